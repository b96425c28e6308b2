"""tests/golden/stiff50_wide_ref.npz + stiff50_wide_tight.npz: the WIDE pin of BASELINE configs[4].

stiff50_ref.npz holds the first 3 vectors of the 4096-vector stiff50 ensemble.  This script pins 32 more, spread
evenly over the ensemble (indices ``WIDE_INDEX``), through the REAL reference ``OdeModel.simulate`` /
``calc_jacobian`` (model/ode_model.py:83-169: odeint -> LSODA, rtol = atol = 1e-10, Dfun=None), exactly as
make_golden_stiff.py does, and -- for every one of them -- the build's tight solution (oracle odeint at
rtol 1e-12 / atol 1e-15, by column groups) that arbitrates where LSODA(1e-10) is the one that is off.

Run in the build container only (it reads /root/reference); ~80 s per reference vector, ~25 s per tight one:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stiff_wide.py [workers]      (4 workers x 2 BLAS threads)

Only numbers are stored: P (the 32 parameter vectors), index (their rows in the ensemble), the grid, the 16 sampled
rows of Y and S.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

N_WIDE = 32
TMP = os.path.join(HERE, '_wide_tmp')
WIDE_INDEX = np.linspace(3, 4095, N_WIDE).astype(int)
TIGHT_EVERY = 1          # a tight solution for EVERY vector (cheap since they are integrated by column groups)


def _inputs():
    from sysbio_modeling_amd import models_zoo
    _, P = models_zoo.stiff_ensemble(4096, n=50)
    grid = np.linspace(0, models_zoo.STIFF_T_END, 1000)
    idx = np.searchsorted(grid, models_zoo.STIFF_MEASURE_TIMES)
    return P[WIDE_INDEX], grid, idx


def reference_vector(j):
    sys.path.insert(0, os.path.join(REF, 'model'))
    import ode_model as ref_ode_model  # the reference
    from make_golden_stiff import c_callable
    from sysbio_modeling_amd.symbolic import zoo_model
    gm = zoo_model('stiff50')
    lib = gm.c_library()
    m = ref_ode_model.OdeModel(c_callable(lib.sbm_rhs), c_callable(lib.sbm_sens_rhs), gm.n_vars,
                               list(gm.param_order), use_jit=False)
    P, grid, idx = _inputs()
    n, k = gm.n_vars, gm.n_sens
    Y = m.simulate(P[j], grid)[idx]
    S = m.calc_jacobian(P[j], grid, np.zeros(n + n * k))[idx]
    print('reference vector', j, 'done', flush=True)
    return Y, S


def tight_vector(j):
    """the build's tight solution of vector j: the oracle's odeint call at rtol 1e-12 / atol 1e-15 on the augmented system
    integrated in groups of 10 sensitivity columns (odeint_oracle.tight_stiff_solution_by_columns: the columns couple
    only through the state; 25 s per vector where the full 2550-equation system takes 15 - 45 minutes, equal to it to
    1e-4 tolerance units on the vectors of stiff50_tight.npz)"""
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd.symbolic import zoo_model
    gm = zoo_model('stiff50')
    P, grid, idx = _inputs()
    Y, S = oo.tight_stiff_solution_by_columns(gm, P[j], np.concatenate([[0.0], grid[idx]]))
    print('tight vector', j, 'done', flush=True)
    return np.concatenate([Y[1:], S[1:]], axis=1)


def _cached(kind, j, fn):
    """every finished vector is written to its own file under _wide_tmp/ at once (a run that is interrupted loses nothing)"""
    os.makedirs(TMP, exist_ok=True)
    path = os.path.join(TMP, '%s_%02d.npz' % (kind, j))
    if os.path.exists(path):
        with np.load(path) as f:
            return f['Y'], f['S']
    Y, S = fn(j)
    np.savez(path + '.tmp.npz', Y=Y, S=S)
    os.replace(path + '.tmp.npz', path)
    return Y, S


def _ref_job(j):
    return _cached('ref', j, reference_vector)


def _tight_job(j):
    def run(jj):
        sol = tight_vector(jj)
        return sol[:, :50], sol[:, 50:]
    return _cached('tight', j, run)


def main(workers=4):
    sys.path.insert(0, HERE)
    from sysbio_modeling_amd.symbolic import zoo_model
    zoo_model('stiff50').c_library()
    P, grid, idx = _inputs()
    tight_rows = list(range(0, N_WIDE, TIGHT_EVERY))
    with mp.get_context('spawn').Pool(workers) as pool:
        ref = pool.map(_ref_job, range(N_WIDE), chunksize=1)          # the reference's vectors first
        Y = np.stack([r[0] for r in ref])
        S = np.stack([r[1] for r in ref])
        np.savez_compressed(os.path.join(HERE, 'stiff50_wide_ref.npz'), P=P, index=WIDE_INDEX, t=grid, idx=idx, Y=Y, S=S)
        print('stiff50_wide_ref.npz written', flush=True)
        tight = pool.map(_tight_job, tight_rows, chunksize=1)
    tY = np.stack([r[0] for r in tight])
    tS = np.stack([r[1] for r in tight])
    np.savez_compressed(os.path.join(HERE, 'stiff50_wide_tight.npz'), P=P[tight_rows], rows=np.array(tight_rows),
                        index=WIDE_INDEX[tight_rows], t=grid, idx=idx, Y=tY, S=tS)
    from oracle.tolerances import parity_err
    print('LSODA(1e-10) golden vs tight: y %.2f S %.2f tolerance units' % (
        parity_err(Y[tight_rows], tY), parity_err(S[tight_rows], tS)))


if __name__ == '__main__':
    # LSODA factors dense 2550 x 2550 Jacobians through LAPACK: one process with all BLAS threads, or several with few --
    # several with all of them oversubscribe the cores several times over (measured: 20 minutes per vector instead of 1)
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '2')
    os.environ.setdefault('OMP_NUM_THREADS', '2')
    sys.path.insert(0, HERE)
    main(*[int(x) for x in sys.argv[1:2]])
