"""tests/golden/stiff50_wide_ref.npz + stiff50_wide_tight.npz: the WIDE pin of BASELINE configs[4].

stiff50_ref.npz holds the first 3 vectors of the 4096-vector stiff50 ensemble.  This script pins 32 more, spread
evenly over the ensemble (indices ``WIDE_INDEX``), through the REAL reference ``OdeModel.simulate`` /
``calc_jacobian`` (model/ode_model.py:83-169: odeint -> LSODA, rtol = atol = 1e-10, Dfun=None), exactly as
make_golden_stiff.py does, and -- for every fourth of them -- the build's tight solution (oracle odeint at
rtol 1e-12 / atol 1e-15) that arbitrates where LSODA(1e-10) is the one that is off.

Run in the build container only (it reads /root/reference); ~40 s per reference vector, ~3 min per tight one:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stiff_wide.py [workers]

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
WIDE_INDEX = np.linspace(3, 4095, N_WIDE).astype(int)
TIGHT_EVERY = 4


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
    from scipy.integrate import odeint
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd.symbolic import zoo_model
    gm = zoo_model('stiff50')
    P, grid, idx = _inputs()
    N = gm.n_vars * (1 + gm.n_sens)
    fw = oo._wrap_c(gm.c_library().sbm_sens_rhs, N, P[j])
    t = np.concatenate([[0.0], grid[idx]])
    sol, info = odeint(fw, np.zeros(N), t, rtol=1e-12, atol=1e-15, mxstep=200000, full_output=True)
    print('tight vector', j, 'done, steps', info['nst'][-1], flush=True)
    return sol[1:]


def main(workers=5):
    sys.path.insert(0, HERE)
    from sysbio_modeling_amd.symbolic import zoo_model
    zoo_model('stiff50').c_library()
    P, grid, idx = _inputs()
    tight_rows = list(range(0, N_WIDE, TIGHT_EVERY))
    with mp.get_context('spawn').Pool(workers) as pool:
        tight_job = pool.map_async(tight_vector, tight_rows)
        ref = pool.map(reference_vector, range(N_WIDE), chunksize=1)
        tight = np.stack(tight_job.get())
    Y = np.stack([r[0] for r in ref])
    S = np.stack([r[1] for r in ref])
    np.savez_compressed(os.path.join(HERE, 'stiff50_wide_ref.npz'), P=P, index=WIDE_INDEX, t=grid, idx=idx, Y=Y, S=S)
    np.savez_compressed(os.path.join(HERE, 'stiff50_wide_tight.npz'), P=P[tight_rows], rows=np.array(tight_rows),
                        index=WIDE_INDEX[tight_rows], t=grid, idx=idx, Y=tight[:, :, :50], S=tight[:, :, 50:])
    from oracle.tolerances import parity_err
    print('LSODA(1e-10) golden vs tight: y %.2f S %.2f tolerance units' % (
        parity_err(Y[tight_rows], tight[:, :, :50]), parity_err(S[tight_rows], tight[:, :, 50:])))


if __name__ == '__main__':
    sys.path.insert(0, HERE)
    main(*[int(x) for x in sys.argv[1:2]])
