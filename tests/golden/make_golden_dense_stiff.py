"""Generate tests/golden/dstiff48_ref.npz: the REAL reference OdeModel on a DENSE stiff network (models_zoo.dense_stiff_spec:
48 states, half of all couplings present, degradation rates over four decades, 48 sensitivity columns = 2352 ODEs), and
dstiff48_tight.npz (LSODA at rtol 1e-12 by column groups, oracle.odeint_oracle.tight_stiff_solution_by_columns).

Run in the build container only (it reads /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dense_stiff.py [n_vectors=2]     (~10 minutes per vector)

As in make_golden_stiff.py the reference's ``OdeModel.simulate`` / ``calc_jacobian`` (model/ode_model.py:83-169: odeint ->
LSODA, rtol = atol = 1e-10, Dfun=None) get right-hand sides with the reference callback contract f(y, t, yout, p) that call
the build's generated C code (the role numba plays in the reference).  Only numbers are stored.
"""
import ctypes
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, 'model'))

import ode_model as ref_ode_model  # noqa: E402  (reference)

from sysbio_modeling_amd import models_zoo  # noqa: E402
from sysbio_modeling_amd.symbolic import GeneratedModel  # noqa: E402
from oracle import odeint_oracle as oo  # noqa: E402


def c_callable(cfn):
    dp = ctypes.POINTER(ctypes.c_double)

    def f(y, t, yout, p):
        yc = np.ascontiguousarray(y, dtype=np.float64)
        pc = np.ascontiguousarray(p, dtype=np.float64)
        cfn(yc.ctypes.data_as(dp), ctypes.c_double(t), yout.ctypes.data_as(dp), pc.ctypes.data_as(dp))
    return f


def main(n_vectors=2):
    gm = GeneratedModel(models_zoo.dense_stiff_spec())
    lib = gm.c_library()
    m = ref_ode_model.OdeModel(c_callable(lib.sbm_rhs), c_callable(lib.sbm_sens_rhs), gm.n_vars, list(gm.param_order),
                               use_jit=False)
    _, P = models_zoo.dense_stiff_ensemble(4096)
    P = P[:n_vectors]
    grid = np.linspace(0, models_zoo.DENSE_STIFF_T_END, 1000)
    idx = np.searchsorted(grid, models_zoo.DENSE_STIFF_MEASURE_TIMES)
    n, k = gm.n_vars, gm.n_sens
    Ys, Ss, Yt, St = [], [], [], []
    for p in P:
        t0 = time.time()
        Ys.append(m.simulate(p, grid)[idx])
        Ss.append(m.calc_jacobian(p, grid, np.zeros(n + n * k))[idx])
        print('reference vector done in %.0f s' % (time.time() - t0), flush=True)
        t0 = time.time()
        y, s_ = oo.tight_stiff_solution_by_columns(gm, p, np.concatenate([[0.0], grid[idx]]), group=8)
        Yt.append(y[1:])
        St.append(s_[1:])
        print('tight solution done in %.0f s' % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, 'dstiff48_ref.npz'), P=P, t=grid, idx=idx, Y=np.stack(Ys), S=np.stack(Ss))
    np.savez_compressed(os.path.join(HERE, 'dstiff48_tight.npz'), P=P, Y=np.stack(Yt), S=np.stack(St))


if __name__ == '__main__':
    main(*[int(x) for x in sys.argv[1:2]])
