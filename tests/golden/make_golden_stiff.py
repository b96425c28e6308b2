"""Generate tests/golden/stiff50_ref.npz (BASELINE configs[4]) and stiff80_ref.npz (the same cascade with 80 states:
more state variables than a wavefront has lanes) with the REAL reference OdeModel.

Run in the build container only (it reads /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stiff.py            (~2 minutes)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stiff.py 80 2       (80 states, 2 vectors: ~40 minutes)

The reference's ``OdeModel.simulate`` / ``calc_jacobian`` (model/ode_model.py:83-169: odeint -> LSODA,
rtol = atol = 1e-10, Dfun=None) integrate the build's stiff50 model; the right-hand sides handed to it
follow the reference callback contract f(y, t, yout, p) and call the build's generated C code
(the role numba plays in the reference, :50-51; the generated Python callables give the same numbers
but need hours here: LSODA differences a dense 2550 x 2550 Jacobian).  Only numbers are stored.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, 'model'))

import ode_model as ref_ode_model  # noqa: E402  (reference)

from sysbio_modeling_amd.symbolic import zoo_model  # noqa: E402
from sysbio_modeling_amd import models_zoo  # noqa: E402


def c_callable(cfn):
    dp = ctypes.POINTER(ctypes.c_double)

    def f(y, t, yout, p):
        yc = np.ascontiguousarray(y, dtype=np.float64)
        pc = np.ascontiguousarray(p, dtype=np.float64)
        cfn(yc.ctypes.data_as(dp), ctypes.c_double(t), yout.ctypes.data_as(dp), pc.ctypes.data_as(dp))
    return f


def main(n_states=50, n_vectors=3):
    from sysbio_modeling_amd.symbolic import GeneratedModel
    name = 'stiff%d' % n_states
    gm = zoo_model('stiff50') if n_states == 50 else GeneratedModel(models_zoo.stiff_spec(n_states, name=name))
    lib = gm.c_library()
    m = ref_ode_model.OdeModel(c_callable(lib.sbm_rhs), c_callable(lib.sbm_sens_rhs), gm.n_vars,
                               list(gm.param_order), use_jit=False)
    _, P = models_zoo.stiff_ensemble(4096, n=n_states)
    P = P[:n_vectors]
    grid = np.linspace(0, models_zoo.STIFF_T_END, 1000)
    idx = np.searchsorted(grid, models_zoo.STIFF_MEASURE_TIMES)
    n, k = gm.n_vars, gm.n_sens
    Ys, Ss = [], []
    for p in P:
        Ys.append(m.simulate(p, grid)[idx])
        Ss.append(m.calc_jacobian(p, grid, np.zeros(n + n * k))[idx])
        print(name, 'vector done', flush=True)
    np.savez_compressed(os.path.join(HERE, name + '_ref.npz'), P=P, t=grid, idx=idx, Y=np.stack(Ys), S=np.stack(Ss))


if __name__ == '__main__':
    main(*[int(x) for x in sys.argv[1:3]])
