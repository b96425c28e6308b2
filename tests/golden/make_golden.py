"""Generate tests/golden/*.npz by running the REAL reference implementation.

Run in the build container only (it reads /root/reference, which does not exist
on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is imported from the reference:
  * model/ode_model.py ``OdeModel`` (+ abstract_model.py) as top-level modules,
    constructed with use_jit=False (numba is absent; its import is lazy, :48);
  * tests/test_utils/{jittable,sens_jittable}{,_mm}_model.py -- the reference's
    own fixture right-hand sides;
  * project/utils.py as a plain file (the four sampling helpers, :10-89).
Only NUMBERS are stored: inputs and the reference's outputs.  No reference
source or bytecode enters the repository.
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, 'model'))
sys.path.insert(0, os.path.join(REF, 'tests', 'test_utils'))

import ode_model as ref_ode_model  # noqa: E402  (reference)
import jittable_model as ref_simple  # noqa: E402
import sens_jittable_model as ref_simple_sens  # noqa: E402
import jittable_mm_model as ref_mm  # noqa: E402
import sens_jittable_mm_model as ref_mm_sens  # noqa: E402

from sysbio_modeling_amd.symbolic import zoo_model  # noqa: E402
from sysbio_modeling_amd import models_zoo  # noqa: E402


def load_ref_utils():
    spec = importlib.util.spec_from_file_location('ref_project_utils', os.path.join(REF, 'project', 'utils.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_model(model, sens_model, n_vars, order):
    return ref_ode_model.OdeModel(model, sens_model, n_vars, order, use_jit=False)


def rhs_samples(fn, n_out, ys, ps):
    out = np.zeros((len(ys), n_out))
    for i, (y, p) in enumerate(zip(ys, ps)):
        yout = np.zeros(n_out)
        fn(np.asarray(y, dtype=float), 0.0, yout, np.asarray(p, dtype=float))
        out[i] = yout
    return out


def main():
    rng = np.random.default_rng(12345)
    t1000 = np.linspace(0, 100, 1000)

    # ---- (i) 1-state model: reference fixture RHS through the reference OdeModel ----
    m = ref_model(ref_simple.model, ref_simple_sens.sens_model, 1, ['k_deg', 'k_synt'])
    P = np.array([[0.001, 0.01], [0.01, 0.01]])
    Y = np.stack([m.simulate(p, t1000) for p in P])
    S = np.stack([m.calc_jacobian(p, t1000, np.zeros(3)) for p in P])
    t10 = np.linspace(0, 100, 10)
    Y10 = m.simulate(P[0], t10)
    S10 = m.calc_jacobian(P[0], t10, np.zeros(3))
    # RHS point evaluations: pin the build's generated callables to the reference fixture's
    ys = rng.uniform(0.0, 2.0, size=(8, 3))
    ps = rng.uniform(0.001, 0.1, size=(8, 2))
    np.savez(os.path.join(HERE, 'simple_ref.npz'), P=P, t=t1000, Y=Y, S=S, t10=t10, Y10=Y10, S10=S10,
             rhs_y=ys, rhs_p=ps, rhs_state=rhs_samples(ref_simple.model, 1, ys[:, :1], ps),
             rhs_sens=rhs_samples(ref_simple_sens.sens_model, 3, ys, ps))

    # ---- (ii) Michaelis-Menten at the parameters of tests/test_Project.py:284-289 ----
    m = ref_model(ref_mm.model, ref_mm_sens.sens_model, 2, list(ref_mm.ordered_params))
    p_mm = np.array([[1e-3, 0.001, 0.01, 0.01, 0.001], [3e-4, 0.005, 0.001, 0.01, 0.01]])
    Y = np.stack([m.simulate(p, t1000) for p in p_mm])
    S = np.stack([m.calc_jacobian(p, t1000, np.zeros(12)) for p in p_mm])
    ys = rng.uniform(0.0, 1.0, size=(8, 12))
    ps = rng.uniform(0.001, 0.1, size=(8, 5))
    np.savez(os.path.join(HERE, 'mm_ref.npz'), P=p_mm, t=t1000, Y=Y, S=S, rhs_y=ys, rhs_p=ps,
             rhs_state=rhs_samples(ref_mm.model, 2, ys[:, :2], ps),
             rhs_sens=rhs_samples(ref_mm_sens.sens_model, 12, ys, ps))

    # ---- (iii) cascade20: the build's generated Python RHS driven by the reference OdeModel ----
    gm = zoo_model('cascade20')
    m = ref_model(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
    _, P = models_zoo.cascade_ensemble(4096)
    P = P[:4]
    idx = np.searchsorted(t1000, models_zoo.CASCADE_MEASURE_TIMES)
    Ys, Ss = [], []
    for p in P:
        Ys.append(m.simulate(p, t1000)[idx])
        Ss.append(m.calc_jacobian(p, t1000, np.zeros(20 + 800))[idx])
        print('cascade vector done', flush=True)
    np.savez(os.path.join(HERE, 'cascade20_ref.npz'), P=P, t=t1000, idx=idx, Y=np.stack(Ys), S=np.stack(Ss))

    # ---- (iv) the reference's sampling helpers on a known trajectory ----
    U = load_ref_utils()

    class M(object):
        def __init__(self, tp):
            self.tp = np.asarray(tp, dtype=float)

        def get_nonzero_measurements(self):
            keep = self.tp != 0
            return None, None, self.tp[keep]

    class Ex(object):
        param_global_vector_idx = dict(a=0, b=1, c=2)

    t_meas = np.array([0.0, 11.11111111, 50.0, 50.05005005005005, 99.99, 100.0])
    sim = np.stack([np.sin(0.1 * t1000), np.cos(0.07 * t1000), t1000 ** 0.5], axis=1)
    jac = rng.standard_normal((1000, 3 * 3))
    d_sim, d_t = U.direct_model_var_to_measure(sim, t1000, Ex(), M(t_meas), 1)
    s_sim, s_t = U.sum_model_vars_to_measure(sim, t1000, Ex(), M(t_meas), [0, 2])
    d_jac = U.direct_model_jac_to_measure_jac(jac, t1000, Ex(), M(t_meas), 2)
    s_jac = U.sum_model_jac_to_measure_jac(jac, t1000, Ex(), M(t_meas), [0, 1])
    np.savez(os.path.join(HERE, 'sampling_ref.npz'), t=t1000, t_meas=t_meas, sim=sim, jac=jac,
             direct_sim=d_sim, direct_t=d_t, sum_sim=s_sim, sum_t=s_t, direct_jac=d_jac, sum_jac=s_jac)
    print('golden files written to', HERE)


if __name__ == '__main__':
    main()
