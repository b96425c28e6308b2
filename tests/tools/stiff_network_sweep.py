"""Sweep: random rate-law networks made STIFF (degradation rates of a third of the species x 1e4) -- irregular Jacobian
patterns: the redundant and the row-distributed LU of the implicit stepper, not the chain prefix -- integrated with
method='auto' (DOPRI45 with a budget, then the stiff integrator SBM_IMPLICIT_EXTRAP) and with the stiff integrator
directly, against odeint with the generated analytic Jacobian (LSODA switches to BDF).  One line per network; exits
non-zero when a result is off by more than 1.5 tolerance units.  Records of the last runs: profiles/r03/stiff_network_sweep.txt, stiff_tri_network_sweep.txt

    python tests/tools/stiff_network_sweep.py [n_networks] [tri]     (GPU box; compiles one plugin per network)

``tri``: lower-triangular networks instead (tests/test_generated_header_host.py::_triangular_network: several sub-diagonal
entries and several J_p entries per row) -- the fused substitution forms of emit_implicit.py (im_solve_tri_pick for the
Newton update, im_sens_tri for the sensitivity step of SBM_IMPLICIT_EXTRAP) on patterns the zoo does not have."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter('ignore')
from tests.test_gpu_user_models import _random_network
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
from oracle import odeint_oracle as oo

n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 8
triangular = len(sys.argv) > 2 and sys.argv[2] == 'tri'
if triangular:
    from tests.test_generated_header_host import _triangular_network
rng0 = np.random.default_rng(777)
worst = 0.0
for k in range(n_models):
    seed, n = int(rng0.integers(10, 10000)), int(rng0.integers(4, 16))
    if triangular:
        n += 8
    gm = GeneratedModel(_triangular_network(seed, n) if triangular else _random_network(seed, n))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    rng = np.random.default_rng(seed)
    names = list(gm.param_order)
    P = np.exp(rng.uniform(np.log(0.2), np.log(2.0), (3, len(names))))
    fast = [i for i in range(n) if i % 3 == 1]
    for i in fast:                                     # fast species: production and degradation x 1e4
        for nm in ('d%d' % i, 'k%d' % i, 'k%d_0' % i, 'k%d_1' % i):
            if nm in names:
                P[:, names.index(nm)] *= 1e4
    t = np.linspace(0, 20.0, 1000); idx = np.array([0, 333, 999])
    t0 = time.time()
    Yr = oo.simulate(gm, P[1], t, use_c=True, model_jac=gm.model_jac)[idx]
    Sr = oo.calc_jacobian(gm, P[1], t, use_c=True, sens_model_jac=gm.sens_model_jac)[idx]
    t_ref = time.time() - t0
    t0 = time.time()
    S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, method='auto', max_steps=20000)
    dt = time.time() - t0
    info = m.last_info
    S2, Y2 = m.calc_jacobian_batch(P, t[idx], return_states=True, method='implicit_controlled')
    i2 = m.last_info
    e2 = max(np.max(np.abs(Y2[1] - Yr) / (1e-8 * np.abs(Yr) + 5e-9)), np.max(np.abs(S2[1] - Sr) / (1e-8 * np.abs(Sr) + 5e-9)))
    ey = np.max(np.abs(Y[1] - Yr) / (1e-8 * np.abs(Yr) + 5e-9)); es = np.max(np.abs(S[1] - Sr) / (1e-8 * np.abs(Sr) + 5e-9))
    print("%-14s n=%2d k=%2d fast %d  status %s stiff %s steps %s  err (tol units) y %.2f s %.2f  gpu %.2f s lsoda %.1f s"
          % (gm.spec.name, n, gm.n_sens, len(fast), info['status'].tolist(), info['stiff'].astype(int).tolist(),
             info['n_steps'].tolist(), ey, es, dt, t_ref)
          + "  | stiff integrator on all 3: status %s macro steps %s err %.2f" % (i2['status'].tolist(), i2['n_steps'].tolist(), e2),
          flush=True)
    worst = max(worst, ey, es, e2)
    if info['status'].any() or i2['status'].any():
        worst = max(worst, 99.0)
print("worst", worst)
sys.exit(0 if worst <= 1.5 else 1)
