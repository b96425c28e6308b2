"""Developer check: models beyond one column per lane.  usage: dev_big.py [n_states] [n_vectors]
cascade(n): n states, 2n parameters.  n <= 64: chunked row-group kernel (AUTO) against the per-wave kernel;
n > 64: per-wave kernels only (several columns per lane, scratch spills).  Errors against odeint in tolerance units."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
from oracle import odeint_oracle as oo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 70
V = int(sys.argv[2]) if len(sys.argv) > 2 else 64
gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade%d' % n)
rng = np.random.default_rng(0)
P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.2 * rng.standard_normal((V, 2 * n)))
t = np.linspace(0, 50, 1000); idx = np.array([0, 400, 999])
Sr = oo.calc_jacobian(gm, P[0], t, use_c=True)[idx]
Yr = oo.simulate(gm, P[0], t, use_c=True)[idx]
m.calc_jacobian_batch(P[:2], t[idx])     # load / warm up
cases = [('dopri45', 'auto', {}), ('dopri45', 'per_wave', {}), ('rk4', 'auto', {'n_steps': 4096})]
if n <= 64:  # implicit kernel: one state row per lane
    cases.append(('implicit_midpoint', 'auto', {'n_steps': 4096, 'extrapolate': 1, 'rtol': 1e-11, 'atol': 1e-13}))
for meth, variant, kw in cases:
    t0 = time.time()
    S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, method=meth, variant=variant, **kw)
    dt = time.time() - t0
    ey = np.max(np.abs(Y[0] - Yr) / (1e-8 * np.abs(Yr) + 5e-9)); es = np.max(np.abs(S[0] - Sr) / (1e-8 * np.abs(Sr) + 5e-9))
    ns = int(m.last_info['n_steps'].sum())
    print("n=%d V=%d %-18s %-9s status %d steps/traj %d err (tol units) y %.2f s %.2f  %.3f s  %.3g steps/s" %
          (n, V, meth, variant, m.last_info['status'].max(), m.last_info['n_steps'][0], ey, es, dt, ns / dt), flush=True)
