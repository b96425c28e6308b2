"""Developer check: a model beyond one row / one column per lane (70 states, 140 parameters: 9870 ODEs per
trajectory) runs through the per-wave kernels (three columns per lane, scratch spills) and agrees with odeint."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
from oracle import odeint_oracle as oo
gm = GeneratedModel(models_zoo.cascade_spec(70, name='cascade70'))
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade70')
rng = np.random.default_rng(0)
P = models_zoo.cascade_nominal_params(70)[None, :] * np.exp(0.2 * rng.standard_normal((64, 140)))
t = np.linspace(0, 50, 1000); idx = np.array([0, 400, 999])
Sr = oo.calc_jacobian(gm, P[0], t, use_c=True)[idx]
Yr = oo.simulate(gm, P[0], t, use_c=True)[idx]
for meth, kw in (('dopri45', {}), ('rk4', {'n_steps': 4096})):
    t0 = time.time()
    S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, method=meth, **kw)
    dt = time.time() - t0
    ey = np.max(np.abs(Y[0] - Yr) / (1e-8 * np.abs(Yr) + 5e-9)); es = np.max(np.abs(S[0] - Sr) / (1e-8 * np.abs(Sr) + 5e-9))
    print(meth, "status", m.last_info['status'].max(), "steps", m.last_info['n_steps'][0], "err (tol units) y %.2f s %.2f" % (ey, es), "%.2f s" % dt)
