"""fit_batch on the configs[3] project (256 starts, 100 iterations): time and costs by integrator and trial budget."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    proj, th0 = models_zoo.cascade_config4_project(m, reference_compat=False)
starts = th0[None, :] + 0.15 * np.random.default_rng(1).standard_normal((256, th0.size))
proj.fit_batch(starts[:8], max_iter=3)
for method, budgets in (('dopri45', (None, -2600, -6500)), ('dop853', (None, -600, -900, -1200, -2000))):
    for b in budgets:
        kw = {} if b is None else {'max_steps': b}
        proj.fit_batch(starts[:8], max_iter=2, method=method, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fit = proj.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8, method=method, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('%s budget %s: %.3f s  cost min %.3f median %.3f max %.3f converged %d' % (method, b, dt, fit['cost'].min(), np.median(fit['cost']), fit['cost'].max(), fit['converged'].sum()), flush=True)
