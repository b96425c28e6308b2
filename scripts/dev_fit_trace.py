"""fit_batch on the configs[3] project with trace=True: per iteration the time of the trial launch, the slowest / mean / 99th
percentile trajectory of it, trial points that failed, starts finished."""
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
method = sys.argv[1] if len(sys.argv) > 1 else 'dopri45'
proj.fit_batch(starts[:8], max_iter=3, method=method)
fit = proj.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8, method=method, trace=True)
tot = 0.0
for h in fit['history']:
    L = h['launch']
    tot += L['ms']
    print('it %3d  %6.2f ms  max %5d  p99 %6.0f  mean %6.0f  failed %3d  done %3d  accepted %3d' % (
        h['iteration'], L['ms'], L['steps_max'], L['steps_p99'], L['steps_mean'], L['failed'], L['done'], h['accepted']))
print('launch total %.1f ms' % tot)
for meth in ('dopri45', 'dop853'):
    for lazy in (False, True):
        proj.fit_batch(starts[:8], max_iter=2, method=meth, lazy_jacobian=lazy)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f = proj.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8, method=meth, lazy_jacobian=lazy)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print('%s lazy %s: %.3f s  cost min %.3f median %.3f max %.3f converged %d  evals %d jac %d' % (
            meth, lazy, dt, f['cost'].min(), np.median(f['cost']), f['cost'].max(), f['converged'].sum(),
            f['n_evaluations'], f['n_jacobian_evaluations']), flush=True)
