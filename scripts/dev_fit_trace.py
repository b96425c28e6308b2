"""Iteration-by-iteration trace of Project.fit_batch on the configs[3] project (GPU box): accepted steps, damping,
cost decrease -- what the convergence tests of levenberg_marquardt_batch see."""
import sys
import warnings

import numpy as np

sys.path.insert(0, '.')
from sysbio_modeling_amd import models_zoo                      # noqa: E402
from sysbio_modeling_amd.model import OdeModel                  # noqa: E402
from sysbio_modeling_amd.symbolic import zoo_model              # noqa: E402

gm = zoo_model('cascade20')
model = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade20')
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    proj, th0 = models_zoo.cascade_config4_project(model, reference_compat=False)
starts = th0[None, :] + 0.15 * np.random.default_rng(1).standard_normal((64, th0.size))
kw = dict(max_iter=int(sys.argv[1]) if len(sys.argv) > 1 else 100)
fit = proj.fit_batch(starts, trace=True, **kw)
for h in fit['history']:
    if h['iteration'] < 12 or h['iteration'] % 8 == 0:
        print(h)
print('converged', int(fit['converged'].sum()), 'of', len(starts), 'n_iter', np.bincount(fit['n_iter'])[-5:],
      'cost min/median/max', fit['cost'].min(), np.median(fit['cost']), fit['cost'].max())
J = proj.evaluate_batch(fit['theta'][:1], jacobian=True, want=('jacobian',))
Jm = J['jacobian'][0]
r = J['residuals'][0]
g = Jm.T @ r
sv = np.linalg.svd(Jm, compute_uv=False)
print('gradient norm at the end', np.linalg.norm(g), 'cost', 0.5 * r @ r, 'singular values of J: max %.3g min %.3g' % (sv[0], sv[-1]))

# Marquardt up/down damping against the lmder trust region: cost reached and time, 256 starts x 100 iterations
import time
import torch
starts256 = th0[None, :] + 0.15 * np.random.default_rng(1).standard_normal((256, th0.size))
for label, kw in (('trust_region, budget 2000', dict(algorithm='trust_region', max_steps=-2000)),
                  ('trust_region, budget 3000', dict(algorithm='trust_region', max_steps=-3000)),
                  ('marquardt', dict(algorithm='marquardt')),
                  ('trust_region, clip 2, factor 100', dict(algorithm='trust_region')),
                  ('trust_region, clip 20, factor 100', dict(algorithm='trust_region', max_step=20.0)),
                  ('trust_region, clip 2, factor 1', dict(algorithm='trust_region', factor=1.0)),
                  ('trust_region, clip 2, factor 0.1', dict(algorithm='trust_region', factor=0.1)),
                  ('trust_region, clip 1, factor 1', dict(algorithm='trust_region', factor=1.0, max_step=1.0)),
                  ('trust_region, clip 2, factor 1, budget 5000', dict(algorithm='trust_region', factor=1.0, max_steps=-5000))):
    proj.fit_batch(starts256[:8], max_iter=2, **kw)
    for n_it in (30, 100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = proj.fit_batch(starts256, max_iter=n_it, **kw)
        torch.cuda.synchronize()
        print("%-44s %3d iterations %.2f s: cost min %.3f median %.3f max %.3f, converged %d; start #0: %.3f"
              % (label, n_it, time.perf_counter() - t0, f['cost'].min(), np.median(f['cost']), f['cost'].max(),
                 int(f['converged'].sum()), f['cost'][0]), flush=True)
sys.exit(0)
# eager against lazy Jacobians at several batch sizes
import time
import torch
for n_starts in (256, 1024, 4096):
    st = th0[None, :] + 0.15 * np.random.default_rng(2).standard_normal((n_starts, th0.size))
    for lazy in (False, True):
        proj.fit_batch(st[:8], max_iter=2, lazy_jacobian=lazy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f = proj.fit_batch(st, max_iter=30, lazy_jacobian=lazy)
        torch.cuda.synchronize()
        print("%5d starts, 30 iterations, lazy_jacobian=%s: %.3f s, %d evaluations, %d with sensitivities, median cost %.4f"
              % (n_starts, lazy, time.perf_counter() - t0, f['n_evaluations'], f['n_jacobian_evaluations'], np.median(f['cost'])))
