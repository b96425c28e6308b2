import os, sys, warnings, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd import models_zoo
warnings.simplefilter('ignore')
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
t0 = time.time()
proj, th0 = models_zoo.cascade_config4_project(m, noise=0.0, reference_compat=False)
print("build", time.time() - t0)
starts = th0[None, :] + 0.15 * np.random.default_rng(1).standard_normal((32, th0.size))
c0 = proj.calc_sum_square_residuals_batch(starts)
for it in (10, 20, 40):
    t0 = time.time()
    fit = proj.fit_batch(starts, max_iter=it)
    print("iters", it, "time %.2f s" % (time.time() - t0), "cost0 median %.3e" % np.median(c0), "cost median %.3e max %.3e min %.3e" % (np.median(fit['cost']), fit['cost'].max(), fit['cost'].min()), "converged", fit['converged'].sum(), flush=True)
