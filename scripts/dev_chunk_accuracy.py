"""Developer check: does cutting the columns into chunks (own step control per chunk) cost accuracy?  cascade40,
row-group (5 chunks) against the per-wave kernel (one error norm over all columns), both against rtol 1e-12."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
n = 40
gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
rng = np.random.default_rng(2026)
P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.3 * rng.standard_normal((56, 2 * n)))
grid = np.linspace(0, 60.0, 1000); t_out = np.concatenate([[0.0], grid[[100, 300, 600, 999]]])
St = m.calc_jacobian_batch(P, t_out, rtol=1e-12, atol=1e-15, variant='per_wave')
def err(S): return np.array([np.max(np.abs(S[v, 1:] - St[v, 1:]) / (1e-8 * np.abs(St[v, 1:]) + 5e-9)) for v in range(len(P))])
for variant in ('row_group', 'per_wave'):
    for rtol in (1e-9, 1e-10):
        S = m.calc_jacobian_batch(P, t_out, variant=variant, rtol=rtol, atol=1e-12)
        e = err(S)
        print("%-9s rtol %.0e: steps/vector %.0f  error vs tight: median %.3f max %.3f" % (variant, rtol, m.last_info['n_steps'].mean(), np.median(e), e.max()))
S12 = m.calc_jacobian_batch(P, t_out, rtol=1e-12, atol=1e-15)
print("row_group at rtol 1e-12 vs per_wave at 1e-12: max %.4f" % err(S12).max())
