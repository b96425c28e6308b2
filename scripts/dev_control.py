"""Developer check: convergence of the extrapolated implicit midpoint results on stiff50 with the step count."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'stiff50_ref.npz'))
P = g['P']; t_out = np.concatenate([[0.0], g['t'][g['idx']]])
def sc(x):
    return 1e-9 * np.maximum(np.abs(x), 1e-3 * np.abs(x).reshape(3, -1).max(axis=1)[:, None, None]) + 1e-12
kw = dict(method='implicit_midpoint_graded', rtol=1e-11, atol=1e-14)
E = {}
for n in (256, 512, 1024, 2048, 4096, 8192, 16384):
    E[n] = m.calc_jacobian_batch(P, t_out, n_steps=n, extrapolate=1, **kw)
fine2 = m.calc_jacobian_batch(P, t_out, n_steps=8192, extrapolate=2, **kw)
fine = E[16384]
for n in sorted(E):
    e1 = np.max(np.abs(E[n] - fine) / sc(fine), axis=(1, 2))
    e2 = np.max(np.abs(E[n] - fine2) / sc(fine2), axis=(1, 2))
    d = np.max(np.abs(E[n] - E[n // 2]) / sc(E[n]), axis=(1, 2)) if n // 2 in E else np.full(3, np.nan)
    print("n=%6d  err vs E(16384) %s   vs T2(8192) %s   |E(n)-E(n/2)| %s" % (n, np.round(e1, 2), np.round(e2, 2), np.round(d, 2)))
print("golden vs fine:", np.max(np.abs(g['S'] - fine[:, 1:]) / (1e-8 * np.abs(g['S']) + 5e-9)), " vs T2:", np.max(np.abs(g['S'] - fine2[:, 1:]) / (1e-8 * np.abs(g['S']) + 5e-9)))
S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled')
print("controlled: levels", m.last_info['levels'], "status", m.last_info['status'], "steps", m.last_info['n_steps'])
print("  S err vs E(16384)", np.round(np.max(np.abs(S - fine) / sc(fine), axis=(1, 2)), 2), " vs T2", np.round(np.max(np.abs(S - fine2) / sc(fine2), axis=(1, 2)), 2))
E32 = m.calc_jacobian_batch(P, t_out, n_steps=32768, extrapolate=1, **kw)
print("  E(32768) vs E(16384)", np.round(np.max(np.abs(E32 - fine) / sc(fine), axis=(1, 2)), 2), " vs T2", np.round(np.max(np.abs(E32 - fine2) / sc(fine2), axis=(1, 2)), 2))
print("  S vs E32", np.round(np.max(np.abs(S - E32) / sc(fine), axis=(1, 2)), 2))
