"""Developer check: Romberg control loop on stiff50 at default and loose tolerances, with the per-level estimates."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'stiff50_ref.npz'))
P = g['P']; t_out = np.concatenate([[0.0], g['t'][g['idx']]]); Sr = g['S']
for rtol, atol in ((1e-9, 1e-12), (1e-7, 1e-10), (1e-5, 1e-8), (1e-3, 1e-6)):
    m._control_trace = []
    S = m.calc_jacobian_batch(P, t_out, method='implicit_controlled', rtol=rtol, atol=atol)
    err = np.max(np.abs(S[:, 1:] - Sr) / (rtol * np.maximum(np.abs(Sr), 1e-3 * np.abs(Sr).max()) + atol), axis=(1, 2))
    print("rtol %.0e: status %s levels %s steps %s  true error / tolerance %s" % (rtol, m.last_info['status'], m.last_info['levels'], m.last_info['n_steps'], np.round(err, 2)))
    for lv, idx, e in m._control_trace:
        print("   level %d  vectors %s  estimates %s" % (lv, idx, np.array2string(e, precision=3)))
