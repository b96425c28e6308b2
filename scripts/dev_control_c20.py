"""Developer check: estimates per level of implicit_controlled on the cascade20 / MM goldens."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
def pe(a, r): return np.max(np.abs(a - r) / (1e-8 * np.abs(r) + 5e-9))
for name, file in (('cascade20', 'cascade20_ref.npz'), ('michaelis_menten', 'mm_ref.npz')):
    gm = zoo_model(name)
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=name)
    g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', file))
    t = np.concatenate([[0.0], g['t'][g['idx']]]) if 'idx' in g else g['t']
    sl = slice(1, None) if 'idx' in g else slice(None)
    for what in ('sim', 'sens'):
        m._control_trace = []
        if what == 'sim':
            Y = m.simulate_batch(g['P'], t, method='implicit_controlled'); err = pe(Y[:, sl], g['Y'])
        else:
            S = m.calc_jacobian_batch(g['P'], t, method='implicit_controlled'); err = pe(S[:, sl], g['S'])
        print(name, what, "status", m.last_info['status'], "levels", m.last_info['levels'], "parity %.3g" % err)
        for lv, idx, e in m._control_trace:
            print("   level %d  vectors %s  estimates %s" % (lv, idx, np.round(e, 2)))
