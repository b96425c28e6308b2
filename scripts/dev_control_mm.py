"""Developer check: implicit_controlled on the Michaelis-Menten golden (full 1000-point grid)."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('michaelis_menten')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='michaelis_menten')
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'mm_ref.npz'))
P, t = g['P'], g['t']
def pe(a, r): return np.max(np.abs(a - r) / (1e-8 * np.abs(r) + 5e-9))
Y = m.simulate_batch(P, t, method='implicit_controlled')
print("controlled: status", m.last_info['status'], "levels", m.last_info['levels'], "steps", m.last_info['n_steps'], "parity", pe(Y, g['Y']))
for n in (256, 1024, 4096, 16384, 65536):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Yn = m.simulate_batch(P, t, method='implicit_midpoint_graded', n_steps=n, extrapolate=1, rtol=1e-11, atol=1e-14)
        Yp = m.simulate_batch(P, t, method='implicit_midpoint', n_steps=n, extrapolate=1, rtol=1e-11, atol=1e-14)
    print("n=%6d graded: status %s steps %s parity %.3g   plain: status %s parity %.3g" % (n, m.last_info['status'], m.last_info['n_steps'], pe(Yn, g['Y']), m.last_info['status'], pe(Yp, g['Y'])))
Yd = m.simulate_batch(P, t)
print("dopri parity", pe(Yd, g['Y']))
