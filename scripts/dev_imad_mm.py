import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
from oracle.tolerances import parity_err
warnings.simplefilter('ignore')
for name, f in (('michaelis_menten', 'mm_ref.npz'), ('simple', 'simple_ref.npz')):
    gm = zoo_model(name)
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
    g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', f))
    for kw in (dict(), dict(rtol=1e-9, atol=1e-12)):
        Y = m.simulate_batch(g['P'], g['t'], method='implicit_controlled', **kw)
        i = dict(m.last_info)
        S, Y2 = m.calc_jacobian_batch(g['P'], g['t'], return_states=True, method='implicit_controlled', **kw)
        j = m.last_info
        print(name, kw, 'state-only: status', i['status'], 'steps', i['n_steps'], 'rej', i['n_rejected'], 'err', parity_err(Y, g['Y']),
              '| sens: status', j['status'], 'steps', j['n_steps'], 'rej', j['n_rejected'], 'err y', parity_err(Y2, g['Y']), 'S', parity_err(S, g['S']), flush=True)
