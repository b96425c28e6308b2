import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model
from oracle.tolerances import parity_err
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
g = np.load(os.path.join(here, 'stiff50_ref.npz')); tg = np.load(os.path.join(here, 'stiff50_tight.npz'))
t_out = np.concatenate([[0.0], g['t'][g['idx']]])
for K, rtol in ((8, 3e-9), (8, 1.9e-9)):
  for atol in (3e-12, 1e-13, 1e-14, 1e-15, 1e-16, 1e-17):
    S, Y = m.calc_jacobian_batch(g['P'], t_out, return_states=True, method='implicit_extrap', rtol=rtol, atol=atol, order=K, max_steps=3000)
    print('stiff50 K %d rtol %g atol %g: status %s steps %s rej %s | vs tight y %.3f S %.3f' % (K, rtol, atol, m.last_info['status'], m.last_info['n_steps'], m.last_info['n_rejected'],
          parity_err(Y[:, 1:], tg['Y']), parity_err(S[:, 1:], tg['S'])), flush=True)
