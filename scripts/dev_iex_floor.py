"""A/B of the relative floor of the error norm (SBM_PLUGIN_FLAGS=-DSBM_IEX_FLOOR=...): stiff50 cost, cascade70 accuracy."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import GeneratedModel, zoo_model
from oracle.tolerances import parity_err
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
print('flags', os.environ.get('SBM_PLUGIN_FLAGS'))
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
g = np.load(os.path.join(here, 'stiff50_ref.npz')); tg = np.load(os.path.join(here, 'stiff50_tight.npz'))
t_out = np.concatenate([[0.0], g['t'][g['idx']]])
for K, rtol in ((8, 3e-9), (7, 3e-9), (6, 3e-9)):
    S, Y = m.calc_jacobian_batch(g['P'], t_out, return_states=True, method='implicit_extrap', rtol=rtol, atol=1e-18, order=K, max_steps=3000)
    print('stiff50 K %d rtol %g atol 1e-18: status %s steps %s | vs tight y %.3f S %.3f' % (K, rtol, m.last_info['status'], m.last_info['n_steps'],
          parity_err(Y[:, 1:], tg['Y']), parity_err(S[:, 1:], tg['S'])), flush=True)
for n in (40, 70):
    gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade%d' % n)
    rng = np.random.default_rng(70)
    P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.2 * rng.standard_normal((2, 2 * n)))
    t_out = np.array([0.0, 10.0, 30.0, 60.0])
    Se, Ye = m.calc_jacobian_batch(P, t_out, return_states=True)
    for K, rtol in ((8, 3e-9), (8, 1e-9), (7, 3e-9), (6, 3e-9)):
        Si, Yi = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_extrap', rtol=rtol, atol=1e-18, order=K)
        print('cascade%d K %d rtol %g: status %s steps %s rej %s | y %.3f S %.3f units' % (
            n, K, rtol, m.last_info['status'], m.last_info['n_steps'], m.last_info['n_rejected'],
            parity_err(Yi[:, 1:], Ye[:, 1:]), parity_err(Si[:, 1:], Se[:, 1:])), flush=True)
