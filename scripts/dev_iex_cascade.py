"""Developer check of SBM_IMPLICIT_EXTRAP on non-stiff cascades (n states) against DOPRI45 at tight tolerance.
usage: python scripts/dev_iex_cascade.py 20,40,70"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import GeneratedModel
from oracle.tolerances import parity_err

for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '20,70').split(',')]:
    gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade%d' % n)
    rng = np.random.default_rng(70)
    P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.2 * rng.standard_normal((2, 2 * n)))
    t_out = np.array([0.0, 10.0, 30.0, 60.0])
    Se, Ye = m.calc_jacobian_batch(P, t_out, return_states=True)
    print('n', n, 'dopri45 steps', m.last_info['n_steps'])
    for K, rtol, atol in ((8, 3e-9, 3e-12), (8, 3e-9, 1e-15), (8, 3e-9, 1e-18), (8, 1e-9, 1e-18), (6, 3e-9, 1e-18), (8, 3e-9, 1e-30)):
        if True:
            Si, Yi = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_extrap', rtol=rtol, atol=atol, order=K)
            print('  K %d rtol %g atol %g: status %s steps %s rej %s | y %.3f S %.3f units' % (
                K, rtol, atol, m.last_info['status'], m.last_info['n_steps'], m.last_info['n_rejected'],
                parity_err(Yi[:, 1:], Ye[:, 1:]), parity_err(Si[:, 1:], Se[:, 1:])), flush=True)
    Sa, Ya = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_adaptive')
    print('  old adaptive midpoint: y %.3f S %.3f units' % (parity_err(Ya[:, 1:], Ye[:, 1:]), parity_err(Sa[:, 1:], Se[:, 1:])))
