import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
from oracle.tolerances import parity_err
warnings.simplefilter('ignore')
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'stiff50_ref.npz'))
t_out = np.concatenate([[0.0], g['t'][g['idx']]])
_, P = models_zoo.stiff_ensemble(4096)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for sens in (False, True):
    for rtol, atol in [(1e-6, 1e-9), (1e-9, 1e-12)]:
        kw = dict(method='implicit_adaptive', rtol=rtol, atol=atol, max_steps=int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
        t0 = time.perf_counter()
        if sens:
            S, Y = m.calc_jacobian_batch(P[:V], t_out, return_states=True, **kw)
        else:
            Y = m.simulate_batch(P[:V], t_out, **kw)
        dt = time.perf_counter() - t0
        i = m.last_info
        print('sens', sens, rtol, 'time %.2fs' % dt, 'status', i['status'][:8], 'steps', i['n_steps'][:8], 'rej', i['n_rejected'][:8], flush=True)
        ok = i['status'][:3] == 0
        if ok.all():
            print('   err y %.2f' % parity_err(Y[:3, 1:], g['Y']), ('S %.2f' % parity_err(S[:3, 1:], g['S'])) if sens else '', flush=True)
        else:
            print('   first NaN output index per vector:', [int(np.argmax(np.isnan(Y[v, :, 0]))) for v in range(min(V, 8))])
