"""Developer timing: the control loops on the full configs[4] ensemble (stiff50, 4096 vectors, 2 output times)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
P = models_zoo.stiff_ensemble(4096)[1]
t = np.array([0.0, 5.0, models_zoo.STIFF_T_END])
m.calc_jacobian_batch(P[:8], t, method='implicit_midpoint', n_steps=64)
for rtol, atol in ((1e-5, 1e-8), (1e-7, 1e-10), (1e-9, 1e-12)):
    t0 = time.time()
    S = m.calc_jacobian_batch(P, t, method='implicit_controlled', rtol=rtol, atol=atol)
    dt = time.time() - t0
    i = m.last_info
    print("implicit_controlled rtol %.0e: %.2f s, steps/vector mean %d, levels %d..%d, not converged %d" %
          (rtol, dt, i['n_steps'].mean(), i['levels'].min(), i['levels'].max(), int((i['status'] != 0).sum())), flush=True)
t0 = time.time()
S = m.calc_jacobian_batch(P, t, method='auto', max_steps=20000, rtol=1e-7, atol=1e-10)
print("auto (budget 20000 explicit steps) rtol 1e-7: %.2f s, stiff %d of %d" % (time.time() - t0, int(m.last_info['stiff'].sum()), len(P)))
