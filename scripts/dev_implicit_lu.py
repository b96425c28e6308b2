"""Developer timing: the implicit kernels on models whose Newton matrix is not triangular -- row-distributed LU
(default) against the redundant per-lane LU (SBM_PLUGIN_FLAGS=-DSBM_IMPLICIT_REDUNDANT_LU builds that variant under its
own name).  Run once with and once without the variable."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import torch
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import GeneratedModel, zoo_model

print('SBM_PLUGIN_FLAGS =', os.environ.get('SBM_PLUGIN_FLAGS', ''))
V = 1024
for name in ('cascade20', 'dense20_50', 'dense20'):
    gm = zoo_model('cascade20') if name == 'cascade20' else GeneratedModel(models_zoo.dense_spec(density=0.5 if name.endswith('50') else 1.0))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    _, P = models_zoo.cascade_ensemble(V)
    t_out = np.array([0.0, 50.0, 100.0])
    Pd = P
    kw = dict(method='implicit_midpoint', n_steps=512, rtol=1e-10, atol=1e-12)
    S = m.calc_jacobian_batch(Pd, t_out, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        S = m.calc_jacobian_batch(Pd, t_out, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    info = m.last_info
    print("%-11s fixed 512 steps x %d vectors with sensitivities: %.2f ms, Newton iterations / step %.2f, failed %d, checksum %.12g"
          % (name, V, dt * 1e3, 1.0 + float(np.mean(info['n_rejected'])) / 512.0, int(np.count_nonzero(info['status'])),
             float(torch.as_tensor(S).double().abs().sum())))
