"""Developer timing: the implicit kernel on configs[4] (stiff50, 4096 vectors, 2048 steps), kernel time by events.
A/B builds through SBM_PLUGIN_FLAGS (e.g. -DSBM_IMID_MIN_WAVES=2)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
dm = m.device_model
V = 4096
P = torch.from_numpy(models_zoo.stiff_ensemble(V)[1]).cuda()
t = torch.tensor([5.0, models_zoo.STIFF_T_END], dtype=torch.float64, device='cuda')
Y = torch.empty((V, 2, 50), dtype=torch.float64, device='cuda'); S = torch.empty((V, 2, 50, 50), dtype=torch.float64, device='cuda')
st = torch.empty((V,), dtype=torch.int32, device='cuda'); ns = torch.empty((V,), dtype=torch.int32, device='cuda'); nw = torch.empty((V,), dtype=torch.int32, device='cuda')
o = _lib.make_opts('implicit_midpoint', rtol=1e-10, atol=1e-12, n_steps=2048, t_end=models_zoo.STIFF_T_END)
dm.sens_dev(P, t, None, o, Y, S, st, ns, nw); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); dm.sens_dev(P, t, None, o, Y, S, st, ns, nw); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b); n = int(ns.sum())
print("flags %r: %.2f ms, %.3g steps/s, newton/step %.2f, failed %d, checksum %.12e" % (os.environ.get('SBM_PLUGIN_FLAGS', ''), ms, n / ms * 1e3, 1 + float(nw.sum()) / n, int((st != 0).sum()), float(S.sum())))
