"""Developer timing: one row-group split against another (SBM_RG_FORCE_PLAN="G,C,CPL,RPG,NCH") on cascade(n)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
n = int(sys.argv[1]); V = int(sys.argv[2])
gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
import re
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
dm = m.device_model
P = torch.from_numpy(models_zoo.cascade_ensemble(V, n=n, spread=0.3)[1]).cuda()
t = torch.tensor([50.0, 100.0], dtype=torch.float64, device='cuda')
Y = torch.empty((V, 2, n), dtype=torch.float64, device='cuda'); S = torch.empty((V, 2, n, 2 * n), dtype=torch.float64, device='cuda')
st = torch.empty((V,), dtype=torch.int32, device='cuda'); ns = torch.empty((V,), dtype=torch.int32, device='cuda')
o = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12)
dm.sens_dev(P, t, None, o, Y, S, st, ns, None); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); dm.sens_dev(P, t, None, o, Y, S, st, ns, None); b.record(); torch.cuda.synchronize()
lay = re.findall(r"RG_G = [^;]*;", gm.hip_source)[0] + ' ' + re.findall(r"RG_NCH = \d+", gm.hip_source)[0]
print("cascade%d V=%d plan %s | %s: %.2f ms, %.3g steps/s" % (n, V, os.environ.get('SBM_RG_FORCE_PLAN', 'default'), lay, a.elapsed_time(b), int(ns.sum()) / a.elapsed_time(b) * 1e3), flush=True)
