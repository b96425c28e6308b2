"""Developer aid: the in-kernel adaptive implicit integrator on the stiff50 ensemble -- error against the real
reference's LSODA results (tests/golden/stiff50_ref.npz) and time for 4096 vectors, per tolerance."""
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
tp = os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'stiff50_tight.npz')
gt = np.load(tp) if os.path.exists(tp) else None
t_out = np.concatenate([[0.0], g['t'][g['idx']]])
_, P = models_zoo.stiff_ensemble(4096)
dev = torch.device('cuda')
Pd = torch.from_numpy(P).to(dev); td = torch.from_numpy(t_out).to(dev)
V, nt = 4096, len(t_out)
Y = torch.empty((V, nt, 50), dtype=torch.float64, device=dev); S = torch.empty((V, nt, 50, 50), dtype=torch.float64, device=dev)
st = torch.empty((V,), dtype=torch.int32, device=dev); ns = torch.empty_like(st); nr = torch.empty_like(st)
for rtol, atol in [(1e-6, 1e-9), (1e-7, 1e-10), (1e-8, 1e-11), (1e-9, 1e-12)]:
    o = _lib.make_opts('implicit_adaptive', rtol=rtol, atol=atol)
    m.device_model.sens_dev(Pd, td, None, o, Y, S, st, ns, nr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.device_model.sens_dev(Pd, td, None, o, Y, S, st, ns, nr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ey = parity_err(Y[:3, 1:].cpu().numpy(), g['Y']); es = parity_err(S[:3, 1:].cpu().numpy().reshape(3, nt - 1, 2500), g['S'])
    print("rtol %.0e atol %.0e: %.1f ms, macro steps mean %.0f max %d, rejected mean %.1f, failed %d, err vs golden: y %.2f S %.2f"
          % (rtol, atol, 1e3 * dt, ns.float().mean().item(), ns.max().item(), nr.float().mean().item(), int((st != 0).sum()), ey, es), flush=True)
    if gt is not None:
        print("      vs TIGHT: y %.3f S %.3f" % (parity_err(Y[:3, 1:].cpu().numpy(), gt['Y']), parity_err(S[:3, 1:].cpu().numpy().reshape(3, nt - 1, 2500), gt['S'])), flush=True)
# state only
o = _lib.make_opts('implicit_adaptive', rtol=1e-9, atol=1e-12)
m.device_model.simulate_dev(Pd, td, None, o, Y, st, ns, nr); torch.cuda.synchronize()
t0 = time.perf_counter(); m.device_model.simulate_dev(Pd, td, None, o, Y, st, ns, nr); torch.cuda.synchronize()
print("state only rtol 1e-9: %.1f ms, steps %.0f, err y %.2f" % (1e3 * (time.perf_counter() - t0), ns.float().mean().item(), parity_err(Y[:3, 1:].cpu().numpy(), g['Y'])))
