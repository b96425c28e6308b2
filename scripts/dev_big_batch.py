"""Developer check: a large batch (32768 vectors x 820 ODEs x 17 output rows = 3.6 GB of sensitivities) gives, vector for
vector, the numbers of a small one -- index arithmetic beyond 2^31 elements, launch order, both explicit pairs."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import torch
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade20')
V = 32768
_, P = models_zoo.cascade_ensemble(V)
grid = np.linspace(0, 100.0, 1000)
t_out = np.concatenate([[0.0], grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]])
dm = m.device_model
Pt, tt = torch.from_numpy(P).cuda(), torch.from_numpy(t_out).cuda()
pick = np.array([0, 1, 4095, 4096, 20000, 32767])
for method, tol in (('dopri45', dict(rtol=1e-9, atol=1e-18)), ('dop853', dict(rtol=1e-10, atol=1e-18))):
    o = _lib.make_opts(method, **tol)
    Y = torch.empty((V, len(t_out), 20), dtype=torch.float64, device='cuda')
    S = torch.empty((V, len(t_out), 20, 40), dtype=torch.float64, device='cuda')
    st = torch.empty((V,), dtype=torch.int32, device='cuda'); ns = torch.empty_like(st)
    dm.sens_dev(Pt, tt, None, o, Y, S, st, ns, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dm.sens_dev(Pt, tt, None, o, Y, S, st, ns, None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    Ps = Pt[torch.from_numpy(pick).cuda()].contiguous()
    Ys = torch.empty((len(pick), len(t_out), 20), dtype=torch.float64, device='cuda')
    Ss = torch.empty((len(pick), len(t_out), 20, 40), dtype=torch.float64, device='cuda')
    dm.sens_dev(Ps, tt, None, o, Ys, Ss, None, None, None)
    torch.cuda.synchronize()
    idx = torch.from_numpy(pick).cuda()
    same = bool(torch.equal(S[idx], Ss)) and bool(torch.equal(Y[idx], Ys))
    print("%-8s V = %d: %.1f ms per pass (%.0f steps per vector, %d failed), S is %.2f GB; picked vectors equal a small batch's bit for bit: %s"
          % (method, V, dt * 1e3, ns.float().mean().item(), int((st != 0).sum()), S.numel() * 8 / 1e9, same), flush=True)
    del Y, S
