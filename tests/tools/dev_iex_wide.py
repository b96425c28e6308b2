"""Default tolerances of SBM_IMPLICIT_EXTRAP against the 35 real-reference stiff50 vectors (and their tight solutions):
worst error over the vectors and time of the 4096-vector launch, per (rtol, atol).  usage: python tests/tools/dev_iex_wide.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from oracle.tolerances import parity_err
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
g, gt, w, wt = (np.load(os.path.join(G, n)) for n in ('stiff50_ref.npz', 'stiff50_tight.npz', 'stiff50_wide_ref.npz', 'stiff50_wide_tight.npz'))
P = np.concatenate([g['P'], w['P']]); Yr = np.concatenate([g['Y'], w['Y']]); Sr = np.concatenate([g['S'], w['S']])
Yt = np.concatenate([gt['Y'], wt['Y']]); St = np.concatenate([gt['S'], wt['S']])
t_out = np.concatenate([[0.0], g['t'][g['idx']]])
gm = zoo_model('stiff50')
lm = _lib.LoadedModel(_lib.default_context(), gm.plugin_path(build_if_missing=True))
dev = torch.device('cuda:0')
_, Pall = models_zoo.stiff_ensemble(4096)
Pd = torch.tensor(Pall, device=dev); td = torch.tensor(t_out, device=dev)
Y = torch.empty((4096, len(t_out), 50), device=dev, dtype=torch.float64); S = torch.empty((4096, len(t_out), 50, 50), device=dev, dtype=torch.float64)
st = torch.zeros(4096, device=dev, dtype=torch.int32); ns = torch.zeros_like(st); nr = torch.zeros_like(st)
print('reference LSODA vs tight over 35 vectors: y %.2f S %.2f' % (max(parity_err(Yr[v], Yt[v]) for v in range(35)), max(parity_err(Sr[v], St[v]) for v in range(35))))
for rtol, atol in ((3e-9, 3e-12), (3e-9, 1e-12), (3e-9, 3e-13), (2e-9, 2e-12), (2e-9, 6e-13), (1e-9, 1e-12), (1e-9, 3e-13)):
    o = _lib.make_opts('implicit_extrap', rtol=rtol, atol=atol, order=8)
    Yh, Sh, sth, nsh, nrh = lm.sens_host(P, t_out, None, o)
    Sh = Sh.reshape(35, len(t_out), -1)
    ey = [parity_err(Yh[v, 1:], Yt[v]) for v in range(35)]; es = [parity_err(Sh[v, 1:], St[v]) for v in range(35)]
    er = [parity_err(Sh[v, 1:], Sr[v]) for v in range(35)]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time(); lm.sens_dev(Pd, td, None, o, Y, S, st, ns, nr); torch.cuda.synchronize(); dt = time.time() - t0
    print('rtol %g atol %g: vs tight worst y %.2f S %.2f (median S %.2f); vs reference worst S %.2f; steps %d-%d | 4096 vectors %.1f ms, mean steps %.0f'
          % (rtol, atol, max(ey), max(es), np.median(es), max(er), nsh.min(), nsh.max(), 1e3 * dt, ns.double().mean().item()), flush=True)
