"""Developer check of SBM_IMPLICIT_EXTRAP on the stiff50 ensemble: golden parity + time per 4096 vectors.
usage: python tests/tools/dev_iex.py [n_vectors] [orders] [rtols]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from oracle.tolerances import parity_err

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
orders = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '8').split(',')]
rtols = [float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '3e-9').split(',')]
atol_abs = float(sys.argv[4]) if len(sys.argv) > 4 else None
gm = zoo_model('stiff50')
ctx = _lib.default_context()
lm = _lib.LoadedModel(ctx, gm.plugin_path(build_if_missing=True))
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
g = np.load(os.path.join(here, 'stiff50_ref.npz')); tg = np.load(os.path.join(here, 'stiff50_tight.npz'))
t_out = g['t'][g['idx']]
_, P = models_zoo.stiff_ensemble(4096, n=50)
dev = torch.device('cuda:0')
Pd = torch.tensor(P[:V], device=dev); td = torch.tensor(t_out, device=dev)
Y = torch.empty((V, len(t_out), 50), device=dev, dtype=torch.float64)
S = torch.empty((V, len(t_out), 50, 50), device=dev, dtype=torch.float64)
st = torch.zeros(V, device=dev, dtype=torch.int32); ns = torch.zeros_like(st); nr = torch.zeros_like(st)
for K in orders:
    for rtol in rtols:
        o = _lib.make_opts('implicit_extrap', rtol=rtol, atol=atol_abs if atol_abs else 1e-3 * rtol, order=K)
        Yh, Sh, sth, nsh, nrh = lm.sens_host(g['P'], t_out, None, o)
        Sh = Sh.reshape(Sh.shape[0], Sh.shape[1], -1)
        print('K %d rtol %g golden: status %s steps %s rej %s | vs ref y %.3f S %.3f | vs tight y %.3f S %.3f' % (
            K, rtol, sth, nsh, nrh, parity_err(Yh, g['Y']), parity_err(Sh, g['S']), parity_err(Yh, tg['Y']), parity_err(Sh, tg['S'])), flush=True)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            lm.sens_dev(Pd, td, None, o, Y, S, st, ns, nr)
            torch.cuda.synchronize(); dt = time.time() - t0
        print('   V %d sens: %.3f s; status!=0: %d; steps mean %.1f max %d rej mean %.1f (with SBM_IEX_COUNT_NEWTON: evaluations; %.2f per Euler step if no rejections)' % (
            V, dt, int((st != 0).sum()), ns.double().mean().item(), ns.max().item(), nr.double().mean().item(),
            nr.double().mean().item() / (ns.double().mean().item() * K * (K + 1) / 2)), flush=True)
        torch.cuda.synchronize(); t0 = time.time()
        lm.simulate_dev(Pd, td, None, o, Y, st, ns, nr)
        torch.cuda.synchronize(); dt = time.time() - t0
        print('   V %d state only: %.3f s; steps mean %.1f' % (V, dt, ns.double().mean().item()), flush=True)
