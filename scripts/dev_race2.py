import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
ctx = _lib.default_context()
gm = zoo_model('cascade20')
from sysbio_modeling_amd import build
import os
hdr = os.path.join(build.MODELS_DIR, 'cascade20.hpp')
MODE = int(os.environ.get('MASKMODE', '0'))
path = gm.plugin_path() if MODE == 0 else build.build_plugin('cascade20_dbg%d' % MODE, hdr, extra_flags=('-DSBM_DBG_MASK=%d' % MODE,))
path = os.environ.get('PLUGIN', path)
dm = _lib.LoadedModel(ctx, path)
V = 64
_, P = models_zoo.cascade_ensemble(V)
t = np.concatenate([[0.0], models_zoo.CASCADE_MEASURE_TIMES])
Pd, td = torch.from_numpy(P).cuda(), torch.from_numpy(t).cuda()
def run(label, opts, fill):
    Y = torch.full((V, len(t), 20), fill, dtype=torch.float64, device='cuda')
    S = torch.full((V, len(t), 20, 40), fill, dtype=torch.float64, device='cuda')
    st = torch.full((V,), -1, dtype=torch.int32, device='cuda'); ns = torch.zeros_like(st); nr = torch.zeros_like(st)
    dm.sens_dev(Pd, td, None, opts, Y, S, st, ns, nr)
    torch.cuda.synchronize()
    print("%-40s bad=%d status0=%d acc0=%d rej0=%d  Y[0,1,:3]=%s S[0,1,0,:3]=%s" % (label, int((st!=0).sum()), int(st[0]), int(ns[0]), int(nr[0]),
          Y[0,1,:3].cpu().numpy(), S[0,1,0,:3].cpu().numpy()), flush=True)
for fill in (float('nan'), 0.0, 1e300):
    run("dopri row_lane fill=%g" % fill, _lib.make_opts('dopri45', variant='row_lane'), fill)
    run("dopri row_lane h0=1e-3 fill=%g" % fill, _lib.make_opts('dopri45', variant='row_lane', h0=1e-3), fill)
    run("rk4 row_lane fill=%g" % fill, _lib.make_opts('rk4', n_steps=1000, t_end=100., variant='row_lane'), fill)
    run("dopri per_wave fill=%g" % fill, _lib.make_opts('dopri45', variant='per_wave'), fill)
