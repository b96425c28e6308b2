"""Developer timing of the stiff path (configs[4] ensemble, default options) without the bench scaffolding:
    python tests/tools/dev_iex_time.py [V]
Environment: SBM_PLUGIN_FLAGS="-DFOO=1" times a developer build of the plugin, SBM_IEX_SEQ=0 the round-3 kernel."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from sysbio_modeling_amd import _lib, models_zoo          # noqa: E402
from sysbio_modeling_amd.symbolic import zoo_model        # noqa: E402
from sysbio_modeling_amd.model import OdeModel            # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda', 0)
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, use_jit=False)
m.enable_jit(_lib.Context(0))
_, Pn = models_zoo.stiff_ensemble(V)
P = torch.from_numpy(Pn).to(dev)
t = torch.from_numpy(np.concatenate([[0.0], models_zoo.STIFF_MEASURE_TIMES])).to(dev)
nt = len(t)
Y = torch.empty((V, nt, 50), dtype=torch.float64, device=dev)
S = torch.empty((V, nt, 50, 50), dtype=torch.float64, device=dev)
st, ns, nr = (torch.empty((V,), dtype=torch.int32, device=dev) for _ in range(3))
o = dict(m.integrator_options, method='implicit_extrap')
_lib.implicit_adaptive_defaults(o, ())
opts = _lib.make_opts('implicit_extrap', order=8, rtol=o['rtol'], atol=o['atol'])
dm = m.device_model


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


ms = timed(lambda: dm.sens_dev(P, t, None, opts, Y, S, st, ns, nr))
macro, rej = int(ns.sum()), int(nr.sum())
if 'SBM_SEQ_PROFILE' in os.environ.get('SBM_PLUGIN_FLAGS', ''):
    print("profile build: kilocycles phase A %d, phase B %d, trajectories %d (sum over %d trajectories; kernel %.2f ms)"
          % (macro, rej, int(st.sum()), V, ms))
ms0 = timed(lambda: dm.simulate_dev(P, t, None, opts, Y, st, ns, nr))
print("flags=%r seq=%s V=%d: sens %.2f ms (%d macro + %d rejected, failed %d, checksum %.12e); state only %.2f ms (%d macro)"
      % (os.environ.get('SBM_PLUGIN_FLAGS', ''), os.environ.get('SBM_IEX_SEQ', '1'), V, ms, macro, rej, int((st != 0).sum()),
         float(S[:, -1].abs().sum()), ms0, int(ns.sum())))
