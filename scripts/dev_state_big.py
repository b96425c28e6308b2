"""Developer timing: state-only kernels on very large batches (lane-per-trajectory vs one / several per wavefront)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import _lib, models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
dm = m.device_model
t = torch.from_numpy(np.concatenate([[0.0], models_zoo.CASCADE_MEASURE_TIMES])).cuda()
for V in (4096, 32768, 65535, 65536, 262144):
    P = torch.from_numpy(models_zoo.cascade_ensemble(V)[1]).cuda()
    Y = torch.empty((V, 17, 20), dtype=torch.float64, device='cuda')
    ns = torch.empty((V,), dtype=torch.int32, device='cuda')
    for variant in ('auto', 'per_wave', 'row_lane'):
        o = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=variant)
        dm.simulate_dev(P, t, None, o, Y, None, ns, None); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); dm.simulate_dev(P, t, None, o, Y, None, ns, None); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        print("V=%7d %-9s %8.3f ms  %.3g steps/s" % (V, variant, ms, int(ns.sum()) / ms * 1e3), flush=True)
