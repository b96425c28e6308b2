"""Developer timing of the sensitivity kernels on the configs[2] ensemble (not a test, not the bench).

usage: python scripts/dev_time.py [variant ...]     variants: auto per_wave row_lane row_group
Prints ms per launch (HIP events on the library's stream = torch's current stream) for DOPRI45 and
RK4-4096, plus a checksum against the row_lane variant so a fast-but-wrong build is visible."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd import models_zoo, _lib

variants = sys.argv[1:] or ['row_lane', 'row_group']
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
dm = m.device_model
V = int(os.environ.get('V', '4096'))
_, P = models_zoo.cascade_ensemble(V)
grid = np.linspace(0, 100, 1000)
t_meas = grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]
Pd = torch.from_numpy(P).cuda()
td = torch.from_numpy(t_meas).cuda()
Y = torch.empty((V, len(t_meas), 20), dtype=torch.float64, device='cuda')
S = torch.empty((V, len(t_meas), 20, 40), dtype=torch.float64, device='cuda')
st = torch.empty(V, dtype=torch.int32, device='cuda')
ns = torch.empty_like(st)
nr = torch.empty_like(st)
ref = {}
for meth in ('dopri45', 'rk4'):
    for var in variants:
        o = (_lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=var) if meth == 'dopri45'
             else _lib.make_opts('rk4', n_steps=4096, t_end=100.0, variant=var))
        dm.sens_dev(Pd, td, None, o, Y, S, st, ns, nr)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        a.record()
        for _ in range(reps):
            dm.sens_dev(Pd, td, None, o, Y, S, st, ns, nr)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        steps = int(ns.sum().item())
        chk = S.double().abs().sum().item()
        ref.setdefault(meth, chk)
        print("%-8s %-10s %8.3f ms  steps %9d  rej %7d  %.3e steps/s  frac %.3f  bad %d  checksum rel diff %.1e" % (
            meth, var, ms, steps, int(nr.sum().item()), steps / ms * 1e3, steps / ms * 1e3 * 13120 / 8e12,
            int((st != 0).sum().item()), abs(chk - ref[meth]) / ref[meth]), flush=True)
