"""Developer check: how much of the sens kernel's time is load imbalance?  Simulates the hardware's
in-order workgroup dispatch onto S slots with per-trajectory cost = accepted steps."""
import os, sys, heapq
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd import models_zoo, _lib

gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
V = 4096
_, P = models_zoo.cascade_ensemble(V)
grid = np.linspace(0, 100, 1000)
t_meas = grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]
m.calc_jacobian_batch(P, np.concatenate([[0.0], t_meas]))
ns = m.last_info['n_steps'].astype(np.int64) + m.last_info.get('n_rejected', m.last_info.get('n_rej', 0))
print("steps: mean %.1f  min %d  max %d  p99 %d" % (ns.mean(), ns.min(), ns.max(), np.percentile(ns, 99)))
for S in (1024, 2048):
    for name, order in (('in order', np.arange(V)), ('longest first', np.argsort(-ns))):
        heap = [0] * S
        heapq.heapify(heap)
        for v in order:
            t = heapq.heappop(heap)
            heapq.heappush(heap, t + ns[v])
        print("slots %d  %-14s makespan %d  ideal %.0f  ratio %.3f" % (S, name, max(heap), ns.sum() / S, max(heap) / (ns.sum() / S)))
