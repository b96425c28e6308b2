import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd import models_zoo
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'stiff50_ref.npz'))
P, Yr, Sr = g['P'], g['Y'], g['S']
t_out = np.concatenate([[0.0], g['t'][g['idx']]])
IM = dict(method='implicit_midpoint', rtol=1e-10, atol=1e-12)
scale = np.abs(Sr).max()
def err(Y, S):
    return (np.max(np.abs(Y[:, 1:] - Yr) / (np.abs(Yr) + 0.5)), np.max(np.abs(S[:, 1:] - Sr) / (np.abs(Sr) + 0.5)), np.max(np.abs(S[:, 1:] - Sr)) / scale)
for n in (1024, 4096):
    for mult in (1, 2, 4):
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=n, step_mult=mult, **IM)
        print("n %d mult %d steps %d newton+ %d  err y %.2e s %.2e (s/scale %.2e)  status %s" % ((n, mult, m.last_info['n_steps'][0], m.last_info['n_rejected'][0]) + err(Y, S) + (m.last_info['status'].tolist(),)), flush=True)
    for ex in (1, 2):
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=n, extrapolate=ex, **IM)
        print("n %d extrapolate %d: err y %.2e s %.2e (s/scale %.2e)" % ((n, ex) + err(Y, S)), flush=True)
_, P4 = models_zoo.stiff_ensemble(4096)
import torch
for n in (512, 2048):
    torch.cuda.synchronize(); t0 = time.time()
    S = m.calc_jacobian_batch(P4, np.array([0.0, 5.0, 10.0]), n_steps=n, **IM)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("V=4096 n_steps %d: %.1f ms incl. host copies; steps %d newton+ %d" % (n, dt * 1e3, m.last_info['n_steps'].sum(), m.last_info['n_rejected'].sum()))
