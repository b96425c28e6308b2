"""CPU experiment on the scheme oracle (oracle/iex_oracle.py): Newton evaluations per implicit-Euler step of the extrapolated
scheme with the kernel's predictor (polynomial through the sequence's own last points) and with a predictor that takes the
first steps of sequence j from the PREVIOUS sequence of the same macro step:  yb = ya + q(tau) - q(tau - h),  q = the
quadratic (linear for j = 2) through the nearest states of sequence j - 1.  usage: python tests/tools/dev_iex_predictor.py [t_end]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import iex_oracle
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd import models_zoo

t_end = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
upto = int(sys.argv[2]) if len(sys.argv) > 2 else 3         # own points from which the sequence's own polynomial is used
gm = zoo_model('stiff50')
_, P = models_zoo.stiff_ensemble(4096, n=50)


def cross(j, m, h, Hs, y_n, ya, default, prev):
    if m >= upto or j < 2 or not prev:
        return default
    hp = Hs / (j - 1)
    xs = [0.0] + [(i + 1) * hp for i in range(len(prev))]
    ys = [y_n] + list(prev)
    tau = (m + 1) * h
    if len(xs) == 2:
        return ya + (ys[1] - ys[0]) * (h / hp)
    c = min(max(int(round(tau / hp)), 1), len(xs) - 2)       # centre of the three nearest grid points
    x0, x1, x2 = xs[c - 1], xs[c], xs[c + 1]

    def q(x):
        return (ys[c - 1] * (x - x1) * (x - x2) / ((x0 - x1) * (x0 - x2)) + ys[c] * (x - x0) * (x - x2) / ((x1 - x0) * (x1 - x2))
                + ys[c + 1] * (x - x0) * (x - x1) / ((x2 - x0) * (x2 - x1)))
    return ya + q(tau) - q(tau - h)


for v in (0, 1500, 3000):
    for name, pred in (('own points (kernel)', None), ('previous sequence', cross)):
        t0 = time.time()
        Y, S, info = iex_oracle.integrate(gm, P[v], [t_end], rtol=1e-9, atol=3e-13, predictor=pred)
        print('vector %4d %-20s macro steps %3d (+%d rejected) evaluations %6d = %.3f per Euler step  (%.0f s)' % (
            v, name, info['n_steps'], info['n_reject'], info['n_eval'], info['n_eval'] / max(info['n_euler'], 1), time.time() - t0),
            flush=True)
