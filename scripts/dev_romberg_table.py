"""Developer check: Romberg table of the graded implicit midpoint rule on stiff50 -- error of every entry against the
finest diagonal entry and against the once-extrapolated finest pair (units: 1e-9 * max(|x|, 1e-3 max|x|) + 1e-12)."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
name = sys.argv[1] if len(sys.argv) > 1 else 'stiff50'
method = sys.argv[2] if len(sys.argv) > 2 else 'implicit_midpoint_graded'
gm = zoo_model(name)
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=name)
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', name + '_ref.npz'))
P = g['P'][:1]; t_out = np.concatenate([[0.0], g['t'][g['idx']]])
K = 9
rows = []
for k in range(K):
    T0 = m.calc_jacobian_batch(P, t_out, method=method, n_steps=256, step_mult=2 ** k, rtol=1e-11, atol=1e-14)[0]
    row = [T0]
    for j in range(1, min(k, 4) + 1):
        f = 4.0 ** j
        row.append((f * row[j - 1] - rows[k - 1][j - 1]) / (f - 1.0))
    rows.append(row)
ref = rows[-1][1]          # finest pair, once extrapolated
sc = 1e-9 * np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max()) + 1e-12
print(name, method, "errors vs finest T[k][1] (tolerance units at rtol 1e-9):")
for k, row in enumerate(rows):
    print("k=%d n=%6d " % (k, 256 * 2 ** k) + "  ".join("%10.3g" % np.max(np.abs(t - ref) / sc) for t in row))
