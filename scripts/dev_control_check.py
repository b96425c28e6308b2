"""Developer check: is what implicit_controlled returns on stiff50 (3 output times) really within its tolerance?"""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.model import OdeModel
gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
P = models_zoo.stiff_ensemble(4096)[1][:32]
t = np.array([0.0, 5.0, models_zoo.STIFF_T_END])
fine = m.calc_jacobian_batch(P, t, method='implicit_midpoint_graded', n_steps=256, step_mult=256, extrapolate=1, rtol=1e-12, atol=1e-15)
for rtol, atol in ((1e-5, 1e-8), (1e-7, 1e-10), (1e-9, 1e-12)):
    m._control_trace = []
    S = m.calc_jacobian_batch(P, t, method='implicit_controlled', rtol=rtol, atol=atol)
    sc = rtol * np.maximum(np.abs(fine), 1e-3 * np.abs(fine).reshape(32, -1).max(axis=1)[:, None, None]) + atol
    err = np.max(np.abs(S - fine) / sc, axis=(1, 2))
    print("rtol %.0e: levels %s  true error / tolerance: max %.2f  (per vector %s)" % (rtol, np.bincount(m.last_info['levels']), err.max(), np.round(err[:8], 2)))
    for lv, idx, e in m._control_trace:
        print("    level %d: %d vectors, estimates %s" % (lv, len(idx), np.array2string(e[:6], precision=2)))
