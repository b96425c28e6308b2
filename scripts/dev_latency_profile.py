"""cProfile of the single-vector Project calls a serial optimiser makes (where the host time of one call goes)."""
import os, sys, time, warnings, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
proj, th = models_zoo.cascade_config4_project(m)
for name, fn in (('residuals', lambda: proj.residuals(th)), ('calc_project_jacobian', lambda: proj.calc_project_jacobian(th))):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(100): fn()
    print("%s: %.3f ms per call" % (name, (time.perf_counter() - t0) * 10))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): fn()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[4:26]))
