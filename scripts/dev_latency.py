"""Developer timing: latency of the reference-style single-vector calls (what a serial optimiser sees)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model
gm = zoo_model('cascade20')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
proj, th = models_zoo.cascade_config4_project(m)
for method in ('dopri45', 'dop853'):
  m.integrator_options['method'] = method
  print('--- model option method =', method)
  for name, fn in (('residuals', lambda: proj.residuals(th)), ('calc_project_jacobian', lambda: proj.calc_project_jacobian(th)),
                   ('calc_rss_gradient', lambda: proj.calc_rss_gradient(th)),
                   ('OdeModel.simulate', lambda: m.simulate(models_zoo.cascade_nominal_params(), np.linspace(0, 100, 1000))),
                   ('OdeModel.calc_jacobian', lambda: m.calc_jacobian(models_zoo.cascade_nominal_params(), np.linspace(0, 100, 1000), np.zeros(820)))):
      fn()
      t0 = time.perf_counter()
      for _ in range(20):
          fn()
      print("%-24s %.2f ms per call" % (name, (time.perf_counter() - t0) / 20 * 1e3))

m.integrator_options['method'] = 'dopri45'
# where the time of one residuals() call goes: kernel time by events around the C call alone
import torch
from sysbio_modeling_amd import _lib
import ctypes
th_d = torch.from_numpy(np.asarray(th)[None, :]).cuda()
for jac in (False, True):
    o = proj.evaluate_batch(th_d, jacobian=jac, want=('jacobian',) if jac else ('residuals',))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(20):
        o = proj.evaluate_batch(th_d, jacobian=jac, want=('jacobian',) if jac else ('residuals',))
    b.record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e3
    print("evaluate_batch(device tensor, V=1, jacobian=%s): wall %.3f ms per call, GPU busy %.3f ms per call" % (jac, wall, a.elapsed_time(b) / 20))
