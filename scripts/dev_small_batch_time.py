"""Developer timing: one small-batch split against another (SBM_RG_FORCE_PLAN_SMALL="G,C,CPL,RPG,NCH"): GPU-busy time of
a single-vector Jacobian evaluation of the configs[3] project (8 trajectories of cascade20)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel
from sysbio_modeling_amd.model import OdeModel
import re
gm = GeneratedModel(models_zoo.cascade_spec(name='cascade20sb'))
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade20sb')
proj, th = models_zoo.cascade_config4_project(m)
th_d = torch.from_numpy(np.asarray(th)[None, :]).cuda()
kw = dict(jacobian=True, want=('jacobian',), variant='small_batch')
proj.evaluate_batch(th_d, **kw); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    o = proj.evaluate_batch(th_d, **kw)
b.record(); torch.cuda.synchronize()
src = gm.hip_source
lay = src[src.index('struct RG1'):] if 'struct RG1' in src else 'RG1 = RG0'
print("small-batch plan %s | %s %s: GPU busy %.3f ms per Jacobian call" % (os.environ.get('SBM_RG_FORCE_PLAN_SMALL', 'default'),
      re.findall(r"RG_G = [^;]*;", lay)[0] if 'RG_G' in lay else lay, re.findall(r"RG_NCH = \d+", lay)[0] if 'RG_NCH' in lay else '', a.elapsed_time(b) / 20), flush=True)
