"""Developer check of the early exit of DOPRI45's step budget: rejection rates of stiff and non-stiff runs."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model

def model(name):
    gm = zoo_model(name)
    return OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=name)

m = model('stiff50')
_, P = models_zoo.stiff_ensemble(4)
t = np.array([0.0, 5.0, 10.0])
for ms in (20000, -20000, -50000):
    m.simulate_batch(P, t, method='dopri45', max_steps=ms)
    print('stiff50 state-only max_steps', ms, 'status', m.last_info['status'], 'acc', m.last_info['n_steps'], 'rej', m.last_info['n_rejected'])
    m.calc_jacobian_batch(P, t, method='dopri45', max_steps=ms)
    print('stiff50 sens       max_steps', ms, 'status', m.last_info['status'], 'acc', m.last_info['n_steps'], 'rej', m.last_info['n_rejected'])
c = model('cascade20')
_, Pc = models_zoo.cascade_ensemble(4)
for t_end in (100.0, 3.0e4, 1.0e5, 1.0e6):
    for ms in (-50000, 1000000):
        c.calc_jacobian_batch(Pc, np.array([0.0, t_end]), max_steps=ms)
        print('cascade20 sens t_end %g max_steps %d' % (t_end, ms), 'status', c.last_info['status'], 'acc', c.last_info['n_steps'], 'rej', c.last_info['n_rejected'])
mm = model('michaelis_menten')
from tests import reference_cases as rc
Pm = np.tile(rc.MM_PARAMS, (4, 1)) * np.array([[1.0], [10.0], [100.0], [1000.0]])
for ms in (-50000, 1000000):
    mm.calc_jacobian_batch(Pm, np.array([0.0, 100.0, 1.0e4]), max_steps=ms)
    print('mm rates x1..x1000 t_end 1e4 max_steps', ms, 'status', mm.last_info['status'], 'acc', mm.last_info['n_steps'], 'rej', mm.last_info['n_rejected'])
