"""Developer check: every public entry point of OdeModel / Project / fit_batch / the sampler on a STIFF project, at default
options and with each integrator option -- looking for option plumbing that only a default-path call would trip over."""
import sys, warnings, traceback
import numpy as np
sys.path.insert(0, '.')
warnings.simplefilter('ignore')
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model
from sysbio_modeling_amd.experiment import Experiment
from sysbio_modeling_amd.measurement import TimecourseMeasurement
from sysbio_modeling_amd.project import Project

def attempt(name, fn):
    try:
        out = fn()
        desc = ''
        if isinstance(out, np.ndarray):
            desc = 'shape %s finite %s' % (out.shape, bool(np.all(np.isfinite(out))))
        elif isinstance(out, (float, np.floating)):
            desc = '%.6g' % out
        elif isinstance(out, dict):
            desc = 'keys %s' % sorted(out)[:6]
        print('OK   %-55s %s' % (name, desc), flush=True)
    except Exception as e:
        print('FAIL %-55s %r' % (name, e), flush=True)
        traceback.print_exc()

gm = zoo_model('stiff50')
m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
g = np.load('tests/golden/stiff50_ref.npz')
P, Yr = g['P'], g['Y']
species = (0, 10, 49)
ms = [TimecourseMeasurement('s%02d' % v, Yr[0][:, v] * 1.3, models_zoo.STIFF_MEASURE_TIMES.copy(), 0.05 * np.abs(Yr[0][:, v]) + 0.01) for v in species]
exp = Experiment('E', ms, fixed_parameters={'b%d' % i: 0.5 for i in range(50)})
def make(**kw):
    return Project(m, [exp], {'Global': ['a%d' % i for i in range(50)], 'Fixed': ['b%d' % i for i in range(50)]},
                   {('s%02d' % v): ('direct', v) for v in species}, **kw)
proj = make(reference_compat=False)
names = list(m.param_order)
theta = np.zeros(50)
for j in range(50):
    theta[proj.get_param_index('a%d' % j, 'Global')] = np.log(P[0, names.index('a%d' % j)])
grid = np.linspace(0, 10, 1000)
attempt('OdeModel.simulate default (stiff)', lambda: m.simulate(P[0], grid))
attempt('OdeModel.calc_jacobian default (stiff)', lambda: m.calc_jacobian(P[0], grid, np.zeros(50 + 2500)))
attempt('Project.residuals default (stiff)', lambda: proj.residuals(theta))
attempt('Project.calc_project_jacobian default', lambda: proj.calc_project_jacobian(theta))
attempt('Project.calc_rss_gradient default', lambda: proj.calc_rss_gradient(theta))
attempt('Project.calc_sum_square_residuals default', lambda: proj.calc_sum_square_residuals(theta))
gr = np.zeros(50)
attempt('Project.nlopt_fcn default', lambda: proj.nlopt_fcn(theta, gr))
attempt('Project(reference_compat=True).residuals', lambda: make().residuals(theta))
attempt('Project(reference_compat=True).calc_project_jacobian', lambda: make().calc_project_jacobian(theta))
th2 = np.stack([theta, theta + 0.05])
attempt('evaluate_batch auto', lambda: proj.evaluate_batch(th2, jacobian=True, want=('jacobian',), method='auto'))
attempt('evaluate_batch implicit_controlled', lambda: proj.evaluate_batch(th2, jacobian=True, want=('jacobian',), method='implicit_controlled'))
attempt('evaluate_batch implicit_romberg', lambda: proj.evaluate_batch(th2, method='implicit_romberg', rtol=1e-7, atol=1e-10))
attempt('evaluate_batch implicit_midpoint fixed + extrapolate', lambda: proj.evaluate_batch(th2, jacobian=True, want=('jacobian',), method='implicit_midpoint', n_steps=1024, extrapolate=1))
attempt('fit_batch method=auto (4 starts, 3 iterations)', lambda: proj.fit_batch(np.stack([theta + 0.02 * k for k in range(4)]), max_iter=3, method='auto', max_steps=20000))
attempt('fit_batch default options on the stiff project', lambda: proj.fit_batch(np.stack([theta + 0.02 * k for k in range(4)]), max_iter=2))
from sysbio_modeling_amd.project.ensembles import ensemble_log_params_batch
attempt('sampler method=auto (4 chains, 3 steps)', lambda: ensemble_log_params_batch(proj, np.tile(theta, (4, 1)), steps=3, seeds=1, method='auto', max_steps=20000)[0])
attempt('sampler recalc_hess_alg method=implicit_controlled', lambda: ensemble_log_params_batch(proj, np.tile(theta, (2, 1)), steps=2, seeds=1, recalc_hess_alg=True, method='implicit_controlled')[0])
proj.integrator_options.update(method='implicit_controlled')
attempt('Project.residuals with project option implicit_controlled', lambda: proj.residuals(theta))
m.integrator_options.update(method='auto')
attempt('OdeModel.simulate_batch with model option auto', lambda: m.simulate_batch(P, np.array([0.0, 5.0, 10.0])))
attempt('OdeModel.calc_jacobian_batch with model option auto', lambda: m.calc_jacobian_batch(P, np.array([0.0, 5.0, 10.0])))
# custom observable on the stiff model with auto
proj_c = Project(m, [Experiment('E', [TimecourseMeasurement('ratio', Yr[0][:, 10] / (Yr[0][:, 0] + 1.0), models_zoo.STIFF_MEASURE_TIMES.copy(), 0.05 + 0 * Yr[0][:, 0])],
                            fixed_parameters={'b%d' % i: 0.5 for i in range(50)})],
                 {'Global': ['a%d' % i for i in range(50)], 'Fixed': ['b%d' % i for i in range(50)]},
                 {'ratio': ('custom', 'x10 / (x0 + 1)')}, reference_compat=False)
attempt('custom observable, residuals (model option auto)', lambda: proj_c.residuals(theta))
attempt('custom observable, jacobian', lambda: proj_c.calc_project_jacobian(theta))
