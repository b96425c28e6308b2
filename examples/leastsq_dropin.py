"""The reference's own way of fitting, unchanged: SciPy's leastsq on Project.residuals with Project.calc_project_jacobian
as Dfun (tests/test_Project.py:202-213 of the reference) -- one parameter vector per call, every call on the GPU.

    python examples/leastsq_dropin.py

8 experiments x 4 measured species x 16 time points = 512 residual rows, 68 parameters, 4 scale factors
(BASELINE configs[3]); each Jacobian call integrates 8 x 820 coupled ODEs."""
import os
import sys
import time
import warnings

import numpy as np
import scipy.optimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import zoo_model


def main():
    warnings.simplefilter('ignore')
    gm = zoo_model('cascade20')
    model = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
    proj, theta_true = models_zoo.cascade_config4_project(model, reference_compat=False)
    rng = np.random.default_rng(3)
    x0 = theta_true + 0.1 * rng.standard_normal(theta_true.size)
    # MINPACK's unconstrained steps in log-parameters can reach rates that make the system stiff for a trial or two;
    # the default step budget (50000 attempts, given up early when hopeless) keeps such trials short: they come back as
    # inf residuals, which leastsq rejects
    calls = {'f': 0, 'J': 0}

    def f(x):
        calls['f'] += 1
        return proj.residuals(x)

    def J(x):
        calls['J'] += 1
        return proj.calc_project_jacobian(x)
    f(x0), J(x0)                                   # plugin load, first-call allocations
    calls.update(f=0, J=0)
    t0 = time.time()
    x, cov, info, msg, ier = scipy.optimize.leastsq(f, x0, Dfun=J, full_output=True, maxfev=400)
    dt = time.time() - t0
    c0, c1 = 0.5 * np.sum(f(x0) ** 2), 0.5 * np.sum(info['fvec'] ** 2)
    print("leastsq: %d residual calls + %d Jacobian calls in %.2f s (%.2f ms per call incl. MINPACK), ier = %d"
          % (calls['f'], calls['J'], dt, 1e3 * dt / max(calls['f'] + calls['J'], 1), ier))
    t0 = time.time()
    for _ in range(20):
        J(x)
    print("at the fit: calc_project_jacobian %.2f ms per call" % (1e3 * (time.time() - t0) / 20))
    print("cost 0.5 |r|^2: %.1f at the start -> %.3f at the fit (data generated with 5 %% noise: %.3f at the true parameters)"
          % (c0, c1, 0.5 * np.sum(proj.residuals(theta_true) ** 2)))
    # A start further away: some of MINPACK's trial points are stiff.  With a plain step budget they come back as inf
    # residuals; method='auto' integrates them with the implicit rule instead -- what LSODA does for the reference.
    x0_far = theta_true + 0.3 * rng.standard_normal(theta_true.size)
    for label, opts in (("default (explicit, step budget)", dict(method='dopri45')),
                        ("method='auto'", dict(method='auto'))):
        proj.integrator_options.update(opts)
        calls.update(f=0, J=0)
        t0 = time.time()
        x, cov, info, msg, ier = scipy.optimize.leastsq(f, x0_far, Dfun=J, full_output=True, maxfev=400)
        dt = time.time() - t0
        print("far start, %-31s %3d + %3d calls in %5.2f s, cost %.3f" % (label + ':', calls['f'], calls['J'], dt,
                                                                           0.5 * np.sum(info['fvec'] ** 2)))
    # The same serial fit with the eighth-order explicit pair: a seventh of the sequential steps per call.  The model's
    # option names it; the single-vector methods keep their stiff fallback (method='auto', explicit_method='dop853').
    proj.integrator_options.clear()
    model.integrator_options['method'] = 'dop853'
    f(x0), J(x0)
    calls.update(f=0, J=0)
    t0 = time.time()
    x, cov, info, msg, ier = scipy.optimize.leastsq(f, x0, Dfun=J, full_output=True, maxfev=400)
    dt = time.time() - t0
    print("model option method='dop853': %d residual calls + %d Jacobian calls in %.2f s (%.2f ms per call), cost %.3f"
          % (calls['f'], calls['J'], dt, 1e3 * dt / max(calls['f'] + calls['J'], 1), 0.5 * np.sum(info['fvec'] ** 2)))


if __name__ == '__main__':
    main()
