"""Multi-start fit and posterior sampling of the 8-experiment cascade project (BASELINE configs[3]) on one GPU.

    python examples/fit_cascade.py [n_starts]

Builds the project from synthetic data (5 % noise), runs Levenberg-Marquardt from n_starts scattered
starts at once, then walks 64 Metropolis chains from the best fit.  Everything the loops evaluate -- ODEs,
forward sensitivities, scale factors, residuals, Jacobians, normal equations -- runs on the device."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.project.ensembles import ensemble_log_params_batch
from sysbio_modeling_amd.symbolic import zoo_model


def main():
    n_starts = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    warnings.simplefilter('ignore')
    gm = zoo_model('cascade20')
    model = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order)
    proj, theta_true = models_zoo.cascade_config4_project(model, noise=0.05, reference_compat=False)
    print("project: %d experiments, %d residual rows, %d parameters, %d scale factors"
          % (len(list(proj.experiments)), proj.n_project_residuals, proj.n_project_params, len(proj.scale_factors)))
    rng = np.random.default_rng(0)
    starts = theta_true[None, :] + 0.3 * rng.standard_normal((n_starts, theta_true.size))
    c0 = proj.calc_sum_square_residuals_batch(starts)
    t0 = time.time()
    fit = proj.fit_batch(starts, max_iter=150)
    dt = time.time() - t0
    best = int(np.argmin(fit['cost']))
    print("LM: %d starts x up to 150 iterations in %.2f s (%d trajectory integrations with sensitivities)"
          % (n_starts, dt, fit['n_evaluations'] * 8))
    print("    cost: start median %.1f -> fit median %.3f, best %.3f (truth: %.3f)"
          % (np.median(c0), np.median(fit['cost']), fit['cost'][best], proj.calc_sum_square_residuals(theta_true)))
    t0 = time.time()
    ens, ens_F, ratio = ensemble_log_params_batch(proj, np.tile(fit['theta'][best], (64, 1)), steps=200, seeds=1,
                                                  sing_val_cutoff=1e-4, step_scale=0.3, energy='rss')
    print("MCMC: 64 chains x 200 steps in %.2f s, acceptance %.2f" % (time.time() - t0, ratio.mean()))
    sd = ens[50:].reshape(-1, ens.shape[-1]).std(axis=0)
    names = [n for n, _ in proj.get_ordered_project_params()]
    tight = np.argsort(sd)[:3]
    loose = np.argsort(sd)[-3:]
    print("    best constrained (log-units sd): " + ", ".join("%s %.3f" % (names[i], sd[i]) for i in tight))
    print("    sloppiest:                       " + ", ".join("%s %.3f" % (names[i], sd[i]) for i in loose))


if __name__ == '__main__':
    main()
