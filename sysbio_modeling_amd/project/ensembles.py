"""Multi-chain Metropolis sampling of the parameter posterior (a15).

The reference's sampler (project/Ensembles.py:55-178, taken from SloppyCell) walks ONE chain: a
candidate from a Gaussian whose axes come from the Hessian (``_sampling_matrix`` :226-258,
``_trial_move`` :260-264), one ``free_energy`` evaluation -- a full set of LSODA runs -- per step,
Metropolis acceptance ``rand < exp(-dF / T)`` (:193-198).  Here C chains take their steps together: the C
candidates of a step are ONE batched device evaluation (``Project.free_energy_batch`` or, without
scale-factor priors, 0.5 |r|^2 from ``sbm_residuals_batch``).  Same candidate density, same acceptance
rule, same outputs, with a leading chain axis.

The reference module itself cannot run (Python-2 ``print`` / ``cPickle``, and it calls
``Project.hessian``, which does not exist): the Hessian used here is the Gauss-Newton J^T J of the
project Jacobian, what SloppyCell's ``GetJandJtJInLogParameters`` -- the function the docstring names --
returns.
"""
from __future__ import annotations

import numpy as np


def sampling_matrix(hessian, cutoff=0.0, temperature=1.0, step_scale=1.0):
    """Candidate-move matrix M: a move is M @ z with z standard normal, i.e. Gaussian with covariance M M^T.

    The recipe is SloppyCell's, as the reference carries it (_sampling_matrix, Ensembles.py:226-258): along each
    principal axis v_i of A = hessian / 2 (eigenvalue a_i) the step has standard deviation 1 / sqrt(max(a_i, c)) with
    c = cutoff * max a, so flat directions are not followed without bound; everything is then divided by
    sqrt(n_eff), n_eff = sum_i min(a_i / c, 1) (= the number of axes when nothing is clipped), which makes the
    expected quadratic cost increase of a move about 1, and multiplied by step_scale * sqrt(temperature).
    A is symmetric, so the axes come from ``eigh`` (the reference takes an SVD; the two agree up to the sign of each
    column, which a Gaussian candidate does not see)."""
    A = 0.5 * np.asarray(hessian, dtype=float)
    a, V = np.linalg.eigh(0.5 * (A + A.T))
    a = np.abs(a)                                   # singular values of a symmetric matrix
    c = cutoff * a.max()
    stiffness = np.maximum(a, max(c, np.finfo(float).tiny))
    n_eff = float(np.sum(np.minimum(a / c, 1.0))) if c > 0.0 else float(len(a))
    return (V / np.sqrt(stiffness)) * (step_scale * np.sqrt(temperature / n_eff))


def ensemble_log_params_batch(project, params, hess=None, steps=1000, temperature=1.0, step_scale=1.0,
                              sing_val_cutoff=0.0, seeds=None, skip_elems=0, energy='auto',
                              **integrator_overrides):
    """C Metropolis chains in log-parameter space, advanced together.

    params : (q,) start shared by all chains, or (C, q) one start per chain (``n_chains`` = C).
    hess   : (q, q) Hessian for the candidate density (default: J^T J at the first start).
    energy : 'free_energy' (rss - scale-factor entropy, needs a log prior on every scale factor, as the
             reference), 'rss' (0.5 |r|^2), or 'auto' (free energy when the priors are there).
    Returns (ens, ens_Fs, ratio): ens (n_kept, C, q) parameter sets including the starts, ens_Fs
    (n_kept, C) their energies, ratio (C,) accepted / attempted per chain.
    """
    starts = np.atleast_2d(np.asarray(params, dtype=float))
    C, q = starts.shape
    rng = np.random.default_rng(seeds)
    sfs = list(project.scale_factors.values()) if project.scale_factors is not None else []
    if energy == 'auto':
        energy = 'free_energy' if sfs and all(sf.log_prior is not None for sf in sfs) else 'rss'

    def F(th):
        if energy == 'free_energy':
            return project.free_energy_batch(th, temperature, **integrator_overrides)
        out = 0.5 * project.evaluate_batch(th, **integrator_overrides)['norms']
        return np.where(np.isfinite(out), out, np.inf)

    if hess is None:
        J = project.evaluate_batch(starts[:1], jacobian=True, want=('jacobian',), **integrator_overrides)['jacobian'][0]
        if project.reference_compat:
            J = J.copy()
            J[:project.n_project_residuals] /= project.descriptor_arrays()['row_sigma'][:, None]
        hess = J.T @ J
    samp = sampling_matrix(hess, sing_val_cutoff, temperature, step_scale)
    curr = starts.copy()
    curr_F = F(curr)
    ens, ens_F = [curr.copy()], [curr_F.copy()]
    accepted = np.zeros(C)
    for step in range(1, int(steps) + 1):
        trial = curr + rng.standard_normal((C, q)) @ samp.T          # _trial_move, one per chain
        next_F = F(trial)
        with np.errstate(over='ignore', invalid='ignore'):
            acc = rng.random(C) < np.exp(-(next_F - curr_F) / temperature)   # _accept_move
        acc &= np.isfinite(next_F)
        curr = np.where(acc[:, None], trial, curr)
        curr_F = np.where(acc, next_F, curr_F)
        accepted += acc
        if step % (skip_elems + 1) == 0:
            ens.append(curr.copy())
            ens_F.append(curr_F.copy())
    return np.stack(ens), np.stack(ens_F), accepted / max(int(steps), 1)
