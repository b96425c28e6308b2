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


def sampling_axes(hessian, cutoff=0.0, temperature=1.0, step_scale=1.0):
    """Principal axes V (q, q) and step lengths s (q,) of the candidate density: a move is V @ (s * z), z standard
    normal.  ``hessian`` may carry leading batch axes (one Hessian per chain).

    The recipe is SloppyCell's, as the reference carries it (_sampling_matrix, Ensembles.py:226-258): along each
    principal axis v_i of A = hessian / 2 (eigenvalue a_i) the step has standard deviation 1 / sqrt(max(a_i, c)) with
    c = cutoff * max a, so flat directions are not followed without bound; everything is then divided by
    sqrt(n_eff), n_eff = sum_i min(a_i / c, 1) (= the number of axes when nothing is clipped), which makes the
    expected quadratic cost increase of a move about 1, and multiplied by step_scale * sqrt(temperature).
    A is symmetric, so the axes come from ``eigh`` (the reference takes an SVD; the two agree up to the sign of each
    column, which a Gaussian candidate does not see)."""
    A = 0.5 * np.asarray(hessian, dtype=float)
    a, V = np.linalg.eigh(0.5 * (A + np.swapaxes(A, -1, -2)))
    a = np.abs(a)                                   # singular values of a symmetric matrix
    c = cutoff * a.max(axis=-1, keepdims=True)
    stiffness = np.maximum(a, np.maximum(c, np.finfo(float).tiny))
    with np.errstate(divide='ignore', invalid='ignore'):
        n_eff = np.where(c > 0.0, np.minimum(a / np.where(c > 0.0, c, 1.0), 1.0), 1.0).sum(axis=-1, keepdims=True)
    return V, step_scale * np.sqrt(temperature / n_eff) / np.sqrt(stiffness)


def sampling_matrix(hessian, cutoff=0.0, temperature=1.0, step_scale=1.0):
    """Candidate-move matrix M = V diag(s) of ``sampling_axes``: a move is M @ z, Gaussian with covariance M M^T."""
    V, s = sampling_axes(hessian, cutoff, temperature, step_scale)
    return V * s[..., None, :]


def _log_candidate_density(step, V, s):
    """log of the Gaussian density N(0, V diag(s^2) V^T) at ``step`` up to the constant (2 pi)^(-q/2), per chain:
    -0.5 step^T Sigma^-1 step - 0.5 log det Sigma  (reference _accept_move_recalc_alg, Ensembles.py:200-224)."""
    u = np.einsum('cji,cj->ci', V, step) / s       # coordinates along the axes, in units of their step lengths
    return -0.5 * np.sum(u * u, axis=1) - np.sum(np.log(s), axis=1)


def ensemble_log_params_batch(project, params, hess=None, steps=1000, temperature=1.0, step_scale=1.0,
                              sing_val_cutoff=0.0, seeds=None, skip_elems=0, energy='auto', recalc_hess_alg=False,
                              **integrator_overrides):
    """C Metropolis chains in log-parameter space, advanced together.

    params : (q,) start shared by all chains, or (C, q) one start per chain (``n_chains`` = C).
    hess   : (q, q) Hessian for the candidate density (default: J^T J at the first start).
    recalc_hess_alg : the reference's second algorithm (Ensembles.py:153-157, 200-224): every chain draws its
             candidate from the Gauss-Newton Hessian J^T J AT ITS CURRENT POINT and the move is accepted with the
             Metropolis-Hastings ratio pi(y) q(y -> x) / (pi(x) q(x -> y)) -- the candidate density now differs between
             the two ends of a move.  The Jacobians of all C trial points come from the same batched device call that
             integrates them.
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

    inv_sigma = None
    if project.reference_compat:        # J comes undivided by sigma there
        inv_sigma = 1.0 / project.descriptor_arrays()['row_sigma']

    def hessians(th):
        J = project.evaluate_batch(th, jacobian=True, want=('jacobian',), **integrator_overrides)['jacobian']
        if inv_sigma is not None:
            J = J.copy()
            J[:, :project.n_project_residuals] *= inv_sigma[None, :, None]
        return np.einsum('crj,crk->cjk', J, J)

    if hess is None and not recalc_hess_alg:
        hess = hessians(starts[:1])[0]
    curr = starts.copy()
    curr_F = F(curr)
    if recalc_hess_alg:
        V, sv = sampling_axes(hessians(curr) if hess is None else np.broadcast_to(hess, (C, q, q)),
                              sing_val_cutoff, temperature, step_scale)
    else:
        V1, s1 = sampling_axes(hess, sing_val_cutoff, temperature, step_scale)
        V, sv = np.broadcast_to(V1, (C, q, q)), np.broadcast_to(s1, (C, q))
    ens, ens_F = [curr.copy()], [curr_F.copy()]
    accepted = np.zeros(C)
    for step in range(1, int(steps) + 1):
        delta = np.einsum('cij,cj->ci', V, sv * rng.standard_normal((C, q)))     # _trial_move, one per chain
        trial = curr + delta
        next_F = F(trial)
        log_ratio = -(next_F - curr_F) / temperature                               # _accept_move
        if recalc_hess_alg:
            ok = np.isfinite(next_F)
            # chains whose trial point cannot be integrated are rejected anyway: their Hessian is not needed
            Vn, sn = sampling_axes(hessians(np.where(ok[:, None], trial, curr)), sing_val_cutoff, temperature, step_scale)
            log_ratio = log_ratio + _log_candidate_density(-delta, Vn, sn) - _log_candidate_density(delta, V, sv)
        with np.errstate(over='ignore', invalid='ignore'):
            acc = np.log(rng.random(C)) < log_ratio
        acc &= np.isfinite(next_F)
        curr = np.where(acc[:, None], trial, curr)
        curr_F = np.where(acc, next_F, curr_F)
        if recalc_hess_alg:
            V = np.where(acc[:, None, None], Vn, V)
            sv = np.where(acc[:, None], sn, sv)
        accepted += acc
        if step % (skip_elems + 1) == 0:
            ens.append(curr.copy())
            ens_F.append(curr_F.copy())
    return np.stack(ens), np.stack(ens_F), accepted / max(int(steps), 1)
