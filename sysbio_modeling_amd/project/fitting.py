"""Batched multi-start Levenberg-Marquardt on top of the device evaluator.

The reference fits one start at a time: ``scipy.optimize.leastsq(project.residuals, x0,
Dfun=project.calc_project_jacobian)`` (tests/test_Project.py:202-213, :352-357) -- every function and
Jacobian evaluation a serial LSODA run per experiment.  Here V starts advance together: one
``sbm_jacobian_batch`` call integrates all trial points (residuals AND Jacobians from the same augmented
integration), one ``sbm_lm_step`` call solves the V damped normal equations on the device, and the
accept / reject bookkeeping is a handful of tensor selects.  Nothing leaves the GPU inside the loop except
a convergence count per iteration.
"""
from __future__ import annotations


import numpy as np

from .. import _lib


def _trial_budget(project, th, integrator_overrides, n_steps_sum=None, status=None):
    """Step budget of the TRIAL integrations when the caller named none: three times what the slowest TRAJECTORY of the
    starting points needs (at least 300 attempts: DOP853 takes ~170 steps where DOPRI45 takes ~1000, and a floor of 2000 let its
    stiff trial points run twelve times the typical trajectory), with the early exit (negative ``max_steps``, include/sbm.h).  One
    launch lasts as long as its slowest trajectory, and an optimiser free to wander along unconstrained parameter
    directions finds regions where the model is stiff and a trajectory takes 20 times the usual steps: such a trial point
    is worth rejecting by its price alone -- the trust region then shrinks away from it.  (Measured on the sloppy
    configs[3] project, 256 starts x 100 iterations of the trust-region algorithm: 4.4 s with a budget of 20 000 attempts,
    1.7 s with 5 000, same costs.)

    The budget is per trajectory and so is the count it is derived from (``sbm_project_trajectory_steps``: round 2
    divided a vector's SUM over its experiments by their number -- with one slow experiment among eight the starts
    themselves ran out of budget).  The STARTING points are integrated with the caller's / the model's own budget
    before this is called (``n_steps_sum`` / ``status`` of that evaluation, if the caller has them); only trial points
    get the derived one.  Returns the budget set (None if the caller named one)."""
    import torch
    if 'max_steps' in integrator_overrides:
        return None
    if n_steps_sum is None:
        out = project.evaluate_batch(th, want=('n_steps',), **integrator_overrides)
        n_steps_sum, status = out['n_steps'], out['status']
    V = int(th.shape[0])
    E = max(1, len(project._experiments))
    worst = 0
    try:
        per = torch.empty((V, E), dtype=torch.int32, device=th.device)
        _lib.check(_lib.load_library().sbm_project_trajectory_steps(project._device(), V, _lib.dev_ptr(per)),
                   'sbm_project_trajectory_steps')
        ok = (status == 0) if status is not None else torch.ones((V,), dtype=torch.bool, device=th.device)
        if bool(ok.any()):
            worst = int(per[ok].max())
    except _lib.SbmError:
        worst = int(n_steps_sum.max()) if n_steps_sum.numel() else 0      # (an upper bound of every trajectory's count)
    # the same cost-quality trade-off for both pairs (scripts/dev_fit_budget.py, 256 starts x 100 iterations of the configs[3]
    # project: median cost 184.6 - 184.8 at 1.2 s with 3x for DOPRI45 (~3 900 attempts) and 6x for DOP853 (~1 200); half the
    # budget: 184.75 / 185.0 at 1.2 / 1.0 s, 185.6 at 0.8 s with 3x for DOP853; twice: 184.3 / 184.5 at 1.9 / 1.4 s)
    method = str(project._options(**integrator_overrides).get('method', 'dopri45')).lower()
    factor = 6.0 if method in _lib.DOP853 else 3.0
    budget = -max(300, int(factor * worst))
    integrator_overrides['max_steps'] = budget
    return budget


def levenberg_marquardt_batch(project, thetas0, max_iter=60, lambda0=1e-2, lambda_up=4.0, lambda_down=3.0,
                              ftol=1.49012e-8, xtol=1.49012e-8, max_step=2.0, trace=False, lazy_jacobian='auto',
                              algorithm='trust_region', factor=100.0, **integrator_overrides):
    """Minimise 0.5 |r(theta)|^2 from every row of ``thetas0`` (V, q), independently.

    ``algorithm='trust_region'`` (default since round 2): MINPACK's lmder, batched (`_trust_region_batch` below) -- the
    damping of every step comes from a scaled trust region whose radius follows the ratio of actual to predicted
    reduction; ``factor`` is lmder's initial step bound (leastsq's default 100).  On the sloppy 68-parameter configs[3]
    project, 64 starts: median cost 185.8 / 184.5 / 184.0 after 50 / 100 / 200 iterations against 187.4 / 187.0 / 186.8
    for ``algorithm='marquardt'`` (round 1's loop, kept), and where leastsq itself stands at 186.2 after 52 Jacobian
    evaluations of the same start.  ``algorithm='marquardt'``:

    Marquardt damping per start: a step is accepted when the cost decreases (lambda /= lambda_down),
    rejected otherwise (lambda *= lambda_up).  (Nielsen's gain-ratio update was tried and did no better on
    the sloppy 68-parameter test problem.)  A start has converged when a lightly damped (lambda <= 1)
    accepted step lowers the cost by less than ftol * cost with a predicted decrease just as small, or
    changes no parameter by more than xtol * (|theta| + xtol); the defaults are scipy.optimize.leastsq's (the reference's
    optimiser, tests/test_Project.py:202-213).
    Failed integrations (inf cost) count as rejections.  Two safeguards keep wild trial points from
    stalling the whole batch (one launch waits for its slowest trajectory): every component of a step is
    clipped to ``max_step`` log-units, and trial integrations get a step budget
    (``max_steps``; default: five times what the starting points need, `_trial_budget`) -- a trial that exhausts it is
    simply rejected.

    ``lazy_jacobian``: a trial point is first integrated WITHOUT sensitivities (the state-only kernels: a
    tenth of the cost) to get its residuals; the state + sensitivity system is integrated only at the trial points that
    were accepted -- MINPACK's economy (lmder evaluates the Jacobian once per successful step, the function once per
    trial), batched: about half of the trials are rejected on a sloppy problem.  ``False``: one state + sensitivity
    integration of every trial point per iteration.  'auto' (default) = ``False`` since round 3 (it was lazy from 8192
    trajectories per iteration on): a launch lasts as long as its slowest trajectory, so two launches per iteration cost
    more than one (256 starts x 8 experiments: 1.23 s eager, 1.74 s lazy per 100 iterations) -- and the cost the
    state-only kernel returns differs from the sensitivity kernel's by the integration tolerance (its steps are chosen by
    the state's error alone), which is what lmder's ftol = 1.5e-8 test resolves: with lazy trial points NO start is ever
    reported converged (costs and parameters are as good).  Pass ``lazy_jacobian=True`` together with looser
    ``ftol`` / tighter ``rtol`` where the saved sensitivity integrations matter.

    With a handful of starts the chip is mostly empty: ``variant='small_batch'`` (an integrator override) lets the
    sensitivity kernel use its small-batch split while starts x experiments x chunks <= 2048.

    Returns a dict of numpy arrays: theta (V, q), cost (V,) = 0.5 |r|^2, n_iter (V,) iterations until
    convergence (max_iter if never), converged (V,) bool, n_evaluations (total trial points integrated),
    n_jacobian_evaluations (those integrated with sensitivities); with
    ``trace=True`` also 'history': per iteration the number of accepted steps, starts still running, median cost,
    damping, relative decrease and largest step component (costs a device synchronisation per iteration).
    """
    import torch
    if algorithm in ('trust_region', 'lmder', 'minpack', 'trust_region_torch'):
        # the loop's bookkeeping in two device launches per iteration (sbm_lm_update / sbm_lm_accept) whenever the
        # integration is ONE device call; the control loops of _control.py (method='auto', 'implicit_romberg') and
        # algorithm='trust_region_torch' (round 2's spelling in tensor selects, kept as the cross-check) take the other
        from .. import _control
        o = project._options(**integrator_overrides)
        one_call = str(o.get('method', 'dopri45')).lower() not in _control.IMPLICIT_CONTROLLED + _control.AUTO
        fn = _trust_region_fused if (one_call and algorithm != 'trust_region_torch') else _trust_region_batch
        return fn(project, thetas0, max_iter=max_iter, ftol=ftol, xtol=xtol, factor=factor, trace=trace,
                  lazy_jacobian=lazy_jacobian, max_step=max_step, **integrator_overrides)
    if algorithm != 'marquardt':
        raise ValueError("fit_batch: unknown algorithm %r ('marquardt', 'trust_region' or 'trust_region_torch')" % (algorithm,))
    if lazy_jacobian == 'auto':
        lazy_jacobian = False        # (see levenberg_marquardt_batch: lazy is opt-in since round 3)
    if project.reference_compat and project.n_total_rows != project.n_project_residuals:
        raise ValueError("fit_batch needs reference_compat=False when priors are set: the reference leaves the "
                         "prior rows of the Jacobian zero (SURVEY.md section 8a, quirk 4)")
    lib = _lib.load_library()
    ctx = project._model.device_model.ctx      # the context (device, stream) the project's model lives on
    th, _ = project._theta_dev(np.asarray(thetas0, dtype=np.float64) if not hasattr(thetas0, 'device') else thetas0)
    th = th.clone()
    dev = th.device
    V, q = th.shape
    f64, i32 = torch.float64, torch.int32
    want = ('jacobian',)
    # reference_compat leaves J undivided by sigma (quirk 3): divide here, the gradient needs d r / d theta
    inv_sigma = None
    if project.reference_compat:
        a = project.descriptor_arrays()
        inv_sigma = torch.from_numpy(1.0 / a['row_sigma']).to(dev)

    last = {}

    def evaluate(t):
        out = project.evaluate_batch(t, jacobian=True, want=want, **integrator_overrides)
        last['n_steps'], last['status'] = out['n_steps'], out['status']
        J = out['jacobian']
        if inv_sigma is not None:
            J = J * inv_sigma[None, :, None]
        cost = 0.5 * out['norms']
        cost = torch.where(torch.isfinite(cost) & (out['status'] == 0), cost, torch.full_like(cost, float('inf')))
        return out['residuals'], J, cost

    r, J, cost = evaluate(th)            # the starting points: the caller's / the model's own step budget
    _trial_budget(project, th, integrator_overrides, n_steps_sum=last['n_steps'], status=last['status'])
    M = r.shape[1]
    lam = torch.full((V,), float(lambda0), dtype=f64, device=dev)
    delta = torch.empty((V, q), dtype=f64, device=dev)
    pred = torch.empty((V,), dtype=f64, device=dev)
    st = torch.empty((V,), dtype=i32, device=dev)
    done = ~torch.isfinite(cost)                      # a start that cannot be integrated stays where it is
    n_iter = torch.full((V,), int(max_iter), dtype=torch.int64, device=dev)
    n_eval = V
    n_jac = V                   # state + sensitivity integrations among them
    history = []
    p = _lib.dev_ptr
    for it in range(max_iter):
        # finite stand-ins where the current point is unusable (their steps are discarded below)
        Jc = torch.where(torch.isfinite(cost)[:, None, None], J, torch.zeros_like(J)).contiguous()
        rc = torch.where(torch.isfinite(cost)[:, None], r, torch.zeros_like(r)).contiguous()
        _lib.check(lib.sbm_lm_step(ctx.handle, p(Jc), p(rc), p(lam), V, M, q, p(delta), p(pred), p(st)), 'sbm_lm_step')
        # per parameter, not by shrinking the whole step: a parameter the data barely constrain gets an
        # enormous (and harmless) Gauss-Newton step, which must not scale everybody else's down to nothing
        delta = delta.clamp(-max_step, max_step)
        trial = torch.where(done[:, None], th, th + delta)
        if lazy_jacobian:
            out_t = project.evaluate_batch(trial, **integrator_overrides)
            cost_t = 0.5 * out_t['norms']
            cost_t = torch.where(torch.isfinite(cost_t) & (out_t['status'] == 0), cost_t, torch.full_like(cost_t, float('inf')))
        else:
            r_t, J_t, cost_t = evaluate(trial)
        n_eval += V
        ok = (st == 0) & (cost_t < cost) & ~done
        # MINPACK-style tests, on lightly damped steps only: a heavily damped step is short whatever the
        # distance to the optimum
        free = lam <= 1.0
        small_f = ok & free & ((cost - cost_t) <= ftol * cost) & (pred <= ftol * cost)
        # a negligible step ends the search whether or not it still lowers the cost (at the rounding floor it does not)
        small_x = (st == 0) & ~done & free & (delta.abs() <= xtol * (th.abs() + xtol)).all(dim=1)
        cost_prev = cost
        if lazy_jacobian:
            # residuals and Jacobian of the accepted points, from ONE integration each (so that J^T r is consistent)
            sel = torch.nonzero(ok).flatten()
            if sel.numel():
                r_a, J_a, cost_a = evaluate(trial[sel])
                n_jac += int(sel.numel())
                # (the state-only and the state + sensitivity integrations agree to the integrator's tolerance, not to
                # the bit: a point whose second integration fails or no longer improves stays where it was)
                good = torch.isfinite(cost_a) & (cost_a < cost[sel])
                sel, r_a, J_a, cost_a = sel[good], r_a[good], J_a[good], cost_a[good]
                th[sel] = trial[sel]
                r[sel] = r_a
                J[sel] = J_a
                cost = cost.clone()
                cost[sel] = cost_a
                ok = torch.zeros_like(ok)
                ok[sel] = True
            else:
                ok = torch.zeros_like(ok)
        else:
            th = torch.where(ok[:, None], trial, th)
            r = torch.where(ok[:, None], r_t, r)
            J = torch.where(ok[:, None, None], J_t, J)
            cost = torch.where(ok, cost_t, cost)
            n_jac += V
        lam = torch.where(ok, lam / lambda_down, lam * lambda_up).clamp(1e-15, 1e15)
        newly = (small_f | small_x) & ~done
        if trace:
            live = ~done
            history.append(dict(iteration=it, accepted=int((ok & live).sum()), live=int(live.sum()),
                                cost_median=float(cost.median()), lambda_median=float(lam[live].median()) if bool(live.any()) else 0.0,
                                rel_decrease_median=float(((cost_prev - cost) / cost_prev)[live].median()) if bool(live.any()) else 0.0,
                                step_max_median=float(delta.abs().max(dim=1).values[live].median()) if bool(live.any()) else 0.0))
        n_iter = torch.where(newly, torch.full_like(n_iter, it + 1), n_iter)
        done = done | newly
        if bool(done.all()):
            break
    return {'theta': th.cpu().numpy(), 'cost': cost.cpu().numpy(), 'n_iter': n_iter.cpu().numpy(),
            'converged': (done & torch.isfinite(cost)).cpu().numpy(), 'n_evaluations': n_eval,
            'n_jacobian_evaluations': n_jac,
            **({'history': history} if trace else {})}


def _trust_region_batch(project, thetas0, max_iter=60, ftol=1.49012e-8, xtol=1.49012e-8, factor=100.0, trace=False,
                        lazy_jacobian='auto', max_step=2.0, **integrator_overrides):
    """MINPACK's lmder for V starts at once -- the optimiser behind the reference's ``scipy.optimize.leastsq`` calls
    (tests/test_Project.py:202-213, 351-357), with its bookkeeping as tensor selects and its inner problem (lmpar: the
    Levenberg-Marquardt parameter of a scaled trust region) solved per start on the device (``sbm_lm_trust_step``, one
    launch per iteration).

    Per start: D = the largest column norm of J seen so far; the step solves (J^T J + lambda D^2) delta = -J^T r with
    ||D delta|| within 10 % of the radius Delta (lambda = 0 if the Gauss-Newton step is inside); ratio = actual / predicted
    reduction of |r|^2; ratio <= 1/4 shrinks Delta (by 1/2, or by the parabola-fit factor down to 1/10 when the cost went
    up), ratio >= 3/4 or lambda = 0 sets Delta = 2 ||D delta||; the step is taken when ratio >= 1e-4.  Converged (lmder's
    info 1 / 2): actual and predicted relative reductions both <= ftol with ratio <= 2, or Delta <= xtol ||D theta||.
    Trial points are integrated as in the Marquardt loop (``lazy_jacobian``, the step budget with early exit); a trial
    that cannot be integrated counts as an increase of the cost.  ``max_step`` only keeps exp(theta) finite.
    """
    import torch
    if project.reference_compat and project.n_total_rows != project.n_project_residuals:
        raise ValueError("fit_batch needs reference_compat=False when priors are set: the reference leaves the "
                         "prior rows of the Jacobian zero (SURVEY.md section 8a, quirk 4)")
    if lazy_jacobian == 'auto':
        lazy_jacobian = False        # (see levenberg_marquardt_batch: lazy is opt-in since round 3)
    lib = _lib.load_library()
    ctx = project._model.device_model.ctx
    th, _ = project._theta_dev(np.asarray(thetas0, dtype=np.float64) if not hasattr(thetas0, 'device') else thetas0)
    th = th.clone()
    dev = th.device
    V, q = th.shape
    f64, i32 = torch.float64, torch.int32
    inv_sigma = None
    if project.reference_compat:
        inv_sigma = torch.from_numpy(1.0 / project.descriptor_arrays()['row_sigma']).to(dev)

    last = {}

    def evaluate(t):
        out = project.evaluate_batch(t, jacobian=True, want=('jacobian',), **integrator_overrides)
        last['n_steps'], last['status'] = out['n_steps'], out['status']
        J = out['jacobian']
        if inv_sigma is not None:
            J = J * inv_sigma[None, :, None]
        c = 0.5 * out['norms']
        c = torch.where(torch.isfinite(c) & (out['status'] == 0), c, torch.full_like(c, float('inf')))
        return out['residuals'], J, c

    def cost_only(t):
        out = project.evaluate_batch(t, **integrator_overrides)
        c = 0.5 * out['norms']
        return torch.where(torch.isfinite(c) & (out['status'] == 0), c, torch.full_like(c, float('inf')))

    r, J, cost = evaluate(th)            # the starting points: the caller's / the model's own step budget
    if not bool(torch.isfinite(cost).all()):
        import warnings
        warnings.warn("fit_batch: %d of %d starting points could not be integrated; they are returned as they came"
                      % (int((~torch.isfinite(cost)).sum()), V))
    _trial_budget(project, th, integrator_overrides, n_steps_sum=last['n_steps'], status=last['status'])
    M = r.shape[1]
    done = ~torch.isfinite(cost)
    dscale = torch.zeros((V, q), dtype=f64, device=dev)
    lam = torch.zeros((V,), dtype=f64, device=dev)
    radius = torch.full((V,), 1.0, dtype=f64, device=dev)        # set from ||D theta|| once D exists (first pass)
    delta = torch.empty((V, q), dtype=f64, device=dev)
    pred = torch.empty((V,), dtype=f64, device=dev)
    dxnorm = torch.empty((V,), dtype=f64, device=dev)
    st = torch.empty((V,), dtype=i32, device=dev)
    n_iter = torch.full((V,), int(max_iter), dtype=torch.int64, device=dev)
    n_eval, n_jac = V, V
    history = []
    p = _lib.dev_ptr
    first = True
    for it in range(max_iter):
        Jc = torch.where(torch.isfinite(cost)[:, None, None], J, torch.zeros_like(J)).contiguous()
        rc = torch.where(torch.isfinite(cost)[:, None], r, torch.zeros_like(r)).contiguous()
        if first:
            # lmder: D from the first Jacobian, Delta = factor * ||D theta|| (factor itself where that is zero)
            col = torch.sqrt((Jc * Jc).sum(dim=1))
            d0 = torch.where(col > 0, col, torch.ones_like(col))
            xn = (d0 * th).norm(dim=1)
            radius = torch.where(xn > 0, factor * xn, torch.full_like(xn, float(factor)))
        _lib.check(lib.sbm_lm_trust_step(ctx.handle, p(Jc), p(rc), p(dscale), p(radius), p(lam), V, M, q, p(delta), p(pred),
                                         p(dxnorm), p(st)), 'sbm_lm_trust_step')
        step = delta.clamp(-max_step, max_step)
        clipped = (step != delta).any(dim=1)
        if bool(clipped.any()):
            # the quantities of the step TAKEN (round 2 judged the clipped step by the unclipped one's prediction):
            # pred = -g.x - x^T J^T J x / 2, ||D x||
            Jx = torch.einsum('vmq,vq->vm', Jc, step)
            gx = torch.einsum('vm,vm->v', rc, Jx)
            pred = torch.where(clipped, -gx - 0.5 * (Jx * Jx).sum(dim=1), pred)
            dxnorm = torch.where(clipped, (dscale * step).norm(dim=1), dxnorm)
        if first:
            # lmder: on the first iteration, Delta = min(Delta, ||D p||) -- of the step taken (the clipped one)
            radius = torch.where(st == 0, torch.minimum(radius, dxnorm), radius)
            first = False
        trial = torch.where(done[:, None], th, th + step)
        if lazy_jacobian:
            cost_t = cost_only(trial)
        else:
            r_t, J_t, cost_t = evaluate(trial)
        n_eval += V
        usable = (st == 0) & ~done
        # lmder's quantities, relative to |r|^2 = 2 cost:  prered = (|J p|^2 + 2 lam |D p|^2) / |r|^2 = pred / cost
        safe_cost = torch.where(cost > 0, cost, torch.ones_like(cost))
        actred = torch.where(0.1 * torch.sqrt(cost_t) < torch.sqrt(cost), 1.0 - cost_t / safe_cost, -torch.ones_like(cost))
        actred = torch.where(torch.isfinite(cost_t), actred, -torch.ones_like(cost))
        prered = pred / safe_cost
        # dirder = g . p / |r|^2  (= -(|J p|^2 + lam |D p|^2) / |r|^2 for an unclipped step: lmder's expression)
        dirder = torch.einsum('vm,vm->v', rc, torch.einsum('vmq,vq->vm', Jc, step)) / (2.0 * safe_cost)
        ratio = torch.where(prered > 0, actred / torch.where(prered > 0, prered, torch.ones_like(prered)), torch.zeros_like(prered))
        # radius update
        shrink = ratio <= 0.25
        temp = torch.where(actred >= 0, torch.full_like(actred, 0.5),
                           0.5 * dirder / torch.where((dirder + 0.5 * actred) != 0, dirder + 0.5 * actred, -torch.ones_like(dirder)))
        worse10 = ~(0.1 * torch.sqrt(cost_t) < torch.sqrt(cost)) | ~torch.isfinite(cost_t)
        temp = torch.where(worse10 | (temp < 0.1) | ~torch.isfinite(temp), torch.full_like(temp, 0.1), temp)
        new_radius = torch.where(shrink, temp * torch.minimum(radius, dxnorm / 0.1),
                                 torch.where((lam == 0) | (ratio >= 0.75), dxnorm / 0.5, radius))
        new_lam = torch.where(shrink, lam / temp, torch.where((lam == 0) | (ratio >= 0.75), 0.5 * lam, lam))
        # a system that could not be solved (status 1): halve the radius, keep the point
        new_radius = torch.where(st == 0, new_radius, 0.5 * radius)
        radius = torch.where(done, radius, new_radius)
        lam = torch.where(done | (st != 0), lam, new_lam)
        ok = usable & (ratio >= 1.0e-4) & torch.isfinite(cost_t)
        cost_prev = cost
        th_before = th.clone() if lazy_jacobian else th
        if lazy_jacobian:
            sel = torch.nonzero(ok).flatten()
            ok = torch.zeros_like(ok)
            if sel.numel():
                r_a, J_a, cost_a = evaluate(trial[sel])
                n_jac += int(sel.numel())
                good = torch.isfinite(cost_a)
                sel, r_a, J_a, cost_a = sel[good], r_a[good], J_a[good], cost_a[good]
                th[sel] = trial[sel]
                r[sel] = r_a
                J[sel] = J_a
                cost = cost.clone()
                cost[sel] = cost_a
                ok[sel] = True
        else:
            th = torch.where(ok[:, None], trial, th)
            r = torch.where(ok[:, None], r_t, r)
            J = torch.where(ok[:, None, None], J_t, J)
            cost = torch.where(ok, cost_t, cost)
            n_jac += V
        # lmder's convergence tests (info 1, 2); ||D theta|| of the point the iteration started from, as sbm_lm_update has it
        xnorm = (dscale * th_before).norm(dim=1)
        conv_f = usable & (actred.abs() <= ftol) & (prered <= ftol) & (0.5 * ratio <= 1.0)
        conv_x = usable & (radius <= xtol * xnorm)
        newly = (conv_f | conv_x) & ~done
        if trace:
            live = ~done
            md = lambda t: float(t[live].median()) if bool(live.any()) else 0.0
            history.append(dict(iteration=it, accepted=int((ok & live).sum()), live=int(live.sum()), cost_median=float(cost.median()),
                                lambda_median=md(lam), radius_median=md(radius), ratio_median=md(ratio),
                                rel_decrease_median=md((cost_prev - cost) / cost_prev)))
        n_iter = torch.where(newly, torch.full_like(n_iter, it + 1), n_iter)
        done = done | newly
        if bool(done.all()):
            break
    return {'theta': th.cpu().numpy(), 'cost': cost.cpu().numpy(), 'n_iter': n_iter.cpu().numpy(),
            'converged': (done & torch.isfinite(cost)).cpu().numpy(), 'n_evaluations': n_eval,
            'n_jacobian_evaluations': n_jac, **({'history': history} if trace else {})}


def _trust_region_fused(project, thetas0, max_iter=60, ftol=1.49012e-8, xtol=1.49012e-8, factor=100.0, trace=False,
                        lazy_jacobian='auto', max_step=2.0, **integrator_overrides):
    """`_trust_region_batch` with the bookkeeping on the device: per iteration ONE ``sbm_lm_trust_step_ex`` (lmpar, the
    clipped step and the trial point), the integration of the trial points (``sbm_jacobian_batch`` /
    ``sbm_residuals_batch`` into preallocated buffers), ONE ``sbm_lm_update`` (lmder's ratio / radius / acceptance /
    convergence logic) and ONE ``sbm_lm_accept`` (the accepted points' theta, r, J, cost) -- about eight launches and one
    4-byte read-back where round 2's loop issued ~70 tensor selects (68 000 micro-launches in a 100-iteration fit,
    profiles/r02/fit_kernel_stats.csv).  Same algorithm, same numbers up to the order of a few sums."""
    import ctypes
    import warnings
    import torch
    if project.reference_compat and project.n_total_rows != project.n_project_residuals:
        raise ValueError("fit_batch needs reference_compat=False when priors are set: the reference leaves the "
                         "prior rows of the Jacobian zero (SURVEY.md section 8a, quirk 4)")
    if lazy_jacobian == 'auto':
        lazy_jacobian = False        # (see levenberg_marquardt_batch: lazy is opt-in since round 3)
    lib = _lib.load_library()
    ctx = project._model.device_model.ctx
    th, _ = project._theta_dev(np.asarray(thetas0, dtype=np.float64) if not hasattr(thetas0, 'device') else thetas0)
    th = th.clone()
    dev = th.device
    V, q = th.shape
    proj = project._device()
    R, M = project._n_residuals, project.n_total_rows
    G = len(project._loss_function.groups) if hasattr(project._loss_function, 'groups') else 0
    f64, i32 = torch.float64, torch.int32
    p = _lib.dev_ptr

    def buffers(n, with_J=True):
        b = dict(sims=torch.empty((n, R), dtype=f64, device=dev), r=torch.empty((n, M), dtype=f64, device=dev),
                 sf=torch.empty((n, max(G, 1)), dtype=f64, device=dev), norms=torch.empty((n,), dtype=f64, device=dev),
                 status=torch.empty((n,), dtype=i32, device=dev), nsteps=torch.empty((n,), dtype=i32, device=dev))
        if with_J:
            b['J'] = torch.empty((n, M, q), dtype=f64, device=dev)
        return b

    def integrate(theta_t, opts, b, jac):
        n = theta_t.shape[0]
        if jac:
            _lib.check(lib.sbm_jacobian_batch(proj, p(theta_t), n, ctypes.byref(opts), p(b['sims']), p(b['r']), p(b['J']), None,
                                              p(b['sf']) if G else None, None, p(b['norms']), None, p(b['status']),
                                              p(b['nsteps'])), 'sbm_jacobian_batch')
        else:
            _lib.check(lib.sbm_residuals_batch(proj, p(theta_t), n, ctypes.byref(opts), p(b['sims']), p(b['r']),
                                               p(b['sf']) if G else None, p(b['norms']), p(b['status']), p(b['nsteps'])),
                       'sbm_residuals_batch')

    cur, tr = buffers(V), buffers(V)
    integrate(th, project._opts(**integrator_overrides), cur, True)      # the starts: the caller's / the model's own budget
    cost = torch.where(torch.isfinite(cur['norms']) & (cur['status'] == 0), 0.5 * cur['norms'],
                       torch.full_like(cur['norms'], float('inf')))
    bad = ~torch.isfinite(cost)
    if bool(bad.any()):
        warnings.warn("fit_batch: %d of %d starting points could not be integrated; they are returned as they came"
                      % (int(bad.sum()), V))
        cur['J'][bad] = 0.0
        cur['r'][bad] = 0.0
    _trial_budget(project, th, integrator_overrides, n_steps_sum=cur['nsteps'], status=cur['status'])
    opts_t = project._opts(**integrator_overrides)
    row_scale = None
    if project.reference_compat:
        row_scale = torch.from_numpy(1.0 / project.descriptor_arrays()['row_sigma']).to(dev).contiguous()
    done = bad.to(i32)
    dscale = torch.zeros((V, q), dtype=f64, device=dev)
    lam = torch.zeros((V,), dtype=f64, device=dev)
    # lmder: D from the first Jacobian, Delta = factor * ||D theta|| (factor itself where that is zero)
    Js = cur['J'] if row_scale is None else cur['J'] * row_scale[None, :, None]
    col = torch.sqrt((Js * Js).sum(dim=1))
    del Js
    xn = (torch.where(col > 0, col, torch.ones_like(col)) * th).norm(dim=1)
    radius = torch.where(xn > 0, factor * xn, torch.full_like(xn, float(factor))).contiguous()
    delta = torch.empty((V, q), dtype=f64, device=dev)
    trial = torch.empty((V, q), dtype=f64, device=dev)
    pred, dxnorm, gtx = (torch.empty((V,), dtype=f64, device=dev) for _ in range(3))
    ratio = torch.empty((V,), dtype=f64, device=dev) if trace else None
    st = torch.empty((V,), dtype=i32, device=dev)
    accept = torch.zeros((V,), dtype=i32, device=dev)
    n_iter = torch.full((V,), int(max_iter), dtype=i32, device=dev)
    counters = torch.zeros((2,), dtype=i32, device=dev)
    n_eval, n_jac = V, V
    history = []
    for it in range(max_iter):
        _lib.check(lib.sbm_lm_trust_step_ex(ctx.handle, p(cur['J']), p(cur['r']), p(dscale), p(radius), p(lam), V, M, q,
                                            p(row_scale), p(done), float(max_step), p(th), p(trial), p(delta), p(pred),
                                            p(dxnorm), p(gtx), p(st)), 'sbm_lm_trust_step_ex')
        if trace:
            import time
            torch.cuda.synchronize(dev)
            t_launch = time.perf_counter()
        integrate(trial, opts_t, tr, not lazy_jacobian)
        n_eval += V
        if trace:
            # what the trial launch cost and why: it lasts as long as its slowest trajectory (scripts/dev_fit_trace.py)
            torch.cuda.synchronize(dev)
            t_launch = time.perf_counter() - t_launch
            per = torch.empty((V, max(1, len(project._experiments))), dtype=i32, device=dev)
            _lib.check(lib.sbm_project_trajectory_steps(proj, V, p(per)), 'sbm_project_trajectory_steps')
            launch = dict(ms=1e3 * t_launch, steps_max=int(per.max()), steps_mean=float(per.double().mean()),
                          steps_p99=float(per.double().flatten().quantile(0.99)), failed=int((tr['status'] != 0).sum()),
                          done=int((done != 0).sum()))
        cost_prev = cost.clone() if trace else None
        _lib.check(lib.sbm_lm_update(ctx.handle, p(cost), p(tr['norms']), p(tr['status']), p(pred), p(dxnorm), p(gtx), p(st),
                                     p(th), p(dscale), V, q, float(ftol), float(xtol), it, 1 if it == 0 else 0, p(radius),
                                     p(lam), p(done), p(accept), p(n_iter), p(counters), p(ratio)), 'sbm_lm_update')
        veto_live = 0
        if lazy_jacobian:
            # MINPACK's economy: the state + sensitivity system only at the trial points that were accepted
            sel = torch.nonzero(accept).flatten()
            if sel.numel():
                sub = buffers(int(sel.numel()))
                integrate(trial[sel].contiguous(), opts_t, sub, True)
                n_jac += int(sel.numel())
                good = torch.isfinite(sub['norms']) & (sub['status'] == 0)
                vetoed = sel[~good]
                accept[vetoed] = 0
                # sbm_lm_update ran BEFORE this re-integration: a start whose accepted step is vetoed here (the sensitivity
                # system could not be integrated at the trial point) has not moved, so it has not converged on that step
                # either -- it goes on from where it was
                veto_live = int(vetoed.numel())
                if veto_live:
                    done[vetoed] = 0
                    n_iter[vetoed] = int(max_iter)
                tr['r'][sel] = sub['r']
                tr['norms'][sel] = sub['norms']
                if 'J' not in tr:
                    tr['J'] = torch.empty((V, M, q), dtype=f64, device=dev)
                tr['J'][sel] = sub['J']
            elif 'J' not in tr:
                tr['J'] = torch.empty((V, M, q), dtype=f64, device=dev)
        else:
            n_jac += V
        _lib.check(lib.sbm_lm_accept(ctx.handle, p(accept), V, M, q, p(trial), p(tr['r']), p(tr['J']), p(tr['norms']), p(th),
                                     p(cur['r']), p(cur['J']), p(cost)), 'sbm_lm_accept')
        live, n_acc = (int(x) for x in counters.cpu())          # (the one read-back of the iteration)
        if lazy_jacobian and veto_live:
            live, n_acc = int((done == 0).sum()), n_acc - veto_live
        if trace:
            running = done == 0
            md = lambda t: float(t[running].median()) if bool(running.any()) else 0.0     # noqa: E731
            history.append(dict(iteration=it, accepted=n_acc, live=live, launch=launch, cost_median=float(cost.median()),
                                lambda_median=md(lam), radius_median=md(radius), ratio_median=md(ratio),
                                rel_decrease_median=md((cost_prev - cost) / cost_prev)))
        if live == 0:
            break
    torch.cuda.synchronize(dev)
    return {'theta': th.cpu().numpy(), 'cost': cost.cpu().numpy(), 'n_iter': n_iter.cpu().numpy().astype(np.int64),
            'converged': ((done != 0) & torch.isfinite(cost)).cpu().numpy(), 'n_evaluations': n_eval,
            'n_jacobian_evaluations': n_jac, **({'history': history} if trace else {})}
