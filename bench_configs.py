"""Side workloads of bench.py: BASELINE.json's configs[1], [3], [4], the end-to-end fitting figure and the kernel
variants.  Every runner returns one dict with the GPU timing (HIP events on the context's stream = torch's
current stream), a `roofline` object for its dominant kernel, and -- with ``cpu=True`` -- a `cpu_baseline` object:
the oracle (SciPy restatement of the reference) timed on one host core over a bounded sample of the same workload.

Runner signature: run(model, gm, dev, reps, cpu) with (model, gm) the cascade20 OdeModel / GeneratedModel.
"""
import time
import warnings

import numpy as np

import bench as B


def _events(torch, dev, fn, reps, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / reps


# ---------------------------------------------------------------------------
# configs[1]: 20-state model, 4096 vectors, no sensitivities
# ---------------------------------------------------------------------------
def run_configs1(model, gm, dev, reps=3, cpu=True):
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    V = 4096
    theta, P = models_zoo.cascade_ensemble(V)
    dm = model.device_model
    Pd = torch.from_numpy(P).to(dev)
    grid = np.linspace(0, models_zoo.CASCADE_T_END, 1000)
    tg = torch.from_numpy(np.concatenate([[0.0], grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]])).to(dev)
    Y = torch.empty((V, len(tg), 20), dtype=torch.float64, device=dev)
    ns = torch.empty((V,), dtype=torch.int32, device=dev)
    st = torch.empty((V,), dtype=torch.int32, device=dev)
    out = {"workload": "configs[1]: cascade20, 4096 parameter vectors, state only (20 ODEs), 16 output times"}
    for label, o, bytes_note in (
            ('dopri45', _lib.make_opts('dopri45', **{k: model.integrator_options[k] for k in ('rtol', 'atol')}),
             'DOPRI45 at the default tolerances rtol=%g atol=%g' % (model.integrator_options['rtol'], model.integrator_options['atol'])),
            ('rk4_fixed_4096', _lib.make_opts('rk4', n_steps=4096, t_end=100.0), 'RK4, 4096 fixed steps')):
        ms = _events(torch, dev, lambda: dm.simulate_dev(Pd, tg, None, o, Y, st, ns, None), reps)
        steps = int(ns.sum().item())
        r = {"ms": ms, "steps": steps, "value": steps / (ms * 1e-3), "unit": "ODE-steps/s", "integrator": bytes_note,
             "failed_vectors": int((st != 0).sum().item()),
             "roofline": B.hbm_roofline("sbm_state_packed_kernel<cascade20,%s>" % label, 'state_packed_cascade20_' + label,
                                        ms, steps, 2 * 8 * 20,
                                        "SURVEY.md section 8(d): 320 B per accepted step (read + write 20 doubles); the "
                                        "state lives in registers, real traffic is the parameter load and 17 output rows")}
        rv = B.valu_roofline('state_packed_cascade20_' + label, ms, steps)
        if rv:
            r["roofline_valu_issue"] = rv
        out[label] = r
    if cpu:
        out["cpu_baseline"] = B.cpu_baseline(gm, theta, budget_s=5.0, sens=False)
    return out


# ---------------------------------------------------------------------------
# configs[3]: 8 experiment settings x 1024 vectors, residual + Jacobian assembly
# ---------------------------------------------------------------------------
def _config3(model):
    from sysbio_modeling_amd import models_zoo
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        p4, th4 = models_zoo.cascade_config4_project(model)
    return p4, th4, models_zoo.config4_ensemble(th4, 1024)


def run_configs3(model, gm, dev, reps=3, cpu=True):
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    p4, th4, thetas = _config3(model)
    t4 = torch.from_numpy(thetas).to(dev)
    V, E = thetas.shape[0], 8
    holder = {}

    def go():
        holder['o'] = p4.evaluate_batch(t4, jacobian=True, want=('jacobian',))
    ms = _events(torch, dev, go, reps)
    o4 = holder['o']
    steps = int(o4['n_steps'].sum().item())
    # the integrator kernel of that pass alone: the same V*E trajectories through sbm_sens_batch
    a = p4.descriptor_arrays()
    pmap, pfixed = np.asarray(a['pmap']), np.asarray(a['pfixed'])
    Pall = np.where(pmap[None] >= 0, np.exp(thetas[:, np.maximum(pmap, 0)]), pfixed[None])      # (V, E, n_par)
    Pd = torch.from_numpy(np.ascontiguousarray(Pall.reshape(V * E, -1))).to(dev)
    grid = np.linspace(0, models_zoo.CASCADE_T_END, 1000)
    tg = torch.from_numpy(np.concatenate([[0.0], grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]])).to(dev)
    dm = model.device_model
    Yk = torch.empty((V * E, len(tg), 20), dtype=torch.float64, device=dev)
    Sk = torch.empty((V * E, len(tg), 20, 40), dtype=torch.float64, device=dev)
    nk = torch.empty((V * E,), dtype=torch.int32, device=dev)
    opts = _lib.make_opts('dopri45', **{k: model.integrator_options[k] for k in ('rtol', 'atol')})   # as the pass above
    k_ms = _events(torch, dev, lambda: dm.sens_dev(Pd, tg, None, opts, Yk, Sk, None, nk, None), reps)
    k_steps = int(nk.sum().item())
    out = {"workload": "configs[3]: 8 experiment settings x 1024 vectors (8192 trajectories of 820 ODEs), residuals + "
                       "Jacobian: 512 rows x 68 parameters per vector, 4 scale factors; on ONE GPU (the driver's "
                       "multi-GPU run shards the vector axis of the headline workload the same way)",
           "ms": ms, "steps": steps, "value": steps / (ms * 1e-3), "unit": "ODE-steps/s", "rows": 512, "params": 68,
           "failed_vectors": int((o4['status'] != 0).sum().item()),
           "roofline": B.hbm_roofline("sbm_sens_rowgroup_kernel<cascade20,dopri45>", 'sens_rowgroup_cascade20_dopri45',
                                      k_ms, k_steps, B.BYTES_PER_STEP,
                                      "the pass's integrator launch alone (8192 trajectories), 13 120 B per accepted step"),
           "integrator_share_of_pass": k_ms / ms}
    rv = B.valu_roofline('sens_rowgroup_cascade20_dopri45', k_ms, k_steps)
    if rv:
        out["roofline_valu_issue"] = rv
    if cpu:
        from oracle import tolerances as tol
        po = B.project_oracle_of(gm, p4)
        # parity of the timed pass on two vectors, then the timed sample
        R = o4['residuals'].cpu().numpy()
        J = o4['jacobian'].cpu().numpy()
        worst = [0.0, 0.0]
        for v in (0, V - 1):
            rr, sims, Bf = po.residuals(thetas[v], return_parts=True)
            Jm = po.model_jacobian(thetas[v])
            Jr = po.calc_project_jacobian(thetas[v])
            ts, tj = tol.lsoda_taus(a, thetas[v], sims, Jm)
            t = tol.project_tolerances(a, sims, Bf, ts, Jm, tj)
            worst[0] = max(worst[0], tol.tol_ratio(R[v], rr, t['residuals']))
            worst[1] = max(worst[1], tol.tol_ratio(J[v], Jr, t['jacobian']))
        po.lsoda_steps = 0
        n, t0 = 0, time.perf_counter()
        for v in range(1, V - 1):
            po.calc_project_jacobian(thetas[v])
            n += 1
            if time.perf_counter() - t0 > 10.0:
                break
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": po.lsoda_steps / dt, "unit": "ODE-steps/s", "cores": 1, "kind": "port",
            "sample": "%d project vectors (8 experiments each): ProjectOracle.calc_project_jacobian = odeint "
                      "rtol=atol=1e-10 on the 1000-point grid per experiment + numpy assembly; %.1f s, %d LSODA steps"
                      % (n, dt, po.lsoda_steps),
            "ms_per_vector": 1e3 * dt / max(n, 1),
            "parity_of_timed_pass": {"vectors_checked": 2, "residual_err_in_tolerance_units": worst[0],
                                     "jacobian_err_in_tolerance_units": worst[1],
                                     "tolerance": "oracle/tolerances.py::project_tolerances over lsoda_taus"}}
    return out


# ---------------------------------------------------------------------------
# configs[4]: stiff 50-state cascade (2550 coupled ODEs), 4096 vectors
# ---------------------------------------------------------------------------
def stiff_pass(torch, dev, dm, P5, t5, bufs):
    """One evaluation of the 4096-vector stiff ensemble AT THE PARITY-MEETING SETTING (tests/test_gpu_implicit.py:
    within 1e-8 |ref| + 5e-9 of the real reference's LSODA results): see STIFF_SETTING."""
    from sysbio_modeling_amd import _lib, models_zoo
    Yc, Sc, Yf, Sf, st, ns, nw, st2, ns2, nw2 = bufs
    o1 = _lib.make_opts('implicit_midpoint', rtol=1e-10, atol=1e-12, n_steps=4096, t_end=models_zoo.STIFF_T_END, step_mult=1)
    o2 = _lib.make_opts('implicit_midpoint', rtol=1e-10, atol=1e-12, n_steps=4096, t_end=models_zoo.STIFF_T_END, step_mult=2)
    dm.sens_dev(P5, t5, None, o1, Yc, Sc, st, ns, nw)
    dm.sens_dev(P5, t5, None, o2, Yf, Sf, st2, ns2, nw2)
    # Richardson: the symmetric rule's error expands in h^2
    torch.add(Yf, Yf - Yc, alpha=1.0 / 3.0, out=Yf)
    torch.add(Sf, Sf - Sc, alpha=1.0 / 3.0, out=Sf)
    return Yf, Sf


STIFF_SETTING = "implicit midpoint, 4096 + 8192 fixed steps, Richardson-extrapolated on the device"


def run_configs4(model, gm, dev, reps=3, cpu=True):
    import torch
    import os
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import zoo_model
    from sysbio_modeling_amd.model import OdeModel
    gm5 = zoo_model('stiff50')
    m5 = OdeModel(gm5.model, gm5.sens_model, gm5.n_vars, gm5.param_order, use_jit=False)
    m5.enable_jit(model.device_model.ctx)
    dm = m5.device_model
    V = 4096
    theta5, Pn = models_zoo.stiff_ensemble(V)
    g = np.load(os.path.join(B.REPO, 'tests', 'golden', 'stiff50_ref.npz'))
    assert np.array_equal(Pn[:3], g['P'])
    P5 = torch.from_numpy(Pn).to(dev)
    t_np = np.concatenate([[0.0], g['t'][g['idx']]])
    t5 = torch.from_numpy(t_np).to(dev)
    nt = len(t_np)
    f64, i32 = torch.float64, torch.int32
    bufs = (torch.empty((V, nt, 50), dtype=f64, device=dev), torch.empty((V, nt, 50, 50), dtype=f64, device=dev),
            torch.empty((V, nt, 50), dtype=f64, device=dev), torch.empty((V, nt, 50, 50), dtype=f64, device=dev)) + \
        tuple(torch.empty((V,), dtype=i32, device=dev) for _ in range(6))
    holder = {}

    def go():
        holder['YS'] = stiff_pass(torch, dev, dm, P5, t5, bufs)
    ms = _events(torch, dev, go, reps)
    Yg, Sg = holder['YS']
    steps = int(bufs[5].sum().item()) + int(bufs[8].sum().item())
    newton = int(bufs[6].sum().item()) + int(bufs[9].sum().item())
    failed = int(((bufs[4] != 0) | (bufs[7] != 0)).sum().item())
    # parity of the timed pass against the REAL reference's LSODA results (tests/golden/stiff50_ref.npz: 3 vectors)
    from oracle.tolerances import parity_err
    ey = parity_err(Yg[:3, 1:].cpu().numpy(), g['Y'])
    es = parity_err(Sg[:3, 1:].cpu().numpy().reshape(3, nt - 1, 2500), g['S'])
    out = {"workload": "configs[4]: stiff50 (50 states, 50 sensitivity parameters: 2550 coupled ODEs, rates spanning "
                       "1e6), 4096 vectors, 16 output times, " + STIFF_SETTING,
           "ms": ms, "steps": steps, "value": steps / (ms * 1e-3), "unit": "ODE-steps/s",
           "vectors_per_s": V / (ms * 1e-3), "launches_per_pass": 2, "newton_iterations_per_step": 1.0 + newton / max(steps, 1),
           "failed_vectors": failed,
           "parity_of_timed_pass": {"vectors_checked": 3, "state_err_in_tolerance_units": ey,
                                    "sens_err_in_tolerance_units": es,
                                    "tolerance": "|gpu - ref| <= 1e-8 |ref| + 5e-9 against the real reference "
                                                 "OdeModel's results (tests/golden/stiff50_ref.npz)"},
           "roofline": B.hbm_roofline("sbm_imid_kernel<stiff50>", 'imid_stiff50', ms, steps, 2 * 8 * 2550,
                                      "SURVEY.md section 8(d): 40 800 B per step under the state-streaming model; a "
                                      "fraction above 1 only says that a streaming integrator could not run this "
                                      "fast -- the kernel is register-resident, see roofline_valu_issue")}
    rv = B.valu_roofline('imid_stiff50', ms, steps)
    if rv:
        out["roofline_valu_issue"] = rv
    if cpu:
        # LSODA on the 2550-equation system differences (and factors) a dense 2550 x 2550 Jacobian: 45 - 150 s per
        # vector on one core.  The bounded sample is ONE vector over the first tenth of the time span (the first 100
        # of the reference's 1000 grid points); the rate is steps per second either way.
        from oracle import odeint_oracle as oo
        gm5.c_library()
        t0 = time.perf_counter()
        (_, _), info = oo.calc_jacobian(gm5, Pn[3], g['t'][:101], use_c=True, return_states=True, full_output=True)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": int(info['nst'][-1]) / dt, "unit": "ODE-steps/s", "cores": 1, "kind": "port",
            "sample": "1 vector (#3 of the ensemble) over t in [0, 1] (the first 100 of the 1000 grid points; the whole "
                      "span takes 45 - 150 s per vector): scipy.integrate.odeint rtol=atol=1e-10, 2550 ODEs, compiled C "
                      "RHS, Dfun=None as in the reference's default call; %.1f s, %d LSODA steps, %d Jacobian "
                      "evaluations" % (dt, int(info['nst'][-1]), int(info['nje'][-1]))}
    return out


# ---------------------------------------------------------------------------
# end-to-end fitting
# ---------------------------------------------------------------------------
def run_fit(model, gm, dev, reps=1, cpu=True):
    """fits/s of Project.fit_batch (multi-start Levenberg-Marquardt, everything on the device) on the configs[3]
    project with noise-free data, against the reference's pattern
    leastsq(project.residuals, x0, Dfun=project.calc_project_jacobian) (tests/test_Project.py:202-213,351-357) run
    serially on the CPU oracle; the optima are compared."""
    import torch
    from sysbio_modeling_amd import models_zoo
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(model, reference_compat=False)     # 5 % noise: the cost has a floor
    n_starts = 256
    starts = th0[None, :] + 0.15 * np.random.default_rng(1).standard_normal((n_starts, th0.size))
    proj.fit_batch(starts[:8], max_iter=3)            # warm-up (scratch allocation)
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        fit = proj.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8)     # leastsq's default tolerances
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = {"workload": "multi-start least-squares fit of the configs[3] project (8 experiments, 512 residual rows, "
                       "68 parameters, 5 %% noise): %d starts at 0.15 log-units from the generating parameters, "
                       "Project.fit_batch, max 100 iterations" % n_starts,
           "seconds": best, "fits_per_s": n_starts / best, "starts": n_starts,
           "converged": int(fit['converged'].sum()), "cost_min": float(np.min(fit['cost'])),
           "cost_median": float(np.median(fit['cost'])),
           "cost_max": float(np.max(fit['cost'])), "evaluations": int(fit['n_evaluations']),
           "evaluations_with_sensitivities": int(fit['n_jacobian_evaluations']),
           "starts_within_1pct_of_the_best_cost": int(np.sum(fit['cost'] <= 1.01 * np.min(fit['cost']))),
           "algorithm": "trust_region (MINPACK lmder, batched; sbm_lm_trust_step)",
           "convergence_note": "the smallest singular value of the Jacobian at the optimum is 4e-20 (a sloppy model: some "
                               "parameter combinations are not constrained by the data at all), so leastsq's tests "
                               "(ftol = xtol = 1.49e-8) are met by few starts within 100 iterations -- the cost still creeps "
                               "down along the flat directions; the fits are compared by the cost they reach",
           "distance_to_truth_max": float(np.max(np.abs(fit['theta'][fit['converged']] - th0[None, :])))
           if fit['converged'].any() else None}
    # round 1's multiplicative damping on the same starts, for the record
    proj.fit_batch(starts[:8], max_iter=3, algorithm='marquardt')
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fit_m = proj.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8, algorithm='marquardt')
    torch.cuda.synchronize(dev)
    out["algorithm_marquardt"] = {"seconds": time.perf_counter() - t0, "cost_min": float(np.min(fit_m['cost'])),
                                  "cost_median": float(np.median(fit_m['cost'])), "cost_max": float(np.max(fit_m['cost'])),
                                  "converged": int(fit_m['converged'].sum()),
                                  "note": "sbm_lm_step + lambda multiplied up / down by trial integrations (round 1)"}
    # the same fit with the trajectories integrated by DOP853 (method='dop853': a seventh of the steps at these tolerances)
    try:
        proj.fit_batch(starts[:8], max_iter=3, method='dop853')
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fit_8 = proj.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8, method='dop853')
        torch.cuda.synchronize(dev)
        dt8 = time.perf_counter() - t0
        out["integrated_with_dop853"] = {"seconds": dt8, "fits_per_s": n_starts / dt8, "cost_min": float(np.min(fit_8['cost'])),
                                         "cost_median": float(np.median(fit_8['cost'])), "cost_max": float(np.max(fit_8['cost'])),
                                         "converged": int(fit_8['converged'].sum())}
    except Exception as e:   # noqa: BLE001
        out["integrated_with_dop853"] = {"error": repr(e)[:200]}
    if cpu:
        from scipy.optimize import leastsq
        from oracle.project_oracle import ProjectOracle
        po = ProjectOracle(gm, list(proj.experiments), proj._model_parameter_settings,
                           dict(proj._measurement_to_model_map_raw),
                           sf_groups=['s%d' % v for v in models_zoo.CASCADE_MEASURED_SPECIES], reference_compat=False)
        calls = [0, 0]
        best_x = [starts[0].copy(), np.inf]
        budget_s = 30.0

        class _OutOfTime(Exception):
            pass

        def res(x):
            if time.perf_counter() - t0 > budget_s:
                raise _OutOfTime()       # (a trial point that makes the model stiff costs LSODA seconds: bound the leg)
            calls[0] += 1
            r = po.residuals(x)
            c0 = 0.5 * float(np.sum(r ** 2))
            if c0 < best_x[1]:
                best_x[0], best_x[1] = np.array(x, copy=True), c0
            return r

        def jac(x):
            if time.perf_counter() - t0 > budget_s:
                raise _OutOfTime()
            calls[1] += 1
            return po.calc_project_jacobian(x)
        t0 = time.perf_counter()
        try:
            x, _, info, _, ier = leastsq(res, starts[0], Dfun=jac, full_output=True, maxfev=400)
        except _OutOfTime:
            x, ier = best_x[0], -1       # stopped by the time budget: best point so far
        dt = time.perf_counter() - t0
        t0 = time.perf_counter() + 1e9   # (the cost evaluation below is not part of the budget)
        out["cpu_baseline"] = {
            "value": 1.0 / dt, "unit": "fits/s", "cores": 1, "kind": "port",
            "sample": "ONE start (#0): scipy.optimize.leastsq(residuals, x0, Dfun=calc_project_jacobian, maxfev=400) "
                      "over the CPU oracle, stopped after 30 s if not converged (ier = -1); %.1f s, %d residual + %d Jacobian "
                      "evaluations, ier=%d"
                      % (dt, calls[0], calls[1], ier),
            "cost": float(0.5 * np.sum(res(x) ** 2)),
            "gpu_cost_of_same_start": float(fit['cost'][0]),
            "distance_to_gpu_optimum_of_same_start": float(np.max(np.abs(x - fit['theta'][0]))),
            "note": "the problem is sloppy (68 parameters, many barely constrained by 512 rows): optima are compared "
                    "by their COST; parameter vectors of equal cost lie far apart along the sloppy directions"}
        out["speedup_vs_one_core"] = out["fits_per_s"] / out["cpu_baseline"]["value"]
    return out


# ---------------------------------------------------------------------------
# kernel variants side by side (no CPU leg: same workload as the headline)
# ---------------------------------------------------------------------------
def variant_extras(model, dev, theta_p, tg, rk4_steps):
    import torch
    from sysbio_modeling_amd import _lib
    dm = model.device_model
    V = theta_p.shape[0]
    Yk = torch.empty((V, len(tg), 20), dtype=torch.float64, device=dev)
    Sk = torch.empty((V, len(tg), 20, 40), dtype=torch.float64, device=dev)
    ns = torch.empty((V,), dtype=torch.int32, device=dev)

    def t(o):
        ms = _events(torch, dev, lambda: dm.sens_dev(theta_p, tg, None, o, Yk, Sk, None, ns, None), 3)
        st = int(ns.sum().item())
        return {"ms": ms, "steps": st, "steps_per_s": st / (ms * 1e-3)}
    ex = {}
    for variant in ('auto', 'row_lane', 'per_wave'):
        ex["sens_dopri45_%s" % variant] = t(_lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=variant))
        ex["sens_rk4_fixed_%d_%s" % (rk4_steps, variant)] = t(_lib.make_opts('rk4', n_steps=rk4_steps, t_end=100.0,
                                                                             variant=variant))
    # the eighth-order pair on the headline workload at OdeModel's default tolerances (rtol cut by ten for it, as the
    # Python classes do: _lib.implicit_adaptive_defaults) next to DOPRI45 at the same defaults: the pass, not steps/s, is
    # what a fit waits for
    tol = {k: model.integrator_options[k] for k in ('rtol', 'atol')}
    ex["sens_dopri45_default_tolerances"] = t(_lib.make_opts('dopri45', **tol))
    ex["sens_dop853_default_tolerances"] = dict(t(_lib.make_opts('dop853', rtol=0.1 * tol['rtol'], atol=tol['atol'])),
                                                note="DOP853 (SBM_DOP853): twelve stages per step, a seventh of the steps; "
                                                     "same parity tests as DOPRI45 (tests/test_gpu_dop853.py)")
    ex["sens_dop853_rtol1e-9_atol1e-12"] = t(_lib.make_opts('dop853', rtol=1e-9, atol=1e-12))
    # the reference's own fixture size: Michaelis-Menten (2 states, 5 parameters: 12 coupled ODEs), 4096 vectors with
    # sensitivities -- one trajectory per wavefront (row-group / row-lane) against eight per wavefront (packed)
    try:
        from sysbio_modeling_amd.symbolic import zoo_model
        from sysbio_modeling_amd.model import OdeModel
        gmm = zoo_model('michaelis_menten')
        mm = OdeModel(gmm.model, gmm.sens_model, gmm.n_vars, gmm.param_order, use_jit=False)
        mm.enable_jit(dm.ctx)
        rng = np.random.default_rng(1)
        Pm = torch.from_numpy(np.array([1e-3, 1e-3, 0.01, 0.01, 1e-3])[None, :] * np.exp(0.3 * rng.standard_normal((4096, 5)))).to(dev)
        tm = torch.linspace(0.0, 100.0, 17, dtype=torch.float64, device=dev)
        Ym = torch.empty((4096, 17, 2), dtype=torch.float64, device=dev)
        Sm = torch.empty((4096, 17, 2, 5), dtype=torch.float64, device=dev)
        nsm = torch.empty((4096,), dtype=torch.int32, device=dev)
        for variant in ('row_group', 'row_lane', 'packed'):
            o = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=variant)
            ms = _events(torch, dev, lambda: mm.device_model.sens_dev(Pm, tm, None, o, Ym, Sm, None, nsm, None), 5)
            stp = int(nsm.sum().item())
            ex["small_model_michaelis_menten_sens_%s" % variant] = {"ms": ms, "steps": stp, "steps_per_s": stp / (ms * 1e-3)}
    except Exception as e:   # noqa: BLE001
        ex["small_model_michaelis_menten_sens"] = {"error": repr(e)[:200]}
    # a model beyond one row / one column per lane: 70 states, 140 parameters, 9870 coupled ODEs per trajectory
    try:
        from sysbio_modeling_amd import models_zoo
        from sysbio_modeling_amd.symbolic import GeneratedModel
        from sysbio_modeling_amd.model import OdeModel
        gm7 = GeneratedModel(models_zoo.cascade_spec(70, name='cascade70'))
        m7 = OdeModel(gm7.model, gm7.sens_model, gm7.n_vars, gm7.param_order, use_jit=False)
        m7.enable_jit(dm.ctx)
        V7 = 1024
        P7 = torch.from_numpy(models_zoo.cascade_ensemble(V7, n=70, spread=0.3)[1]).to(dev)
        t7 = torch.tensor([50.0, 100.0], dtype=torch.float64, device=dev)
        Y7 = torch.empty((V7, 2, 70), dtype=torch.float64, device=dev)
        S7 = torch.empty((V7, 2, 70, 140), dtype=torch.float64, device=dev)
        st7 = torch.empty((V7,), dtype=torch.int32, device=dev)
        ns7 = torch.empty((V7,), dtype=torch.int32, device=dev)
        o7 = _lib.make_opts(**{k: v for k, v in m7.integrator_options.items() if k in ('method', 'rtol', 'atol')})
        ms7 = _events(torch, dev, lambda: m7.device_model.sens_dev(P7, t7, None, o7, Y7, S7, st7, ns7, None), 1)
        n7 = int(ns7.sum().item())
        ex["large_model_cascade70_dopri45"] = {
            "ms": ms7, "steps": n7, "steps_per_s": n7 / (ms7 * 1e-3), "n_equations": 70 + 70 * 140, "vectors": V7,
            "rtol": o7.rtol, "algorithmic_GBps": n7 / (ms7 * 1e-3) * 2 * 8 * (70 + 70 * 140) / 1e9,
            "failed_vectors": int((st7 != 0).sum().item())}
    except Exception as e:   # noqa: BLE001
        ex["large_model_cascade70_dopri45"] = {"error": repr(e)[:200]}
    return ex


# ---------------------------------------------------------------------------
# the MFMA question: dense Jacobian x S on the matrix cores against the scalar kernels
# ---------------------------------------------------------------------------
def run_dense(model, gm, dev, reps=3, cpu=True):
    """SURVEY.md section 8(d) / BASELINE north_star: "MFMA only if the sensitivity-RHS Jacobian x S product is
    actually dense enough to pay".  The same DOPRI45 driver, the same ensemble shape (4096 vectors, 16 outputs) on
    20-state networks whose df/dy has 40 (the cascade), 60, 120, 220 and 400 (dense) non-zeros: the scalar kernels
    (the row-group kernel, what AUTO picks for sparse Jacobians) against SBM_VARIANT_MFMA (v_mfma_f64_16x16x4_f64 tiles, cost independent of the
    sparsity).  No CPU leg: same workload family as the headline."""
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    V = 4096
    _, P = models_zoo.cascade_ensemble(V, spread=0.3)
    Pd = torch.from_numpy(P).to(dev)
    tg = torch.from_numpy(np.concatenate([[0.0], np.linspace(6.25, 100.0, 16)])).to(dev)
    Y = torch.empty((V, 17, 20), dtype=torch.float64, device=dev)
    S = torch.empty((V, 17, 20, 40), dtype=torch.float64, device=dev)
    ns = torch.empty((V,), dtype=torch.int32, device=dev)
    st = torch.empty((V,), dtype=torch.int32, device=dev)
    out = {"workload": "20 states / 40 parameters with forward sensitivities (820 ODEs), 4096 vectors, DOPRI45 rtol 1e-9 "
                       "atol 1e-12, 16 output times; by the number of non-zeros of df/dy"}
    models = [('cascade20', model)]
    for dens in (0.1, 0.25, 0.5, 1.0):
        spec = models_zoo.dense_spec(density=dens)
        g = GeneratedModel(spec)
        mm = OdeModel(g.model, g.sens_model, g.n_vars, g.param_order, model_name=spec.name, use_jit=False)
        mm.enable_jit(model.device_model.ctx)
        models.append((spec.name, mm))
    for name, mm in models:
        row = {"nnz_jy": int(mm.generated.hip_source.split('NNZ_JY = ')[1].split(';')[0])}
        for label, variant in (('valu', 'row_group'), ('mfma', 'mfma')):
            o = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant=variant)
            ms = _events(torch, dev, lambda: mm.device_model.sens_dev(Pd, tg, None, o, Y, S, st, ns, None), reps)
            steps = int(ns.sum().item())
            row[label] = {"ms": ms, "steps": steps, "steps_per_s": steps / (ms * 1e-3), "failed_vectors": int((st != 0).sum().item())}
        row["mfma_over_valu_time"] = row['mfma']['ms'] / row['valu']['ms']
        out[name] = row
    return out


def run_dop853(model, gm, dev, reps=3, cpu=True):
    """The headline ensemble (4096 vectors, 820 ODEs, 16 output times) through the DOP853 kernel alone, at OdeModel's
    default tolerances as the Python classes hand them to this method (rtol cut by ten): a profiling workload
    (`bench.py --only dop853`); the full run reports it under extras."""
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    V = 4096
    _, P = models_zoo.cascade_ensemble(V)
    dm = model.device_model
    Pd = torch.from_numpy(P).to(dev)
    grid = np.linspace(0, models_zoo.CASCADE_T_END, 1000)
    tg = torch.from_numpy(np.concatenate([[0.0], grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]])).to(dev)
    Yk = torch.empty((V, len(tg), 20), dtype=torch.float64, device=dev)
    Sk = torch.empty((V, len(tg), 20, 40), dtype=torch.float64, device=dev)
    ns = torch.empty((V,), dtype=torch.int32, device=dev)
    o = _lib.make_opts('dop853', rtol=0.1 * model.integrator_options['rtol'], atol=model.integrator_options['atol'])
    ms = _events(torch, dev, lambda: dm.sens_dev(Pd, tg, None, o, Yk, Sk, None, ns, None), reps)
    steps = int(ns.sum().item())
    return {"workload": "headline ensemble through DOP853 (kernel alone)", "ms": ms, "steps": steps,
            "steps_per_s": steps / (ms * 1e-3), "algorithmic_GBps": steps * B.BYTES_PER_STEP / (ms * 1e-3) / 1e9}


RUNNERS = {'configs1': run_configs1, 'configs3': run_configs3, 'configs4': run_configs4, 'fit': run_fit, 'dense': run_dense,
           'dop853': run_dop853}
ORDER = ['configs1', 'configs3', 'configs4', 'fit', 'dense']
