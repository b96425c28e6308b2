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
        r["roofline"]["achieved_fp64"] = B.fp64_rate(*B.flops_per_step(gm, 'dopri45' if label == 'dopri45' else 'rk4', sens=False),
                                                     steps / (ms * 1e-3))
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
    out["roofline"]["achieved_fp64"] = B.fp64_rate(*B.flops_per_step(gm, 'dopri45'), k_steps / (k_ms * 1e-3))
    rv = B.valu_roofline('sens_rowgroup_cascade20_dopri45', k_ms, k_steps)
    if rv:
        out["roofline_valu_issue"] = rv
    if cpu:
        po = B.project_oracle_of(gm, p4)
        # parity of the timed pass on 16 vectors spread over the batch (a pool of spawned workers runs the oracle on
        # them), then the timed one-core sample
        R = o4['residuals'].cpu().numpy()
        J = o4['jacobian'].cpu().numpy()
        picks = [int(x) for x in np.linspace(0, V - 1, 16).astype(int)]
        parity = oracle_parity_pool(p4, a, thetas, R, J, picks)
        po.lsoda_steps = 0
        n, t0 = 0, time.perf_counter()
        for v in range(1, V - 1):
            po.calc_project_jacobian(thetas[v])
            n += 1
            if time.perf_counter() - t0 > 5.0:
                break
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": po.lsoda_steps / dt, "unit": "ODE-steps/s", "cores": 1, "kind": "port",
            "sample": "%d project vectors (8 experiments each): ProjectOracle.calc_project_jacobian = odeint "
                      "rtol=atol=1e-10 on the 1000-point grid per experiment + numpy assembly; %.1f s, %d LSODA steps"
                      % (n, dt, po.lsoda_steps),
            "ms_per_vector": 1e3 * dt / max(n, 1),
            "parity_of_timed_pass": parity}
    return out


def run_configs3_sharded(model, gm, dev, world, rank, steps=5, warmup=1):
    """BASELINE configs[3] as it is named there -- "8 experiment settings x 1024 vectors, residual + Jacobian assembly,
    sharded over 8 MI355X via RCCL": the 1024 vectors are split into contiguous blocks, one per rank, every rank
    evaluates ALL 8 experiments of its block (sysbio_modeling_amd/distributed.py: splitting by experiment would cut
    through scale-factor groups), no data-path collective, then the all-gather of the per-vector residual norms.
    STRONG scaling: the total work is fixed.  Collective: every rank of the job must call this."""
    import torch
    import torch.distributed as dist
    from sysbio_modeling_amd import distributed as D
    # everything that can fail on ONE rank alone happens before the first collective of the timed part, and the ranks
    # agree on having got through it (a rank that raised would leave the others waiting in the all-gather)
    err = None
    try:
        p4, th4, thetas = _config3(model)
        V = thetas.shape[0]
        lo, hi = D.shard_range(V, rank, world)
        t4 = torch.from_numpy(np.ascontiguousarray(thetas[lo:hi])).to(dev)
        p4.evaluate_batch(t4, jacobian=True, want=('jacobian',))
        torch.cuda.synchronize(dev)
    except Exception as e:   # noqa: BLE001
        err = repr(e)[:300]
    if world > 1:
        flag = torch.tensor([0.0 if err else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 0.5:
            return {"error": err or "another rank failed to set the workload up"}
    elif err:
        return {"error": err}
    holder = {}

    def step():
        o = p4.evaluate_batch(t4, jacobian=True, want=('jacobian',))
        holder['o'] = o
        holder['g'] = D.gather_norms(o['norms'], V)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(max(1, warmup)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    o, g = holder['o'], holder['g']
    stats = torch.tensor([dt, float(o['n_steps'].sum().item()), float((o['status'] != 0).sum().item())],
                         dtype=torch.float64, device=dev)
    mine = bool(torch.equal(g[lo:hi], o['norms'])) and int(g.shape[0]) == V
    if world > 1:
        mx, sm = stats.clone(), stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        ok = torch.tensor([1.0 if mine else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        dt, total, failed, mine = float(mx[0]), float(sm[1]), int(sm[2]), bool(ok.item() > 0.5)
    else:
        total, failed = float(stats[1]), int(stats[2])
    ms = 1e3 * dt / steps
    return {"workload": "configs[3] sharded: 1024 project vectors x 8 experiments (8192 trajectories of 820 ODEs, 512 rows x "
                        "68 parameters per vector) split by vector over %d rank(s), all-gather of the residual norms" % world,
            "n_gpus": world, "vectors_per_rank": [D.shard_range(V, r, world)[1] - D.shard_range(V, r, world)[0] for r in range(world)],
            "ms": ms, "steps": total, "value": total / (ms * 1e-3), "unit": "ODE-steps/s", "scaling": "strong",
            "failed_vectors": failed, "gathered_norms_match_local_block_on_every_rank": mine}


def _oracle_rows(args):
    """one worker of oracle_parity_pool: the assembly oracle (SciPy odeint per experiment + numpy assembly) on its share,
    and -- with ``tight`` -- the same assembly formulas over a TIGHT integration (DOP853, rtol 1e-13) of the same vectors"""
    experiments, settings, mapping, sf_groups, rows, tight = args
    from sysbio_modeling_amd.symbolic import zoo_model
    from oracle.project_oracle import ProjectOracle
    po = ProjectOracle(zoo_model('cascade20'), experiments, settings, mapping, sf_groups=sf_groups)
    out = []
    for th in rows:
        po.tight = False
        rr, sims, Bf = po.residuals(th, return_parts=True)
        row = [rr, sims, Bf, po.model_jacobian(th), po.calc_project_jacobian(th), None, None]
        if tight:
            po.tight = True
            row[5], row[6] = po.residuals(th), po.calc_project_jacobian(th)
        out.append(tuple(row))
    return out


def oracle_parity_pool(proj, a, thetas, R, J, picks, tight=True):
    """Residual / Jacobian rows of a timed GPU pass against the assembly oracle for the vectors `picks`, the oracle runs
    spread over the host cores (spawned workers; the oracle of one 8-experiment vector costs ~2.5 s of one core).
    ``tight``: ALSO arbitrate the timed pass -- residual rows, ||r||^2 and Jacobian rows of every picked vector from a tight
    integration pushed through the same assembly formulas, with the GPU's and LSODA's relative distance to it (north_star:
    "residuals within 1e-8 rel of SciPy"; LSODA at rtol = atol = 1e-10 is itself ~1e-9 .. 1e-8 off, so a GPU-vs-SciPy
    difference of that size has to be attributed)."""
    import multiprocessing as mp
    import os
    from oracle import tolerances as tol
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16, len(picks)))
    sf_groups = [g if len(g) > 1 else g[0] for g in proj._loss_function.groups]
    shares = [picks[i::cores] for i in range(cores)]
    jobs = [(list(proj.experiments), proj._model_parameter_settings, dict(proj._measurement_to_model_map_raw), sf_groups,
             [thetas[v] for v in sh], tight) for sh in shares]
    t0 = time.perf_counter()
    with mp.get_context('spawn').Pool(cores) as pool:
        res = pool.map(_oracle_rows, jobs)
    worst_r = worst_j = worst_rel = worst_n = 0.0
    worst_v = [None, None]
    vs_tight = {"gpu_norm_rel_err_vs_tight": 0.0, "scipy_norm_rel_err_vs_tight": 0.0,
                "gpu_residual_rel_err_vs_tight": 0.0, "scipy_residual_rel_err_vs_tight": 0.0,
                "gpu_jacobian_rel_err_vs_tight": 0.0, "scipy_jacobian_rel_err_vs_tight": 0.0}
    n_gpu_closer = 0

    def rel(x, ref):
        return float(np.linalg.norm(np.asarray(x) - ref) / np.linalg.norm(ref))
    for sh, rows in zip(shares, res):
        for v, (rr, sims, Bf, Jm, Jr, rt, Jt) in zip(sh, rows):
            ts, tj = tol.lsoda_taus(a, thetas[v], sims, Jm)
            t = tol.project_tolerances(a, sims, Bf, ts, Jm, tj)
            er, ej = tol.tol_ratio(R[v], rr, t['residuals']), tol.tol_ratio(J[v], Jr, t['jacobian'])
            if er > worst_r:
                worst_r, worst_v[0] = er, v
            if ej > worst_j:
                worst_j, worst_v[1] = ej, v
            worst_rel = max(worst_rel, rel(R[v], rr))
            worst_n = max(worst_n, abs(float(np.sum(R[v] ** 2)) / float(np.sum(rr ** 2)) - 1.0))
            if rt is not None:
                nt = float(np.sum(rt ** 2))
                g = {"gpu_norm_rel_err_vs_tight": abs(float(np.sum(R[v] ** 2)) / nt - 1.0),
                     "scipy_norm_rel_err_vs_tight": abs(float(np.sum(rr ** 2)) / nt - 1.0),
                     "gpu_residual_rel_err_vs_tight": rel(R[v], rt), "scipy_residual_rel_err_vs_tight": rel(rr, rt),
                     "gpu_jacobian_rel_err_vs_tight": rel(J[v], Jt), "scipy_jacobian_rel_err_vs_tight": rel(Jr, Jt)}
                n_gpu_closer += int(g["gpu_norm_rel_err_vs_tight"] <= g["scipy_norm_rel_err_vs_tight"])
                for k_, x in g.items():
                    vs_tight[k_] = max(vs_tight[k_], x)
    out = {"vectors_checked": len(picks), "residual_err_in_tolerance_units": worst_r, "jacobian_err_in_tolerance_units": worst_j,
           "residual_rel_err": worst_rel, "norm_rel_err": worst_n,
           "worst_vectors": worst_v, "oracle_wall_seconds": time.perf_counter() - t0, "oracle_worker_processes": cores,
           "tolerance": "oracle/tolerances.py::project_tolerances over lsoda_taus (trajectories within 1e-8 |ref| + 5e-9 "
                        "of the reference's LSODA, propagated to first order through the reference's formulas); <= 1 passes. "
                        "residual_rel_err / norm_rel_err: worst ||r_gpu - r_scipy|| / ||r_scipy|| and | ||r_gpu||^2 / ||r_scipy||^2 - 1 |"}
    if tight:
        out.update(vs_tight)
        out["vectors_where_gpu_norm_is_closer_to_tight_than_scipy"] = n_gpu_closer
        out["gpu_within_1e-8_of_tight"] = bool(vs_tight["gpu_norm_rel_err_vs_tight"] <= 1e-8
                                               and vs_tight["gpu_residual_rel_err_vs_tight"] <= 1e-8)
        out["arbitration"] = ("*_vs_tight: the same vectors integrated by DOP853 at rtol 1e-13 (oracle.odeint_oracle.tight_solution) and "
                              "pushed through the same assembly formulas; worst over the picked vectors of | ||r||^2 / ||r_tight||^2 - 1 | "
                              "and ||x - x_tight|| / ||x_tight|| for the GPU pass and for the SciPy (LSODA rtol = atol = 1e-10) oracle")
    return out


# ---------------------------------------------------------------------------
# configs[4]: stiff 50-state cascade (2550 coupled ODEs), 4096 vectors
# ---------------------------------------------------------------------------
IEX_ORDER = 8
STIFF_SETTING = ("extrapolated implicit Euler (SBM_IMPLICIT_EXTRAP, order 8: 36 implicit-Euler steps per macro step; chain model: "
                 "sequences side by side, persistent wavefronts), LOCAL error control in the kernel at the method's default "
                 "tolerances, ONE launch, no step count chosen")
STIFF_FIXED_SETTING = "implicit midpoint, 4096 + 8192 fixed steps, Richardson-extrapolated on the device (round 2's timed setting)"


def _stiff_setup(model, dev, V=4096):
    import os
    import torch
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import zoo_model
    from sysbio_modeling_amd.model import OdeModel
    gm5 = zoo_model('stiff50')
    m5 = OdeModel(gm5.model, gm5.sens_model, gm5.n_vars, gm5.param_order, use_jit=False)
    m5.enable_jit(model.device_model.ctx)
    B._CURRENT_STAMPS.update(B.build_stamps(gm5))
    theta5, Pn = models_zoo.stiff_ensemble(V)
    gdir = os.path.join(B.REPO, 'tests', 'golden')
    g = np.load(os.path.join(gdir, 'stiff50_ref.npz'))
    assert np.array_equal(Pn[:3], g['P'])
    P5 = torch.from_numpy(Pn).to(dev)
    t_np = np.concatenate([[0.0], g['t'][g['idx']]])
    t5 = torch.from_numpy(t_np).to(dev)
    return gm5, m5, Pn, P5, t_np, t5, gdir


def stiff_golden_parity(gdir, Pn, Yg, Sg):
    """Parity of a timed configs[4] pass against every stiff50 vector the REAL reference OdeModel was run on
    (tests/golden/stiff50_ref.npz: vectors 0-2 of the ensemble; stiff50_wide_ref.npz: 32 more spread over it,
    make_golden_stiff_wide.py), arbitrated -- where the reference's own LSODA run is the one that is off -- by the tight
    LSODA solutions (stiff50_tight.npz, stiff50_wide_tight.npz) exactly as tests/conftest.py::check_parity does:
    a vector passes if it is within |gpu - ref| <= 1e-8 |ref| + 5e-9 of the reference's result, or within that of the
    tight solution AND closer to it than the reference's result is."""
    import os
    from oracle.tolerances import parity_err
    sets = [('stiff50_ref.npz', 'stiff50_tight.npz', np.arange(3), None)]
    if os.path.exists(os.path.join(gdir, 'stiff50_wide_ref.npz')):
        w = np.load(os.path.join(gdir, 'stiff50_wide_ref.npz'))
        sets.append(('stiff50_wide_ref.npz', 'stiff50_wide_tight.npz', w['index'], None))
    worst = {'state_vs_reference': 0.0, 'sens_vs_reference': 0.0, 'state_vs_tight': 0.0, 'sens_vs_tight': 0.0}
    n_checked = n_tight = n_arbitrated = n_failed = 0
    for ref_name, tight_name, index, _ in sets:
        g = np.load(os.path.join(gdir, ref_name))
        tight = np.load(os.path.join(gdir, tight_name)) if os.path.exists(os.path.join(gdir, tight_name)) else None
        trow = {}
        if tight is not None:
            rows = tight['rows'] if 'rows' in tight.files else np.arange(len(tight['P']))
            trow = {int(r): i for i, r in enumerate(rows)}
        assert np.array_equal(Pn[index], g['P'])
        nt = g['Y'].shape[1]
        for j, v in enumerate(index):
            Y = Yg[int(v), 1:].cpu().numpy()
            S = Sg[int(v), 1:].cpu().numpy().reshape(nt, -1)
            ey, es = parity_err(Y, g['Y'][j]), parity_err(S, g['S'][j])
            worst['state_vs_reference'] = max(worst['state_vs_reference'], ey)
            worst['sens_vs_reference'] = max(worst['sens_vs_reference'], es)
            n_checked += 1
            ok = ey <= 1.0 and es <= 1.0
            if j in trow:
                i = trow[j]
                n_tight += 1
                ty, ts = parity_err(Y, tight['Y'][i]), parity_err(S, tight['S'][i])
                worst['state_vs_tight'] = max(worst['state_vs_tight'], ty)
                worst['sens_vs_tight'] = max(worst['sens_vs_tight'], ts)
                if not ok:
                    ly, ls = parity_err(g['Y'][j], tight['Y'][i]), parity_err(g['S'][j], tight['S'][i])
                    ok = ty <= 1.0 and ts <= 1.0 and ty <= max(ly, 1e-3) and ts <= max(ls, 1e-3)
                    n_arbitrated += int(ok)
            n_failed += int(not ok)
    return {"vectors_checked": n_checked, "vectors_with_a_tight_solution": n_tight,
            "worst_state_err_vs_reference_in_tolerance_units": worst['state_vs_reference'],
            "worst_sens_err_vs_reference_in_tolerance_units": worst['sens_vs_reference'],
            "worst_state_err_vs_tight_solution": worst['state_vs_tight'],
            "worst_sens_err_vs_tight_solution": worst['sens_vs_tight'],
            "vectors_passed_by_arbitration": n_arbitrated, "vectors_failed": n_failed,
            "tolerance": "|gpu - ref| <= 1e-8 |ref| + 5e-9 against the real reference OdeModel's results (LSODA at rtol = "
                         "atol = 1e-10, itself about one unit off its own tight solution on this model); a vector beyond "
                         "that must be within it of the tight solution and closer to it than the reference is",
            "note": "vs-reference figures above 1 with vectors_failed = 0 are the reference's own LSODA error"}


def run_configs4(model, gm, dev, reps=3, cpu=True):
    """BASELINE configs[4] at the product's DEFAULT options for a stiff model: what method='auto' / 'implicit_controlled'
    run -- SBM_IMPLICIT_EXTRAP with the tolerances _lib.implicit_adaptive_defaults derives (no hand-chosen step count)."""
    import torch
    from sysbio_modeling_amd import _lib
    gm5, m5, Pn, P5, t_np, t5, gdir = _stiff_setup(model, dev)
    dm = m5.device_model
    V, nt = Pn.shape[0], len(t_np)
    f64, i32 = torch.float64, torch.int32
    Y = torch.empty((V, nt, 50), dtype=f64, device=dev)
    S = torch.empty((V, nt, 50, 50), dtype=f64, device=dev)
    st, ns, nr = (torch.empty((V,), dtype=i32, device=dev) for _ in range(3))
    o = dict(m5.integrator_options, method='implicit_extrap')
    _lib.implicit_adaptive_defaults(o, ())
    tol = {k: o[k] for k in ('rtol', 'atol')}
    opts = _lib.make_opts('implicit_extrap', order=IEX_ORDER, **tol)
    ms = _events(torch, dev, lambda: dm.sens_dev(P5, t5, None, opts, Y, S, st, ns, nr), reps)
    macro, rej = int(ns.sum().item()), int(nr.sum().item())
    euler = (macro + rej) * (IEX_ORDER * (IEX_ORDER + 1) // 2)
    failed = int((st != 0).sum().item())
    # state only, the same options (every residual-only evaluation of a stiff Project)
    ms_state = _events(torch, dev, lambda: dm.simulate_dev(P5, t5, None, opts, Y, st, ns, nr), reps)
    macro_state = int(ns.sum().item())
    dm.sens_dev(P5, t5, None, opts, Y, S, st, ns, nr)      # (Y of the sensitivity pass again, for the parity check)
    evals = 2.9      # Newton evaluations per Euler step (measured on sbm_iex_kernel: profiles/r03, SBM_IEX_COUNT_NEWTON; the
    #                  sequences-side-by-side kernel follows the same stopping rule per sequence)
    F, formula = B.flops_per_step(gm5, 'implicit_euler', evals_per_step=evals)
    out = {"workload": "configs[4]: stiff50 (50 states, 50 sensitivity parameters: 2550 coupled ODEs, rates spanning "
                       "1e6), 4096 vectors, 16 output times, " + STIFF_SETTING + " (rtol %g, atol %g)" % (tol['rtol'], tol['atol']),
           "ms": ms, "macro_steps": macro, "rejected_macro_steps": rej, "euler_steps": euler, "steps": euler,
           "value": euler / (ms * 1e-3), "unit": "ODE-steps/s",
           "unit_note": "a step = one implicit-Euler step of the 2550-equation system (Newton on the 50 states + one sparse "
                        "solve per sensitivity column); a macro step is 36 of them, accepted or rejected as a whole",
           "vectors_per_s": V / (ms * 1e-3), "launches_per_pass": 1, "failed_vectors": failed,
           "macro_steps_per_vector": macro / V, "state_only": {"ms": ms_state, "macro_steps_per_vector": macro_state / V},
           "parity_of_timed_pass": stiff_golden_parity(gdir, Pn, Y, S),
           "roofline": B.hbm_roofline("sbm_iex_seq_kernel<stiff50, rotated columns>", 'iex_stiff50', ms, euler, 2 * 8 * 2550,
                                      "SURVEY.md section 8(d): 40 800 B per step under the state-streaming model (a "
                                      "streaming integrator reads and writes the augmented state once per implicit-Euler "
                                      "step); a fraction above 1 says only that a streaming integrator could not run this "
                                      "fast -- the kernel keeps S in registers and LDS, see roofline_valu_issue")}
    out["roofline"]["achieved_fp64"] = B.fp64_rate(F, formula, euler / (ms * 1e-3))
    rv = B.valu_roofline('iex_stiff50', ms, euler)
    if rv:
        out["roofline_valu_issue"] = rv
    ri = B.issue_roofline('iex_stiff50', ms, euler)
    if ri:
        # the bound that applies to a kernel at ONE wavefront per SIMD: a wavefront issues one instruction of ANY kind per
        # four cycles, so vector, scalar and LDS instructions queue behind each other (the byte model's fraction above 1
        # is kept as roofline_hbm_model: it says a streaming integrator could not run this fast, nothing else)
        out["roofline_hbm_model"] = out["roofline"]
        out["roofline"] = dict(ri, achieved_fp64=out["roofline_hbm_model"].get("achieved_fp64"),
                               traffic=out["roofline_hbm_model"].get("traffic"), kernel=out["roofline_hbm_model"].get("kernel"),
                               kernel_ms=ms)
    # round 2's timed setting beside it: the hand-chosen fixed-step Richardson pair
    try:
        fx = run_configs4_fixed(model, gm, dev, reps=reps, cpu=False, setup=(gm5, m5, Pn, P5, t_np, t5, gdir))
        out["fixed_step_pair"] = {k: fx[k] for k in ("ms", "steps", "value", "parity_of_timed_pass", "setting")}
        out["speedup_over_fixed_step_pair"] = fx["ms"] / ms
        out["fixed_step_pair"]["note"] = ("round 2 chose 4096 + 8192 steps on the three vectors of stiff50_ref.npz; on the 35 "
                                          "vectors of the wide pin it is NOT at parity (vectors_failed above): the time ratio "
                                          "compares a controlled integrator with a setting that misses the tolerance")
    except Exception as e:   # noqa: BLE001
        out["fixed_step_pair"] = {"error": repr(e)[:200]}
    if cpu:
        out["cpu_baseline"] = stiff_cpu_legs(gm5, Pn, t_np)
    return out


def stiff_pass_fixed(torch, dev, dm, P5, t5, bufs):
    """One evaluation of the 4096-vector stiff ensemble with the fixed-step pair (STIFF_FIXED_SETTING)."""
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


def run_configs4_fixed(model, gm, dev, reps=3, cpu=True, setup=None):
    """configs[4] with round 2's hand-chosen fixed-step pair (kept for comparison and as a profiling workload)."""
    import torch
    gm5, m5, Pn, P5, t_np, t5, gdir = setup or _stiff_setup(model, dev)
    dm = m5.device_model
    V, nt = Pn.shape[0], len(t_np)
    f64, i32 = torch.float64, torch.int32
    bufs = (torch.empty((V, nt, 50), dtype=f64, device=dev), torch.empty((V, nt, 50, 50), dtype=f64, device=dev),
            torch.empty((V, nt, 50), dtype=f64, device=dev), torch.empty((V, nt, 50, 50), dtype=f64, device=dev)) + \
        tuple(torch.empty((V,), dtype=i32, device=dev) for _ in range(6))
    holder = {}

    def go():
        holder['YS'] = stiff_pass_fixed(torch, dev, dm, P5, t5, bufs)
    ms = _events(torch, dev, go, reps)
    Yg, Sg = holder['YS']
    steps = int(bufs[5].sum().item()) + int(bufs[8].sum().item())
    newton = int(bufs[6].sum().item()) + int(bufs[9].sum().item())
    out = {"workload": "configs[4] with " + STIFF_FIXED_SETTING, "setting": STIFF_FIXED_SETTING,
           "ms": ms, "steps": steps, "value": steps / (ms * 1e-3), "unit": "ODE-steps/s", "launches_per_pass": 2,
           "newton_iterations_per_step": 1.0 + newton / max(steps, 1),
           "failed_vectors": int(((bufs[4] != 0) | (bufs[7] != 0)).sum().item()),
           "parity_of_timed_pass": stiff_golden_parity(gdir, Pn, Yg, Sg),
           "roofline": B.hbm_roofline("sbm_imid_kernel<stiff50>", 'imid_stiff50', ms, steps, 2 * 8 * 2550,
                                      "40 800 B per implicit-midpoint step under the state-streaming model")}
    F, formula = B.flops_per_step(gm5, 'implicit_midpoint', evals_per_step=1.0 + newton / max(steps, 1))
    out["roofline"]["achieved_fp64"] = B.fp64_rate(F, formula, steps / (ms * 1e-3))
    rv = B.valu_roofline('imid_stiff50', ms, steps)
    if rv:
        out["roofline_valu_issue"] = rv
    return out


def _silence_fortran_unit6():
    """ODEPACK writes its warnings ("lsoda-- ...") to Fortran unit 6 = the C-level stdout / stderr of the process; a
    CPU leg that provokes hundreds of them buries bench.py's own output.  Returns a function that restores the
    descriptors."""
    import os
    import sys
    sys.stdout.flush()
    sys.stderr.flush()
    saved = (os.dup(1), os.dup(2))
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    os.dup2(null, 2)
    os.close(null)

    def restore():
        os.dup2(saved[0], 1)
        os.dup2(saved[1], 2)
        os.close(saved[0])
        os.close(saved[1])
    return restore


def stiff_cpu_legs(gm5, Pn, t_np, n_pts=11):
    """CPU legs of configs[4] on ONE host core (BLAS limited to one thread), vector #3 of the ensemble, BOUNDED: both
    calls cover the first ``n_pts`` points of the reference's 1000-point grid (t in [0, 0.1]: the stiff transient, where
    LSODA takes the same kind of step it takes later -- dense 2550 x 2550 Jacobian, factor, few Newton iterations):
      as_reference   the reference's default call -- odeint, Dfun=None: LSODA differences and factors the dense Jacobian;
      analytic_dfun  the reference's use_jac path (model/ode_model.py:114-120) with the generated analytic Jacobian of the
                     augmented system as Dfun.
    The FULL span (36 s and 34 s of one core per vector) is measured once per round by tests/tools/cpu_leg_stiff50.py and
    quoted from profiles/rNN/stiff50_cpu_full_span.json (full_span_measured_once): a default bench run stays short."""
    from oracle import odeint_oracle as oo
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        import contextlib
        threadpool_limits = lambda limits=None: contextlib.nullcontext()     # noqa: E731
    gm5.c_library()
    p = Pn[3]
    grid = np.linspace(0.0, 10.0, 1000)
    restore = _silence_fortran_unit6()
    try:
      with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        (_, _), info = oo.calc_jacobian(gm5, p, grid[:n_pts], use_c=True, return_states=True, full_output=True)
        dt = time.perf_counter() - t0
        legs = {"value": int(info['nst'][-1]) / dt, "unit": "ODE-steps/s", "cores": 1, "kind": "port",
                "sample": "1 vector (#3 of the ensemble), t in [0, %.2f] on the reference's 1000-point grid: "
                          "scipy.integrate.odeint rtol=atol=1e-10, 2550 ODEs, compiled C RHS, Dfun=None as in the reference's "
                          "default call, BLAS on one thread; %.1f s, %d LSODA steps, %d Jacobian evaluations"
                          % (grid[n_pts - 1], dt, int(info['nst'][-1]), int(info['nje'][-1])),
                "seconds": dt, "span_covered": [0.0, float(grid[n_pts - 1])]}
        # the use_jac path (the Python callback fills a 2550 x 2550 matrix per Jacobian evaluation)
        jac = gm5.sens_model_jac
        t0 = time.perf_counter()
        (_, _), inf2 = oo.calc_jacobian(gm5, p, grid[:n_pts], use_c=True, return_states=True, full_output=True,
                                        sens_model_jac=jac)
        d2 = time.perf_counter() - t0
        legs["analytic_dfun"] = {
            "value": int(inf2['nst'][-1]) / d2, "unit": "ODE-steps/s", "cores": 1, "kind": "port", "seconds": d2,
            "span_covered": [0.0, float(grid[n_pts - 1])],
            "sample": "the same vector, Dfun = the generated analytic Jacobian of the augmented system "
                      "(GeneratedModel.sens_model_jac, col_deriv layout: the reference's use_jac path), t in [0, %.2f]: "
                      "%.1f s, %d LSODA steps, %d Jacobian evaluations" % (grid[n_pts - 1], d2, int(inf2['nst'][-1]),
                                                                          int(inf2['nje'][-1]))}
        import json
        import os
        for rnd in ('r04', 'r03'):
            fp = os.path.join(B.REPO, 'profiles', rnd, 'stiff50_cpu_full_span.json')
            if os.path.exists(fp):
                with open(fp) as fh:
                    legs["full_span_measured_once"] = dict(json.load(fh), source='profiles/%s/stiff50_cpu_full_span.json' % rnd)
                fs = legs["full_span_measured_once"].get("as_reference", {})
                if fs.get("seconds"):
                    legs["seconds_per_vector_full_span"] = fs["seconds"]
                break
    finally:
        restore()
    return legs


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
    # ---- the same project made well-posed: a log-normal prior (sigma = 1 log-unit, the reference's
    # set_parameter_log_prior, base_project.py:675-691) on every parameter.  The sloppy directions are then constrained, every
    # start converges by leastsq's own tests, and so does the reference's serial pattern on the CPU oracle -- a baseline
    # that FINISHES (without priors one start wanders into stiff corners of parameter space and costs LSODA minutes).
    def add_priors(pr):
        for g_, slots in pr.project_param_idx.items():
            for key, gi in slots.items():
                pr.set_parameter_log_prior(g_, key, float(th0[gi]), 1.0)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            projp, _ = models_zoo.cascade_config4_project(model, reference_compat=False)
            add_priors(projp)
        projp.fit_batch(starts[:8], max_iter=3)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fitp = projp.fit_batch(starts, max_iter=100, ftol=1.49012e-8, xtol=1.49012e-8)
        torch.cuda.synchronize(dev)
        dtp = time.perf_counter() - t0
        wp = {"workload": "the same project with a log-normal prior (sigma 1) on each of the 68 parameters (580 residual rows), "
                          "%d starts, Project.fit_batch, leastsq's default tolerances" % n_starts,
              "seconds": dtp, "fits_per_s": n_starts / dtp, "starts": n_starts, "converged": int(fitp['converged'].sum()),
              "iterations_median": float(np.median(fitp['n_iter'])), "cost_min": float(np.min(fitp['cost'])),
              "cost_median": float(np.median(fitp['cost'])), "cost_max": float(np.max(fitp['cost'])),
              "evaluations": int(fitp['n_evaluations'])}
        if cpu:
            from scipy.optimize import leastsq
            from oracle.project_oracle import ProjectOracle
            po = ProjectOracle(gm, list(projp.experiments), projp._model_parameter_settings,
                               dict(projp._measurement_to_model_map_raw),
                               sf_groups=['s%d' % v for v in models_zoo.CASCADE_MEASURED_SPECIES], reference_compat=False)
            add_priors(po)
            calls = [0, 0]
            budget_s = 25.0

            class _OutOfTime(Exception):
                pass

            def res(x):
                if time.perf_counter() - t0 > budget_s:
                    raise _OutOfTime()
                calls[0] += 1
                return po.residuals(x)

            def jac(x):
                if time.perf_counter() - t0 > budget_s:
                    raise _OutOfTime()
                calls[1] += 1
                return po.calc_project_jacobian(x)
            restore = _silence_fortran_unit6()
            t0 = time.perf_counter()
            try:
                x, _, info, _, ier = leastsq(res, starts[0], Dfun=jac, full_output=True, maxfev=400)
            except _OutOfTime:
                x, ier = starts[0], -1
            finally:
                restore()
            dt = time.perf_counter() - t0
            t0 = time.perf_counter() + 1e9
            cost_cpu = float(0.5 * np.sum(po.residuals(x) ** 2))
            finished = ier in (1, 2, 3, 4)
            agree = finished and abs(cost_cpu - float(fitp['cost'][0])) <= 1e-6 * abs(cost_cpu)
            wp["cpu_baseline"] = {
                "value": (1.0 / dt) if finished else None, "unit": "fits/s", "cores": 1, "kind": "port", "finished": finished,
                "seconds": dt, "ier": int(ier), "residual_evaluations": calls[0], "jacobian_evaluations": calls[1],
                "cost": cost_cpu, "gpu_cost_of_same_start": float(fitp['cost'][0]), "costs_agree_to_1e-6": bool(agree),
                "distance_to_gpu_optimum_of_same_start": float(np.max(np.abs(x - fitp['theta'][0]))),
                "sample": "ONE start (#0): scipy.optimize.leastsq(residuals, x0, Dfun=calc_project_jacobian) over the CPU oracle "
                          "(SciPy odeint per experiment + numpy assembly), the reference's pattern tests/test_Project.py:202-213"}
            wp["speedup_vs_one_core"] = (wp["fits_per_s"] * dt) if (finished and agree) else None
        out["with_priors"] = wp
    except Exception as e:   # noqa: BLE001
        out["with_priors"] = {"error": repr(e)[:300]}
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
    # the cost of a Jacobian pass AT PARITY for both explicit pairs, side by side (round 2's tightening of the default atol
    # doubled DOPRI45's steps per pass; DOP853 gives it back), and what explicit_method='auto' would pick for this batch
    pick = _lib.predict_explicit_pair(tol['rtol'], int(V), 1, 2)
    ex["ms_per_pass_at_parity"] = {
        "dopri45": ex["sens_dopri45_default_tolerances"].get("ms"), "dop853": ex["sens_dop853_default_tolerances"].get("ms"),
        "dopri45_round1_tolerances_rtol1e-9_atol1e-12": ex.get("sens_dopri45_auto", {}).get("ms"),
        "explicit_method_auto_picks": pick[0], "predicted_time_ratio_dopri45_over_dop853": pick[1],
        "measured_time_ratio": (ex["sens_dopri45_default_tolerances"].get("ms") or 0.0) / max(ex["sens_dop853_default_tolerances"].get("ms") or 1.0, 1e-9),
        "note": "kernel alone, 4096 vectors x 820 ODEs, OdeModel's default tolerances (DOP853 at a tenth of the inherited rtol, as "
                "the Python classes run it); both meet SURVEY 8(d) against tight solutions (tests/test_gpu_parity_sweeps.py, "
                "tests/test_gpu_dop853.py)"}
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
        if dens == 1.0:
            B._CURRENT_STAMPS['dense20'] = B.build_stamps(g).get(g.name)      # (the PMC entry of the dense network's kernels)
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


def run_dense_stiff(model, gm, dev, reps=1, cpu=True):
    """BASELINE configs[4] names a "dense Jacobian x S product on MFMA"; stiff50, the timed workload, has two non-zeros per
    row.  This leg times a DENSE stiff network of the same size (models_zoo.dense_stiff_spec: 48 states, 1200 non-zeros of
    df/dy, 48 sensitivity columns = 2352 ODEs, degradation rates over four decades) at the stiff integrator's default
    options: sbm_iex_kernel with the row-distributed dense LU (IM_DIST) and per-column substitutions on the VALU -- there
    are no MFMA tiles for the implicit path.  1024 vectors (the dense factorisation costs ~n^3 / 3 per
    Newton matrix per trajectory); parity on the vectors the real reference was run on."""
    import os
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    g5 = GeneratedModel(models_zoo.dense_stiff_spec())
    m5 = OdeModel(g5.model, g5.sens_model, g5.n_vars, g5.param_order, model_name=g5.spec.name, use_jit=False)
    m5.enable_jit(model.device_model.ctx)
    V = 1024
    _, Pn = models_zoo.dense_stiff_ensemble(V)
    P = torch.from_numpy(Pn).to(dev)
    t_np = np.concatenate([[0.0], np.linspace(0, models_zoo.DENSE_STIFF_T_END, 1000)[np.searchsorted(
        np.linspace(0, models_zoo.DENSE_STIFF_T_END, 1000), models_zoo.DENSE_STIFF_MEASURE_TIMES)]])
    t5 = torch.from_numpy(t_np).to(dev)
    n, k = g5.n_vars, g5.n_sens
    Y = torch.empty((V, len(t_np), n), dtype=torch.float64, device=dev)
    S = torch.empty((V, len(t_np), n, k), dtype=torch.float64, device=dev)
    st, ns, nr = (torch.empty((V,), dtype=torch.int32, device=dev) for _ in range(3))
    o = dict(m5.integrator_options, method='implicit_extrap')
    _lib.implicit_adaptive_defaults(o, ())
    opts = _lib.make_opts('implicit_extrap', order=IEX_ORDER, rtol=o['rtol'], atol=o['atol'])
    ms = _events(torch, dev, lambda: m5.device_model.sens_dev(P, t5, None, opts, Y, S, st, ns, nr), reps)
    macro, rej = int(ns.sum().item()), int(nr.sum().item())
    euler = (macro + rej) * (IEX_ORDER * (IEX_ORDER + 1) // 2)
    out = {"workload": "dense stiff network dstiff48 (48 states, 1200 non-zeros of df/dy, 48 sensitivity columns: 2352 ODEs), "
                       "%d vectors, 16 output times, SBM_IMPLICIT_EXTRAP order 8 at default tolerances, dense LU row-distributed "
                       "over the lanes + per-column substitution on the VALU (no MFMA tiles on the implicit path)" % V,
           "ms": ms, "vectors": V, "macro_steps_per_vector": macro / V, "rejected_macro_steps": rej, "euler_steps": euler,
           "value": euler / (ms * 1e-3), "unit": "ODE-steps/s", "failed_vectors": int((st != 0).sum().item()),
           "ms_per_1000_vectors": ms * 1000.0 / V}
    gdir = os.path.join(B.REPO, 'tests', 'golden')
    if os.path.exists(os.path.join(gdir, 'dstiff48_ref.npz')):
        from oracle.tolerances import parity_err
        g, gt = np.load(os.path.join(gdir, 'dstiff48_ref.npz')), np.load(os.path.join(gdir, 'dstiff48_tight.npz'))
        nv = len(g['P'])
        assert np.array_equal(Pn[:nv], g['P'])
        Yg, Sg = Y[:nv, 1:].cpu().numpy(), S[:nv, 1:].cpu().numpy().reshape(nv, len(t_np) - 1, -1)
        out["parity_of_timed_pass"] = {
            "vectors_checked": nv,
            "worst_state_err_vs_reference_in_tolerance_units": max(parity_err(Yg[v], g['Y'][v]) for v in range(nv)),
            "worst_sens_err_vs_reference_in_tolerance_units": max(parity_err(Sg[v], g['S'][v]) for v in range(nv)),
            "worst_state_err_vs_tight_solution": max(parity_err(Yg[v], gt['Y'][v]) for v in range(nv)),
            "worst_sens_err_vs_tight_solution": max(parity_err(Sg[v], gt['S'][v]) for v in range(nv)),
            "reference_vs_tight_sens": max(parity_err(g['S'][v], gt['S'][v]) for v in range(nv))}
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


RUNNERS = {'configs1': run_configs1, 'configs3': run_configs3, 'configs4': run_configs4, 'configs4_fixed': run_configs4_fixed,
           'fit': run_fit, 'dense': run_dense, 'dop853': run_dop853, 'dense_stiff': run_dense_stiff}
ORDER = ['configs1', 'configs3', 'configs4', 'fit', 'dense', 'dense_stiff']
