#!/usr/bin/env python
"""bench.py -- ensemble ODE-steps/sec of the 20-state / 40-parameter model with forward
sensitivities (BASELINE.json metric; workload = configs[2], plus the residual/Jacobian
assembly that consumes it).

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: theta -> p gather, Dormand-Prince 5(4) integration of the 820-equation
augmented system for 4096 parameter vectors (per GPU), and the fused sample + residual +
Jacobian assembly of a one-experiment Project on top (64 rows x 40 parameters per vector).
value = accepted integrator steps of all trajectories of all ranks / wall time.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the ensemble is sharded by vector index (weak scaling: 4096 vectors per GPU, no
data-path collective); RCCL carries only the all-gather of the per-vector residual norms.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_AUG = 820                  # 20 + 20*40 coupled ODEs
BYTES_PER_STEP = 2 * 8 * N_AUG   # read + write the augmented state once per step
V_PER_GPU = 4096


def build_workload(model, gm, n_vectors, rank):
    """configs[2] as a Project: one experiment, species 4/9/14/19 measured at 16 times."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    import warnings
    # synthetic data: nominal trajectory x (1 + 5 % noise), seed 7; generated ON THE GPU PATH
    # (bench inputs must not depend on the oracle)
    p_nom = models_zoo.cascade_nominal_params()
    grid = np.linspace(0, models_zoo.CASCADE_T_END, 1000)
    idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)
    y = model.simulate(p_nom, np.concatenate([[0.0], grid[idx]]))[1:]
    rng = np.random.default_rng(7)
    ms = []
    for v in models_zoo.CASCADE_MEASURED_SPECIES:
        data = y[:, v] * (1.0 + 0.05 * rng.standard_normal(len(idx)))
        ms.append(TimecourseMeasurement('s%d' % v, data, models_zoo.CASCADE_MEASURE_TIMES.copy(),
                                        0.05 * np.abs(data) + 0.01))
    exp = Experiment('exp_0', ms)
    mapping = {('s%d' % v): ('direct', v) for v in models_zoo.CASCADE_MEASURED_SPECIES}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj = Project(model, [exp], {'Global': list(gm.param_order)}, mapping,
                       sf_groups=['s%d' % v for v in models_zoo.CASCADE_MEASURED_SPECIES])
    # global ensemble of world*4096 vectors; this rank owns a contiguous block
    theta_all, _ = models_zoo.cascade_ensemble(n_vectors * (rank + 1))
    theta = theta_all[rank * n_vectors:(rank + 1) * n_vectors]
    # project vector order == model order here (all Global, listed in model order)
    order = [gm.param_order.index(name) for name, _ in proj.get_ordered_project_params()]
    return proj, np.ascontiguousarray(theta[:, order]), grid[idx]


def cpu_baseline(gm, theta_rows, budget_s=12.0, max_vectors=4096):
    """The oracle (SciPy odeint restatement of OdeModel.calc_jacobian, compiled C RHS standing in
    for the reference's numba) timed on ONE host core over a bounded sample of the same ensemble."""
    from oracle import odeint_oracle as oo
    grid = np.linspace(0, 100.0, 1000)
    steps, n, t0 = 0, 0, time.perf_counter()
    gm.c_library()
    for row in theta_rows[:max_vectors]:
        _, info = oo.calc_jacobian(gm, np.exp(row), grid, use_c=True, full_output=True)
        steps += int(info['nst'][-1])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "ODE-steps/s", "cores": 1, "kind": "port",
            "sample": "first %d vectors of the ensemble, state+sensitivity system (820 ODEs), "
                      "scipy.integrate.odeint rtol=atol=1e-10 on the reference's 1000-point grid, "
                      "compiled C RHS; %.1f s, %d LSODA steps" % (n, dt, steps),
            "ms_per_vector": 1e3 * dt / max(n, 1)}


def residual_parity(gm, proj, theta_rows, res_gpu, jac_gpu, n_check=3):
    """BASELINE.json's second figure: residual (and Jacobian) error of the timed GPU pass against the
    SciPy restatement of the reference's Project on the same inputs, for the first few vectors.  Reported
    in the units of the parity tolerance |gpu - ref| <= 1e-8 |ref| + 5e-9 of the underlying trajectories;
    residual rows are trajectories divided by sigma ~ 0.06, so their floor scales by 1 / sigma;
    ``residual_rel_err`` is |r_gpu - r_scipy|_2 / |r_scipy|_2, the worst of the checked vectors."""
    from oracle.project_oracle import ProjectOracle
    po = ProjectOracle(gm, list(proj.experiments), proj._model_parameter_settings,
                       {k: v for k, v in proj._measurement_to_model_map_raw.items()},
                       sf_groups=[g if len(g) > 1 else g[0] for g in proj._loss_function.groups])
    sig = proj.descriptor_arrays()['row_sigma']
    worst_r = worst_rel = worst_j = 0.0
    for v in range(n_check):
        rr = po.residuals(theta_rows[v])
        Jr = po.calc_project_jacobian(theta_rows[v])
        d = np.abs(res_gpu[v] - rr)
        worst_r = max(worst_r, float(np.max(d / (1e-8 * np.abs(rr) + 5e-9 / sig))))
        worst_rel = max(worst_rel, float(np.linalg.norm(res_gpu[v] - rr) / np.linalg.norm(rr)))
        worst_j = max(worst_j, float(np.max(np.abs(jac_gpu[v] - Jr) / (1e-8 * np.abs(Jr) + 1e-8 * np.abs(Jr).max()))))
    return {"vectors_checked": n_check, "residual_err_in_tolerance_units": worst_r,
            "residual_rel_err": worst_rel, "jacobian_err_in_tolerance_units": worst_j,
            "tolerance": "|gpu - scipy| <= 1e-8 |scipy| + 5e-9 / sigma (residuals); 1e-8 (|J| + max|J|) (Jacobian)",
            "reference": "ProjectOracle: scipy.integrate.odeint rtol=atol=1e-10, reference_compat Jacobian"}


def _cpu_worker(rows_budget):
    """one process of the all-cores baseline: integrates its share of the sample for `budget` seconds"""
    rows, budget = rows_budget
    from sysbio_modeling_amd.symbolic import zoo_model
    from oracle import odeint_oracle as oo
    gm = zoo_model('cascade20')
    gm.c_library()
    grid = np.linspace(0, 100.0, 1000)
    steps, n, t0 = 0, 0, time.perf_counter()
    for row in rows:
        _, info = oo.calc_jacobian(gm, np.exp(row), grid, use_c=True, full_output=True)
        steps += int(info['nst'][-1])
        n += 1
        if time.perf_counter() - t0 > budget:
            break
    return steps, n, time.perf_counter() - t0


def cpu_baseline_all_cores(theta_rows, budget_s=10.0):
    """The same oracle on every host core the process may use: one worker
    process per core, each timing its own share.  Runs BEFORE this process touches the GPU (the workers
    are spawned, not forked)."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))     # a one-GPU box's CPU share
    per = max(1, len(theta_rows) // cores)
    # every worker starts at its own offset of the ensemble and stops on its time budget
    chunks = [(np.roll(theta_rows, -i * per, axis=0), budget_s) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context('spawn').Pool(cores) as pool:
        out = pool.map(_cpu_worker, chunks)
    rate = sum(st / dt for st, _, dt in out)          # every worker timed on its own clock
    return {"value": rate, "unit": "ODE-steps/s", "cores": cores, "kind": "port",
            "sample": "%d vectors over %d worker processes, %.1f s each (wall %.1f s incl. start-up); same "
                      "odeint call as cpu_baseline" % (sum(n for _, n, _ in out), cores, budget_s,
                                                       time.perf_counter() - t0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--vectors', type=int, default=V_PER_GPU, help="parameter vectors per GPU")
    ap.add_argument('--method', default='dopri45', choices=['dopri45', 'rk4'])
    ap.add_argument('--rk4-steps', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help="time the CPU oracle and exit (no GPU needed)")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from sysbio_modeling_amd import models_zoo as _mz
        cpu_all = cpu_baseline_all_cores(_mz.cascade_ensemble(args.vectors)[0])
    if args.cpu_baseline_only:
        from sysbio_modeling_amd.symbolic import zoo_model as _zm
        from sysbio_modeling_amd import models_zoo as _mz
        print(json.dumps({"cpu_baseline": cpu_baseline(_zm('cascade20'), _mz.cascade_ensemble(args.vectors)[0]),
                          "cpu_baseline_all_cores": cpu_all}))
        return
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal knob (not used by the driver): SBM_BENCH_REHEARSAL=1 runs every rank on GPU 0 over gloo, so
    # that the multi-rank code path can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get('SBM_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
        os.environ['LOCAL_RANK'] = '0'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from sysbio_modeling_amd import _lib
    from sysbio_modeling_amd.symbolic import zoo_model
    from sysbio_modeling_amd.model import OdeModel
    gm = zoo_model('cascade20')
    model = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, use_jit=False)
    model.enable_jit(_lib.Context(local_rank))
    V = args.vectors
    proj, theta, t_meas = build_workload(model, gm, V, rank)
    lib = _lib.load_library()
    pj = proj._device()
    q, R, G = proj.n_project_params, proj.n_project_residuals, 4
    f64, i32 = torch.float64, torch.int32
    th = torch.from_numpy(theta).to(dev)
    out = dict(sims=torch.empty((V, R), dtype=f64, device=dev), res=torch.empty((V, R), dtype=f64, device=dev),
               J=torch.empty((V, R, q), dtype=f64, device=dev), sf=torch.empty((V, G), dtype=f64, device=dev),
               norms=torch.empty((V,), dtype=f64, device=dev), status=torch.empty((V,), dtype=i32, device=dev),
               nsteps=torch.empty((V,), dtype=i32, device=dev))
    gathered = torch.empty((world * V,), dtype=f64, device=dev) if world > 1 else None
    if args.method == 'dopri45':
        opts = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12)
    else:
        opts = _lib.make_opts('rk4', n_steps=args.rk4_steps, t_end=100.0)
    p = _lib.dev_ptr

    def step():
        _lib.check(lib.sbm_jacobian_batch(pj, p(th), V, ctypes.byref(opts), p(out['sims']), p(out['res']),
                                          p(out['J']), None, p(out['sf']), None, p(out['norms']), None,
                                          p(out['status']), p(out['nsteps'])), 'sbm_jacobian_batch')
        if world > 1:
            dist.all_gather_into_tensor(gathered, out['norms'])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    steps_per_pass = int(out['nsteps'].sum().item())
    n_bad = int((out['status'] != 0).sum().item())
    stats = torch.tensor([dt, float(steps_per_pass), float(n_bad)], dtype=f64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, total_steps_per_pass, n_bad = float(mx[0]), float(sm[1]), int(sm[2])
    else:
        total_steps_per_pass = float(steps_per_pass)
    value = total_steps_per_pass * args.steps / dt

    # ---- roofline of the dominant kernel: the sensitivity integrator alone, HIP events on its stream ----
    dm = model.device_model
    theta_p = torch.exp(th)  # all parameters Global and in model order: p = exp(theta)
    tg = torch.from_numpy(np.concatenate([[0.0], t_meas])).to(dev)
    Yk = torch.empty((V, len(tg), 20), dtype=f64, device=dev)
    Sk = torch.empty((V, len(tg), 20, 40), dtype=f64, device=dev)
    ns_k = torch.empty((V,), dtype=i32, device=dev)
    for _ in range(max(1, args.warmup)):
        dm.sens_dev(theta_p, tg, None, opts, Yk, Sk, None, ns_k, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(args.steps):
        dm.sens_dev(theta_p, tg, None, opts, Yk, Sk, None, ns_k, None)
    e1.record()
    torch.cuda.synchronize(dev)
    k_ms = e0.elapsed_time(e1) / args.steps
    k_steps = int(ns_k.sum().item())
    achieved = k_steps * BYTES_PER_STEP / (k_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as fh:
            traffic = json.load(fh).get(args.method, {}).get('hbm_bytes_per_launch')
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "sbm_sens_rowgroup_kernel<cascade20,%s>" % args.method, "kernel_ms": k_ms,
                "steps_per_launch": k_steps, "algorithmic_bytes_per_step": BYTES_PER_STEP,
                "kernel_steps_per_s": k_steps / (k_ms * 1e-3),
                "note": "algorithmic bytes (2*8*820 B per accepted step) / kernel time; the kernel keeps the "
                        "state in VGPRs, so real HBM traffic ('traffic', PMC) is far below this figure and the "
                        "binding resource is VALU issue (profiles/r01e/pmc_summary.json)"}

    extras = {}
    if not args.no_extras and rank == 0 and world == 1:   # single-GPU micro-benchmarks
        def time_kernel(kind, o, reps=3):
            for _ in range(1):
                (dm.sens_dev(theta_p, tg, None, o, Yk, Sk, None, ns_k, None) if kind == 'sens'
                 else dm.simulate_dev(theta_p, tg, None, o, Yk, None, ns_k, None))
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                (dm.sens_dev(theta_p, tg, None, o, Yk, Sk, None, ns_k, None) if kind == 'sens'
                 else dm.simulate_dev(theta_p, tg, None, o, Yk, None, ns_k, None))
            b.record()
            torch.cuda.synchronize(dev)
            ms = a.elapsed_time(b) / reps
            st = int(ns_k.sum().item())
            return {"ms": ms, "steps": st, "steps_per_s": st / (ms * 1e-3)}
        rk = _lib.make_opts('rk4', n_steps=args.rk4_steps, t_end=100.0)
        dp = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12)
        dp_rl = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant='row_lane')
        rk_rl = _lib.make_opts('rk4', n_steps=args.rk4_steps, t_end=100.0, variant='row_lane')
        dp_pw = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12, variant='per_wave')
        rk_pw = _lib.make_opts('rk4', n_steps=args.rk4_steps, t_end=100.0, variant='per_wave')
        extras = {"sens_rk4_fixed_%d" % args.rk4_steps: time_kernel('sens', rk),
                  "sens_dopri45": time_kernel('sens', dp),
                  "sens_dopri45_row_lane_variant": time_kernel('sens', dp_rl),
                  "sens_rk4_fixed_%d_row_lane_variant" % args.rk4_steps: time_kernel('sens', rk_rl),
                  "sens_dopri45_per_wave_variant": time_kernel('sens', dp_pw),
                  "sens_rk4_fixed_%d_per_wave_variant" % args.rk4_steps: time_kernel('sens', rk_pw),
                  "state_only_dopri45_configs1": time_kernel('state', dp),
                  "state_only_rk4_fixed_%d" % args.rk4_steps: time_kernel('state', rk)}
        # BASELINE configs[3]: 8 experiment settings x 1024 vectors, residual + Jacobian assembly
        import warnings
        from sysbio_modeling_amd import models_zoo
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            p4, th4 = models_zoo.cascade_config4_project(model)
        t4 = torch.from_numpy(models_zoo.config4_ensemble(th4, 1024)).to(dev)
        p4.evaluate_batch(t4, jacobian=True, want=('jacobian',))
        torch.cuda.synchronize(dev)
        a4, b4 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a4.record()
        for _ in range(3):
            o4 = p4.evaluate_batch(t4, jacobian=True, want=('jacobian',))
        b4.record()
        torch.cuda.synchronize(dev)
        ms4 = a4.elapsed_time(b4) / 3
        st4 = int(o4['n_steps'].sum().item())
        extras["configs3_project_8exp_x_1024vec"] = {"ms": ms4, "steps": st4, "steps_per_s": st4 / (ms4 * 1e-3),
                                                     "rows": 512, "params": 68,
                                                     "failed_vectors": int((o4['status'] != 0).sum().item())}

        # BASELINE configs[4]: stiff 50-state cascade, 2550 coupled ODEs, implicit midpoint, 4096 vectors.
        # One launch per Richardson level; the parity setting of tests/test_gpu_implicit.py is 4096 + 8192
        # steps (extrapolate=1), timed here: the 2048-step launch, whose cost scales linearly.
        gm5 = zoo_model('stiff50')
        m5 = OdeModel(gm5.model, gm5.sens_model, gm5.n_vars, gm5.param_order, use_jit=False)
        P5 = torch.from_numpy(models_zoo.stiff_ensemble(4096)[1]).to(dev)
        t5 = torch.tensor([5.0, models_zoo.STIFF_T_END], dtype=f64, device=dev)
        Y5 = torch.empty((4096, 2, 50), dtype=f64, device=dev)
        S5 = torch.empty((4096, 2, 50, 50), dtype=f64, device=dev)
        st5 = torch.empty((4096,), dtype=i32, device=dev)
        ns5 = torch.empty((4096,), dtype=i32, device=dev)
        nw5 = torch.empty((4096,), dtype=i32, device=dev)
        o5 = _lib.make_opts('implicit_midpoint', rtol=1e-10, atol=1e-12, n_steps=2048, t_end=models_zoo.STIFF_T_END)
        m5.device_model.sens_dev(P5, t5, None, o5, Y5, S5, st5, ns5, nw5)
        torch.cuda.synchronize(dev)
        a5, b5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a5.record()
        m5.device_model.sens_dev(P5, t5, None, o5, Y5, S5, st5, ns5, nw5)
        b5.record()
        torch.cuda.synchronize(dev)
        ms5 = a5.elapsed_time(b5)
        n5 = int(ns5.sum().item())
        extras["configs4_stiff50_implicit_midpoint_2048_steps"] = {
            "ms": ms5, "steps": n5, "steps_per_s": n5 / (ms5 * 1e-3), "n_equations": 2550,
            "newton_iterations_per_step": 1.0 + float(nw5.sum().item()) / n5,
            "algorithmic_GBps": n5 / (ms5 * 1e-3) * 2 * 8 * 2550 / 1e9,
            "failed_vectors": int((st5 != 0).sum().item())}

        # A model beyond one row / one column per lane: 70 states, 140 parameters, 9870 coupled ODEs per trajectory
        # (two state rows per lane, sensitivity columns in 14 chunks of 10, one wavefront each).  The plugin is
        # built here (hipcc, ~15 s); a failure to build must not cost the headline line.
        try:
            from sysbio_modeling_amd.symbolic import GeneratedModel
            gm7 = GeneratedModel(models_zoo.cascade_spec(70, name='cascade70'))
            m7 = OdeModel(gm7.model, gm7.sens_model, gm7.n_vars, gm7.param_order, use_jit=False)
            V7 = 1024
            P7 = torch.from_numpy(models_zoo.cascade_ensemble(V7, n=70, spread=0.3)[1]).to(dev)
            t7 = torch.tensor([50.0, 100.0], dtype=f64, device=dev)
            Y7 = torch.empty((V7, 2, 70), dtype=f64, device=dev)
            S7 = torch.empty((V7, 2, 70, 140), dtype=f64, device=dev)
            st7 = torch.empty((V7,), dtype=i32, device=dev)
            ns7 = torch.empty((V7,), dtype=i32, device=dev)
            o7 = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12)
            m7.device_model.sens_dev(P7, t7, None, o7, Y7, S7, st7, ns7, None)
            torch.cuda.synchronize(dev)
            a7, b7 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a7.record()
            m7.device_model.sens_dev(P7, t7, None, o7, Y7, S7, st7, ns7, None)
            b7.record()
            torch.cuda.synchronize(dev)
            ms7 = a7.elapsed_time(b7)
            n7 = int(ns7.sum().item())
            extras["large_model_cascade70_dopri45"] = {
                "ms": ms7, "steps": n7, "steps_per_s": n7 / (ms7 * 1e-3), "n_equations": 70 + 70 * 140, "vectors": V7,
                "algorithmic_GBps": n7 / (ms7 * 1e-3) * 2 * 8 * (70 + 70 * 140) / 1e9,
                "failed_vectors": int((st7 != 0).sum().item())}
        except Exception as e:   # noqa: BLE001
            extras["large_model_cascade70_dopri45"] = {"error": repr(e)[:200]}

    result = {
        "metric": "ensemble ODE-steps/sec (20-state model + fwd sens)",
        "value": value, "unit": "ODE-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: cascade20 (20 states, 40 params) with full forward sensitivities "
                               "(820 coupled ODEs), %d parameter vectors per GPU, %s; step = theta->p gather + "
                               "integration + fused residual/Jacobian assembly (64 rows x 40 params, 4 scale "
                               "factors)%s" % (V, "DOPRI45 rtol=1e-9 atol=1e-12, 16 output times"
                                               if args.method == 'dopri45' else "RK4 fixed, %d steps" % args.rk4_steps,
                                               " + RCCL all-gather of residual norms" if world > 1 else ""),
                   "vectors_per_gpu": V, "n_equations": N_AUG, "integrator": args.method,
                   "accepted_steps_per_pass": total_steps_per_pass, "failed_vectors": n_bad},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(gm, theta)
        result["cpu_baseline_all_cores"] = cpu_all
        result["cpu_baseline"]["parity_of_timed_pass"] = residual_parity(
            gm, proj, theta, out['res'][:3].cpu().numpy(), out['J'][:3].cpu().numpy())
    elif rank == 0:
        result["cpu_baseline"] = None
    if extras:
        result["extras"] = extras
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()   # rank 0 arrives late (extras run after the timed region)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
