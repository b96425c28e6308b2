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
    python bench.py --gpus N ...          spawns N ranks itself (torch.distributed.run, one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the ensemble is sharded by vector index (weak scaling: 4096 vectors per GPU, no
data-path collective); RCCL carries only the all-gather of the per-vector residual norms.

At N = 1 the line also carries, under "configs", BASELINE.json's other configurations -- configs[1]
(state only), configs[3] (8-experiment Project), configs[4] (stiff 50-state model) -- each with its own
`roofline` and `cpu_baseline` objects, "fit" (end-to-end multi-start fitting against the reference's
serial leastsq pattern) and "dense" (the MFMA question).  `--only NAME` runs one of them alone (what the
rocprofv3 passes of scripts/profile_gpu.sh use).  The side workloads live in bench_configs.py.
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
N_SIMD = 1024                # 256 CUs x 4 SIMDs
MAX_CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md: max clock
N_AUG = 820                  # 20 + 20*40 coupled ODEs
BYTES_PER_STEP = 2 * 8 * N_AUG   # read + write the augmented state once per step
V_PER_GPU = 4096
COUNTERS = os.path.join(REPO, 'profiles', 'kernel_counters.json')   # written by scripts/summarize_profile.py


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


# ---------------------------------------------------------------------------
# CPU legs: the oracle (SciPy odeint restatement of the reference, compiled C RHS standing in for its numba)
# timed on the GPU box's host cores over BOUNDED samples of the same workloads
# ---------------------------------------------------------------------------
def cpu_baseline(gm, theta_rows, budget_s=8.0, max_vectors=4096, sens=True, t_end=100.0):
    """One host core; the reference's exact odeint call per vector on its 1000-point grid."""
    from oracle import odeint_oracle as oo
    grid = np.linspace(0, t_end, 1000)
    steps, n, t0 = 0, 0, time.perf_counter()
    gm.c_library()
    for row in theta_rows[:max_vectors]:
        if sens:
            _, info = oo.calc_jacobian(gm, np.exp(row), grid, use_c=True, full_output=True)
        else:
            _, info = oo.simulate(gm, np.exp(row), grid, use_c=True, full_output=True)
        steps += int(info['nst'][-1])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "ODE-steps/s", "cores": 1, "kind": "port",
            "sample": "first %d vectors of the ensemble, %s, scipy.integrate.odeint rtol=atol=1e-10 on the "
                      "reference's 1000-point grid, compiled C RHS; %.1f s, %d LSODA steps"
                      % (n, "state+sensitivity system (%d ODEs)" % (gm.n_vars * (1 + gm.n_sens)) if sens
                         else "state system (%d ODEs)" % gm.n_vars, dt, steps),
            "ms_per_vector": 1e3 * dt / max(n, 1)}


def project_oracle_of(gm, proj):
    from oracle.project_oracle import ProjectOracle
    return ProjectOracle(gm, list(proj.experiments), proj._model_parameter_settings,
                         {k: v for k, v in proj._measurement_to_model_map_raw.items()},
                         sf_groups=[g if len(g) > 1 else g[0] for g in proj._loss_function.groups])


def residual_parity(gm, proj, theta_rows, res_gpu, jac_gpu, picks):
    """BASELINE.json's second figure: residual (and Jacobian) error of the timed GPU pass against the SciPy
    restatement of the reference's Project on the same inputs, for ``picks`` (indices spread over the batch), AND the
    arbitration of that difference: the same rows from a tight integration, with the GPU's and SciPy's distance to it.
    Tolerance (oracle/tolerances.py): the sampled trajectories agree with the reference's LSODA to
    |gpu - ref| <= 1e-8 |ref| + 5e-9 (the absolute term is LSODA's own noise at atol = 1e-10); residual and
    Jacobian rows get the first-order propagation of exactly that through the reference's formulas.  The oracle runs
    are spread over the host cores (bench_configs.oracle_parity_pool)."""
    import bench_configs as bc
    out = bc.oracle_parity_pool(proj, proj.descriptor_arrays(), theta_rows, res_gpu, jac_gpu, picks, tight=True)
    out["reference"] = "ProjectOracle: scipy.integrate.odeint rtol=atol=1e-10, reference_compat Jacobian"
    return out


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


def cpu_baseline_all_cores(theta_rows, budget_s=6.0):
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


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks through torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1) as a CHILD process and hand its stdout / exit code through.  Nothing in this
    process initialises a GPU."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------
# rooflines
# ---------------------------------------------------------------------------
_CURRENT_STAMPS = {}      # module name -> build stamp of what this process runs (filled by main / the runners)


def _counters():
    """profiles/kernel_counters.json, minus the entries whose module (core library / model plugin) has been rebuilt
    from different sources since the profile was taken: stale counters are dropped, not printed."""
    try:
        with open(COUNTERS) as fh:
            raw = json.load(fh)
    except (OSError, ValueError):
        return {}
    out = {}
    for key, e in raw.items():
        mod, st = e.get('module'), e.get('build_stamp')
        if mod is None or st is None or _CURRENT_STAMPS.get(mod) != st:
            continue
        out[key] = e
    return out


FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X: half the 157.3 TFLOP/s fp32 vector rate of MI355X_MICROARCH.md


def flops_per_step(gm, method, sens=True, evals_per_step=None):
    """USEFUL fp64 operations of one accepted step of one trajectory, from the emitter's own operation counts
    (GeneratedModel.flop_counts): right-hand-side evaluations + J_y S products + the method's linear combinations.
    Selects, the step controller, LDS hand-offs and redundant work are NOT counted -- that is the point of the figure.
    Returns (F_step, formula)."""
    c = gm.flop_counts()
    n, k = gm.n_vars, (gm.n_sens if sens else 0)
    N = n * (1 + k)
    if sens:
        rhs = c['f_jac'] + 2 * c['nnz_jy'] * k + c['nnz_jp']
        rhs_txt = "(f + J_y + J_p: %d) + 2 nnz(J_y) k (%d) + nnz(J_p) (%d)" % (c['f_jac'], 2 * c['nnz_jy'] * k, c['nnz_jp'])
    else:
        rhs = c['f']
        rhs_txt = "f: %d" % c['f']
    if method == 'dopri45':      # 21 a_ij + 7 b_j (the FSAL stage argument) + the error estimate's e_j: 38 multiply-adds per element
        return 6 * rhs + 2 * 38 * N, "6 stages x [%s] + 38 multiply-adds x %d elements (21 a_ij, 7 b_j, 7 e_j, accept / scale)" % (rhs_txt, N)
    if method == 'dop853':       # 50 a_ij, 8 b_j, 8 + 8 coefficients of the two embedded estimates
        return 12 * rhs + 2 * 74 * N, "12 stages x [%s] + 74 multiply-adds x %d elements" % (rhs_txt, N)
    if method == 'rk4':
        return 4 * rhs + 2 * 8 * N, "4 stages x [%s] + 8 multiply-adds x %d elements" % (rhs_txt, N)
    if method in ('implicit_euler', 'implicit_midpoint'):
        # one implicit step: `evals_per_step` Newton evaluations of f / J_y / J_p, each with one solve of the sparse
        # factors (2 nnz(L + U) - n operations) for the state, + one such solve and the J_p term per sensitivity column
        ev = float(evals_per_step or 1.0)
        solve = 2 * c['nnz_lu'] - n
        F = ev * (c['f_jac'] + solve + 2 * n) + k * (solve + (2 if method == 'implicit_midpoint' else 0) * n) + 2 * c['nnz_jp']
        return F, ("%.2f Newton evaluations x [(f + J_y + J_p: %d) + one sparse solve (%d) + residual (%d)] + %d columns x "
                   "one sparse solve + 2 nnz(J_p)" % (ev, c['f_jac'], solve, 2 * n, k))
    raise ValueError(method)


def fp64_rate(flops_step, formula, steps_per_s):
    t = flops_step * steps_per_s / 1e12
    return {"achieved": t, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": t / FP64_VECTOR_PEAK_TFLOPS,
            "useful_flops_per_step": flops_step, "formula": formula,
            "note": "USEFUL operations only (emitter's operation counts: right-hand sides, J_y S, the method's linear "
                    "combinations) against the fp64 vector peak; what the kernel ISSUES is roofline_valu_issue"}


def build_stamps(*models):
    """Fingerprints (flags + toolchain + source contents: sysbio_modeling_amd/build.py) of the native modules whose
    kernels the rooflines quote PMC counters for.  profiles/kernel_counters.json stores them next to every entry
    (scripts/summarize_profile.py); an entry whose module has been rebuilt since is NOT printed."""
    from sysbio_modeling_amd import build
    out = {}

    def stamp(path):
        try:
            with open(path + '.stamp') as fh:
                return fh.read().strip()[:16]
        except OSError:
            return None
    out['core'] = stamp(build.CORE_LIB)
    for gm in models:
        try:
            out[gm.name] = stamp(gm.plugin_path(build_if_missing=False))
        except Exception:   # noqa: BLE001
            out[gm.name] = None
    return out


def hbm_roofline(kernel, key, k_ms, k_steps, bytes_per_step, note):
    """Algorithmic-byte roofline of SURVEY.md section 8(d): bytes_per_step x accepted steps of one launch / the
    launch's duration (HIP events on the kernel's stream), against the HBM peak.  `traffic` = HBM bytes per launch
    from the PMC passes of this round's profile (profiles/kernel_counters.json: (2 FETCH_SIZE + WRITE_SIZE) KiB,
    separate passes, the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md), null when no profile is committed."""
    c = _counters().get(key, {})
    achieved = k_steps * bytes_per_step / (k_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": c.get('hbm_bytes_per_launch'),
            "traffic_source": c.get('source'),
            "kernel": kernel, "kernel_ms": k_ms, "steps_per_launch": k_steps,
            "algorithmic_bytes_per_step": bytes_per_step, "kernel_steps_per_s": k_steps / (k_ms * 1e-3), "note": note}


def valu_roofline(key, k_ms, k_steps):
    """The resource that actually binds the register-resident integrators: VALU instruction issue.
    achieved = (VALU wave-instructions per accepted step, PMC: SQ_INSTS_VALU / steps) x steps per second;
    peak = 1024 SIMDs x 2.4 GHz / (cycles one such instruction holds its SIMD's issue port, PMC:
    4 SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU -- the counter ticks in quad-cycles).  Both PMC figures come from the
    committed profile of the same kernel; the rate is measured live."""
    c = _counters().get(key, {})
    if not c.get('valu_insts_per_step') or not c.get('cycles_per_valu_inst'):
        return None
    achieved = c['valu_insts_per_step'] * k_steps / (k_ms * 1e-3)
    peak = N_SIMD * MAX_CLOCK_HZ / c['cycles_per_valu_inst']
    return {"bound": "valu_issue", "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
            "frac": achieved / peak, "valu_insts_per_step": c['valu_insts_per_step'],
            "cycles_per_valu_inst": c['cycles_per_valu_inst'],
            "valu_busy_fraction_pmc": c.get('valu_busy_fraction'), "source": c.get('source'),
            "note": "peak at the 2.4 GHz maximum clock; the chip holds less under load, so frac understates how "
                    "close to the issue limit the kernel runs (valu_busy_fraction_pmc is the direct reading)"}


def issue_roofline(key, k_ms, k_steps):
    """Kernels at ONE wavefront per SIMD (the stiff integrators: three copies of a sensitivity column in registers): a
    wavefront issues one instruction of any kind per four cycles, so the bound is instruction issue over ALL kinds.
    achieved = (VALU + SALU + LDS wave-instructions per step, PMC) x steps per second -- branches, waits and vector-memory
    instructions are not counted by these counters, so the figure understates what is issued; peak = 1024 SIMDs x
    2.4 GHz / 4."""
    c = _counters().get(key, {})
    if not c.get('valu_insts_per_step') or c.get('waves_per_simd', 2) > 1.01:
        return None
    per_step = c['valu_insts_per_step'] + c.get('salu_insts_per_step', 0.0) + c.get('lds_insts_per_step', 0.0)
    achieved = per_step * k_steps / (k_ms * 1e-3)
    peak = N_SIMD * MAX_CLOCK_HZ / 4.0
    return {"bound": "instruction_issue", "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
            "frac": achieved / peak, "counted_insts_per_step": per_step, "valu_insts_per_step": c['valu_insts_per_step'],
            "salu_insts_per_step": c.get('salu_insts_per_step'), "lds_insts_per_step": c.get('lds_insts_per_step'),
            "any_inst_active_share_pmc": c.get('any_inst_active_share_of_wave_lifetime'),
            "valu_busy_fraction_pmc": c.get('valu_busy_fraction'), "lds_conflict_share_pmc": c.get('lds_conflict_share'),
            "source": c.get('source'),
            "note": "one wavefront per SIMD: one instruction of any kind per 4 cycles (scripts/dev_latency_ubench.hip)"}


_REAL_STDOUT = None
T_START = time.perf_counter()


def protect_stdout():
    """The JSON line must be the only thing on stdout, but ODEPACK (the CPU legs' LSODA) writes its warnings to
    Fortran unit 6 = the C-level stdout, buffered until exit.  So: keep a private duplicate of the real stdout for
    the JSON line and point file descriptor 1 at /dev/null for everything else, for the life of the process."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(null, 1)         # (not stderr: hundreds of "lsoda-- ..." lines from one CPU leg bury everything else there)
        os.close(null)


def _write_all(fd, data):
    """os.write may write less than it was given (a pipe that is full): loop until the buffer is out."""
    view = memoryview(data)
    while len(view):
        n = os.write(fd, view)
        view = view[n:]


def emit(obj):
    line = (json.dumps(obj, separators=(',', ':')) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _write_all(1, line)
    else:
        _write_all(_REAL_STDOUT, line)


LINE_LIMIT = 4096          # the driver keeps an 8 KB tail of stdout: the headline line stays well inside it
FULL_OUT = os.path.join(REPO, 'gpurun_out', 'bench_full.json')


def _sig(x, digits=6):
    """numbers at six significant digits, everything else unchanged (the full record keeps full precision)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float('%.*g' % (digits, x)) if np.isfinite(x) else None
    if isinstance(x, (int, str)):
        return x
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _compact_roofline(r, rv=None):
    """{bound, achieved, peak, unit, frac, traffic, kernel, kernel_ms, achieved_fp64{...}} (+ valu_issue{...}): numbers only."""
    if not isinstance(r, dict):
        return None
    out = _pick(r, ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'kernel_ms'))
    out['traffic'] = r.get('traffic')
    if isinstance(r.get('achieved_fp64'), dict):
        out['achieved_fp64'] = _pick(r['achieved_fp64'], ('achieved', 'peak', 'unit', 'frac'))
    if isinstance(rv, dict):
        out['valu_issue'] = _pick(rv, ('achieved', 'peak', 'unit', 'frac', 'valu_busy_fraction_pmc'))
    return out


def _compact_side(name, c):
    """one side workload in a few numbers: time, rate, the fraction of the bound that applies, parity of the timed pass"""
    if not isinstance(c, dict):
        return None
    if 'error' in c:
        return {"error": str(c['error'])[:80]}
    out = _pick(c, ('ms', 'value', 'seconds', 'fits_per_s', 'converged', 'starts', 'cost_median', 'failed_vectors',
                    'n_gpus', 'scaling', 'macro_steps_per_vector', 'speedup_vs_one_core', 'vectors_per_rank',
                    'gathered_norms_match_local_block_on_every_rank'))
    r = c.get('roofline') or (c.get('dopri45') or {}).get('roofline')
    if name == 'configs1' and isinstance(c.get('dopri45'), dict):
        out.update(_pick(c['dopri45'], ('ms', 'value', 'failed_vectors')))
    rv = c.get('roofline_valu_issue') or (c.get('dopri45') or {}).get('roofline_valu_issue')
    if isinstance(r, dict):
        # the bound that applies: VALU issue where the byte model says > 1 or the PMC figures exist, else the byte model
        if r.get('bound') == 'instruction_issue':
            out['roofline'] = {"bound": "instruction_issue", "frac": r.get('frac'), "valu_busy": r.get('valu_busy_fraction_pmc'),
                               "any_inst_active": r.get('any_inst_active_share_pmc'),
                               "hbm_model_frac": (c.get('roofline_hbm_model') or {}).get('frac')}
        elif isinstance(rv, dict):
            out['roofline'] = {"bound": "valu_issue", "frac": rv.get('frac'), "valu_busy": rv.get('valu_busy_fraction_pmc'),
                               "hbm_model_frac": r.get('frac')}
        else:
            out['roofline'] = {"bound": r.get('bound'), "frac": r.get('frac')}
        if isinstance(r.get('achieved_fp64'), dict):
            out['roofline']['fp64_frac'] = r['achieved_fp64'].get('frac')
    par = c.get('parity_of_timed_pass') or (c.get('cpu_baseline') or {}).get('parity_of_timed_pass')
    if isinstance(par, dict):
        keep = ('vectors_checked', 'residual_err_in_tolerance_units', 'jacobian_err_in_tolerance_units', 'gpu_norm_rel_err_vs_tight',
                'scipy_norm_rel_err_vs_tight', 'gpu_within_1e-8_of_tight', 'worst_state_err_vs_tight_solution',
                'worst_sens_err_vs_tight_solution', 'vectors_passed_by_arbitration', 'vectors_failed')
        out['parity'] = {k: par[k] for k in keep if k in par}
    cb = c.get('cpu_baseline')
    if isinstance(cb, dict):
        out['cpu'] = _pick(cb, ('value', 'unit', 'cores', 'finished', 'seconds'))
    if name == 'fit' and isinstance(c.get('with_priors'), dict) and 'error' not in c['with_priors']:
        wp = c['with_priors']
        out['with_priors'] = _pick(wp, ('seconds', 'fits_per_s', 'converged', 'starts', 'cost_median', 'speedup_vs_one_core'))
        if isinstance(wp.get('cpu_baseline'), dict):
            out['with_priors']['cpu'] = _pick(wp['cpu_baseline'], ('value', 'unit', 'cores', 'finished', 'seconds', 'costs_agree_to_1e-6'))
    if name == 'dense':
        out = {k: {"valu_ms": v['valu']['ms'], "mfma_ms": v['mfma']['ms']} for k, v in c.items()
               if isinstance(v, dict) and 'valu' in v and 'mfma' in v}
    return out


def compact_line(full, full_path=None):
    """The LAST stdout line of bench.py: the contract's keys + `roofline` + `cpu_baseline`, numbers and short labels
    only, at most LINE_LIMIT bytes.  Everything else (side configurations in full, kernel variants, formulas, notes)
    is in the side file `full_path`.  Sheds the optional parts (smallest value first) if a line would still be too long."""
    cfg = full.get('config', {})
    line = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                     'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    line['config'] = dict(_pick(cfg, ('vectors_per_gpu', 'n_equations', 'integrator', 'rtol', 'atol',
                                      'accepted_steps_per_pass', 'failed_vectors')),
                          workload=str(cfg.get('workload_short') or cfg.get('workload', ''))[:200])
    line['roofline'] = _compact_roofline(full.get('roofline'), full.get('roofline_valu_issue'))
    cb = full.get('cpu_baseline')
    if isinstance(cb, dict):
        c = _pick(cb, ('value', 'unit', 'cores', 'kind'))
        c['sample'] = str(cb.get('sample_short') or cb.get('sample', ''))[:160]
        if isinstance(full.get('cpu_baseline_all_cores'), dict):
            c['all_cores'] = _pick(full['cpu_baseline_all_cores'], ('value', 'cores'))
        if isinstance(cb.get('parity_of_timed_pass'), dict):
            c['parity_of_timed_pass'] = {k: v for k, v in cb['parity_of_timed_pass'].items()
                                         if isinstance(v, (int, float, bool)) and not k.startswith('oracle_')}
        line['cpu_baseline'] = c
    else:
        line['cpu_baseline'] = None
    rk = full.get('ranks')
    if isinstance(rk, dict):
        line['ranks'] = _pick(rk, ('world_size_seen', 'backend', 'nccl_version', 'ms_per_step_by_rank',
                                   'gathered_norms_match_local_block'))
    if isinstance(full.get('host_inclusive'), dict):
        line['host_inclusive'] = _pick(full['host_inclusive'], ('value_host_inclusive',
                                                                'value_host_inclusive_with_jacobian_download'))
    if isinstance(full.get('product_default'), dict):
        line['product_default'] = _pick(full['product_default'], ('integrator', 'ms_per_step'))
    sides = {}
    for name, c in (full.get('configs') or {}).items():
        if name == 'configs3_sharded' and full.get('n_gpus') == 1:
            continue            # (at N = 1 that entry repeats configs3)
        s = _compact_side(name, c)
        if s:
            sides[name] = s
    if sides:
        line['configs'] = sides
    line['full'] = full_path
    line['build'] = full.get('build')
    line = _sig(line)
    # never longer than the limit: drop optional detail, least important first
    for drop in (('configs', 'dense'), ('configs', 'dense_stiff'), ('host_inclusive',), ('configs', 'fit'), ('configs', 'configs1'), ('build',),
                 ('configs',), ('ranks', 'ms_per_step_by_rank'), ('product_default',)):
        if len(json.dumps(line, separators=(',', ':'))) + 1 <= LINE_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict):
            d.pop(drop[-1], None)
    return line


def write_full(full, path=None):
    path = path or FULL_OUT
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = path + '.tmp.%d' % os.getpid()
        with open(tmp, 'w') as fh:
            json.dump(full, fh, indent=1)
        os.replace(tmp, path)
        return os.path.relpath(path, REPO)
    except OSError as e:
        sys.stderr.write("bench.py: could not write %s (%s); the full record goes to stderr\n" % (path, e))
        sys.stderr.write(json.dumps(full) + "\n")
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--vectors', type=int, default=V_PER_GPU, help="parameter vectors per GPU")
    ap.add_argument('--method', default='dopri45', choices=['dopri45', 'rk4', 'dop853'],
                    help="integrator of the headline pass (BASELINE's metric is quoted on the default, the RK45 pair; "
                         "'dop853' takes a seventh of the steps -- its steps/s is not comparable, its ms_per_step is)")
    ap.add_argument('--rk4-steps', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help="headline only: no other configs, no variants")
    ap.add_argument('--only', default=None, help="run one workload's GPU part alone, --steps times (profiling aid; "
                    "headline, configs1, configs3, configs4, fit, dense, dop853); no JSON contract")
    ap.add_argument('--full-out', default=None, help="where the full record goes (default gpurun_out/bench_full.json); "
                    "stdout carries ONE compact line")
    ap.add_argument('--cpu-baseline-only', action='store_true', help="time the CPU oracle and exit (no GPU needed)")
    args = ap.parse_args()
    global T_START
    T_START = time.perf_counter()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher.  This process never imports torch or
        # touches a GPU; the N ranks (one per device, backend nccl = RCCL) are children and rank 0 prints the line.
        sys.exit(self_launch(args.gpus))
    protect_stdout()
    # bench_configs imports this file as module `bench` while it runs as `__main__`: both must see ONE stamp table
    import bench as _as_module
    _as_module._CURRENT_STAMPS = _CURRENT_STAMPS
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started inside a world of %d ranks: launch with "
                         "torch.distributed.run --nproc-per-node %d (or plain `python bench.py --gpus %d`, which "
                         "spawns the ranks itself)" % (args.gpus, world, args.gpus, args.gpus))
    with_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.only is None
    cpu_all = None
    if with_cpu:
        from sysbio_modeling_amd import models_zoo as _mz
        cpu_all = cpu_baseline_all_cores(_mz.cascade_ensemble(args.vectors)[0])
    if args.cpu_baseline_only:
        from sysbio_modeling_amd.symbolic import zoo_model as _zm
        from sysbio_modeling_amd import models_zoo as _mz
        emit({"cpu_baseline": cpu_baseline(_zm('cascade20'), _mz.cascade_ensemble(args.vectors)[0]),
              "cpu_baseline_all_cores": cpu_all})
        return
    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()      # (counting devices does not initialise the GPU)
    if n_dev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # rehearsal knob (not used by the driver): SBM_BENCH_REHEARSAL=1 runs every rank on GPU 0 over gloo, so
    # that the multi-rank code path can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get('SBM_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
        os.environ['LOCAL_RANK'] = '0'
    if local_rank >= n_dev:
        raise SystemExit("bench.py --gpus %d: rank %d has no device (%d visible); one process per GPU"
                         % (args.gpus, rank, n_dev))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    from sysbio_modeling_amd import _lib
    from sysbio_modeling_amd.symbolic import zoo_model
    from sysbio_modeling_amd.model import OdeModel
    import bench_configs as bc
    gm = zoo_model('cascade20')
    model = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, use_jit=False)
    model.enable_jit(_lib.Context(local_rank))
    _CURRENT_STAMPS.update(build_stamps(gm))
    if args.only not in (None, 'headline'):
        out = bc.RUNNERS[args.only](model, gm, dev, reps=max(1, args.steps), cpu=False)
        emit({args.only: out, "build": dict(_CURRENT_STAMPS)})
        return
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
    tol = {k: model.integrator_options[k] for k in ('rtol', 'atol')}    # the product's DEFAULT tolerances
    if args.method == 'dopri45':
        opts = _lib.make_opts('dopri45', **tol)
    elif args.method == 'dop853':
        tol = dict(tol, rtol=0.1 * tol['rtol'])        # an inherited rtol is cut by ten for it, as the Python classes do
        opts = _lib.make_opts('dop853', **tol)
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
        per_rank = torch.empty((world,), dtype=f64, device=dev)
        dist.all_gather_into_tensor(per_rank, stats[:1].contiguous())
        per_rank_ms = [1e3 * float(x) / args.steps for x in per_rank.cpu()]
        dt, total_steps_per_pass, n_bad = float(mx[0]), float(sm[1]), int(sm[2])
        world_seen = dist.get_world_size()
        backend = dist.get_backend()
        norms_ok = bool(torch.equal(gathered[rank * V:(rank + 1) * V], out['norms']))
        try:
            nccl_version = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            nccl_version = None
        # the multi-GPU line is an RCCL line or no line at all: outside the one-GPU rehearsal a run that did not see
        # N ranks over nccl (= RCCL), or whose gathered norms differ from the local block, fails with a non-zero exit
        if not rehearsal and (backend != 'nccl' or world_seen != args.gpus or not norms_ok):
            raise SystemExit("bench.py --gpus %d: backend %r, %d ranks seen, gathered norms match local block: %s -- "
                             "refusing to print a multi-GPU line" % (args.gpus, backend, world_seen, norms_ok))
    else:
        total_steps_per_pass = float(steps_per_pass)
        per_rank_ms, world_seen, backend, norms_ok, nccl_version = [1e3 * dt / args.steps], 1, None, True, None
    value = total_steps_per_pass * args.steps / dt
    if args.only == 'headline':
        emit({"headline": {"ms_per_step": 1e3 * dt / args.steps, "value": value,
                           "steps_per_pass": total_steps_per_pass}, "build": dict(_CURRENT_STAMPS)})
        return

    # ---- what a user's call runs: explicit_method='auto' picks the pair by tolerance and batch (DOP853 at these defaults); the same
    # pass (gather + integrate + assemble) timed with it -- its steps are not comparable with DOPRI45's, its ms per pass is ----
    product_default = None
    if world == 1 and args.method == 'dopri45':
        try:
            pick = _lib.predict_explicit_pair(tol['rtol'], int(V), 1, 2)[0]
            o2 = _lib.make_opts(pick, rtol=(0.1 if pick == 'dop853' else 1.0) * tol['rtol'], atol=tol['atol'])

            def step2():
                _lib.check(lib.sbm_jacobian_batch(pj, p(th), V, ctypes.byref(o2), p(out['sims']), p(out['res']), p(out['J']), None,
                                                  p(out['sf']), None, p(out['norms']), None, p(out['status']), p(out['nsteps'])),
                           'sbm_jacobian_batch')
            step2()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            for _ in range(args.steps):
                step2()
            torch.cuda.synchronize(dev)
            product_default = {"integrator": pick, "ms_per_step": 1e3 * (time.perf_counter() - t2) / args.steps,
                               "accepted_steps_per_pass": int(out['nsteps'].sum().item()),
                               "failed_vectors": int((out['status'] != 0).sum().item()),
                               "note": "the pass of the headline workload as OdeModel / Project run it by default "
                                       "(explicit_method='auto'); `value` stays on DOPRI45, the pair BASELINE.json's metric names"}
            step()      # (the buffers hold the DOPRI45 pass again for the parity check below)
            torch.cuda.synchronize(dev)
        except Exception as e:   # noqa: BLE001
            product_default = {"error": repr(e)[:200]}

    # ---- SURVEY.md section 8(d)'s host-inclusive figure: P upload and download of the sampled rows / norms ----
    host_incl = None
    if world == 1:
        th_pin = torch.from_numpy(theta).pin_memory()
        res_pin = torch.empty((V, R), dtype=f64).pin_memory()
        nrm_pin = torch.empty((V,), dtype=f64).pin_memory()
        J_pin = torch.empty((V, R, q), dtype=f64).pin_memory()

        def host_step(with_J):
            th.copy_(th_pin, non_blocking=True)
            step()
            res_pin.copy_(out['res'], non_blocking=True)
            nrm_pin.copy_(out['norms'], non_blocking=True)
            if with_J:
                J_pin.copy_(out['J'], non_blocking=True)
        host_incl = {}
        for with_J, key in ((False, "value_host_inclusive"), (True, "value_host_inclusive_with_jacobian_download")):
            host_step(with_J)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                host_step(with_J)
            torch.cuda.synchronize(dev)
            host_incl[key] = total_steps_per_pass * args.steps / (time.perf_counter() - t1)
        host_incl["note"] = ("SURVEY.md section 8(d): per pass, theta (V x 40 f64) uploaded from pinned host memory "
                             "and residual rows + norms (V x 65 f64) downloaded; the second figure also downloads "
                             "the Jacobian (V x 64 x 40 f64 = 84 MB per pass); never reported as `value`")

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
    kkey = 'sens_rowgroup_cascade20_%s' % args.method
    roofline = hbm_roofline("sbm_sens_rowgroup_kernel<cascade20,%s>" % args.method, kkey, k_ms, k_steps,
                            BYTES_PER_STEP,
                            "algorithmic bytes (2*8*820 B per accepted step) / kernel time; the kernel keeps the "
                            "state in VGPRs, so real HBM traffic ('traffic', PMC) is far below this figure and the "
                            "binding resource is VALU issue: see roofline_valu_issue")
    roofline_valu = valu_roofline(kkey, k_ms, k_steps)
    roofline["achieved_fp64"] = fp64_rate(*flops_per_step(gm, args.method), k_steps / (k_ms * 1e-3))

    result = {
        "metric": "ensemble ODE-steps/sec (20-state model + fwd sens)",
        "value": value, "unit": "ODE-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: cascade20 (20 states, 40 params) + full forward sensitivities = 820 ODEs, %d "
                               "vectors/GPU, %s, 16 output times; gather + integrate + residual/Jacobian assembly"
                               % (V, {"dopri45": "DOPRI45", "dop853": "DOP853", "rk4": "RK4 fixed %d steps" % args.rk4_steps}[args.method]),
                   "description": "step = theta->p gather + integration at OdeModel's DEFAULT tolerances + fused sample / residual / "
                                  "Jacobian assembly (64 rows x 40 params, 4 scale factors)%s"
                                  % (" + RCCL all-gather of residual norms" if world > 1 else ""),
                   "vectors_per_gpu": V, "n_equations": N_AUG, "integrator": args.method,
                   "rtol": tol.get('rtol'), "atol": tol.get('atol'),
                   "accepted_steps_per_pass": total_steps_per_pass, "failed_vectors": n_bad},
        "roofline": roofline,
        "ranks": {"world_size_seen": world_seen, "backend": backend, "nccl_version": nccl_version,
                  "ms_per_step_by_rank": per_rank_ms, "gathered_norms_match_local_block": norms_ok},
    }
    if roofline_valu:
        result["roofline_valu_issue"] = roofline_valu
    if host_incl:
        result["host_inclusive"] = host_incl
    if product_default:
        result["product_default"] = product_default
    if with_cpu:
        result["cpu_baseline"] = cpu_baseline(gm, theta)
        result["cpu_baseline_all_cores"] = cpu_all
        picks = [int(x) for x in np.linspace(0, V - 1, 64).astype(int)]
        result["cpu_baseline"]["parity_of_timed_pass"] = residual_parity(
            gm, proj, theta, out['res'].cpu().numpy(), out['J'].cpu().numpy(), picks)
    elif rank == 0:
        result["cpu_baseline"] = None

    if world > 1 and not args.no_extras:
        # the configuration BASELINE names for the multi-GPU run, sharded the same way (collective: all ranks)
        try:
            c3s = bc.run_configs3_sharded(model, gm, dev, world, rank, steps=max(2, args.steps // 2), warmup=1)
        except Exception as e:   # noqa: BLE001
            c3s = {"error": repr(e)[:300]}
        result["configs"] = {"configs3_sharded": c3s}
    if not args.no_extras and rank == 0 and world == 1:
        # BASELINE.json's other configurations, each with its own roofline and CPU leg; then kernel variants
        cfgs = {}
        for name in bc.ORDER:
            try:
                cfgs[name] = bc.RUNNERS[name](model, gm, dev, reps=3, cpu=not args.no_cpu_baseline)
            except Exception as e:   # noqa: BLE001 -- a failing side config must not cost the headline line
                cfgs[name] = {"error": repr(e)[:300]}
        if isinstance(cfgs.get('configs3'), dict) and 'ms' in cfgs['configs3']:
            c3 = cfgs['configs3']
            cfgs['configs3_sharded'] = {"n_gpus": 1, "ms": c3['ms'], "steps": c3['steps'], "value": c3['value'],
                                        "unit": "ODE-steps/s", "scaling": "strong",
                                        "note": "N = 1: the configs3 figures above (the N > 1 line times the same 1024 x 8 "
                                                "workload split by vector over the ranks)"}
        result["configs"] = cfgs
        try:
            result["extras"] = bc.variant_extras(model, dev, theta_p, tg, args.rk4_steps)
        except Exception as e:   # noqa: BLE001
            result["extras"] = {"error": repr(e)[:300]}
    if rank == 0:
        result["build"] = dict(_CURRENT_STAMPS)
        result["wall_seconds"] = time.perf_counter() - T_START
        path = write_full(result, args.full_out)
        emit(compact_line(result, path))
    if world > 1:
        dist.barrier()   # rank 0 arrives late
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
