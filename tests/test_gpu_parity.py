"""Parity tests proper (need an MI355X): the HIP path, called through the C ABI, against
  * golden vectors produced by the real reference OdeModel (tests/golden/),
  * the oracle (SciPy odeint restatement) run live on the same seeded inputs,
  * the reference's own known answers (tests/test_OdeModel.py, tests/test_Project.py),
and, at BASELINE.json's full ensemble size, through size-independent properties.

Tolerance: |gpu - ref| <= 1e-8 |ref| + 5e-9  (conftest.PARITY_*; the absolute term is the
reference integrator's own noise: LSODA at atol = 1e-10 is ~1e-9 accurate in absolute terms,
measured against DOP853 at rtol 1e-13 in test_gpu_is_closer_to_tight_solution_than_lsoda)."""
import ctypes
import warnings

import numpy as np
import pytest

from tests import reference_cases as rc
from tests.conftest import parity_err, survey_err, tol_ratio, lsoda_taus, tight_taus, project_tolerances, PARITY_RTOL, PARITY_ATOL

pytestmark = pytest.mark.gpu

METHODS = [('dopri45', {}), ('rk4', {'n_steps': 65536}), ('implicit_controlled', {}), ('auto', {})]


def _from_zero(t_pts):
    """Output grid starting at the initial time 0 (odeint semantics: t_sim[0] is t0)."""
    return np.concatenate([[0.0], np.asarray(t_pts, dtype=float)])


# ---------------------------------------------------------------------------
# OdeModel.simulate / calc_jacobian
# ---------------------------------------------------------------------------
def test_reference_test_odemodel_simulate(gpu_models):
    """tests/test_OdeModel.py:20-29."""
    m = gpu_models('simple')
    assert m.n_vars == 1
    y = m.simulate(rc.SIMPLE_P, rc.SIMPLE_T10)
    assert y.shape == (10, 1)
    assert np.allclose(y[:, 0], rc.SIMPLE_Y10_GOLDEN, rtol=0.05)
    exact, _, _ = rc.simple_closed_form(rc.SIMPLE_P[0], rc.SIMPLE_P[1], rc.SIMPLE_T10)
    assert np.allclose(y[:, 0], exact, rtol=1e-9, atol=1e-12)


def test_reference_test_odemodel_calc_jacobian(gpu_models):
    """tests/test_OdeModel.py:31-52 (with the correct closed form for d/dk_deg)."""
    m = gpu_models('simple')
    s = m.calc_jacobian(rc.SIMPLE_P, rc.SIMPLE_T10, np.zeros(3))
    assert s.shape == (10, 2)
    _, d_kdeg, d_ksynt = rc.simple_closed_form(rc.SIMPLE_P[0], rc.SIMPLE_P[1], rc.SIMPLE_T10)
    assert np.allclose(s[:, 1], d_ksynt, rtol=0.05) and np.allclose(s[:, 0], d_kdeg, rtol=0.05)
    assert np.allclose(s[:, 1], d_ksynt, rtol=1e-9, atol=1e-12)
    assert np.allclose(s[:, 0], d_kdeg, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize('method,kw', METHODS)
@pytest.mark.parametrize('name,file', [('simple', 'simple_ref.npz'), ('michaelis_menten', 'mm_ref.npz')])
def test_golden_full_grid(gpu_models, golden, name, file, method, kw):
    """1000-point trajectories of the REAL reference OdeModel, both parameter sets."""
    m = gpu_models(name)
    g = golden(file)
    Y = m.simulate_batch(g['P'], g['t'], method=method, **kw)
    S, Y2 = m.calc_jacobian_batch(g['P'], g['t'], return_states=True, method=method, **kw)
    assert Y.shape == g['Y'].shape and S.shape == g['S'].shape
    assert parity_err(Y, g['Y']) <= 1.0
    assert parity_err(Y2, g['Y']) <= 1.0
    assert parity_err(S, g['S']) <= 1.0


@pytest.mark.parametrize('method,kw', METHODS)
def test_golden_cascade20(gpu_models, golden, method, kw):
    """20-state / 40-parameter network, 820 coupled ODEs: rows the reference OdeModel produced on
    its 1000-point grid at the 16 measurement times."""
    m = gpu_models('cascade20')
    g = golden('cascade20_ref.npz')
    t_out = _from_zero(g['t'][g['idx']])
    Y = m.simulate_batch(g['P'], t_out, method=method, **kw)[:, 1:]
    S, Y2 = m.calc_jacobian_batch(g['P'], t_out, return_states=True, method=method, **kw)
    assert parity_err(Y, g['Y']) <= 1.0
    assert parity_err(Y2[:, 1:], g['Y']) <= 1.0
    assert parity_err(S[:, 1:], g['S']) <= 1.0
    assert np.all(Y2[:, 0] == 0) and np.all(S[:, 0] == 0)     # initial condition row is returned untouched


@pytest.mark.parametrize('method,kw', [('dopri45', {}), ('rk4', {'n_steps': 4096})])
def test_kernel_variants_agree(gpu_models, golden, method, kw):
    """The per-wave kernel and the row-lane kernel integrate the same
    equations; both match the reference golden and differ from each other only by rounding.
    7 vectors: 7 vectors: an odd batch."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('cascade20')
    _, P = models_zoo.cascade_ensemble(7)
    g = golden('cascade20_ref.npz')
    assert np.array_equal(P[:4], g['P'])
    t_out = _from_zero(g['t'][g['idx']])
    res = {}
    for variant in ('per_wave', 'row_lane', 'row_group'):
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method=method, variant=variant, **kw)
        assert m.last_info['status'].tolist() == [0] * 7
        res[variant] = (Y, S, m.last_info['n_steps'].copy())
        if method == 'dopri45':
            assert parity_err(Y[:4, 1:], g['Y']) <= 1.0 and parity_err(S[:4, 1:], g['S']) <= 1.0
    Ya, Sa, na = res['per_wave']
    for other in ('row_lane', 'row_group'):
        Yb, Sb, nb = res[other]
        assert np.allclose(Ya, Yb, rtol=1e-9, atol=1e-11) and np.allclose(Sa, Sb, rtol=1e-9, atol=1e-10)
        assert np.all(np.abs(na - nb) <= 2)          # same controller, same step sequence up to rounding
    # failures stay per trajectory
    for variant in ('row_lane', 'row_group'):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            Pbad = P.copy()
            Pbad[2, 5] = np.nan
            S = m.calc_jacobian_batch(Pbad, t_out, method=method, variant=variant, **kw)
        assert m.last_info['status'][2] != 0 and np.all(np.isnan(S[2, -1]))
        ok = [0, 1, 3, 4, 5, 6]
        assert m.last_info['status'][ok].tolist() == [0] * 6 and np.array_equal(S[ok], res[variant][1][ok])


def test_row_group_kernel_on_two_state_model(gpu_models, golden):
    """Michaelis-Menten (2 rows, 5 columns) splits into G = 2 groups of one row each: every J_y
    term crosses a group boundary, i.e. goes through the LDS halo.  Same golden as the other
    variants, plus initial conditions for S (the s0 path of the kernel)."""
    m = gpu_models('michaelis_menten')
    g = golden('mm_ref.npz')
    for variant in ('per_wave', 'row_lane', 'row_group'):
        S, Y = m.calc_jacobian_batch(g['P'], g['t'], return_states=True, variant=variant)
        assert parity_err(Y, g['Y']) <= 1.0 and parity_err(S, g['S']) <= 1.0
    rng = np.random.default_rng(2)
    y0 = np.concatenate([[0.3, 0.1], rng.uniform(-1, 1, 10)])
    t_out = np.linspace(0, 50, 6)
    a = m.calc_jacobian_batch(g['P'], t_out, y0, variant='per_wave')
    b = m.calc_jacobian_batch(g['P'], t_out, y0, variant='row_group')
    assert np.allclose(a, b, rtol=1e-9, atol=1e-11) and np.array_equal(b[0, 0], y0[2:])


def test_live_oracle_random_vectors(gpu_models, zoo):
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd import models_zoo
    gm = zoo('cascade20')
    m = gpu_models('cascade20')
    _, P = models_zoo.cascade_ensemble(4096)
    pick = [5, 777, 4095]
    grid = np.linspace(0, 100, 1000)
    idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)
    S, Y = m.calc_jacobian_batch(P[pick], _from_zero(grid[idx]), return_states=True)
    for k, v in enumerate(pick):
        (Sr, Yr) = oo.calc_jacobian(gm, P[v], grid, use_c=True, return_states=True)
        assert parity_err(Y[k, 1:], Yr[idx]) <= 1.0
        assert parity_err(S[k, 1:], Sr[idx]) <= 1.0


def test_gpu_is_closer_to_tight_solution_than_lsoda(gpu_models, zoo):
    """'GPU more accurate than the reference' must be distinguishable from 'GPU wrong'."""
    from oracle import odeint_oracle as oo
    gm = zoo('michaelis_menten')
    m = gpu_models('michaelis_menten')
    t = np.linspace(0, 100, 50)
    tight = oo.tight_solution(gm, rc.MM_PARAMS, t)
    S, Y = m.calc_jacobian_batch(rc.MM_PARAMS[None, :], t, return_states=True, rtol=1e-11, atol=1e-14)
    Sl, Yl = oo.calc_jacobian(gm, rc.MM_PARAMS, t, return_states=True)
    e_gpu = max(np.max(np.abs(Y[0] - tight[:, :2])), np.max(np.abs(S[0] - tight[:, 2:])))
    e_lsoda = max(np.max(np.abs(Yl - tight[:, :2])), np.max(np.abs(Sl - tight[:, 2:])))
    assert e_gpu < 1e-9 and e_gpu <= e_lsoda


def test_initial_conditions_and_start_time(gpu_models, zoo):
    from oracle import odeint_oracle as oo
    gm = zoo('michaelis_menten')
    m = gpu_models('michaelis_menten')
    t = np.linspace(3.0, 40.0, 12)            # odeint: the initial condition holds at t_sim[0] = 3
    y0 = np.array([0.01, 0.002])
    Y = m.simulate(rc.MM_PARAMS, t, y0)
    assert np.array_equal(Y[0], y0)
    assert parity_err(Y, oo.simulate(gm, rc.MM_PARAMS, t, y0)) <= 1.0
    rng = np.random.default_rng(2)
    yS0 = np.concatenate([y0, 1e-3 * rng.standard_normal(10)])
    S = m.calc_jacobian(rc.MM_PARAMS, t, yS0)
    assert np.array_equal(S[0], yS0[2:])
    assert parity_err(S, oo.calc_jacobian(gm, rc.MM_PARAMS, t, yS0)) <= 1.0
    with pytest.raises(ValueError):
        m.simulate(rc.MM_PARAMS, t, np.zeros(3))
    with pytest.raises(ValueError):
        m.simulate(rc.MM_PARAMS, t[::-1])


def test_edge_grids(gpu_models):
    m = gpu_models('simple')
    # one output = the initial time only; repeated output times; empty ensemble
    y = m.simulate(rc.SIMPLE_P, np.array([0.0]))
    assert y.shape == (1, 1) and y[0, 0] == 0.0
    t = np.array([0.0, 10.0, 10.0, 10.0, 50.0])
    y = m.simulate(rc.SIMPLE_P, t)[:, 0]
    exact, _, _ = rc.simple_closed_form(rc.SIMPLE_P[0], rc.SIMPLE_P[1], t)
    assert y[1] == y[2] == y[3] and np.allclose(y, exact, rtol=1e-9, atol=1e-13)
    assert m.simulate_batch(np.zeros((0, 2)), t).shape == (0, 5, 1)
    assert m.calc_jacobian_batch(np.zeros((0, 2)), t).shape == (0, 5, 2)
    with pytest.raises(ValueError):
        m.simulate_batch(np.zeros((3, 5)), t)


def test_failures_are_reported_not_hidden(gpu_models):
    """The reference never checks LSODA's status; here every trajectory carries one and failed
    rows are NaN (the Project layer turns them into the reference's inf rows)."""
    m = gpu_models('simple')
    P = np.array([[0.001, 0.01], [np.nan, 0.01], [0.001, 0.01]])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        Y = m.simulate_batch(P, rc.SIMPLE_T10)
        assert any('failed' in str(x.message) for x in w)
    assert m.last_info['status'].tolist() == [0, 2, 0]
    assert np.all(np.isnan(Y[1, 1:])) and np.all(np.isfinite(Y[0])) and np.array_equal(Y[0], Y[2])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        S = m.calc_jacobian_batch(P[:1], rc.SIMPLE_T10, max_steps=3)
    assert m.last_info['status'].tolist() == [1] and np.all(np.isnan(S[0, -1]))


def test_c_abi_device_pointers_and_determinism(gpu_models):
    """Straight through sbm_sens_batch with caller-owned device buffers; a vector's result does
    not depend on what else is in the batch (bitwise)."""
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    m = gpu_models('cascade20')
    dm = m.device_model
    _, P = models_zoo.cascade_ensemble(256)
    t = _from_zero(models_zoo.CASCADE_MEASURE_TIMES)
    Pd, td = torch.from_numpy(P).cuda(), torch.from_numpy(t).cuda()
    V, nt = P.shape[0], len(t)
    Y = torch.full((V, nt, 20), float('nan'), dtype=torch.float64, device='cuda')
    S = torch.full((V, nt, 20, 40), float('nan'), dtype=torch.float64, device='cuda')
    st = torch.full((V,), -1, dtype=torch.int32, device='cuda')
    ns = torch.zeros_like(st)
    opts = _lib.make_opts(**{k: m.integrator_options[k] for k in ('method', 'rtol', 'atol')})   # the model's defaults
    dm.sens_dev(Pd, td, None, opts, Y, S, st, ns, None)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0 and int(ns.min()) > 50 and bool(torch.isfinite(S).all())
    S1, Y1 = m.calc_jacobian_batch(P[17:18], t, return_states=True)
    assert np.array_equal(S1[0].reshape(nt, 20, 40), S[17].cpu().numpy())
    assert np.array_equal(Y1[0], Y[17].cpu().numpy())
    # NULL arguments are refused, not dereferenced
    rc_ = dm.lib.sbm_sens_batch(dm.handle, None, V, _lib.dev_ptr(td), nt, None, ctypes.byref(opts), None, None,
                                None, None, None)
    assert rc_ != 0 and b'NULL' in dm.lib.sbm_last_error()


# ---------------------------------------------------------------------------
# full ensemble size: size-independent properties (BASELINE.json configs[1], configs[2])
# ---------------------------------------------------------------------------
def test_full_ensemble_properties(gpu_models):
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    m = gpu_models('cascade20')
    dm = m.device_model
    theta, P = models_zoo.cascade_ensemble(4096)
    t = _from_zero(models_zoo.CASCADE_MEASURE_TIMES)
    V, nt = 4096, len(t)
    Pd, td = torch.from_numpy(P).cuda(), torch.from_numpy(t).cuda()

    def run(kind, opts, Pdev=Pd):
        Y = torch.empty((V, nt, 20), dtype=torch.float64, device='cuda')
        S = torch.empty((V, nt, 20, 40), dtype=torch.float64, device='cuda') if kind == 'sens' else None
        st = torch.empty((V,), dtype=torch.int32, device='cuda')
        if kind == 'sens':
            dm.sens_dev(Pdev, td, None, opts, Y, S, st, None, None)
        else:
            dm.simulate_dev(Pdev, td, None, opts, Y, st, None, None)
        torch.cuda.synchronize()
        assert int((st != 0).sum()) == 0
        return Y, S

    dop = _lib.make_opts('dopri45', rtol=1e-9, atol=1e-12)
    rk4 = _lib.make_opts('rk4', n_steps=16384, t_end=100.0)
    Ya, Sa = run('sens', dop)
    Yb, Sb = run('sens', rk4)
    Yc, _ = run('state', dop)

    def perr(a, b):
        return float(torch.max(torch.abs(a - b) / (PARITY_ATOL + PARITY_RTOL * torch.abs(b))))
    # 1. two unrelated integrators agree on all 4096 x 16 x 820 values
    assert perr(Ya, Yb) <= 1.0 and perr(Sa, Sb) <= 1.0
    # 2. the state computed alone (one trajectory per lane) equals the state column of the augmented run
    assert perr(Yc, Ya) <= 1.0
    # 3. sensitivities are derivatives: central differences of the state w.r.t. three parameters
    for j in (0, 23, 39):
        d = 1e-5 * P[:, j]
        Pp, Pm = P.copy(), P.copy()
        Pp[:, j] += d
        Pm[:, j] -= d
        Yp, _ = run('state', dop, torch.from_numpy(Pp).cuda())
        Ym, _ = run('state', dop, torch.from_numpy(Pm).cuda())
        fd = (Yp - Ym) / torch.from_numpy(2 * d).cuda()[:, None, None]
        s = Sa[:, :, :, j]
        scale = torch.amax(torch.abs(s), dim=(1, 2), keepdim=True)
        assert float(torch.max(torch.abs(fd - s) / (scale + 1e-12))) < 1e-5


def test_single_vector_calls_take_the_small_batch_split(gpu_models, golden):
    """OdeModel.calc_jacobian and the single-vector methods of Project -- what a serial optimiser calls -- run the
    row-group kernel's small-batch split (cascade20: five chunks of eight columns, 3 equations per lane instead of 14):
    another step sequence, the same numbers to the parity tolerance; batch calls never switch by themselves, so their
    rows do not depend on the size of the batch."""
    m = gpu_models('cascade20')
    g = golden('cascade20_ref.npz')
    t_out = _from_zero(g['t'][g['idx']])
    S1 = m.calc_jacobian(g['P'][0], t_out)
    steps_single = int(m.last_info['n_steps'][0])
    Sb = m.calc_jacobian_batch(g['P'], t_out)
    steps_batch = int(m.last_info['n_steps'][0])
    assert parity_err(S1[1:], g['S'][0]) <= 1.0 and parity_err(Sb[0][1:], g['S'][0]) <= 1.0
    assert not np.array_equal(S1, Sb[0]) and steps_single != steps_batch        # two splits, two step sequences
    assert np.array_equal(m.calc_jacobian_batch(g['P'][:1], t_out)[0], Sb[0])    # a batch of one is still a batch
    assert np.array_equal(m.calc_jacobian_batch(g['P'][:1], t_out, variant='small_batch')[0], S1)
    # beyond 1024 wavefronts the request falls back to the throughput split
    from sysbio_modeling_amd import models_zoo
    _, P = models_zoo.cascade_ensemble(512)
    assert np.array_equal(m.calc_jacobian_batch(P, t_out, variant='small_batch'), m.calc_jacobian_batch(P, t_out))


@pytest.mark.parametrize('name,n_traj', [('cascade20', 4096), ('cascade20', 2049), ('michaelis_menten', 3003)])
def test_packed_state_kernel_equals_the_unpacked_one(gpu_models, zoo, name, n_traj):
    """From 2048 trajectories on the state-only path packs several trajectories into one wavefront (two segments of
    32 lanes for the 20-state model, four of 16 for the 2-state one), each with its own time, step size and
    accept / reject decisions.  The segment reductions return the same bits in every lane, so the numbers are
    those of the one-trajectory-per-wavefront kernel (selected with a kernel variant) bit for bit -- also for a
    trajectory count that leaves the last wavefront partly empty, ragged step counts, and the fixed-step method."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models(name)
    rng = np.random.default_rng(17)
    if name == 'cascade20':
        _, P = models_zoo.cascade_ensemble(n_traj)
        t = _from_zero(models_zoo.CASCADE_MEASURE_TIMES)
    else:
        P = np.array([1e-3, 1e-3, 0.01, 0.01, 1e-3]) * np.exp(0.8 * rng.standard_normal((n_traj, 5)))
        t = np.linspace(0, 100, 7)
    for method, kw in (('dopri45', {}), ('rk4', {'n_steps': 512})):
        Ya = m.simulate_batch(P, t, method=method, **kw)
        na, sa = m.last_info['n_steps'].copy(), m.last_info['status'].copy()
        Yb = m.simulate_batch(P, t, method=method, variant='row_lane', **kw)
        assert sa.max() == 0 and np.array_equal(na, m.last_info['n_steps'])
        assert np.array_equal(Ya, Yb)
    assert na.min() == na.max() or method == 'dopri45'
    # a batch below the threshold takes the unpacked kernel either way: same numbers for the same vectors
    Ys = m.simulate_batch(P[:100], t)
    assert np.array_equal(Ys, m.simulate_batch(P, t)[:100])


# ---------------------------------------------------------------------------
# Project
# ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def simple_case(gpu_models, zoo):
    from oracle.project_oracle import ProjectOracle
    from sysbio_modeling_amd.project import Project
    exps, settings, mapping, sf = rc.simple_project_case()
    proj = Project(gpu_models('simple'), exps, settings, mapping, sf_groups=sf)
    exps2, _, _, _ = rc.simple_project_case()
    po = ProjectOracle(zoo('simple'), exps2, settings, mapping, sf_groups=sf)
    theta = rc.simple_project_theta(proj.get_param_index)
    return proj, po, theta


def test_reference_test_sim_experiments(simple_case):
    """tests/test_Project.py:96-117."""
    proj, po, theta = simple_case
    res = proj.residuals(theta)
    assert res.shape == (35,)
    sims = proj.get_simulations()
    assert len(set(sims.index.get_level_values(0))) == 2
    for exp in proj.experiments:
        data, _, _ = exp.get_variable_measurements('Variable_1').get_nonzero_measurements()
        sim = sims.loc[(exp.name, 'Variable_1'), 'mean'].values
        assert np.allclose(data / 3.75, sim, rtol=0.05)
    assert np.allclose(proj.scale_factors['Variable_1'].sf, 3.75, rtol=0.05)
    ref, ref_sims, B = po.residuals(theta, return_parts=True)
    assert parity_err(res, ref) <= 1.0 and parity_err(sims['mean'].values, ref_sims) <= 1.0
    assert abs(proj.scale_factors['Variable_1'].sf - B[0]) <= 1e-8 * B[0]
    scaled = proj.get_simulations(scaled=True)['mean'].values
    assert parity_err(scaled, ref_sims * B[0]) <= 1.0
    # sampled grid times, not measurement times (project/utils.py:21)
    assert sims['timepoints'].values[0] == pytest.approx(np.linspace(0, HIGH_T_END, 1000)[
        np.searchsorted(np.linspace(0, HIGH_T_END, 1000), rc.HIGH_DEG_T[0])])


HIGH_T_END = rc.HIGH_DEG_T[-1]


def test_reference_test_variable_jacobian(simple_case):
    """tests/test_Project.py:119-143."""
    proj, po, theta = simple_case
    proj.calc_project_jacobian(theta)
    model_jac = proj.get_model_jacobian_df()
    p = np.exp(theta)
    for exp in proj.experiments:
        ks, kd = exp.param_global_vector_idx['k_synt'], exp.param_global_vector_idx['k_deg']
        t = exp.get_variable_measurements('Variable_1').timepoints
        t = t[t != 0]
        anal = rc.simple_model_analytical_jac(p[kd], p[ks], t)
        blk = model_jac.loc[(exp.name, 'Variable_1'), :].values
        assert np.allclose(anal[1], blk[:, ks], rtol=0.05) and np.allclose(anal[0], blk[:, kd], rtol=0.05)
    assert parity_err(model_jac.values, po.model_jacobian(theta)) <= 1.0


def test_reference_test_calc_project_jacobian(simple_case):
    """tests/test_Project.py:145-176: FD identities, then parity with the oracle."""
    proj, po, theta = simple_case
    J = proj.calc_project_jacobian(theta)
    assert J.shape == (35, 3)

    def scaled_sims(x):
        proj.residuals(x)
        return proj.get_simulations(scaled=True)['mean'].values
    assert np.allclose(rc.central_fd_jacobian(scaled_sims, theta), J, atol=1e-6)
    th2 = theta.copy()
    th2[proj.get_param_index('Group_1', ('High',))] = np.log(0.02)
    th2[proj.get_param_index('Group_1', ('Low',))] = np.log(0.003)
    th2[proj.get_param_index('k_synt', 'Global')] = np.log(0.05)
    g = proj.calc_rss_gradient(th2)
    gnum = rc.central_fd_jacobian(lambda x: np.array([proj.calc_sum_square_residuals(x)]), th2)[0]
    assert np.allclose(g, gnum, atol=1e-6)
    assert parity_err(J, po.calc_project_jacobian(theta)) <= 1.0
    assert np.allclose(g, po.calc_rss_gradient(th2), rtol=1e-7, atol=1e-9)
    grad = np.zeros(3)
    val = proj.nlopt_fcn(th2, grad)                                   # :829-852
    assert np.allclose(grad, g) and val == pytest.approx(po.calc_sum_square_residuals(th2), rel=1e-8)
    sfg = proj.scale_factors['Variable_1'].gradient
    assert sfg.shape == (3,)


def test_reference_test_optimization(simple_case):
    """tests/test_Project.py:202-213: leastsq driven by residuals + Dfun runs and fits."""
    from scipy.optimize import leastsq
    proj, _, theta = simple_case
    base_guess = np.log(np.ones(3) * 0.01)
    out, ier = leastsq(proj.residuals, base_guess, Dfun=proj.calc_project_jacobian)
    assert ier in (1, 2, 3, 4)
    assert proj.calc_sum_square_residuals(out) <= proj.calc_sum_square_residuals(base_guess)


def test_reference_test_sum_variables(gpu_models, zoo):
    """tests/test_Project.py:263-349: Michaelis-Menten, 'sum' of both species, no scale factors."""
    from oracle import odeint_oracle as oo
    from oracle.project_oracle import ProjectOracle
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    gm = zoo('michaelis_menten')
    sim = oo.simulate(gm, rc.MM_PARAMS, rc.MM_T)
    mk = lambda: Experiment('Standard', TimecourseMeasurement('Total', sim.sum(axis=1), rc.MM_T))
    with pytest.warns(UserWarning):
        proj = Project(gpu_models('michaelis_menten'), [mk()], {}, {'Total': ('sum', [0, 1])})
    po = ProjectOracle(gm, [mk()], {}, {'Total': ('sum', [0, 1])})
    for param in proj.project_param_idx:
        assert list(proj.project_param_idx[param].keys()) == ['Global']
    pdict = {n: {'Global': v} for n, v in zip(gm.param_order, rc.MM_PARAMS)}
    theta = np.log(proj.project_param_dict_to_vect(pdict))
    res = proj.residuals(theta)
    assert np.allclose(res, 0, atol=1e-3)
    g = proj.calc_rss_gradient(theta)
    gnum = rc.central_fd_jacobian(lambda x: np.array([proj.calc_sum_square_residuals(x)]), theta)[0]
    assert np.allclose(g, gnum, atol=1e-5)
    J = proj.calc_project_jacobian(theta)

    def all_sims(x):
        proj.residuals(x)
        return proj.get_simulations()['mean'].values
    assert np.allclose(J, rc.central_fd_jacobian(all_sims, theta), atol=1e-5)
    assert parity_err(J, po.calc_project_jacobian(theta)) <= 1.0
    assert parity_err(all_sims(theta), po.residuals(theta, return_parts=True)[1]) <= 1.0


def _cascade_project(gpu_models, zoo, n_exp=3, compat=True, fixed=False, priors=False):
    """A small version of BASELINE.json configs[3]: shared d0..d3 by condition, several measures per
    experiment, one scale-factor group per measured species (plus a two-measure group)."""
    from oracle.project_oracle import ProjectOracle
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    gm = zoo('cascade20')
    rng = np.random.default_rng(7)
    p_nom = models_zoo.cascade_nominal_params()
    names = gm.param_order
    settings = {'Shared': {'deg': {('d%d' % i): ('cond',) for i in range(4)}},
                'Global': [n for n in names if n not in ('d0', 'd1', 'd2', 'd3')]}
    mapping = {'s4': ('direct', 4), 's9': ('direct', 9), 's14': ('direct', 14), 's19': ('direct', 19),
               'tot': ('sum', [0, 1, 2])}
    sf = ['s4', frozenset(['s9', 's14']), 's19']

    def build():
        exps = []
        for c in range(n_exp):
            p = p_nom.copy()
            p[20:24] *= (1.0 + 0.25 * c)
            tmax = 100.0 - 10.0 * c                                   # ragged grids: t_end differs per experiment
            tm = np.linspace(tmax / 8, tmax, 8)
            grid = np.linspace(0, tmax, 1000)
            y = oo.simulate(gm, p, grid, use_c=True)[np.searchsorted(grid, tm)]
            ms = []
            for nm, (kind, arg) in mapping.items():
                val = y[:, arg] if kind == 'direct' else y[:, arg].sum(axis=1)
                val = val * (2.0 if nm != 'tot' else 1.0) * (1 + 0.05 * rng.standard_normal(8))
                ms.append(TimecourseMeasurement(nm, val, tm.copy(), 0.05 * np.abs(val) + 0.01))
            fx = {'k3': 1.1} if (fixed and c == 1) else None
            exps.append(Experiment('exp_%d' % c, ms, fixed_parameters=fx, experiment_settings={'cond': c}))
        return exps
    rng_state = rng.bit_generator.state
    exps_a = build()
    rng.bit_generator.state = rng_state
    exps_b = build()
    proj = Project(gpu_models('cascade20'), exps_a, settings, mapping, sf_groups=sf, reference_compat=compat)
    po = ProjectOracle(gm, exps_b, settings, mapping, sf_groups=sf, reference_compat=compat)
    if priors:
        for obj in (proj, po):
            obj.set_parameter_log_prior('k5', 'Global', np.log(1.2), 0.5)
            obj.set_parameter_log_prior('deg', (1,), np.log(0.1), 0.3)
            obj.set_scale_factor_log_prior('s19', np.log(2.0), 0.2)
    return proj, po


@pytest.mark.parametrize('compat,fixed,priors', [(True, False, False), (False, True, True), (True, True, True)])
def test_project_cascade_vs_oracle(gpu_models, zoo, compat, fixed, priors):
    proj, po = _cascade_project(gpu_models, zoo, compat=compat, fixed=fixed, priors=priors)
    # the four parameters of group 'deg' share ONE slot per condition (reference semantics,
    # base_project.py:252-264): their Jacobian contributions accumulate in that column
    assert proj.n_project_params == po.n_project_params == 36 + 3
    assert proj.project_param_idx == po.project_param_idx
    rng = np.random.default_rng(99)
    thetas = 0.1 * rng.standard_normal((3, proj.n_project_params))
    from sysbio_modeling_amd import models_zoo
    p_nom = models_zoo.cascade_nominal_params()
    for name, slots in proj.project_param_idx.items():
        for key, gi in slots.items():
            base = p_nom[proj._model.param_order.index(name)] if name != 'deg' else 0.1
            thetas[:, gi] += np.log(base)
    out = proj.evaluate_batch(thetas, jacobian=True,
                              want=('jacobian', 'model_jacobian', 'gradient', 'sf_gradient'))
    assert out['status'].tolist() == [0, 0, 0]
    a = proj.descriptor_arrays()
    tols = []
    for v in range(3):
        ref, sims, B = po.residuals(thetas[v], return_parts=True)
        Jref = po.calc_project_jacobian(thetas[v])
        Jm = po.model_jacobian(thetas[v])
        # trajectories agree with the reference's LSODA to 1e-8 |ref| + 5e-9; everything downstream gets the
        # first-order propagation of exactly that through the reference's formulas (oracle/tolerances.py)
        tau_s, tau_Jm = lsoda_taus(a, thetas[v], sims, Jm)
        t = project_tolerances(a, sims, B, tau_s, Jm, tau_Jm)
        tols.append(t)
        assert tol_ratio(out['sims'][v], sims, t['sims']) <= 1.0
        assert tol_ratio(out['sf'][v], B, t['sf']) <= 1.0
        assert tol_ratio(out['residuals'][v], ref, t['residuals']) <= 1.0
        assert out['norms'][v] == pytest.approx(np.sum(ref ** 2), abs=2.0 * np.sum(np.abs(ref) * t['residuals']))
        assert tol_ratio(out['jacobian'][v], Jref, t['jacobian']) <= 1.0
        assert tol_ratio(out['model_jacobian'][v], Jm, t['model_jacobian']) <= 1.0
        g_tol = (np.abs(Jref) * t['residuals'][:, None] + np.abs(ref)[:, None] * t['jacobian']).sum(axis=0)
        assert tol_ratio(out['gradient'][v], (Jref.T * ref).sum(axis=1), g_tol) <= 1.0
    # V = 1 through the reference-named methods: residuals() (state-only kernel) gives bit-identical numbers to the
    # batch; calc_project_jacobian() runs the augmented kernel's small-batch split (another step sequence) and agrees
    # to the integration tolerance -- and bit for bit when the project's options pin the variant
    res_only = proj.evaluate_batch(thetas)
    assert np.array_equal(proj.residuals(thetas[1]), res_only['residuals'][1])
    J1 = proj.calc_project_jacobian(thetas[1])
    assert tol_ratio(J1, out['jacobian'][1], tols[1]['jacobian']) <= 1.0
    proj.integrator_options['variant'] = 'auto'
    assert np.array_equal(proj.calc_project_jacobian(thetas[1]), out['jacobian'][1])
    del proj.integrator_options['variant']
    # the two kernels agree with each other to the parity tolerance
    for v in range(3):
        assert tol_ratio(res_only['residuals'][v], out['residuals'][v], tols[v]['residuals']) <= 1.0


def test_project_failed_vector_gives_inf_rows(gpu_models, zoo):
    """NaN simulations -> inf residuals and inf Jacobian (squared_loss_function.py:28-32,46-50)."""
    proj, _ = _cascade_project(gpu_models, zoo, n_exp=2)
    th = np.zeros((2, proj.n_project_params))
    th[1, 3] = np.nan
    out = proj.evaluate_batch(th, jacobian=True, want=('jacobian',))
    assert out['status'][0] == 0 and out['status'][1] != 0
    assert np.all(np.isinf(out['residuals'][1])) and np.all(np.isinf(out['jacobian'][1]))
    assert np.all(np.isfinite(out['residuals'][0])) and np.all(np.isfinite(out['jacobian'][0]))
    with pytest.warns(UserWarning, match="integration failed"):
        r = proj.residuals(th[1])
    assert np.all(np.isinf(r))


# ---------------------------------------------------------------------------
# BASELINE.json configs[3] at full size: 8 experiment settings x 1024 vectors
# ---------------------------------------------------------------------------
def test_config4_full_size(gpu_models, zoo):
    """R = 512 rows, q = 68 parameters, 8192 trajectories of 820 equations in one launch.  Checked by
    properties that do not need the oracle at this size (norms, gradient identity, bitwise batch
    independence, directional finite differences) and against the oracle on three of the vectors."""
    import torch
    from oracle import odeint_oracle as oo
    from oracle.project_oracle import ProjectOracle
    from sysbio_modeling_amd import models_zoo
    gm = zoo('cascade20')
    model = gpu_models('cascade20')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(model)
    assert proj.n_project_params == 68 and proj.n_project_residuals == 512
    thetas = models_zoo.config4_ensemble(th0, 1024)
    out = proj.evaluate_batch(torch.from_numpy(thetas).cuda(), jacobian=True, want=('jacobian', 'gradient'))
    torch.cuda.synchronize()
    R, J, g = out['residuals'], out['jacobian'], out['gradient']
    assert R.shape == (1024, 512) and J.shape == (1024, 512, 68)
    assert int((out['status'] != 0).sum()) == 0 and bool(torch.isfinite(J).all())
    assert torch.allclose(out['norms'], (R ** 2).sum(dim=1), rtol=1e-12)
    assert torch.allclose(g, torch.einsum('vrq,vr->vq', J, R), rtol=1e-10, atol=1e-9)
    # a vector's rows do not depend on its batch mates (bitwise)
    sub = proj.evaluate_batch(torch.from_numpy(thetas[500:503]).cuda(), jacobian=True, want=('jacobian',))
    assert torch.equal(sub['residuals'], R[500:503]) and torch.equal(sub['jacobian'], J[500:503])
    # J is d(B*sim)/dtheta: directional central differences of the SCALED simulations (state-only kernel)
    rng = np.random.default_rng(0)
    u = rng.standard_normal(68)
    u /= np.linalg.norm(u)
    eps = 1e-5
    pick = slice(0, 64)

    def scaled(th):
        o = proj.evaluate_batch(torch.from_numpy(th).cuda())
        grp = torch.from_numpy(proj._rows['sf'].astype(np.int64)).cuda()
        return o['sims'] * o['sf'][:, grp]
    fd = (scaled(thetas[pick] + eps * u) - scaled(thetas[pick] - eps * u)) / (2 * eps)
    jv = torch.einsum('vrq,q->vr', J[pick], torch.from_numpy(u).cuda())
    assert float(torch.max(torch.abs(fd - jv)) / torch.max(torch.abs(jv))) < 1e-6
    # three vectors against the oracle (same experiments rebuilt from SciPy-generated data would differ in
    # the noise draw: reuse the project's own experiments)
    exps = [proj.get_experiment(i) for i in range(8)]
    po = ProjectOracle(gm, exps, proj._model_parameter_settings, {k: (v['type'], v['variables'][0])
                       for k, v in proj._measurement_to_model_map.items()},
                       sf_groups=['s%d' % v for v in models_zoo.CASCADE_MEASURED_SPECIES])
    assert po.project_param_idx == proj.project_param_idx
    a = proj.descriptor_arrays()
    for v in (0, 511, 1023):
        ref, sims, B = po.residuals(thetas[v], return_parts=True)
        Jref = po.calc_project_jacobian(thetas[v])
        Jm = po.model_jacobian(thetas[v])
        tau_s, tau_Jm = lsoda_taus(a, thetas[v], sims, Jm)
        t = project_tolerances(a, sims, B, tau_s, Jm, tau_Jm)
        assert tol_ratio(R[v].cpu().numpy(), ref, t['residuals']) <= 1.0
        assert tol_ratio(J[v].cpu().numpy(), Jref, t['jacobian']) <= 1.0


# ---------------------------------------------------------------------------
# next row f3: log-square and normalized losses
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('loss', ['log', 'normalized'])
@pytest.mark.parametrize('compat', [True, False])
def test_project_other_losses_vs_oracle(gpu_models, zoo, loss, compat):
    from oracle import odeint_oracle as oo
    from oracle.project_oracle import ProjectOracle
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    from sysbio_modeling_amd.project.loss_functions import LogSquareLossFunction, NormalizedSquareLossFunction
    gm = zoo('michaelis_menten')
    p_true = np.array([0.05, 0.3, 0.4, 0.05, 0.02])
    t = np.linspace(0, 60, 13)
    grid = np.linspace(0, 60, 1000)
    y = oo.simulate(gm, p_true, grid)[np.searchsorted(grid, t)]
    rng = np.random.default_rng(12)

    def exps():
        r = np.random.default_rng(12)
        a = TimecourseMeasurement('S', 3.0 * y[:, 0] * (1 + 0.05 * r.standard_normal(13)) + 1e-3, t.copy(),
                                  0.1 * np.abs(y[:, 0]) + 0.05)
        b = TimecourseMeasurement('P', 0.5 * y[:, 1] * (1 + 0.05 * r.standard_normal(13)) + 1e-3, t.copy(),
                                  0.1 * np.abs(y[:, 1]) + 0.05)
        return [Experiment('e1', [a, b]), Experiment('e2', [TimecourseMeasurement('Tot', y.sum(axis=1) + 0.01, t.copy())])]
    mapping = {'S': ('direct', 0), 'P': ('direct', 1), 'Tot': ('sum', [0, 1])}
    cls = LogSquareLossFunction if loss == 'log' else NormalizedSquareLossFunction
    with pytest.warns(UserWarning):
        proj = Project(gpu_models('michaelis_menten'), exps(), {}, mapping, sf_groups=['S', 'P'], loss_function=cls,
                       reference_compat=compat)
    po = ProjectOracle(gm, exps(), {}, mapping, sf_groups=['S', 'P'], reference_compat=compat, loss=loss)
    for obj in (proj, po):
        obj.set_scale_factor_log_prior('S', np.log(2.5), 0.3)
        obj.set_parameter_log_prior('km', 'Global', np.log(0.2), 1.0)
    thetas = np.log(p_true)[None, :] + 0.2 * rng.standard_normal((4, 5))
    out = proj.evaluate_batch(thetas, jacobian=True, want=('jacobian', 'gradient', 'sf_gradient'))
    assert out['status'].tolist() == [0] * 4
    a = proj.descriptor_arrays()
    for v in range(4):
        ref, sims, B = po.residuals(thetas[v], return_parts=True)
        Jref = po.calc_project_jacobian(thetas[v])
        Jm = po.model_jacobian(thetas[v])
        tau_s, tau_Jm = lsoda_taus(a, thetas[v], sims, Jm)
        t = project_tolerances(a, sims, B, tau_s, Jm, tau_Jm)      # the loss type comes with the descriptor arrays
        assert tol_ratio(out['sf'][v], B, t['sf']) <= 1.0
        assert tol_ratio(out['residuals'][v], ref, t['residuals']) <= 1.0
        assert tol_ratio(out['jacobian'][v], Jref, t['jacobian']) <= 1.0
    if not compat:
        # with J divided by sigma the gradient is the true derivative of 0.5*sum r^2
        g = rc.central_fd_jacobian(lambda x: np.array([proj.calc_sum_square_residuals(x)]), thetas[0])[0]
        assert np.allclose(out['gradient'][0], g, rtol=1e-5, atol=1e-6 * np.max(np.abs(g)))
    # the reference refuses non-positive data under the log loss (log_squared_loss_function.py:55-56)
    if loss == 'log':
        bad = [Experiment('e', [TimecourseMeasurement('S', np.array([0.0, 1.0, 2.0]), np.array([1.0, 2.0, 3.0]))])]
        with pytest.warns(UserWarning):
            pb = Project(gpu_models('michaelis_menten'), bad, {}, {'S': ('direct', 0)}, loss_function=cls)
        with pytest.raises(ValueError, match="smaller or equal to zero"):
            pb.residuals(np.log(p_true))


def test_launch_order_does_not_change_results(gpu_models):
    """From 2048 trajectories on, the library launches the sensitivity kernel longest-trajectory-first,
    using the step counts of the previous launch of the same size (sbm_core.hip::launch_sens).  The
    first call runs in index order, the second sorted, the third sorted by the second's counts on
    DIFFERENT parameters: every output must be bitwise what index order gives."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('cascade20')
    _, P = models_zoo.cascade_ensemble(2048)
    t_out = _from_zero(np.linspace(6.25, 100.0, 4))
    S1, Y1 = m.calc_jacobian_batch(P, t_out, return_states=True)
    n1 = m.last_info['n_steps'].copy()
    S2, Y2 = m.calc_jacobian_batch(P, t_out, return_states=True)
    assert np.array_equal(S1, S2) and np.array_equal(Y1, Y2) and np.array_equal(n1, m.last_info['n_steps'])
    assert n1.min() < n1.max()                      # there is something to sort
    P3 = P[::-1].copy()                             # stale order: worst case for the predictor
    S3 = m.calc_jacobian_batch(P3, t_out)
    assert np.array_equal(S3, S1[::-1])
    S4 = m.calc_jacobian_batch(P3[:100], t_out)     # small batches bypass the ordering
    assert np.array_equal(S4, S3[:100])


def test_free_energy_and_scale_factor_entropy(gpu_models, zoo):
    """Project.free_energy = rss - T * sum_groups log-integral (reference base_project.py:854-892,
    linear_scale_factor.py:13-18,63-81), against an independent evaluation: oracle simulations, the
    integral by the trapezoid rule on a dense u grid."""
    from oracle.project_oracle import ProjectOracle
    from sysbio_modeling_amd.project import Project
    exps, settings, mapping, sf = rc.simple_project_case()
    proj = Project(gpu_models('simple'), exps, settings, mapping, sf_groups=sf)
    exps2, _, _, _ = rc.simple_project_case()
    po = ProjectOracle(zoo('simple'), exps2, settings, mapping, sf_groups=sf)
    theta = rc.simple_project_theta(proj.get_param_index) + 0.05
    with pytest.raises(ValueError):
        proj.free_energy(theta)                       # no log prior on the scale factor
    proj.set_scale_factor_log_prior('Variable_1', np.log(3.0), 0.5)
    po.set_scale_factor_log_prior('Variable_1', np.log(3.0), 0.5)
    for T in (1.0, 2.5):
        fe = proj.free_energy(theta, T)
        res, sims, B = po.residuals(theta, return_parts=True)
        rows = po.rows()
        d = np.array([r[2] for r in rows]); sg = np.array([r[3] for r in rows])
        ak, bk = np.sum(sims ** 2 / sg ** 2), np.sum(sims * d / sg ** 2)
        u = np.linspace(-6, 6, 400001)
        b = np.exp(u) * bk / ak
        f = np.exp(-ak / (2 * T) * (b - bk / ak) ** 2 - (u + np.log(bk / ak) - np.log(3.0)) ** 2 / (2 * 0.25))
        integral = np.sum(0.5 * (f[1:] + f[:-1])) * (u[1] - u[0])
        ref = 0.5 * np.sum(res ** 2) - T * np.log(integral)
        assert fe == pytest.approx(ref, rel=1e-7)
    fb = proj.free_energy_batch(np.stack([theta, theta + 0.1]), 1.0)
    assert fb[0] == pytest.approx(proj.free_energy(theta, 1.0), rel=1e-12) and np.isfinite(fb[1])


def test_c_abi_allgather_norms_single_rank():
    """sbm_allgather_norms with a communicator the HOST creates through RCCL's own C API (as a non-Python
    host would): one rank here -- the gather of 1 rank is a copy, but the call goes through RCCL on the
    context's stream.  The multi-rank sharding logic is covered on CPU (tests/test_distributed_cpu.py)."""
    import ctypes
    import torch
    from sysbio_modeling_amd import _lib
    torch.zeros(1, device='cuda')
    rccl = ctypes.CDLL('librccl.so.1')

    class UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        ctx = _lib.default_context()
        send = torch.arange(1000, dtype=torch.float64, device='cuda') * 0.5
        recv = torch.full((1000,), -1.0, dtype=torch.float64, device='cuda')
        _lib.check(ctx.lib.sbm_allgather_norms(ctx.handle, comm, _lib.dev_ptr(send), 1000, _lib.dev_ptr(recv)),
                   'sbm_allgather_norms')
        ctx.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_large_batch_indexing(gpu_models):
    """16 384 vectors x 17 rows x 820 values = 1.8 GB of sensitivities in one call (past 2^31 bytes and
    2^28 elements): every trajectory must equal what a small batch gives for the same parameters."""
    import torch
    from sysbio_modeling_amd import models_zoo, _lib
    m = gpu_models('cascade20')
    V = 16384
    _, P = models_zoo.cascade_ensemble(V)
    grid = np.linspace(0, 100, 1000)
    t = np.concatenate([[0.0], grid[np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)]])
    dm = m.device_model
    Pd, td = torch.from_numpy(P).cuda(), torch.from_numpy(t).cuda()
    Y = torch.empty((V, 17, 20), dtype=torch.float64, device='cuda')
    S = torch.full((V, 17, 20, 40), float('nan'), dtype=torch.float64, device='cuda')
    st = torch.empty(V, dtype=torch.int32, device='cuda')
    ns = torch.empty(V, dtype=torch.int32, device='cuda')
    dm.sens_dev(Pd, td, None, _lib.make_opts(t0=0.0, **{k: m.integrator_options[k] for k in ('method', 'rtol', 'atol')}),
                Y, S, st, ns, None)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0 and bool(torch.isfinite(S).all())
    pick = [0, 1, 4095, 4096, 8191, 12345, V - 1]
    S_small, Y_small = m.calc_jacobian_batch(P[pick], t, return_states=True)
    assert np.array_equal(S[pick].cpu().numpy().reshape(len(pick), 17, 800), S_small)
    assert np.array_equal(Y[pick].cpu().numpy(), Y_small)
    del S, Y
    torch.cuda.empty_cache()


def test_step_budget_with_early_exit_stays_quiet_on_slow_starting_runs(gpu_models):
    """The default options carry a step budget with an early exit (max_steps = -50000: a trajectory whose current step
    size could not finish within four budgets gives up at once -- an explicit method on a stiff system).  A run that is
    NOT stiff but starts with small steps on a long horizon must not be mistaken for one: the cascade's transient (t < 100)
    takes ~1000 steps of 0.1, at which rate a horizon of 3 10^4 or 10^5 looks like 3 10^5 / 10^6 steps -- but the step size
    grows to ~3 afterwards and 5 000 / 15 000 steps do.  During the transient the controller never rejects, which is what
    tells it from a step size pinned by stability.  Same numbers as with a plain budget; a horizon the budget really
    cannot cover (10^6: 100 000+ steps) and a genuinely exhausted plain budget are still reported."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('cascade20')
    _, P = models_zoo.cascade_ensemble(8)
    for t_end in (3.0e4, 1.0e5):
        t_out = np.array([0.0, 1.0e2, t_end])
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True)                      # default: budget with early exit
        info = {k: np.array(v, copy=True) for k, v in m.last_info.items() if k in ('status', 'n_steps')}
        assert not info['status'].any() and info['n_steps'].max() < 50000
        S2, Y2 = m.calc_jacobian_batch(P, t_out, return_states=True, max_steps=1000000)   # plain budget
        assert np.array_equal(S, S2) and np.array_equal(Y, Y2) and np.array_equal(info['n_steps'], m.last_info['n_steps'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.simulate_batch(P[:2], np.array([0.0, 1.0e6]))
        assert m.last_info['status'].tolist() == [1, 1]
        m.simulate_batch(P[:2], np.array([0.0, 1.0e2]), max_steps=100)
        assert m.last_info['status'].tolist() == [1, 1]
    # ... and method='auto' hands the long horizon to the implicit integrator
    Ya = m.simulate_batch(P[:2], np.array([0.0, 1.0e6]), method='auto')
    assert m.last_info['status'].tolist() == [0, 0] and m.last_info['stiff'].tolist() == [True, True]
    Yl = m.simulate_batch(P[:2], np.array([0.0, 1.0e6]), max_steps=1000000)
    assert parity_err(Ya[:, 1:], Yl[:, 1:]) <= 1.0
