"""Batched multi-start Levenberg-Marquardt against the reference's way of
fitting: scipy.optimize.leastsq(project.residuals, x0, Dfun=project.calc_project_jacobian)
(tests/test_Project.py:202-213, :352-357), which the same Project object also supports (batch of one)."""
import warnings

import numpy as np
import pytest
from scipy.optimize import leastsq


pytestmark = pytest.mark.gpu


def test_lm_step_solves_damped_normal_equations():
    """sbm_lm_step against numpy on random systems, including a rank-deficient and a non-finite one."""
    import torch
    from sysbio_modeling_amd import _lib
    rng = np.random.default_rng(0)
    V, M, q = 7, 45, 11
    J = rng.standard_normal((V, M, q))
    r = rng.standard_normal((V, M))
    lam = 10.0 ** rng.uniform(-6, 2, V)
    J[3, :, 4] = 0.0                 # a parameter no residual depends on: delta stays 0 there
    J[5, 2, 1] = np.nan              # unusable input
    Jd, rd, ld = (torch.from_numpy(x).cuda() for x in (J, r, lam))
    delta = torch.empty((V, q), dtype=torch.float64, device='cuda')
    pred = torch.empty((V,), dtype=torch.float64, device='cuda')
    st = torch.empty((V,), dtype=torch.int32, device='cuda')
    ctx = _lib.default_context()
    p = _lib.dev_ptr
    _lib.check(ctx.lib.sbm_lm_step(ctx.handle, p(Jd), p(rd), p(ld), V, M, q, p(delta), p(pred), p(st)), 'sbm_lm_step')
    torch.cuda.synchronize()
    delta, pred, st = delta.cpu().numpy(), pred.cpu().numpy(), st.cpu().numpy()
    assert st.tolist() == [0, 0, 0, 0, 0, 1, 0] and np.all(delta[5] == 0)
    for v in (0, 1, 2, 3, 4, 6):
        A = J[v].T @ J[v]
        g = J[v].T @ r[v]
        D = np.diag(np.where(np.diag(A) > 0, np.diag(A) * (1 + lam[v]), 1.0))
        ref = np.linalg.solve(A - np.diag(np.diag(A)) + D, -g)
        assert np.allclose(delta[v], ref, rtol=1e-9, atol=1e-12)
        assert pred[v] == pytest.approx(-(g @ ref) - 0.5 * ref @ A @ ref, rel=1e-9)
    assert delta[3, 4] == 0.0


def test_lm_trust_step_finds_the_levenberg_marquardt_parameter():
    """sbm_lm_trust_step (MINPACK's lmpar on the device): on random systems the step solves
    (J^T J + lambda D^2) delta = -J^T r with ||D delta|| within 10 % of the radius -- or lambda = 0 and the Gauss-Newton step
    inside it -- D is the running maximum of the column norms, and the predicted decrease is that of the Gauss-Newton
    model; a rank-deficient J is handled by the damping, non-finite input is reported."""
    import torch
    from sysbio_modeling_amd import _lib
    rng = np.random.default_rng(3)
    V, M, q = 9, 40, 12
    J = rng.standard_normal((V, M, q)) * 10.0 ** rng.uniform(-3, 1, (V, 1, q))
    r = rng.standard_normal((V, M))
    J[2, :, 5] = 0.0                                   # a column nothing depends on
    J[4, :, 7] = J[4, :, 3]                            # rank deficient
    J[6, 1, 1] = np.inf                                # unusable
    gn = np.array([np.linalg.norm(np.linalg.lstsq(J[v], -r[v], rcond=None)[0] * np.linalg.norm(J[v], axis=0))
                   if v != 6 else 1.0 for v in range(V)])
    radius = gn * np.array([10.0, 0.5, 0.3, 0.01, 0.2, 1e-4, 1.0, 2.0, 0.05])      # some inside, most outside
    d_in = np.zeros((V, q))
    d_in[1] = 3.0 * np.linalg.norm(J[1], axis=0)       # a larger scaling from earlier iterations stays
    lam_in = np.zeros(V)
    lam_in[3] = 7.0                                    # a poor starting guess
    dev = 'cuda'
    Jd, rd, Dd, Rd, Ld = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (J, r, d_in, radius, lam_in))
    delta = torch.empty((V, q), dtype=torch.float64, device=dev)
    pred = torch.empty((V,), dtype=torch.float64, device=dev)
    dxn = torch.empty((V,), dtype=torch.float64, device=dev)
    st = torch.empty((V,), dtype=torch.int32, device=dev)
    ctx = _lib.default_context()
    p = _lib.dev_ptr
    _lib.check(ctx.lib.sbm_lm_trust_step(ctx.handle, p(Jd), p(rd), p(Dd), p(Rd), p(Ld), V, M, q, p(delta), p(pred), p(dxn),
                                         p(st)), 'sbm_lm_trust_step')
    torch.cuda.synchronize()
    delta, pred, dxn, st, D, lam = (x.cpu().numpy() for x in (delta, pred, dxn, st, Dd, Ld))
    assert st.tolist() == [0, 0, 0, 0, 0, 0, 1, 0, 0] and np.all(delta[6] == 0)
    for v in range(V):
        if v == 6:
            continue
        cn = np.linalg.norm(J[v], axis=0)
        want_D = np.maximum(d_in[v], cn)
        want_D[want_D == 0] = 1.0
        assert np.allclose(D[v], want_D, rtol=1e-13)
        A, g = J[v].T @ J[v], J[v].T @ r[v]
        assert lam[v] >= 0.0
        assert np.allclose((A + lam[v] * np.diag(D[v] ** 2)) @ delta[v], -g, rtol=1e-7, atol=1e-9 * np.abs(g).max())
        assert dxn[v] == pytest.approx(np.linalg.norm(D[v] * delta[v]), rel=1e-10)
        if lam[v] == 0.0:
            assert dxn[v] <= 1.1 * radius[v]
        else:
            assert abs(dxn[v] - radius[v]) <= 0.1 * radius[v] * (1 + 1e-9)
        assert pred[v] == pytest.approx(-(g @ delta[v]) - 0.5 * delta[v] @ A @ delta[v], rel=1e-7, abs=1e-12)
    assert lam[0] == 0.0 and lam[7] == 0.0 and np.all(lam[[1, 2, 3, 5, 8]] > 0)


def test_lm_trust_step_ex_clip_skip_scale_and_near_singular_systems():
    """sbm_lm_trust_step_ex: the clipped step's own prediction (pred = -g.x - x^T J^T J x / 2, ||D x||, g.x), skipped
    vectors, row scaling, trial = theta + x; near-singular J^T J (columns equal to rounding: a Cholesky factorisation that
    fails for small damping and succeeds for larger) returns a SOLVED system or status 1, never the right-hand side; and
    q = 128 -- the largest the entry point admits -- fits the LDS with its smaller row tile."""
    import torch
    from sysbio_modeling_amd import _lib
    rng = np.random.default_rng(5)
    V, M, q = 8, 48, 10
    J = rng.standard_normal((V, M, q))
    r = rng.standard_normal((V, M))
    rs = rng.uniform(0.5, 2.0, M)
    # near-singular: two / three columns identical up to 1e-16 relative noise, large radius => tiny damping tried first
    J[3, :, 4] = J[3, :, 2] * (1.0 + 1e-16 * rng.standard_normal(M))
    J[5, :, 1] = J[5, :, 0]
    J[5, :, 9] = J[5, :, 0] * (1.0 + 1e-16 * rng.standard_normal(M))
    Js = J * rs[None, :, None]
    theta = rng.standard_normal((V, q))
    radius = np.array([1e3, 1e-2, 5.0, 1e6, 1.0, 1e8, 0.3, 1e3])
    skip = np.zeros(V, dtype=np.int32)
    skip[6] = 1
    max_step = 0.05
    dev = 'cuda'
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)     # noqa: E731
    Jd, rd, rsd, thd, Rd, skd = t(J), t(r), t(rs), t(theta), t(radius), t(skip)
    Dd = torch.zeros((V, q), dtype=torch.float64, device=dev)
    Ld = torch.zeros((V,), dtype=torch.float64, device=dev)
    delta, trial = (torch.empty((V, q), dtype=torch.float64, device=dev) for _ in range(2))
    pred, dxn, gtx = (torch.empty((V,), dtype=torch.float64, device=dev) for _ in range(3))
    st = torch.empty((V,), dtype=torch.int32, device=dev)
    ctx = _lib.default_context()
    p = _lib.dev_ptr
    _lib.check(ctx.lib.sbm_lm_trust_step_ex(ctx.handle, p(Jd), p(rd), p(Dd), p(Rd), p(Ld), V, M, q, p(rsd), p(skd), max_step,
                                            p(thd), p(trial), p(delta), p(pred), p(dxn), p(gtx), p(st)), 'sbm_lm_trust_step_ex')
    torch.cuda.synchronize()
    delta, trial, pred, dxn, gtx, st, D, lam = (x.cpu().numpy() for x in (delta, trial, pred, dxn, gtx, st, Dd, Ld))
    assert st[6] == 2 and np.all(delta[6] == 0) and np.array_equal(trial[6], theta[6])
    assert set(st[[0, 1, 2, 4, 7]].tolist()) == {0}
    for v in range(V):
        if st[v] != 0:
            assert np.all(delta[v] == 0) and pred[v] == 0 and np.array_equal(trial[v], theta[v])
            continue
        A, g = Js[v].T @ Js[v], Js[v].T @ r[v]
        assert np.all(np.abs(delta[v]) <= max_step * (1 + 1e-15)) and np.allclose(trial[v], theta[v] + delta[v], rtol=0, atol=1e-15)
        assert gtx[v] == pytest.approx(g @ delta[v], rel=1e-9, abs=1e-13)
        assert dxn[v] == pytest.approx(np.linalg.norm(D[v] * delta[v]), rel=1e-10)
        assert pred[v] == pytest.approx(-(g @ delta[v]) - 0.5 * delta[v] @ A @ delta[v], rel=1e-7, abs=1e-12)
        clipped = np.any(np.abs(delta[v]) >= max_step * (1 - 1e-12))
        if not clipped:          # an unclipped answer solves its damped system: never the right-hand side -g
            x = np.linalg.solve(A + lam[v] * np.diag(D[v] ** 2), -g)
            assert np.allclose(delta[v], x, rtol=1e-5, atol=1e-8 * np.abs(x).max())
        else:                    # a clipped one is the clipped solution of its system (the near-singular vectors included)
            x = np.linalg.lstsq(A + lam[v] * np.diag(D[v] ** 2), -g, rcond=None)[0]
            if v not in (3, 5):
                assert np.allclose(delta[v], np.clip(x, -max_step, max_step), rtol=1e-5, atol=1e-9)
        assert pred[v] >= -1e-12                       # a descent step of the model
    # the largest system: q = 128 (round 2 asked for 169 KB of LDS there and failed inside the HIP call)
    V2, M2, q2 = 2, 140, 128
    J2 = t(rng.standard_normal((V2, M2, q2)))
    r2 = t(rng.standard_normal((V2, M2)))
    D2 = torch.zeros((V2, q2), dtype=torch.float64, device=dev)
    R2 = torch.full((V2,), 1e3, dtype=torch.float64, device=dev)
    L2 = torch.zeros((V2,), dtype=torch.float64, device=dev)
    d2 = torch.empty((V2, q2), dtype=torch.float64, device=dev)
    p2, n2 = (torch.empty((V2,), dtype=torch.float64, device=dev) for _ in range(2))
    s2 = torch.empty((V2,), dtype=torch.int32, device=dev)
    _lib.check(ctx.lib.sbm_lm_trust_step(ctx.handle, p(J2), p(r2), p(D2), p(R2), p(L2), V2, M2, q2, p(d2), p(p2), p(n2), p(s2)),
               'sbm_lm_trust_step')
    torch.cuda.synchronize()
    assert s2.tolist() == [0, 0]
    for v in range(V2):
        Jv, rv = J2[v].cpu().numpy(), r2[v].cpu().numpy()
        assert np.allclose(d2[v].cpu().numpy(), np.linalg.lstsq(Jv, -rv, rcond=None)[0], rtol=1e-6, atol=1e-9)
    lam_buf = torch.zeros((V2,), dtype=torch.float64, device=dev)
    _lib.check(ctx.lib.sbm_lm_step(ctx.handle, p(J2), p(r2), p(lam_buf), V2, M2, q2, p(d2), p(p2), p(s2)), 'sbm_lm_step')
    torch.cuda.synchronize()
    assert s2.tolist() == [0, 0]


def test_fused_trust_region_loop_equals_the_tensor_select_loop(gpu_models):
    """fit_batch(algorithm='trust_region') -- lmder's bookkeeping in sbm_lm_update / sbm_lm_accept, two launches per
    iteration -- against round 2's spelling of the same algorithm in tensor selects (algorithm='trust_region_torch'):
    same iterations, same acceptances, costs and parameters equal to rounding; with and without the lazy Jacobian, and
    with a step bound that binds (the clipped step is judged by ITS predicted reduction in both)."""
    from sysbio_modeling_amd import models_zoo
    import warnings
    m = gpu_models('cascade20')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(m, n_exp=2, reference_compat=False)
    starts = th0[None, :] + 0.15 * np.random.default_rng(2).standard_normal((24, th0.size))
    for kw in (dict(), dict(lazy_jacobian=True), dict(max_step=0.02)):
        a = proj.fit_batch(starts, max_iter=12, algorithm='trust_region', trace=True, **kw)
        b = proj.fit_batch(starts, max_iter=12, algorithm='trust_region_torch', trace=True, **kw)
        assert np.array_equal(a['n_iter'], b['n_iter']) and np.array_equal(a['converged'], b['converged']), kw
        # the two spell the same sums in different orders (g . x from J^T r against r . (J x)): a start whose ratio sits on
        # one of lmder's thresholds may take the other branch -- most starts agree to rounding, all to 1e-3
        rel = np.abs(a['cost'] / b['cost'] - 1)
        print("fused vs tensor-select loop %s: %d of %d starts equal to 1e-9, worst %.2g; accepted per iteration %s / %s"
              % (kw, int(np.sum(rel <= 1e-9)), len(rel), rel.max(), [h['accepted'] for h in a['history']],
                 [h['accepted'] for h in b['history']]))
        assert np.sum(rel <= 1e-9) >= 0.8 * len(rel) and rel.max() <= 1e-3, (kw, rel)
        same = rel <= 1e-9
        assert np.allclose(a['theta'][same], b['theta'][same], rtol=0, atol=1e-6), kw
        acc_a, acc_b = [h['accepted'] for h in a['history']], [h['accepted'] for h in b['history']]
        assert len(acc_a) == len(acc_b) and sum(abs(x - y) for x, y in zip(acc_a, acc_b)) <= 3, (acc_a, acc_b)
        assert a['n_evaluations'] == b['n_evaluations']
    # reference_compat Jacobians (not divided by sigma) are scaled inside the step kernel
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj_c, _ = models_zoo.cascade_config4_project(m, n_exp=2, reference_compat=True)
    a = proj_c.fit_batch(starts[:8], max_iter=8)
    b = proj_c.fit_batch(starts[:8], max_iter=8, algorithm='trust_region_torch')
    assert np.allclose(a['cost'], b['cost'], rtol=1e-3) and np.sum(np.abs(a['cost'] / b['cost'] - 1) <= 1e-9) >= 6


def _exact_simple_project(m, theta_true, sf_groups=None):
    """Two experiments of the one-state model (shared k_synt, one k_deg each: the structure of
    tests/test_Project.py:27-72) with data the model itself produces at ``theta_true``."""
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    t = np.linspace(5.0, 100.0, 20)
    grid = np.linspace(0, 100.0, 1000)
    t_on_grid = grid[np.searchsorted(grid, t)]
    exps = []
    for name, kd in (('Low', theta_true[1]), ('High', theta_true[0])):
        y = m.simulate(np.exp([kd, theta_true[2]]), np.concatenate([[0.0], t_on_grid]))[1:, 0]
        exps.append(Experiment('%s_Deg_Exp' % name, TimecourseMeasurement('Variable_1', y, t),
                               experiment_settings={'Deg_Rate': name}))
    settings = {'Global': ['k_synt'], 'Shared': {'Group_1': {'k_deg': ('Deg_Rate',)}}}
    return Project(m, exps, settings, {'Variable_1': ('direct', 0)}, sf_groups=sf_groups, reference_compat=False)


@pytest.mark.parametrize('algorithm', ['trust_region', 'marquardt'])
def test_multi_start_fit_matches_leastsq(gpu_models, algorithm):
    """The reference fits with leastsq(proj.residuals, x0, Dfun=proj.calc_project_jacobian)
    (tests/test_Project.py:202-213).  Same call through this package's Project (batch of one), and 64
    starts at once with fit_batch: all reach the parameters that generated the data."""
    m = gpu_models('simple')
    proj = _exact_simple_project(m, np.log([0.05, 0.02, 0.3]))
    truth = np.zeros(3)
    truth[proj.get_param_index('Group_1', ('High',))] = np.log(0.05)
    truth[proj.get_param_index('Group_1', ('Low',))] = np.log(0.02)
    truth[proj.get_param_index('k_synt', 'Global')] = np.log(0.3)
    assert proj.calc_sum_square_residuals(truth) < 1e-16
    x0 = truth + np.array([0.4, -0.3, 0.5])
    x_ref, _, info, _, ier = leastsq(proj.residuals, x0, Dfun=proj.calc_project_jacobian, full_output=True)
    assert ier in (1, 2, 3, 4) and np.allclose(x_ref, truth, atol=1e-6)
    rng = np.random.default_rng(5)
    starts = truth[None, :] + rng.uniform(-1.0, 1.0, (64, 3))
    starts[0] = x0
    fit = proj.fit_batch(starts, max_iter=60, algorithm=algorithm)
    assert fit['converged'].all()
    assert np.allclose(fit['theta'], truth[None, :], atol=1e-6)
    assert fit['cost'].max() < 1e-14
    # one batched evaluation per iteration: far fewer launches than 64 serial fits
    assert fit['n_evaluations'] <= 64 * 61 and fit['n_iter'].max() <= 40
    assert info['nfev'] >= 4          # leastsq needed several serial evaluations for ONE start


@pytest.mark.parametrize('algorithm', ['trust_region', 'marquardt'])
def test_fit_batch_on_the_config4_project(gpu_models, algorithm):
    """configs[3]-style project (8 experiments, 512 rows, 68 parameters, 4 free scale factors) with
    noise-free data: a sloppy problem (rate constants and scale factors trade off), so the test is on the
    cost, not on the parameters: 32 starts scattered around the truth, 40 iterations = 41 batched
    evaluations of 256 trajectories each."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('cascade20')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(m, noise=0.0, reference_compat=False)
        assert proj.calc_sum_square_residuals(th0) < 1e-12
        starts = th0[None, :] + 0.15 * np.random.default_rng(1).standard_normal((32, th0.size))
        c0 = proj.calc_sum_square_residuals_batch(starts)
        fit = proj.fit_batch(starts, max_iter=40, algorithm=algorithm)
    assert fit['n_evaluations'] == 32 * 41
    assert np.all(fit['cost'] <= c0)
    assert np.median(fit['cost']) < 1e-5 * np.median(c0)
    assert (fit['cost'] < 1e-3 * c0).sum() >= 28


def test_multi_chain_sampler_reproduces_the_gaussian_posterior(gpu_models):
    """ensemble_log_params_batch (reference Ensembles.py:55-178, C chains at once).  Noise-free data with
    1 % error bars: the posterior is the Gaussian N(theta*, (J^T J)^-1) to a good approximation, so the
    pooled chains must reproduce its mean and standard deviations; the acceptance ratio of the reference's
    candidate density (expected quadratic cost increase ~ 1) sits around one half."""
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    from sysbio_modeling_amd.project.ensembles import ensemble_log_params_batch, sampling_matrix
    m = gpu_models('simple')
    truth_p = np.array([0.05, 0.3])            # k_deg, k_synt
    t = np.linspace(5.0, 100.0, 20)
    grid = np.linspace(0, 100.0, 1000)
    y = m.simulate(truth_p, np.concatenate([[0.0], grid[np.searchsorted(grid, t)]]))[1:, 0]
    exp = Experiment('E', TimecourseMeasurement('Variable_1', y, t, 0.01 * y))
    proj = Project(m, [exp], {'Global': ['k_deg', 'k_synt']}, {'Variable_1': ('direct', 0)}, reference_compat=False)
    truth = np.zeros(2)
    truth[proj.get_param_index('k_deg', 'Global')] = np.log(0.05)
    truth[proj.get_param_index('k_synt', 'Global')] = np.log(0.3)
    J = proj.calc_project_jacobian(truth)
    cov = np.linalg.inv(J.T @ J)
    # the candidate density: samp samp^T = (0.5 H)^-1 / q
    samp = sampling_matrix(J.T @ J)
    assert np.allclose(samp @ samp.T, np.linalg.inv(0.5 * J.T @ J) / 2, rtol=1e-9)
    ens, ens_F, ratio = ensemble_log_params_batch(proj, np.tile(truth, (128, 1)), steps=400, seeds=11)
    assert ens.shape == (401, 128, 2) and ens_F.shape == (401, 128) and ratio.shape == (128,)
    assert 0.3 < ratio.mean() < 0.75
    assert np.all(ens_F[0] < 1e-12) and np.all(ens_F >= 0)
    pooled = ens[100:].reshape(-1, 2)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(pooled.mean(axis=0) - truth) < 0.1 * sd)
    assert np.allclose(pooled.std(axis=0), sd, rtol=0.15)
    assert np.corrcoef(pooled.T)[0, 1] == pytest.approx(cov[0, 1] / (sd[0] * sd[1]), abs=0.1)
    # skip_elems thins the record, not the walk
    ens2, _, _ = ensemble_log_params_batch(proj, truth, steps=50, seeds=3, skip_elems=4)
    assert ens2.shape == (11, 1, 2)


def test_sampler_with_recalculated_hessian(gpu_models):
    """ensemble_log_params_batch(recalc_hess_alg=True): the reference's second algorithm (Ensembles.py:153-157, :200-224),
    J^T J at every trial point from the batched device Jacobian and the Metropolis-Hastings ratio with both candidate
    densities.  On the one-species model with noise-free data the posterior is N(theta*, (J^T J)^-1) to a good
    approximation: pooled chains reproduce its standard deviations."""
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    from sysbio_modeling_amd.project.ensembles import ensemble_log_params_batch
    m = gpu_models('simple')
    t = np.linspace(5.0, 100.0, 20)
    grid = np.linspace(0, 100.0, 1000)
    y = m.simulate(np.array([0.05, 0.3]), np.concatenate([[0.0], grid[np.searchsorted(grid, t)]]))[1:, 0]
    exp = Experiment('E', TimecourseMeasurement('Variable_1', y, t, 0.01 * y))
    proj = Project(m, [exp], {'Global': ['k_deg', 'k_synt']}, {'Variable_1': ('direct', 0)}, reference_compat=False)
    truth = np.zeros(2)
    truth[proj.get_param_index('k_deg', 'Global')] = np.log(0.05)
    truth[proj.get_param_index('k_synt', 'Global')] = np.log(0.3)
    J = proj.calc_project_jacobian(truth)
    sd = np.sqrt(np.diag(np.linalg.inv(J.T @ J)))
    ens, ens_F, ratio = ensemble_log_params_batch(proj, np.tile(truth, (128, 1)), steps=250, seeds=3, recalc_hess_alg=True)
    assert ens.shape == (251, 128, 2) and 0.3 < ratio.mean() < 0.8 and np.all(ens_F >= 0)
    pooled = ens[60:].reshape(-1, 2)
    assert np.all(np.abs(pooled.mean(axis=0) - truth) < 0.12 * sd)
    assert np.allclose(pooled.std(axis=0), sd, rtol=0.15)


def test_fit_and_sampler_on_a_model_with_its_own_context(gpu_models):
    """A model enabled on an explicit Context (what bench.py does per rank: `model.enable_jit(_lib.Context(local_rank))`)
    is fitted and sampled through THAT context's handle -- the optimiser's device calls (sbm_lm_trust_step, sbm_lm_step)
    take the context of the project's model, not the process default -- with the same numbers as on the default one."""
    from sysbio_modeling_amd import _lib
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import zoo_model
    from sysbio_modeling_amd.project.ensembles import ensemble_log_params_batch
    gm = zoo_model('simple')
    own = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='simple')
    ctx = _lib.Context(0)
    own.enable_jit(ctx)
    assert own.device_model.ctx is ctx and ctx is not _lib.default_context()
    truth = np.log([0.05, 0.02, 0.3])
    starts = None
    results = []
    for m in (own, gpu_models('simple')):
        proj = _exact_simple_project(m, truth)
        th = np.zeros(3)
        th[proj.get_param_index('Group_1', ('High',))] = truth[0]
        th[proj.get_param_index('Group_1', ('Low',))] = truth[1]
        th[proj.get_param_index('k_synt', 'Global')] = truth[2]
        if starts is None:
            starts = th[None, :] + np.random.default_rng(2).uniform(-0.5, 0.5, (8, 3))
        fits = [proj.fit_batch(starts, max_iter=25, algorithm=a) for a in ('trust_region', 'marquardt')]
        ens = ensemble_log_params_batch(proj, np.tile(th, (4, 1)), steps=5, seeds=9)[0]
        results.append((fits[0]['theta'], fits[1]['theta'], ens))
        assert np.allclose(fits[0]['theta'], th[None, :], atol=1e-6) and np.allclose(fits[1]['theta'], th[None, :], atol=1e-6)
    for a, b in zip(*results):
        assert np.array_equal(a, b)
    own.disable_jit()
    ctx.close()


@pytest.mark.gpu
def test_trajectory_steps_belong_to_the_last_evaluation(gpu_models):
    """sbm_project_trajectory_steps hands out the per-trajectory step counts of the project's LAST evaluation: asking
    with another V -- one that the scratch buffers of an earlier, larger batch could hold -- is an argument error, not a
    success with stale counts (round 3 checked the capacity only; the fit's trial budget is derived from this call)."""
    import torch
    from sysbio_modeling_amd import _lib, models_zoo
    model = gpu_models('cascade20')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(model, n_exp=2)
    lib = _lib.load_library()
    th = th0[None, :] + 0.05 * np.random.default_rng(3).standard_normal((12, th0.size))
    proj.evaluate_batch(th, want=('n_steps',))
    per = torch.empty((12, 2), dtype=torch.int32, device='cuda')
    _lib.check(lib.sbm_project_trajectory_steps(proj._device(), 12, _lib.dev_ptr(per)), 'steps')
    total = np.asarray(proj.evaluate_batch(th[:5], want=('n_steps',))['n_steps'])
    per5 = torch.empty((5, 2), dtype=torch.int32, device='cuda')
    _lib.check(lib.sbm_project_trajectory_steps(proj._device(), 5, _lib.dev_ptr(per5)), 'steps')
    assert np.array_equal(per5.sum(dim=1).cpu().numpy(), total) and torch.equal(per5, per[:5])
    with pytest.raises(_lib.SbmError, match="last evaluation had 5"):
        _lib.check(lib.sbm_project_trajectory_steps(proj._device(), 12, _lib.dev_ptr(per)), 'steps')
