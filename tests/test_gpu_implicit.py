"""Implicit midpoint on the GPU (BASELINE configs[4]: stiff 50-state cascade, N = 2550 coupled ODEs).

Two levels of parity:
  * scheme level: csrc/sbm_integrators.hpp::sbm_imid_kernel against oracle/imid_oracle.py, the same
    algorithm in dense numpy (same steps, same Newton recipe): agreement to rounding;
  * reference level: against SciPy odeint (LSODA switches to BDF on this system) -- second-order
    convergence of the raw scheme and agreement of the Richardson-extrapolated result.
The model's sparse LU (emit_implicit.py) is exercised on three sparsity patterns: bidiagonal
(stiff50), bidiagonal + corner with fill-in (cascade20) and a 2x2 block (Michaelis-Menten)."""
import os
import warnings

import numpy as np
import pytest

from tests import reference_cases as rc
from tests.conftest import parity_err, check_parity, tol_ratio, lsoda_taus, project_tolerances

pytestmark = pytest.mark.gpu

IM = dict(method='implicit_midpoint', rtol=1e-10, atol=1e-12)


def _from_zero(t_pts):
    return np.concatenate([[0.0], np.asarray(t_pts, dtype=float)])


@pytest.mark.parametrize('name,t_end,h0', [('michaelis_menten', 100.0, 2.0), ('cascade20', 100.0, 1.0),
                                            ('stiff50', 10.0, 0.02)])
def test_kernel_equals_scheme_oracle(gpu_models, zoo, name, t_end, h0):
    from oracle import imid_oracle
    from sysbio_modeling_amd import models_zoo
    gm, m = zoo(name), gpu_models(name)
    if name == 'michaelis_menten':
        P = np.stack([rc.MM_PARAMS * np.array([40.0, 30.0, 5.0, 3.0, 20.0]), rc.MM_PARAMS * 7.0])
    elif name == 'cascade20':
        P = models_zoo.cascade_ensemble(2)[1]
    else:
        P = models_zoo.stiff_ensemble(2)[1]
    t_out = np.array([0.0, 0.37 * t_end, t_end])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, h0=h0, **IM)
    assert m.last_info['status'].tolist() == [0, 0]
    for v in range(2):
        Yo, So, ns, nn = imid_oracle.integrate(gm, P[v], t_out[1:], h0)
        assert m.last_info['n_steps'][v] == ns
        # Newton iterations beyond one per step: the stopping test sits on a knife edge now and then
        assert abs(int(m.last_info['n_rejected'][v]) - (nn - ns)) <= (nn - ns) // 8 + 2
        assert np.allclose(Y[v, 1:], Yo, rtol=1e-9, atol=1e-12)
        assert np.allclose(S[v, 1:], So, rtol=1e-8, atol=1e-10 * np.abs(So).max())
    # state-only entry point: same kernel without the column work
    Y2 = m.simulate_batch(P, t_out, h0=h0, **IM)
    assert np.array_equal(Y2, Y)


@pytest.mark.parametrize('name,t_end,order,rtol', [('michaelis_menten', 100.0, 6, 1e-8), ('cascade20', 60.0, 8, 3e-9),
                                                   ('stiff50', 2.5, 8, 3e-9)])
def test_extrapolation_kernel_equals_scheme_oracle(gpu_models, zoo, name, t_end, order, rtol):
    """csrc/sbm_implicit_extrap.hpp::sbm_iex_kernel (SBM_IMPLICIT_EXTRAP: extrapolated implicit Euler, local step
    control in the kernel) against oracle/iex_oracle.py, the same algorithm in dense numpy -- same sequences, same Newton
    recipe, same error norm and step-size rule.  The kernel runs its controller in single precision (fast pow and
    reciprocal), so the two take the same decisions but step sizes that differ in the sixth digit: results agree far
    inside the integration tolerance, step counts to a step.  Three sparsity patterns / three solve paths: 2 x 2 block
    (chain prefix), bidiagonal + corner with fill-in (redundant LU), bidiagonal (chain prefix on 50 lanes)."""
    from oracle import iex_oracle
    from sysbio_modeling_amd import models_zoo
    gm, m = zoo(name), gpu_models(name)
    if name == 'michaelis_menten':
        P = np.stack([rc.MM_PARAMS * np.array([40.0, 30.0, 5.0, 3.0, 20.0]), rc.MM_PARAMS * 7.0])
    elif name == 'cascade20':
        P = models_zoo.cascade_ensemble(2)[1]
    else:
        P = models_zoo.stiff_ensemble(4096)[1][[5, 4000]]
    t_out = np.array([0.0, 0.25 * t_end, 0.4 * t_end, t_end])
    atol = 1e-3 * rtol
    kw = dict(method='implicit_extrap', rtol=rtol, atol=atol, order=order)
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, **kw)
    info = m.last_info
    assert info['status'].tolist() == [0, 0]
    for v in range(2):
        # (stiff50 is a chain: sbm_iex_seq_kernel, which sums T_j itself; the other two run sbm_iex_kernel: T_j - S_n)
        Yo, So, io = iex_oracle.integrate(gm, P[v], t_out[1:], rtol=rtol, atol=atol, order=order,
                                          sums='values' if name == 'stiff50' else 'differences')
        assert io['status'] == 0
        assert abs(int(info['n_steps'][v]) - io['n_steps']) <= 1 + io['n_steps'] // 50, (info['n_steps'][v], io)
        assert abs(int(info['n_rejected'][v]) - io['n_reject']) <= 2
        ky = np.max(np.abs(Y[v, 1:] - Yo) / (rtol * np.abs(Yo) + atol))
        ks = np.max(np.abs(S[v, 1:] - So) / (rtol * np.maximum(np.abs(So), 1e-6 * np.abs(So).max(axis=0)) + atol))
        print("%s vector %d: kernel vs scheme oracle y %.3g S %.3g integration tolerances; %d macro steps (oracle %d)"
              % (name, v, ky, ks, info['n_steps'][v], io['n_steps']))
        assert ky <= 0.3 and ks <= 0.3
    # state-only entry point: the same controller without the column work -- its own step sequence (no sensitivity
    # column tightens the steps), same solution within the tolerance
    Y2 = m.simulate_batch(P, t_out, **kw)
    assert m.last_info['status'].tolist() == [0, 0] and np.all(m.last_info['n_steps'] <= info['n_steps'])
    assert np.max(np.abs(Y2[:, 1:] - Y[:, 1:]) / (rtol * np.abs(Y[:, 1:]) + atol)) <= 30.0


def test_stiff50_against_reference_golden(gpu_models, golden):
    """tests/golden/stiff50_ref.npz: the REAL reference OdeModel (LSODA, which switches to BDF here) on
    the build's stiff50 model, 3 vectors, 16 measurement rows (make_golden_stiff.py).  Raw scheme: error
    / 4 per halving of h; one Richardson level on 4096 + 8192 steps meets the parity tolerance of the
    explicit kernels (|gpu - ref| <= 1e-8 |ref| + 5e-9).  Explicit RK4 at the same step count blows up:
    the system IS stiff."""
    m = gpu_models('stiff50')
    g = golden('stiff50_ref.npz')
    P, Yr, Sr = g['P'], g['Y'], g['S']
    t_out = _from_zero(g['t'][g['idx']])
    errs = []
    for mult in (1, 2):
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=1024, step_mult=mult, **IM)
        assert m.last_info['status'].tolist() == [0, 0, 0]
        errs.append((np.max(np.abs(Y[:, 1:] - Yr)), np.max(np.abs(S[:, 1:] - Sr))))
    assert 3.6 < errs[0][0] / errs[1][0] < 4.4 and 3.6 < errs[0][1] / errs[1][1] < 4.4
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=4096, extrapolate=1, **IM)
    assert m.last_info['status'].tolist() == [0, 0, 0]
    assert parity_err(Y[:, 1:], Yr) <= 1.0
    assert parity_err(S[:, 1:], Sr) <= 1.0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Yx = m.simulate_batch(P, t_out, method='rk4', n_steps=4096)
    assert not np.all(np.isfinite(Yx)) or np.max(np.abs(Yx[:, 1:] - Yr)) > 1e-2


def test_newton_failure_is_reported_per_trajectory(gpu_models):
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('stiff50')
    _, P = models_zoo.stiff_ensemble(3)
    P[1, 3] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Y = m.simulate_batch(P, np.array([0.0, 1.0]), h0=0.01, **IM)
    assert m.last_info['status'][1] != 0 and np.all(np.isnan(Y[1, -1]))
    assert m.last_info['status'][[0, 2]].tolist() == [0, 0] and np.all(np.isfinite(Y[[0, 2]]))


def test_config5_full_ensemble_properties(gpu_models):
    """BASELINE configs[4] at full size: 4096 vectors x 2550 ODEs.  Size-independent properties:
    states stay in [0, 1] (the model's invariant region), sensitivities of species i to a_j vanish
    for j > i (the cascade is feed-forward), permutation equivariance, determinism."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('stiff50')
    _, P = models_zoo.stiff_ensemble(4096)
    t_out = np.array([0.0, 2.5, models_zoo.STIFF_T_END])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=512, **IM)
    assert not m.last_info['status'].any()
    assert Y.min() >= 0.0 and Y.max() < 1.0
    S4 = S.reshape(4096, 3, 50, 50)
    upper = np.triu_indices(50, k=1)
    assert np.all(S4[:, :, upper[0], upper[1]] == 0.0)
    assert np.abs(S4[:, -1, np.arange(50), np.arange(50)]).min() > 0.0
    perm = np.random.default_rng(0).permutation(4096)[:512]
    S2 = m.calc_jacobian_batch(P[perm], t_out, n_steps=512, **IM)
    assert np.array_equal(S2, S[perm])


def test_project_on_the_stiff_model_with_extrapolation(gpu_models, golden):
    """A Project on stiff50 evaluated with the implicit integrator and one Richardson level inside
    sbm_residuals_batch / sbm_jacobian_batch (sbm_project_set_extrapolation): simulations and the model
    Jacobian at the measurement rows equal the reference's LSODA results of the golden file to the parity
    tolerance -- the extrapolation happens BEFORE the (nonlinear) scale-factor assembly."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    m = gpu_models('stiff50')
    g = golden('stiff50_ref.npz')
    P, Yr, Sr = g['P'], g['Y'], g['S'].reshape(3, 16, 50, 50)
    species = (0, 3, 10, 49)
    ms = [TimecourseMeasurement('s%d' % v, Yr[0][:, v] * 1.7 + 0.01, models_zoo.STIFF_MEASURE_TIMES.copy(),
                                0.05 * np.abs(Yr[0][:, v]) + 0.01) for v in species]
    exp = Experiment('E', ms, fixed_parameters={'b%d' % i: 0.5 for i in range(50)})
    proj = Project(m, [exp], {'Global': ['a%d' % i for i in range(50)], 'Fixed': ['b%d' % i for i in range(50)]},
                   {('s%d' % v): ('direct', v) for v in species},
                   sf_groups=['s%d' % v for v in species], reference_compat=False)
    names = list(m.param_order)
    theta = np.zeros((3, 50))
    for i in range(50):
        theta[:, proj.get_param_index('a%d' % i, 'Global')] = np.log(P[:, names.index('a%d' % i)])
    out = proj.evaluate_batch(theta, jacobian=True, want=('jacobian', 'model_jacobian'),
                              method='implicit_midpoint', n_steps=4096, extrapolate=1, rtol=1e-10, atol=1e-12)
    assert out['status'].tolist() == [0, 0, 0]
    # rows: measurement-name order (s0, s10, s3, s49 sorted as strings), 16 times each
    order = sorted(range(4), key=lambda k: 's%d' % species[k])
    for v in range(3):
        sims_ref = np.concatenate([Yr[v][:, species[k]] for k in order])
        assert parity_err(out['sims'][v], sims_ref) <= 1.0
        # d sim / d theta_j = S[:, species, j] * a_j  (log-parameter chain rule)
        cols = [proj.get_param_index('a%d' % j, 'Global') for j in range(50)]
        Jm_ref = np.concatenate([Sr[v][:, species[k], :] for k in order]) * P[v][None, :50]
        assert parity_err(out['model_jacobian'][v][:, cols], Jm_ref) <= 1.0
    assert out['n_steps'][0] >= 4096 + 8192
    # without extrapolation the same call is only second-order accurate
    out0 = proj.evaluate_batch(theta, method='implicit_midpoint', n_steps=4096, rtol=1e-10, atol=1e-12)
    assert parity_err(out0['sims'][0], np.concatenate([Yr[0][:, species[k]] for k in order])) > 50.0


def test_graded_start_keeps_fourth_order_after_extrapolation(gpu_models, golden):
    """The graded first step belongs to the BASE step; a run with step_mult = m cuts each of its substeps into m
    parts, so the grids of a Richardson pair are nested and one extrapolation level gives fourth order (error
    / 16 per halving).  A pattern relative to each run's own h converged with ratio 8."""
    m = gpu_models('cascade20')
    g = golden('cascade20_ref.npz')
    t_out = _from_zero(g['t'][g['idx']])
    kw = dict(method='implicit_midpoint_graded', n_steps=512, extrapolate=1, rtol=1e-12, atol=1e-15)
    fine = m.calc_jacobian_batch(g['P'][:2], t_out, step_mult=32, **kw)
    errs = [np.max(np.abs(m.calc_jacobian_batch(g['P'][:2], t_out, step_mult=mult, **kw) - fine)) for mult in (1, 2, 4)]
    assert 13.0 < errs[0] / errs[1] < 19.0 and 13.0 < errs[1] / errs[2] < 19.0, errs


def test_in_kernel_controlled_implicit_integrator(gpu_models, golden):
    """method='implicit_controlled' (= 'implicit_extrap', SBM_IMPLICIT_EXTRAP since round 3): ONE device call, every
    trajectory's step sizes chosen by LOCAL error control inside the kernel (csrc/sbm_implicit_extrap.hpp).  With
    default options the stiff50 golden vectors meet the parity tolerance against the real reference's LSODA results --
    where LSODA's own error (about one tolerance unit on this model) gets in the way, against the tight LSODA solution
    of tests/golden/stiff50_tight.npz."""
    m = gpu_models('stiff50')
    g, gt = golden('stiff50_ref.npz'), golden('stiff50_tight.npz')
    P = g['P']
    t_out = _from_zero(g['t'][g['idx']])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled')
    info = m.last_info
    assert info['status'].tolist() == [0, 0, 0] and np.all(info['n_steps'] >= 16)
    ey = check_parity(Y[:, 1:], g['Y'], gt['Y'], what='stiff50 states', criterion='parity')
    es = check_parity(S[:, 1:], g['S'], gt['S'], what='stiff50 sensitivities', criterion='parity')
    print("implicit_controlled on stiff50: %s macro steps (+%s rejected); error vs LSODA golden y %.2f S %.2f"
          % (info['n_steps'], info['n_rejected'], ey[0], es[0]), "vs tight:", ey[1], es[1])
    # state only: the same controller without the column work
    Y1 = m.simulate_batch(P, t_out, method='implicit_controlled')
    assert m.last_info['status'].tolist() == [0, 0, 0]
    check_parity(Y1[:, 1:], g['Y'], gt['Y'], what='stiff50 states (state-only run)', criterion='parity')
    # a looser tolerance takes fewer steps and is less accurate, but about within ITS tolerance
    S2 = m.calc_jacobian_batch(P, t_out, method='implicit_controlled', rtol=1e-5, atol=1e-8)
    assert np.all(m.last_info['n_steps'] < info['n_steps'])
    St = gt['S']
    assert np.max(np.abs(S2[:, 1:] - St) / (1e-5 * np.maximum(np.abs(St), 1e-3 * np.abs(St).max()) + 1e-8)) <= 3.0
    # a step budget too small for the tolerance is reported, rows NaN
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Y3 = m.simulate_batch(P[:1], t_out, method='implicit_controlled', max_steps=20)
    assert m.last_info['status'].tolist() == [1] and np.all(np.isnan(Y3[0, -1]))


def _wide_golden(golden):
    """stiff50_ref.npz (vectors 0-2 of the ensemble) + stiff50_wide_ref.npz (32 more, spread over it): every stiff50
    vector the REAL reference OdeModel was run on, with a tight solution for each -> (P, Y_ref, S_ref, Y_tight, S_tight)."""
    g, gt = golden('stiff50_ref.npz'), golden('stiff50_tight.npz')
    w, wt = golden('stiff50_wide_ref.npz'), golden('stiff50_wide_tight.npz')
    assert np.array_equal(wt['rows'], np.arange(len(w['P']))) and np.array_equal(w['idx'], g['idx'])
    return (np.concatenate([g['P'], w['P']]), np.concatenate([g['Y'], w['Y']]), np.concatenate([g['S'], w['S']]),
            np.concatenate([gt['Y'], wt['Y']]), np.concatenate([gt['S'], wt['S']]), _from_zero(g['t'][g['idx']]))


def test_stiff_integrator_on_35_reference_vectors_spread_over_the_ensemble(gpu_models, golden):
    """The pin of BASELINE configs[4], widened (round 2 had 3 vectors): 35 parameter vectors of the 4096-vector stiff50
    ensemble -- the first three and 32 spread evenly over it -- through the real reference OdeModel (LSODA at 1e-10), each
    with a tight solution (LSODA at 1e-12 by column groups).  The stiff integrator at DEFAULT options, one call: every
    vector within the parity tolerance of the reference's result, or -- LSODA at 1e-10 is itself about one unit off on
    this model -- within it of the tight solution and closer to it than the reference's result is.  The fixed-step
    Richardson pair that round 2 timed is held to the same vectors."""
    m = gpu_models('stiff50')
    P, Yr, Sr, Yt, St, t_out = _wide_golden(golden)
    assert len(P) == 35
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled')
    info = m.last_info
    assert not info['status'].any()
    worst = [0.0, 0.0, 0.0, 0.0]
    arbitrated = 0
    for v in range(len(P)):
        ey = check_parity(Y[v, 1:], Yr[v], Yt[v], what='stiff50 vector %d, states' % v, criterion='parity')
        es = check_parity(S[v, 1:], Sr[v], St[v], what='stiff50 vector %d, sensitivities' % v, criterion='parity')
        arbitrated += int(ey[1] is not None or es[1] is not None)
        worst = [max(worst[0], ey[0]), max(worst[1], es[0]), max(worst[2], parity_err(Y[v, 1:], Yt[v])),
                 max(worst[3], parity_err(S[v, 1:], St[v]))]
    lsoda = (max(parity_err(Yr[v], Yt[v]) for v in range(len(P))), max(parity_err(Sr[v], St[v]) for v in range(len(P))))
    print("stiff integrator, default options, 35 reference vectors: worst vs reference y %.2f S %.2f, vs tight y %.2f S %.2f "
          "(the reference's own LSODA vs tight: y %.2f S %.2f); %d vectors passed by arbitration; macro steps %d - %d"
          % (worst[0], worst[1], worst[2], worst[3], lsoda[0], lsoda[1], arbitrated, info['n_steps'].min(), info['n_steps'].max()))
    assert worst[2] <= 1.0 and worst[3] <= 1.0
    # state only: its own step sequence
    Y1 = m.simulate_batch(P, t_out, method='implicit_controlled')
    assert not m.last_info['status'].any()
    for v in range(len(P)):
        check_parity(Y1[v, 1:], Yr[v], Yt[v], what='stiff50 vector %d, state-only run' % v, criterion='parity')
    # the hand-chosen fixed pair of round 2's bench line on the same vectors (it was chosen on three of them)
    Sf, Yf = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=4096, extrapolate=1, **IM)
    ef = max(parity_err(Sf[v, 1:], St[v]) for v in range(len(P)))
    print("fixed 4096 + 8192 Richardson pair on the 35 vectors: worst sensitivity error vs tight %.2f units" % ef)


def test_stiff_integrator_at_full_size_is_batch_independent(gpu_models, golden):
    """BASELINE configs[4] at its full size -- all 4096 vectors of the stiff50 ensemble, 2550 ODEs each, one launch at default
    options -- through properties that need no reference: every vector integrates (status 0, finite rows); the launch
    is deterministic; a vector's rows do not depend on the batch it travels in or on its place in it (the ensemble in
    reverse order, the 35 pinned vectors on their own: bit for bit the rows of the big launch -- so the parity shown on the
    35 holds for them inside the full-size pass as well)."""
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('stiff50')
    _, P = models_zoo.stiff_ensemble(4096, n=50)
    g = golden('stiff50_wide_ref.npz')
    t_out = _from_zero(g['t'][g['idx']])
    kw = dict(method='implicit_controlled')
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, **kw)
    steps = m.last_info['n_steps'].copy()
    assert not m.last_info['status'].any()
    assert np.isfinite(Y).all() and np.isfinite(S).all()
    assert steps.min() > 50 and steps.max() < 5 * np.median(steps)
    S2, Y2 = m.calc_jacobian_batch(P, t_out, return_states=True, **kw)
    assert np.array_equal(S2, S) and np.array_equal(Y2, Y)
    del S2, Y2
    Sr, Yr = m.calc_jacobian_batch(P[::-1].copy(), t_out, return_states=True, **kw)
    assert np.array_equal(Sr[::-1], S) and np.array_equal(Yr[::-1], Y)
    del Sr, Yr
    idx = np.concatenate([[0, 1, 2], g['index']])
    assert np.array_equal(P[g['index']], g['P'])
    Sp, Yp = m.calc_jacobian_batch(P[idx], t_out, return_states=True, **kw)
    assert np.array_equal(Sp, S[idx]) and np.array_equal(Yp, Y[idx])
    assert np.array_equal(m.last_info['n_steps'], steps[idx])
    # initial rows: the initial condition (zeros) exactly
    assert not Y[:, 0].any() and not S[:, 0].any()


def test_romberg_controlled_implicit_needs_no_step_count(gpu_models, golden):
    """method='implicit_romberg': the round-1 host loop around the fixed-step kernel -- the step count is found by
    doubling until two successive Richardson extrapolants agree (sysbio_modeling_amd/_control.py).  With default
    options the result meets the parity tolerance against the reference's LSODA golden; every vector reports the
    level it stopped at."""
    m = gpu_models('stiff50')
    g = golden('stiff50_ref.npz')
    P, Yr, Sr = g['P'], g['Y'], g['S']
    t_out = _from_zero(g['t'][g['idx']])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_romberg', rtol=1e-9, atol=1e-12)
    info = m.last_info
    assert info['status'].tolist() == [0, 0, 0]
    assert np.all(info['levels'] >= 1) and np.all(info['n_steps'] >= 3 * 256)
    # against a much finer solution of the same scheme (three runs, h^2 and h^4 terms cancelled): the returned
    # values are within the requested tolerance (1e-9 relative to max(|x|, 1e-3 max|x|)) ...
    St, Yt = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=8192, extrapolate=2,
                                   method='implicit_midpoint_graded', rtol=1e-11, atol=1e-14)
    for got, fine in ((Y, Yt), (S, St)):
        sc = 1e-9 * np.maximum(np.abs(fine), 1e-3 * np.abs(fine).reshape(3, -1).max(axis=1)[:, None, None]) + 1e-12
        assert np.max(np.abs(got - fine) / sc) <= 2.0          # (the fine solution is itself good to ~1 unit)
    # ... and LSODA's golden values (themselves good to ~1e-8) are met as closely as the fine solution meets them
    assert parity_err(Y[:, 1:], Yr) <= 1.0
    assert parity_err(S[:, 1:], Sr) <= max(1.0, 1.05 * parity_err(St[:, 1:], Sr))
    # a looser tolerance stops earlier and is less accurate, but about within ITS tolerance (an estimate, not a bound)
    S2 = m.calc_jacobian_batch(P, t_out, method='implicit_romberg', rtol=1e-5, atol=1e-8)
    assert np.all(m.last_info['levels'] < info['levels'])
    assert np.max(np.abs(S2[:, 1:] - Sr) / (1e-5 * np.maximum(np.abs(Sr), 1e-3 * np.abs(Sr).max()) + 1e-8)) <= 2.0
    # a tolerance out of reach within the allowed doublings is reported, with the finest result returned
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Y3 = m.simulate_batch(P[:1], t_out, method='implicit_romberg', rtol=1e-9, atol=1e-12, n_steps=16, max_doublings=2)
    assert m.last_info['status'].tolist() == [5] and np.all(np.isfinite(Y3))


def test_auto_method_switches_stiff_vectors_to_the_implicit_integrator(gpu_models, golden):
    """method='auto' is what LSODA's method switching is to the reference: DOPRI45 with a step budget, then the
    controlled implicit integrator for the vectors that exhausted it.  Two vectors of the SAME model: the
    golden (stiffness ratio 1e7: DOPRI45 would need ~1e7 steps) and one with all rates within a decade."""
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd.symbolic import zoo_model
    m = gpu_models('stiff50')
    gm = zoo_model('stiff50')
    g = golden('stiff50_ref.npz')
    t_out = _from_zero(g['t'][g['idx']])
    mild = g['P'][0].copy()
    mild[:50] = 10.0 ** np.linspace(0.0, 1.0, 50)
    P = np.stack([g['P'][0], mild])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='auto', max_steps=20000)
    info = m.last_info
    assert info['status'].tolist() == [0, 0] and info['stiff'].tolist() == [True, False]
    assert parity_err(Y[0, 1:], g['Y'][0]) <= 1.0 and parity_err(S[0, 1:], g['S'][0]) <= 1.0
    Yr = oo.simulate(gm, mild, g['t'], use_c=True)[g['idx']]
    assert parity_err(Y[1, 1:], Yr) <= 1.0
    S1 = m.calc_jacobian_batch(P[1:], t_out)                 # plain DOPRI45: the very same numbers
    assert np.array_equal(S[1], S1[0])
    # the explicit attempt gives up early on the stiff vector (negative max_steps: budget with early exit) instead of
    # burning its budget, and is untouched on the mild one
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        Yq = m.simulate_batch(P, t_out, method='dopri45', max_steps=-20000)
        early = m.last_info['n_steps'].copy()
        st_early = m.last_info['status'].copy()
        m.simulate_batch(P[:1], t_out, method='dopri45', max_steps=20000)
        full = m.last_info['n_steps'].copy()
    assert st_early.tolist() == [1, 0] and early[0] <= 2048 and full[0] >= 15000
    assert np.array_equal(Yq[1], m.simulate_batch(P[1:], t_out)[0])
    # state-only path, and a model that is not stiff at all: nothing switches
    Ys = m.simulate_batch(P, t_out, method='auto', max_steps=20000)
    assert m.last_info['stiff'].tolist() == [True, False] and parity_err(Ys[0, 1:], g['Y'][0]) <= 1.0
    c = gpu_models('cascade20')
    from sysbio_modeling_amd import models_zoo
    _, Pc = models_zoo.cascade_ensemble(8)
    tc = np.linspace(0, 100, 5)
    Sa = c.calc_jacobian_batch(Pc, tc, method='auto')
    assert not c.last_info['stiff'].any()
    assert np.array_equal(Sa, c.calc_jacobian_batch(Pc, tc))


def test_project_auto_method_on_the_stiff_model(gpu_models, golden):
    """Project.evaluate_batch(method='auto'): residual rows and the model Jacobian of a stiff vector come from
    the controlled implicit integrator, those of a mild vector from DOPRI45, in one call."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    m = gpu_models('stiff50')
    g = golden('stiff50_ref.npz')
    P, Yr, Sr = g['P'], g['Y'], g['S'].reshape(3, 16, 50, 50)
    species = (0, 10, 49)
    ms = [TimecourseMeasurement('s%02d' % v, Yr[0][:, v] * 1.3, models_zoo.STIFF_MEASURE_TIMES.copy(),
                                0.05 * np.abs(Yr[0][:, v]) + 0.01) for v in species]
    exp = Experiment('E', ms, fixed_parameters={'b%d' % i: 0.5 for i in range(50)})
    proj = Project(m, [exp], {'Global': ['a%d' % i for i in range(50)], 'Fixed': ['b%d' % i for i in range(50)]},
                   {('s%02d' % v): ('direct', v) for v in species}, reference_compat=False)
    names = list(m.param_order)
    cols = [proj.get_param_index('a%d' % j, 'Global') for j in range(50)]
    theta = np.zeros((2, 50))
    theta[0, cols] = np.log(P[0, [names.index('a%d' % j) for j in range(50)]])
    theta[1, cols] = np.log(10.0) * np.linspace(0.0, 1.0, 50)
    out = proj.evaluate_batch(theta, jacobian=True, want=('jacobian', 'model_jacobian'), method='auto',
                              max_steps=20000)
    assert out['status'].tolist() == [0, 0] and out['stiff'].tolist() == [True, False]
    sims_ref = np.concatenate([Yr[0][:, v] for v in species])
    assert parity_err(out['sims'][0], sims_ref) <= 1.0
    Jm_ref = np.concatenate([Sr[0][:, v, :] for v in species]) * P[0][None, :50]
    assert parity_err(out['model_jacobian'][0][:, cols], Jm_ref) <= 1.0
    plain = proj.evaluate_batch(theta[1:], jacobian=True, want=('jacobian', 'model_jacobian'))
    assert np.array_equal(out['jacobian'][1], plain['jacobian'][0])
    assert np.array_equal(out['residuals'][1], plain['residuals'][0])
    # DEFAULT options: the single-vector methods run 'auto' by themselves (what leastsq sees when a trial point is stiff)
    r_def = proj.residuals(theta[0])
    J_def = proj.calc_project_jacobian(theta[0])
    assert np.all(np.isfinite(r_def)) and np.all(np.isfinite(J_def)) and r_def.shape == (48,) and J_def.shape == (48, 50)
    # the single-vector API of the reference takes the method from the project's options
    proj.integrator_options.update(method='auto', max_steps=20000)
    r = proj.residuals(theta[0])
    # (a state-only run of the controlled implicit integrator: its own step count, the same solution to the parity
    # tolerance, propagated to the residual rows)
    a = proj.descriptor_arrays()
    tau_s, _ = lsoda_taus(a, theta[0], out['sims'][0])
    tol = project_tolerances(a, out['sims'][0], np.zeros(0), tau_s)
    assert tol_ratio(r, out['residuals'][0], tol['residuals']) <= 1.0
    assert tol_ratio(r_def, out['residuals'][0], tol['residuals']) <= 1.0
    assert np.allclose(J_def, out['jacobian'][0], rtol=1e-6, atol=1e-8 * np.abs(J_def).max())


STIFF_MOTIF = """
#*! Parameters Start
    k_on = p[0]
    k_off = p[1]
    k_cat = p[2]
    k_deg = p[3]
#*! Parameters End
#*! Variables Start
    _c = y[0]
    _p = y[1]
#*! Variables End
#*! Differential Equations Start
    d__c = k_on * (1.5 - _c) * (1.0 / (1.0 + _p)) - k_off * _c - k_cat * _c
    d__p = k_cat * _c - k_deg * _p
#*! Differential Equations End
"""


def test_graded_first_step_for_inconsistent_initial_conditions():
    """The reference always starts from y = 0 (ode_model.py:151-152); with binding rates of 2e3 that is far
    off the fast manifold, and the initial layer (width ~3e-4) is thinner than any affordable fixed step.
    SBM_IMPLICIT_MIDPOINT_GRADED cuts the first step into 13 geometrically growing substeps (nested across
    step_mult): scheme-level agreement with the oracle, two orders of magnitude closer to the reference's LSODA
    result than the plain rule at the same cost (13 extra steps), and with the step count left to
    method='implicit_controlled' the 1e-8 parity tolerance is met."""
    from oracle import imid_oracle, odeint_oracle as oo
    from sysbio_modeling_amd.symbolic import make_ode_model
    from sysbio_modeling_amd.model import OdeModel
    gm = make_ode_model(STIFF_MOTIF, name='stiff_motif2')
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff_motif2')
    P = np.array([[2e3, 5e2, 1e3, 0.1], [1e3, 8e2, 2e3, 0.15]])
    grid = np.linspace(0, 30.0, 1000)
    idx = np.array([100, 500, 999])
    t_out = _from_zero(grid[idx])
    Yr = np.stack([oo.simulate(gm, p, grid)[idx] for p in P])
    Sr = np.stack([oo.calc_jacobian(gm, p, grid)[idx] for p in P])
    # scheme level
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_midpoint_graded', h0=0.05,
                                 rtol=1e-11, atol=1e-13)
    Yo, So, ns, _ = imid_oracle.integrate(gm, P[0], t_out[1:], 0.05, rtol=1e-11, atol=1e-13, graded=True)
    assert m.last_info['n_steps'][0] == ns
    assert np.allclose(Y[0, 1:], Yo, rtol=1e-9, atol=1e-12) and np.allclose(S[0, 1:], So, rtol=1e-8, atol=1e-10 * np.abs(So).max())
    # ... and with every step halved: two substeps of each size of the pattern
    S2, Y2 = m.calc_jacobian_batch(P[:1], t_out, return_states=True, method='implicit_midpoint_graded', h0=0.05,
                                   step_mult=2, rtol=1e-11, atol=1e-13)
    Yo2, So2, ns2, _ = imid_oracle.integrate(gm, P[0], t_out[1:], 0.05, rtol=1e-11, atol=1e-13, graded=True, step_mult=2)
    assert m.last_info['n_steps'][0] == ns2 == 2 * ns
    assert np.allclose(Y2[0, 1:], Yo2, rtol=1e-9, atol=1e-12) and np.allclose(S2[0, 1:], So2, rtol=1e-8, atol=1e-10 * np.abs(So2).max())
    # reference level
    def rel(A, B):
        return np.max(np.abs(A - B) / (np.abs(B) + 1e-6 * np.abs(B).max()))
    kw = dict(n_steps=4096, extrapolate=1, rtol=1e-11, atol=1e-13)
    S_g, Y_g = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_midpoint_graded', **kw)
    S_p, Y_p = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_midpoint', **kw)
    e_g = max(rel(Y_g[:, 1:], Yr), rel(S_g[:, 1:], Sr))
    e_p = max(rel(Y_p[:, 1:], Yr), rel(S_p[:, 1:], Sr))
    assert e_g < 2e-7, e_g
    assert e_p > 50.0 * e_g, (e_p, e_g)
    S_c, Y_c = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled')
    assert m.last_info['status'].tolist() == [0, 0]
    gm.c_library()
    for v in range(2):
        tight = lambda v=v: oo.tight_solution(gm, P[v], t_out, use_c=True)[1:]
        check_parity(np.concatenate([Y_c[v, 1:], S_c[v, 1:]], axis=1), np.concatenate([Yr[v], Sr[v]], axis=1), tight,
                     what='binding motif, vector %d' % v, criterion='parity')


def test_single_vector_methods_integrate_a_stiff_vector_under_default_options(gpu_models, golden):
    """The reference's LSODA integrates a stiff parameter vector like any other (it switches to BDF by itself,
    model/ode_model.py:122-123).  The reference-named single-vector methods -- what a serial optimiser calls -- do too:
    with DEFAULT options a stiff vector goes through method='auto' and comes back finite and at parity, not as NaN."""
    m = gpu_models('stiff50')
    g, gt = golden('stiff50_ref.npz'), golden('stiff50_tight.npz')
    assert m.integrator_options['method'] == 'dopri45'              # the default; nothing was configured
    t_out = _from_zero(g['t'][g['idx']])
    y = m.simulate(g['P'][1], t_out)
    assert m.last_info['status'].tolist() == [0] and m.last_info['stiff'].tolist() == [True]
    check_parity(y[1:], g['Y'][1], gt['Y'][1], what='simulate, stiff vector', criterion='parity')
    s = m.calc_jacobian(g['P'][1], t_out, np.zeros(50 + 2500))
    assert m.last_info['status'].tolist() == [0] and np.all(np.isfinite(s))
    check_parity(s[1:], g['S'][1], gt['S'][1], what='calc_jacobian, stiff vector', criterion='parity')
    # a mild vector of the same model stays on DOPRI45, bit for bit
    mild = g['P'][0].copy()
    mild[:50] = 10.0 ** np.linspace(0.0, 1.0, 50)
    y2 = m.simulate(mild, t_out)
    assert m.last_info['stiff'].tolist() == [False]
    assert np.array_equal(y2, m.simulate_batch(mild[None, :], t_out)[0])


# ----------------------------------------------------------------------------------------------------------------
# more state variables than a wavefront has lanes: state component i lives on lane i mod 64 (sbm_implicit_stepper.hpp)
# ----------------------------------------------------------------------------------------------------------------
_big = {}


def _big_model(kind, n):
    """(GeneratedModel, OdeModel) of an n-state stiff cascade ('stiff') or activation cascade ('cascade')"""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import GeneratedModel
    if (kind, n) not in _big:
        spec = (models_zoo.stiff_spec(n, name='stiff%d' % n) if kind == 'stiff'
                else models_zoo.cascade_spec(n, name='cascade%d' % n))
        gm = GeneratedModel(spec)
        _big[(kind, n)] = (gm, OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=spec.name))
    return _big[(kind, n)]


def test_eighty_state_stiff_cascade_equals_the_scheme_oracle():
    """stiff80 (80 states, 80 sensitivity columns: 6480 coupled ODEs, two rows per lane and two column chunks) through
    the fixed-step kernel against the dense numpy restatement of the scheme (oracle/imid_oracle.py)."""
    from oracle import imid_oracle
    from sysbio_modeling_amd import models_zoo
    gm, m = _big_model('stiff', 80)
    P = models_zoo.stiff_ensemble(2, n=80)[1]
    t_out = np.array([0.0, 3.7, 10.0])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, h0=0.02, **IM)
    assert m.last_info['status'].tolist() == [0, 0]
    for v in range(2):
        Yo, So, ns, nn = imid_oracle.integrate(gm, P[v], t_out[1:], 0.02)
        assert m.last_info['n_steps'][v] == ns
        assert np.allclose(Y[v, 1:], Yo, rtol=1e-9, atol=1e-12)
        assert np.allclose(S[v, 1:], So, rtol=1e-8, atol=1e-10 * np.abs(So).max())
    assert np.array_equal(m.simulate_batch(P, t_out, h0=0.02, **IM), Y)
    # feed-forward structure survives the two-rows-per-lane bookkeeping: d x_i / d a_j = 0 for j > i
    S4 = S.reshape(2, 3, 80, 80)
    up = np.triu_indices(80, k=1)
    assert np.all(S4[:, :, up[0], up[1]] == 0.0) and np.abs(S4[:, -1, np.arange(80), np.arange(80)]).min() > 0.0


def test_eighty_state_stiff_cascade_against_reference_golden(golden):
    """tests/golden/stiff80_ref.npz: the REAL reference OdeModel (LSODA on its BDF branch) on stiff80, 2 vectors
    (make_golden_stiff.py 80 2); stiff80_tight.npz: the oracle's LSODA call at rtol 1e-12 (make_golden_stiff_tight.py
    80), which arbitrates where LSODA at the reference's 1e-10 is itself about one tolerance unit off (it is, as on
    stiff50).  The in-kernel controlled integrator with default options, a fixed-step Richardson pair, and method='auto'
    (DOPRI45 gives up, the implicit kernel takes over) all meet the parity tolerance."""
    g, gt = golden('stiff80_ref.npz'), golden('stiff80_tight.npz')
    gm, m = _big_model('stiff', 80)
    P = g['P']
    t_out = _from_zero(g['t'][g['idx']])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled')
    info = m.last_info
    assert not info['status'].any()
    ey = check_parity(Y[:, 1:], g['Y'], gt['Y'], what='stiff80 states', criterion='parity')
    es = check_parity(S[:, 1:], g['S'], gt['S'], what='stiff80 sensitivities', criterion='parity')
    print("implicit_controlled on stiff80: %s coarse steps (+%s abandoned); error vs LSODA golden y %.2f S %.2f units,"
          % (info['n_steps'], info['n_rejected'], ey[0], es[0]), "vs tight:", ey[1], es[1])
    Sf, Yf = m.calc_jacobian_batch(P, t_out, return_states=True, n_steps=8192, extrapolate=1, **IM)
    assert not m.last_info['status'].any()
    eyf = check_parity(Yf[:, 1:], g['Y'], gt['Y'], what='stiff80 states, fixed pair', criterion='parity')
    esf = check_parity(Sf[:, 1:], g['S'], gt['S'], what='stiff80 sensitivities, fixed pair', criterion='parity')
    print("fixed 8192 + 16384 Richardson on stiff80: vs golden y %.2f S %.2f units," % (eyf[0], esf[0]), "vs tight:", eyf[1], esf[1])
    Sa = m.calc_jacobian_batch(P[:1], t_out, method='auto', max_steps=20000)
    assert m.last_info['stiff'].tolist() == [True]
    check_parity(Sa[:, 1:], g['S'][:1], gt['S'][:1], what='stiff80 through method=auto', criterion='parity')


def test_controlled_implicit_kernel_on_a_seventy_state_model_agrees_with_dopri45():
    """cascade70 (not stiff): the error-controlled implicit kernel and DOPRI45 solve the same problem -- state and all
    70 x 140 sensitivities -- to the parity tolerance.  The cascade is an AMPLIFIER (gain k/d = 5 - 10 per stage while a
    stage is far from saturation): the relative error of a component that is still tiny reaches every stage downstream,
    so the test has to be relative -- DOPRI45's default atol is 1e-18 for that reason (model/ode_model.py) and the
    implicit kernel, whose inherited atol is 1e-3 rtol (stiff models carry sensitivities that are numerical noise:
    _lib.implicit_adaptive_defaults), is given the same here.  With the default atol it is 240 parity units off on this
    model -- as the reference's LSODA at atol 1e-10 is (310 units)."""
    from sysbio_modeling_amd import models_zoo
    gm, m = _big_model('cascade', 70)
    rng = np.random.default_rng(70)
    P = models_zoo.cascade_nominal_params(70)[None, :] * np.exp(0.2 * rng.standard_normal((2, 140)))
    t_out = np.array([0.0, 10.0, 30.0, 60.0])
    Se, Ye = m.calc_jacobian_batch(P, t_out, return_states=True)
    assert not m.last_info['status'].any()
    Si, Yi = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled', rtol=1e-9, atol=1e-18)
    assert not m.last_info['status'].any()
    ey, es = parity_err(Yi[:, 1:], Ye[:, 1:]), parity_err(Si[:, 1:], Se[:, 1:])
    print("cascade70 implicit_controlled vs dopri45: y %.3f S %.3f units, %s macro steps" % (ey, es, m.last_info['n_steps']))
    assert ey <= 1.0 and es <= 1.0


def test_dense_coupled_stiff_network_row_distributed_lu():
    """dense20 (every species coupled to every other: J_y full, 400 non-zeros) with degradation rates spread over four
    decades: a stiff problem whose Newton matrix is dense.  The factorisation runs distributed over the row lanes
    (emit_implicit.py IM_DIST, sbm_implicit_stepper.hpp::factor_rows).  Scheme level: the fixed-step kernel equals the
    dense numpy restatement; reference level: the error-controlled kernel meets the parity tolerance against the
    oracle's LSODA call."""
    from oracle import imid_oracle
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(models_zoo.dense_spec(density=1.0))
    assert 'IM_DIST = true' in gm.hip_source
    gm.c_library()
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    rng = np.random.default_rng(12)
    d = 10.0 ** np.linspace(0.0, 4.0, 20)
    P = np.stack([np.concatenate([d * rng.uniform(0.5, 2.0, 20), d * rng.uniform(0.8, 1.25, 20)]) for _ in range(2)])
    t_out = np.array([0.0, 0.4, 2.0])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, h0=0.002, **IM)
    assert m.last_info['status'].tolist() == [0, 0]
    for v in range(2):
        Yo, So, ns, nn = imid_oracle.integrate(gm, P[v], t_out[1:], 0.002)
        assert m.last_info['n_steps'][v] == ns
        assert np.allclose(Y[v, 1:], Yo, rtol=1e-9, atol=1e-12)
        assert np.allclose(S[v, 1:], So, rtol=1e-8, atol=1e-10 * np.abs(So).max())
    # a step five times longer starts too far from the fast species' initial layer for Newton on the second vector: the
    # restatement gives up there too (same predictor, same iteration limit), and the kernel says which vector it was
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.simulate_batch(P, t_out, h0=0.01, **IM)
    assert m.last_info['status'].tolist() == [0, 4]
    with pytest.raises(RuntimeError):
        imid_oracle.integrate(gm, P[1], t_out[1:], 0.01)
    grid = np.linspace(0.0, 2.0, 1000)
    idx = np.array([200, 999])
    t2 = _from_zero(grid[idx])
    Sc, Yc = m.calc_jacobian_batch(P, t2, return_states=True, method='implicit_controlled')
    assert m.last_info['status'].tolist() == [0, 0]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.simulate_batch(P, t2, method='dopri45', max_steps=-20000)
    explicit_status = m.last_info['status'].tolist()
    for v in range(2):
        Sr, Yr = oo.calc_jacobian(gm, P[v], grid, use_c=True, return_states=True)
        tight = oo.tight_solution(gm, P[v], t2, use_c=True)[1:]
        check_parity(np.concatenate([Yc[v, 1:], Sc[v, 1:]], axis=1), np.concatenate([Yr[idx], Sr[idx]], axis=1), tight,
                     what='dense20 stiff vector %d' % v, criterion='parity')
    print("dense20 stiff: implicit_controlled %s coarse steps; DOPRI45 with a 20000-step budget: status %s"
          % (m.last_info['n_steps'], explicit_status))


def test_stiff_integrator_edge_cases(gpu_models, zoo):
    """SBM_IMPLICIT_EXTRAP at the edges of its input space: a single output at the initial time; a start time other than
    zero with non-zero initial state AND sensitivities (odeint semantics, model/ode_model.py:122,167); a parameter vector
    that cannot be integrated next to ones that can (NaN rows and a non-zero status for it alone); a step budget that
    runs out; every extrapolation order the entry point admits."""
    from oracle import odeint_oracle as oo
    m, gm = gpu_models('michaelis_menten'), zoo('michaelis_menten')
    p = rc.MM_PARAMS * np.array([40.0, 30.0, 5.0, 3.0, 20.0])
    P = np.stack([p, 0.5 * p, 2.0 * p])
    kw = dict(method='implicit_extrap')
    # one output row, at t0: the initial condition, zero steps
    S, Y = m.calc_jacobian_batch(P, np.array([0.0]), return_states=True, **kw)
    assert m.last_info['status'].tolist() == [0, 0, 0] and m.last_info['n_steps'].tolist() == [0, 0, 0]
    assert np.all(Y == 0.0) and np.all(S == 0.0) and S.shape == (3, 1, 10)
    # t_sim[0] = 7.5 is the time of the initial condition; y0 and S0 given
    rng = np.random.default_rng(2)
    y0 = np.concatenate([[0.4, 0.9], 0.1 * rng.standard_normal(10)])
    t2 = np.linspace(7.5, 60.0, 300)
    pick = [0, 40, 299]
    Sr, Yr = oo.calc_jacobian(gm, P[1], t2, init_conditions=y0, return_states=True)
    S2, Y2 = m.calc_jacobian_batch(P[1:2], t2[pick], init_conditions=y0, return_states=True, **kw)
    assert np.array_equal(Y2[0, 0], y0[:2]) and np.array_equal(S2[0, 0], y0[2:])
    assert parity_err(Y2[0], Yr[pick]) <= 1.0 and parity_err(S2[0], Sr[pick]) <= 1.0
    # a vector that cannot be integrated does not take its neighbours with it
    Pbad = P.copy()
    Pbad[1, 0] = np.nan
    t3 = np.array([0.0, 10.0, 50.0])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        S3, Y3 = m.calc_jacobian_batch(Pbad, t3, return_states=True, **kw)
        st = m.last_info['status'].copy()
        Sg, Yg = m.calc_jacobian_batch(P, t3, return_states=True, **kw)
    assert st[0] == 0 and st[2] == 0 and st[1] != 0
    assert np.all(np.isnan(Y3[1, 1:])) and np.all(np.isnan(S3[1, 1:]))
    assert np.array_equal(Y3[[0, 2]], Yg[[0, 2]]) and np.array_equal(S3[[0, 2]], Sg[[0, 2]])
    # the step budget counts macro steps
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        m.calc_jacobian_batch(P[:1], t3, max_steps=3, **kw)
    assert m.last_info['status'].tolist() == [1]
    # every order: higher orders take fewer macro steps at a tight tolerance; all agree with the reference
    tt = np.linspace(0, 50.0, 1000)
    Srr, Yrr = oo.calc_jacobian(gm, P[0], tt, return_states=True)
    steps = []
    for K in (2, 3, 4, 6, 8, 10):
        So, Yo = m.calc_jacobian_batch(P[:1], tt[[0, 400, 999]], return_states=True, rtol=1e-8 if K > 3 else 1e-6, atol=1e-12, order=K, **kw)
        assert m.last_info['status'].tolist() == [0]
        steps.append(int(m.last_info['n_steps'][0]))
        tol = 1.0 if K > 3 else 300.0
        assert parity_err(Yo[0], Yrr[[0, 400, 999]]) <= tol and parity_err(So[0], Srr[[0, 400, 999]]) <= tol, K
    assert steps[3] > steps[4] and steps[2] > steps[3], steps


@pytest.mark.gpu
def test_stiff_integrator_with_a_vanishing_atol(gpu_models):
    """The controllers scale their errors in single precision: with atol below the smallest float (1e-40) the scale of a
    component that is exactly zero -- every sensitivity at t0 -- used to become 1 / 0: every macro step was rejected and
    the trajectory ended in SBM_STEP_UNDERFLOW after 200 000 attempts.  The scale is clamped in double precision: both
    kernels (a chain model: sbm_iex_seq_kernel; another: sbm_iex_kernel) integrate, and agree with the run at an ordinary
    atol within the tolerance."""
    cases = (('simple', np.array([[0.001, 0.01], [0.01, 0.01]]), np.array([0.0, 10.0, 50.0, 100.0])),
             ('michaelis_menten', np.stack([rc.MM_PARAMS * np.array([40.0, 30.0, 5.0, 3.0, 20.0])] * 2), np.array([0.0, 10.0, 50.0])))
    for name, P, tt in cases:
        m = gpu_models(name)
        S0, Y0 = m.calc_jacobian_batch(P, tt, return_states=True, method='implicit_extrap', rtol=1e-7, atol=1e-13)
        assert not m.last_info['status'].any()
        S1, Y1 = m.calc_jacobian_batch(P, tt, return_states=True, method='implicit_extrap', rtol=1e-7, atol=1e-40)
        assert not m.last_info['status'].any(), (name, m.last_info)
        assert np.isfinite(S1).all() and np.isfinite(Y1).all()
        assert np.max(np.abs(Y1 - Y0) / (1e-6 * np.abs(Y0) + 1e-12)) <= 1.0, name
        assert np.max(np.abs(S1 - S0) / (1e-6 * np.maximum(np.abs(S0), 1e-6 * np.abs(S0).max()) + 1e-12)) <= 1.0, name


@pytest.mark.gpu
def test_dense_stiff_network_against_reference_golden(golden):
    """A DENSE stiff network at the size of BASELINE configs[4] (models_zoo.dense_stiff_spec: 48 states, 1200 non-zeros of
    df/dy, degradation rates over four decades, 48 sensitivity columns = 2352 ODEs) against the REAL reference OdeModel
    (tests/golden/make_golden_dense_stiff.py), with LSODA at rtol 1e-12 by column groups as the tight solution.  The Newton
    matrix is dense: factored row-distributed over the lanes, substituted per column on the VALU (there are no MFMA tiles for
    the implicit path: DESIGN.md section 9).  Default options of the stiff integrator, one call."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import GeneratedModel
    g, gt = golden('dstiff48_ref.npz'), golden('dstiff48_tight.npz')
    gm = GeneratedModel(models_zoo.dense_stiff_spec())
    assert 'IM_DIST = true' in gm.hip_source and gm.n_vars == 48 and gm.n_sens == 48
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    assert np.array_equal(models_zoo.dense_stiff_ensemble(4096)[1][:len(g['P'])], g['P'])
    t_out = _from_zero(g['t'][g['idx']])
    S, Y = m.calc_jacobian_batch(g['P'], t_out, return_states=True, method='implicit_controlled')
    assert not m.last_info['status'].any()
    for v in range(len(g['P'])):
        ey = check_parity(Y[v, 1:], g['Y'][v], gt['Y'][v], what='dstiff48 vector %d, states' % v, criterion='parity')
        es = check_parity(S[v, 1:], g['S'][v], gt['S'][v], what='dstiff48 vector %d, sensitivities' % v, criterion='parity')
        print("dstiff48 vector %d: vs reference y %.2f S %.2f units%s; macro steps %d (+%d rejected)"
              % (v, ey[0], es[0], '' if es[1] is None else ' (vs tight %.2f)' % es[1], m.last_info['n_steps'][v],
                 m.last_info['n_rejected'][v]))


@pytest.mark.gpu
def test_stiff_chain_restarted_from_given_sensitivities(gpu_models, golden):
    """A chain model normally integrates with its sensitivity columns held rotated (sbm_iex_seq_kernel<M, true>: rows above a
    column's J_p entry are structurally zero).  Initial sensitivities handed in by the caller need not respect that
    structure, so such a call runs the un-rotated variant of the kernel (generated column step with inverse-ballot picks).
    Both on stiff50: 0 -> t2 in one call (rotated) against 0 -> t1, then t1 -> t2 from (y, S)(t1) (un-rotated) -- the same
    solution within the integration tolerance, and within parity of the reference golden at t2."""
    m = gpu_models('stiff50')
    g = golden('stiff50_ref.npz')
    P = g['P'][:2]
    tg = g['t'][g['idx']]
    t1, t2 = float(tg[5]), float(tg[-1])
    kw = dict(method='implicit_controlled')
    S2, Y2 = m.calc_jacobian_batch(P, np.array([0.0, t1, t2]), return_states=True, **kw)
    assert not m.last_info['status'].any()
    for v in range(2):
        y0 = np.concatenate([Y2[v, 1], S2[v, 1].ravel()])
        Sb, Yb = m.calc_jacobian_batch(P[v:v + 1], np.array([t1, t2]), init_conditions=y0, return_states=True, **kw)
        assert not m.last_info['status'].any()
        assert np.array_equal(Yb[0, 0], Y2[v, 1]) and np.array_equal(Sb[0, 0].ravel(), S2[v, 1].ravel())
        ey = np.max(np.abs(Yb[0, 1] - Y2[v, 2]) / (1e-8 * np.abs(Y2[v, 2]) + 5e-9))
        es = np.max(np.abs(Sb[0, 1] - S2[v, 2]) / (1e-8 * np.abs(S2[v, 2]) + 5e-9))
        print("stiff50 vector %d restarted at t = %.3g: y %.2f S %.2f parity units from the one-call solution" % (v, t1, ey, es))
        assert ey <= 1.0 and es <= 1.0
        assert parity_err(Yb[0, 1], g['Y'][v][-1]) <= 1.5 and parity_err(Sb[0, 1].ravel(), g['S'][v][-1]) <= 2.5


_SEQ_VS_OLD = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.symbolic import GeneratedModel
out = {}
for n in (24, 34):
    gm = GeneratedModel(models_zoo.stiff_spec(n, name='stiff%d_free' % n, fixed_deactivation=False))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    P = models_zoo.stiff_ensemble(4, n=n)[1][:3]
    t = np.array([0.0, 1.0, 4.0, 10.0])
    S, Y = m.calc_jacobian_batch(P, t, return_states=True, method='implicit_controlled')
    out['Y%d' % n], out['S%d' % n] = Y, S
    out['st%d' % n], out['ns%d' % n] = m.last_info['status'], m.last_info['n_steps']
np.savez(sys.argv[2], **out)
'''


@pytest.mark.gpu
def test_general_seq_kernel_equals_round3_kernel_with_two_entries_per_row_and_two_chunks(tmp_path):
    """Chains whose rows have TWO J_p entries (rate a_i and deactivation strength b_i both free: 2 n sensitivity columns) do
    not qualify for rotated columns: sbm_iex_seq_kernel<M, false> walks them with the generated column step (inverse-ballot
    picks, RL_MAXJP = 2) -- with 24 states in one chunk of 48 columns, with 34 states in TWO chunks (68 columns: two
    workgroups per trajectory, each redoing the state).  Same scheme as round 3's sbm_iex_kernel (SBM_IEX_SEQ=0, a second
    process: the switch is read once): the two agree far inside the integration tolerance and take the same steps."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'seq_vs_old.py'
    script.write_text(_SEQ_VS_OLD)
    res = {}
    for tag, env in (('seq', {}), ('old', {'SBM_IEX_SEQ': '0'})):
        p = subprocess.run([sys.executable, str(script), repo, str(tmp_path / (tag + '.npz'))], env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = np.load(tmp_path / (tag + '.npz'))
    for n in (24, 34):
        a, b = res['seq'], res['old']
        assert not a['st%d' % n].any() and not b['st%d' % n].any()
        assert np.max(np.abs(a['ns%d' % n].astype(int) - b['ns%d' % n].astype(int))) <= 2, (a['ns%d' % n], b['ns%d' % n])
        ey = np.max(np.abs(a['Y%d' % n] - b['Y%d' % n]) / (1e-9 * np.abs(b['Y%d' % n]) + 3e-13))
        Sb = b['S%d' % n]
        es = np.max(np.abs(a['S%d' % n] - Sb) / (1e-9 * np.maximum(np.abs(Sb), 1e-6 * np.abs(Sb).max()) + 3e-13))
        print("stiff%d_free (%d columns): seq vs round-3 kernel y %.3g S %.3g integration tolerances; macro steps %s / %s"
              % (n, 2 * n, ey, es, a['ns%d' % n], b['ns%d' % n]))
        # two runs of one scheme whose sums are associated differently (T_j against T_j - S_n): each within the integration
        # tolerance of the solution, so within a few tolerances of each other -- and far inside the parity tolerance
        assert ey <= 3.0 and es <= 3.0
        assert parity_err(a['Y%d' % n], b['Y%d' % n]) <= 0.3 and parity_err(a['S%d' % n], Sb) <= 0.3


def _tight_stiff_worker(args):
    """one worker of the pool below: LSODA at rtol 1e-12 by column groups for its share of the vectors"""
    P, t = args
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd.symbolic import zoo_model
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        import contextlib
        threadpool_limits = lambda limits=None: contextlib.nullcontext()     # noqa: E731
    gm = zoo_model('stiff50')
    gm.c_library()
    with threadpool_limits(limits=1):        # (a pool of processes with a BLAS thread pool each thrashes the box)
        return [oo.tight_stiff_solution_by_columns(gm, p, t, group=10) for p in P]


@pytest.mark.gpu
def test_stiff_defaults_on_vectors_that_never_took_part_in_choosing_them(gpu_models, golden):
    """The stiff integrator's default tolerances (rtol 1e-9, atol 3e-4 rtol) were settled on the 35 pinned vectors of the
    stiff50 ensemble.  Eight OTHER vectors, drawn by a fixed seed from the 4061 that were never looked at, each with a tight
    solution computed here (LSODA at rtol 1e-12 by column groups, ~20 s of a core each, a pool of workers): the same default
    call is within the parity tolerance of every one of them -- the defaults are not a fit to the pinned sample."""
    import multiprocessing as mp
    from sysbio_modeling_amd import models_zoo
    m = gpu_models('stiff50')
    g = golden('stiff50_wide_ref.npz')
    pinned = set(int(i) for i in g['index']) | {0, 1, 2}
    rng = np.random.default_rng(424242)
    fresh = [int(i) for i in rng.permutation(4096) if int(i) not in pinned][:8]
    _, P = models_zoo.stiff_ensemble(4096)
    t_out = _from_zero(g['t'][g['idx']])
    S, Y = m.calc_jacobian_batch(P[fresh], t_out, return_states=True, method='implicit_controlled')
    assert not m.last_info['status'].any()
    workers = min(8, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 4)
    shares = [fresh[i::workers] for i in range(workers)]
    with mp.get_context('spawn').Pool(workers) as pool:
        res = pool.map(_tight_stiff_worker, [(P[sh], t_out) for sh in shares if sh])
    tight = {}
    for sh, rows in zip([s_ for s_ in shares if s_], res):
        for v, (Yt, St) in zip(sh, rows):
            tight[v] = (Yt[1:], St[1:])
    worst = [0.0, 0.0]
    for k, v in enumerate(fresh):
        ey, es = parity_err(Y[k, 1:], tight[v][0]), parity_err(S[k, 1:].reshape(len(t_out) - 1, -1), tight[v][1])
        worst = [max(worst[0], ey), max(worst[1], es)]
    print("stiff50, 8 vectors outside the pinned 35 (%s): worst error vs tight y %.2f S %.2f parity units" % (fresh, worst[0], worst[1]))
    assert worst[0] <= 1.0 and worst[1] <= 1.0
