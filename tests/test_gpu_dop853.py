"""DOP853 (SBM_DOP853, method='dop853'): the eighth-order Dormand-Prince pair as an alternative to DOPRI45 at tight
tolerances -- same parity bar, a fraction of the steps.  The reference has one integrator (odeint = LSODA,
model/ode_model.py:122-123,167-168); which explicit pair stands in for it on the GPU is this build's choice, so the
checks are the ones DOPRI45 passes: the real reference's golden vectors, tight solutions under SURVEY 8(d), the project
rows against the assembly oracle."""
import multiprocessing as mp

import numpy as np
import pytest

from tests.conftest import parity_err, survey_err, check_parity, tol_ratio, lsoda_taus, project_tolerances

pytestmark = pytest.mark.gpu


def _from_zero(t_pts):
    return np.concatenate([[0.0], np.asarray(t_pts, dtype=float)])


def test_cascade20_golden_vectors(gpu_models, golden, zoo):
    """tests/golden/cascade20_ref.npz (the real reference OdeModel): states and all 800 sensitivities at the sampled grid
    points, default tolerances; a seventh of DOPRI45's steps."""
    from oracle import odeint_oracle as oo
    m = gpu_models('cascade20')
    g = golden('cascade20_ref.npz')
    P = g['P']
    t_out = _from_zero(g['t'][g['idx']])
    S8, Y8 = m.calc_jacobian_batch(P, t_out, return_states=True, method='dop853')
    info8 = {k: np.array(v, copy=True) for k, v in m.last_info.items() if k in ('status', 'n_steps')}
    assert not info8['status'].any()
    S5, Y5 = m.calc_jacobian_batch(P, t_out, return_states=True)
    assert np.all(info8['n_steps'] * 5 < m.last_info['n_steps'])
    gm = zoo('cascade20')
    for v in range(len(P)):
        tight = lambda v=v: oo.tight_solution(gm, P[v], t_out, use_c=True, atol=1e-30)[1:]
        check_parity(np.concatenate([Y8[v, 1:], S8[v, 1:]], axis=1), np.concatenate([g['Y'][v], g['S'][v]], axis=1), tight,
                     what='cascade20 golden vector %d' % v)
    assert parity_err(Y8, Y5) <= 1.0 and parity_err(S8, S5) <= 1.0
    # the state-only entry points: one trajectory per wavefront, and packed from 2048 trajectories on
    assert parity_err(m.simulate_batch(P, t_out, method='dop853')[:, 1:], g['Y']) <= 1.0
    from sysbio_modeling_amd import models_zoo
    _, Pb = models_zoo.cascade_ensemble(2048)
    Yb8 = m.simulate_batch(Pb, t_out, method='dop853')
    assert not m.last_info['status'].any()
    Yb5 = m.simulate_batch(Pb, t_out)
    # the vectors on which the two pairs differ most, against LSODA and (where needed) a tight solution
    diff = np.array([parity_err(Yb8[v], Yb5[v]) for v in range(len(Pb))])
    print("state only, 2048 vectors: DOP853 vs DOPRI45 median %.2f max %.2f parity units" % (np.median(diff), diff.max()))
    for v in np.argsort(diff)[-3:]:
        ref = oo.simulate(gm, Pb[v], g['t'], use_c=True)[g['idx']]
        tight = lambda v=v: oo.tight_solution(gm, Pb[v], t_out, sens=False, use_c=True, atol=1e-30)[1:]
        check_parity(Yb8[v, 1:], ref, tight, what='state-only DOP853, vector %d' % v)
    # small batch: the single-vector methods take the latency split
    m.integrator_options['method'] = 'dop853'
    try:
        S1 = m.calc_jacobian(P[0], g['t'], np.zeros(820))
    finally:
        m.integrator_options['method'] = 'dopri45'
    assert parity_err(S1[g['idx']], g['S'][0]) <= 1.0 or survey_err(S1[g['idx']], S8[0, 1:]) <= 1.0


def test_reference_fixture_models(gpu_models, golden):
    """The reference's own fixtures (1 and 2 states): DOP853 on the row-lane kernel, against the real reference's
    outputs on its 1000-point grid."""
    for name, fixture in (('simple', 'simple_ref.npz'), ('michaelis_menten', 'mm_ref.npz')):
        m = gpu_models(name)
        g = golden(fixture)
        S, Y = m.calc_jacobian_batch(g['P'], g['t'], return_states=True, method='dop853')
        assert not m.last_info['status'].any()
        assert parity_err(Y, g['Y']) <= 1.0 and parity_err(S, g['S']) <= 1.0, name
        assert parity_err(m.simulate_batch(g['P'], g['t'], method='dop853'), g['Y']) <= 1.0, name


def _tight_worker(args):
    n, P, t_out = args
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel, zoo_model
    gm = zoo_model('cascade20') if n == 20 else GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
    return [oo.tight_solution(gm, p, t_out, use_c=True, atol=1e-30)[1:] for p in P]


@pytest.mark.parametrize('n,n_vec', [(40, 6), (70, 3)])
def test_default_options_meet_section_8d_on_deep_cascades(n, n_vec):
    """DOP853 with OdeModel's default (size-aware) tolerances against DOP853-in-SciPy tight solutions (rtol 1e-13) under
    SURVEY section 8(d), on the models where DOPRI45 needed the size-aware default."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
    gm.c_library()
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    rng = np.random.default_rng(2026)
    P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.3 * rng.standard_normal((n_vec, 2 * n)))
    grid = np.linspace(0, 60.0, 1000)
    t_out = _from_zero(grid[[100, 300, 600, 999]])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='dop853')
    steps8 = m.last_info['n_steps'].mean()
    assert not m.last_info['status'].any()
    m.calc_jacobian_batch(P, t_out)
    with mp.get_context('spawn').Pool(min(n_vec, 6)) as pool:
        tight = [x for part in pool.map(_tight_worker, [(n, P[i:i + 1], t_out) for i in range(n_vec)]) for x in part]
    ey = max(survey_err(Y[v, 1:], tight[v][:, :n]) for v in range(n_vec))
    es = max(survey_err(S[v, 1:], tight[v][:, n:]) for v in range(n_vec))
    print("cascade%d dop853 default: state %.2f sens %.2f section-8(d) units, %.0f steps per vector (dopri45: %.0f)"
          % (n, ey, es, steps8, m.last_info['n_steps'].mean()))
    assert ey <= 1.0 and es <= 1.0


def test_project_rows_with_dop853(gpu_models, zoo):
    """Residual rows, scale factors and Jacobian rows of a two-experiment cascade project integrated with DOP853, against
    the assembly oracle (LSODA-driven) within the propagated tolerances."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    from oracle import odeint_oracle as oo
    from oracle.project_oracle import ProjectOracle
    gm = zoo('cascade20')
    m = gpu_models('cascade20')
    grid = np.linspace(0, 100, 1000)
    idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)

    def exps():
        out = []
        for c in range(2):
            p = models_zoo.cascade_nominal_params()
            p[20] *= 1.0 + 0.5 * c
            y = oo.simulate(gm, p, grid, use_c=True)[idx]
            ms = [TimecourseMeasurement('s%d' % v, 2.0 * y[:, v], models_zoo.CASCADE_MEASURE_TIMES.copy(),
                                        0.05 * np.abs(y[:, v]) + 0.01) for v in (4, 19)]
            out.append(Experiment('exp_%d' % c, ms, experiment_settings={'cond': c}))
        return out
    settings = {'Shared': {'deg': {'d0': ('cond',)}}, 'Global': [n for n in gm.param_order if n != 'd0']}
    mapping = {'s4': ('direct', 4), 's19': ('direct', 19)}
    proj = Project(m, exps(), settings, mapping, sf_groups=['s4', 's19'], reference_compat=False)
    po = ProjectOracle(gm, exps(), settings, mapping, sf_groups=['s4', 's19'], reference_compat=False)
    theta = np.zeros(proj.n_project_params)
    for name, slots in proj.project_param_idx.items():
        for key, gi in slots.items():
            theta[gi] = np.log(0.1 if name == 'deg' else models_zoo.cascade_nominal_params()[gm.param_order.index(name)])
    thetas = theta[None, :] + 0.1 * np.random.default_rng(4).standard_normal((3, theta.size))
    out = proj.evaluate_batch(thetas, jacobian=True, want=('jacobian',), method='dop853')
    assert not out['status'].any()
    res_only = proj.evaluate_batch(thetas, method='dop853')
    a = proj.descriptor_arrays()
    for v in range(3):
        rr, sims, B = po.residuals(thetas[v], return_parts=True)
        Jr, Jm = po.calc_project_jacobian(thetas[v]), po.model_jacobian(thetas[v])
        tau_s, tau_Jm = lsoda_taus(a, thetas[v], sims, Jm)
        t = project_tolerances(a, sims, B, tau_s, Jm, tau_Jm)
        assert tol_ratio(out['residuals'][v], rr, t['residuals']) <= 1.0
        assert tol_ratio(out['sf'][v], B, t['sf']) <= 1.0
        assert tol_ratio(out['jacobian'][v], Jr, t['jacobian']) <= 1.0
        assert tol_ratio(res_only['residuals'][v], rr, t['residuals']) <= 1.0
    fit = proj.fit_batch(thetas, max_iter=5, method='dop853')
    assert np.all(fit['cost'] <= 0.5 * out['norms'] + 1e-12)


def test_single_vector_methods_keep_the_stiff_fallback(gpu_models, golden):
    """A model whose options name 'dop853' keeps what the single-vector methods do by default: method='auto', with the
    eighth-order pair as the explicit attempt -- a stiff vector (the stiff50 golden) still reaches the implicit
    integrator, a mild one carries the numbers of a plain DOP853 call."""
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.symbolic import zoo_model
    gm = zoo_model('stiff50')
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='stiff50')
    m.integrator_options['method'] = 'dop853'
    g = golden('stiff50_ref.npz')
    y = m.simulate(g['P'][0], g['t'])
    assert m.last_info['stiff'].tolist() == [True]
    assert parity_err(y[g['idx']], g['Y'][0]) <= 1.0
    mild = g['P'][0].copy()
    mild[:50] = 10.0 ** np.linspace(0.0, 1.0, 50)
    t_out = _from_zero(g['t'][g['idx']])
    y1 = m.simulate(mild, t_out)
    assert m.last_info['stiff'].tolist() == [False]
    assert np.array_equal(y1, m.simulate_batch(mild[None, :], t_out)[0])
    # and a batch call with the method spelled out: 'auto' with explicit_method
    Y2 = m.simulate_batch(np.stack([g['P'][0], mild]), t_out, method='auto', explicit_method='dop853', max_steps=20000)
    assert m.last_info['stiff'].tolist() == [True, False] and np.array_equal(Y2[1], y1)


def _lsoda_worker(args):
    P, grid, idx = args
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd.symbolic import zoo_model
    gm = zoo_model('cascade20')
    out = []
    for p in P:
        S, Y = oo.calc_jacobian(gm, p, grid, use_c=True, return_states=True)
        out.append(np.concatenate([Y[idx], S[idx]], axis=1))
    return out


def test_headline_ensemble_256_vectors_against_the_oracle(gpu_models, zoo):
    """256 vectors spread over the headline ensemble, every sampled state and sensitivity against the oracle's LSODA call
    (as tests/test_gpu_parity_sweeps.py does for DOPRI45 on 1024): where a vector exceeds the parity tolerance the
    disagreement must be LSODA's own error, shown against a tight solution."""
    from oracle import odeint_oracle as oo
    from sysbio_modeling_amd import models_zoo
    gm = zoo('cascade20')
    gm.c_library()
    m = gpu_models('cascade20')
    _, P = models_zoo.cascade_ensemble(4096)
    pick = np.arange(0, 4096, 16)
    grid = np.linspace(0, 100.0, 1000)
    idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)
    t_out = _from_zero(grid[idx])
    S, Y = m.calc_jacobian_batch(P[pick], t_out, return_states=True, method='dop853')
    assert not m.last_info['status'].any()
    chunks = np.array_split(np.arange(len(pick)), 12)
    with mp.get_context('spawn').Pool(12) as pool:
        ref = [x for part in pool.map(_lsoda_worker, [(P[pick[c]], grid, idx) for c in chunks]) for x in part]
    got = np.concatenate([Y[:, 1:], S[:, 1:]], axis=2)
    err = np.array([parity_err(got[v], ref[v]) for v in range(len(pick))])
    over = np.flatnonzero(err > 1.0)
    for v in over:
        tight = oo.tight_solution(gm, P[pick[v]], t_out, use_c=True, atol=1e-30)[1:]
        check_parity(got[v], ref[v], tight, what='vector %d' % pick[v])
    assert len(over) <= 8
    print("DOP853, 256 vectors: error against LSODA median %.3f p99 %.3f max %.3f parity units; %d beyond 1 (LSODA's own error)"
          % (np.median(err), np.percentile(err, 99), err.max(), len(over)))


@pytest.mark.parametrize('model,variant', [('cascade20', 'row_group'), ('cascade20', 'row_lane'), ('cascade20', 'per_wave'),
                                           ('cascade20', 'mfma'), ('michaelis_menten', 'packed'), ('michaelis_menten', 'row_lane'),
                                           ('michaelis_menten', 'per_wave'), ('simple', 'packed')])
def test_dop853_on_every_sensitivity_kernel_variant(gpu_models, golden, zoo, model, variant):
    """DOP853 answers on every sensitivity kernel (round 2: row-group and the smallest row-lane models only; the other
    variants returned an error): the real reference's golden vectors through each forced variant, and the same numbers
    as the method's default kernel to the integration tolerance (the variants take their own step sequences)."""
    m = gpu_models(model)
    g = golden({'cascade20': 'cascade20_ref.npz', 'michaelis_menten': 'mm_ref.npz', 'simple': 'simple_ref.npz'}[model])
    if model == 'cascade20':
        P, t_out, Yr, Sr = g['P'][:3], _from_zero(g['t'][g['idx']]), g['Y'][:3], g['S'][:3]
    else:
        P, t = g['P'], g['t']                       # fixture models on the reference's 1000-point grid (t[0] = 0)
        idx = np.arange(60, len(t), 90)
        t_out, Yr, Sr = _from_zero(t[idx]), g['Y'][:, idx], g['S'][:, idx]
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method='dop853', variant=variant)
    assert not m.last_info['status'].any()
    S0, Y0 = m.calc_jacobian_batch(P, t_out, return_states=True, method='dop853')
    ey, es = parity_err(Y[:, 1:], Yr), parity_err(S[:, 1:], Sr)
    print("%s dop853 variant %s: vs reference golden y %.3f S %.3f; vs the default kernel y %.3g S %.3g; steps %s"
          % (model, variant, ey, es, parity_err(Y, Y0), parity_err(S, S0), m.last_info['n_steps']))
    assert ey <= 1.0 and es <= 1.0
    assert parity_err(Y, Y0) <= 1.0 and parity_err(S, S0) <= 1.0
