"""The oracle is pinned before it is trusted (CPU only).

  * integration half: against the reference's golden vector and analytic sensitivities
    (tests/test_OdeModel.py:20-52) and against tests/golden/*.npz, which hold outputs of the
    REAL reference OdeModel (tests/golden/make_golden.py);
  * sampling: against outputs of the reference's project/utils.py helpers (sampling_ref.npz);
  * assembly half: against the known answers of tests/test_Project.py and
    tests/test_Loss_Functions.py.
"""
import numpy as np
import pytest

from oracle import odeint_oracle as oo
from oracle.project_oracle import ProjectOracle
from tests import reference_cases as rc
from sysbio_modeling_amd.experiment import Experiment
from sysbio_modeling_amd.measurement import TimecourseMeasurement


# ---------------------------------------------------------------------------
# integration half
# ---------------------------------------------------------------------------
def test_simulate_matches_reference_test_golden(zoo):
    gm = zoo('simple')
    y = oo.simulate(gm, rc.SIMPLE_P, rc.SIMPLE_T10)[:, 0]
    assert np.allclose(y, rc.SIMPLE_Y10_GOLDEN, rtol=0.05)          # the reference's own tolerance (:29)
    assert np.max(np.abs(y - rc.SIMPLE_Y10_GOLDEN)) < 5e-8          # and far tighter than that


def test_calc_jacobian_matches_closed_form(zoo):
    gm = zoo('simple')
    s = oo.calc_jacobian(gm, rc.SIMPLE_P, rc.SIMPLE_T10)
    _, d_kdeg, d_ksynt = rc.simple_closed_form(rc.SIMPLE_P[0], rc.SIMPLE_P[1], rc.SIMPLE_T10)
    assert np.allclose(s[:, 1], d_ksynt, rtol=0.05)                 # tests/test_OdeModel.py:45
    assert np.allclose(s[1:, 1], d_ksynt[1:], rtol=1e-8)
    assert np.allclose(s[1:, 0], d_kdeg[1:], rtol=1e-7)


@pytest.mark.parametrize('name,file', [('simple', 'simple_ref.npz'), ('michaelis_menten', 'mm_ref.npz')])
def test_oracle_equals_real_reference_odemodel(zoo, golden, name, file):
    """Same odeint call on a mathematically identical RHS: agreement to LSODA's round-off
    sensitivity (the reference fixture RHS and the generated RHS differ in operation order)."""
    gm = zoo(name)
    g = golden(file)
    for v, p in enumerate(g['P']):
        y = oo.simulate(gm, p, g['t'])
        s = oo.calc_jacobian(gm, p, g['t'])
        assert np.allclose(y, g['Y'][v], rtol=1e-9, atol=1e-10)
        assert np.allclose(s, g['S'][v], rtol=1e-8, atol=1e-9)


def test_oracle_c_rhs_equals_python_rhs(zoo, golden):
    gm = zoo('cascade20')
    g = golden('cascade20_ref.npz')
    for v in range(2):
        y = oo.simulate(gm, g['P'][v], g['t'], use_c=True)[g['idx']]
        s = oo.calc_jacobian(gm, g['P'][v], g['t'], use_c=True)[g['idx']]
        # golden = reference OdeModel driving the generated PYTHON rhs; here the compiled C rhs
        assert np.allclose(y, g['Y'][v], rtol=1e-9, atol=1e-10)
        assert np.allclose(s, g['S'][v], rtol=1e-8, atol=1e-9)


def test_generated_rhs_equals_reference_fixture_rhs(zoo, golden):
    """Point evaluations of the reference's fixture callbacks
    (tests/test_utils/sens_jittable_model.py, sens_jittable_mm_model.py) vs the generated ones."""
    for name, file in (('simple', 'simple_ref.npz'), ('michaelis_menten', 'mm_ref.npz')):
        gm = zoo(name)
        g = golden(file)
        n, k = gm.n_vars, gm.n_sens
        for y, p, ref_state, ref_sens in zip(g['rhs_y'], g['rhs_p'], g['rhs_state'], g['rhs_sens']):
            out = np.zeros(n)
            gm.model(y[:n].copy(), 0.0, out, p)
            assert np.allclose(out, ref_state, rtol=1e-12, atol=1e-15)
            out = np.zeros(n + n * k)
            gm.sens_model(y.copy(), 0.0, out, p)
            assert np.allclose(out, ref_sens, rtol=1e-10, atol=1e-13)


# ---------------------------------------------------------------------------
# sampling (project/utils.py:10-89)
# ---------------------------------------------------------------------------
def test_sampling_matches_reference_helpers(golden):
    g = golden('sampling_ref.npz')
    t, tm = g['t'], g['t_meas']
    tm = tm[tm != 0]
    idx = np.searchsorted(t, tm)
    assert np.array_equal(g['direct_t'], t[idx])
    assert np.array_equal(g['direct_sim'], g['sim'][idx, 1])
    assert np.array_equal(g['sum_sim'], g['sim'][idx, 0] + g['sim'][idx, 2])
    assert np.array_equal(g['direct_jac'], g['jac'][idx, 6:9])
    assert np.allclose(g['sum_jac'], g['jac'][idx, 0:3] + g['jac'][idx, 3:6], rtol=0, atol=0)
    # the quirk that matters: at-or-after grid point, no interpolation (t = 50 -> 50.05005)
    assert abs(g['direct_t'][1] - 50.05005005005005) < 1e-12


# ---------------------------------------------------------------------------
# assembly half: tests/test_Project.py
# ---------------------------------------------------------------------------
@pytest.fixture(scope='module')
def simple_oracle(zoo):
    exps, settings, mapping, sf = rc.simple_project_case()
    po = ProjectOracle(zoo('simple'), exps, settings, mapping, sf_groups=sf)
    theta = rc.simple_project_theta(lambda g, s: po.project_param_idx[g][s])
    return po, theta


def test_project_initialisation(simple_oracle):
    po, _ = simple_oracle
    assert list(po.project_param_idx['k_synt'].keys()) == ['Global']   # test_Project.py:89-90
    assert len(po.project_param_idx['Group_1']) == 2                   # :93-94
    assert po.n_project_params == 3
    assert len(po.rows()) == 35                                        # :104
    for ei, exp in enumerate(po.experiments):
        assert po.project_param_idx['Group_1'][(exp.settings['Deg_Rate'],)] == po.exp_param_idx[ei]['k_deg']


def test_project_scale_factor_recovered(simple_oracle):
    po, theta = simple_oracle
    res, sims, B = po.residuals(theta, return_parts=True)
    assert np.allclose(B[0], 3.75, rtol=0.05)                           # test_Project.py:114
    data = np.array([r[2] for r in po.rows()]) / 3.75
    assert np.allclose(data, sims, rtol=0.05)                           # :111
    assert res.shape == (35,)


def test_project_model_jacobian_analytic(simple_oracle):
    po, theta = simple_oracle
    J = po.model_jacobian(theta)
    p = np.exp(theta)
    row = 0
    for ei, exp in enumerate(po.experiments):
        t = exp.get_variable_measurements('Variable_1').timepoints
        t = t[t != 0]
        ks, kd = po.exp_param_idx[ei]['k_synt'], po.exp_param_idx[ei]['k_deg']
        anal = rc.simple_model_analytical_jac(p[kd], p[ks], t)
        blk = J[row:row + len(t)]
        assert np.allclose(anal[1], blk[:, ks], rtol=0.05)              # test_Project.py:141-142
        assert np.allclose(anal[0], blk[:, kd], rtol=0.05)
        row += len(t)


def test_project_jacobian_finite_differences(simple_oracle):
    po, theta = simple_oracle
    J = po.calc_project_jacobian(theta)

    def scaled_sims(x):
        _, sims, B = po.residuals(x, return_parts=True)
        return sims * B[0]
    num = rc.central_fd_jacobian(scaled_sims, theta)
    assert np.allclose(num, J, atol=1e-6)                               # test_Project.py:167
    th2 = theta.copy()
    th2[po.project_param_idx['Group_1'][('High',)]] = np.log(0.02)
    th2[po.project_param_idx['Group_1'][('Low',)]] = np.log(0.003)
    th2[po.project_param_idx['k_synt']['Global']] = np.log(0.05)
    g = po.calc_rss_gradient(th2)
    gnum = rc.central_fd_jacobian(lambda x: np.array([po.calc_sum_square_residuals(x)]), th2)[0]
    assert np.allclose(g, gnum, atol=1e-6)                              # :176


def test_project_sum_mapping_michaelis_menten(zoo):
    gm = zoo('michaelis_menten')
    sim = oo.simulate(gm, rc.MM_PARAMS, rc.MM_T)   # the reference generates its data with odeint too (:294)
    total = TimecourseMeasurement('Total', sim.sum(axis=1), rc.MM_T)
    exp = Experiment('Standard', total)
    po = ProjectOracle(gm, [exp], {}, {'Total': ('sum', [0, 1])})
    assert all(list(v.keys()) == ['Global'] for v in po.project_param_idx.values())   # :309-311
    theta = np.log(np.array([rc.MM_PARAMS[gm.param_order.index(p)] for p in po.project_param_idx]))
    res = po.residuals(theta)
    assert np.allclose(res, 0, atol=1e-3)                               # :322-323
    g = po.calc_rss_gradient(theta)
    gnum = rc.central_fd_jacobian(lambda x: np.array([po.calc_sum_square_residuals(x)]), theta)[0]
    assert np.allclose(g, gnum, atol=1e-5)                              # :330-332
    J = po.calc_project_jacobian(theta)
    Jnum = rc.central_fd_jacobian(lambda x: po.residuals(x, return_parts=True)[1], theta)
    assert np.allclose(J, Jnum, atol=1e-5)                              # :347-349


# ---------------------------------------------------------------------------
# assembly half: tests/test_Loss_Functions.py (hand-built rows, no integration)
# ---------------------------------------------------------------------------
from tests.loss_cases import RowsOnlyOracle as _RowsOnlyOracle, lin_square_rows as _lin_square_rows  # noqa: E402


def test_loss_residuals_without_scale_factors():
    rng = np.random.default_rng(0)
    noise = rng.standard_normal(202)
    rows, sims = _lin_square_rows(noise=noise)
    po = _RowsOnlyOracle(rows, sims, None, [], 1)
    assert np.allclose(po.residuals(np.zeros(1)), noise)                 # :44-45
    std = np.abs(rng.standard_normal(202)) + 0.1
    rows = [(r[0], r[1], r[2], s, r[4]) for r, s in zip(rows, std)]
    po = _RowsOnlyOracle(rows, sims, None, [], 1)
    d = np.array([r[2] for r in rows])
    assert np.allclose(po.residuals(np.zeros(1)), (sims - d) / std)      # :52-56


def test_loss_scale_factors_exact():
    rows, sims = _lin_square_rows(2.0, 3.6)
    po = _RowsOnlyOracle(rows, sims, None, ['Lin', 'Square'], 1)
    res, _, B = po.residuals(np.zeros(1), return_parts=True)
    assert B[0] == pytest.approx(2.0, abs=1e-14) and B[1] == pytest.approx(3.6, abs=1e-14)   # :69-70
    assert np.allclose(res, 0)                                           # :79-80


def test_loss_scale_factor_gradient_and_scaled_jacobian():
    t = np.linspace(0, 10, 11)                                           # :109
    p1 = np.array([0.3, 0.5, 1.3])

    def model_fcn(p):
        return p[0] * np.sin(p[1] * t) - p[2] * t ** 2

    jac = np.stack([np.sin(p1[1] * t), p1[0] * t * np.cos(p1[1] * t), -t ** 2], axis=1)   # :99-105
    rng = np.random.default_rng(1)
    data = model_fcn(p1) * 5 - rng.standard_normal(11)
    rows = [(0, 'Val', d, 1.0, tt) for d, tt in zip(data, t)]

    def sf_of(p):
        po = _RowsOnlyOracle(rows, model_fcn(p), None, ['Val'], 3)
        return np.array([po.residuals(np.zeros(3), return_parts=True)[2][0]])

    po = _RowsOnlyOracle(rows, model_fcn(p1), jac, ['Val'], 3)
    B, dB, *_ = po._sf(rows, model_fcn(p1), jac)
    assert np.allclose(rc.central_fd_jacobian(sf_of, p1)[0], dB[0], rtol=0.01)            # :144

    def res_of(p):
        return _RowsOnlyOracle(rows, model_fcn(p), None, ['Val'], 3).residuals(np.zeros(3))

    assert np.allclose(po.calc_project_jacobian(np.zeros(3)), rc.central_fd_jacobian(res_of, p1), rtol=0.01)  # :164
    # no scale factors: the Jacobian is returned unchanged (:118-121)
    po0 = _RowsOnlyOracle(rows, model_fcn(p1), jac, [], 3)
    assert np.array_equal(po0.calc_project_jacobian(np.zeros(3)), jac)


def test_loss_scale_factor_prior_residual():
    rows, sims = _lin_square_rows(5.0, 5.0)
    rows = [r for r, s in zip(rows, sims) if s != 0]                     # :200-202
    sims = sims[sims != 0]
    po = _RowsOnlyOracle(rows, sims, None, ['Lin', 'Square'], 1)
    po.sf_priors = {0: (1.0, 2.0)}                                       # :208
    res = po.residuals(np.zeros(1))
    assert np.allclose(res[-1], (np.log(5) - 1) / 2.0)                   # :232-233


# ---------------------------------------------------------------------------
# log-square and normalized losses (tests/test_Loss_Functions.py:166-194, 235-312)
# ---------------------------------------------------------------------------
def test_log_loss_residuals_identities():
    rows, sims = _lin_square_rows()
    keep = sims != 0                                                     # :170-172
    rows = [r for r, k in zip(rows, keep) if k]
    sims = sims[keep]
    rows = [(r[0], r[1], r[2] / 2.3, r[3], r[4]) for r in rows]           # :176-177
    po = _RowsOnlyOracle(rows, sims, None, [], 1)
    po.loss = 'log'
    assert np.allclose(po.residuals(np.zeros(1)), np.log(2.3))           # :182-184
    po = _RowsOnlyOracle(rows, sims, None, ['Lin', 'Square'], 1)
    po.loss = 'log'
    res, _, B = po.residuals(np.zeros(1), return_parts=True)
    assert np.allclose(res, 0) and np.allclose(B, 1 / 2.3)               # :190-194


def test_log_loss_jacobian_identities():
    t = np.linspace(0, 11, 11)                                           # :253
    p1 = np.array([0.3, 0.5, 1.3])

    def model_fcn(p):
        return p[0] + t * p[2] * p[1] ** 2                               # :238-241

    jac = np.stack([np.ones_like(t), 2 * t * p1[2] * p1[1], t * p1[1] ** 2], axis=1)     # :243-249
    rng = np.random.default_rng(4)
    data = model_fcn(p1) * 5 + np.abs(rng.standard_normal(11))           # :267-269
    rows = [(0, 'Val', d, 1.0, tt) for d, tt in zip(data, t)]

    def mk(p, J=None, groups=('Val',)):
        o = _RowsOnlyOracle(rows, model_fcn(p), J, list(groups), 3)
        o.loss = 'log'
        return o

    B, dB, *_ = mk(p1, jac)._sf(rows, model_fcn(p1), jac)
    num = rc.central_fd_jacobian(lambda p: np.array([mk(p).residuals(np.zeros(3), return_parts=True)[2][0]]), p1)[0]
    assert np.allclose(num, dB[0], rtol=0.01)                            # :281-284
    Jnum = rc.central_fd_jacobian(lambda p: mk(p).residuals(np.zeros(3)), p1)
    assert np.allclose(mk(p1, jac).calc_project_jacobian(np.zeros(3)), Jnum, rtol=0.01)   # :295
    # no scale factors: J / sim (:67-69)
    assert np.allclose(mk(p1, jac, groups=()).calc_project_jacobian(np.zeros(3)), jac / model_fcn(p1)[:, None])


def test_normalized_loss_identity():
    """Normalized residuals x measurement mean == plain residuals (:297-312)."""
    rng = np.random.default_rng(6)
    noise = rng.standard_normal(202)
    rows, sims = _lin_square_rows(noise=noise)
    rows = [r for r in rows if r[2] != 0]
    sims_nz = np.array([s for s, r in zip(sims, _lin_square_rows(noise=noise)[0]) if r[2] != 0])
    plain = _RowsOnlyOracle(rows, sims_nz, None, [], 1).residuals(np.zeros(1))
    # the oracle applies sigma *= mean when it builds rows(); emulate that on the hand-built rows
    nrows = [(r[0], r[1], r[2], r[3] * r[2], r[4]) for r in rows]
    norm = _RowsOnlyOracle(nrows, sims_nz, None, [], 1).residuals(np.zeros(1))
    d = np.array([r[2] for r in rows])
    assert np.allclose(norm * d, plain)


def _log_sf_cases(golden):
    g = golden('log_scale_factor_ref.npz')
    for c in range(int(g['n_cases'])):
        yield c, {k: g['%s_%d' % (k, c)] for k in ('sim', 'data', 'std', 'jac', 'prior', 'sf', 'sf_gradient',
                                                    'prior_residual', 'prior_gradient')}


def test_log_scale_factor_restatement_equals_the_real_reference_class(golden):
    """The one class of the assembly half that runs here as it stands -- the reference's LogScaleFactor (needs numpy
    only; tests/golden/make_golden_log_scale_factor.py) -- against the oracle's restatement of it
    (project_oracle.py::_sf, log branch; the prior row of residuals / calc_project_jacobian): scale factor, its gradient,
    the prior residual and the prior's Jacobian row to rounding, on six random cases."""
    for c, g in _log_sf_cases(golden):
        n, q = g['jac'].shape
        rows = [(0, 'M', d, s, float(i)) for i, (d, s) in enumerate(zip(g['data'], g['std']))]
        po = _RowsOnlyOracle(rows, g['sim'], g['jac'], ['M'], q)
        po.loss = 'log'
        B, dB, _, _, _ = po._sf(rows, g['sim'], g['jac'])
        assert B[0] == pytest.approx(float(g['sf']), rel=1e-13), c
        assert np.allclose(dB[0], g['sf_gradient'], rtol=1e-12, atol=1e-15 * np.abs(g['sf_gradient']).max()), c
        if np.isfinite(g['prior'][0]):
            po.sf_priors = {0: (float(g['prior'][0]), float(g['prior'][1]))}
            res = po.residuals(np.zeros(q))
            assert res[-1] == pytest.approx(float(g['prior_residual']), rel=1e-12, abs=1e-14), c
            Jp = po.calc_project_jacobian(np.zeros(q))
            assert np.allclose(Jp[-1], g['prior_gradient'], rtol=1e-12, atol=1e-15), c      # reference_compat row: (dB/dtheta)/B
            # the data rows of the log loss: J / sim + (dB/dtheta) / B   (log_squared_loss_function.py:92-93)
            assert np.allclose(Jp[:-1], g['jac'] / g['sim'][:, None] + (g['sf_gradient'] / float(g['sf']))[None, :], rtol=1e-12)
        else:
            assert np.isnan(g['prior_residual'])


def test_experiment_and_measurement_containers_equal_the_real_reference_classes(golden):
    """experiment/experiments.py and measurement/timecourse_measurement.py run here as they stand
    (tests/golden/make_golden_containers.py recorded what the reference's objects answer on random measurement sets); this
    package's Experiment / TimecourseMeasurement on the same inputs: the order the measurements are kept in, the default
    error bars, get_unique_timepoints with and without zero, get_nonzero_measurements, drop_timepoint_zero of one variable
    and of all."""
    g = golden('containers_ref.npz')
    for c in range(int(g['n_cases'])):
        names = [str(x) for x in g['names_%d' % c]]
        ms = []
        for j, nm in enumerate(names):
            s = g['s_%d_%d' % (c, j)]
            ms.append(TimecourseMeasurement(nm, g['v_%d_%d' % (c, j)].copy(), g['t_%d_%d' % (c, j)].copy(),
                                            None if np.isnan(s).all() else s.copy()))
        e = Experiment('Exp%d' % c, ms)
        assert [m.variable_name for m in e.measurements] == [str(x) for x in g['order_%d' % c]]
        assert np.array_equal(e.get_unique_timepoints(), g['unique_%d' % c])
        assert np.array_equal(e.get_unique_timepoints(include_zero=True), g['unique0_%d' % c])
        for j, m in enumerate(e.measurements):
            assert np.array_equal(np.asarray(m.std, dtype=float), g['std_%d_%d' % (c, j)])
            v, s, t = m.get_nonzero_measurements()
            assert np.array_equal(v, g['nz_v_%d_%d' % (c, j)]) and np.array_equal(s, g['nz_s_%d_%d' % (c, j)]) \
                and np.array_equal(t, g['nz_t_%d_%d' % (c, j)])
        e.drop_timepoint_zero(e.measurements[0].variable_name)
        for j, m in enumerate(e.measurements):
            assert np.array_equal(np.asarray(m.timepoints, dtype=float), g['d1_t_%d_%d' % (c, j)])
        e.drop_timepoint_zero()
        for j, m in enumerate(e.measurements):
            assert np.array_equal(np.asarray(m.timepoints, dtype=float), g['d2_t_%d_%d' % (c, j)])
            assert np.array_equal(np.asarray(m.values, dtype=float), g['d2_v_%d_%d' % (c, j)])
        assert np.array_equal(e.get_unique_timepoints(include_zero=True), g['unique_after_%d' % c])
    with pytest.raises(KeyError):                      # experiments.py:131-134: one timeseries per variable
        Experiment('E', [TimecourseMeasurement('a', np.ones(2), np.arange(2.0)), TimecourseMeasurement('a', np.ones(2), np.arange(2.0))])
    with pytest.raises(ValueError):                    # abstract_measurement.py:16-17
        TimecourseMeasurement('a', np.ones(2), np.arange(2.0), np.array([1.0, 0.0]))
    with pytest.raises(ValueError):                    # experiments.py:33-36
        Experiment('_x', [TimecourseMeasurement('a', np.ones(2), np.arange(2.0))])


def _grouping_cases(golden):
    g = golden('loss_grouping_ref.npz')
    for c in range(int(g['n_cases'])):
        d = {k: g['%s_%d' % (k, c)] for k in ('exp', 'measure', 'sim', 'data', 'std', 'time', 'groups', 'prior', 'sf',
                                               'scaled', 'after_prior_update')}
        d['groups'] = [str(x).split('|') for x in d['groups']]
        yield c, d


def test_scale_factor_grouping_equals_the_real_reference_base_class(golden):
    """abstract_loss_function.py::LossFunctionWithScaleFactors -- the class that decides which rows of which experiments
    enter which scale factor, lets measures share one, applies the factors and fills the prior row -- runs here as it
    stands, with the real LogScaleFactor (tests/golden/make_golden_loss_grouping.py).  The oracle's restatement on the
    same frames: every group's factor, the scaled simulations, log B in the prior row."""
    for c, g in _grouping_cases(golden):
        data_rows = np.array([not str(e).startswith('~~') for e in g['exp']])
        exps = {}
        rows = [(exps.setdefault(str(e), len(exps)), str(m), d, s, t) for e, m, d, s, t in
                zip(g['exp'][data_rows], g['measure'][data_rows], g['data'][data_rows], g['std'][data_rows], g['time'][data_rows])]
        po = _RowsOnlyOracle(rows, g['sim'][data_rows], None, [tuple(x) if len(x) > 1 else x[0] for x in g['groups']], 1)
        po.loss = 'log'
        B, _, grp, _, _ = po._sf(rows, g['sim'][data_rows])
        assert np.allclose(B, g['sf'], rtol=1e-13), c
        scaled = g['sim'][data_rows] * np.where(grp >= 0, B[np.clip(grp, 0, None)], 1.0)
        assert np.allclose(scaled, g['scaled'][data_rows], rtol=1e-13), c
        # rows of measures outside every group are left alone
        assert np.array_equal(scaled[grp < 0], g['sim'][data_rows][grp < 0])
        if np.isfinite(g['prior'][0]):
            po.sf_priors = {0: (float(g['prior'][0]), float(g['prior'][1]))}
            res = po.residuals(np.zeros(1))
            log_b = g['after_prior_update'][~data_rows][0]            # what the reference wrote into the prior row
            assert log_b == pytest.approx(np.log(g['sf'][0]), rel=1e-13)
            assert res[-1] == pytest.approx((log_b - g['prior'][0]) / g['prior'][1], rel=1e-12, abs=1e-14), c


def test_dense_stiff_golden_is_the_oracles_call(golden):
    """tests/golden/dstiff48_ref.npz (the REAL reference OdeModel on the dense stiff network, make_golden_dense_stiff.py):
    the oracle's odeint call on the same generated C right-hand side gives the same numbers -- vector 0, ~12 s -- and the
    tight solution by column groups agrees with it to LSODA's own accuracy."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    g, gt = golden('dstiff48_ref.npz'), golden('dstiff48_tight.npz')
    gm = GeneratedModel(models_zoo.dense_stiff_spec())
    assert np.array_equal(models_zoo.dense_stiff_ensemble(4096)[1][:len(g['P'])], g['P'])
    S = oo.calc_jacobian(gm, g['P'][0], g['t'], use_c=True)
    Y = oo.simulate(gm, g['P'][0], g['t'], use_c=True)
    # (the same call, but LSODA factors a dense 2352 x 2352 Jacobian with the BLAS of the day: the number of its threads moves
    # the factors by rounding and the trajectory by ~1e-9 relative -- far inside the parity tolerance, not bit for bit)
    from oracle.tolerances import parity_err
    assert parity_err(Y[g['idx']], g['Y'][0]) <= 0.2 and parity_err(S[g['idx']], g['S'][0]) <= 0.2
    assert parity_err(g['Y'][0], gt['Y'][0]) <= 3.0 and parity_err(g['S'][0], gt['S'][0]) <= 5.0
