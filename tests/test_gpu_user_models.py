"""User models end to end on the GPU box: model text -> SymPy -> generated HIP -> hipcc at run time
-> plugin -> kernels, against SciPy odeint on the generated Python callables (the reference's own
pipeline is parse -> process -> make_ode_model -> exec, symbolic/sympy_tools.py:162-397).  Nothing here
is a committed header: these are the code paths a user's own model takes."""
import numpy as np
import pytest

from tests.conftest import parity_err

pytestmark = pytest.mark.gpu

# a small signalling motif with a conservation law, rate laws and a fixed parameter
MOTIF_TEXT = """
#*! Parameters Start
    k_on = p[0]
    k_off = p[1]
    k_cat = p[2]
    k_deg = p[3]
    e_tot = p[4]
#*! Parameters End

#*! Variables Start
    _c = y[0]
    _p = y[1]
    _q = y[2]
#*! Variables End

#*! Conservation Laws Start
    _e = e_tot - _c
#*! Conservation Laws End

#*! Rate Laws Start
    v_bind = k_on * _e * (1.0 / (1.0 + _p))
    v_cat = k_cat * _c
#*! Rate Laws End

#*! Differential Equations Start
    d__c = v_bind - k_off * _c - v_cat
    d__p = v_cat - k_deg * _p
    d__q = k_deg * _p - 0.05 * _q
#*! Differential Equations End
"""


def _odeint_ref(gm, p, t):
    from oracle import odeint_oracle as oo
    return oo.simulate(gm, p, t), oo.calc_jacobian(gm, p, t)


def test_model_text_with_laws_and_fixed_parameter(tmp_path):
    """Conservation law + rate laws substituted, one parameter 'fixed' (no sensitivity column: the
    compact column index of SURVEY.md section 8a, quirk 6); every kernel variant the model supports."""
    from sysbio_modeling_amd.symbolic import make_ode_model
    from sysbio_modeling_amd.model import OdeModel
    gm = make_ode_model(MOTIF_TEXT, name='motif', fixed_params=['e_tot'])
    assert gm.n_vars == 3 and len(gm.param_order) == 5 and gm.n_sens == 4
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='motif')
    rng = np.random.default_rng(4)
    P = np.array([0.8, 0.2, 0.5, 0.1, 1.5]) * np.exp(0.3 * rng.standard_normal((5, 5)))
    t = np.linspace(0, 30.0, 1000)
    idx = np.arange(0, 1000, 111)
    for variant in ('per_wave', 'row_lane', 'row_group', 'auto'):
        S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant)
        assert m.last_info['status'].tolist() == [0] * 5 and S.shape == (5, len(idx), 3 * 4)
        for v in (0, 4):
            Yr, Sr = _odeint_ref(gm, P[v], t)
            assert parity_err(Y[v], Yr[idx]) <= 1.0 and parity_err(S[v], Sr[idx]) <= 1.0
    Y = m.simulate_batch(P, t[idx])
    assert parity_err(Y[0], _odeint_ref(gm, P[0], t)[0][idx]) <= 1.0
    # stiff-capable path on the same model
    S2 = m.calc_jacobian_batch(P[:2], t[idx], method='implicit_midpoint', n_steps=4096, extrapolate=1,
                               rtol=1e-11, atol=1e-13)
    assert parity_err(S2[0], _odeint_ref(gm, P[0], t)[1][idx]) <= 1.0


FORCED_TEXT = """
#*! Parameters Start
    k_in = p[0]
    w = p[1]
    d1 = p[2]
    k2 = p[3]
    d2 = p[4]
#*! Parameters End
#*! Variables Start
    _a = y[0]
    _b = y[1]
#*! Variables End
#*! Differential Equations Start
    d__a = k_in * (1.0 + 0.5 * sin(w * t)) - d1 * _a
    d__b = k2 * _a / (1.0 + _a) - d2 * _b * exp(-0.01 * t)
#*! Differential Equations End
"""


def test_explicit_time_dependence():
    """A non-autonomous right-hand side (periodic input, slowly decaying degradation): the callback contract carries t
    (f(y, t, yout, p), the reference's fixtures ignore it), so every stage must evaluate the model at ITS time.  All
    kernels, the implicit rule (midpoint time) and a start time other than zero against odeint."""
    from sysbio_modeling_amd.symbolic import make_ode_model
    from sysbio_modeling_amd.model import OdeModel
    gm = make_ode_model(FORCED_TEXT, name='forced')
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='forced')
    rng = np.random.default_rng(8)
    P = np.array([1.0, 0.7, 0.2, 0.5, 0.3]) * np.exp(0.2 * rng.standard_normal((4, 5)))
    t = np.linspace(0, 40.0, 1000)
    idx = np.array([0, 250, 600, 999])
    Yr, Sr = _odeint_ref(gm, P[2], t)
    for variant in ('auto', 'per_wave', 'row_lane', 'row_group', 'small_batch'):
        for method, kw in (('dopri45', {}), ('rk4', {'n_steps': 8192})):
            S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant, method=method, **kw)
            assert m.last_info['status'].max() == 0
            assert parity_err(Y[2], Yr[idx]) <= 1.0 and parity_err(S[2], Sr[idx]) <= 1.0, (variant, method)
    assert parity_err(m.simulate_batch(P, t[idx])[2], Yr[idx]) <= 1.0
    S_im = m.calc_jacobian_batch(P, t[idx], method='implicit_midpoint', n_steps=8192, extrapolate=1, rtol=1e-11, atol=1e-13)
    assert parity_err(S_im[2], Sr[idx]) <= 1.0
    S_c = m.calc_jacobian_batch(P, t[idx], method='implicit_controlled')
    assert m.last_info['status'].max() == 0 and parity_err(S_c[2], Sr[idx]) <= 1.0
    # t_sim[0] is the time of the initial condition (odeint semantics): start at t = 7.5 from a given state
    t2 = np.linspace(7.5, 40.0, 400)
    y0 = np.concatenate([[0.4, 0.9], np.zeros(10)])
    from oracle import odeint_oracle as oo
    Sr2, Yr2 = oo.calc_jacobian(gm, P[2], t2, init_conditions=y0, return_states=True)
    S2, Y2 = m.calc_jacobian_batch(P[2:3], t2[[0, 150, 399]], init_conditions=y0, return_states=True)
    assert parity_err(Y2[0], Yr2[[0, 150, 399]]) <= 1.0 and parity_err(S2[0], Sr2[[0, 150, 399]]) <= 1.0


HILL_TEXT = """
#*! Parameters Start
    vmax = p[0]
    K = p[1]
    d = p[2]
    k2 = p[3]
#*! Parameters End
#*! Variables Start
    _s = y[0]
    _q = y[1]
#*! Variables End
#*! Rate Laws Start
    hill = vmax * _q**4 / (K**4 + _q**4)
#*! Rate Laws End
#*! Differential Equations Start
    d__s = k2 + hill - d * _s
    d__q = k2 * sqrt(1.0 + _s) - d * _q * tanh(_q) - 0.1 * _q + log(1.0 + _s**2)
#*! Differential Equations End
"""


def test_transcendental_rate_laws():
    """Hill kinetics (fourth powers), sqrt, tanh and log in the rate laws: the device code of every kernel against
    odeint on the generated Python callables (same expressions, printed for two languages)."""
    from sysbio_modeling_amd.symbolic import make_ode_model
    from sysbio_modeling_amd.model import OdeModel
    gm = make_ode_model(HILL_TEXT, name='hill')
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='hill')
    rng = np.random.default_rng(2)
    P = np.array([1.0, 0.7, 0.2, 0.5]) * np.exp(0.25 * rng.standard_normal((5, 4)))
    t = np.linspace(0, 30.0, 1000)
    idx = np.array([0, 120, 500, 999])
    Yr, Sr = _odeint_ref(gm, P[3], t)
    for variant in ('auto', 'per_wave', 'row_lane', 'small_batch'):
        S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant)
        assert m.last_info['status'].max() == 0
        assert parity_err(Y[3], Yr[idx]) <= 1.0 and parity_err(S[3], Sr[idx]) <= 1.0, variant
    assert parity_err(m.simulate_batch(P, t[idx])[3], Yr[idx]) <= 1.0
    S_c = m.calc_jacobian_batch(P, t[idx], method='implicit_controlled')
    assert m.last_info['status'].max() == 0 and parity_err(S_c[3], Sr[idx]) <= 1.0


ODD_TEXT = """
#*! Parameters Start
    k1 = p[0]
    unused = p[1]
    d1 = p[2]
    k2 = p[3]
#*! Parameters End
#*! Variables Start
    _a = y[0]
    _b = y[1]
    _c = y[2]
#*! Variables End
#*! Differential Equations Start
    d__a = k1 - d1 * _a
    d__b = k2 * _a - d1 * _b
    d__c = 0.3 - 0.1 * _c
#*! Differential Equations End
"""


def test_degenerate_structure():
    """A parameter no equation uses (an all-zero sensitivity column, as the reference's expanded equations would
    carry it), a state decoupled from everything with literal constants only, a parameter shared by two rows."""
    from sysbio_modeling_amd.symbolic import make_ode_model
    from sysbio_modeling_amd.model import OdeModel
    gm = make_ode_model(ODD_TEXT, name='odd')
    assert gm.n_vars == 3 and gm.n_sens == 4
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='odd')
    P = np.array([[0.8, 5.0, 0.3, 0.6], [1.2, -1.0, 0.2, 0.9]])
    t = np.linspace(0, 25.0, 1000)
    idx = np.array([0, 200, 999])
    for v in range(2):
        Yr, Sr = _odeint_ref(gm, P[v], t)
        for variant in ('auto', 'per_wave', 'row_lane', 'small_batch'):
            S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant)
            assert m.last_info['status'].max() == 0
            assert parity_err(Y[v], Yr[idx]) <= 1.0 and parity_err(S[v], Sr[idx]) <= 1.0, variant
            Sv = S[v].reshape(len(idx), 3, 4)
            assert np.all(Sv[:, :, 1] == 0.0) and np.all(Sv[:, 2, :] == 0.0)      # unused parameter; decoupled state
    S_c = m.calc_jacobian_batch(P, t[idx], method='implicit_controlled')
    assert m.last_info['status'].max() == 0 and parity_err(S_c[1], Sr[idx]) <= 1.0


def test_more_sensitivity_columns_than_lanes():
    """35 species, 70 parameters: more columns than a wavefront has lanes.  The row-group kernel cuts them into
    chunks (one wavefront each, every chunk with its own copy of the state and its own step control); the
    per-wave kernel runs two columns per lane; the implicit kernel runs one wavefront per 64 columns.  All
    agree with LSODA."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    gm = GeneratedModel(models_zoo.cascade_spec(35, name='cascade35'))
    assert gm.n_sens == 70 and 'RG_OK = true' in gm.hip_source and 'RG_NCH = 1;' not in gm.hip_source
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade35')
    rng = np.random.default_rng(9)
    P = models_zoo.cascade_nominal_params(35)[None, :] * np.exp(0.3 * rng.standard_normal((3, 70)))
    t = np.linspace(0, 60.0, 1000)
    idx = np.array([0, 300, 999])
    Yr, Sr = _odeint_ref(gm, P[1], t)
    steps = {}
    for variant in ('auto', 'row_group', 'per_wave'):
        S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant)
        assert m.last_info['status'].tolist() == [0, 0, 0]
        assert parity_err(Y[1], Yr[idx]) <= 1.0 and parity_err(S[1], Sr[idx]) <= 1.0, variant
        steps[variant] = m.last_info['n_steps'].copy()
    assert np.array_equal(steps['auto'], steps['row_group'])           # AUTO picks the chunked row-group kernel
    # the single-vector call takes the small-batch split (eight chunks instead of three): same numbers to the tolerance
    S_one = m.calc_jacobian(P[1], t[idx])
    assert m.last_info['status'].tolist() == [0] and parity_err(S_one, Sr[idx]) <= 1.0
    assert int(m.last_info['n_steps'][0]) != int(steps['auto'][1])
    S_rk = m.calc_jacobian_batch(P, t[idx], method='rk4', n_steps=8192)
    assert np.allclose(S_rk, S, rtol=1e-7, atol=1e-9)
    S_im = m.calc_jacobian_batch(P, t[idx], method='implicit_midpoint', n_steps=8192, extrapolate=1,
                                 rtol=1e-11, atol=1e-13)
    assert m.last_info['status'].tolist() == [0, 0, 0]
    assert parity_err(S_im[1], Sr[idx]) <= 1.0
    # initial sensitivities reach the right chunk: S(t0) = S0 comes back unchanged at the first output time
    S0 = rng.standard_normal((35, 70))
    y0 = np.concatenate([np.full(35, 0.3), S0.ravel()])
    S_ic = m.calc_jacobian_batch(P[:1], np.array([0.0, 1.0]), init_conditions=y0)
    assert np.array_equal(S_ic[0, 0], S0.ravel())
    # a failing vector is reported once for all its chunks (worst status), its rows are NaN, its neighbours untouched
    Pbad = P.copy()
    Pbad[1, 3] = np.nan
    with pytest.warns(UserWarning, match='integration failed for 1 of 3'):
        Sb = m.calc_jacobian_batch(Pbad, t[idx])
    assert m.last_info['status'].tolist()[0] == 0 and m.last_info['status'][1] != 0 and m.last_info['status'][2] == 0
    S_ref = m.calc_jacobian_batch(P, t[idx])
    assert np.all(np.isnan(Sb[1, 1:])) and np.array_equal(Sb[0], S_ref[0]) and np.array_equal(Sb[2], S_ref[2])


def test_project_on_a_model_with_more_parameters_than_lanes():
    """A Project on the 35-state / 70-parameter cascade (sensitivity columns in chunks): two experiments, a
    'Shared' degradation rate, scale factors.  The project Jacobian equals central finite differences of the
    batched residuals (the reference's own test style, tests/test_Project.py:145-176), column by column."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    from sysbio_modeling_amd.experiment import Experiment
    from sysbio_modeling_amd.measurement import TimecourseMeasurement
    from sysbio_modeling_amd.project import Project
    n = 35
    gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade35'))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade35')
    p_nom = models_zoo.cascade_nominal_params(n)
    times = np.linspace(5.0, 60.0, 8)
    grid = np.linspace(0, 60.0, 1000)
    rows = np.searchsorted(grid, times)
    exps = []
    for e, scale in enumerate((1.0, 1.6)):
        p = p_nom.copy()
        p[n] *= scale                                            # d0 differs between the experiments
        Y = m.simulate(p, grid)[rows]
        ms = [TimecourseMeasurement('x%d' % v, 2.0 * Y[:, v] * (1 + 0.03 * np.cos(np.arange(8) + v)), times.copy(),
                                    0.05 * np.abs(Y[:, v]) + 0.01) for v in (7, 20, 34)]
        exps.append(Experiment('exp_%d' % e, ms, experiment_settings={'cond': e}))
    settings = {'Global': [k for k in gm.param_order if k != 'd0'], 'Shared': {'deg0': {'d0': ('cond',)}}}
    proj = Project(m, exps, settings, {('x%d' % v): ('direct', v) for v in (7, 20, 34)},
                   sf_groups=['x7', 'x20', 'x34'], reference_compat=False)
    q = proj.n_project_params
    assert q == 71
    rng = np.random.default_rng(12)
    theta = np.zeros(q)
    names = list(gm.param_order)
    for g, slots in proj.project_param_idx.items():
        for key, gi in slots.items():
            theta[gi] = np.log(p_nom[names.index('d0' if g == 'deg0' else g)])
    theta += 0.05 * rng.standard_normal(q)
    J = proj.calc_project_jacobian(theta)
    assert J.shape == (48, q) and np.all(np.isfinite(J))
    h = 1e-4
    Th = np.repeat(theta[None, :], 2 * q, axis=0)
    for k in range(q):
        Th[2 * k, k] += h
        Th[2 * k + 1, k] -= h
    R = proj.residuals_batch(Th, rtol=1e-11, atol=1e-13)
    J_fd = ((R[0::2] - R[1::2]) / (2 * h)).T
    # columns of parameters with next to no influence are judged against the largest column (difference noise)
    scale = np.abs(J).max(axis=0, keepdims=True) + 1e-2 * np.abs(J).max()
    assert np.max(np.abs(J - J_fd) / scale) <= 1e-3
    assert np.sum(np.abs(J).max(axis=0) > 1e-2 * np.abs(J).max()) >= 10      # not a test of zeros


def test_more_state_variables_than_lanes():
    """70 species: a lane of the row kernels carries two state rows (lane, lane + 64).  18 of the 140 parameters
    have sensitivity columns (the rest are 'fixed': keeps the per-wave comparison kernel small).  Row-group
    (AUTO), per-wave and LSODA agree; so does the state-only path (state-rows kernel, two rows per lane)."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    n = 70
    fixed = ['d%d' % i for i in range(n)] + ['k%d' % i for i in range(n) if i % 4]
    gm = GeneratedModel(models_zoo.cascade_spec(n, name='cascade70f', fixed=fixed))
    assert gm.n_vars == 70 and gm.n_sens == 18 and 'RG_OK = true' in gm.hip_source
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='cascade70f')
    rng = np.random.default_rng(4)
    P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.3 * rng.standard_normal((3, 2 * n)))
    t = np.linspace(0, 60.0, 1000)
    idx = np.array([0, 300, 999])
    Yr, Sr = _odeint_ref(gm, P[2], t)
    res = {}
    for variant in ('auto', 'per_wave'):
        S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant)
        assert m.last_info['status'].tolist() == [0, 0, 0]
        assert parity_err(Y[2], Yr[idx]) <= 1.0 and parity_err(S[2], Sr[idx]) <= 1.0, variant
        res[variant] = (S, m.last_info['n_steps'].copy())
    assert not np.array_equal(res['auto'][0], res['per_wave'][0])      # two kernels, not one
    Ys = m.simulate_batch(P, t[idx])
    assert m.last_info['status'].tolist() == [0, 0, 0] and parity_err(Ys[2], Yr[idx]) <= 1.0
    S_rk = m.calc_jacobian_batch(P, t[idx], method='rk4', n_steps=8192)
    assert np.allclose(S_rk, res['auto'][0], rtol=1e-7, atol=1e-9)
    y0 = np.concatenate([0.2 + 0.01 * np.arange(n), rng.standard_normal(n * 18)])
    S_ic, Y_ic = m.calc_jacobian_batch(P[:1], np.array([0.0, 1.0]), init_conditions=y0, return_states=True)
    assert np.array_equal(Y_ic[0, 0], y0[:n]) and np.array_equal(S_ic[0, 0], y0[n:])
    # the implicit kernels take two state rows per lane as well (sbm_implicit_stepper.hpp): same solution
    # (an amplifier cascade wants a relative test: tests/test_gpu_implicit.py, the seventy-state test)
    S_im = m.calc_jacobian_batch(P, t[idx], method='implicit_controlled', rtol=1e-9, atol=1e-18)
    assert m.last_info['status'].tolist() == [0, 0, 0] and parity_err(S_im[2], Sr[idx]) <= 1.0
    assert np.array_equal(m.calc_jacobian_batch(P, t[idx], method='auto'), res['auto'][0])
    assert not m.last_info['stiff'].any()


def test_three_state_rows_per_lane_against_finite_differences():
    """A random network of 130 species and ~300 parameters (39 000 sensitivity ODEs per trajectory: three state
    rows per lane, columns in chunks).  LSODA on the augmented system is out of reach here, so the columns are
    checked another way: d y / d p_j by central differences of the STATE-ONLY kernel (a different kernel, no
    sensitivity code involved), and the states against LSODA on the 130 state equations."""
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    from oracle import odeint_oracle as oo
    gm = GeneratedModel(_random_network(23, 130))
    n, k = gm.n_vars, gm.n_sens
    assert n == 130 and k > 256 and 'RG_OK = true' in gm.hip_source
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    rng = np.random.default_rng(5)
    p = np.exp(rng.uniform(np.log(0.3), np.log(1.5), len(gm.param_order)))
    t = np.linspace(0, 10.0, 1000)
    idx = np.array([0, 400, 999])
    S, Y = m.calc_jacobian_batch(p[None, :], t[idx], return_states=True)
    assert m.last_info['status'].tolist() == [0]
    Yr = oo.simulate(gm, p, t)[idx]
    assert parity_err(Y[0], Yr) <= 1.0
    S = S[0].reshape(len(idx), n, k)
    cols = rng.choice(k, size=12, replace=False)
    names = list(gm.param_order)
    P2 = np.repeat(p[None, :], 2 * len(cols), axis=0)
    hs = []
    for q, j in enumerate(cols):
        pj = names.index(gm.sens_params[j])
        h = 1e-4 * p[pj]
        P2[2 * q, pj] += h
        P2[2 * q + 1, pj] -= h
        hs.append(h)
    Yp = m.simulate_batch(P2, t[idx], rtol=1e-12, atol=1e-14)
    assert m.last_info['status'].max() == 0
    for q, j in enumerate(cols):
        fd = (Yp[2 * q] - Yp[2 * q + 1]) / (2 * hs[q])
        scale = np.abs(fd).max() + 1e-12
        assert np.max(np.abs(S[:, :, j] - fd)) <= 1e-5 * scale, (j, gm.sens_params[j])
    # the fixed-step integrator on the same right-hand side lands on the same numbers
    S_rk = m.calc_jacobian_batch(p[None, :], t[idx], method='rk4', n_steps=4096)[0].reshape(len(idx), n, k)
    assert np.allclose(S_rk, S, rtol=1e-6, atol=1e-8 * np.abs(S).max())
    # the implicit kernels keep a whole column of S per lane in registers: 128 state variables at most, a clear error
    # beyond, and 'auto' runs without its fallback
    with pytest.raises(Exception, match='n_vars <= 128'):
        m.simulate_batch(p[None, :], t[idx], method='implicit_midpoint', n_steps=64)
    assert np.array_equal(m.simulate_batch(p[None, :], t[idx], method='auto'), m.simulate_batch(p[None, :], t[idx]))
    assert not m.last_info['status'].any()


def _random_network(seed, n):
    """Random rate-law network: every species is produced from one or two others (mass action or
    saturating), degraded linearly, some with product inhibition; bounded by construction."""
    import sympy
    from collections import OrderedDict
    from sysbio_modeling_amd.symbolic.emit import ModelSpec
    rng = np.random.default_rng(seed)
    xs = [sympy.Symbol('x%d' % i) for i in range(n)]
    params, eq = [], OrderedDict()

    def par(name):
        params.append(name)
        return sympy.Symbol(name)
    for i in range(n):
        kind = int(rng.integers(0, 4))
        d = par('d%d' % i)
        if i == 0 or kind == 0:
            rhs = par('k%d' % i) - d * xs[i]
        elif kind == 1:
            j = int(rng.integers(0, n))
            rhs = par('k%d' % i) * xs[j] / (1 + xs[j]) - d * xs[i]
        elif kind == 2:
            j, l = int(rng.integers(0, n)), int(rng.integers(0, n))
            rhs = par('k%d' % i) * xs[j] / ((1 + xs[j]) * (1 + xs[l] / par('K%d' % i))) - d * xs[i]
        else:
            j = int(rng.integers(0, i))
            rhs = par('k%d' % i) * xs[j] - d * xs[i] * xs[i]
        eq['x%d' % i] = rhs
    return ModelSpec(name='rand%d_%d' % (n, seed), variables=[str(x) for x in xs], params=params, equations=eq)


@pytest.mark.parametrize('seed,n', [(1, 6), (2, 11), (3, 17), (4, 30)])   # the last: 66 parameters, 3 column chunks
def test_random_networks_all_variants(seed, n):
    """Emitter + kernels on networks nobody tuned them for: whatever classes, row splits and halo terms
    the generator comes up with, every variant must reproduce SciPy odeint on the generated callables."""
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.model import OdeModel
    gm = GeneratedModel(_random_network(seed, n))
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    rng = np.random.default_rng(100 + seed)
    P = np.exp(rng.uniform(np.log(0.2), np.log(2.0), (4, len(gm.param_order))))
    t = np.linspace(0, 20.0, 1000)
    idx = np.array([0, 333, 999])
    ref = [_odeint_ref(gm, P[v], t) for v in (0, 3)]
    for variant in ('per_wave', 'row_lane', 'row_group'):
        for method, kw in (('dopri45', {}), ('rk4', {'n_steps': 8192})):
            S, Y = m.calc_jacobian_batch(P, t[idx], return_states=True, variant=variant, method=method, **kw)
            assert m.last_info['status'].tolist() == [0, 0, 0, 0]
            for k, v in enumerate((0, 3)):
                assert parity_err(Y[v], ref[k][0][idx]) <= 1.0, (variant, method)
                assert parity_err(S[v], ref[k][1][idx]) <= 1.0, (variant, method)
    Y = m.simulate_batch(P, t[idx])
    assert parity_err(Y[0], ref[0][0][idx]) <= 1.0
    S = m.calc_jacobian_batch(P, t[idx], method='implicit_midpoint', n_steps=4096, extrapolate=1, rtol=1e-11, atol=1e-13)
    assert parity_err(S[3], ref[1][1][idx]) <= 1.0


OSCILLATOR_TEXT = """
#*! Parameters Start
    k = p[0]
    c = p[1]
    F = p[2]
    w = p[3]
#*! Parameters End
#*! Variables Start
    _x = y[0]
    _v = y[1]
#*! Variables End
#*! Differential Equations Start
    d__x = _v
    d__v = F * sin(w * t) - k * _x - c * _v
#*! Differential Equations End
"""


def test_sign_changing_states_and_sensitivities_under_the_default_tolerances():
    """A driven, damped oscillator: both states and all eight sensitivities cross zero again and again.  The default
    atol of the explicit pairs is 1e-18 -- effectively a relative test (model/ode_model.py::default_tolerances) -- and an
    entry that passes through zero has a vanishing scale at that instant; the error norm is an RMS over a column (and
    the packed / row kernels share it), so a zero crossing costs a rejected attempt now and then, not a collapse of the
    step size: parity with the reference's LSODA on the reference's grid, no step-budget exit, and a step count within
    a factor 1.6 of the run with atol = 1e-12, for DOPRI45, DOP853 and the stiff integrator."""
    from sysbio_modeling_amd.symbolic import make_ode_model
    from sysbio_modeling_amd.model import OdeModel
    gm = make_ode_model(OSCILLATOR_TEXT, name='oscillator')
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name='oscillator')
    rng = np.random.default_rng(12)
    P = np.array([4.0, 0.3, 1.0, 1.3]) * np.exp(0.15 * rng.standard_normal((6, 4)))
    t = np.linspace(0, 40.0, 1000)
    idx = np.arange(40, 1000, 60)
    Yr, Sr = _odeint_ref(gm, P[1], t)
    assert np.sum(np.diff(np.sign(Yr[idx, 0])) != 0) >= 5 and np.sum(np.diff(np.sign(Sr[idx, 0])) != 0) >= 3     # they do cross zero
    t_out = np.concatenate([[0.0], t[idx]])
    for method in ('dopri45', 'dop853'):
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, method=method)
        steps = m.last_info['n_steps'].copy()
        assert m.last_info['status'].max() == 0
        assert parity_err(Y[1, 1:], Yr[idx]) <= 1.0 and parity_err(S[1, 1:], Sr[idx]) <= 1.0, method
        S12 = m.calc_jacobian_batch(P, t_out, method=method, atol=1e-12)
        steps12 = m.last_info['n_steps'].copy()
        assert parity_err(S12[1, 1:], Sr[idx]) <= 1.0
        print("%s on the oscillator: steps at atol 1e-18 %s, at 1e-12 %s" % (method, steps.tolist(), steps12.tolist()))
        assert np.all(steps <= 1.6 * steps12 + 20), (method, steps, steps12)
    # the single-vector methods (method='auto': explicit attempt with the early-exit budget) do not mistake it for stiff
    S1 = m.calc_jacobian(P[1], t_out)
    assert m.last_info['stiff'].tolist() == [False] and parity_err(S1[1:], Sr[idx]) <= 1.0
    S_c, Y_c = m.calc_jacobian_batch(P, t_out, return_states=True, method='implicit_controlled')
    assert m.last_info['status'].max() == 0
    assert parity_err(Y_c[1, 1:], Yr[idx]) <= 1.0 and parity_err(S_c[1, 1:], Sr[idx]) <= 1.0
