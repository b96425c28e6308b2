"""Plain f(y, t, yout, p) callables in the reference's emitted form are read from their source and compiled
(sysbio_modeling_amd/symbolic/ingest.py): the construction pattern of the reference's own tests,
OdeModel(jittable_model.model, sens_jittable_model.sens_model, n_vars, ordered_params)
(tests/test_OdeModel.py:12-18), works without going through this package's generator."""
import importlib.util
import os

import numpy as np
import pytest
import sympy

from sysbio_modeling_amd.symbolic import ingest, zoo_model, GeneratedModel
from tests.support import plain_models as pm

REF_UTILS = '/root/reference/tests/test_utils'


def test_spec_from_hand_written_callable():
    spec = ingest.spec_from_callables(pm.model, None, pm.n_vars, pm.ordered_params)
    assert spec.params == pm.ordered_params and spec.variables == ['y0', 'y1', 'y2'] and spec.fixed == []
    a, b, c, t = sympy.symbols('y0 y1 y2 t')
    k_in, k_conv, K, k_out, tau = sympy.symbols(' '.join(pm.ordered_params))
    want = [k_in * (1 - sympy.exp(-t / tau)) - k_conv * a ** 2 / (K ** 2 + a ** 2),
            k_conv * a ** 2 / (K ** 2 + a ** 2) - k_conv * b,
            k_conv * b - k_out * sympy.sqrt(c + 1) * c]
    for v, w in zip(spec.variables, want):
        assert sympy.simplify(spec.equations[v] - w) == 0
    # the callables generated from the parsed spec reproduce the hand-written one numerically
    gm = GeneratedModel(spec)
    rng = np.random.default_rng(0)
    y, p = rng.uniform(0.1, 1, 3), rng.uniform(0.2, 2, 5)
    o1, o2 = np.zeros(3), np.zeros(3)
    pm.model(y, 0.7, o1, p)
    gm.model(y, 0.7, o2, p)
    assert np.allclose(o1, o2, rtol=1e-14)


def test_fixed_parameters_are_found_from_the_sens_model():
    spec = ingest.spec_from_callables(pm.model, pm.sens_model, pm.n_vars, pm.ordered_params)
    assert spec.fixed == ['K_half', 'tau'] and spec.sens_params == ['k_in', 'k_conv', 'k_out']
    gm = GeneratedModel(spec)
    rng = np.random.default_rng(1)
    z, p = rng.uniform(0.1, 1, 12), rng.uniform(0.2, 2, 5)
    o1, o2 = np.zeros(12), np.zeros(12)
    pm.sens_model(z, 0.3, o1, p)
    gm.sens_model(z, 0.3, o2, p)
    assert np.allclose(o1, o2, rtol=1e-12, atol=1e-14)


def test_a_sens_model_that_disagrees_is_refused():
    def wrong(y, t, yout, p):
        pm.sens_model(y, t, yout, p)
        yout[5] = (yout[5] * 1.001)
    # (defined in a test: its source is available, its numbers are off)
    with pytest.raises(TypeError, match="does not agree|no choice"):
        ingest.spec_from_callables(pm.model, wrong, pm.n_vars, pm.ordered_params)


def test_what_cannot_be_read_is_refused():
    def data_loop(y, t, yout, p):
        i = 0
        while y[i] > 0:                       # (a loop whose trip count depends on the data)
            yout[i] = (-p[0] * y[i])
            i += 1

    def data_bound(y, t, yout, p):
        for i in range(len(y)):               # (not a static integer)
            yout[i] = (-p[0] * y[i])

    def data_branch(y, t, yout, p):
        if y[0] > 1.0:                        # (a right-hand side with a kink has no sensitivities)
            yout[0] = -p[0]
        else:
            yout[0] = (-p[0] * y[0])
        yout[1] = 0.0 * y[1]

    def unknown_call(y, t, yout, p):
        yout[0] = (np.interp(t, [0, 1], [0, 1]) - p[0] * y[0])

    def missing_row(y, t, yout, p):
        yout[1] = (-p[0] * y[0])
    for fn in (data_loop, data_bound, data_branch, unknown_call, missing_row, lambda y, t, yout, p: None, np.sin):
        with pytest.raises(TypeError):
            ingest.spec_from_callables(fn, None, 2, ['k'])


def test_static_loops_and_branches_unroll_to_the_straight_line_form():
    """The reference takes any callable (model/ode_model.py:27-44); a hand-written cascade is a loop over the species with
    the first one special-cased.  Loops with static bounds, branches on loop variables, integer index expressions and
    accumulators are unrolled at parse time (round 4) into exactly the equations of the straight-line form."""
    n = 6

    def looped(y, t, yout, p):
        n = 6
        total = 0.0
        for i in range(n):
            total += y[i]
        for i in range(0, n):
            if i == 0:
                yout[i] = (p[0] / (1.0 + y[n - 1]) - p[n] * y[0])          # feedback from the last species
            else:
                yout[i] = (p[i] * y[i - 1] / (1.0 + y[i - 1]) - p[n + i] * y[i])
        for i in range(1, n, 2):
            yout[i] -= 0.01 * total * y[i]                                   # (odd species leak in proportion to the total)

    def straight(y, t, yout, p):
        total = y[0] + y[1] + y[2] + y[3] + y[4] + y[5]
        yout[0] = (p[0] / (1.0 + y[5]) - p[6] * y[0])
        yout[1] = (p[1] * y[0] / (1.0 + y[0]) - p[7] * y[1] - 0.01 * total * y[1])
        yout[2] = (p[2] * y[1] / (1.0 + y[1]) - p[8] * y[2])
        yout[3] = (p[3] * y[2] / (1.0 + y[2]) - p[9] * y[3] - 0.01 * total * y[3])
        yout[4] = (p[4] * y[3] / (1.0 + y[3]) - p[10] * y[4])
        yout[5] = (p[5] * y[4] / (1.0 + y[4]) - p[11] * y[5] - 0.01 * total * y[5])
    names = ['k%d' % i for i in range(n)] + ['d%d' % i for i in range(n)]
    a = ingest.spec_from_callables(looped, None, n, names)
    b = ingest.spec_from_callables(straight, None, n, names)
    for v in a.variables:
        assert sympy.simplify(a.equations[v] - b.equations[v]) == 0
    # ... and they ARE the function: the parsed equations reproduce a call of the Python callable
    rng = np.random.default_rng(5)
    yv, pv = rng.uniform(0.2, 1.5, n), rng.uniform(0.2, 1.5, 2 * n)
    out = np.zeros(n)
    looped(yv, 0.0, out, pv)
    at = {sympy.Symbol('y%d' % i): yv[i] for i in range(n)}
    at.update({sympy.Symbol(nm): pv[i] for i, nm in enumerate(names)})
    assert np.allclose([float(a.equations[v].subs(at)) for v in a.variables], out, rtol=1e-13)
    # nested loops and an index expression of two loop variables
    def nested(y, t, yout, p):
        for i in range(2):
            yout[i] = 0.0 * y[i]
            for j in range(2):
                yout[i] += p[2 * i + j] * y[j]
    c = ingest.spec_from_callables(nested, None, 2, ['a', 'b', 'c', 'd'])
    ys = [sympy.Symbol('y0'), sympy.Symbol('y1')]
    assert sympy.simplify(c.equations['y1'] - (sympy.Symbol('c') * ys[0] + sympy.Symbol('d') * ys[1])) == 0


@pytest.mark.skipif(not os.path.isdir(REF_UTILS), reason="the reference tree exists in the build container only")
def test_the_references_own_emitted_fixtures_are_read():
    """tests/test_utils/jittable_model.py / sens_jittable_model.py / jittable_mm_model.py / sens_jittable_mm_model.py
    of the reference: same equations as the zoo restatements of the two known-answer systems."""
    def load(name):
        s = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF_UTILS, name + '.py'))
        m = importlib.util.module_from_spec(s)
        s.loader.exec_module(m)
        return m
    for plain, sens, n, params, zoo in (('jittable_model', 'sens_jittable_model', 1, ['k_deg', 'k_synt'], 'simple'),
                                        ('jittable_mm_model', 'sens_jittable_mm_model', 2, None, 'michaelis_menten')):
        mod, smod = load(plain), load(sens)
        params = params or list(mod.ordered_params)
        spec = ingest.spec_from_callables(mod.model, smod.sens_model, n, params)
        z = zoo_model(zoo).spec
        assert spec.params == z.params and spec.fixed == []
        ren = {sympy.Symbol(v): sympy.Symbol('y%d' % i) for i, v in enumerate(z.variables)}
        for v, zv in zip(spec.variables, z.variables):
            assert sympy.simplify(spec.equations[v] - sympy.sympify(z.equations[zv]).subs(ren)) == 0


@pytest.mark.gpu
def test_odemodel_from_plain_callables_on_the_gpu():
    """The reference's construction pattern end to end: plain callables in, GPU results out, checked against
    scipy.integrate.odeint on the very same callables (the reference's exact call)."""
    from scipy.integrate import odeint
    from sysbio_modeling_amd.model import OdeModel
    from tests.conftest import parity_err
    m = OdeModel(pm.model, pm.sens_model, pm.n_vars, pm.ordered_params, model_name='plain chain')
    assert m.n_vars == 3 and m.sens_params == ['k_in', 'k_conv', 'k_out']
    p = np.array([0.8, 0.5, 0.4, 0.3, 5.0])
    t = np.linspace(0, 40, 30)

    def wrap(fn, n):
        out = np.zeros(n)

        def f(y, tt):
            fn(y, tt, out, p)
            return out
        return f
    Yr = odeint(wrap(pm.model, 3), np.zeros(3), t, rtol=1e-10, atol=1e-10)
    Sr = odeint(wrap(pm.sens_model, 12), np.zeros(12), t, rtol=1e-10, atol=1e-10)[:, 3:]
    assert parity_err(m.simulate(p, t), Yr) <= 1.0
    assert parity_err(m.calc_jacobian(p, t, np.zeros(12)), Sr) <= 1.0


@pytest.mark.gpu
def test_a_hand_written_looped_cascade_runs_on_the_gpu():
    """A right-hand side written the way a person writes a cascade -- a loop over the species, the first one special-cased --
    goes through OdeModel (no sens_model handed in: the sensitivity system is derived) and agrees with odeint on the very
    same callable, states and finite-difference-free sensitivities (odeint on the derived system of the oracle emitter)."""
    from scipy.integrate import odeint
    from sysbio_modeling_amd.model import OdeModel
    from tests.conftest import parity_err
    n = 8

    def looped(y, t, yout, p):
        n = 8
        for i in range(n):
            if i == 0:
                yout[i] = (p[0] / (1.0 + y[n - 1]) - p[n] * y[0])
            else:
                yout[i] = (p[i] * y[i - 1] / (1.0 + y[i - 1]) - p[n + i] * y[i])
    names = ['k%d' % i for i in range(n)] + ['d%d' % i for i in range(n)]
    m = OdeModel(looped, None, n, names, model_name='looped cascade')
    p = np.concatenate([np.full(n, 1.0), 0.1 * (1.0 + np.arange(n) / n)])
    t = np.linspace(0, 60, 25)
    out = np.zeros(n)

    def f(y, tt):
        looped(y, tt, out, p)
        return out
    Yr = odeint(f, np.zeros(n), t, rtol=1e-10, atol=1e-10)
    assert parity_err(m.simulate(p, t), Yr) <= 1.0
    # sensitivities: central differences of odeint at 1e-12 (the derived system itself is pinned by tests/test_symbolic.py)
    S = m.calc_jacobian(p, t, np.zeros(n + n * 2 * n)).reshape(len(t), n, 2 * n)
    for j in (0, 3, n + 2):
        dp = 1e-6 * p[j]
        pp, pm_ = p.copy(), p.copy()
        pp[j] += dp
        pm_[j] -= dp

        def g(q):
            o = np.zeros(n)

            def ff(y, tt):
                looped(y, tt, o, q)
                return o
            return odeint(ff, np.zeros(n), t, rtol=1e-12, atol=1e-14)
        fd = (g(pp) - g(pm_)) / (2 * dp)
        assert np.max(np.abs(S[:, :, j] - fd)) <= 1e-6 * max(1.0, np.max(np.abs(fd)))
