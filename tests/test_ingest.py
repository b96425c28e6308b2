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
    def loops(y, t, yout, p):
        for i in range(2):
            yout[i] = (-p[0] * y[i])

    def unknown_call(y, t, yout, p):
        yout[0] = (np.interp(t, [0, 1], [0, 1]) - p[0] * y[0])

    def missing_row(y, t, yout, p):
        yout[1] = (-p[0] * y[0])
    for fn in (loops, unknown_call, missing_row, lambda y, t, yout, p: None, np.sin):
        with pytest.raises(TypeError):
            ingest.spec_from_callables(fn, None, 2, ['k'])


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
