"""Model definitions used by the tests and the benchmark.

``simple`` and ``michaelis_menten`` restate, in this package's model-text
format, the two known-answer systems of the reference's tests
(math of tests/test_utils/simple_model.py:6-23 and
michelis_menten_model.py:10-33; parameter order of jittable_model.py:1 and
jittable_mm_model.py:1).  ``cascade20`` and ``stiff50`` are this build's
synthetic benchmark networks.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import sympy
from sympy import Symbol

from .symbolic.emit import ModelSpec

SIMPLE_MODEL_TEXT = """
#*! Parameters Start
    k_deg = p[0]
    k_synt = p[1]
#*! Parameters End

#*! Variables Start
    _y = y[0]
#*! Variables End

#*! Differential Equations Start
    d__y = k_synt - k_deg * _y
#*! Differential Equations End
"""

MICHAELIS_MENTEN_TEXT = """
#*! Parameters Start
    vmax = p[0]
    km = p[1]
    k_synt_s = p[2]
    k_deg_s = p[3]
    k_deg_p = p[4]
#*! Parameters End

#*! Variables Start
    _s = y[0]
    _p = y[1]
#*! Variables End

#*! Rate Laws Start
    v_conv = vmax * (_s / (km + _s))
#*! Rate Laws End

#*! Differential Equations Start
    d__s = k_synt_s - v_conv - k_deg_s * _s
    d__p = v_conv - k_deg_p * _p
#*! Differential Equations End
"""


def simple_spec():
    return ModelSpec.from_text(SIMPLE_MODEL_TEXT, name='simple')


def michaelis_menten_spec():
    return ModelSpec.from_text(MICHAELIS_MENTEN_TEXT, name='michaelis_menten')


# ----------------------------------------------------------------------------
# cascade20: 20 species / 40 parameters, negative feedback + saturating cascade
# ----------------------------------------------------------------------------
CASCADE_N = 20
CASCADE_K_FEEDBACK = 1.0


def cascade_spec(n=CASCADE_N, name=None, fixed=()):
    """x_0' = k_0/(1 + x_{n-1}/K) - d_0 x_0 ;  x_i' = k_i x_{i-1}/(1 + x_{i-1}) - d_i x_i.

    Parameter order: k_0..k_{n-1}, d_0..d_{n-1}.  K = 1 is a literal constant.  ``fixed``: names of
    parameters without a sensitivity column.
    """
    xs = [Symbol('x%d' % i) for i in range(n)]
    ks = [Symbol('k%d' % i) for i in range(n)]
    ds = [Symbol('d%d' % i) for i in range(n)]
    eq = OrderedDict()
    eq['x0'] = ks[0] / (1 + xs[n - 1] / sympy.Float(CASCADE_K_FEEDBACK)) - ds[0] * xs[0]
    for i in range(1, n):
        eq['x%d' % i] = ks[i] * xs[i - 1] / (1 + xs[i - 1]) - ds[i] * xs[i]
    # Float(1.0) factors print as 1.0*..., fold them
    eq = OrderedDict((v, sympy.nsimplify(e, rational=True)) for v, e in eq.items())
    return ModelSpec(name=name or ('cascade%d' % n), variables=[str(x) for x in xs],
                     params=[str(k) for k in ks] + [str(d) for d in ds], equations=eq, fixed=list(fixed))


def cascade_nominal_params(n=CASCADE_N):
    """k_i = 1, d_i = 0.1 (1 + i/n)."""
    k = np.ones(n)
    d = 0.1 * (1.0 + np.arange(n) / float(n))
    return np.concatenate([k, d])


def cascade_ensemble(n_vectors=4096, n=CASCADE_N, seed=20261003, spread=0.5):
    """theta_v = log(p_nom) + spread * z, z ~ N(0, I); returns (theta, p = exp(theta)), row-major (V, 2n)."""
    rng = np.random.default_rng(seed)
    theta = np.log(cascade_nominal_params(n))[None, :] + spread * rng.standard_normal((n_vectors, 2 * n))
    return theta, np.exp(theta)


def dense_spec(n=CASCADE_N, density=1.0, name=None, seed=11):
    """The DENSE-coupled variant of the 20-state model ("a dense variant for MFMA
    evaluation"): every species is produced at a rate that saturates in a weighted sum over other species,

        x_i' = k_i u_i / (1 + u_i) - d_i x_i,    u_i = 1/10 + sum_o w_o x_{(i + o) mod n},

    with fixed literal weights w_o in [0.2, 1) on a set of offsets o of the given density (a circulant coupling: the
    same kinetic form in every row, so the row lanes evaluate ONE class, as for the cascade) and a basal input so that
    the network starts from y = 0.  Parameters k_0.., d_0.. as in the cascade (40 sensitivity columns at n = 20):
    df/dy has 1 + density (n - 1) non-zeros per row -- 400 in all at density 1 against the cascade's 40."""
    rng = np.random.default_rng(seed)
    xs = [Symbol('x%d' % i) for i in range(n)]
    ks = [Symbol('k%d' % i) for i in range(n)]
    ds = [Symbol('d%d' % i) for i in range(n)]
    m = max(1, int(round(density * (n - 1))))
    offsets = sorted(int(o) for o in rng.choice(np.arange(1, n), size=m, replace=False))
    eq = OrderedDict()
    for i in range(n):
        u = sympy.Rational(1, 10)
        for o in offsets:
            u = u + sympy.Rational(1, 2) * xs[(i + o) % n]
        eq['x%d' % i] = ks[i] * u / (1 + u) - ds[i] * xs[i]
    return ModelSpec(name=name or ('dense%d' % n if density >= 1.0 else 'dense%d_%02d' % (n, int(round(100 * density)))),
                     variables=[str(x) for x in xs], params=[str(k) for k in ks] + [str(d) for d in ds], equations=eq)


def dense_stiff_spec(n=48, density=0.5, name=None):
    """A DENSE stiff network (BASELINE configs[4] names a "dense Jacobian x S product"; stiff50 itself has two non-zeros per
    row): the densely coupled model of ``dense_spec`` -- every species produced at a rate that saturates in a weighted sum
    over ``density`` (n - 1) others -- with the degradation rates d_i FIXED (no sensitivity columns: n columns k_0 .. k_{n-1},
    n (1 + n) coupled ODEs: 2352 at n = 48, the size of stiff50's system) and, in ``dense_stiff_ensemble``, spread over four
    decades.  Its Newton matrix I - h J_y is dense: factored row-distributed over the lanes (emit_implicit.py IM_DIST)."""
    base = dense_spec(n=n, density=density, name=name or 'dstiff%d' % n)
    return ModelSpec(name=base.name, variables=base.variables, params=base.params, equations=base.equations,
                     fixed=[p for p in base.params if p.startswith('d')])


DENSE_STIFF_T_END = 2.0
DENSE_STIFF_MEASURE_TIMES = np.linspace(0.125, 2.0, 16)


def dense_stiff_ensemble(n_vectors=4096, n=48, seed=20261003, spread=0.25):
    """(theta, P): production rates k_i = d_i x lognormal(spread), degradation rates d_i = 10^(4 i / (n - 1)) x lognormal(spread)
    (time scales from 1 to 1e-4 against t_end = 2: stiff), parameters in model order (k_0.., d_0..)."""
    rng = np.random.default_rng(seed)
    d_nom = 10.0 ** np.linspace(0.0, 4.0, n)
    theta = np.concatenate([np.log(d_nom), np.log(d_nom)])[None, :] + spread * rng.standard_normal((n_vectors, 2 * n))
    return theta, np.exp(theta)


CASCADE_T_END = 100.0
CASCADE_MEASURE_TIMES = np.linspace(6.25, 100.0, 16)
CASCADE_MEASURED_SPECIES = (4, 9, 14, 19)


# ----------------------------------------------------------------------------
# stiff50: 50-state signalling cascade with rate constants spanning 1e6
# ----------------------------------------------------------------------------
def stiff_spec(n=50, name=None, fixed_deactivation=True):
    """Activation cascade with saturating deactivation, every species with its own time scale
    (BASELINE configs[4]):

        x_0' = a_0 ((1 - x_0)         - b_0 x_0 / (1/2 + x_0))
        x_i' = a_i (x_{i-1} (1 - x_i) - b_i x_i / (1/2 + x_i))

    a_i is the RATE of species i (its steady state does not depend on it); nominal a_i = 10^(6 i/(n-1)):
    1 at the top of the cascade, 10^6 at the bottom -- the fast species follow the slow ones
    quasi-statically, which is what makes the system stiff (stiffness ratio 10^7 over t_end = 10) while
    the stage gains stay O(1) (a well-conditioned problem: an ultrasensitive variant amplified rounding
    errors by 10^9 along the cascade and no two integrators agreed on it).  Starting from y0 = 0 every
    fast species starts on its quasi-steady state: no initial layer.  Parameters a_0.. (rates) and b_0..
    (deactivation strengths, nominal 0.5); with ``fixed_deactivation`` the b_i are 'fixed' parameters
    (no sensitivity columns): k = n sensitivity parameters, N = n + n*n = 2550 coupled ODEs at n = 50, as
    SURVEY.md section 8(d) sizes config 5.
    """
    xs = [Symbol('x%d' % i) for i in range(n)]
    a = [Symbol('a%d' % i) for i in range(n)]
    b = [Symbol('b%d' % i) for i in range(n)]
    half = sympy.Rational(1, 2)
    eq = OrderedDict()
    eq['x0'] = a[0] * ((1 - xs[0]) - b[0] * xs[0] / (half + xs[0]))
    for i in range(1, n):
        eq['x%d' % i] = a[i] * (xs[i - 1] * (1 - xs[i]) - b[i] * xs[i] / (half + xs[i]))
    return ModelSpec(name=name or ('stiff%d' % n), variables=[str(x) for x in xs],
                     params=[str(s) for s in a] + [str(s) for s in b], equations=eq,
                     fixed=[str(s) for s in b] if fixed_deactivation else [])


def stiff_nominal_params(n=50):
    a = 10.0 ** np.linspace(0.0, 6.0, n)
    b = 0.5 * np.ones(n)
    return np.concatenate([a, b])


STIFF_T_END = 10.0
STIFF_MEASURE_TIMES = np.linspace(0.625, 10.0, 16)


def stiff_ensemble(n_vectors=4096, n=50, seed=20261003, spread=0.25):
    """theta_v = log(p_nom) + spread * z on the activation rates; deactivation rates stay nominal (they are
    'fixed' parameters of the model).  Returns (theta, p), row-major (V, 2n)."""
    rng = np.random.default_rng(seed)
    p_nom = stiff_nominal_params(n)
    theta = np.tile(np.log(p_nom), (n_vectors, 1))
    theta[:, :n] += spread * rng.standard_normal((n_vectors, n))
    return theta, np.exp(theta)


# ----------------------------------------------------------------------------
# BASELINE.json configs[3]: multi-experiment Project on the cascade model
# ----------------------------------------------------------------------------
def cascade_config4_project(model, n_exp=8, seed=7, simulate=None, project_cls=None, noise=0.05, **project_kw):
    """E experiments 'exp_0'..; setting cond = e; d0..d3 each in its own 'Shared' group keyed on cond
    (4 x E slots), the other 36 parameters Global => q = 36 + 4 E (68 at E = 8).  Each experiment:
    species 4/9/14/19 'direct', 16 timepoints linspace(6.25, 100, 16), sigma = 0.05 |data| + 0.01,
    data = model at nominal parameters (d0..d3 scaled by 1 + 0.1 cond) x (1 + 5 % noise), one scale
    factor per measured species => R = 64 E (512), q = 68, as SURVEY.md section 8(d) specifies.

    ``simulate(p, t_out) -> (len(t_out), n)``: where the synthetic data come from (default: the
    model's own GPU simulate; the oracle-side tests pass the SciPy restatement instead).
    Returns (project, theta_nominal)."""
    from .experiment import Experiment
    from .measurement import TimecourseMeasurement
    if project_cls is None:
        from .project import Project as project_cls
    names = list(model.param_order)
    p_nom = cascade_nominal_params()
    grid = np.linspace(0, CASCADE_T_END, 1000)
    idx = np.searchsorted(grid, CASCADE_MEASURE_TIMES)
    t_out = np.concatenate([[0.0], grid[idx]])
    if simulate is None:
        def simulate(p, t):
            return model.simulate(p, t)
    rng = np.random.default_rng(seed)
    exps = []
    for c in range(n_exp):
        p = p_nom.copy()
        p[20:24] *= 1.0 + 0.1 * c
        y = simulate(p, t_out)[1:]
        ms = []
        for v in CASCADE_MEASURED_SPECIES:
            data = y[:, v] * (1.0 + noise * rng.standard_normal(len(idx)))
            ms.append(TimecourseMeasurement('s%d' % v, data, CASCADE_MEASURE_TIMES.copy(), 0.05 * np.abs(data) + 0.01))
        exps.append(Experiment('exp_%d' % c, ms, experiment_settings={'cond': c}))
    shared = OrderedDict(('deg%d' % i, {('d%d' % i): ('cond',)}) for i in range(4))
    settings = {'Global': [n for n in names if n not in ('d0', 'd1', 'd2', 'd3')], 'Shared': shared}
    mapping = {('s%d' % v): ('direct', v) for v in CASCADE_MEASURED_SPECIES}
    proj = project_cls(model, exps, settings, mapping, sf_groups=['s%d' % v for v in CASCADE_MEASURED_SPECIES],
                       **project_kw)
    theta = np.zeros(proj.n_project_params)
    idx_map = proj.project_param_idx
    for g, slots in idx_map.items():
        for key, gi in slots.items():
            if g.startswith('deg'):
                theta[gi] = np.log(p_nom[20 + int(g[3:])] * (1.0 + 0.1 * key[0]))
            else:
                theta[gi] = np.log(p_nom[names.index(g)])
    return proj, theta


def config4_ensemble(theta_nominal, n_vectors=1024, seed=20261003, spread=0.5):
    rng = np.random.default_rng(seed)
    return theta_nominal[None, :] + spread * rng.standard_normal((n_vectors, theta_nominal.size))
