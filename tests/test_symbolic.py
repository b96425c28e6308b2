"""Emitter tests (CPU): the three generated targets agree, and the HIP text -- compiled
on the host with device qualifiers stubbed -- computes the same augmented RHS as the
Python callables the oracle integrates."""
import ctypes
import io
import os
import subprocess

import numpy as np
import pytest
import sympy

from sysbio_modeling_amd import build, models_zoo
from sysbio_modeling_amd.symbolic import (make_ode_model, make_jit_model, parse_model_file,
                                          process_model_dict, zoo_model, ZOO_NAMES)
from oracle.expanded_sens import reference_style_sensitivity_equations as _derive_sensitivity_equations

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parse_sections():
    md = parse_model_file(models_zoo.MICHAELIS_MENTEN_TEXT)
    assert list(md['Parameters']) == ['vmax', 'km', 'k_synt_s', 'k_deg_s', 'k_deg_p']
    assert list(md['Variables']) == ['_s', '_p']
    assert 'v_conv' in md['Rate Laws']
    assert list(md['Differential Equations']) == ['d__s', 'd__p']


def test_parse_reference_style_function():
    """The reference writes models as python functions with #*! markers and `d_y` for `_y`
    (tests/test_utils/simple_model.py:6-23)."""
    def simple_model(y, t, *args):
        p = args[0]
        #*! Parameters Start
        k_deg = p[0]
        k_synt = p[1]
        #*! Parameters End
        #*! Variables Start
        _y = y[0]
        #*! Variables End
        #*! Differential Equations Start
        d_y = k_synt - k_deg * _y
        #*! Differential Equations End
        return d_y
    gm = make_ode_model(simple_model, name='fn_style')
    assert gm.param_order == ['k_deg', 'k_synt'] and gm.n_vars == 1
    out = np.zeros(3)
    gm.sens_model(np.array([2.0, 0.5, 0.25]), 0.0, out, np.array([0.1, 3.0]))
    assert np.allclose(out, [3.0 - 0.2, -2.0 - 0.05, 1.0 - 0.025])


def test_rate_laws_substituted():
    gm = zoo_model('michaelis_menten')
    y, p = np.array([0.3, 0.7]), np.array([2.0, 0.5, 0.1, 0.2, 0.3])
    out = np.zeros(2)
    gm.model(y, 0.0, out, p)
    v = p[0] * y[0] / (p[1] + y[0])
    assert np.allclose(out, [p[2] - v - p[3] * y[0], v - p[4] * y[1]])


def test_sparse_form_equals_expanded_sensitivity_equations():
    """S' = J_y S + J_p, printed sparsely, equals the reference's expanded definition
    d/dt sens_i_j = df_i/dp_j + sum_m df_i/dy_m sens_m_j (symbolic/sympy_tools.py:130-146)."""
    gm = zoo_model('michaelis_menten')
    spec = gm.spec
    params = {p: sympy.Symbol(p) for p in spec.params}
    expanded = _derive_sensitivity_equations(spec.equations, params)
    rng = np.random.default_rng(3)
    n, k = spec.n_vars, spec.n_sens
    for _ in range(4):
        y = rng.uniform(0.1, 1.0, n + n * k)
        p = rng.uniform(0.1, 1.0, len(spec.params))
        subs = {sympy.Symbol(v): y[i] for i, v in enumerate(spec.variables)}
        subs.update({sympy.Symbol(q): p[i] for i, q in enumerate(spec.params)})
        for i, vi in enumerate(spec.variables):
            for j, pj in enumerate(spec.sens_params):
                subs[sympy.Symbol('sens_%s_%s' % (vi, pj))] = y[n + i * k + j]
        want = [float(e.subs(subs)) for e in expanded.values()]
        out = np.zeros(n + n * k)
        gm.sens_model(y, 0.0, out, p)
        assert np.allclose(out[n:], want, rtol=1e-12)


def test_fixed_parameters_drop_sensitivity_columns():
    gm = make_ode_model(models_zoo.MICHAELIS_MENTEN_TEXT, name='mm_fixed', fixed_params=['km'])
    assert gm.sens_params == ['vmax', 'k_synt_s', 'k_deg_s', 'k_deg_p'] and gm.n_sens == 4
    full = zoo_model('michaelis_menten')
    rng = np.random.default_rng(5)
    y5 = rng.uniform(0.1, 1, 12)
    p = rng.uniform(0.1, 1, 5)
    keep = [0, 2, 3, 4]
    y4 = np.concatenate([y5[:2], y5[2:].reshape(2, 5)[:, keep].ravel()])
    o5, o4 = np.zeros(12), np.zeros(10)
    full.sens_model(y5, 0.0, o5, p)
    gm.sens_model(y4, 0.0, o4, p)
    assert np.allclose(o4[2:].reshape(2, 4), o5[2:].reshape(2, 5)[:, keep])
    with pytest.raises(KeyError):
        process_model_dict(parse_model_file(models_zoo.SIMPLE_MODEL_TEXT), fixed_params=['nope'])


def test_make_jit_model_facade_and_output_fh():
    fh = io.StringIO()
    fn = make_jit_model(models_zoo.SIMPLE_MODEL_TEXT, fh, calculate_sensitivities=False, name='facade')
    out = np.zeros(1)
    fn(np.array([1.0]), 0.0, out, np.array([0.5, 2.0]))
    assert out[0] == 1.5
    assert 'def sens_model(y, t, yout, p):' in fh.getvalue()


@pytest.mark.parametrize('name', ZOO_NAMES)
def test_committed_headers_are_current(name):
    """csrc/models/<name>.hpp is exactly what the emitter prints today (the GPU box builds
    plugins from the committed text)."""
    gm = zoo_model(name)
    with open(os.path.join(build.MODELS_DIR, name + '.hpp')) as fh:
        assert fh.read() == gm.hip_source


@pytest.mark.parametrize('name', ZOO_NAMES)
def test_hip_text_equals_python_and_c_on_host(name, tmp_path):
    gm = zoo_model(name)
    hdr = os.path.join(build.MODELS_DIR, name + '.hpp')
    so = str(tmp_path / ('h_%s.so' % name))
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-DSBM_MODEL_HEADER="%s"' % hdr,
                           os.path.join(HERE, 'support', 'host_model_harness.cpp'), '-o', so])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    n, k = gm.n_vars, gm.n_sens
    assert (lib.h_n_vars(), lib.h_n_params(), lib.h_n_sens()) == (n, len(gm.param_order), k)
    clib = gm.c_library()
    rng = np.random.default_rng(11)
    for _ in range(5):
        y = rng.uniform(0.05, 2.0, n + n * k)
        p = rng.uniform(0.05, 2.0, len(gm.param_order))
        ref = np.zeros(n + n * k)
        gm.sens_model(y, 0.0, ref, p)
        for fn in (lib.h_sens_rhs, lib.h_sens_rhs_fused, lib.h_sens_rhs_rowlane, clib.sbm_sens_rhs):
            out = np.zeros(n + n * k)
            fn(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp), p.ctypes.data_as(dp))
            assert np.allclose(out, ref, rtol=1e-12, atol=1e-14)
        out = np.zeros(n + n * k)
        rc_ = lib.h_sens_rhs_rowgroup(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp),
                                      p.ctypes.data_as(dp))
        # a single row cannot be split (stiff50: five chunks of ten columns, nine rows per lane)
        assert rc_ == (-1 if name == 'simple' else 0)
        if rc_ == 0:
            assert np.allclose(out, ref, rtol=1e-12, atol=1e-14)
            # the small-batch layout (more, smaller column chunks) of the same form
            out = np.zeros(n + n * k)
            assert lib.h_sens_rhs_rowgroup_small_batch(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp),
                                                       p.ctypes.data_as(dp)) == 0
            assert np.allclose(out, ref, rtol=1e-12, atol=1e-14)
        ref_s = np.zeros(n)
        gm.model(y[:n].copy(), 0.0, ref_s, p)
        out = np.zeros(n)
        lib.h_rhs(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp), p.ctypes.data_as(dp))
        assert np.allclose(out, ref_s, rtol=1e-13, atol=1e-15)
        assert np.allclose(ref[:n], ref_s, rtol=1e-13, atol=1e-15)
    # which models have a distinct small-batch layout (Michaelis-Menten: 10 lanes either way)
    assert lib.h_rowgroup_layouts_differ() == (1 if name in ('cascade20', 'stiff50') else 0)


def test_cascade_definition():
    """the synthetic benchmark network."""
    spec = models_zoo.cascade_spec()
    assert spec.n_vars == 20 and spec.n_params == 40 and spec.n_sens == 40
    gm = zoo_model('cascade20')
    assert len(gm.derived.jy) == 40 and len(gm.derived.jp) == 40   # <= 2 non-zeros per row each
    p = models_zoo.cascade_nominal_params()
    assert np.allclose(p[:20], 1.0) and np.allclose(p[20:], 0.1 * (1 + np.arange(20) / 20.0))
    y = np.linspace(0.1, 2.0, 20)
    out = np.zeros(20)
    gm.model(y, 0.0, out, p)
    assert out[0] == pytest.approx(1.0 / (1 + y[19]) - 0.1 * y[0])
    assert out[7] == pytest.approx(y[6] / (1 + y[6]) - p[27] * y[7])
    theta, P = models_zoo.cascade_ensemble(16)
    assert P.shape == (16, 40) and np.allclose(np.exp(theta), P)
    theta2, _ = models_zoo.cascade_ensemble(16)
    assert np.array_equal(theta, theta2)   # seeded


def _irregular_model():
    """11 species, mixed kinetic forms, couplings at irregular distances."""
    import sympy
    from collections import OrderedDict
    from sysbio_modeling_amd.symbolic import GeneratedModel
    from sysbio_modeling_amd.symbolic.emit import ModelSpec
    from sysbio_modeling_amd.symbolic import emit_rowgroup
    n = 11
    xs = [sympy.Symbol('x%d' % i) for i in range(n)]
    ks = [sympy.Symbol('k%d' % i) for i in range(n)]
    d, K = sympy.Symbol('d'), sympy.Symbol('K')
    eq = OrderedDict()
    for i in range(n):
        a, b = xs[(i * 3 + 1) % n], xs[(i + 5) % n]
        if i % 3 == 0:
            rhs = ks[i] * a / (K + a) - d * xs[i]
        elif i % 3 == 1:
            rhs = ks[i] * a * b - d * xs[i] * xs[i]
        else:
            rhs = ks[i] / (1 + b) - d * xs[i] + a
        eq['x%d' % i] = rhs
    spec = ModelSpec(name='irregular', variables=[str(x) for x in xs], params=[str(k) for k in ks] + ['d', 'K'],
                     equations=eq)
    return GeneratedModel(spec)


def test_rowgroup_form_of_an_irregular_network(tmp_path):
    """Row-group emitter on a network whose rows do NOT line up across groups: mixed kinetic forms,
    couplings at irregular distances (so most terms become LDS halo terms, several groups lack
    terms others have), 11 rows over 13 columns.  The lane-by-lane emulation must still equal the
    Python emitter's sensitivity RHS, and padding must stay exactly zero."""
    from sysbio_modeling_amd.symbolic import emit_rowgroup
    gm = _irregular_model()
    spec, n = gm.spec, gm.n_vars
    lay = emit_rowgroup.layout(spec, gm.derived)
    assert lay is not None and lay['G'] >= 2 and len(lay['hsrc']) > 0
    hdr = tmp_path / 'irregular.hpp'
    hdr.write_text(gm.hip_source)
    so = str(tmp_path / 'h_irregular.so')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-DSBM_MODEL_HEADER="%s"' % hdr,
                           os.path.join(HERE, 'support', 'host_model_harness.cpp'), '-o', so])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    k = gm.n_sens
    rng = np.random.default_rng(3)
    for _ in range(4):
        y = rng.uniform(0.05, 2.0, n + n * k)
        p = rng.uniform(0.05, 2.0, len(gm.param_order))
        ref = np.zeros(n + n * k)
        gm.sens_model(y, 0.0, ref, p)
        for fn in (lib.h_sens_rhs_rowlane, lib.h_sens_rhs_rowgroup):
            out = np.zeros(n + n * k)
            rc_ = fn(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp), p.ctypes.data_as(dp))
            assert rc_ in (0, None) or fn is lib.h_sens_rhs_rowlane
            assert np.allclose(out, ref, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('seed,n', [(21, 45), (22, 70), (23, 130)])
def test_rowgroup_chunks_of_large_irregular_networks_on_host(tmp_path, seed, n):
    """Random rate-law networks beyond one column per lane (45 species, ~110 parameters) and beyond one state row
    per lane (70 and 130 species): several classes, couplings at random distances (most terms are halo terms),
    columns in chunks.  The lane-by-lane, chunk-by-chunk emulation of the row-group form equals the Python
    emitter's sensitivity RHS and padding stays exactly zero."""
    from tests.test_gpu_user_models import _random_network
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(_random_network(seed, n))
    k = gm.n_sens
    assert gm.n_vars == n and k > 64 and 'RG_OK = true' in gm.hip_source and 'RG_NCH = 1;' not in gm.hip_source
    hdr = tmp_path / 'net.hpp'
    hdr.write_text(gm.hip_source)
    so = str(tmp_path / 'h_net.so')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-DSBM_MODEL_HEADER="%s"' % hdr,
                           os.path.join(HERE, 'support', 'host_model_harness.cpp'), '-o', so])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(seed)
    for _ in range(2):
        y = rng.uniform(0.05, 2.0, n + n * k)
        p = rng.uniform(0.05, 2.0, len(gm.param_order))
        ref = np.zeros(n + n * k)
        gm.sens_model(y, 0.0, ref, p)
        for fn in (lib.h_sens_rhs_rowgroup, lib.h_sens_rhs_rowgroup_small_batch):
            out = np.zeros(n + n * k)
            assert fn(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp), p.ctypes.data_as(dp)) == 0
            assert np.allclose(out, ref, rtol=1e-12, atol=1e-14)


def test_rowgroup_plan():
    from sysbio_modeling_amd.symbolic.emit_rowgroup import plan
    assert plan(20, 40) == (3, 20, 2, 7, 1)    # cascade20: 14 elements on 60 lanes instead of 20 on 40
    assert plan(1, 2) is None                  # nothing to split
    assert plan(257, 3) is None                # more than four rows per lane
    G, C, CPL, RPG, NCH = plan(30, 20)
    assert NCH == 1 and G * C <= 64 and C * CPL >= 20 and G * RPG >= 30 and RPG * CPL <= 0.8 * 30
    # more columns than one wavefront holds in registers: chunks of columns, one wavefront each
    # ... and more state variables than lanes: up to four rows per lane
    for n, nk in ((30, 64), (40, 80), (50, 50), (64, 200), (5, 300), (70, 140), (70, 10), (128, 256)):
        G, C, CPL, RPG, NCH = plan(n, nk)
        assert G * C <= 64 and G * RPG >= n and (G - 1) * RPG < n
        assert C * CPL * NCH >= nk and C * CPL * (NCH - 1) < nk       # every chunk holds a column
        assert RPG * CPL <= 15                                       # DOPRI45's stage vectors fit 256 registers
    assert plan(40, 80)[4] > 1 and plan(50, 50)[4] > 1
    # small batches: the work of ONE wavefront counts, extra wavefronts are free
    from sysbio_modeling_amd.symbolic.emit_rowgroup import plan_latency, latency_plan_or_none
    G, C, CPL, RPG, NCH = plan_latency(20, 40)
    assert NCH > 1 and RPG * CPL <= 5 and C * CPL * NCH >= 40
    assert latency_plan_or_none(20, 40) == (G, C, CPL, RPG, NCH)
    assert latency_plan_or_none(2, 5) is None          # Michaelis-Menten: nothing to gain
    assert latency_plan_or_none(130, 289) is None      # the small-batch split would leave the register budget


@pytest.mark.parametrize('n_states,n_chunks', [(40, 5), (70, 14)])
def test_rowgroup_column_chunks_on_host(tmp_path, n_states, n_chunks):
    """Cascades beyond one column per lane (40 states / 80 parameters) and beyond one state row per lane (70 /
    140): the chunked row-group form, emulated lane by lane and chunk by chunk, equals the Python emitter's
    sensitivity RHS."""
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(models_zoo.cascade_spec(n_states, name='cascade%d' % n_states))
    n, k = gm.n_vars, gm.n_sens
    assert (n, k) == (n_states, 2 * n_states) and 'RG_NCH = %d;' % n_chunks in gm.hip_source
    hdr = tmp_path / 'cascade.hpp'
    hdr.write_text(gm.hip_source)
    so = str(tmp_path / 'h_cascade.so')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-DSBM_MODEL_HEADER="%s"' % hdr,
                           os.path.join(HERE, 'support', 'host_model_harness.cpp'), '-o', so])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    rng = np.random.default_rng(5)
    for _ in range(3):
        y = rng.uniform(0.05, 2.0, n + n * k)
        p = rng.uniform(0.05, 2.0, len(gm.param_order))
        ref = np.zeros(n + n * k)
        gm.sens_model(y, 0.0, ref, p)
        out = np.zeros(n + n * k)
        assert lib.h_sens_rhs_rowgroup(y.ctypes.data_as(dp), ctypes.c_double(0.0), out.ctypes.data_as(dp),
                                       p.ctypes.data_as(dp)) == 0
        assert np.allclose(out, ref, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('name', ZOO_NAMES + ('irregular',))
def test_generated_sparse_lu_solves_newton_matrix(name, tmp_path):
    """im_build / im_factor / im_solve (emit_implicit.py) against numpy's dense solve of
    (I - gamma*J_y) x = b, J_y by central differences of the generated Python RHS."""
    if name == 'irregular':
        gm = _irregular_model()
        hdr = tmp_path / 'irregular.hpp'
        hdr.write_text(gm.hip_source)
        hdr = str(hdr)
    else:
        gm = zoo_model(name)
        hdr = os.path.join(build.MODELS_DIR, name + '.hpp')
    so = str(tmp_path / ('him_%s.so' % name))
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-fPIC', '-shared', '-DSBM_MODEL_HEADER="%s"' % hdr,
                           os.path.join(HERE, 'support', 'host_model_harness.cpp'), '-o', so])
    lib = ctypes.CDLL(so)
    dp = ctypes.POINTER(ctypes.c_double)
    n = gm.n_vars
    rng = np.random.default_rng(8)
    for gamma in (1e-3, 0.3, 5.0):
        y = rng.uniform(0.1, 1.5, n)
        p = rng.uniform(0.1, 1.5, len(gm.param_order))
        J = np.zeros((n, n))
        for m in range(n):
            e = np.zeros(n); e[m] = 1e-6
            fp, fm = np.zeros(n), np.zeros(n)
            gm.model(y + e, 0.0, fp, p); gm.model(y - e, 0.0, fm, p)
            J[:, m] = (fp - fm) / 2e-6
        b = rng.standard_normal(n)
        x = b.copy()
        lib.h_im_solve(ctypes.c_double(gamma), y.ctypes.data_as(dp), ctypes.c_double(0.0), p.ctypes.data_as(dp),
                       x.ctypes.data_as(dp))
        ref = np.linalg.solve(np.eye(n) - gamma * J, b)
        assert np.allclose(x, ref, rtol=1e-6, atol=1e-8)
        # distributed form of lower-triangular patterns (feed-forward models)
        x2 = b.copy()
        rc_ = lib.h_im_solve_tri(ctypes.c_double(gamma), y.ctypes.data_as(dp), ctypes.c_double(0.0),
                                 p.ctypes.data_as(dp), x2.ctypes.data_as(dp))
        assert rc_ == (0 if name in ('simple', 'michaelis_menten', 'stiff50') else -1)
        if rc_ == 0:
            assert np.allclose(x2, x, rtol=1e-12, atol=1e-14)


def test_symbolic_lu_fill_in():
    from sysbio_modeling_amd.symbolic.emit_implicit import symbolic_lu
    # arrow matrix pointing the wrong way: eliminating column 0 fills the whole trailing block
    n = 5
    ent = [(i, 0) for i in range(n)] + [(0, j) for j in range(n)]
    pattern, _ = symbolic_lu(n, ent)
    assert len(pattern) == n * n
    # lower bidiagonal: no fill
    pattern, ops = symbolic_lu(n, [(i, i - 1) for i in range(1, n)] + [(i, i) for i in range(n)])
    assert len(pattern) == 2 * n - 1 and all(j <= i for i, j in pattern)


def test_simplification_is_done_once_per_structure_and_renamed_back():
    """emit._cheapest works on a placeholder form of the expression and renames the result: the same kinetic form with
    other symbols gives the same result up to that renaming, equal in value to the input, and costs one dictionary
    look-up the second time."""
    from sysbio_modeling_amd.symbolic import emit
    a, b, x, y = sympy.symbols('a b x y')
    k, K, u, v = sympy.symbols('k K u v')
    e1 = sympy.diff(a * x / (b + x) - y * x, x)
    e2 = sympy.diff(k * u / (K + u) - v * u, u)
    emit._cheapest_memo.clear()
    r1 = emit._cheapest(e1)
    n_after_first = len(emit._cheapest_memo)
    r2 = emit._cheapest(e2)
    assert len(emit._cheapest_memo) == n_after_first == 1
    assert sympy.simplify(r1 - e1) == 0 and sympy.simplify(r2 - e2) == 0
    assert r1.xreplace({a: k, b: K, x: u, y: v}) == r2
    assert sympy.count_ops(r1) <= sympy.count_ops(e1)
    # no placeholder leaks into the result
    assert r1.free_symbols <= {a, b, x, y} and r2.free_symbols <= {k, K, u, v}
