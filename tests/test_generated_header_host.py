"""The generated HIP header on the host: it is plain C++ once the device qualifiers and the few macros of
csrc/sbm_integrators.hpp are defined away, so its members can be compiled with g++ and called through ctypes -- no GPU.
Checked here, on random networks (lower-triangular ones with several sub-diagonal entries and several J_p entries per
row -- patterns the zoo models do not have -- and general ones with feedback):

  * the row-lane form (emit_rowlane.py: class bodies + operand / result tables) reproduces f, J_y and J_p of the generated
    Python model;
  * ``im_sens_tri`` (the fused sensitivity step of csrc/sbm_implicit_extrap.hpp: J_p pick next to the substitution, tables
    loaded a block of rows ahead) returns BIT FOR BIT what the two-pass form it replaced returns -- the J_p pick of
    sbm_implicit_stepper.hpp::sens_euler followed by ``im_solve_tri`` -- for every sensitivity column and for a lane
    without one;
  * ``im_solve_tri`` with the factors ``im_build`` / ``im_factor`` produce solves (I - gamma J_y) x = b (numpy);
  * ``im_solve_tri_pick`` hands lane i component i of the same solution;
  * the symbolic LU of a NON-triangular J_y (fill-in worked out at generation time) solves its Newton matrix;
  * the row-DISTRIBUTED LU of densely coupled networks (lane-level code with SBM_LANE_BCAST), emulated one lane at a time,
    solves it too.

The GPU tests exercise these members inside the kernels on the zoo's patterns (tests/test_gpu_implicit.py)."""
import ctypes
import subprocess
from collections import OrderedDict

import numpy as np
import pytest

HARNESS = r'''
#include <cmath>
#define __device__
#define __forceinline__ inline
#define __constant__ static const
#define SBM_RCP(x) (1.0 / (x))
#define SBM_SEL(c, a, b) ((c) ? (a) : (b))
#define SBM_PICK(scol, c, v, otherwise) ((scol) == (c) ? (v) : (otherwise))
#define SBM_PICK_COL(col, c, v, otherwise) ((col) == (c) ? (v) : (otherwise))
#include <vector>
// SBM_LANE_BCAST(v, src) = the value of v in lane src.  One lane at a time on the host: the j-th broadcast of a (straight-line)
// function is learnt by running the source lane up to it -- it depends on earlier broadcasts only -- and replayed to the others.
struct Probe { int src; double v; };
static std::vector<double> g_bc;
static size_t g_idx;
static inline double lane_bcast(double v, int src) {
  const size_t j = g_idx++;
  if (j < g_bc.size()) return g_bc[j];
  throw Probe{src, v};
}
#define SBM_LANE_BCAST(v, src) lane_bcast((v), (src))
#define SBM_LDS_FENCE()
using std::fma;
#include "%(header)s"
typedef SbmModel M;
extern "C" {
int nv() { return M::NV; }
int nk() { return M::NK; }
int maxjp() { return M::RL_MAXJP; }
int maxjy() { return M::RL_MAXJY; }
int im_mf() { return M::IM_MF; }
int im_nm() { return M::IM_NM; }
int njy() { return M::NNZ_JY; }
int is_tri() { return M::IM_TRI ? 1 : 0; }
int has_fused() { return M::IM_SENS_TRI ? 1 : 0; }
int jpcol(int q, int i) { return M::rl_jpcol(q, i); }
int rstart(int i) { return M::im_rstart(i); }
int mfpos(int s, int i) { return M::im_mfpos(s, i); }
int diagslot(int i) { return M::im_diagslot(i); }
int jycol(int s, int i) { return M::rl_jycol(s, i); }
// the two-pass form: sbm_implicit_stepper.hpp::sens_euler before the fusion
void two_pass(const double* mf, const double* ja, double hh, int col, double* z_io) {
  double z[M::NV];
  for (int i = 0; i < M::NV; ++i) z[i] = z_io[i];
  for (int i = 0; i < M::NV; ++i) {
    double a = 0.0;
    for (int q = 0; q < M::RL_MAXJP; ++q) a = (M::rl_jpcol(q, i) == col) ? ja[i * M::RL_MAXJP + q] : a;
    z[i] = fma(hh, a, z[i]);
  }
  M::im_solve_tri(mf, z);
  for (int i = 0; i < M::NV; ++i) z_io[i] = z[i];
}
void fused(const double* mf, const double* ja, double hh, int col, double* z_io) {
  double z[M::NV];
  for (int i = 0; i < M::NV; ++i) z[i] = z_io[i];
  M::im_sens_tri(mf, ja, hh, col, z);
  for (int i = 0; i < M::NV; ++i) z_io[i] = z[i];
}
void solve_tri(const double* mf, double* b_io) {
  double b[M::NV];
  for (int i = 0; i < M::NV; ++i) b[i] = b_io[i];
  M::im_solve_tri(mf, b);
  for (int i = 0; i < M::NV; ++i) b_io[i] = b[i];
}
void solve_tri_pick(const double* mf, const double* g, int lane, double* d_out) {
  constexpr int RPL = (M::NV + 63) / 64;
  double d[RPL];
  for (int r = 0; r < RPL; ++r) d[r] = 0.0;
  M::im_solve_tri_pick<RPL>(mf, g, lane, d);
  for (int r = 0; r < RPL; ++r) d_out[r] = d[r];
}
// what row lane `row` evaluates: operands gathered through the row-lane tables, one class body
void eval_row(int row, double t, const double* y, const double* p, double* f, double* jy, double* jp) {
  double ys[M::RL_MAXYS], ps[M::RL_MAXPS], jyo[M::RL_MAXJY], jpo[M::RL_MAXJP];
  for (int s = 0; s < M::RL_MAXYS; ++s) ys[s] = y[M::rl_ys(s, row)];
  for (int s = 0; s < M::RL_MAXPS; ++s) ps[s] = p[M::rl_ps(s, row)];
  for (int s = 0; s < M::RL_MAXJY; ++s) jyo[s] = 0.0;
  for (int s = 0; s < M::RL_MAXJP; ++s) jpo[s] = 0.0;
  double ff = 0.0;
  M::class_dispatch(M::rl_class(row), t, ys, ps, ff, jyo, jpo);
  *f = ff;
  for (int s = 0; s < M::RL_MAXJY; ++s) jy[s] = jyo[s];
  for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = jpo[s];
}
// the row-distributed factorisation (lane i owns row i of M) in lockstep emulation, then im_solve_lds on the published rows;
// returns the number of broadcasts, -1 on an inconsistency
int dist_factor_solve(const double* Md, double* b_io) {
  constexpr int N = M::NV;
  g_bc.clear();
  double rd[N], row[N];
  auto run = [&](int lane) {
    for (int j = 0; j < N; ++j) row[j] = Md[lane * N + j];
    g_idx = 0;
    double (&m)[N] = row;
    M::im_factor_rows(m, lane, rd);
  };
  for (;;) {
    try { run(0); break; }
    catch (const Probe& p) {
      try { run(p.src); return -1; }
      catch (const Probe& q) { g_bc.push_back(q.v); }
    }
  }
  std::vector<double> mf((size_t)N * M::IM_LD + 2, 0.0);
  for (int lane = 0; lane < N; ++lane) {
    run(lane);
    for (int j = 0; j < N; ++j) mf[(size_t)lane * M::IM_LD + j] = row[j];
  }
  double b[N];
  for (int i = 0; i < N; ++i) b[i] = b_io[i];
  M::im_solve_lds(mf.data(), rd, b);
  for (int i = 0; i < N; ++i) b_io[i] = b[i];
  return (int)g_bc.size();
}
int is_dist() { return M::IM_DIST ? 1 : 0; }
void build_factor_solve(double gamma, const double* jy, double* b_io) {
  double m[M::IM_NM], b[M::NV];
  M::im_build(gamma, jy, m);
  M::im_factor(m);
  for (int i = 0; i < M::NV; ++i) b[i] = b_io[i];
  M::im_solve(m, b);
  for (int i = 0; i < M::NV; ++i) b_io[i] = b[i];
}
}
'''


def _triangular_network(seed, n):
    """species i is produced from one to three species j < i (mass action, saturating, or a product of two) with its own
    rate constants and degraded linearly: J_y lower triangular with up to three sub-diagonal entries per row, up to four
    parameters per row."""
    import sympy
    from sysbio_modeling_amd.symbolic.emit import ModelSpec
    rng = np.random.default_rng(seed)
    xs = [sympy.Symbol('x%d' % i) for i in range(n)]
    params, eq = [], OrderedDict()

    def par(name):
        params.append(name)
        return sympy.Symbol(name)
    for i in range(n):
        rhs = -par('d%d' % i) * xs[i]
        if i == 0:
            rhs += par('k0')
        else:
            for term in range(int(rng.integers(1, 4))):
                j = int(rng.integers(0, i))
                kind = int(rng.integers(0, 3))
                if term == 2 and kind != 1:
                    kind = 0            # (at most four parameters per row: the sparse J_p table of the implicit kernels)
                k = par('k%d_%d' % (i, term)) if term < 2 else sympy.Symbol('k%d_1' % i)
                if kind == 0:
                    rhs += k * xs[j]
                elif kind == 1:
                    rhs += k * xs[j] / (1 + xs[j])
                else:
                    rhs += k * xs[j] * xs[int(rng.integers(0, i))]
        eq['x%d' % i] = rhs
    return ModelSpec(name='tri%d_%d' % (n, seed), variables=[str(x) for x in xs], params=params, equations=eq)


def _host_library(gm, tmp_path):
    header = tmp_path / (gm.name + '.hpp')
    header.write_text(gm.hip_source)
    src = tmp_path / 'harness.cpp'
    src.write_text(HARNESS % dict(header=str(header)))
    so = tmp_path / 'harness.so'
    # -ffp-contract=off: only the fma() calls the generator wrote are fused, in both forms alike
    subprocess.run(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', '-Wno-unknown-pragmas',
                    str(src), '-o', str(so)], check=True, capture_output=True)
    lib = ctypes.CDLL(str(so))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.two_pass.argtypes = lib.fused.argtypes = [dp, dp, ctypes.c_double, ctypes.c_int, dp]
    lib.solve_tri.argtypes = [dp, dp]
    lib.solve_tri_pick.argtypes = [dp, dp, ctypes.c_int, dp]
    lib.build_factor_solve.argtypes = [ctypes.c_double, dp, dp]
    lib.dist_factor_solve.argtypes = [dp, dp]
    lib.eval_row.argtypes = [ctypes.c_int, ctypes.c_double, dp, dp, dp, dp, dp]
    for f in ('jpcol', 'mfpos', 'jycol'):
        getattr(lib, f).argtypes = [ctypes.c_int, ctypes.c_int]
    for f in ('rstart', 'diagslot'):
        getattr(lib, f).argtypes = [ctypes.c_int]
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


@pytest.mark.parametrize('seed,n', [(1, 5), (2, 13), (3, 24), (4, 40), (5, 70)])
def test_fused_sensitivity_step_equals_the_two_pass_form(tmp_path, seed, n):
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(_triangular_network(seed, n))
    lib = _host_library(gm, tmp_path)
    assert lib.nv() == n and lib.is_tri() == 1 and lib.has_fused() == 1
    nk, maxjp, n_mf = lib.nk(), lib.maxjp(), lib.im_mf()
    assert 2 <= maxjp <= 4
    rng = np.random.default_rng(100 + seed)
    rows_with = {q: sum(1 for i in range(n) if lib.jpcol(q, i) >= 0) for q in range(maxjp)}
    assert rows_with[maxjp - 1] >= 1 and rows_with[0] == n           # several J_p entries per row do occur
    sub = [sum(1 for s in range(lib.maxjy()) if 0 <= lib.mfpos(s, i) < n_mf) for i in range(n)]
    assert max(sub) >= 2 or n < 10                                   # ... and several sub-diagonal entries
    mf = rng.standard_normal(n_mf + 2)
    ja = rng.standard_normal(n * maxjp + 2)
    for col in list(range(nk)) + [nk + 3]:
        z0 = rng.standard_normal(n)
        a, b = z0.copy(), z0.copy()
        lib.two_pass(_p(mf), _p(ja), 0.37, col, _p(a))
        lib.fused(_p(mf), _p(ja), 0.37, col, _p(b))
        assert np.array_equal(a, b), col
        if col < nk:
            assert not np.array_equal(a, _solve_only(lib, mf, z0))    # the column's J_p entries did enter


def _solve_only(lib, mf, z0):
    b = z0.copy()
    lib.solve_tri(_p(mf), _p(b))
    return b


@pytest.mark.parametrize('seed,n', [(2, 13), (5, 70)])
def test_triangular_factors_solve_the_newton_matrix(tmp_path, seed, n):
    """im_build + im_factor + im_solve against numpy on (I - gamma J_y) x = b; the distributed table form (reciprocal
    pivot at rstart, scaled entries at mfpos -- what the row lanes write on the GPU) through im_solve_tri and
    im_solve_tri_pick gives the same solution."""
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(_triangular_network(seed, n))
    lib = _host_library(gm, tmp_path)
    d = gm.derived
    rng = np.random.default_rng(7 + seed)
    njy = lib.njy()
    jy = rng.standard_normal(njy) * 3.0
    gamma = 0.21
    J = np.zeros((n, n))
    for i in range(n):
        for e, c in d.jy_rows[i]:
            J[i, c] = jy[e]
    Mx = np.eye(n) - gamma * J
    b0 = rng.standard_normal(n)
    x_ref = np.linalg.solve(Mx, b0)
    x = b0.copy()
    lib.build_factor_solve(gamma, _p(np.concatenate([jy, [0.0, 0.0]])), _p(x))
    assert np.allclose(x, x_ref, rtol=1e-11, atol=1e-13)
    # the table of the distributed form, filled as the row lanes do: 1 / M_ii at rstart(i), gamma J_ij / M_ii at mfpos
    mf = np.zeros(lib.im_mf() + 2)
    for i in range(n):
        mf[lib.rstart(i)] = 1.0 / Mx[i, i]
        for s, (e, c) in enumerate(d.jy_rows[i]):
            if c != i:
                mf[lib.mfpos(s, i)] = gamma * J[i, c] / Mx[i, i]
    x2 = b0.copy()
    lib.solve_tri(_p(mf), _p(x2))
    assert np.allclose(x2, x_ref, rtol=1e-11, atol=1e-13)
    rpl = (n + 63) // 64
    for lane in (0, 1, n // 2, min(n - 1, 63)):
        dd = np.zeros(rpl)
        lib.solve_tri_pick(_p(mf), _p(b0), lane, _p(dd))
        for r in range(rpl):
            if lane + 64 * r < n:
                assert dd[r] == x2[lane + 64 * r]


def _general_network(seed, n):
    """tests/test_gpu_user_models.py::_random_network: feedback, saturation, product inhibition -- not triangular"""
    from tests.test_gpu_user_models import _random_network
    return _random_network(seed, n)


@pytest.mark.parametrize('make,seed,n', [(_triangular_network, 3, 24), (_general_network, 2, 11), (_general_network, 4, 30)])
def test_row_lane_tables_reproduce_the_right_hand_side(tmp_path, make, seed, n):
    """The row-lane form every kernel evaluates (emit_rowlane.py: isomorphic equations share one class body, operands
    come through the tables SBM_RL_YS / PS, results go out through JYOUT / JYCOL / JPCOL) against the generated Python
    model the oracle integrates: f, every non-zero of J_y and of J_p, and nothing else."""
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(make(seed, n))
    lib = _host_library(gm, tmp_path)
    k, n_par = gm.n_sens, len(gm.param_order)
    assert lib.nv() == n and lib.nk() == k
    rng = np.random.default_rng(seed)
    y = rng.uniform(0.2, 2.0, n)
    p = rng.uniform(0.3, 3.0, n_par)
    t = 0.7
    f_ref = np.zeros(n)
    gm.model(y, t, f_ref, p)
    aug, out = np.zeros(n + n * k), np.zeros(n + n * k)
    aug[:n] = y
    gm.sens_model(aug, t, out, p)
    Jp_ref = out[n:].reshape(n, k)
    jac = np.zeros((n, n))
    gm.model_jac(y, t, jac, p)
    Jy_ref = jac.T                                   # jacout[b, a] = d f_a / d y_b
    maxjy, maxjp = lib.maxjy(), lib.maxjp()
    Jy, Jp = np.zeros((n, n)), np.zeros((n, k))
    f = np.zeros(n)
    for i in range(n):
        fi, jy, jp = np.zeros(1), np.zeros(maxjy), np.zeros(maxjp)
        lib.eval_row(i, t, _p(y), _p(p), _p(fi), _p(jy), _p(jp))
        f[i] = fi[0]
        for s in range(maxjy):
            c = lib.jycol(s, i)
            if c >= 0:
                Jy[i, c] += jy[s]
        for s in range(maxjp):
            c = lib.jpcol(s, i)
            if c >= 0:
                Jp[i, c] += jp[s]
    assert np.allclose(f, f_ref, rtol=1e-12, atol=1e-14)
    assert np.allclose(Jy, Jy_ref, rtol=1e-12, atol=1e-14)
    assert np.allclose(Jp, Jp_ref, rtol=1e-12, atol=1e-14)
    assert np.count_nonzero(Jy_ref) > n and np.count_nonzero(Jp_ref) >= n


@pytest.mark.parametrize('seed,n', [(2, 11), (3, 17), (4, 30)])
def test_symbolic_lu_with_fill_in_solves_the_newton_matrix(tmp_path, seed, n):
    """Networks with feedback: J_y is not triangular, the elimination worked out at generation time (symbolic_lu:
    natural order, fill-in entries added to the pattern) is straight-line code in im_factor / im_solve -- against numpy
    on (I - gamma J_y) x = b, for a mild and a stiff gamma."""
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(_general_network(seed, n))
    lib = _host_library(gm, tmp_path)
    assert lib.is_tri() == 0 and lib.has_fused() == 0
    d = gm.derived
    assert lib.im_nm() >= lib.njy()
    fill_in = lib.im_nm() - lib.njy()
    assert fill_in > 0 or n < 30, fill_in             # the largest network does have fill-in
    rng = np.random.default_rng(seed)
    jy = rng.standard_normal(lib.njy())
    J = np.zeros((n, n))
    for i in range(n):
        for e, c in d.jy_rows[i]:
            J[i, c] = jy[e]
    for gamma in (0.05, 40.0):
        Mx = np.eye(n) - gamma * J
        b0 = rng.standard_normal(n)
        x = b0.copy()
        lib.build_factor_solve(gamma, _p(np.concatenate([jy, [0.0, 0.0]])), _p(x))
        x_ref = np.linalg.solve(Mx, b0)
        # no pivoting (the Newton matrices of the implicit integrators are diagonally dominant for the steps they take;
        # a random J at gamma = 40 is not): judged by the residual relative to the conditioning
        scale = np.linalg.cond(Mx) * 1e-13
        assert np.allclose(x, x_ref, rtol=max(1e-10, scale), atol=max(1e-12, scale * np.abs(x_ref).max())), gamma


def _hub_network(n=14, fan_in=11):
    """a lower-triangular network whose last species is produced from ``fan_in`` earlier ones with one shared rate constant:
    one row of the table is wider than a whole block of im_sens_tri (8 doubles)"""
    import sympy
    from sysbio_modeling_amd.symbolic.emit import ModelSpec
    xs = [sympy.Symbol('x%d' % i) for i in range(n)]
    params, eq = [], OrderedDict()
    for i in range(n):
        params += ['d%d' % i, 'k%d' % i]
        d, k = sympy.Symbol('d%d' % i), sympy.Symbol('k%d' % i)
        if i == 0:
            rhs = k - d * xs[0]
        elif i < n - 1:
            rhs = k * xs[i - 1] - d * xs[i]
        else:
            rhs = k * sum(xs[j] for j in range(fan_in)) - d * xs[i]
        eq['x%d' % i] = rhs
    return ModelSpec(name='hub%d_%d' % (n, fan_in), variables=[str(x) for x in xs], params=params, equations=eq)


def test_fused_sensitivity_step_with_a_row_wider_than_a_block(tmp_path):
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(_hub_network())
    lib = _host_library(gm, tmp_path)
    n, n_mf, maxjp = lib.nv(), lib.im_mf(), lib.maxjp()
    assert lib.is_tri() == 1 and lib.has_fused() == 1
    wide = max(sum(1 for s in range(lib.maxjy()) if 0 <= lib.mfpos(s, i) < n_mf) for i in range(n))
    assert wide >= 11
    rng = np.random.default_rng(3)
    mf, ja = rng.standard_normal(n_mf + 2), rng.standard_normal(n * maxjp + 2)
    for col in range(lib.nk()):
        z0 = rng.standard_normal(n)
        a, b = z0.copy(), z0.copy()
        lib.two_pass(_p(mf), _p(ja), 0.11, col, _p(a))
        lib.fused(_p(mf), _p(ja), 0.11, col, _p(b))
        assert np.array_equal(a, b), col


@pytest.mark.parametrize('n,density', [(12, 1.0), (20, 0.5)])
def test_row_distributed_lu_in_lockstep_emulation(tmp_path, n, density):
    """Densely coupled networks get the row-DISTRIBUTED factorisation (emit_implicit.py::emit_distributed: lane i eliminates
    in row i, pivot rows travel by SBM_LANE_BCAST, the rows are published to an LDS table that im_solve_lds reads).  Lane-level
    code, run here one lane at a time with the broadcasts learnt from their source lanes (harness above): the factors of
    (I - gamma J_y) for a random dense J_y solve the system numpy solves; the redundant form (im_factor / im_solve, every
    lane the whole matrix) gives the same solution."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel
    gm = GeneratedModel(models_zoo.dense_spec(n, density, name='dense%d_host%d' % (n, int(100 * density))))
    lib = _host_library(gm, tmp_path)
    assert lib.is_dist() == 1 and lib.is_tri() == 0
    d = gm.derived
    rng = np.random.default_rng(n)
    jy = rng.standard_normal(lib.njy())
    J = np.zeros((n, n))
    for i in range(n):
        for e, c in d.jy_rows[i]:
            J[i, c] = jy[e]
    gamma = 0.07
    Mx = np.eye(n) - gamma * J
    b0 = rng.standard_normal(n)
    x = b0.copy()
    n_bc = lib.dist_factor_solve(_p(np.ascontiguousarray(Mx)), _p(x))
    assert n_bc > n                                              # pivots + the pivot rows' entries
    x_ref = np.linalg.solve(Mx, b0)
    assert np.allclose(x, x_ref, rtol=1e-11, atol=1e-13)
    x2 = b0.copy()
    lib.build_factor_solve(gamma, _p(np.concatenate([jy, [0.0, 0.0]])), _p(x2))
    assert np.allclose(x2, x, rtol=1e-12, atol=1e-14)
