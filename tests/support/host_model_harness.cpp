// Host harness for a generated model header: compiles SbmModel with g++ (device
// qualifiers stubbed out) and exposes the augmented RHS exactly as the sensitivity
// kernel assembles it (eval_jac once, apply_col per column, state in column 0).
// Used by tests/test_symbolic.py to pin the HIP text to the Python/C emitters on CPU.
#include <cmath>
#define __device__
#define __constant__ static const
#define __forceinline__ inline
#define SBM_RCP(x) (1.0 / (x))
#define SBM_PICK(scol, c, v, otherwise) ((scol) == (c) ? (v) : (otherwise))
#define SBM_SEL(c, a, b) ((c) ? (a) : (b))
using std::fma;
#include SBM_MODEL_HEADER

extern "C" {
int h_n_vars() { return SbmModel::NV; }
int h_n_params() { return SbmModel::NP; }
int h_n_sens() { return SbmModel::NK; }

void h_rhs(const double* y, double t, double* yout, const double* p) {
  double yy[SbmModel::NV], f[SbmModel::NV];
  for (int i = 0; i < SbmModel::NV; ++i) yy[i] = y[i];
  SbmModel::eval_f(t, yy, p, f);
  for (int i = 0; i < SbmModel::NV; ++i) yout[i] = f[i];
}

// y: [n + n*k] in the reference layout (state, then S state-major / param-minor)
void h_sens_rhs(const double* y, double t, double* yout, const double* p) {
  constexpr int N = SbmModel::NV, K = SbmModel::NK;
  double yy[N], f[N], jy[SbmModel::NJY], jp[SbmModel::NJP];
  for (int i = 0; i < N; ++i) yy[i] = y[i];
  SbmModel::eval_jac(t, yy, p, f, jy, jp);
  for (int i = 0; i < N; ++i) yout[i] = f[i];
  for (int j = 0; j < K; ++j) {
    double z[N], dz[N];
    for (int i = 0; i < N; ++i) z[i] = y[N + i * K + j];
    SbmModel::apply_col(jy, jp, j, z, dz);
    for (int i = 0; i < N; ++i) yout[N + i * K + j] = dz[i];
  }
}

// the same through the fused per-row form the DOPRI/RK4 kernels call (eval_col)
void h_sens_rhs_fused(const double* y, double t, double* yout, const double* p) {
  constexpr int N = SbmModel::NV, K = SbmModel::NK;
  double yy[N], z[N], dz[N];
  for (int i = 0; i < N; ++i) yy[i] = y[i];
  SbmModel::eval_col(t, yy, p, -1, yy, dz);
  for (int i = 0; i < N; ++i) yout[i] = dz[i];
  for (int j = 0; j < K; ++j) {
    for (int i = 0; i < N; ++i) z[i] = y[N + i * K + j];
    SbmModel::eval_col(t, yy, p, j, z, dz);
    for (int i = 0; i < N; ++i) yout[N + i * K + j] = dz[i];
  }
}

// the row-lane form: every row lane evaluates its class body on its own operands and drops the
// results into the J_y list / additive matrix; every column then runs apply_rowlane
void h_sens_rhs_rowlane(const double* y, double t, double* yout, const double* p) {
  constexpr int N = SbmModel::NV, K = SbmModel::NK;
  static double ash[N * 64 + 2];
  double jysh[SbmModel::NJY + 2], z[N], dz[N];
  for (int i = 0; i < N * 64 + 2; ++i) ash[i] = 0.0;
  for (int row = 0; row < N; ++row) {
    double ys[SbmModel::RL_MAXYS], ps[SbmModel::RL_MAXPS], f = 0.0, jy[SbmModel::RL_MAXJY], jp[SbmModel::RL_MAXJP];
    for (int s = 0; s < SbmModel::RL_MAXYS; ++s) ys[s] = y[SbmModel::rl_ys(s, row)];
    for (int s = 0; s < SbmModel::RL_MAXPS; ++s) ps[s] = p[SbmModel::rl_ps(s, row)];
    for (int s = 0; s < SbmModel::RL_MAXJY; ++s) jy[s] = 0.0;
    for (int s = 0; s < SbmModel::RL_MAXJP; ++s) jp[s] = 0.0;
    SbmModel::class_dispatch(SbmModel::rl_class(row), t, ys, ps, f, jy, jp);
    for (int s = 0; s < SbmModel::RL_MAXJY; ++s) jysh[SbmModel::rl_jyout(s, row)] = jy[s];
    for (int s = 0; s < SbmModel::RL_MAXJP; ++s) ash[SbmModel::rl_apos(s, row)] = jp[s];
    yout[row] = f;
  }
  for (int c = 0; c < K; ++c) {
    for (int i = 0; i < N; ++i) z[i] = y[N + i * K + c];
    SbmModel::apply_rowlane(jysh, ash + c, z, dz);
    for (int i = 0; i < N; ++i) yout[N + i * K + c] = dz[i];
  }
}
}
