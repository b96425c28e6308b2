// Host harness for a generated model header: compiles SbmModel with g++ (device
// qualifiers stubbed out) and exposes the augmented RHS exactly as the sensitivity
// kernels assemble it, in each of the emitted forms:
//   h_sens_rhs          eval_jac once, apply_col per column, state in column 0   (per-wave kernel, CPL > 1)
//   h_sens_rhs_fused    eval_col per column                                      (per-wave kernel, CPL == 1)
//   h_sens_rhs_rowlane  class_dispatch per row lane + apply_rowlane per column   (row-lane kernel)
// Used by tests/test_symbolic.py to pin the HIP text to the Python/C emitters on CPU.
#include <cmath>
#define __device__
#define __constant__ static const
#define __forceinline__ inline
#define SBM_RCP(x) (1.0 / (x))
#define SBM_PICK(scol, c, v, otherwise) ((scol) == (c) ? (v) : (otherwise))
#define SBM_SEL(c, a, b) ((c) ? (a) : (b))
// On the device SBM_LANE_BCAST(jy[slot], lane) reads register jy[slot] of another lane.  Here the
// "lane registers" are the table g_lane_jy[lane][slot]; `v` names jy[slot] of an array whose base
// address is g_bcast_base, which recovers the slot.
static const double* g_bcast_base = nullptr;
static double g_lane_jy[64][8];
static inline double h_lane_bcast(const double* which, int src) { return g_lane_jy[src][which - g_bcast_base]; }
#define SBM_LANE_BCAST(v, src) h_lane_bcast(&(v), (src))
using std::fma;
#include SBM_MODEL_HEADER

extern "C" {
int h_n_vars() { return SbmModel::NV; }
int h_n_params() { return SbmModel::NP; }
int h_n_sens() { return SbmModel::NK; }
int h_n_classes() { return SbmModel::RL_NCLASS; }

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

void h_sens_rhs_rowlane(const double* y, double t, double* yout, const double* p) {
  constexpr int N = SbmModel::NV, K = SbmModel::NK;
  static_assert(SbmModel::RL_MAXJY <= 8, "harness table too small");
  static double ash[N * 64 + 2];
  double z[N], dz[N], acol[N];
  for (int i = 0; i < N * 64 + 2; ++i) ash[i] = 0.0;
  for (int row = 0; row < N; ++row) {
    double ys[SbmModel::RL_MAXYS], ps[SbmModel::RL_MAXPS], f = 0.0, jy[SbmModel::RL_MAXJY], jp[SbmModel::RL_MAXJP];
    for (int s = 0; s < SbmModel::RL_MAXYS; ++s) ys[s] = y[SbmModel::rl_ys(s, row)];
    for (int s = 0; s < SbmModel::RL_MAXPS; ++s) ps[s] = p[SbmModel::rl_ps(s, row)];
    for (int s = 0; s < SbmModel::RL_MAXJY; ++s) jy[s] = 0.0;
    for (int s = 0; s < SbmModel::RL_MAXJP; ++s) jp[s] = 0.0;
    SbmModel::class_dispatch(SbmModel::rl_class(row), t, ys, ps, f, jy, jp);
    for (int s = 0; s < SbmModel::RL_MAXJY; ++s) g_lane_jy[row][s] = jy[s];
    for (int s = 0; s < SbmModel::RL_MAXJP; ++s) ash[SbmModel::rl_apos(s, row)] = jp[s];
    yout[row] = f;
  }
  double jy_names[SbmModel::RL_MAXJY] = {0};  // only its addresses matter (slot recovery)
  g_bcast_base = jy_names;
  // parameter-only J_y entries are broadcast once per kernel (here: from the same lane table)
  double sj[SbmModel::RL_NSTATIC > 0 ? SbmModel::RL_NSTATIC : 1];
  SbmModel::rl_static(jy_names, sj);
  for (int c = 0; c < K; ++c) {
    for (int i = 0; i < N; ++i) { z[i] = y[N + i * K + c]; acol[i] = ash[i * 64 + c]; }
    SbmModel::apply_rowlane(jy_names, sj, acol, z, dz);
    for (int i = 0; i < N; ++i) yout[N + i * K + c] = dz[i];
  }
}
}
