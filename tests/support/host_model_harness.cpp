// Host harness for a generated model header: compiles SbmModel with g++ (device
// qualifiers stubbed out) and exposes the augmented RHS exactly as the sensitivity
// kernels assemble it, in each of the emitted forms:
//   h_sens_rhs          eval_jac once, apply_col per column, state in column 0   (per-wave kernel, CPL > 1)
//   h_sens_rhs_fused    eval_col per column                                      (per-wave kernel, CPL == 1)
//   h_sens_rhs_rowlane  class_dispatch per row lane + apply_rowlane per column   (row-lane kernel)
//   h_sens_rhs_rowgroup class_dispatch per row lane + publish/apply_rowgroup per lane (g, c')
//                       (row-group kernel; returns -1 when the model has no row-group form)
// Used by tests/test_symbolic.py to pin the HIP text to the Python/C emitters on CPU.
#include <cmath>
#include <type_traits>
#define __device__
#define __constant__ static const
#define __forceinline__ inline
#define SBM_RCP(x) (1.0 / (x))
#define SBM_LDS_FENCE() ((void)0)
#define SBM_PICK(scol, c, v, otherwise) ((scol) == (c) ? (v) : (otherwise))
#define SBM_SEL(c, a, b) ((c) ? (a) : (b))
#define SBM_PICK_COL(col, c, v, otherwise) ((col) == (c) ? (v) : (otherwise))
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

// The row-group kernel's right-hand side, lane by lane: LDS tables are plain arrays here, the
// per-lane base pointers / halo offsets are set up exactly as sbm_sens_rowgroup_kernel does.
}  // extern "C"

template <class L>
static int sens_rhs_rowgroup(const double* y, double t, double* yout, const double* p) {
  using M = SbmModel;
  if constexpr (!L::RG_OK) {
    (void)y; (void)t; (void)yout; (void)p;
    return -1;
  } else {
    constexpr int N = M::NV, K = M::NK, G = L::RG_G, C = L::RG_C, CPL = L::RG_CPL, RPG = L::RG_RPG, JYS = L::RG_JYS;
    constexpr int NPAD = G * RPG, NROWS = NPAD + RPG, NE = RPG * CPL;
    constexpr int NH = L::RG_NHALO > 0 ? L::RG_NHALO : 1;
    constexpr int LS = L::RG_LS, ZPOS = RPG * LS;
    static double A[RPG * LS + 2], H[RPG * LS + 4], JYL[NROWS * JYS + 2];
    // one pass per column chunk (on the device: one wavefront each, blockIdx.y)
    for (int chunk = 0; chunk < L::RG_NCH; ++chunk) {
    const int cbase = chunk * C * CPL;
    for (double& v : A) v = 0.0;
    for (double& v : H) v = 0.0;
    for (double& v : JYL) v = 0.0;
    for (int row = 0; row < N; ++row) {
      double ys[M::RL_MAXYS], ps[M::RL_MAXPS], f = 0.0, jy[M::RL_MAXJY], jp[M::RL_MAXJP];
      for (int s = 0; s < M::RL_MAXYS; ++s) ys[s] = y[M::rl_ys(s, row)];
      for (int s = 0; s < M::RL_MAXPS; ++s) ps[s] = p[M::rl_ps(s, row)];
      for (int s = 0; s < M::RL_MAXJY; ++s) jy[s] = 0.0;
      for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = 0.0;
      M::class_dispatch(M::rl_class(row), t, ys, ps, f, jy, jp);
      for (int s = 0; s < M::RL_MAXJP; ++s) {
        const int lc = M::rl_jpcol(s, row) - cbase;
        A[(lc >= 0 && lc < C * CPL && lc + cbase < K) ? L::rg_pos(row, lc) : RPG * LS + 1] = jp[s];
      }
      for (int s = 0; s < M::RL_MAXJY; ++s) {
        const int jpz = L::rg_jypos(s, row);
        JYL[jpz < NPAD * JYS ? jpz : NROWS * JYS + 1] = jy[s];
      }
      yout[row] = f;
    }
    double z[64][NE], dz[64][NE];
    for (int lane = 0; lane < 64; ++lane) {   // every lane publishes before any lane reads
      const bool active = lane < G * C;
      const int g = active ? lane / C : G, cp = lane - g * C;
      for (int cc = 0; cc < CPL; ++cc)
        for (int r = 0; r < RPG; ++r) {
          const int grow = g * RPG + r, col = cbase + cp + C * cc;
          z[lane][r + RPG * cc] = (active && grow < N && col < K) ? y[N + grow * K + col] : 0.0;
        }
      L::publish_rowgroup(H + CPL * lane, z[lane]);
    }
    for (int lane = 0; lane < 64; ++lane) {
      const bool active = lane < G * C;
      const int g = active ? lane / C : G, cp = lane - g * C;
      int hoff[NH];
      for (int tt = 0; tt < NH; ++tt) {
        const int src = (L::RG_NHALO > 0 && active) ? L::rg_hsrc(tt, g) : NPAD;
        hoff[tt] = src < NPAD ? L::rg_pos(src, cp) : ZPOS;
      }
      double acol[NE], coef[RPG * JYS];
      L::load_rowgroup(A + CPL * lane, JYL + (g * RPG) * JYS, acol, coef);
      L::apply_rowgroup(acol, coef, H, hoff, z[lane], dz[lane]);
      for (int cc = 0; cc < CPL; ++cc)
        for (int r = 0; r < RPG; ++r) {
          const int grow = g * RPG + r, col = cbase + cp + C * cc;
          if (active && grow < N && col < K) yout[N + grow * K + col] = dz[lane][r + RPG * cc];
          else if (dz[lane][r + RPG * cc] != 0.0) return -2;   // padding must stay exactly zero
        }
    }
    }
    return 0;
  }
}

extern "C" {
// throughput layout (RG0) and small-batch layout (RG1: more, smaller column chunks; may be the same type)
int h_sens_rhs_rowgroup(const double* y, double t, double* yout, const double* p) {
  return sens_rhs_rowgroup<SbmModel::RG0>(y, t, yout, p);
}
int h_sens_rhs_rowgroup_small_batch(const double* y, double t, double* yout, const double* p) {
  return sens_rhs_rowgroup<SbmModel::RG1>(y, t, yout, p);
}
int h_rowgroup_layouts_differ() { return std::is_same<SbmModel::RG0, SbmModel::RG1>::value ? 0 : 1; }


// b <- (I - gamma*J_y(y))^-1 b through the generated sparse LU (im_build / im_factor / im_solve)
void h_im_solve(double gamma, const double* y, double t, const double* p, double* b) {
  using M = SbmModel;
  double yy[M::NV], f[M::NV], jy[M::NJY], jp[M::NJP], m[M::IM_NM], bb[M::NV];
  for (int i = 0; i < M::NV; ++i) { yy[i] = y[i]; bb[i] = b[i]; }
  M::eval_jac(t, yy, p, f, jy, jp);
  M::im_build(gamma, jy, m);
  M::im_factor(m);
  M::im_solve(m, bb);
  for (int i = 0; i < M::NV; ++i) b[i] = bb[i];
}

// The distributed triangular form (IM_TRI): every row scales itself (what row lane i does in the
// kernel), then the forward substitution reads the table.  Returns -1 when the model has no such form.
int h_im_solve_tri(double gamma, const double* y, double t, const double* p, double* b) {
  using M = SbmModel;
  if constexpr (!M::IM_TRI) {
    (void)gamma; (void)y; (void)t; (void)p; (void)b;
    return -1;
  } else {
    static double MF[M::IM_NM + 2];
    for (double& v : MF) v = 0.0;
    for (int row = 0; row < M::NV; ++row) {
      double ys[M::RL_MAXYS], ps[M::RL_MAXPS], f = 0.0, jy[M::RL_MAXJY], jp[M::RL_MAXJP];
      for (int s = 0; s < M::RL_MAXYS; ++s) ys[s] = y[M::rl_ys(s, row)];
      for (int s = 0; s < M::RL_MAXPS; ++s) ps[s] = p[M::rl_ps(s, row)];
      for (int s = 0; s < M::RL_MAXJY; ++s) jy[s] = 0.0;
      for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = 0.0;
      M::class_dispatch(M::rl_class(row), t, ys, ps, f, jy, jp);
      double jd = 0.0;
      for (int s = 0; s < M::RL_MAXJY; ++s) if (M::im_diagslot(row) == s) jd = jy[s];
      const double rd = 1.0 / (1.0 - gamma * jd);
      MF[M::im_rstart(row)] = rd;
      for (int s = 0; s < M::RL_MAXJY; ++s) MF[M::im_mfpos(s, row)] = gamma * jy[s] * rd;
    }
    double bb[M::NV];
    for (int i = 0; i < M::NV; ++i) bb[i] = b[i];
    M::im_solve_tri(MF, bb);
    for (int i = 0; i < M::NV; ++i) b[i] = bb[i];
    return 0;
  }
}
}
