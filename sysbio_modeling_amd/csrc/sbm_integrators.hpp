// sbm_integrators.hpp -- ODE / forward-sensitivity integrators for gfx950 (CDNA4).
//
// Replaces the per-trajectory hot loop of the reference:
//   OdeModel.simulate       (model/ode_model.py:128-169)  -> state kernels
//   OdeModel.calc_jacobian  (model/ode_model.py:83-126)   -> sensitivity kernels
// which there is scipy.integrate.odeint (LSODA, Fortran) calling a Python RHS.
//
// Two mappings, chosen per kernel kind:
//
//  * sensitivity kernels: ONE TRAJECTORY PER WAVEFRONT.  The augmented state is
//    the n x (1+k) matrix Z = [ y | S ]; lane c of the wave owns column c
//    (lane 0 the state, lane 1+j the sensitivities w.r.t. parameter j) and keeps
//    all n rows of it, for every Runge-Kutta stage, in VGPRs.  A column evolves
//    as  z_c' = J_y(y) z_c + J_p[:, c]  and lane 0 as y' = f(y); y is broadcast
//    from lane 0 (v_readfirstlane -> SGPRs), J_y / J_p / f are evaluated once per
//    stage from scalar operands, and the sparse per-column product is unrolled
//    with static indices (SbmModel::apply_col).  No LDS, no HBM traffic inside
//    the step loop; step-size control is wave-uniform (no divergence).
//
//  * state-only kernels: ONE TRAJECTORY PER LANE (n is tiny; there is nothing to
//    spread over a wave).  Parameters are loaded coalesced and parked in LDS,
//    transposed to [param][lane] so that every access is conflict-free.
//
// The generated model header must define `struct SbmModel` (see
// sysbio_modeling_amd/symbolic/emit.py::emit_hip) and may use SBM_RCP.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sbm_plugin.h"

// ---------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------
// 1/x: v_rcp_f64 (~24 good bits on gfx950... refined by two Newton steps to < 1 ulp-ish).
// The IEEE division sequence (div_scale/div_fmas/div_fixup) costs about twice as much
// and the RHS of rate-law models is dominated by reciprocals.
#ifndef SBM_RCP_NR
#define SBM_RCP_NR 2
#endif
__device__ __forceinline__ double sbm_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int it = 0; it < SBM_RCP_NR; ++it) {
    const double e = fma(-x, r, 1.0);
    r = fma(e, r, r);
  }
  return r;
}
#define SBM_RCP(x) sbm_rcp(x)
// compiler-level ordering point for memory operations (no instruction): generated code uses it to bound
// how far loads are hoisted
#define SBM_LDS_FENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)

// J_p[row, scol] picked per lane.  Arguments BY VALUE: the candidates are computed
// unconditionally and this lowers to v_cndmask -- written as a ?: chain over array
// elements hipcc sinks the J_p arithmetic into exec-masked branches (one per non-zero).
__device__ __forceinline__ double sbm_pick(int scol, int c, double v, double otherwise) {
  return scol == c ? v : otherwise;
}
#define SBM_PICK(scol, c, v, otherwise) sbm_pick(scol, c, v, otherwise)
__device__ __forceinline__ double sbm_sel(bool c, double a, double b) { return c ? a : b; }
#define SBM_SEL(c, a, b) sbm_sel(c, a, b)
// (col == c ? v : otherwise) for col = lane + 64 * chunk (chunk wave-uniform) and a literal c: the lane mask comes from
// scalar instructions (the inverse of a ballot: s_mov / s_cselect of the literal), the select is the v_cndmask pair alone
__device__ __forceinline__ double sbm_pick_col(int col, int c, double v, double otherwise) {
  const int chunk = __builtin_amdgcn_readfirstlane(col) >> 6;       // lane 0 holds 64 * chunk
  const unsigned long long mask = (c >> 6) == chunk ? (1ull << (c & 63)) : 0ull;
  return __builtin_amdgcn_inverse_ballot_w64(mask) ? v : otherwise;
}
#define SBM_PICK_COL(col, c, v, otherwise) sbm_pick_col(col, c, v, otherwise)
// value of `v` in lane `src` (compile-time constant) as a wave-uniform scalar: two v_readlane_b32
__device__ __forceinline__ double sbm_lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
#define SBM_LANE_BCAST(v, src) sbm_lane_bcast(v, src)

__device__ __forceinline__ double sbm_bcast0(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float sbm_bcast0f(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// Wave-wide reductions of the step controller, on DPP (data-parallel primitives: a VALU operand
// taken from another lane of the same row of 16, no LDS crossbar round trip).  The __shfl_xor
// butterflies they replace cost one ds_bpermute + s_waitcnt lgkmcnt(0) per level, ~20 dependent
// LDS round trips per step over the controller's reductions -- at one wave per SIMD nothing
// hides them.  Sequence as in rocPRIM's warp_reduce_dpp: two quad_perms and two row rotations leave
// every lane of a row with the row's result, row_bcast:15 / :31 fold the four rows into lane 63.
// Lanes of rows a row_mask disables read 0: fine for sums and for maxima of values >= 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sbm_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float sbm_lane63f(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// max over the wave of v >= 0 (NaN-free: callers map NaN to +inf first).  Non-negative floats order as their bit
// patterns do, and an integer max takes the DPP operand itself: one v_max_i32_dpp per level where fmaxf costs a
// v_mov_b32 for the fill, the DPP move, a canonicalising v_max_f32 and the max (6 instead of 30 instructions; the Newton
// loops of the implicit kernels reduce once per iteration).  INT_MIN is the identity the masked-off rows keep.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int sbm_dpp_smax(int x) {
  const int o = __builtin_amdgcn_update_dpp((int)0x80000000, x, CTRL, ROW_MASK, 0xf, false);
  return x > o ? x : o;
}
__device__ __forceinline__ float sbm_wave_max(float v) {
  int x = __float_as_int(v);
  x = sbm_dpp_smax<0xb1, 0xf>(x);    // quad_perm:[1,0,3,2]
  x = sbm_dpp_smax<0x4e, 0xf>(x);    // quad_perm:[2,3,0,1]
  x = sbm_dpp_smax<0x124, 0xf>(x);   // row_ror:4
  x = sbm_dpp_smax<0x128, 0xf>(x);   // row_ror:8
  x = sbm_dpp_smax<0x142, 0xa>(x);   // row_bcast:15 -> rows 1, 3
  x = sbm_dpp_smax<0x143, 0xc>(x);   // row_bcast:31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ float sbm_wave_sumf(float v) {
  v += sbm_dpp<0xb1, 0xf>(v);
  v += sbm_dpp<0x4e, 0xf>(v);
  v += sbm_dpp<0x124, 0xf>(v);
  v += sbm_dpp<0x128, 0xf>(v);
  v += sbm_dpp<0x142, 0xa>(v);
  v += sbm_dpp<0x143, 0xc>(v);
  return sbm_lane63f(v);
}
// a failed evaluation (NaN) must surface as a failed step: +inf survives max and sum
__device__ __forceinline__ float sbm_nan_to_inf(float v) { return (v != v) ? __builtin_inff() : v; }
__device__ __forceinline__ double sbm_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return sbm_bcast0(v);
}

// ---------------------------------------------------------------------------
// Dormand-Prince 5(4) tableau (Hairer, Norsett, Wanner, vol. I, table 5.2)
// ---------------------------------------------------------------------------
namespace dp {
constexpr double C2 = 1.0 / 5.0, C3 = 3.0 / 10.0, C4 = 4.0 / 5.0, C5 = 8.0 / 9.0;
constexpr double A21 = 1.0 / 5.0;
constexpr double A31 = 3.0 / 40.0, A32 = 9.0 / 40.0;
constexpr double A41 = 44.0 / 45.0, A42 = -56.0 / 15.0, A43 = 32.0 / 9.0;
constexpr double A51 = 19372.0 / 6561.0, A52 = -25360.0 / 2187.0, A53 = 64448.0 / 6561.0, A54 = -212.0 / 729.0;
constexpr double A61 = 9017.0 / 3168.0, A62 = -355.0 / 33.0, A63 = 46732.0 / 5247.0, A64 = 49.0 / 176.0,
                 A65 = -5103.0 / 18656.0;
constexpr double A71 = 35.0 / 384.0, A73 = 500.0 / 1113.0, A74 = 125.0 / 192.0, A75 = -2187.0 / 6784.0,
                 A76 = 11.0 / 84.0;
constexpr double E1 = 71.0 / 57600.0, E3 = -71.0 / 16695.0, E4 = 71.0 / 1920.0, E5 = -17253.0 / 339200.0,
                 E6 = 22.0 / 525.0, E7 = -1.0 / 40.0;
}  // namespace dp

// ---------------------------------------------------------------------------
// Dormand-Prince 8(5,3) tableau (Hairer, Norsett, Wanner, vol. I, section II.5: DOP853), 12 stages + FSAL;
// A<i>_<j> multiplies k_j in the argument of stage i (1-based), B<j> forms the 8th-order solution, E5_<j> / E3_<j>
// the two embedded error estimates the step controller combines.  Entries that are zero are left out.
// ---------------------------------------------------------------------------
namespace dp8 {
constexpr double C2 = 0.05260015195876773, C3 = 0.0789002279381516, C4 = 0.1183503419072274, C5 = 0.2816496580927726, C6 = 0.3333333333333333, C7 = 0.25, C8 = 0.3076923076923077, C9 = 0.6512820512820513, C10 = 0.6, C11 = 0.8571428571428571, C12 = 1.0;
constexpr double A2_1 = 0.05260015195876773;
constexpr double A3_1 = 0.0197250569845379, A3_2 = 0.0591751709536137;
constexpr double A4_1 = 0.02958758547680685, A4_3 = 0.08876275643042054;
constexpr double A5_1 = 0.2413651341592667, A5_3 = -0.8845494793282861, A5_4 = 0.924834003261792;
constexpr double A6_1 = 0.037037037037037035, A6_4 = 0.17082860872947386, A6_5 = 0.12546768756682242;
constexpr double A7_1 = 0.037109375, A7_4 = 0.17025221101954405, A7_5 = 0.06021653898045596, A7_6 = -0.017578125;
constexpr double A8_1 = 0.03709200011850479, A8_4 = 0.17038392571223998, A8_5 = 0.10726203044637328, A8_6 = -0.015319437748624402, A8_7 = 0.008273789163814023;
constexpr double A9_1 = 0.6241109587160757, A9_4 = -3.3608926294469414, A9_5 = -0.868219346841726, A9_6 = 27.59209969944671, A9_7 = 20.154067550477894, A9_8 = -43.48988418106996;
constexpr double A10_1 = 0.47766253643826434, A10_4 = -2.4881146199716677, A10_5 = -0.590290826836843, A10_6 = 21.230051448181193, A10_7 = 15.279233632882423, A10_8 = -33.28821096898486, A10_9 = -0.020331201708508627;
constexpr double A11_1 = -0.9371424300859873, A11_4 = 5.186372428844064, A11_5 = 1.0914373489967295, A11_6 = -8.149787010746927, A11_7 = -18.52006565999696, A11_8 = 22.739487099350505, A11_9 = 2.4936055526796523, A11_10 = -3.0467644718982196;
constexpr double A12_1 = 2.273310147516538, A12_4 = -10.53449546673725, A12_5 = -2.0008720582248625, A12_6 = -17.9589318631188, A12_7 = 27.94888452941996, A12_8 = -2.8589982771350235, A12_9 = -8.87285693353063, A12_10 = 12.360567175794303, A12_11 = 0.6433927460157636;
constexpr double B1 = 0.054293734116568765, B6 = 4.450312892752409, B7 = 1.8915178993145003, B8 = -5.801203960010585, B9 = 0.3111643669578199, B10 = -0.1521609496625161, B11 = 0.20136540080403034, B12 = 0.04471061572777259;
constexpr double E5_1 = 0.01312004499419488, E5_6 = -1.2251564463762044, E5_7 = -0.4957589496572502, E5_8 = 1.6643771824549864, E5_9 = -0.35032884874997366, E5_10 = 0.3341791187130175, E5_11 = 0.08192320648511571, E5_12 = -0.022355307863886294;
constexpr double E3_1 = -0.18980075407240762, E3_6 = 4.450312892752409, E3_7 = 1.8915178993145003, E3_8 = -5.801203960010585, E3_9 = -0.4226823213237919, E3_10 = -0.1521609496625161, E3_11 = 0.20136540080403034, E3_12 = 0.02265179219836082;
}  // namespace dp8

// every element a lane integrates: CPL columns of NV rows, plus NX extra scalars per lane
// (row-lane kernel: the lane's own state component)
#define SBM_ALL(c, i)                       \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) \
  _Pragma("unroll") for (int i = 0; i < NVX; ++i)
// the NV column rows / the NX extra elements (the stage STATE in the row-lane / row-group kernels)
#define SBM_MAIN(c, i)                      \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) \
  _Pragma("unroll") for (int i = 0; i < NV; ++i)
#define SBM_EXTRA(c, i)                     \
  _Pragma("unroll") for (int c = 0; c < CPL; ++c) \
  _Pragma("unroll") for (int i = NV; i < NVX; ++i)
// A stage runs in three phases:
//   issue  : form the extra elements, make them visible (LDS) and START fetching the row operands;
//   eval   : evaluate f / J_y / J_p of the lane's row, publish them; the extra elements of the
//            stage derivative (kout) are known from here on;
//   finish : form the column rows of the stage vector and their derivative.
// The state path (issue, eval) does not depend on the column rows, so the drivers software-pipeline
// it: stage s+1's issue is placed between eval and finish of stage s (SBM_STAGE_THEN's `next`),
// and the LDS round trip of its operands is covered by the column work of stage s.
// `stmt` forms element [c][i] of the stage vector zt (and may update running sums next to it); it is
// run once for the extra elements (issue) and once for the column rows (finish side).
#define SBM_ISSUE(tt, stmt)                 \
  {                                         \
    SBM_EXTRA(c, i) { stmt }                \
    pend_ = sys.issue((tt), zt);            \
  }
#define SBM_STAGE_THEN(tt, stmt, kout, next) \
  {                                         \
    auto tok_ = sys.eval(pend_, (tt));      \
    sys.extra_out(tok_, kout);              \
    next                                    \
    SBM_MAIN(c, i) { stmt }                 \
    sys.finish(tok_, (tt), zt, kout);       \
  }
#define SBM_STAGE(tt, stmt, kout) SBM_ISSUE(tt, stmt) SBM_STAGE_THEN(tt, stmt, kout, )

// ---------------------------------------------------------------------------
// "System" policies: what differs between the two mappings
// ---------------------------------------------------------------------------
struct SbmLdsParams {  // [param][lane] image in LDS
  const double* base;
  __device__ __forceinline__ double operator[](int c) const { return base[c * 64]; }
};

// one trajectory per wave; lane = column of [y | S]
template <class M>
struct SensSystem {
  static constexpr int NV = M::NV;
  static constexpr int NVX = M::NV;   // no extra per-lane elements
  static constexpr int NCOL = 1 + M::NK;
  static constexpr int CPL = (NCOL + 63) / 64;
  static constexpr int NCS = CPL;
  __device__ __forceinline__ static constexpr int col_of(int c, int) { return c; }
  static constexpr bool kUniform = true;
  const double* __restrict__ p;  // wave-uniform -> scalar loads
  int lane;

  __device__ __forceinline__ void rhs(double t, const double (&z)[CPL][NV], double (&dz)[CPL][NV]) const {
    double y[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) y[i] = sbm_bcast0(z[0][i]);
    if constexpr (CPL == 1) {
      // fused, row by row: J entries are consumed as they are produced
      M::eval_col(t, y, p, lane - 1, z[0], dz[0]);
    } else {
      double f[NV], jy[M::NJY], jp[M::NJP];
      M::eval_jac(t, y, p, f, jy, jp);
#pragma unroll
      for (int c = 0; c < CPL; ++c) M::apply_col(jy, jp, lane + 64 * c - 1, z[c], dz[c]);
      const bool state_lane = (lane == 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) dz[0][i] = sbm_sel(state_lane, f[i], dz[0][i]);
    }
  }
  // error norm: max over columns of the column's RMS (every column, the state
  // included, individually meets the tolerance -- the CVODES-style sens. test)
  __device__ __forceinline__ float norm(const float (&colsum)[CPL], float /*xsum*/) const {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) m = fmaxf(m, sbm_nan_to_inf(colsum[c]));
    return sqrtf(sbm_wave_max(m) * (1.0f / NV));   // +inf: the driver rejects the step
  }
  __device__ __forceinline__ double sum(double v) const { return sbm_wave_sum(v); }
  struct Pending {};
  __device__ __forceinline__ Pending issue(double, const double (&)[CPL][NV]) const { return Pending{}; }
  __device__ __forceinline__ int eval(const Pending&, double) const { return 0; }
  __device__ __forceinline__ void extra_out(int, double (&)[CPL][NV]) const {}
  __device__ __forceinline__ void finish(int, double t, const double (&z)[CPL][NV], double (&dz)[CPL][NV]) const {
    rhs(t, z, dz);
  }
};

// one trajectory per lane; state only
template <class M>
struct StateSystem {
  static constexpr int NV = M::NV;
  static constexpr int NVX = M::NV;
  static constexpr int CPL = 1;
  static constexpr int NCS = 1;
  __device__ __forceinline__ static constexpr int col_of(int, int) { return 0; }
  static constexpr bool kUniform = false;
  SbmLdsParams p;

  __device__ __forceinline__ void rhs(double t, const double (&z)[1][NV], double (&dz)[1][NV]) const {
    M::eval_f(t, z[0], p, dz[0]);
  }
  __device__ __forceinline__ float norm(const float (&colsum)[1], float /*xsum*/) const {
    return sqrtf(colsum[0] * (1.0f / NV));
  }
  __device__ __forceinline__ double sum(double v) const { return v; }
  struct Pending {};
  __device__ __forceinline__ Pending issue(double, const double (&)[1][NV]) const { return Pending{}; }
  __device__ __forceinline__ int eval(const Pending&, double) const { return 0; }
  __device__ __forceinline__ void extra_out(int, double (&)[1][NV]) const {}
  __device__ __forceinline__ void finish(int, double t, const double (&z)[1][NV], double (&dz)[1][NV]) const {
    rhs(t, z, dz);
  }
};

// ---------------------------------------------------------------------------
// per-trajectory driver state shared by both integrators
// ---------------------------------------------------------------------------
struct SbmTrajOut {
  int32_t status;
  int32_t n_acc;
  int32_t n_rej;
};

// Dormand-Prince 5(4) with FSAL, I-controller (safety 0.9, factor in [0.2, 10]),
// landing exactly on every output time (the reference samples the solution AT
// grid points, project/utils.py:18-21 -- no dense output, no interpolation).
// `Store` is called as store(io, z) once per output index.
template <class Sys, class Store>
__device__ __forceinline__ SbmTrajOut sbm_dopri45(const Sys& sys, double (&z)[Sys::CPL][Sys::NVX],
                                                   const double* __restrict__ t_out, int n_t,
                                                   const sbm_integrator_opts& o, Store&& store) {
  constexpr int NV = Sys::NV;
  constexpr int NVX = Sys::NVX;
  constexpr int CPL = Sys::CPL;
  using namespace dp;
  const double rtol = o.rtol, atol = o.atol;
  // max_steps < 0: a budget of |max_steps| attempts with an early exit -- a trajectory whose CURRENT step size
  // would need more than one and a half times what is LEFT of the budget for the rest of the time span gives up at
  // once (round 2: four whole budgets -- a trial point of a fit that was going to miss its budget by less than that
  // ran to the end of it, and one such trajectory sets the duration of the launch; checked every 256
  // attempts from the 512th on, when the controller has settled, and only while the step size has stopped growing
  // from one check to the next).  That is the explicit method on a stiff
  // system, its step size pinned by stability: method='auto' hands such trajectories to the implicit
  // integrator without first burning the whole budget on them.
  const bool early_exit = o.max_steps < 0;
  const int max_steps = o.max_steps > 0 ? o.max_steps : (o.max_steps < 0 ? -o.max_steps : 1000000);

  double k1[CPL][NVX], k2[CPL][NVX], k3[CPL][NVX], k4[CPL][NVX], k5[CPL][NVX], k6[CPL][NVX], zt[CPL][NVX];
  // defined values everywhere from the start: lanes / elements that carry no equation must hold
  // zeros, never whatever the previous kernel left in the register file
  SBM_ALL(c, i) { k1[c][i] = 0.0; k2[c][i] = 0.0; k3[c][i] = 0.0; k4[c][i] = 0.0; k5[c][i] = 0.0; k6[c][i] = 0.0; zt[c][i] = 0.0; }
  double t = o.t0;
  SbmTrajOut out{SBM_OK, 0, 0};
  if (n_t <= 0) return out;
  const double t_span = t_out[n_t - 1] - o.t0;

  sys.rhs(t, z, k1);

  // ---- initial step (Hairer's hinit) ----
  double h = o.h0;
  if (!(h > 0.0)) {
    double dnf = 0.0, dny = 0.0;
    SBM_ALL(c, i) {
      const double sk = atol + rtol * fabs(z[c][i]);
      const double a = k1[c][i] / sk, b = z[c][i] / sk;
      dnf = fma(a, a, dnf);
      dny = fma(b, b, dny);
    }
    dnf = sys.sum(dnf);
    dny = sys.sum(dny);
    h = (dnf <= 1e-10 || dny <= 1e-10) ? 1e-6 : sqrt(dny / dnf) * 0.01;
    h = fmin(h, t_span > 0.0 ? t_span : 1.0);
    SBM_ALL(c, i) zt[c][i] = fma(h, k1[c][i], z[c][i]);
    sys.rhs(t + h, zt, k2);
    double der2 = 0.0;
    SBM_ALL(c, i) {
      const double sk = atol + rtol * fabs(z[c][i]);
      const double a = (k2[c][i] - k1[c][i]) / sk;
      der2 = fma(a, a, der2);
    }
    der2 = sqrt(sys.sum(der2)) / h;
    const double der12 = fmax(fabs(der2), sqrt(dnf));
    const double h1 = (der12 <= 1e-15) ? fmax(1e-6, fabs(h) * 1e-3) : pow(0.01 / der12, 0.2);
    h = fmin(fmin(100.0 * h, h1), t_span > 0.0 ? t_span : 1.0);
    if (!(h > 0.0)) h = 1e-6;
  }

  int n_try = 0;
  bool failed = false;
  float h_mark = 0.f;          // step size at the previous early-exit check
  int rej_mark = 0;            // rejected attempts up to the previous check
  typename Sys::Pending pend_;
  for (int io = 0; io < n_t; ++io) {
    const double target = t_out[io];
    while (!failed && t < target) {
      if (n_try >= max_steps) { out.status = SBM_MAX_STEPS; failed = true; break; }
      if (early_exit && n_try >= 512 && (n_try & 255) == 0) {
        // A step size that is still GROWING (by half or more since the previous check) is not one pinned by stability:
        // a run that merely starts with small steps on a long horizon carries on.
        // and a controller that never rejects is limited by accuracy -- the transient of a long run -- not by stability:
        // there the step size sits on the stability boundary and about one attempt in 40 overshoots it (measured:
        // stiff50 497 rejections in 20 000 attempts, cascade20 beyond its transient 2.5 %, within it 0 of 978).
        const bool growing = (float)h > 1.5f * h_mark;
        const bool smooth = out.n_rej - rej_mark < 3;
        h_mark = (float)h;
        rej_mark = out.n_rej;
        // (the end of the time span is read again here, in the cold path, rather than kept alive across the step loop)
        if (!growing && !smooth && (t_out[n_t - 1] - t) > 1.5 * (double)(max_steps - n_try) * h) { out.status = SBM_MAX_STEPS; failed = true; break; }
      }
      ++n_try;
      // clip to land on the output time
      double hs = h;
      bool last = false;
      if (t + 1.01 * hs >= target) { hs = target - t; last = true; }

      // The tableau entries stay compile-time constants (SGPR literals the compiler can rematerialise at
      // will) and the step size multiplies each finished sum once: the 26 products h*a_ij, h*e_j would
      // otherwise sit in 52 VGPRs for the whole step -- a fifth of the register budget of a wave that
      // needs every register for its stage vectors (there is no scalar fp64 ALU to keep them in SGPRs).
      // Price: one more FMA per element in stages 3-7.
      //
      // Stages 2-4 combine the stored derivatives.  From stage 5 on every remaining linear
      // combination (the inputs of stages 6 and 7 and the error estimate) is carried as a RUNNING SUM
      // instead: when the input of stage 5 is formed, k2 / k3 / k4 are read one last time and their
      // registers take over the partial sums U6 / U7 / E (unscaled by h).  Each derivative is read
      // once instead of up to four times and the live set peaks at 7 stage vectors instead of 8
      // (z, k1, zt, U6, U7, E, k5), falling to 5 by stage 7 -- it is the v_accvgpr traffic of an
      // overflowing register file that this saves.
      const double ha21 = hs * A21;
#define SBM_S2 zt[c][i] = fma(ha21, k1[c][i], z[c][i]);
#define SBM_S3 zt[c][i] = fma(hs, fma(A32, k2[c][i], A31 * k1[c][i]), z[c][i]);
#define SBM_S4 zt[c][i] = fma(hs, fma(A43, k3[c][i], fma(A42, k2[c][i], A41 * k1[c][i])), z[c][i]);
#define SBM_S5                                                                    \
  const double a1_ = k1[c][i], a2_ = k2[c][i], a3_ = k3[c][i], a4_ = k4[c][i];    \
  zt[c][i] = fma(hs, fma(A54, a4_, fma(A53, a3_, fma(A52, a2_, A51 * a1_))), z[c][i]); \
  k2[c][i] = fma(A64, a4_, fma(A63, a3_, fma(A62, a2_, A61 * a1_))); /* U6 */     \
  k3[c][i] = fma(A74, a4_, fma(A73, a3_, A71 * a1_));                /* U7 */     \
  k4[c][i] = fma(E4, a4_, fma(E3, a3_, E1 * a1_));                   /* E  */
#define SBM_S6                                              \
  const double a5_ = k5[c][i];                              \
  zt[c][i] = fma(hs, fma(A65, a5_, k2[c][i]), z[c][i]);     \
  k3[c][i] = fma(A75, a5_, k3[c][i]);                       \
  k4[c][i] = fma(E5, a5_, k4[c][i]);
#define SBM_S7                                              \
  const double a6_ = k6[c][i];                              \
  zt[c][i] = fma(hs, fma(A76, a6_, k3[c][i]), z[c][i]);     \
  k4[c][i] = fma(E6, a6_, k4[c][i]);
      SBM_ISSUE(t + C2 * hs, SBM_S2)
      SBM_STAGE_THEN(t + C2 * hs, SBM_S2, k2, SBM_ISSUE(t + C3 * hs, SBM_S3))
      SBM_STAGE_THEN(t + C3 * hs, SBM_S3, k3, SBM_ISSUE(t + C4 * hs, SBM_S4))
      SBM_STAGE_THEN(t + C4 * hs, SBM_S4, k4, SBM_ISSUE(t + C5 * hs, SBM_S5))
      SBM_STAGE_THEN(t + C5 * hs, SBM_S5, k5, SBM_ISSUE(t + hs, SBM_S6))
      SBM_STAGE_THEN(t + hs, SBM_S6, k6, SBM_ISSUE(t + hs, SBM_S7))
      // zt is the 5th-order solution now.  k1 was last read when the input of stage 5 was formed, so
      // k7 = f(z_new) goes straight into it: an accepted step continues with it (FSAL, no copy), a
      // rejected one -- rare -- re-evaluates f(z).  One stage vector fewer alive through stages 5-7.
      SBM_STAGE_THEN(t + hs, SBM_S7, k1, )
#undef SBM_S2
#undef SBM_S3
#undef SBM_S4
#undef SBM_S5
#undef SBM_S6
#undef SBM_S7

      // embedded error estimate h (E + e7 k7); ratios and norm in f32 (they only steer the controller).
      // The norm is homogeneous: |h| multiplies it once at the end instead of every element.
      // one sum of squares per column a lane holds (a lane of the row-group system holds the
      // rows of several columns next to each other inside one array: Sys::col_of maps them)
      float colsum[Sys::NCS];
#pragma unroll
      for (int c = 0; c < Sys::NCS; ++c) colsum[c] = 0.f;
      float xsum = 0.f;
      auto err_ratio = [&](int c, int i) {
        const double e = fma(E7, k1[c][i], k4[c][i]);
        const double sc = fma(rtol, fmax(fabs(z[c][i]), fabs(zt[c][i])), atol);
        return (float)e * __builtin_amdgcn_rcpf((float)sc);
      };
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float r = err_ratio(c, i);
          colsum[Sys::col_of(c, i)] = fmaf(r, r, colsum[Sys::col_of(c, i)]);
        }
#pragma unroll
        for (int i = NV; i < NVX; ++i) {
          const float r = err_ratio(c, i);
          xsum = fmaf(r, r, xsum);
        }
      }
      const float err = fabsf((float)hs) * sys.norm(colsum, xsum);

      const bool finite = (err == err) && (err < 3.0e38f);
      const bool accept = finite && (err <= 1.0f);
      float fac;
      if (!finite) {
        fac = 0.2f;
      } else {
        // 0.9 * err^(-1/5), clipped
        fac = 0.9f * __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(fmaxf(err, 1e-30f)));
        fac = fminf(10.f, fmaxf(0.2f, fac));
      }
      if (accept) {
        ++out.n_acc;
        t = last ? target : t + hs;
        SBM_ALL(c, i) z[c][i] = zt[c][i];
        const double hn = hs * (double)fac;
        // a step clipped to hit an output time says nothing against the unclipped proposal
        h = last ? fmax(hn, h) : hn;
      } else {
        ++out.n_rej;
        h = hs * (double)fminf(fac, 1.0f);
        if (!(h > 1e-14 * fmax(fabs(t), 1e-3))) {
          out.status = finite ? SBM_STEP_UNDERFLOW : SBM_NON_FINITE;
          failed = true;
        } else {
          sys.rhs(t, z, k1);   // k1 holds f(z_new) of the rejected attempt
        }
      }
    }
    if (failed) {
      SBM_ALL(c, i) z[c][i] = __builtin_nan("");
    }
    store(io, z);
  }
  return out;
}

// Dormand-Prince 8(5,3) (DOP853), for tight tolerances: twelve stages per step instead of six, one seventh of the
// steps at rtol 1e-9 (cascade20 with sensitivities: 123 steps where DOPRI45 takes 843 -- 3.4 times fewer evaluations
// of the right-hand side).  Same driver contract as sbm_dopri45: lands on every output time, error control on state and
// sensitivities with the system's norm, step budget with early exit.  Ten stage vectors are alive at the peak (k2's
// storage is taken over by k4, k3's by k6: the tableau never reads them later), with the state and the stage argument
// that is twelve vectors of a lane's elements -- the kernels give this method the whole register file (one wave per
// SIMD).  Error estimate and controller as in Hairer's code: err = |h| e5^2 / sqrt(e5^2 + 0.01 e3^2) with e5, e3 the
// norms of the fifth- and third-order embedded estimates, new step = h * min(6, max(1/3, 0.9 err^(-1/8))).
// f(y_new) is evaluated after acceptance only (it is the next step's k1).
template <class Sys, class Store>
__device__ __forceinline__ SbmTrajOut sbm_dop853(const Sys& sys, double (&z)[Sys::CPL][Sys::NVX],
                                                  const double* __restrict__ t_out, int n_t,
                                                  const sbm_integrator_opts& o, Store&& store) {
  constexpr int NV = Sys::NV;
  constexpr int NVX = Sys::NVX;
  constexpr int CPL = Sys::CPL;
  using namespace dp8;
  const double rtol = o.rtol, atol = o.atol;
  const bool early_exit = o.max_steps < 0;
  const int max_steps = o.max_steps > 0 ? o.max_steps : (o.max_steps < 0 ? -o.max_steps : 1000000);

  double k1[CPL][NVX], kx[CPL][NVX], ky[CPL][NVX], k5[CPL][NVX], k7[CPL][NVX], k8[CPL][NVX], k9[CPL][NVX],
      k10[CPL][NVX], k11[CPL][NVX], k12[CPL][NVX], zt[CPL][NVX];
  SBM_ALL(c, i) {
    k1[c][i] = 0.0; kx[c][i] = 0.0; ky[c][i] = 0.0; k5[c][i] = 0.0; k7[c][i] = 0.0; k8[c][i] = 0.0; k9[c][i] = 0.0;
    k10[c][i] = 0.0; k11[c][i] = 0.0; k12[c][i] = 0.0; zt[c][i] = 0.0;
  }
  double t = o.t0;
  SbmTrajOut out{SBM_OK, 0, 0};
  if (n_t <= 0) return out;
  const double t_span = t_out[n_t - 1] - o.t0;

  sys.rhs(t, z, k1);

  // ---- initial step (Hairer's hinit, order 8) ----
  double h = o.h0;
  if (!(h > 0.0)) {
    double dnf = 0.0, dny = 0.0;
    SBM_ALL(c, i) {
      const double sk = atol + rtol * fabs(z[c][i]);
      const double a = k1[c][i] / sk, b = z[c][i] / sk;
      dnf = fma(a, a, dnf);
      dny = fma(b, b, dny);
    }
    dnf = sys.sum(dnf);
    dny = sys.sum(dny);
    h = (dnf <= 1e-10 || dny <= 1e-10) ? 1e-6 : sqrt(dny / dnf) * 0.01;
    h = fmin(h, t_span > 0.0 ? t_span : 1.0);
    SBM_ALL(c, i) zt[c][i] = fma(h, k1[c][i], z[c][i]);
    sys.rhs(t + h, zt, kx);
    double der2 = 0.0;
    SBM_ALL(c, i) {
      const double sk = atol + rtol * fabs(z[c][i]);
      const double a = (kx[c][i] - k1[c][i]) / sk;
      der2 = fma(a, a, der2);
    }
    der2 = sqrt(sys.sum(der2)) / h;
    const double der12 = fmax(fabs(der2), sqrt(dnf));
    const double h1 = (der12 <= 1e-15) ? fmax(1e-6, fabs(h) * 1e-3) : pow(0.01 / der12, 0.125);
    h = fmin(fmin(100.0 * h, h1), t_span > 0.0 ? t_span : 1.0);
    if (!(h > 0.0)) h = 1e-6;
  }

  int n_try = 0;
  bool failed = false;
  float h_mark = 0.f;
  int rej_mark = 0;
  typename Sys::Pending pend_;
  for (int io = 0; io < n_t; ++io) {
    const double target = t_out[io];
    while (!failed && t < target) {
      if (n_try >= max_steps) { out.status = SBM_MAX_STEPS; failed = true; break; }
      if (early_exit && n_try >= 512 && (n_try & 255) == 0) {      // (see sbm_dopri45)
        const bool growing = (float)h > 1.5f * h_mark;
        const bool smooth = out.n_rej - rej_mark < 3;
        h_mark = (float)h;
        rej_mark = out.n_rej;
        if (!growing && !smooth && (t_out[n_t - 1] - t) > 1.5 * (double)(max_steps - n_try) * h) { out.status = SBM_MAX_STEPS; failed = true; break; }
      }
      ++n_try;
      double hs = h;
      bool last = false;
      if (t + 1.01 * hs >= target) { hs = target - t; last = true; }

      // The step size multiplies each finished sum once (tableau entries stay literals, as in sbm_dopri45).  Stages are
      // software-pipelined as there: stage s+1's state path (extra elements, LDS hand-off, operand fetch) is issued
      // between the row evaluation and the column work of stage s.
      const double ha21 = hs * A2_1;
#define SBM_D2 zt[c][i] = fma(ha21, k1[c][i], z[c][i]);
#define SBM_D3 zt[c][i] = fma(hs, fma(A3_2, kx[c][i], A3_1 * k1[c][i]), z[c][i]);
#define SBM_D4 zt[c][i] = fma(hs, fma(A4_3, ky[c][i], A4_1 * k1[c][i]), z[c][i]);
#define SBM_D5 zt[c][i] = fma(hs, fma(A5_4, kx[c][i], fma(A5_3, ky[c][i], A5_1 * k1[c][i])), z[c][i]);
#define SBM_D6 zt[c][i] = fma(hs, fma(A6_5, k5[c][i], fma(A6_4, kx[c][i], A6_1 * k1[c][i])), z[c][i]);
#define SBM_D7 zt[c][i] = fma(hs, fma(A7_6, ky[c][i], fma(A7_5, k5[c][i], fma(A7_4, kx[c][i], A7_1 * k1[c][i]))), z[c][i]);
#define SBM_D8                                                                                                      \
  zt[c][i] = fma(hs, fma(A8_7, k7[c][i], fma(A8_6, ky[c][i], fma(A8_5, k5[c][i], fma(A8_4, kx[c][i], A8_1 * k1[c][i])))), \
                 z[c][i]);
#define SBM_D9                                                                                                      \
  zt[c][i] = fma(hs, fma(A9_8, k8[c][i], fma(A9_7, k7[c][i], fma(A9_6, ky[c][i], fma(A9_5, k5[c][i],                 \
                 fma(A9_4, kx[c][i], A9_1 * k1[c][i]))))), z[c][i]);
#define SBM_D10                                                                                                     \
  zt[c][i] = fma(hs, fma(A10_9, k9[c][i], fma(A10_8, k8[c][i], fma(A10_7, k7[c][i], fma(A10_6, ky[c][i],             \
                 fma(A10_5, k5[c][i], fma(A10_4, kx[c][i], A10_1 * k1[c][i])))))), z[c][i]);
#define SBM_D11                                                                                                     \
  zt[c][i] = fma(hs, fma(A11_10, k10[c][i], fma(A11_9, k9[c][i], fma(A11_8, k8[c][i], fma(A11_7, k7[c][i],           \
                 fma(A11_6, ky[c][i], fma(A11_5, k5[c][i], fma(A11_4, kx[c][i], A11_1 * k1[c][i]))))))), z[c][i]);
#define SBM_D12                                                                                                     \
  zt[c][i] = fma(hs, fma(A12_11, k11[c][i], fma(A12_10, k10[c][i], fma(A12_9, k9[c][i], fma(A12_8, k8[c][i],         \
                 fma(A12_7, k7[c][i], fma(A12_6, ky[c][i], fma(A12_5, k5[c][i], fma(A12_4, kx[c][i],                 \
                 A12_1 * k1[c][i])))))))), z[c][i]);
      SBM_ISSUE(t + C2 * hs, SBM_D2)
      SBM_STAGE_THEN(t + C2 * hs, SBM_D2, kx, SBM_ISSUE(t + C3 * hs, SBM_D3))       // k2
      SBM_STAGE_THEN(t + C3 * hs, SBM_D3, ky, SBM_ISSUE(t + C4 * hs, SBM_D4))       // k3
      SBM_STAGE_THEN(t + C4 * hs, SBM_D4, kx, SBM_ISSUE(t + C5 * hs, SBM_D5))       // k4 takes k2's storage
      SBM_STAGE_THEN(t + C5 * hs, SBM_D5, k5, SBM_ISSUE(t + C6 * hs, SBM_D6))
      SBM_STAGE_THEN(t + C6 * hs, SBM_D6, ky, SBM_ISSUE(t + C7 * hs, SBM_D7))       // k6 takes k3's storage
      SBM_STAGE_THEN(t + C7 * hs, SBM_D7, k7, SBM_ISSUE(t + C8 * hs, SBM_D8))
      SBM_STAGE_THEN(t + C8 * hs, SBM_D8, k8, SBM_ISSUE(t + C9 * hs, SBM_D9))
      SBM_STAGE_THEN(t + C9 * hs, SBM_D9, k9, SBM_ISSUE(t + C10 * hs, SBM_D10))
      SBM_STAGE_THEN(t + C10 * hs, SBM_D10, k10, SBM_ISSUE(t + C11 * hs, SBM_D11))
      SBM_STAGE_THEN(t + C11 * hs, SBM_D11, k11, SBM_ISSUE(t + hs, SBM_D12))
      SBM_STAGE_THEN(t + hs, SBM_D12, k12, )
#undef SBM_D2
#undef SBM_D3
#undef SBM_D4
#undef SBM_D5
#undef SBM_D6
#undef SBM_D7
#undef SBM_D8
#undef SBM_D9
#undef SBM_D10
#undef SBM_D11
#undef SBM_D12

      // 8th-order solution into zt; the two error estimates, ratios and norms in f32 (they only steer the controller)
      float cs5[Sys::NCS], cs3[Sys::NCS];
#pragma unroll
      for (int c = 0; c < Sys::NCS; ++c) { cs5[c] = 0.f; cs3[c] = 0.f; }
      float xs5 = 0.f, xs3 = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
          const double a1 = k1[c][i], a6 = ky[c][i], a7 = k7[c][i], a8 = k8[c][i], a9 = k9[c][i], a10 = k10[c][i],
                       a11 = k11[c][i], a12 = k12[c][i];
          const double zn = fma(hs, fma(B12, a12, fma(B11, a11, fma(B10, a10, fma(B9, a9, fma(B8, a8, fma(B7, a7,
                                fma(B6, a6, B1 * a1))))))), z[c][i]);
          const double e5 = fma(E5_12, a12, fma(E5_11, a11, fma(E5_10, a10, fma(E5_9, a9, fma(E5_8, a8, fma(E5_7, a7,
                                fma(E5_6, a6, E5_1 * a1)))))));
          const double e3 = fma(E3_12, a12, fma(E3_11, a11, fma(E3_10, a10, fma(E3_9, a9, fma(E3_8, a8, fma(E3_7, a7,
                                fma(E3_6, a6, E3_1 * a1)))))));
          const float rsc = __builtin_amdgcn_rcpf((float)fma(rtol, fmax(fabs(z[c][i]), fabs(zn)), atol));
          const float r5 = (float)e5 * rsc, r3 = (float)e3 * rsc;
          zt[c][i] = zn;
          if (i < NV) {
            cs5[Sys::col_of(c, i)] = fmaf(r5, r5, cs5[Sys::col_of(c, i)]);
            cs3[Sys::col_of(c, i)] = fmaf(r3, r3, cs3[Sys::col_of(c, i)]);
          } else {
            xs5 = fmaf(r5, r5, xs5);
            xs3 = fmaf(r3, r3, xs3);
          }
        }
      }
      const float n5 = sys.norm(cs5, xs5), n3 = sys.norm(cs3, xs3);
      const float den = n5 * n5 + 0.01f * n3 * n3;
      const float err = fabsf((float)hs) * (den > 0.f ? n5 * n5 * __builtin_amdgcn_rsqf(den) : 0.f);

      const bool finite = (err == err) && (err < 3.0e38f) && (n3 == n3) && (n3 < 3.0e38f);
      const bool accept = finite && (err <= 1.0f);
      float fac;
      if (!finite) {
        fac = 1.0f / 3.0f;
      } else {
        fac = 0.9f * __builtin_amdgcn_exp2f(-0.125f * __builtin_amdgcn_logf(fmaxf(err, 1e-30f)));   // 0.9 err^(-1/8)
        fac = fminf(6.f, fmaxf(1.0f / 3.0f, fac));
      }
      if (accept) {
        ++out.n_acc;
        t = last ? target : t + hs;
        SBM_ALL(c, i) z[c][i] = zt[c][i];
        sys.rhs(t, z, k1);                       // FSAL: f(y_new) is the next step's k1
        const double hn = hs * (double)fac;
        h = last ? fmax(hn, h) : hn;
      } else {
        ++out.n_rej;
        h = hs * (double)fminf(fac, 1.0f);
        if (!(h > 1e-14 * fmax(fabs(t), 1e-3))) {
          out.status = finite ? SBM_STEP_UNDERFLOW : SBM_NON_FINITE;
          failed = true;
        }
      }
    }
    if (failed) {
      SBM_ALL(c, i) z[c][i] = __builtin_nan("");
    }
    store(io, z);
  }
  return out;
}

// one entry for the kernels: the driver of METHOD
template <int METHOD, class Sys, class Store>
__device__ __forceinline__ SbmTrajOut sbm_integrate(const Sys& sys, double (&z)[Sys::CPL][Sys::NVX],
                                                     const double* __restrict__ t_out, int n_t,
                                                     const sbm_integrator_opts& o, Store&& store) {
  if constexpr (METHOD == SBM_DOPRI45) return sbm_dopri45(sys, z, t_out, n_t, o, store);
  else if constexpr (METHOD == SBM_DOP853) return sbm_dop853(sys, z, t_out, n_t, o, store);
  else return sbm_rk4(sys, z, t_out, n_t, o, store);
}

// classic RK4, fixed step: every output interval is cut into ceil(dt / h0) equal steps
template <class Sys, class Store>
__device__ __forceinline__ SbmTrajOut sbm_rk4(const Sys& sys, double (&z)[Sys::CPL][Sys::NVX],
                                               const double* __restrict__ t_out, int n_t,
                                               const sbm_integrator_opts& o, Store&& store) {
  constexpr int NV = Sys::NV;
  constexpr int NVX = Sys::NVX;
  constexpr int CPL = Sys::CPL;
  double k[CPL][NVX], acc[CPL][NVX], zt[CPL][NVX];
  SBM_ALL(c, i) { k[c][i] = 0.0; acc[c][i] = 0.0; zt[c][i] = 0.0; }
  typename Sys::Pending pend_;
  SbmTrajOut out{SBM_OK, 0, 0};
  const double h0 = o.h0;
  const int max_steps = o.max_steps > 0 ? o.max_steps : 1000000000;
  double t = o.t0;
  bool failed = !(h0 > 0.0);
  if (failed) out.status = SBM_STEP_UNDERFLOW;
  for (int io = 0; io < n_t; ++io) {
    const double target = t_out[io];
    const double dt = target - t;
    if (!failed && dt > 0.0) {
      const double nd = ceil(dt / h0 - 1e-9) * (o.step_mult > 0 ? o.step_mult : 1);
      int ns = nd < 1.0 ? 1 : (nd > 2.0e9 ? 2000000000 : (int)nd);
      if (out.n_acc + ns > max_steps) { out.status = SBM_MAX_STEPS; failed = true; }
      if (!failed) {
        const double hs = dt / ns, hh = 0.5 * hs, h6 = hs * (1.0 / 6.0);
        const double t0 = t;
        for (int s = 0; s < ns; ++s) {
          const double ts = fma((double)s, hs, t0);
          sys.rhs(ts, z, k);
          SBM_ALL(c, i) acc[c][i] = k[c][i];
          SBM_STAGE(ts + hh, zt[c][i] = fma(hh, k[c][i], z[c][i]);, k)
          SBM_ALL(c, i) acc[c][i] = fma(2.0, k[c][i], acc[c][i]);
          SBM_STAGE(ts + hh, zt[c][i] = fma(hh, k[c][i], z[c][i]);, k)
          SBM_ALL(c, i) acc[c][i] = fma(2.0, k[c][i], acc[c][i]);
          SBM_STAGE(ts + hs, zt[c][i] = fma(hs, k[c][i], z[c][i]);, k)
          SBM_ALL(c, i) z[c][i] = fma(h6, acc[c][i] + k[c][i], z[c][i]);
        }
        out.n_acc += ns;
        t = target;
      }
    }
    if (failed) {
      SBM_ALL(c, i) z[c][i] = __builtin_nan("");
    }
    store(io, z);
  }
  // RK4 has no error estimate: flag non-finite end states
  bool bad = false;
  SBM_ALL(c, i) bad = bad || !(z[c][i] == z[c][i]);
  if (!failed && bad) out.status = SBM_NON_FINITE;
  return out;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// Sensitivity kernel.  grid = n_traj blocks of one wave; at ~1 wave per SIMD
// (the stage vectors fill most of the 512-VGPR budget) a CU holds 4 trajectories,
// the chip 1024, so a 4096-vector ensemble is 4 waves of work per SIMD.
template <class M, int METHOD>
__global__ void __launch_bounds__(64) sbm_sens_kernel(sbm_kernel_args a) {
  using Sys = SensSystem<M>;
  constexpr int NV = Sys::NV;
  constexpr int CPL = Sys::CPL;
  constexpr int NK = M::NK;
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;  // wave-uniform
  const int lane = threadIdx.x;

  Sys sys{a.P + (size_t)traj * M::NP, lane};
  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;

  double z[CPL][NV];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = lane + 64 * c;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double v = 0.0;
      if (col == 0) {
        if (a.y0) v = a.y0[i];
      } else if (col <= NK) {
        if (a.s0) v = a.s0[i * NK + (col - 1)];
      }
      z[c][i] = v;
    }
  }

  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
  auto store = [&](int io, const double (&zz)[CPL][NV]) {
    if (Yt && lane == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) Yt[(size_t)io * NV + i] = zz[0][i];
    }
    if (St) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int scol = lane + 64 * c - 1;
        if (scol >= 0 && scol < NK) {
#pragma unroll
          for (int i = 0; i < NV; ++i) St[((size_t)io * NV + i) * NK + scol] = zz[c][i];
        }
      }
    }
  };

  SbmTrajOut r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);

  if (lane == 0) {
    if (a.status) a.status[traj] = r.status;
    if (a.n_steps) a.n_steps[traj] = r.n_acc;
    if (a.n_reject) a.n_reject[traj] = r.n_rej;
  }
}

// State-only kernel.  One trajectory per lane, 64 per block.
template <class M, int METHOD>
__global__ void __launch_bounds__(64) sbm_state_kernel(sbm_kernel_args a) {
  using Sys = StateSystem<M>;
  constexpr int NV = Sys::NV;
  constexpr int NP = M::NP;
  __shared__ double p_lds[NP * 64];
  const int lane = threadIdx.x;
  const int traj0 = blockIdx.x * 64;
  const int traj = traj0 + lane;
  const int n_here = min(64, a.n_traj - traj0);

  // coalesced read of the block's 64 x NP parameter slab, transposed into LDS
  const double* Pb = a.P + (size_t)traj0 * NP;
  for (int e = lane; e < n_here * NP; e += 64) {
    const int tl = e / NP, c = e - tl * NP;
    p_lds[c * 64 + tl] = Pb[e];
  }
  __syncthreads();
  if (traj >= a.n_traj) return;

  Sys sys{SbmLdsParams{p_lds + lane}};
  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;

  double z[1][NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) z[0][i] = a.y0 ? a.y0[i] : 0.0;

  double* Yt = a.Y + (size_t)traj * a.n_t * NV;
  auto store = [&](int io, const double (&zz)[1][NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) Yt[(size_t)io * NV + i] = zz[0][i];
  };

  SbmTrajOut r;
  if (METHOD == SBM_DOPRI45) r = sbm_dopri45(sys, z, tg, glen, a.opts, store);
  else r = sbm_rk4(sys, z, tg, glen, a.opts, store);

  if (a.status) a.status[traj] = r.status;
  if (a.n_steps) a.n_steps[traj] = r.n_acc;
  if (a.n_reject) a.n_reject[traj] = r.n_rej;
}

// ===========================================================================
// Row-lane sensitivity kernel: SIMD across isomorphic equations.
//
// Still one trajectory per wavefront and one sensitivity column per lane (lane j owns column j
// of S, all NV rows, every Runge-Kutta stage in VGPRs).  What changes is who evaluates the part
// of the right-hand side that is identical for all columns.  The per-wave kernel above computes
// f, J_y and J_p on all 64 lanes from broadcast operands, then every lane picks "its" J_p entry
// with v_cndmask chains: rocprofv3 shows it VALU-issue bound (one wave per SIMD issues one
// instruction per ~4 cycles whatever its type) with ~60 % of the instructions spent there.
// Here
//   * the state itself lives one component per lane (lane i integrates y_i as an extra
//     element next to its column), so publishing a stage state is ONE ds_write per lane;
//   * rows with the same kinetic form (emit_rowlane.py) are evaluated together, lane i
//     computing f_i and the J_y / J_p entries of row i from per-lane operands (its parameters
//     sit in registers for the whole kernel, its state operands come from LDS by index);
//   * each row lane drops its J_y entries into a list and its J_p entries into the additive
//     matrix A[NV][64] in LDS, and every column lane then evaluates the same
//     dz_i = sum_m J_y[i,m] z_m + A[i][lane]  -- J_y broadcast reads, A one column per lane
//     (conflict-free), no selects, no readfirstlane.
// Only wave-local ordering is needed (a 64-thread workgroup): no cross-wave barrier anywhere.
// ===========================================================================
template <class M>
struct SbmRowLaneShared {
  double Y[64];                 // stage state, one component per row lane
  double A[M::NV * 64 + 2];     // A[i][c] = J_p[i][c]; zero where J_p is structurally zero
};

template <class M>
struct RowLaneSystem {
  static constexpr int NV = M::NV;
  static constexpr int NVX = M::NV + 1;  // column rows + this lane's own state component
  static constexpr int CPL = 1;
  static constexpr int NCS = 1;
  __device__ __forceinline__ static constexpr int col_of(int, int) { return 0; }
  SbmRowLaneShared<M>* sh;
  int lane;
  int cls;                       // class of this lane's row, -1 on lanes without a row
  int yidx[M::RL_MAXYS];         // which state feeds operand slot s
  double ps[M::RL_MAXPS];        // this row's parameters
  int apos[M::RL_MAXJP];         // where this row's J_p entries go in A
  double sj[M::RL_NSTATIC > 0 ? M::RL_NSTATIC : 1];  // parameter-only J_y entries, wave-uniform

  // LDS traffic of ONE wave is processed in issue order, so a ds_read issued after a ds_write of
  // another lane sees that write: no barrier and no waitcnt is needed between the phases below,
  // only a compiler-level fence that keeps the memory operations in program order.
  __device__ __forceinline__ static void lds_order() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

  struct Token {
    double f;                  // derivative of this lane's state component
    double jy[M::RL_MAXJY];    // J_y entries of this lane's row (read by the other lanes via v_readlane)
  };
  struct Pending { double ys[M::RL_MAXYS]; };   // the row's state operands, in flight from LDS
  // issue: publish this lane's stage state and start fetching the operands of the lane's row
  __device__ __forceinline__ Pending issue(double /*t*/, const double (&z)[1][NVX]) const {
    Pending p;
    sh->Y[lane] = z[0][NV];
    lds_order();
#pragma unroll
    for (int s = 0; s < M::RL_MAXYS; ++s) p.ys[s] = sh->Y[yidx[s]];
    lds_order();
    return p;
  }
  // eval: evaluate the lane's row and publish its J_p entries
  __device__ __forceinline__ Token eval(const Pending& p, double t) const {
    Token k;
    double jp[M::RL_MAXJP];
    k.f = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJY; ++s) k.jy[s] = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = 0.0;
    M::class_dispatch(cls, t, p.ys, ps, k.f, k.jy, jp);
    k.f = cls >= 0 ? k.f : 0.0;   // lanes without a row come out of the select chain with the last class's value
    lds_order();
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) sh->A[apos[s]] = jp[s];
    lds_order();
    return k;
  }
  // phase 2: the lane's column, dz = J_y z + A[:, lane]
  __device__ __forceinline__ void finish(const Token& k, double /*t*/, const double (&z)[1][NVX],
                                         double (&dz)[1][NVX]) const {
    double zc[NV], dc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) zc[i] = z[0][i];
    // A[:, lane] is fetched here rather than in begin(): holding 2*NV more registers across the
    // column rows costs more (AGPR traffic) than the exposed LDS latency (measured 12.8 vs 13.3 ms)
    double acol[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acol[i] = sh->A[i * 64 + lane];
    lds_order();
    M::apply_rowlane(k.jy, sj, acol, zc, dc);
#pragma unroll
    for (int i = 0; i < NV; ++i) dz[0][i] = dc[i];
  }
  __device__ __forceinline__ void extra_out(const Token& k, double (&dz)[1][NVX]) const { dz[0][NV] = k.f; }
  __device__ __forceinline__ void rhs(double t, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    const Token k = eval(issue(t, z), t);
    extra_out(k, dz);
    finish(k, t, z, dz);
  }
  // max( RMS of the state error, max over columns of the column RMS ): the same test as the
  // per-wave kernel, with the state spread over the lanes
  __device__ __forceinline__ float norm(const float (&colsum)[1], float xsum) const {
    // lanes without a column / without a row carry no equation: they must not steer the controller
    const float m = (lane < M::NK) ? sbm_nan_to_inf(colsum[0]) : 0.f;
    const float x = (lane < NV) ? sbm_nan_to_inf(xsum) : 0.f;
    const float mx = sbm_wave_max(m);
    const float xs = sbm_wave_sumf(x);
    return sqrtf(fmaxf(mx, xs) * (1.0f / NV));   // +inf: the driver rejects the step
  }
  __device__ __forceinline__ double sum(double v) const { return sbm_wave_sum(v); }
};

template <class M, int METHOD>
__global__ void __launch_bounds__(64) sbm_sens_rowlane_kernel(sbm_kernel_args a) {
  using Sys = RowLaneSystem<M>;
  constexpr int NV = M::NV;
  constexpr int NVX = NV + 1;
  constexpr int NK = M::NK;
  static_assert(NV <= 64 && NK <= 64, "row-lane kernel: one row and one column per lane");
  __shared__ SbmRowLaneShared<M> sh;
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x;

  for (int i = lane; i < NV * 64 + 2; i += 64) sh.A[i] = 0.0;
  sh.Y[lane] = 0.0;

  Sys sys;
  sys.sh = &sh;
  sys.lane = lane;
  const bool has_row = lane < NV;
  const int row = has_row ? lane : 0;
  sys.cls = has_row ? M::rl_class(row) : -1;
  const double* P = a.P + (size_t)traj * M::NP;
#pragma unroll
  for (int s = 0; s < M::RL_MAXYS; ++s) sys.yidx[s] = M::rl_ys(s, row);
#pragma unroll
  for (int s = 0; s < M::RL_MAXPS; ++s) sys.ps[s] = P[M::rl_ps(s, row)];
#pragma unroll
  for (int s = 0; s < M::RL_MAXJP; ++s) sys.apos[s] = has_row ? M::rl_apos(s, row) : NV * 64 + 1;
  __syncthreads();
  {  // parameter-only J_y entries: evaluate the rows once (any state will do) and broadcast them
    double ys0[M::RL_MAXYS], f0 = 0.0, jy0[M::RL_MAXJY], jp0[M::RL_MAXJP];
#pragma unroll
    for (int s = 0; s < M::RL_MAXYS; ++s) ys0[s] = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJY; ++s) jy0[s] = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) jp0[s] = 0.0;
    M::class_dispatch(sys.cls, a.opts.t0, ys0, sys.ps, f0, jy0, jp0);
    M::rl_static(jy0, sys.sj);
  }

  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;

  double z[1][NVX];
#pragma unroll
  for (int i = 0; i < NV; ++i) z[0][i] = (a.s0 && lane < NK) ? a.s0[i * NK + lane] : 0.0;
  z[0][NV] = (a.y0 && has_row) ? a.y0[lane] : 0.0;

  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
  auto store = [&](int io, const double (&zz)[1][NVX]) {
    if (Yt && has_row) Yt[(size_t)io * NV + lane] = zz[0][NV];
    if (St && lane < NK) {
#pragma unroll
      for (int i = 0; i < NV; ++i) St[((size_t)io * NV + i) * NK + lane] = zz[0][i];
    }
  };

  SbmTrajOut r;
  r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);


  if (lane == 0) {
    if (a.status) a.status[traj] = r.status;
    if (a.n_steps) a.n_steps[traj] = r.n_acc;
    if (a.n_reject) a.n_reject[traj] = r.n_rej;
  }
}

// ===========================================================================
// State-only kernel, one trajectory per wavefront: lane i integrates y_i and evaluates f_i with the
// class bodies of the row-lane form.  The lane-per-trajectory kernel above needs 64 trajectories to
// fill ONE wave: an ensemble of 4096 is 64 waves on a chip with 1024 SIMDs, and the launch takes as
// long as one trajectory's serial chain of ~1400-instruction steps.  Here the same ensemble is 4096
// small waves (a handful of registers each, 8 per SIMD) of ~400-instruction steps.  Past ~64k
// trajectories the chip is full either way and the lane-per-trajectory kernel, which does no
// redundant work, wins again (launcher).
// ===========================================================================
template <class M>
struct StateRowSystem {
  static constexpr int NV = 0;           // no column rows: the lane's state components are the "extra" elements
  static constexpr int RPL = (M::NV + 63) / 64;   // state rows per lane: lane, lane + 64, ...
  static constexpr int NVX = RPL;
  static constexpr int CPL = 1;
  static constexpr int NCS = 1;
  static constexpr bool kUniform = true;
  __device__ __forceinline__ static constexpr int col_of(int, int) { return 0; }
  double* Y;                     // [64 * RPL] in LDS: stage state, one component per (lane, r)
  int lane;
  int cls[RPL];
  int yidx[RPL][M::RL_MAXYS];
  double ps[RPL][M::RL_MAXPS];

  struct Pending { double ys[RPL][M::RL_MAXYS]; };
  struct Token { double f[RPL]; };
  __device__ __forceinline__ Pending issue(double, const double (&z)[1][NVX]) const {
    Pending p;
#pragma unroll
    for (int r = 0; r < RPL; ++r) Y[lane + 64 * r] = z[0][r];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
    for (int r = 0; r < RPL; ++r)
#pragma unroll
      for (int s = 0; s < M::RL_MAXYS; ++s) p.ys[r][s] = Y[yidx[r][s]];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return p;
  }
  __device__ __forceinline__ Token eval(const Pending& p, double t) const {
    Token k;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      double jy[M::RL_MAXJY], jp[M::RL_MAXJP];   // dead: the compiler drops the Jacobian arithmetic
      k.f[r] = 0.0;
#pragma unroll
      for (int s = 0; s < M::RL_MAXJY; ++s) jy[s] = 0.0;
#pragma unroll
      for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = 0.0;
      M::class_dispatch(cls[r], t, p.ys[r], ps[r], k.f[r], jy, jp);
      k.f[r] = cls[r] >= 0 ? k.f[r] : 0.0;
    }
    return k;
  }
  __device__ __forceinline__ void extra_out(const Token& k, double (&dz)[1][NVX]) const {
#pragma unroll
    for (int r = 0; r < RPL; ++r) dz[0][r] = k.f[r];
  }
  __device__ __forceinline__ void finish(const Token&, double, const double (&)[1][NVX], double (&)[1][NVX]) const {}
  __device__ __forceinline__ void rhs(double t, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    extra_out(eval(issue(t, z), t), dz);
  }
  __device__ __forceinline__ float norm(const float (&)[1], float xsum) const {
    const float x = (lane < M::NV) ? sbm_nan_to_inf(xsum) : 0.f;   // rows beyond NV hold zeros throughout
    return sqrtf(sbm_wave_sumf(x) * (1.0f / M::NV));
  }
  __device__ __forceinline__ double sum(double v) const { return sbm_wave_sum(v); }
};

template <class M, int METHOD>
__global__ void __launch_bounds__(64) sbm_state_rows_kernel(sbm_kernel_args a) {
  using Sys = StateRowSystem<M>;
  constexpr int RPL = Sys::RPL;
  __shared__ double Ysh[64 * RPL];
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = blockIdx.x;
  const int lane = threadIdx.x;
  Sys sys;
  sys.Y = Ysh;
  sys.lane = lane;
  const double* P = a.P + (size_t)traj * M::NP;
  double z[1][RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    Ysh[lane + 64 * r] = 0.0;
    const bool has_row = lane + 64 * r < M::NV;
    const int row = has_row ? lane + 64 * r : 0;
    sys.cls[r] = has_row ? M::rl_class(row) : -1;
#pragma unroll
    for (int s = 0; s < M::RL_MAXYS; ++s) sys.yidx[r][s] = M::rl_ys(s, row);
#pragma unroll
    for (int s = 0; s < M::RL_MAXPS; ++s) sys.ps[r][s] = P[M::rl_ps(s, row)];
    z[0][r] = (a.y0 && has_row) ? a.y0[row] : 0.0;
  }
  __syncthreads();
  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;
  double* Yt = a.Y + (size_t)traj * a.n_t * M::NV;
  auto store = [&](int io, const double (&zz)[1][RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r)
      if (lane + 64 * r < M::NV) Yt[(size_t)io * M::NV + lane + 64 * r] = zz[0][r];
  };
  SbmTrajOut r;
  r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);

  if (lane == 0) {
    if (a.status) a.status[traj] = r.status;
    if (a.n_steps) a.n_steps[traj] = r.n_acc;
    if (a.n_reject) a.n_reject[traj] = r.n_rej;
  }
}

// ===========================================================================
// Packed state-rows kernel: several trajectories per wavefront.
//
// A model with few state variables leaves most lanes of the state-rows kernel idle (cascade20: 20 of 64), and
// with thousands of trajectories the kernel is bound by instruction issue, not latency: every wavefront costs
// the same issue slots whether 20 or 60 of its lanes carry equations.  Here a wavefront is cut into 64 / SEG
// segments of SEG = 16 or 32 lanes, one trajectory each (lane i of a segment = state component i).  The
// segments share the instruction stream but not the control flow: each trajectory has its own time, step size
// and accept / reject decisions (per-lane values, uniform within a segment), so the wavefront runs until its
// slowest segment is done and a rejected step of one segment idles the others for one evaluation -- a few per
// cent.  Reductions stay inside a segment: DPP within rows of 16, one ds_bpermute across the two rows of a
// 32-lane segment (the fixed-step method reduces nothing and packs segments of any multiple of four lanes).  LDS exchange of the stage state is per segment (operand indices offset by the segment base).
// Used from 2048 trajectories on (below that the chip is not full and the unpacked kernel has the lower latency).
// ===========================================================================
// Sum over the lanes of a segment, the SAME BITS in every lane: the result steers the step size, and lanes of one
// trajectory that disagree in the last bit drift apart in time (measured: 3e-4 relative error with a rotation-based
// reduction whose lanes add the same numbers in different orders).  Every level is an exchange between two lanes that
// both form own + other -- commutative, hence identical: quad_perm (xor 1, xor 2), row_half_mirror (i <-> 7 - i),
// row_mirror (i <-> 15 - i) on DPP, the two rows of a 32-lane segment through ds_bpermute.
template <int SEG>
__device__ __forceinline__ float sbm_seg_sumf(float v) {
  v += sbm_dpp<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v += sbm_dpp<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
  v += sbm_dpp<0x141, 0xf>(v);   // row_half_mirror
  v += sbm_dpp<0x140, 0xf>(v);   // row_mirror: every lane of the row of 16 holds the row's sum
  if constexpr (SEG == 32) v += __shfl_xor(v, 16, 64);
  return v;
}
template <int SEG>
__device__ __forceinline__ double sbm_seg_sum(double v) {
#pragma unroll
  for (int off = SEG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);   // stays inside the aligned segment
  return v;
}

template <class M, int SEG>
struct PackedStateRowSystem {
  static constexpr int NV = 0;           // no column rows: the lane's state component is the "extra" element
  static constexpr int NVX = 1;
  static constexpr int CPL = 1;
  static constexpr int NCS = 1;
  static constexpr bool kUniform = false;
  __device__ __forceinline__ static constexpr int col_of(int, int) { return 0; }
  static constexpr bool kDpp = (SEG == 16 || SEG == 32);   // the widths the segment reductions exist for
  double* Y;                     // [64] in LDS: stage state, segment by segment
  int lane, cls;
  bool has_row;
  int yidx[M::RL_MAXYS];
  double ps[M::RL_MAXPS];

  struct Pending { double ys[M::RL_MAXYS]; };
  struct Token { double f; };
  __device__ __forceinline__ Pending issue(double, const double (&z)[1][1]) const {
    Pending p;
    Y[lane] = z[0][0];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
    for (int s = 0; s < M::RL_MAXYS; ++s) p.ys[s] = Y[yidx[s]];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return p;
  }
  __device__ __forceinline__ Token eval(const Pending& p, double t) const {
    Token k;
    double jy[M::RL_MAXJY], jp[M::RL_MAXJP];   // dead: the compiler drops the Jacobian arithmetic
    k.f = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJY; ++s) jy[s] = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = 0.0;
    M::class_dispatch(cls, t, p.ys, ps, k.f, jy, jp);
    k.f = cls >= 0 ? k.f : 0.0;
    return k;
  }
  __device__ __forceinline__ void extra_out(const Token& k, double (&dz)[1][1]) const { dz[0][0] = k.f; }
  __device__ __forceinline__ void finish(const Token&, double, const double (&)[1][1], double (&)[1][1]) const {}
  __device__ __forceinline__ void rhs(double t, const double (&z)[1][1], double (&dz)[1][1]) const {
    extra_out(eval(issue(t, z), t), dz);
  }
  // (only the adaptive driver reduces: fixed-step runs may use any segment width)
  __device__ __forceinline__ float norm(const float (&)[1], float xsum) const {
    static_assert(kDpp, "segment reductions: widths 16 and 32");
    const float x = has_row ? sbm_nan_to_inf(xsum) : 0.f;
    return sqrtf(sbm_seg_sumf<SEG>(x) * (1.0f / M::NV));
  }
  __device__ __forceinline__ double sum(double v) const {
    static_assert(kDpp, "segment reductions: widths 16 and 32");
    return sbm_seg_sum<SEG>(v);
  }
};

template <class M, int METHOD, int SEG>
__global__ void __launch_bounds__(64) sbm_state_packed_kernel(sbm_kernel_args a) {
  using Sys = PackedStateRowSystem<M, SEG>;
  static_assert(SEG >= 4 && SEG <= 32 && SEG % 4 == 0 && M::NV <= SEG, "one state component per lane of a segment");
  constexpr int TPW = 64 / SEG;          // trajectories per wavefront (lanes beyond TPW * SEG idle along with the last one)
  __shared__ double Ysh[64];
  const int lane = threadIdx.x;
  const int seg = lane / SEG < TPW ? lane / SEG : TPW - 1;
  const int li = lane - seg * SEG;               // >= SEG on the idle tail lanes
  const int traj_raw = (int)blockIdx.x * TPW + seg;
  const bool live = traj_raw < a.n_traj && li < SEG;   // segments beyond the batch integrate a copy of the last trajectory
  const int traj = traj_raw < a.n_traj ? traj_raw : a.n_traj - 1;
  Ysh[lane] = 0.0;
  Sys sys;
  sys.Y = Ysh;
  sys.lane = lane;
  sys.has_row = li < M::NV;
  const int row = sys.has_row ? li : 0;
  sys.cls = sys.has_row ? M::rl_class(row) : -1;
  const double* P = a.P + (size_t)traj * M::NP;
#pragma unroll
  for (int s = 0; s < M::RL_MAXYS; ++s) sys.yidx[s] = M::rl_ys(s, row) + seg * SEG;
#pragma unroll
  for (int s = 0; s < M::RL_MAXPS; ++s) sys.ps[s] = P[M::rl_ps(s, row)];
  __syncthreads();
  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;
  double z[1][1];
  z[0][0] = (a.y0 && sys.has_row) ? a.y0[li] : 0.0;
  double* Yt = a.Y + (size_t)traj * a.n_t * M::NV;
  auto store = [&](int io, const double (&zz)[1][1]) {
    if (sys.has_row && live) Yt[(size_t)io * M::NV + li] = zz[0][0];
  };
  SbmTrajOut r;
  r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);

  if (li == 0 && live) {
    if (a.status) a.status[traj] = r.status;
    if (a.n_steps) a.n_steps[traj] = r.n_acc;
    if (a.n_reject) a.n_reject[traj] = r.n_rej;
  }
}

// ===========================================================================
// Row-group sensitivity kernel: the row-lane kernel with the rows of a column split over G lanes.
//
// One trajectory per wavefront, as before; the state still lives one component per lane and the
// row lanes still evaluate f / J_y / J_p by class.  What changes is the distribution of S:
// lane (g, c') = g*C + c' owns rows [g*RPG, (g+1)*RPG) of the columns c', c'+C, ... (CPL of them),
// RPG*CPL elements instead of NV.  For the 20-state / 40-parameter cascade that is 14 elements
// on 60 lanes instead of 20 elements on 40 lanes: fewer instructions per step in proportion and a
// Runge-Kutta working set that (nearly) fits the 256 architectural VGPRs, so the v_accvgpr
// traffic of the row-lane kernel goes away.  Price: J_y coefficients are per-lane values now (LDS
// table JYL instead of v_readlane scalars) and terms that cross a group boundary go through an LDS
// halo (emit_rowgroup.py).  Still a 64-thread workgroup: wave-local LDS ordering, no barriers.
//
// Column chunks (L::RG_NCH > 1; L = the layout, M::RG0 or M::RG1): the columns of S are coupled only through the state, so a trajectory with
// more columns than one wavefront can hold in registers is cut into RG_NCH chunks of C*CPL columns;
// blockIdx.y = chunk, each chunk integrates (state, its columns) on its own -- own error norm, own step
// sequence, nothing exchanged.  Chunk 0 stores the state; status / step counts are combined with atomicMax
// (the launcher zeroes them first).
// ===========================================================================
#ifndef SBM_RG_LANE_SPARE
#define SBM_RG_LANE_SPARE 0
#endif
#if SBM_RG_LANE_SPARE
#define SBM_RG_SPARE(lane) (lane)
#else
#define SBM_RG_SPARE(lane) 1
#endif
template <class M, class L>
struct SbmRowGroupShared {
  // JYL rows: G groups of RPG (the last one padded), then one more all-padding group for the idle
  // lanes (>= G*C): nothing is ever written there, it stays zero
  static constexpr int NPAD = L::RG_G * L::RG_RPG;
  static constexpr int NROWS = NPAD + L::RG_RPG;
  static constexpr int LS = L::RG_LS;    // A / H are [local row][lane][cc]; element (i, j) at L::rg_pos(i, j)
  static constexpr int ZPOS = L::RG_RPG * LS;   // a slot of H that stays zero (absent halo terms)
  static constexpr int RPL = (M::NV + 63) / 64;   // state rows per lane (rows lane, lane + 64, ...)
  double Y[64 * RPL];                    // stage state, one component per (row lane, r)
  // (+ one spare slot PER LANE for the lanes without a row / the slots without an entry: 44 idle lanes storing to one
  //  address are a 16-way conflict on every store of the hand-over, round 4's PMC pass)
  alignas(16) double JYL[NROWS * L::RG_JYS + 64];  // J_y coefficients [row][term]
  alignas(16) double A[L::RG_RPG * LS + 64];       // J_p entries; idle / padded slots stay 0
  alignas(16) double H[L::RG_RPG * LS + 4];        // published rows of the stage vector (+ zero slot)
};

// EARLY: fetch the lane's A / J_y operands right after the row lanes published them (eval) instead of
// in finish, so that the column rows of the stage vector are formed while they are in flight.  Costs
// 2*(NV + RPG*JYS) more live VGPRs: pays for RK4 (18.0 vs 19.4 ms), spills for DOPRI45 (20.7 vs 7.5 ms).
template <class M, class L, bool EARLY>
struct RowGroupSystem {
  static constexpr int G = L::RG_G, C = L::RG_C, RPG = L::RG_RPG;
  static constexpr int NV = L::RG_RPG * L::RG_CPL;   // elements of S this lane integrates
  static constexpr int RPL = (M::NV + 63) / 64;      // state rows this lane evaluates: lane, lane + 64, ...
  static constexpr int NVX = NV + RPL;               // + this lane's own state component(s)
  static constexpr int CPL = 1;
  static constexpr int NCS = L::RG_CPL;
  __device__ __forceinline__ static constexpr int col_of(int, int i) { return i / L::RG_RPG; }
  static constexpr int NH = L::RG_NHALO > 0 ? L::RG_NHALO : 1;
  SbmRowGroupShared<M, L>* sh;
  int lane, grp, cp;             // lane = grp*C + cp on the active lanes
  int cbase;                     // first column of this wavefront's chunk
  bool active;                   // lane < G*C
  int cls[RPL];
  int yidx[RPL][M::RL_MAXYS];
  double ps[RPL][M::RL_MAXPS];
  int jypos[RPL][M::RL_MAXJY];
  int apos[RPL][M::RL_MAXJP];
  const double* a_lane;
  const double* jy_lane;
  double* h_lane;
  int hoff[NH];

  __device__ __forceinline__ static void lds_order() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

  struct Token {
    double f[RPL];
    double acol[EARLY ? NV : 1], coef[EARLY ? RPG * L::RG_JYS : 1];
  };
  struct Pending { double ys[RPL][M::RL_MAXYS]; };
  __device__ __forceinline__ Pending issue(double /*t*/, const double (&z)[1][NVX]) const {
    Pending p;
#pragma unroll
    for (int r = 0; r < RPL; ++r) sh->Y[lane + 64 * r] = z[0][NV + r];
    lds_order();
#pragma unroll
    for (int r = 0; r < RPL; ++r)
#pragma unroll
      for (int s = 0; s < M::RL_MAXYS; ++s) p.ys[r][s] = sh->Y[yidx[r][s]];
    lds_order();
    return p;
  }
  __device__ __forceinline__ Token eval(const Pending& p, double t) const {
    Token k;
    double jy[RPL][M::RL_MAXJY], jp[RPL][M::RL_MAXJP];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      k.f[r] = 0.0;
#pragma unroll
      for (int s = 0; s < M::RL_MAXJY; ++s) jy[r][s] = 0.0;
#pragma unroll
      for (int s = 0; s < M::RL_MAXJP; ++s) jp[r][s] = 0.0;
      M::class_dispatch(cls[r], t, p.ys[r], ps[r], k.f[r], jy[r], jp[r]);
      k.f[r] = cls[r] >= 0 ? k.f[r] : 0.0;   // lanes without a row come out of the select chain with the last class's value
    }
    lds_order();
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
#pragma unroll
      for (int s = 0; s < M::RL_MAXJP; ++s) sh->A[apos[r][s]] = jp[r][s];
#pragma unroll
      for (int s = 0; s < M::RL_MAXJY; ++s) sh->JYL[jypos[r][s]] = jy[r][s];
    }
    lds_order();
    if constexpr (EARLY) {
      L::load_rowgroup(a_lane, jy_lane, k.acol, k.coef);
      lds_order();
    }
    return k;
  }
  __device__ __forceinline__ void finish(const Token& k, double /*t*/, const double (&z)[1][NVX],
                                         double (&dz)[1][NVX]) const {
    double zc[NV], dc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) zc[i] = z[0][i];
    L::publish_rowgroup(h_lane, zc);
    lds_order();
    if constexpr (EARLY) {
      L::apply_rowgroup(k.acol, k.coef, sh->H, hoff, zc, dc);
    } else {
      double acol[NV], coef[RPG * L::RG_JYS];
      L::load_rowgroup(a_lane, jy_lane, acol, coef);
      L::apply_rowgroup(acol, coef, sh->H, hoff, zc, dc);
    }
    lds_order();
#pragma unroll
    for (int i = 0; i < NV; ++i) dz[0][i] = dc[i];
  }
  __device__ __forceinline__ void extra_out(const Token& k, double (&dz)[1][NVX]) const {
#pragma unroll
    for (int r = 0; r < RPL; ++r) dz[0][NV + r] = k.f[r];
  }
  __device__ __forceinline__ void rhs(double t, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    const Token k = eval(issue(t, z), t);
    extra_out(k, dz);
    finish(k, t, z, dz);
  }
  // max( RMS of the state error, max over columns of the column RMS ); a column's sum of squares
  // is spread over the G lanes cp, C + cp, ...
  __device__ __forceinline__ float norm(const float (&colsum)[NCS], float xsum) const {
    float m = 0.f;
#pragma unroll
    for (int cc = 0; cc < NCS; ++cc) {
      const float v = active ? sbm_nan_to_inf(colsum[cc]) : 0.f;
      float tot = v;
#pragma unroll
      for (int gg = 1; gg < G; ++gg) {
        int src = lane + gg * C;
        src = src >= G * C ? src - G * C : src;
        tot += __shfl(v, active ? src : lane, 64);
      }
      const bool has_col = active && ((L::RG_NCH > 1 ? cbase : 0) + cp + C * cc < M::NK);
      m = fmaxf(m, has_col ? tot : 0.f);
    }
    const float x = (lane < M::NV) ? sbm_nan_to_inf(xsum) : 0.f;
    const float mx = sbm_wave_max(m);
    const float xs = sbm_wave_sumf(x);
    return sqrtf(fmaxf(mx, xs) * (1.0f / M::NV));   // +inf: the driver rejects the step
  }
  __device__ __forceinline__ double sum(double v) const { return sbm_wave_sum(v); }
};

template <class M, class L, int METHOD>
// Two waves per SIMD (256 registers each) when the lane's share of S is small enough for DOPRI45's seven
// stage vectors to fit: 2*(RPG*CPL + 1)*7 + operands <= 256 holds up to 15 elements (cascade20: 14 + 1).
// Larger shares get the whole register file (one wave per SIMD) rather than spill.
#ifndef SBM_RG_MIN_WAVES
#define SBM_RG_MIN_WAVES (METHOD == SBM_DOP853 ? ((L::RG_RPG * L::RG_CPL + (M::NV + 63) / 64 <= 8) ? 2 : 1) \
                          : ((L::RG_RPG * L::RG_CPL + (M::NV + 63) / 64 <= 15 || METHOD == SBM_RK4_FIXED) ? 2 : 1))
#endif
__global__ void __launch_bounds__(64, SBM_RG_MIN_WAVES) sbm_sens_rowgroup_kernel(sbm_kernel_args a) {
  using Sys = RowGroupSystem<M, L, METHOD == SBM_RK4_FIXED>;
  using Sh = SbmRowGroupShared<M, L>;
  constexpr int MNV = M::NV, NK = M::NK;
  constexpr int G = L::RG_G, C = L::RG_C, RPG = L::RG_RPG, CPL = L::RG_CPL;
  constexpr int NE = Sys::NV, NVX = Sys::NVX, NPAD = Sh::NPAD;
  constexpr int NCH = L::RG_NCH;
  constexpr int RPL = Sys::RPL;
  static_assert(G * C <= 64 && C * CPL * NCH >= NK && C * CPL * (NCH - 1) < NK && NPAD >= MNV, "row-group layout");
  __shared__ Sh sh;
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x;
  const int chunk = NCH > 1 ? (int)blockIdx.y : 0;
  const int cbase = chunk * (C * CPL);

  constexpr int NROWS = Sh::NROWS;
  constexpr int LS = Sh::LS;
  for (int i = lane; i < RPG * LS + 64; i += 64) sh.A[i] = 0.0;
  for (int i = lane; i < RPG * LS + 4; i += 64) sh.H[i] = 0.0;
  for (int i = lane; i < NROWS * L::RG_JYS + 64; i += 64) sh.JYL[i] = 0.0;
#pragma unroll
  for (int r = 0; r < RPL; ++r) sh.Y[lane + 64 * r] = 0.0;

  Sys sys;
  sys.sh = &sh;
  sys.lane = lane;
  sys.active = lane < G * C;
  sys.grp = sys.active ? lane / C : G;   // idle lanes form the all-padding group: zeros throughout
  sys.cp = lane - sys.grp * C;
  sys.cbase = cbase;
  const double* P = a.P + (size_t)traj * M::NP;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const bool has_row = lane + 64 * r < MNV;
    const int row = has_row ? lane + 64 * r : 0;
    sys.cls[r] = has_row ? M::rl_class(row) : -1;
#pragma unroll
    for (int s = 0; s < M::RL_MAXYS; ++s) sys.yidx[r][s] = M::rl_ys(s, row);
#pragma unroll
    for (int s = 0; s < M::RL_MAXPS; ++s) sys.ps[r][s] = P[M::rl_ps(s, row)];
#pragma unroll
    for (int s = 0; s < M::RL_MAXJY; ++s) {
      const int jp_ = L::rg_jypos(s, row);
      sys.jypos[r][s] = (has_row && jp_ < NPAD * L::RG_JYS) ? jp_ : NROWS * L::RG_JYS + SBM_RG_SPARE(lane);   // else: spare slot
    }
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) {
      if constexpr (NCH == 1 && RPL == 1) {
        const int ap = M::rl_apos(s, row);                      // row*64 + column; unused slots carry MNV*64
        sys.apos[r][s] = (has_row && ap < MNV * 64) ? L::rg_pos(ap >> 6, ap & 63) : RPG * LS + SBM_RG_SPARE(lane);
      } else {
        const int lc = M::rl_jpcol(s, row) - cbase;             // column within this chunk (unused slots: -1)
        sys.apos[r][s] = (has_row && lc >= 0 && lc < C * CPL && lc + cbase < NK) ? L::rg_pos(row, lc) : RPG * LS + SBM_RG_SPARE(lane);
      }
    }
  }
  sys.a_lane = sh.A + CPL * lane;
  sys.jy_lane = sh.JYL + (sys.grp * RPG) * L::RG_JYS;
  sys.h_lane = sh.H + CPL * lane;
#pragma unroll
  for (int t = 0; t < Sys::NH; ++t) {
    const int src = (L::RG_NHALO > 0 && sys.active) ? L::rg_hsrc(t, sys.grp) : NPAD;   // NPAD: term absent
    sys.hoff[t] = src < NPAD ? L::rg_pos(src, sys.cp) : Sh::ZPOS;
  }
  __syncthreads();

  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;

  // element (r, cc) of this lane = S[grp*RPG + r][cp + C*cc]
  double z[1][NVX];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc)
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
      const int grow = sys.grp * RPG + r, col = cbase + sys.cp + C * cc;
      const bool valid = sys.active && grow < MNV && col < NK;
      z[0][r + RPG * cc] = (a.s0 && valid) ? a.s0[grow * NK + col] : 0.0;
    }
#pragma unroll
  for (int r = 0; r < RPL; ++r) z[0][NE + r] = (a.y0 && lane + 64 * r < MNV) ? a.y0[lane + 64 * r] : 0.0;

  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * MNV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * MNV * NK : nullptr;
  auto store = [&](int io, const double (&zz)[1][NVX]) {
    if (Yt && chunk == 0) {
#pragma unroll
      for (int r = 0; r < RPL; ++r)
        if (lane + 64 * r < MNV) Yt[(size_t)io * MNV + lane + 64 * r] = zz[0][NE + r];
    }
    if (St) {
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc)
#pragma unroll
        for (int r = 0; r < RPG; ++r) {
          const int grow = sys.grp * RPG + r, col = cbase + sys.cp + C * cc;
          if (sys.active && grow < MNV && col < NK) St[((size_t)io * MNV + grow) * NK + col] = zz[0][r + RPG * cc];
        }
    }
  };

  SbmTrajOut r;
  r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);


  if (lane == 0) {
    if constexpr (NCH > 1) {
      // worst status, most steps over the chunks (all non-negative; zeroed by the launcher)
      if (a.status) atomicMax(a.status + traj, r.status);
      if (a.n_steps) atomicMax(a.n_steps + traj, r.n_acc);
      if (a.n_reject) atomicMax(a.n_reject + traj, r.n_rej);
    } else {
      if (a.status) a.status[traj] = r.status;
      if (a.n_steps) a.n_steps[traj] = r.n_acc;
      if (a.n_reject) a.n_reject[traj] = r.n_rej;
    }
  }
}

// ===========================================================================
// Packed row-lane sensitivity kernel: several trajectories per wavefront, for SMALL models.
//
// The reference's own fixtures have 1 - 2 state variables and 2 - 5 parameters: in the row-lane kernel 3 - 5 of the 64
// lanes carry equations and every trajectory costs a whole wavefront's issue slots.  Here a wavefront is cut into
// 64 / SEG segments of SEG = 4, 8, 16 or 32 lanes (SEG >= n_vars and >= n_sens), one trajectory each: lane li of a
// segment owns sensitivity column li (all NV rows) and state component li.  As in the packed state-only kernel the
// segments share the instruction stream but not the control flow: time, step size and accept / reject decisions are
// per-lane values, uniform within a segment; the controller's reductions stay inside a segment and return the same
// bits in every lane of it (exchange-symmetric DPP / butterfly levels).  A row lane hands its J_y entries to the
// segment's columns through a per-segment LDS table (a lane cannot name "its" row lane by a literal v_readlane index
// when the segment base varies) and its J_p entries through the segment's slice of A.  A trajectory's result does not
// depend on which trajectories share its wavefront (tested bitwise).  Used from 2048 trajectories on
// (SBM_VARIANT_PACKED forces it).
// ===========================================================================
template <int SEG>
__device__ __forceinline__ float sbm_seg_sumf_any(float v) {
  v += sbm_dpp<0xb1, 0xf>(v);                              // quad_perm:[1,0,3,2]
  v += sbm_dpp<0x4e, 0xf>(v);                              // quad_perm:[2,3,0,1]
  if constexpr (SEG >= 8) v += sbm_dpp<0x141, 0xf>(v);     // row_half_mirror
  if constexpr (SEG >= 16) v += sbm_dpp<0x140, 0xf>(v);    // row_mirror
  if constexpr (SEG == 32) v += __shfl_xor(v, 16, 64);
  return v;
}
template <int SEG>
__device__ __forceinline__ float sbm_seg_maxf_any(float v) {
  v = fmaxf(v, sbm_dpp<0xb1, 0xf>(v));
  v = fmaxf(v, sbm_dpp<0x4e, 0xf>(v));
  if constexpr (SEG >= 8) v = fmaxf(v, sbm_dpp<0x141, 0xf>(v));
  if constexpr (SEG >= 16) v = fmaxf(v, sbm_dpp<0x140, 0xf>(v));
  if constexpr (SEG == 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
  return v;
}

template <class M, int SEG>
struct SbmPackedSensShared {
  static constexpr int TPW = 64 / SEG;
  double Y[64];                                   // stage state, segment by segment
  double JYL[TPW * M::NV * M::RL_MAXJY + 2];      // [segment][row][slot] (+ spare slot)
  double A[TPW * M::NV * SEG + 2];                // [segment][row][column of the segment] (+ spare slot)
};

template <class M, int SEG>
struct PackedRowLaneSystem {
  static constexpr int NV = M::NV;
  static constexpr int NVX = M::NV + 1;
  static constexpr int CPL = 1;
  static constexpr int NCS = 1;
  static constexpr bool kUniform = false;
  __device__ __forceinline__ static constexpr int col_of(int, int) { return 0; }
  SbmPackedSensShared<M, SEG>* sh;
  int lane, li, cls;
  bool has_row, has_col;
  int yidx[M::RL_MAXYS], jypos[M::RL_MAXJY], apos[M::RL_MAXJP];
  double ps[M::RL_MAXPS];
  const double* jyl;        // this segment's J_y table
  const double* acolp;      // this lane's column of the segment's A: acolp[i * SEG]

  __device__ __forceinline__ static void lds_order() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }
  struct Token { double f; };
  struct Pending { double ys[M::RL_MAXYS]; };
  __device__ __forceinline__ Pending issue(double, const double (&z)[1][NVX]) const {
    Pending p;
    sh->Y[lane] = z[0][NV];
    lds_order();
#pragma unroll
    for (int s = 0; s < M::RL_MAXYS; ++s) p.ys[s] = sh->Y[yidx[s]];
    lds_order();
    return p;
  }
  __device__ __forceinline__ Token eval(const Pending& p, double t) const {
    Token k;
    double jy[M::RL_MAXJY], jp[M::RL_MAXJP];
    k.f = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJY; ++s) jy[s] = 0.0;
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) jp[s] = 0.0;
    M::class_dispatch(cls, t, p.ys, ps, k.f, jy, jp);
    k.f = cls >= 0 ? k.f : 0.0;
    lds_order();
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) sh->A[apos[s]] = jp[s];
#pragma unroll
    for (int s = 0; s < M::RL_MAXJY; ++s) sh->JYL[jypos[s]] = jy[s];
    lds_order();
    return k;
  }
  __device__ __forceinline__ void finish(const Token&, double, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    double zc[NV], dc[NV], acol[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { zc[i] = z[0][i]; acol[i] = acolp[i * SEG]; }
    lds_order();
    M::apply_lds(jyl, acol, zc, dc);
    lds_order();
#pragma unroll
    for (int i = 0; i < NV; ++i) dz[0][i] = dc[i];
  }
  __device__ __forceinline__ void extra_out(const Token& k, double (&dz)[1][NVX]) const { dz[0][NV] = k.f; }
  __device__ __forceinline__ void rhs(double t, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    const Token k = eval(issue(t, z), t);
    extra_out(k, dz);
    finish(k, t, z, dz);
  }
  __device__ __forceinline__ float norm(const float (&colsum)[1], float xsum) const {
    const float m = has_col ? sbm_nan_to_inf(colsum[0]) : 0.f;
    const float x = has_row ? sbm_nan_to_inf(xsum) : 0.f;
    return sqrtf(fmaxf(sbm_seg_maxf_any<SEG>(m), sbm_seg_sumf_any<SEG>(x)) * (1.0f / NV));
  }
  __device__ __forceinline__ double sum(double v) const { return sbm_seg_sum<SEG>(v); }
};

template <class M, int METHOD, int SEG>
__global__ void __launch_bounds__(64) sbm_sens_packed_kernel(sbm_kernel_args a) {
  using Sys = PackedRowLaneSystem<M, SEG>;
  using Sh = SbmPackedSensShared<M, SEG>;
  constexpr int NV = M::NV, NVX = NV + 1, NK = M::NK, TPW = 64 / SEG;
  static_assert(SEG >= 4 && SEG <= 32 && (SEG & (SEG - 1)) == 0 && NV <= SEG && NK <= SEG, "packed sensitivity kernel: a row and a column per lane of a segment");
  __shared__ Sh sh;
  const int lane = threadIdx.x;
  const int seg = lane / SEG, li = lane - seg * SEG;
  const int traj_raw = (int)blockIdx.x * TPW + seg;
  const bool live = traj_raw < a.n_traj;            // segments beyond the batch integrate a copy of the last trajectory
  const int traj = live ? traj_raw : a.n_traj - 1;
  for (int i = lane; i < TPW * NV * M::RL_MAXJY + 2; i += 64) sh.JYL[i] = 0.0;
  for (int i = lane; i < TPW * NV * SEG + 2; i += 64) sh.A[i] = 0.0;
  sh.Y[lane] = 0.0;

  Sys sys;
  sys.sh = &sh;
  sys.lane = lane;
  sys.li = li;
  sys.has_row = li < NV;
  sys.has_col = li < NK;
  const int row = sys.has_row ? li : 0;
  sys.cls = sys.has_row ? M::rl_class(row) : -1;
  const double* P = a.P + (size_t)traj * M::NP;
#pragma unroll
  for (int s = 0; s < M::RL_MAXYS; ++s) sys.yidx[s] = M::rl_ys(s, row) + seg * SEG;
#pragma unroll
  for (int s = 0; s < M::RL_MAXPS; ++s) sys.ps[s] = P[M::rl_ps(s, row)];
#pragma unroll
  for (int s = 0; s < M::RL_MAXJY; ++s)
    sys.jypos[s] = (sys.has_row && M::rl_jycol(s, row) >= 0) ? (seg * NV + row) * M::RL_MAXJY + s : TPW * NV * M::RL_MAXJY + 1;
#pragma unroll
  for (int s = 0; s < M::RL_MAXJP; ++s) {
    const int c = M::rl_jpcol(s, row);
    sys.apos[s] = (sys.has_row && c >= 0) ? (seg * NV + row) * SEG + c : TPW * NV * SEG + 1;
  }
  sys.jyl = sh.JYL + seg * NV * M::RL_MAXJY;
  sys.acolp = sh.A + seg * NV * SEG + li;
  __syncthreads();

  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;

  double z[1][NVX];
#pragma unroll
  for (int i = 0; i < NV; ++i) z[0][i] = (a.s0 && sys.has_col) ? a.s0[i * NK + li] : 0.0;
  z[0][NV] = (a.y0 && sys.has_row) ? a.y0[li] : 0.0;

  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
  auto store = [&](int io, const double (&zz)[1][NVX]) {
    if (Yt && sys.has_row && live) Yt[(size_t)io * NV + li] = zz[0][NV];
    if (St && sys.has_col && live) {
#pragma unroll
      for (int i = 0; i < NV; ++i) St[((size_t)io * NV + i) * NK + li] = zz[0][i];
    }
  };
  SbmTrajOut r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);
  if (li == 0 && live) {
    if (a.status) a.status[traj] = r.status;
    if (a.n_steps) a.n_steps[traj] = r.n_acc;
    if (a.n_reject) a.n_reject[traj] = r.n_rej;
  }
}

// v[lane] of a per-lane array held in registers: a binary select tree over the bits of `lane`.  The
// obvious chain of `lane == i` selects makes the compiler keep N loop-invariant 64-bit lane masks in
// SGPRs (100 of them for N = 50: it spilled them through v_writelane and AGPRs); the tree needs
// log2(N) masks for the same number of selects.
template <int N, int BIT = 0>
__device__ __forceinline__ double sbm_pick_tree(const double (&v)[N], int lane) {
  if constexpr (N == 1) {
    return v[0];
  } else {
    constexpr int H = (N + 1) / 2;
    double t[H];
    const bool odd = ((lane >> BIT) & 1) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) t[j] = (2 * j + 1 < N) ? sbm_sel(odd, v[2 * j + 1 < N ? 2 * j + 1 : 0], v[2 * j]) : v[2 * j];
    return sbm_pick_tree<H, BIT + 1>(t, lane);
  }
}

// ===========================================================================
// Implicit midpoint for stiff systems (BASELINE configs[4]), state + forward sensitivities, FIXED step.
//
// The step itself (Newton on the midpoint, one linear solve per sensitivity column with the factored Newton matrix,
// the lane mapping, models with more than 64 state variables) is sbm_implicit_stepper.hpp.  A-stable and symmetric
// (second order, error expansion in h^2: the caller extrapolates two runs, or uses SBM_IMPLICIT_ADAPTIVE, which does
// that inside the kernel).  Fixed step h0 between output times like RK4; opts.rtol / atol are the Newton tolerances.
// More than 64 columns: blockIdx.y = chunk of 64 columns, each wavefront repeating the (identical, fixed-step)
// state iteration for its own columns; chunk 0 stores the state.
// ===========================================================================
#include "sbm_implicit_stepper.hpp"

template <class M>
struct SbmImidShared {
  // J_p reaches the columns through a compact per-row table (RL_MAXJP values, picked by column index) when rows have few
  // parameter entries -- round 2 kept the dense [row][64] image here (25.6 KB for 50 rows: five wavefronts per CU); with
  // the compact table and the fused Newton update (256 VGPRs) two wavefronts share a SIMD
  static constexpr bool A_SPARSE = (M::RL_MAXJP <= 4);
  static constexpr int NROW = 64 * ((M::NV + 63) / 64);
  static constexpr int A_SIZE = A_SPARSE ? (M::NV * M::RL_MAXJP + 2) : (M::NV * 64 + 2);
  double Y[NROW];               // iterate, one component per row lane (rows lane, lane + 64, ...)
  double G[NROW];               // Newton residual
  double JY[sbm_ijy_size<M>()]; // J_y non-zeros by entry index (+ spare slot): the redundant factorisation's input only
  double A[A_SIZE];             // A[i][c] = J_p[i][c] (+ spare slot)
  static constexpr int MF_SIZE = sbm_imf_size<M>(), RD_SIZE = sbm_ird_size<M>();
  __attribute__((aligned(16))) double MF[MF_SIZE];   // the factors (IM_TRI: reciprocal pivots and scaled entries; IM_DIST: dense rows)
  double RD[RD_SIZE];           // IM_DIST: reciprocal pivots
};

template <class M>
__global__ void __launch_bounds__(64) sbm_imid_kernel(sbm_kernel_args a) {
  constexpr int NV = M::NV, NK = M::NK;
  constexpr int NCH = (NK + 63) / 64;
  using Stepper = SbmImplicitStepper<M, SbmImidShared<M>>;
  constexpr int RPL = Stepper::RPL;
  __shared__ SbmImidShared<M> sh;
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x;
  const int chunk = NCH > 1 ? (int)blockIdx.y : 0;
  const int col = lane + 64 * chunk;       // this lane's column of S
  const bool has_col = col < NK;
  Stepper st;
  st.setup(&sh, lane, chunk, a.P + (size_t)traj * M::NP);
  __syncthreads();

  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;
  const bool with_sens = a.S != nullptr;   // wave-uniform
  const bool graded = a.opts.method == SBM_IMPLICIT_MIDPOINT_GRADED;

  double z[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) z[i] = (a.s0 && has_col) ? a.s0[i * NK + col] : 0.0;
  double y[RPL], dy_prev[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    y[r] = (a.y0 && st.has_row[r]) ? a.y0[lane + 64 * r] : 0.0;
    dy_prev[r] = 0.0;        // increment of the previous step (this lane's components)
  }

  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
  const double rtol = a.opts.rtol > 0.0 ? a.opts.rtol : 1e-10, atol = a.opts.atol > 0.0 ? a.opts.atol : 1e-12;
  const double h0 = a.opts.h0;
  const int max_steps = a.opts.max_steps > 0 ? a.opts.max_steps : 1000000000;
  int status = SBM_OK, n_acc = 0, n_newton = 0;
  double t = a.opts.t0;
  bool failed = !(h0 > 0.0);
  if (failed) status = SBM_STEP_UNDERFLOW;
  double hs_prev = 0.0;      // step size of the previous step

  for (int io = 0; io < glen; ++io) {
    const double target = tg[io];
    const double dt = target - t;
    if (!failed && dt > 0.0) {
      const double nd = ceil(dt / h0 - 1e-9) * (a.opts.step_mult > 0 ? a.opts.step_mult : 1);
      const int ns = nd < 1.0 ? 1 : (nd > 2.0e9 ? 2000000000 : (int)nd);
      if (n_acc + ns > max_steps) { status = SBM_MAX_STEPS; failed = true; }
      if (!failed) {
        const double hs = dt / ns;
        const double t0 = t;
        // Newton starts from the midpoint the previous step's increment predicts (free, and good for the
        // smooth solutions a fixed step resolves): 2.6 -> about 2 iterations per step on stiff50
        const double hsc = (hs_prev > 0.0) ? hs / hs_prev : 0.0;
#pragma unroll
        for (int r = 0; r < RPL; ++r) dy_prev[r] *= hsc;
        hs_prev = hs;
        for (int s = 0; s < ns && !failed; ++s) {
         // SBM_IMPLICIT_MIDPOINT_GRADED: the very first step of a trajectory is cut into GRADE + 1 midpoint
         // substeps of sizes hs * 2^-GRADE, 2^-GRADE, 2^-(GRADE-1), ..., 1/2 (they add up to hs).  An initial
         // condition off a fast manifold -- the reference always starts from y = 0 -- produces a layer far
         // thinner than any affordable fixed step; no one-step method integrates through it accurately
         // without resolving it (a backward-Euler start-up damps the layer but misses its area: measured 60x
         // WORSE than doing nothing).  The geometric grading resolves layers down to hs / 4096 for 12 extra
         // steps.  Richardson needs NESTED grids to keep the h^2 expansion (an irregular first step graded relative
         // to each run's own hs leaves an h^3 term: measured third-order convergence of the extrapolants): the
         // pattern belongs to the first BASE step (step_mult = 1), and a run with step_mult = m cuts every one
         // of its substeps into m equal parts -- m substeps of each size hs * 2^-k, covering the first m steps.
         constexpr int GRADE = 12;
         const int gm_ = a.opts.step_mult > 0 ? a.opts.step_mult : 1;
         const bool grade_now = graded && n_acc == 0;                  // wave-uniform
         const int nsub = grade_now ? (GRADE + 1) * gm_ : 1;
         double t_sub = fma((double)s, hs, t0);
         for (int sub = 0; sub < nsub && !failed; ++sub) {
          const int gj = sub / gm_;                                    // which size of the pattern
          const double hsub = nsub == 1 ? hs : ldexp(hs, -(gj == 0 ? GRADE : GRADE - gj + 1));
          const double hh = 0.5 * hsub;
          const double tm = t_sub + hh;
          t_sub += hsub;
          double yb[RPL];
#pragma unroll
          for (int r = 0; r < RPL; ++r) yb[r] = fma(0.5, dy_prev[r], y[r]);
          const int rc = st.template newton<12>(tm, hh, y, yb, rtol, atol, n_newton);
          if (rc != SBM_OK) { status = rc; failed = true; break; }
          // predictor for the next (sub)step: this increment, rescaled when the next substep is twice as long
          const double psc = (nsub > 1 && gj > 0 && sub % gm_ == gm_ - 1) ? 2.0 : 1.0;
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            dy_prev[r] = 2.0 * (yb[r] - y[r]) * psc;
            y[r] = fma(2.0, yb[r], -y[r]);
          }
          if (with_sens) st.sens(hh, z);
         }
          if (!failed) {
            // the graded block covered the first gm_ steps of this interval
            n_acc += grade_now ? gm_ : 1;
            if (grade_now) s += gm_ - 1;
          }
        }
        if (!failed) t = target;
      }
    }
    if (failed) {
#pragma unroll
      for (int i = 0; i < NV; ++i) z[i] = __builtin_nan("");
#pragma unroll
      for (int r = 0; r < RPL; ++r) y[r] = __builtin_nan("");
    }
    if (Yt && chunk == 0) {
#pragma unroll
      for (int r = 0; r < RPL; ++r)
        if (st.has_row[r]) Yt[(size_t)io * NV + lane + 64 * r] = y[r];
    }
    if (St && has_col) {
#pragma unroll
      for (int i = 0; i < NV; ++i) St[((size_t)io * NV + i) * NK + col] = z[i];
    }
  }
  if (lane == 0 && chunk == 0) {   // the state iteration, hence status and counts, is the same in every chunk
    if (a.status) a.status[traj] = status;
    if (a.n_steps) a.n_steps[traj] = n_acc;
    if (a.n_reject) a.n_reject[traj] = n_newton - n_acc;   // Newton iterations beyond one per step
  }
}

#include "sbm_implicit_adaptive.hpp"
#include "sbm_implicit_extrap.hpp"
#include "sbm_implicit_extrap_seq.hpp"
#include "sbm_sens_mfma.hpp"

// What the implicit kernels can hold: every lane keeps a whole column of S (NV values) and the solver's work vector
// in registers (4 NV VGPRs: beyond about 110 state variables the compiler spills), the error-controlled kernel parks two
// more copies of S in LDS.
template <class M>
struct SbmImplicitFits {
  static constexpr bool fixed = M::NV <= SBM_IMPLICIT_MAX_NV && sizeof(SbmImidShared<M>) <= 160u * 1024u;
  static constexpr bool adaptive = M::NV <= SBM_IMPLICIT_MAX_NV && sizeof(SbmImadShared<M>) <= 160u * 1024u;
};

// ---------------------------------------------------------------------------
// host-side launcher used by sbm_plugin_main.hip
// ---------------------------------------------------------------------------
template <class T>
struct SbmTypeTag { using type = T; };

// ---- scratch of the persistent kernels: one buffer per (device, stream), grown on demand, never shrunk.  Launches on
// one stream run one after the other, so they can share it; two contexts on two streams get one each. ----
#include <map>
#include <mutex>
#include <utility>
#include <stdlib.h>
#include <stdio.h>
struct SbmScratch {
  void* p = nullptr;
  size_t bytes = 0;
  int* counter = nullptr;      // the work counter of a persistent launch (zeroed on the stream before every launch)
};
static hipError_t sbm_scratch_for(hipStream_t stream, size_t need, SbmScratch** out) {
  static std::mutex mu;
  static std::map<std::pair<int, void*>, SbmScratch> table;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  SbmScratch& s = table[std::make_pair(dev, (void*)stream)];
  if (!s.counter) {
    e = hipMalloc((void**)&s.counter, 256);
    if (e != hipSuccess) { s.counter = nullptr; return e; }
  }
  if (s.bytes < need) {
    // work enqueued earlier on this stream may still read the old buffer
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr;
    s.bytes = 0;
    e = hipMalloc(&s.p, need);
    if (e != hipSuccess) { s.p = nullptr; return e; }
    s.bytes = need;
  }
  *out = &s;
  return hipSuccess;
}
// how many workgroups of `kernel` the device holds at once (cached per kernel and device)
static hipError_t sbm_resident_blocks(const void* kernel, int block, int* out) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> cache;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(std::make_pair(dev, kernel));
  if (it == cache.end()) {
    int per_cu = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    it = cache.emplace(std::make_pair(dev, kernel), per_cu * cus).first;
  }
  *out = it->second;
  return hipSuccess;
}
// developer A/B switch: SBM_IEX_SEQ=0 runs chain models through sbm_iex_kernel as round 3 did
static bool sbm_iex_seq_enabled() {
  static const bool on = [] { const char* v = getenv("SBM_IEX_SEQ"); return !(v && v[0] == '0'); }();
  return on;
}

template <class M>
static int sbm_launch_model(int kind, const sbm_kernel_args* args, hipStream_t stream) {
  const sbm_kernel_args a = *args;
  if (a.n_traj <= 0) return (int)hipSuccess;
  if (a.opts.method == SBM_IMPLICIT_EXTRAP) {
    if constexpr (SbmIexFits<M>::value) {
      const int nch = a.S ? (M::NK + 63) / 64 : 1;
      if (nch > 1) {     // chunks combine status / counts with atomicMax
        hipError_t e = hipSuccess;
        if (a.status) e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e == hipSuccess && a.n_steps) e = hipMemsetAsync(a.n_steps, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e == hipSuccess && a.n_reject) e = hipMemsetAsync(a.n_reject, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e != hipSuccess) return (int)e;
      }
      if constexpr (SbmIexSeqFits<M>::value) {
        // chain models, order <= 8: sequences side by side + persistent wavefronts (sbm_implicit_extrap_seq.hpp)
        int K = a.opts.step_mult;
        const double rtol = a.opts.rtol > 0.0 ? a.opts.rtol : 1e-8;
        if (K <= 0) K = rtol >= 1e-4 ? 4 : (rtol >= 1e-6 ? 6 : 8);
        if (K <= SbmIexSeqPlan<M>::KMAX && sbm_iex_seq_enabled()) {
          const int n_work = a.n_traj * nch;
          int resident = 0;
          // columns held rotated (chain + one J_p entry per column: no select in the column step) unless the caller hands
          // in initial sensitivities, which need not respect the structure
          constexpr bool kRot = SbmIexSeqPlan<M>::ROT_OK;
          const bool rot = kRot && a.s0 == nullptr;
          const void* kfn = rot ? (const void*)sbm_iex_seq_kernel<M, kRot> : (const void*)sbm_iex_seq_kernel<M, false>;
          hipError_t e = sbm_resident_blocks(kfn, 64, &resident);
          if (e != hipSuccess) return (int)e;
          const int grid = n_work < resident ? n_work : resident;
          if (getenv("SBM_DEBUG_LAUNCH")) fprintf(stderr, "sbm_iex_seq_kernel: %d pieces of work, %d resident workgroups, grid %d\n", n_work, resident, grid);
          SbmScratch* sc = nullptr;
          e = sbm_scratch_for(stream, (size_t)grid * SbmIexSeqPlan<M>::BLOCK_DOUBLES * sizeof(double), &sc);
          if (e != hipSuccess) return (int)e;
          e = hipMemsetAsync(sc->counter, 0, sizeof(int), stream);
          if (e != hipSuccess) return (int)e;
          if (rot) hipLaunchKernelGGL((sbm_iex_seq_kernel<M, kRot>), dim3(grid), dim3(64), 0, stream, a, (double*)sc->p, sc->counter, n_work, nch);
          else hipLaunchKernelGGL((sbm_iex_seq_kernel<M, false>), dim3(grid), dim3(64), 0, stream, a, (double*)sc->p, sc->counter, n_work, nch);
          return (int)hipGetLastError();
        }
      }
      hipLaunchKernelGGL((sbm_iex_kernel<M>), dim3(a.n_traj, nch), dim3(64), 0, stream, a);
      return (int)hipGetLastError();
    } else {
      return (int)hipErrorInvalidConfiguration;
    }
  }
  if (a.opts.method == SBM_IMPLICIT_ADAPTIVE) {
    if constexpr (SbmImplicitFits<M>::adaptive) {
      const int nch = a.S ? (M::NK + 63) / 64 : 1;
      if (nch > 1) {     // chunks combine status / counts with atomicMax
        hipError_t e = hipSuccess;
        if (a.status) e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e == hipSuccess && a.n_steps) e = hipMemsetAsync(a.n_steps, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e == hipSuccess && a.n_reject) e = hipMemsetAsync(a.n_reject, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e != hipSuccess) return (int)e;
      }
      hipLaunchKernelGGL((sbm_imid_adaptive_kernel<M>), dim3(a.n_traj, nch), dim3(64), 0, stream, a);
      return (int)hipGetLastError();
    } else {
      return (int)hipErrorInvalidConfiguration;
    }
  }
  if (a.opts.method == SBM_IMPLICIT_MIDPOINT || a.opts.method == SBM_IMPLICIT_MIDPOINT_GRADED) {
    // one trajectory per wave for both kinds (state only: S == NULL skips the column work)
    if constexpr (SbmImplicitFits<M>::fixed) {
      // state only: one wavefront; with sensitivities: one per chunk of 64 columns
      const int nch = a.S ? (M::NK + 63) / 64 : 1;
      hipLaunchKernelGGL((sbm_imid_kernel<M>), dim3(a.n_traj, nch), dim3(64), 0, stream, a);
      return (int)hipGetLastError();
    } else {
      return (int)hipErrorInvalidConfiguration;   // see SbmImplicitFits
    }
  }
  // J_y S on the matrix cores (sbm_sens_mfma.hpp) costs the same whatever the sparsity of J_y; the scalar kernels cost
  // 2 FMAs per non-zero and column.  Measured on 20-state networks, 4096 vectors, DOPRI45 (bench.py "dense",
  // profiles/r03, scalar / MFMA ms): 40 non-zeros (cascade20, twice the steps) 5.3 / 21.7; 60: 4.0 / 11.1; 120: 9.3 / 11.5;
  // 220: 25.7 / 12.1; 400 (dense): 67.1 / 13.4 -- the matrix cores win from about 35 % density (round 2, one wavefront per
  // SIMD: 45 %).  AUTO takes them from there (a static property of the model: a given model always runs the same
  // kernel); SBM_VARIANT_MFMA forces them.
  constexpr bool kMfmaPays = M::NV >= 16 && M::NV <= 64 && (long long)M::NNZ_JY * 100 >= 35LL * M::NV * M::NV;
  // (DOP853 keeps twelve stage vectors alive -- on the matrix-core kernel they leave the register file: built, so that a
  // forced variant answers, but AUTO keeps DOP853 on the row kernels)
  if (kind == SBM_KIND_SENS && (a.opts.variant == SBM_VARIANT_MFMA ||
                                (kMfmaPays && a.opts.method != SBM_DOP853 &&
                                 (a.opts.variant == SBM_VARIANT_AUTO || a.opts.variant == SBM_VARIANT_SMALL_BATCH)))) {
    // models beyond one state row per lane fall through to the scalar kernels
    if constexpr (M::NV <= 64) {
      constexpr int nch = SbmMfmaPlan<M>::NCH;
      if (nch > 1) {
        hipError_t e = hipSuccess;
        if (a.status) e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e == hipSuccess && a.n_steps) e = hipMemsetAsync(a.n_steps, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e == hipSuccess && a.n_reject) e = hipMemsetAsync(a.n_reject, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
        if (e != hipSuccess) return (int)e;
      }
      dim3 grid(a.n_traj, nch), block(64);
      if (a.opts.method == SBM_DOPRI45) hipLaunchKernelGGL((sbm_sens_mfma_kernel<M, SBM_DOPRI45>), grid, block, 0, stream, a);
      else if (a.opts.method == SBM_DOP853) hipLaunchKernelGGL((sbm_sens_mfma_kernel<M, SBM_DOP853>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((sbm_sens_mfma_kernel<M, SBM_RK4_FIXED>), grid, block, 0, stream, a);
      return (int)hipGetLastError();
    }
  }
  // small models: several trajectories per wavefront (sbm_sens_packed_kernel)
  if constexpr (M::NV <= 32 && M::NK <= 32 && M::NV * (M::NK + 1) <= 256) {
    constexpr int need = M::NV > M::NK ? M::NV : M::NK;
    constexpr int SEG = need <= 4 ? 4 : (need <= 8 ? 8 : (need <= 16 ? 16 : 32));
    // AUTO: small models ALWAYS run packed (round 2 switched at 2048 trajectories: a vector's result then depended on
    // the size of the batch it travelled in -- the shard a rank owns, the subset a lazy-Jacobian fit re-integrates).  One
    // trajectory alone in its wavefront costs what it costs in the unpacked kernels; the single-vector methods of the
    // Python classes ask for SMALL_BATCH and keep the row kernels' lower latency.
    if (kind == SBM_KIND_SENS && (a.opts.variant == SBM_VARIANT_PACKED || a.opts.variant == SBM_VARIANT_AUTO)) {
      dim3 grid((a.n_traj + 64 / SEG - 1) / (64 / SEG)), block(64);
      if (a.opts.method == SBM_DOPRI45) hipLaunchKernelGGL((sbm_sens_packed_kernel<M, SBM_DOPRI45, SEG>), grid, block, 0, stream, a);
      else if (a.opts.method == SBM_DOP853) hipLaunchKernelGGL((sbm_sens_packed_kernel<M, SBM_DOP853, SEG>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((sbm_sens_packed_kernel<M, SBM_RK4_FIXED, SEG>), grid, block, 0, stream, a);
      return (int)hipGetLastError();
    }
  }
  if (kind == SBM_KIND_SENS) {
    // row-lane / row-group kernels whenever the model fits one row + one column per lane.  Even when
    // every row is a class of its own they evaluate NCLASS <= NV row bodies per stage where the per-wave
    // kernel evaluates all NV rows on every lane (measured on random networks with 6 and 9 classes of 11
    // and 17 rows: 1.6x faster than per-wave)
    constexpr bool kRowLaneOk = (M::NV <= 64 && M::NK <= 64);
    constexpr bool kRowLanePays = kRowLaneOk;
    constexpr bool kRowGroupOk = M::RG0::RG_OK;   // any number of columns (chunks of them), up to four rows per lane
    // The per-wave kernel keeps all NV rows of ceil((NK+1)/64) columns on every lane: for a large model that is
    // minutes of compile time for a kernel whose stage vectors live in scratch.  Where the row-group form exists
    // it is not instantiated beyond 4096 sensitivity entries, and opts.variant becomes a no-op for that model.
    constexpr bool kPerWaveBuilt = !(kRowGroupOk && M::NV * (M::NK + 1) > 4096);
    const bool rowlane = a.opts.variant == SBM_VARIANT_ROW_LANE ||
                         ((a.opts.variant == SBM_VARIANT_AUTO || a.opts.variant == SBM_VARIANT_SMALL_BATCH ||
                           a.opts.variant == SBM_VARIANT_MFMA || a.opts.variant == SBM_VARIANT_PACKED) && kRowLanePays);
    // row-group kernel: the row-lane kernel with the rows of a column split over several lanes,
    // when the emitter found a split that cuts the elements per lane (M::RG_OK)
    if constexpr (kRowGroupOk) {
      if (a.opts.variant == SBM_VARIANT_ROW_GROUP || a.opts.variant == SBM_VARIANT_AUTO ||
          a.opts.variant == SBM_VARIANT_SMALL_BATCH || a.opts.variant == SBM_VARIANT_MFMA ||
          a.opts.variant == SBM_VARIANT_PACKED || !kPerWaveBuilt) {
        // Two splits of the same form (emit_rowgroup.py): RG0 for throughput; RG1 -- more, smaller column chunks,
        // fewer elements per lane -- while all its wavefronts are resident at once (2048: two per SIMD; a wavefront of
        // this split issues ~40 % of the other's instructions per step, so two of them sharing a SIMD still finish a step
        // sooner than one of the other alone): a single parameter vector, a serial optimiser's call, is latency-bound.
        // Opt-in (SBM_VARIANT_SMALL_BATCH: what the single-vector methods of the Python classes ask for): the two
        // splits take different step sequences, and a batch call's rows must not depend on how many rows it has.
        const bool small_batch = !std::is_same<typename M::RG1, typename M::RG0>::value &&
                                 a.opts.variant == SBM_VARIANT_SMALL_BATCH && (long long)a.n_traj * M::RG1::RG_NCH <= 2048;
        auto go = [&](auto layout_tag) -> int {
          using L = typename decltype(layout_tag)::type;
          dim3 grid(a.n_traj, L::RG_NCH), block(64);
          if constexpr (L::RG_NCH > 1) {
            hipError_t e = hipSuccess;
            if (a.status) e = hipMemsetAsync(a.status, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
            if (e == hipSuccess && a.n_steps) e = hipMemsetAsync(a.n_steps, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
            if (e == hipSuccess && a.n_reject) e = hipMemsetAsync(a.n_reject, 0, sizeof(int32_t) * (size_t)a.n_traj, stream);
            if (e != hipSuccess) return (int)e;
          }
          if (a.opts.method == SBM_DOPRI45)
            hipLaunchKernelGGL((sbm_sens_rowgroup_kernel<M, L, SBM_DOPRI45>), grid, block, 0, stream, a);
          else if (a.opts.method == SBM_DOP853) {
            // (instantiated for its own split and the small-batch one only)
            if constexpr (std::is_same<L, typename M::RG2>::value || std::is_same<L, typename M::RG1>::value)
              hipLaunchKernelGGL((sbm_sens_rowgroup_kernel<M, L, SBM_DOP853>), grid, block, 0, stream, a);
            else
              return (int)hipErrorInvalidConfiguration;
          } else
            hipLaunchKernelGGL((sbm_sens_rowgroup_kernel<M, L, SBM_RK4_FIXED>), grid, block, 0, stream, a);
          return (int)hipGetLastError();
        };
        // DOP853 keeps twelve stage vectors alive: its own split, planned for smaller shares per lane (RG2)
        if (a.opts.method == SBM_DOP853 && !small_batch) return go(SbmTypeTag<typename M::RG2>{});
        return small_batch ? go(SbmTypeTag<typename M::RG1>{}) : go(SbmTypeTag<typename M::RG0>{});
      }
    }
    if (a.opts.method == SBM_DOP853) {
      // beside the row-group form: the row-lane kernel (twelve stage vectors of NV rows per lane: beyond ~16 state
      // variables they leave the register file and the kernel runs out of scratch -- correct, slow), then the per-wave one
      if constexpr (kRowLaneOk && M::NV <= 32) {
        if (a.opts.variant != SBM_VARIANT_PER_WAVE) {
          hipLaunchKernelGGL((sbm_sens_rowlane_kernel<M, SBM_DOP853>), dim3(a.n_traj), dim3(64), 0, stream, a);
          return (int)hipGetLastError();
        }
      }
      if constexpr (kPerWaveBuilt && M::NV * ((M::NK + 64) / 64) <= 64) {
        hipLaunchKernelGGL((sbm_sens_kernel<M, SBM_DOP853>), dim3(a.n_traj), dim3(64), 0, stream, a);
        return (int)hipGetLastError();
      } else {
        return (int)hipErrorInvalidConfiguration;
      }
    }
    if constexpr (kRowLaneOk) {
      if (rowlane) {
        dim3 grid(a.n_traj), block(64);
        if (a.opts.method == SBM_DOPRI45)
          hipLaunchKernelGGL((sbm_sens_rowlane_kernel<M, SBM_DOPRI45>), grid, block, 0, stream, a);
        else
          hipLaunchKernelGGL((sbm_sens_rowlane_kernel<M, SBM_RK4_FIXED>), grid, block, 0, stream, a);
        return (int)hipGetLastError();
      }
    }
    if constexpr (kPerWaveBuilt) {
      dim3 grid(a.n_traj), block(64);
      if (a.opts.method == SBM_DOPRI45) hipLaunchKernelGGL((sbm_sens_kernel<M, SBM_DOPRI45>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((sbm_sens_kernel<M, SBM_RK4_FIXED>), grid, block, 0, stream, a);
    }
  } else {
    // one trajectory per wave until the chip is full of lane-per-trajectory waves anyway
    constexpr bool kRowsOk = (M::NV <= 256);   // up to four state rows per lane
    // one trajectory per LANE keeps NV stage-vector rows per lane: beyond 64 rows only the rows kernel is built
    constexpr bool kLaneBuilt = !(kRowsOk && M::NV > 64);
    // Which state-only kernel (measured on cascade20, DOPRI45, scripts/dev_state_big.py): one trajectory per
    // wavefront up to 2047 trajectories (0.2 ms per 1024, lowest latency); several per wavefront from there
    // (0.14 ms per 1024: 0.56 ms at 4096, 3.3 ms at 32768); one trajectory per LANE costs 2.6 - 2.9 ms whatever
    // the batch up to 65536 (one serial chain per lane, 64 of them per wavefront) and wins from ~20000 on --
    // for small models only: beyond 32 state variables its stage vectors leave the register file.
    constexpr int kLaneFrom = (M::NV <= 32) ? 20480 : 65536;
    // several trajectories per wavefront once the chip is full: models of up to 32 state variables
    if constexpr (M::NV <= 32 && M::NV >= 2) {
      if (a.n_traj >= 2048 && a.n_traj < kLaneFrom && a.opts.variant == SBM_VARIANT_AUTO) {
        // DOPRI45 reduces its error norm inside a segment: widths 16 / 32 (DPP).  Segment sums through LDS for
        // other widths were tried (20 lanes: three trajectories per wavefront): 0.53 against 0.49 ms -- the LDS
        // round trip per step and a third trajectory to wait for cost more than the denser packing gains.
        // RK4 reduces nothing: the width is the state variables rounded up to a multiple of four lanes.
        constexpr int SEG_A = M::NV <= 16 ? 16 : 32;
        constexpr int SEG_F = (M::NV + 3) / 4 * 4;
        if (a.opts.method == SBM_DOPRI45) {
          dim3 grid((a.n_traj + 64 / SEG_A - 1) / (64 / SEG_A)), block(64);
          hipLaunchKernelGGL((sbm_state_packed_kernel<M, SBM_DOPRI45, SEG_A>), grid, block, 0, stream, a);
        } else if (a.opts.method == SBM_DOP853) {
          dim3 grid((a.n_traj + 64 / SEG_A - 1) / (64 / SEG_A)), block(64);
          hipLaunchKernelGGL((sbm_state_packed_kernel<M, SBM_DOP853, SEG_A>), grid, block, 0, stream, a);
        } else {
          dim3 grid((a.n_traj + 64 / SEG_F - 1) / (64 / SEG_F)), block(64);
          hipLaunchKernelGGL((sbm_state_packed_kernel<M, SBM_RK4_FIXED, SEG_F>), grid, block, 0, stream, a);
        }
        return (int)hipGetLastError();
      }
    }
    if constexpr (kRowsOk) {
      if (((a.n_traj < kLaneFrom || a.opts.variant == SBM_VARIANT_ROW_LANE || a.opts.variant == SBM_VARIANT_ROW_GROUP) &&
           a.opts.variant != SBM_VARIANT_PER_WAVE) || !kLaneBuilt || a.opts.method == SBM_DOP853) {
        dim3 grid(a.n_traj), block(64);
        if (a.opts.method == SBM_DOPRI45)
          hipLaunchKernelGGL((sbm_state_rows_kernel<M, SBM_DOPRI45>), grid, block, 0, stream, a);
        else if (a.opts.method == SBM_DOP853)
          hipLaunchKernelGGL((sbm_state_rows_kernel<M, SBM_DOP853>), grid, block, 0, stream, a);
        else
          hipLaunchKernelGGL((sbm_state_rows_kernel<M, SBM_RK4_FIXED>), grid, block, 0, stream, a);
        return (int)hipGetLastError();
      }
    }
    if (a.opts.method == SBM_DOP853) return (int)hipErrorInvalidConfiguration;   // (more than 256 state variables)
    if constexpr (kLaneBuilt) {
      dim3 grid((a.n_traj + 63) / 64), block(64);
      if (a.opts.method == SBM_DOPRI45) hipLaunchKernelGGL((sbm_state_kernel<M, SBM_DOPRI45>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((sbm_state_kernel<M, SBM_RK4_FIXED>), grid, block, 0, stream, a);
    }
  }
  return (int)hipGetLastError();
}
