// sbm_implicit_stepper.hpp -- one implicit-midpoint step on a wavefront: the Newton iteration on the midpoint state and
// the linear solve that advances a sensitivity column.  Shared by the fixed-step kernel (sbm_imid_kernel) and the
// error-controlled one (sbm_imid_adaptive_kernel).
//
//   y_{n+1} = y_n + h f(ybar),  ybar = (y_n + y_{n+1}) / 2      Newton on ybar:
//       M(ybar) delta = ybar - y_n - (h/2) f(ybar),  M = I - (h/2) J_y(ybar)
//   S_{n+1} = 2 Sbar - S_n,     M Sbar = S_n + (h/2) J_p(ybar)   -- the EXACT derivative of the scheme:
//       one linear solve per sensitivity column with the matrix Newton just factored.
//
// Mapping: lane j owns column j (of its chunk of 64) of S, all NV rows in VGPRs.  State component i lives on lane
// i mod 64: a model with more than 64 state variables gives every lane RPL = ceil(NV / 64) of them (rows lane,
// lane + 64, ...).  Row lanes evaluate f_i and the J_y / J_p entries of their rows by class (emit_rowlane.py) and
// publish them in LDS.  M is the same for every lane of the wave: each lane applies the sparse LU the model generator
// worked out symbolically for the model's sparsity pattern (emit_implicit.py: straight-line code, static indices,
// fill-in included; for lower-triangular patterns the factors are one reciprocal per ROW, computed by the row's lane
// and read from the table MF) to its own right-hand side: the Newton residual (picked apart again: the lane of row i
// keeps delta_i) and its sensitivity column.
//
// Three forms of the factorisation, chosen by the model's pattern (emit_implicit.py):
//   IM_TRI   lower triangular: no elimination; row lane i publishes 1 / M_ii and its scaled row in MF;
//   IM_DIST  anything else on up to 64 state variables: lane i holds row i of M as a dense register row, the lanes
//            eliminate together (pivot rows travel by v_readlane), the finished rows are published in MF
//            ([row][IM_LD]) with the reciprocal pivots in RD; the substitutions read both with wave-uniform addresses;
//   neither  (more than 64 state variables, not triangular): every lane factors the whole matrix in its own registers.
//
// The shared-memory struct Sh provides: Y[NROW], G[NROW] (NROW = 64 RPL), JY[NJY + 2], MF[MF_SIZE], RD[RD_SIZE],
// A[A_SIZE] and the constants A_SPARSE (J_p stored [row][slot] instead of [row][64 columns]), A_SIZE (last slot =
// spare), MF_SIZE = sbm_imf_size<M>() (last slot = spare) and RD_SIZE = sbm_ird_size<M>().
#pragma once

#include <type_traits>
#include <utility>

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant
template <class F, int... I>
__device__ __forceinline__ void sbm_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sbm_static_for(F&& f) {
  sbm_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// v[LO + lane] for lane < CNT out of a register array: the select tree of sbm_pick_tree over a slice
template <int N, int LO, int CNT>
__device__ __forceinline__ double sbm_pick_slice(const double (&v)[N], int lane) {
  static_assert(LO >= 0 && CNT >= 1 && LO + CNT <= N, "slice");
  if constexpr (LO == 0 && CNT == N) {
    return sbm_pick_tree<N>(v, lane);
  } else {
    double t[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) t[j] = v[LO + j];
    return sbm_pick_tree<CNT>(t, lane);
  }
}

// sizes of the factor tables in LDS
template <class M>
constexpr int sbm_imf_size() { return (M::IM_DIST ? M::NV * M::IM_LD : M::IM_MF) + 2; }
template <class M>
constexpr int sbm_ird_size() { return M::IM_DIST ? M::NV : 1; }
// J_y by entry index: read by im_build only (neither triangular nor row-distributed) -- a dense model's 1200 entries are
// 9.6 KB of LDS nobody reads otherwise
template <class M>
constexpr int sbm_ijy_size() {
#ifdef SBM_IMPLICIT_REDUNDANT_LU
  return M::NJY + 2;
#else
  return (M::IM_TRI || M::IM_DIST) ? 2 : M::NJY + 2;
#endif
}

template <class M, class Sh>
struct SbmImplicitStepper {
  static constexpr int NV = M::NV;
  static constexpr int RPL = (NV + 63) / 64;       // state rows per lane
  static constexpr int NROW = 64 * RPL;
  static constexpr int ASPARE = Sh::A_SIZE - 1;
  Sh* sh;
  int lane, chunk;
  bool has_row[RPL];
  int cls[RPL];
  int yidx[RPL][M::RL_MAXYS], jyout[RPL][M::RL_MAXJY], apos[RPL][M::RL_MAXJP], mfpos[RPL][M::RL_MAXJY];
  int rdpos[RPL], diagslot[RPL];
  double ps[RPL][M::RL_MAXPS];
#ifdef SBM_IMPLICIT_REDUNDANT_LU                   // developer A/B switch: the redundant form where the distributed one applies
  static constexpr bool DIST = false;
#else
  static constexpr bool DIST = M::IM_DIST;         // (RPL == 1 there)
#endif
  static constexpr int MFSPARE = Sh::MF_SIZE - 1;
  double m[(DIST || M::IM_TRI) ? 1 : M::IM_NM];    // the redundant form's factors
  static constexpr bool CHAIN = M::IM_CHAIN && RPL == 1 && !DIST;
  double ch_a, ch_b;                               // CHAIN: this lane's row of the recurrence x_i = b_i + a_i x_{i-1}

  __device__ __forceinline__ static void fence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

  // Clears the LDS tables and loads this lane's rows (operand indices, parameters, output positions).  `chunk` = which
  // 64 columns of S this wavefront advances.  The caller synchronises (one wavefront per block: a fence suffices).
  __device__ __forceinline__ void setup(Sh* shared, int lane_, int chunk_, const double* P) {
    sh = shared;
    lane = lane_;
    chunk = chunk_;
    constexpr int NCH = (M::NK + 63) / 64;
    for (int i = lane; i < Sh::A_SIZE; i += 64) sh->A[i] = 0.0;
    for (int i = lane; i < sbm_ijy_size<M>(); i += 64) sh->JY[i] = 0.0;
    for (int i = lane; i < Sh::MF_SIZE; i += 64) sh->MF[i] = 0.0;
    for (int i = lane; i < Sh::RD_SIZE; i += 64) sh->RD[i] = 1.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      sh->Y[lane + 64 * r] = 0.0;
      sh->G[lane + 64 * r] = 0.0;
      const bool hr = lane + 64 * r < NV;
      const int row = hr ? lane + 64 * r : 0;
      has_row[r] = hr;
      cls[r] = hr ? M::rl_class(row) : -1;
#pragma unroll
      for (int s = 0; s < M::RL_MAXYS; ++s) yidx[r][s] = M::rl_ys(s, row);
#pragma unroll
      for (int s = 0; s < M::RL_MAXPS; ++s) ps[r][s] = P[M::rl_ps(s, row)];
#pragma unroll
      for (int s = 0; s < M::RL_MAXJY; ++s) jyout[r][s] = hr ? M::rl_jyout(s, row) : M::NJY + 1;
#pragma unroll
      for (int s = 0; s < M::RL_MAXJP; ++s) {
        if constexpr (Sh::A_SPARSE) {
          apos[r][s] = hr ? row * M::RL_MAXJP + s : ASPARE;
        } else if constexpr (NCH == 1) {
          apos[r][s] = hr ? M::rl_apos(s, row) : ASPARE;
        } else {
          const int lc = M::rl_jpcol(s, row) - 64 * chunk;       // column within this chunk (unused slots: -1)
          apos[r][s] = (hr && lc >= 0 && lc < 64) ? row * 64 + lc : ASPARE;
        }
      }
      if constexpr (DIST) {
        // where slot s of this lane's row goes in the dense table, and which slot (if any) is the diagonal entry
        diagslot[r] = -1;
#pragma unroll
        for (int s = 0; s < M::RL_MAXJY; ++s) {
          const int c = M::rl_jycol(s, row);
          mfpos[r][s] = (hr && c >= 0) ? row * M::IM_LD + c : MFSPARE;
          if (hr && c == row) diagslot[r] = s;
        }
        rdpos[r] = hr ? row * M::IM_LD : MFSPARE;       // start of this lane's row
      } else {
#pragma unroll
        for (int s = 0; s < M::RL_MAXJY; ++s) mfpos[r][s] = (M::IM_TRI && hr) ? M::im_mfpos(s, row) : MFSPARE;
        rdpos[r] = (M::IM_TRI && hr) ? M::im_rstart(row) : MFSPARE;
        diagslot[r] = (M::IM_TRI && hr) ? M::im_diagslot(row) : -1;
      }
    }
#pragma unroll
    for (int e = 0; e < (int)(sizeof(m) / sizeof(double)); ++e) m[e] = 0.0;
  }

  // IM_DIST: build this lane's row of M = I - hh J_y in the LDS table, read it back as a dense register row, factor
  // with the other lanes, publish.  jy: the class outputs of this lane's row.
  __device__ __forceinline__ void factor_rows(double hh, const double (&jy)[M::RL_MAXJY]) {
    static_assert(!DIST || RPL == 1, "distributed factorisation: one row per lane");
    constexpr int LD = M::IM_LD;
    double* myrow = sh->MF + (has_row[0] ? rdpos[0] : 0);
    if (has_row[0]) {
      // the table still holds the previous factors: clear the row (16-byte stores), then scatter the entries of M
#pragma unroll
      for (int j = 0; j + 1 < LD; j += 2) *reinterpret_cast<double2*>(myrow + j) = double2{0.0, 0.0};
      fence();
      myrow[lane] = 1.0;
    }
    fence();
#pragma unroll
    for (int q = 0; q < M::RL_MAXJY; ++q) {
      const double v = -hh * jy[q];
      sh->MF[mfpos[0][q]] = diagslot[0] == q ? 1.0 + v : v;
    }
    fence();
    double row[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) row[j] = myrow[j];
    fence();
    M::im_factor_rows(row, lane, sh->RD);
    if (has_row[0]) {
#pragma unroll
      for (int j = 0; j < NV; ++j) myrow[j] = row[j];
    }
    fence();
  }

  // Newton on the midpoint state of one step of size 2*hh from y: on entry yb = predictor, on exit the midpoint.
  // Leaves the factors of M = I - hh J_y (m / sh->MF) and J_p (sh->A) of the last iterate for sens(): within the
  // Newton tolerance of the converged midpoint.  Returns SBM_OK, SBM_NEWTON_FAIL or SBM_NON_FINITE (wave-uniform).
  template <int MAXIT>
  __device__ __forceinline__ int newton(double tm, double hh, const double (&y)[RPL], double (&yb)[RPL], double nrtol,
                                        double natol, int& n_iter) {
    // (since round 3 on the pieces below: a chain Jacobian's update is a parallel prefix over the row lanes, a triangular
    // one's the fused substitution -- 36 / ~350 instructions where every lane used to repeat the whole substitution and
    // pick its component out of NV)
    for (int it = 0; it < MAXIT; ++it) {
      ++n_iter;
      eval_factor(tm, hh, y, yb);
      double d[RPL];
      solve_delta(d);
      float rmax = 0.f;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double dd = has_row[r] ? d[r] : 0.0;
        yb[r] -= dd;
        rmax = fmaxf(rmax, has_row[r] ? sbm_nan_to_inf((float)(fabs(dd) / fma(nrtol, fabs(yb[r]), natol))) : 0.f);
      }
      const float rr = sbm_wave_max(rmax);
      if (!(rr < 3.0e38f)) return SBM_NON_FINITE;
      if (rr <= 1.0f) return SBM_OK;
    }
    return SBM_NEWTON_FAIL;
  }

  // ---- the same iteration in pieces (sbm_implicit_extrap.hpp) ----
  // Evaluate f, J_y, J_p at yb (time tm): the Newton residual (yb - y) - hh f goes to G, J_p to A, and M = I - hh J_y is
  // factored (MF / RD / m), ready for solve_delta() and the sensitivity solves.
  __device__ __forceinline__ void eval_factor(double tm, double hh, const double (&y)[RPL], const double (&yb)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) sh->Y[lane + 64 * r] = yb[r];
    fence();
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      double ys[M::RL_MAXYS];
#pragma unroll
      for (int q = 0; q < M::RL_MAXYS; ++q) ys[q] = sh->Y[yidx[r][q]];
      double f = 0.0, jy[M::RL_MAXJY], jp[M::RL_MAXJP];
#pragma unroll
      for (int q = 0; q < M::RL_MAXJY; ++q) jy[q] = 0.0;
#pragma unroll
      for (int q = 0; q < M::RL_MAXJP; ++q) jp[q] = 0.0;
      M::class_dispatch(cls[r], tm, ys, ps[r], f, jy, jp);
      fence();
#pragma unroll
      for (int q = 0; q < M::RL_MAXJP; ++q) sh->A[apos[r][q]] = jp[q];
      if constexpr (!M::IM_TRI && !DIST) {
#pragma unroll
        for (int q = 0; q < M::RL_MAXJY; ++q) sh->JY[jyout[r][q]] = jy[q];
      }
      sh->G[lane + 64 * r] = has_row[r] ? (yb[r] - y[r]) - hh * f : 0.0;
      if constexpr (M::IM_TRI) {
        double jd = 0.0;
#pragma unroll
        for (int q = 0; q < M::RL_MAXJY; ++q) jd = sbm_sel(diagslot[r] == q, jy[q], jd);
        const double rd = sbm_rcp(fma(-hh, jd, 1.0));
        sh->MF[rdpos[r]] = rd;
#pragma unroll
        for (int q = 0; q < M::RL_MAXJY; ++q) sh->MF[mfpos[r][q]] = hh * jy[q] * rd;
        if constexpr (CHAIN) {
          double off = 0.0;       // the one entry left of the diagonal (none in the first row)
#pragma unroll
          for (int q = 0; q < M::RL_MAXJY; ++q) off += sbm_sel(diagslot[r] == q, 0.0, jy[q]);
          ch_a = has_row[r] ? hh * off * rd : 0.0;
          ch_b = has_row[r] ? ((yb[r] - y[r]) - hh * f) * rd : 0.0;
        }
      } else if constexpr (DIST) {
        factor_rows(hh, jy);
      }
    }
    fence();
    if constexpr (!M::IM_TRI && !DIST) {
      M::im_build(hh, sh->JY, m);
      M::im_factor(m);
    }
  }

  // v of the lane `CTRL` names (DPP), `fill` where that lane does not exist or the row is masked off
  template <int CTRL, int ROW_MASK>
  __device__ __forceinline__ static double dpp_f64(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  // one level of the prefix: (a, b)_i <- (a, b)_i o (a, b)_src, i.e. x_i = b_i + a_i x_src expressed through x_(src's source)
  template <int CTRL, int ROW_MASK>
  __device__ __forceinline__ static void chain_level(double& a, double& b) {
    double ap, bp;
    if constexpr (ROW_MASK == 0xf) {
      // every row takes part: a lane without a source reads zeros through bound_ctrl (no fill registers to set up);
      // only the high word of the identity 1.0 needs one
      const int z = 0;
      bp = __hiloint2double(__builtin_amdgcn_update_dpp(z, __double2hiint(b), CTRL, ROW_MASK, 0xf, true),
                            __builtin_amdgcn_update_dpp(z, __double2loint(b), CTRL, ROW_MASK, 0xf, true));
      ap = __hiloint2double(__builtin_amdgcn_update_dpp(0x3ff00000, __double2hiint(a), CTRL, ROW_MASK, 0xf, false),
                            __builtin_amdgcn_update_dpp(z, __double2loint(a), CTRL, ROW_MASK, 0xf, true));
    } else {
      ap = dpp_f64<CTRL, ROW_MASK>(a, 1.0);
      bp = dpp_f64<CTRL, ROW_MASK>(b, 0.0);
    }
    b = fma(a, bp, b);
    a *= ap;
  }

  // d <- the components of this lane's rows of M^-1 G (the Newton update), with the factors eval_factor() left
  __device__ __forceinline__ void solve_delta(double (&d)[RPL]) {
    if constexpr (CHAIN) {
      // x_i = b_i + a_i x_{i-1}, x_{-1} = 0: an inclusive prefix over the row lanes with the associative composition
      // of affine maps, on DPP -- shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then the last lane of a row to the
      // row above (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3), as rocPRIM scans a wavefront.
      // 36 instructions where the substitution every lane repeats for itself takes ~350 (50 rows).
      double a = ch_a, b = ch_b;
      chain_level<0x111, 0xf>(a, b);    // row_shr:1
      chain_level<0x112, 0xf>(a, b);    // row_shr:2
      chain_level<0x114, 0xf>(a, b);    // row_shr:4
      chain_level<0x118, 0xf>(a, b);    // row_shr:8
      if constexpr (NV > 16) chain_level<0x142, 0xa>(a, b);    // row_bcast:15
      if constexpr (NV > 32) chain_level<0x143, 0xc>(a, b);    // row_bcast:31
      d[0] = b;
    } else if constexpr (M::IM_TRI && !DIST) {
      // fused: a solution component lives in a register only while later rows refer to it (emit_implicit.py)
#pragma unroll
      for (int r = 0; r < RPL; ++r) d[r] = 0.0;
      int lo = lane;               // (opaque: the 2 NV lane masks are otherwise hoisted out of every loop and spilled)
      asm volatile("" : "+v"(lo));
      M::template im_solve_tri_pick<RPL>(sh->MF, sh->G, lo, d);
      fence();
    } else {
      double b[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) b[i] = sh->G[i];
      if constexpr (DIST) M::im_solve_lds(sh->MF, sh->RD, b);
      else M::im_solve(m, b);
      fence();
      sbm_static_for<RPL>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        constexpr int CNT = (NV - 64 * r) < 64 ? (NV - 64 * r) : 64;
        d[r] = sbm_pick_slice<NV, 64 * r, CNT>(b, lane);
      });
    }
  }

  // Newton on yb as newton(), ended by the RATE of convergence: the updates fall quadratically, so once two of them
  // are known the error of the iterate just formed is predicted as rr^2 * (rr / rr_prev^2) (in units of the
  // tolerance); below 0.1 the iteration stops one solve early -- f, J_y, J_p are evaluated and M is factored at that
  // iterate (the sensitivity solves want the matrices of the CONVERGED state: they are amplified by the extrapolation
  // weights), the solve for an update that would change nothing is skipped.
  template <int MAXIT>
  __device__ __forceinline__ int newton_rate(double tm, double hh, const double (&y)[RPL], double (&yb)[RPL], double nrtol,
                                             double natol, int& n_iter) {
    float r_prev = 0.f;
    for (int it = 0; it < MAXIT; ++it) {
      ++n_iter;
      eval_factor(tm, hh, y, yb);
      double d[RPL];
      solve_delta(d);
      float rmax = 0.f;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double dd = has_row[r] ? d[r] : 0.0;
        yb[r] -= dd;
        rmax = fmaxf(rmax, has_row[r] ? sbm_nan_to_inf((float)fabs(dd) * __builtin_amdgcn_rcpf((float)fmax(fma(nrtol, fabs(yb[r]), natol), 1e-30))) : 0.f);
      }
      const float rr = sbm_wave_max(rmax);
      if (!(rr < 3.0e38f)) return SBM_NON_FINITE;
      if (rr <= 1.0f) return SBM_OK;
      if (it > 0 && rr < 0.25f * r_prev && rr * rr * (rr / (r_prev * r_prev)) <= 0.1f) {
        ++n_iter;
        eval_factor(tm, hh, y, yb);
        return SBM_OK;
      }
      // the tolerance sits a few hundred ulps above rounding: an iteration that has stopped falling within 10^3
      // tolerances of it has converged as far as the arithmetic allows
      if (it >= 2 && rr >= 0.5f * r_prev && rr <= 1.0e3f) {
        ++n_iter;
        eval_factor(tm, hh, y, yb);
        return SBM_OK;
      }
      r_prev = rr;
    }
    return SBM_NEWTON_FAIL;
  }

  // J_p[i][this lane's column]
  __device__ __forceinline__ double a_of(int i) const {
    if constexpr (Sh::A_SPARSE) {
      double a = 0.0;
#pragma unroll
      for (int q = 0; q < M::RL_MAXJP; ++q)
        a = sbm_sel(M::rl_jpcol(q, i) - 64 * chunk == lane, sh->A[i * M::RL_MAXJP + q], a);
      return a;
    } else {
      return sh->A[i * 64 + lane];
    }
  }

  // one midpoint step of a sensitivity column with the matrices newton() left: z <- 2 M^-1 (z + hh J_p) - z
  __device__ __forceinline__ void sens(double hh, double (&z)[NV]) {
    double b[NV];
    if constexpr (Sh::A_SPARSE) {
      int lo = lane + 64 * chunk;  // (opaque: see solve_delta)
      asm volatile("" : "+v"(lo));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < M::RL_MAXJP; ++q) a = sbm_sel(M::rl_jpcol(q, i) == lo, sh->A[i * M::RL_MAXJP + q], a);
        b[i] = fma(hh, a, z[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) b[i] = fma(hh, a_of(i), z[i]);
    }
    if constexpr (DIST) M::im_solve_lds(sh->MF, sh->RD, b);
    else if constexpr (M::IM_TRI) M::im_solve_tri(sh->MF, b);
    else M::im_solve(m, b);
    fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] = fma(2.0, b[i], -z[i]);
  }

  // one implicit-EULER step (hh = the step) of a sensitivity column with the matrices newton() left, in place:
  // z <- M^-1 (z + hh J_p) -- the exact derivative of y_{n+1} = y_n + hh f(y_{n+1})  (sbm_implicit_extrap.hpp)
  __device__ __forceinline__ void sens_euler(double hh, double (&z)[NV]) {
    if constexpr (Sh::A_SPARSE && M::IM_SENS_TRI && !DIST) {
      // triangular M: pick and substitution row by row, tables loaded a block ahead (emit_implicit.py)
      int lo = lane + 64 * chunk;  // (opaque: see solve_delta)
      asm volatile("" : "+v"(lo));
      M::im_sens_tri(sh->MF, sh->A, hh, lo, z);
      fence();
      return;
    } else if constexpr (Sh::A_SPARSE) {
      int lo = lane + 64 * chunk;  // (opaque: see solve_delta)
      asm volatile("" : "+v"(lo));
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double a = 0.0;
#pragma unroll
        for (int q = 0; q < M::RL_MAXJP; ++q) a = sbm_sel(M::rl_jpcol(q, i) == lo, sh->A[i * M::RL_MAXJP + q], a);
        z[i] = fma(hh, a, z[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) z[i] = fma(hh, a_of(i), z[i]);
    }
    if constexpr (DIST) M::im_solve_lds(sh->MF, sh->RD, z);
    else if constexpr (M::IM_TRI) M::im_solve_tri(sh->MF, z);
    else M::im_solve(m, z);
    fence();
  }
};
