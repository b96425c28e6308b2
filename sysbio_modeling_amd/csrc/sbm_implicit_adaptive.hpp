// sbm_implicit_adaptive.hpp -- implicit midpoint with error control INSIDE the kernel (SBM_IMPLICIT_ADAPTIVE).
//
// The reference never picks an integrator: scipy.integrate.odeint is LSODA, which goes implicit on a stiff system by
// itself and controls its error (model/ode_model.py:122-123,167-168).  The fixed-step kernel of sbm_integrators.hpp
// (sbm_imid_kernel) needed a host loop around it for that (whole-ensemble runs with n, 2n, 4n, ... steps combined in a
// Romberg table: _control.py -- seconds per ensemble).  Here the control sits inside the kernel, one launch:
//
//   THREE solutions are carried side by side and never mixed: steps of size H, H/2 and H/4 on nested grids.  The
//   implicit midpoint rule is symmetric, so all three have global errors that expand in H^2; at an output time
//       T22 = (4 y_{H/2} - y_H) / 3   and   T32 = (4 y_{H/4} - y_{H/2}) / 3
//   are two fourth-order results (passive extrapolation), T32 is written out and |T32 - T22| / 3 bounds its error as
//   long as the error falls by at least 4 per halving -- the acceptance rule of the round-1 host loop (_control.py:
//   it held on stiff50 where order reduction leaves second order, and in the asymptotic regime where the ratio is
//   16), now per trajectory inside the kernel.  Too large at any output time: the trajectory starts over with more
//   steps per unit time (the estimate falls as n^-4 at best: n <- n (est / 0.5)^(1/4) with a margin; a pass that is
//   clearly failing is abandoned at its first bad output).  Every pass uses ONE step size per output interval -- see
//   below why.  rtol / atol mean what they mean for DOPRI45: no calibration constant.
//
// Three designs that were built first and measured, and why they are not here:
//  * Step doubling with local extrapolation (advance with (4 fine - coarse) / 3 after every macro step) is UNSTABLE
//    on stiff systems: the midpoint rule's amplification factor tends to -1 for h lambda -> -infinity (A- but not
//    L-stable), two half steps give +1, so the locally extrapolated step multiplies a stiff deviation by
//    (4 - (-1)) / 3 = 5/3.  Its controller saved the run by keeping h |lambda| = O(1): 275 000 steps per vector on
//    stiff50 where 4096 suffice.
//  * Keeping the solutions apart but choosing the step count per OUTPUT INTERVAL (retry an interval from its start)
//    works for the state and fails for the sensitivities: every change of the step size leaves a deviation of the
//    stiff components from the numerical slow manifold, the rule damps it by only 1 - O(1/(h lambda)) per step, and
//    the estimator then sees an oscillation of fixed size that no refinement of the interval removes (a third of the
//    stiff50 vectors stalled around the fifth interval).  A uniform grid from t0 excites nothing.
//  * TWO solutions (H, H/2) with (fine - coarse) / 3 as the estimate: that is the error of the SECOND-order fine
//    solution, while the fourth-order combination is what is returned -- e4 = (C4 / C2^2) e2^2 with a problem-dependent
//    constant.  Calibrated on stiff50 (estimate held at sqrt(tol)) it took 0.95 s per 4096 vectors, 7000 coarse steps
//    on average where 2048 + 4096 + 8192 fixed steps give the same accuracy, and was 4 tolerance units off on a
//    binding motif with an initial layer.  The third solution costs 7 midpoint solves per coarse step instead of 3
//    and buys an estimate of the error that is actually returned.
//
// Error norm as in the explicit kernels: max(RMS over the state, max over the sensitivity columns of the column RMS),
// every element against atol + rtol max(|value|, 1e-6 x the largest entry of its column so far).  The first step of a
// trajectory is graded (13 geometric substeps, as SBM_IMPLICIT_MIDPOINT_GRADED; cut again in the finer solutions): the
// reference always starts from y = 0, possibly off a fast manifold.
//
// Mapping and the step itself: sbm_implicit_stepper.hpp (lane j = column j of S with all NV rows in registers, state
// component i on lane i mod 64, row lanes evaluate f_i / J_y / J_p by class; sparse LU from emit_implicit.py,
// distributed for triangular patterns).
// Registers: the S columns of the solution being advanced and the solver's work vector b, as in the fixed-step
// kernel; the other two solutions' columns wait in LDS ([row][column], 2 x 8 NV NKc bytes).  J_p reaches the columns through a compact table
// (RL_MAXJP values per row, picked by column index) when rows have few parameter entries, through the dense
// [row][64] table otherwise.
#pragma once

template <class M>
struct SbmImadShared {
  static constexpr bool A_SPARSE = (M::RL_MAXJP <= 4);
  static constexpr int NROW = 64 * ((M::NV + 63) / 64);
  static constexpr int A_SIZE = A_SPARSE ? (M::NV * M::RL_MAXJP + 2) : (M::NV * 64 + 2);
  double Y[NROW];               // iterate, one component per row lane (rows lane, lane + 64, ...)
  double G[NROW];               // Newton residual
  double JY[sbm_ijy_size<M>()]; // J_y non-zeros by entry index (+ spare slot): the redundant factorisation's input only
  static constexpr int MF_SIZE = sbm_imf_size<M>(), RD_SIZE = sbm_ird_size<M>();
  __attribute__((aligned(16))) double MF[MF_SIZE];   // the factors (IM_TRI: reciprocal pivots and scaled entries; IM_DIST: dense rows)
  double RD[RD_SIZE];           // IM_DIST: reciprocal pivots
  double A[A_SIZE];             // J_p: [row][slot] or [row][column]
  static constexpr int ZC = M::NK < 64 ? M::NK : 64;     // columns of a chunk that exist
  static constexpr int ZS = ZC < 64 ? ZC + 1 : 64;       // + one spare column that the idle lanes share (all zeros)
  double ZO[2][M::NV * ZS];     // S of the two solutions that are NOT being advanced at the moment, [row][column]
};

template <class M>
__global__ void __launch_bounds__(64) sbm_imid_adaptive_kernel(sbm_kernel_args a) {
  constexpr int NV = M::NV, NK = M::NK;
  constexpr int NCH = (NK + 63) / 64;
  using Sh = SbmImadShared<M>;
  using Stepper = SbmImplicitStepper<M, Sh>;
  constexpr int RPL = Stepper::RPL;
  __shared__ Sh sh;
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x;
  const int chunk = NCH > 1 ? (int)blockIdx.y : 0;
  const int col = lane + 64 * chunk;
  const bool has_col = col < NK;
  Stepper st;
  st.setup(&sh, lane, chunk, a.P + (size_t)traj * M::NP);
  __syncthreads();

  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;
  const bool with_sens = a.S != nullptr;   // wave-uniform

  // the three solutions: index 0 = steps H, 1 = H/2, 2 = H/4.  One of them is "active" (state component in ya, S
  // columns in za), the other two wait in sh.ZO / yw.
  double za[NV];
  double yw[3][RPL];            // this lane's state components (rows lane, lane + 64, ...) of the three solutions
  constexpr int ZS = Sh::ZS;
  const int zl = lane < Sh::ZC ? lane : ZS - 1; // idle lanes (beyond the chunk's columns) share the spare column: all zeros
  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
  const double rtol = a.opts.rtol > 0.0 ? a.opts.rtol : 1e-9, atol = a.opts.atol > 0.0 ? a.opts.atol : 1e-12;
  const double nrtol = 0.03 * rtol;
  const long long max_steps = a.opts.max_steps > 0 ? a.opts.max_steps : (a.opts.max_steps < 0 ? -(long long)a.opts.max_steps : 4000000LL);
  constexpr int GRADE = 12;
  constexpr int MAXPASS = 12;
  int status = SBM_OK, n_newton = 0;
  long long n_acc = 0, n_rej = 0;
  const double t_span = glen > 0 ? tg[glen - 1] - a.opts.t0 : 0.0;
  // coarse steps per unit time of the first pass
  double density = a.opts.h0 > 0.0 ? 1.0 / a.opts.h0 : 64.0 / (t_span > 0.0 ? t_span : 1.0);

  bool done = false, complete = false;
  for (int pass = 0; pass < MAXPASS && !done; ++pass) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      za[i] = (a.s0 && has_col) ? a.s0[i * NK + col] : 0.0;
      sh.ZO[0][i * ZS + zl] = za[i];
      sh.ZO[1][i * ZS + zl] = za[i];
    }
#pragma unroll
    for (int r = 0; r < RPL; ++r)
      yw[0][r] = yw[1][r] = yw[2][r] = (a.y0 && st.has_row[r]) ? a.y0[lane + 64 * r] : 0.0;
    // who is where: active solution, and which solution each LDS slot holds.  The steps run 0, 1, 2 then 2, 1, 0
    // then 0, 1, 2 ...: two exchanges with LDS per coarse step instead of three.
    int active = 0, slot0 = 1, slot1 = 2;
    double t = a.opts.t0;
    double colmax = 0.0;          // largest |S| entry of this lane's column so far
    float err_max = 0.f;
    int rc = SBM_OK;
    long long n_pass = 0;
    bool first_step = true, abandoned = false, forward = true;
    double dyp[3][RPL];                // previous increment of each solution over a coarse step: Newton predictors
#pragma unroll
    for (int r = 0; r < RPL; ++r) dyp[0][r] = dyp[1][r] = dyp[2][r] = 0.0;
    double H_prev = 0.0;
    int io = 0;
    for (; io < glen && rc == SBM_OK && !abandoned; ++io) {
      const double target = tg[io];
      const double dt = target - t;
      if (dt > 0.0) {
        const double nd = ceil(dt * density - 1e-9);
        const int n = nd < 1.0 ? 1 : (nd > 2.0e9 ? 2000000000 : (int)nd);
        if (n_acc + n_rej + n_pass + n > max_steps) { rc = SBM_MAX_STEPS; break; }
        const double H = dt / n;
        float yloc = 0.f;
#pragma unroll
        for (int r = 0; r < RPL; ++r) yloc = fmaxf(yloc, st.has_row[r] ? (float)fabs(yw[2][r]) : 0.f);
        const float ymax = sbm_wave_max(yloc);
        const double natol = fmax(0.03 * atol, 4.0e-16 * (double)ymax);
        const double sc_h = H_prev > 0.0 ? H / H_prev : 0.0;
#pragma unroll
        for (int r = 0; r < RPL; ++r) { dyp[0][r] *= sc_h; dyp[1][r] *= sc_h; dyp[2][r] *= sc_h; }
        H_prev = H;
        for (int s = 0; s < n && rc == SBM_OK; ++s) {
          const double ts = fma((double)s, H, t);
          // ONE copy of the step code (Newton + column solve) serves the three solutions and the graded first step:
          // two inlined copies, a third NV-row array to work in, or two solutions in registers overflow the register
          // file (measured: 999, 380 and 262 scratch instructions).
#pragma unroll 1
          for (int ph = 0; ph < 3; ++ph) {
            const int want = forward ? ph : 2 - ph;
            if (want != active) {
              // bring solution `want` in, park the active one in its slot
              const int sl = slot0 == want ? 0 : 1;     // (an LDS offset: no register array is indexed by it)
              if (with_sens) {
                double* zo = sh.ZO[sl];
#pragma unroll
                for (int i = 0; i < NV; ++i) { const double tmp = zo[i * ZS + zl]; zo[i * ZS + zl] = za[i]; za[i] = tmp; }
              }
              if (sl == 0) slot0 = active; else slot1 = active;
              active = want;
            }
            if (rc == SBM_OK) {
              const int parts = 1 << want;
              double ya[RPL], ystart[RPL], dy_pred[RPL];
#pragma unroll
              for (int r = 0; r < RPL; ++r) {
                ya[r] = want == 0 ? yw[0][r] : (want == 1 ? yw[1][r] : yw[2][r]);
                ystart[r] = ya[r];
                dy_pred[r] = (want == 0 ? dyp[0][r] : (want == 1 ? dyp[1][r] : dyp[2][r])) / parts;
              }
              const int nops = first_step ? (GRADE + 1) * parts : parts;
              double tt = ts;
#pragma unroll 1
              for (int k = 0; k < nops && rc == SBM_OK; ++k) {
                double h = H / parts;
                if (first_step) {
                  const int gj = k / parts;
                  h = ldexp(H, -(gj == 0 ? GRADE : GRADE - gj + 1)) / parts;
                }
                double y_before[RPL], yb[RPL];
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                  if (first_step) dy_pred[r] = 0.0;
                  y_before[r] = ya[r];
                  yb[r] = fma(0.5, dy_pred[r], ya[r]);
                }
                rc = st.template newton<10>(tt + 0.5 * h, 0.5 * h, ya, yb, nrtol, natol, n_newton);
                if (rc == SBM_OK) {
#pragma unroll
                  for (int r = 0; r < RPL; ++r) ya[r] = fma(2.0, yb[r], -ya[r]);
                  if (with_sens) st.sens(0.5 * h, za);
                }
#pragma unroll
                for (int r = 0; r < RPL; ++r) dy_pred[r] = ya[r] - y_before[r];   // the next piece starts from this piece's increment
                tt += h;
              }
#pragma unroll
              for (int r = 0; r < RPL; ++r) {
                const double inc = ya[r] - ystart[r];
                if (want == 0) { yw[0][r] = ya[r]; dyp[0][r] = inc; }
                else if (want == 1) { yw[1][r] = ya[r]; dyp[1][r] = inc; }
                else { yw[2][r] = ya[r]; dyp[2][r] = inc; }
              }
            }
          }
          forward = !forward;
          first_step = false;
        }
        n_pass += n;
        t = target;
      }
      if (rc != SBM_OK) break;
      // ---- output time: T22 = (4 y1 - y0) / 3, T32 = (4 y2 - y1) / 3; write T32, |T32 - T22| / 3 bounds its error ----
      // (the active solution is 0 or 2 here; fetch the other two from their slots)
      float cs = 0.f;
      if (with_sens) {
        const int s1 = slot0 == 1 ? 0 : 1;                 // slot of solution 1 (never active at an output time)
        const int so = 1 - s1;                              // slot of the other waiting solution
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const double z1 = sh.ZO[s1][i * ZS + zl], zo = sh.ZO[so][i * ZS + zl];
          const double z0 = active == 0 ? za[i] : zo, z2 = active == 0 ? zo : za[i];
          const double t32 = fma(z2 - z1, 1.0 / 3.0, z2), t22 = fma(z1 - z0, 1.0 / 3.0, z1);
          colmax = fmax(colmax, fabs(t32));
          const float r = (float)((t32 - t22) * (1.0 / 3.0)) *
                          __builtin_amdgcn_rcpf((float)fma(rtol, fmax(fabs(t32), 1e-6 * colmax), atol));
          cs = fmaf(r, r, cs);
          if (St && has_col) St[((size_t)io * NV + i) * NK + col] = t32;
        }
      }
      double y32[RPL];
      float yloc = 0.f;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        y32[r] = fma(yw[2][r] - yw[1][r], 1.0 / 3.0, yw[2][r]);
        yloc = fmaxf(yloc, st.has_row[r] ? (float)fabs(y32[r]) : 0.f);
      }
      const float ymax_now = sbm_wave_max(yloc);
      float ry2 = 0.f;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double y22 = fma(yw[1][r] - yw[0][r], 1.0 / 3.0, yw[1][r]);
        const float ry = st.has_row[r] ? (float)((y32[r] - y22) * (1.0 / 3.0)) *
                                             __builtin_amdgcn_rcpf((float)fma(rtol, fmax(fabs(y32[r]), 1e-6 * (double)ymax_now), atol)) : 0.f;
        ry2 += sbm_nan_to_inf(ry * ry);
      }
      const float xs = sbm_wave_sumf(ry2);
      const float mx = sbm_wave_max(has_col ? sbm_nan_to_inf(cs) : 0.f);
      const float err = sqrtf(fmaxf(mx, xs) * (1.0f / NV));
      if (!(err < 3.0e38f)) { rc = SBM_NON_FINITE; break; }
      err_max = fmaxf(err_max, err);
      if (Yt && chunk == 0) {
#pragma unroll
        for (int r = 0; r < RPL; ++r)
          if (st.has_row[r]) Yt[(size_t)io * NV + lane + 64 * r] = y32[r];
      }
      if (err_max > 16.0f && pass + 1 < MAXPASS) abandoned = true;    // clearly not good enough: do not finish the pass
    }
    complete = rc == SBM_OK && !abandoned;
    if (complete && err_max <= 1.0f) {
      n_acc += n_pass;
      done = true;
    } else if (rc == SBM_MAX_STEPS) {
      n_rej += n_pass;
      status = SBM_MAX_STEPS;
      break;
    } else {
      n_rej += n_pass;
      double grow = 4.0;                                     // Newton failure / non-finite values: a much finer grid
      // the estimate falls by 16 per halving in the asymptotic regime, by 4 under order reduction: between the two
      if (rc == SBM_OK) grow = fmin(8.0, fmax(1.3, 1.15 * pow((double)err_max / 0.5, 0.3)));
      density *= grow;
      if (pass + 1 == MAXPASS) status = rc != SBM_OK ? rc : SBM_TOL_NOT_REACHED;
    }
  }
  if (!done && status == SBM_OK) status = SBM_TOL_NOT_REACHED;
  if (!done && !(complete && status == SBM_TOL_NOT_REACHED)) {
    // a failed trajectory reports NaN rows (the last pass may have written a part of them); one that only missed
    // the tolerance keeps its finest result, as the host loop does (status SBM_TOL_NOT_REACHED)
    for (int io = 0; io < glen; ++io) {
      if (Yt && chunk == 0) {
#pragma unroll
        for (int r = 0; r < RPL; ++r)
          if (st.has_row[r]) Yt[(size_t)io * NV + lane + 64 * r] = __builtin_nan("");
      }
      if (St && has_col) {
#pragma unroll
        for (int i = 0; i < NV; ++i) St[((size_t)io * NV + i) * NK + col] = __builtin_nan("");
      }
    }
  }
  if (lane == 0) {
    const int na = (int)(n_acc > 2000000000LL ? 2000000000LL : n_acc), nr = (int)(n_rej > 2000000000LL ? 2000000000LL : n_rej);
    // chunks of a trajectory control their steps separately (each carries a copy of the state next to its own
    // columns): worst status, most steps
    if constexpr (NCH > 1) {
      if (a.status) atomicMax(a.status + traj, status);
      if (a.n_steps) atomicMax(a.n_steps + traj, na);
      if (a.n_reject) atomicMax(a.n_reject + traj, nr);
    } else {
      if (a.status) a.status[traj] = status;
      if (a.n_steps) a.n_steps[traj] = na;
      if (a.n_reject) a.n_reject[traj] = nr;
    }
  }
  (void)n_newton;
}
