// sbm_implicit_extrap.hpp -- SBM_IMPLICIT_EXTRAP: extrapolated implicit Euler with LOCAL step-size control in the kernel.
//
// The reference never picks an integrator: scipy.integrate.odeint is LSODA, which goes implicit on a stiff system by
// itself and controls its LOCAL error step by step (model/ode_model.py:122-123,167-168).  This kernel is the GPU's
// answer for stiff systems: an L-stable embedded one-step scheme with a per-trajectory step size chosen inside the
// kernel, one launch per ensemble, no step count to choose and no restarts.
//
// One macro step of size H from (t, y, S):
//   for j = 1 .. K:   T_j = j implicit-Euler steps of size H / j from (y, S)          (harmonic sequence)
//       y_{m+1} = y_m + h f(y_{m+1}):     Newton, M = I - h J_y, factored per iterate (sbm_implicit_stepper.hpp)
//       M S_{m+1} = S_m + h J_p(y_{m+1}): the EXACT derivative of the step, one solve per column with Newton's factors
//   T_KK    = sum_j wH_j T_j   (polynomial extrapolation to h = 0 through all K results: order K)
//   T_K,K-1 = sum_j wL_j T_j   (the same through T_2 .. T_K: order K - 1)
//   err = || T_KK - T_K,K-1 ||  (state and every sensitivity column, scaled by atol + rtol |value|);
//   err <= 1: continue from T_KK (local extrapolation); either way H <- H * 0.9 err^(-1/K) (within [0.2, 4]).
// Every T_j is a composition of implicit-Euler steps, R(z) = 1 / (1 - z): stiff components are DAMPED by every T_j and
// hence by any combination of them (R(infinity) = 0, A(alpha)-stable with alpha > 87 degrees up to K = 8, Hairer &
// Wanner IV.9) -- which is what makes local extrapolation and a change of step size harmless here, where the
// implicit midpoint rule (R(infinity) = -1) amplifies stiff deviations by 5/3 under local extrapolation and keeps
// the trace of every step-size change (csrc/sbm_implicit_adaptive.hpp: why that kernel integrates three solutions
// on uniform grids from t0 and restarts whole trajectories).  The combination of exact discrete sensitivities is the
// exact derivative of the combined scheme.
//
// Why not ESDIRK / Rosenbrock-W: an s-stage method keeps s - 1 stage derivatives of every sensitivity column alive
// (order 4: five vectors of NV values per lane, order 5: six -- 20 KB of LDS each for 50 state variables, or the
// register file three times over).  The extrapolation needs the column being advanced and TWO running sums,
// whatever the order: zh = sum wH_j (T_j - S_n), ze = sum (wH_j - wL_j) (T_j - S_n); S_n waits in LDS.
//
// Cost: K (K + 1) / 2 Euler steps per macro step (36 at K = 8), each a Newton iteration on the state (2 - 3 solves)
// plus one solve per sensitivity column.  Measured against the hand-chosen fixed-step Richardson pair of
// sbm_imid_kernel (4096 + 8192 midpoint steps per trajectory): see DESIGN.md section 5.
//
// opts: rtol / atol as for DOPRI45 (state AND sensitivities, column by column); h0 = first step (<= 0: automatic);
// step_mult = extrapolation order K (0: chosen by rtol; 2 .. SBM_IEX_KMAX); max_steps = macro-step attempts per
// trajectory (0: 200000).  n_steps = accepted macro steps, n_reject = rejected ones.
#pragma once

constexpr int SBM_IEX_KMAX = 10;

struct SbmIexWeights {
  double wh[SBM_IEX_KMAX + 1][SBM_IEX_KMAX + 1];   // [K][j]: weight of T_j in T_KK
  double we[SBM_IEX_KMAX + 1][SBM_IEX_KMAX + 1];   // [K][j]: weight of T_j in T_KK - T_K,K-1
};
constexpr SbmIexWeights sbm_iex_make_weights() {
  SbmIexWeights w{};
  for (int K = 2; K <= SBM_IEX_KMAX; ++K) {
    for (int j = 1; j <= K; ++j) {
      double h = 1.0, l = j >= 2 ? 1.0 : 0.0;
      for (int i = 1; i <= K; ++i) {
        if (i == j) continue;
        h *= (double)j / (double)(j - i);
        if (i >= 2 && j >= 2) l *= (double)j / (double)(j - i);
      }
      w.wh[K][j] = h;
      w.we[K][j] = h - l;
    }
  }
  return w;
}
__constant__ SbmIexWeights SBM_IEX_W = sbm_iex_make_weights();

template <class M>
struct SbmIexShared {
  static constexpr bool A_SPARSE = (M::RL_MAXJP <= 4);
  static constexpr int NROW = 64 * ((M::NV + 63) / 64);
  static constexpr int A_SIZE = A_SPARSE ? (M::NV * M::RL_MAXJP + 2) : (M::NV * 64 + 2);
  double Y[NROW];               // iterate, one component per row lane (rows lane, lane + 64, ...)
  double G[NROW];               // Newton residual
  double JY[sbm_ijy_size<M>()]; // J_y non-zeros by entry index (+ spare slot): the redundant factorisation's input only
  static constexpr int MF_SIZE = sbm_imf_size<M>(), RD_SIZE = sbm_ird_size<M>();
  __attribute__((aligned(16))) double MF[MF_SIZE];   // the factors (sbm_implicit_stepper.hpp)
  double RD[RD_SIZE];
  __attribute__((aligned(16))) double A[A_SIZE];   // J_p: [row][slot] or [row][column]; aligned: im_sens_tri reads pairs
  static constexpr int ZC = M::NK < 64 ? M::NK : 64;     // columns of a chunk that exist
  static constexpr int ZS = ZC < 64 ? ZC + 1 : 64;       // + one spare column that the idle lanes share (all zeros)
  double ZN[M::NV * ZS];        // S at the start of the macro step, [row][column]
};

template <class M>
struct SbmIexFits {
  static constexpr bool value = M::NV <= SBM_IMPLICIT_MAX_NV && sizeof(SbmIexShared<M>) <= 160u * 1024u;
};

template <class M>
__global__ void __launch_bounds__(64) sbm_iex_kernel(sbm_kernel_args a) {
  constexpr int NV = M::NV, NK = M::NK;
  constexpr int NCH = (NK + 63) / 64;
  using Sh = SbmIexShared<M>;
  using Stepper = SbmImplicitStepper<M, Sh>;
  constexpr int RPL = Stepper::RPL;
  constexpr int ZS = Sh::ZS;
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
  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
  const double rtol = a.opts.rtol > 0.0 ? a.opts.rtol : 1e-8, atol = a.opts.atol > 0.0 ? a.opts.atol : 1e-11;
  // The extrapolation multiplies what the Newton iteration leaves in a T_j by weights of up to 10^3: the iteration is
  // converged far below the integration tolerance (quadratic convergence: the last of its 2 - 3 iterations is the one
  // this buys; measured on stiff50: with 0.03 rtol, the fixed-step kernel's setting, the noise drives the step-size
  // control into rejections and the run costs MORE)
  // -- and the sensitivity solves take J_y, J_p at the iterate of the LAST evaluation, which is off by the last update:
  // an evaluation point good to 1e-3 rtol leaves 1e-3 rtol x sum |w| = 3 rtol of noise in the error estimate (seen as
  // step counts that depend on the predictor's quality: 142 / 171 / 156 macro steps for three predictors on the same
  // vector; all 142 since).  Hence 1e-5 rtol: the final evaluation sits at the converged state to rounding.
  const double nrtol = fmax(1e-5 * rtol, 4e-15);
  int K = a.opts.step_mult;
  if (K <= 0) K = rtol >= 1e-4 ? 4 : (rtol >= 1e-6 ? 6 : 8);
  K = K < 2 ? 2 : (K > SBM_IEX_KMAX ? SBM_IEX_KMAX : K);
  const float expo = -1.0f / (float)K;
#ifndef SBM_IEX_FLOOR
#define SBM_IEX_FLOOR 1e-6
#endif
  const double floor_rel = SBM_IEX_FLOOR;
  const long long max_steps = a.opts.max_steps > 0 ? a.opts.max_steps : (a.opts.max_steps < 0 ? -(long long)a.opts.max_steps : 200000LL);
  const int zl = lane < Sh::ZC ? lane : ZS - 1; // idle lanes (beyond the chunk's columns) share the spare column: all zeros

  // S_n in LDS, the state components of this lane's rows in registers
  double yn[RPL], ydot[RPL];
#pragma unroll
  for (int i = 0; i < NV; ++i) sh.ZN[i * ZS + zl] = (a.s0 && has_col) ? a.s0[i * NK + col] : 0.0;
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    yn[r] = (a.y0 && st.has_row[r]) ? a.y0[lane + 64 * r] : 0.0;
    ydot[r] = 0.0;                // slope of the last accepted macro step: predictor of a sequence's first Newton iteration
  }
  Stepper::fence();

  int status = SBM_OK, n_newton = 0;
  long long n_acc = 0, n_rej = 0;
  double t = a.opts.t0;
  const double t_span = glen > 0 ? tg[glen - 1] - a.opts.t0 : 0.0;
  double H = a.opts.h0 > 0.0 ? a.opts.h0 : 1e-3 * (t_span > 0.0 ? t_span : 1.0);
  double colmax = 0.0;            // largest |S| entry of this lane's column so far
  bool after_reject = false;

  for (int io = 0; io < glen; ++io) {
    const double target = tg[io];
    while (status == SBM_OK && t < target) {
      if (n_acc + n_rej >= max_steps) { status = SBM_MAX_STEPS; break; }
      const double rem = target - t;
      const bool landing = H * 1.0001 >= rem;       // (wave-uniform)
      const double Hs = landing ? rem : H;
      if (!(Hs > 1e-14 * fmax(fabs(t), fabs(target)))) { status = SBM_STEP_UNDERFLOW; break; }
      float yloc = 0.f;
#pragma unroll
      for (int r = 0; r < RPL; ++r) yloc = fmaxf(yloc, st.has_row[r] ? (float)fabs(yn[r]) : 0.f);
      const float ymax = sbm_wave_max(yloc);
      const double natol = fmax(1e-5 * atol, 4.0e-16 * (double)ymax);

      double zs[NV], zh[NV], ze[NV];
      double yh[RPL], ye[RPL];
#pragma unroll
      for (int i = 0; i < NV; ++i) { zh[i] = 0.0; ze[i] = 0.0; }
#pragma unroll
      for (int r = 0; r < RPL; ++r) { yh[r] = 0.0; ye[r] = 0.0; }
      int rc = SBM_OK;
      // ONE copy of the Euler step (Newton + column solve) serves all sequences
#pragma unroll 1
      for (int j = 1; j <= K && rc == SBM_OK; ++j) {
        const double h = Hs / (double)j;
        double ya[RPL], yp[RPL], yp2[RPL], yp3[RPL];     // the last four points of this sequence
#pragma unroll
        for (int r = 0; r < RPL; ++r) { ya[r] = yn[r]; yp[r] = fma(-h, ydot[r], yn[r]); yp2[r] = yp3[r] = 0.0; }
        if (with_sens) {
#pragma unroll
          for (int i = 0; i < NV; ++i) zs[i] = sh.ZN[i * ZS + zl];
        }
#pragma unroll 1
        for (int m = 0; m < j && rc == SBM_OK; ++m) {
          // predictor: the polynomial through the last 2 / 3 / 4 points of the sequence (first step: H/j x the slope
          // of the last macro step) -- 3.2 -> 2.8 evaluations per Euler step against the linear one
          double yb[RPL];
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const double lin = fma(2.0, ya[r], -yp[r]);
            const double quad = fma(3.0, ya[r] - yp[r], yp2[r]);
            const double cub = fma(4.0, ya[r] + yp2[r], fma(-6.0, yp[r], -yp3[r]));
            yb[r] = m < 2 ? lin : (m == 2 ? quad : cub);
          }
          rc = st.template newton_rate<8>(fma((double)(m + 1), h, t), h, ya, yb, nrtol, natol, n_newton);
          if (rc == SBM_OK) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) { yp3[r] = yp2[r]; yp2[r] = yp[r]; yp[r] = ya[r]; ya[r] = yb[r]; }
            if (with_sens) st.sens_euler(h, zs);
          }
        }
        if (rc == SBM_OK) {
          const double wh = SBM_IEX_W.wh[K][j], we = SBM_IEX_W.we[K][j];
          if (with_sens) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              const double d = zs[i] - sh.ZN[i * ZS + zl];
              zh[i] = fma(wh, d, zh[i]);
              ze[i] = fma(we, d, ze[i]);
            }
          }
#pragma unroll
          for (int r = 0; r < RPL; ++r) {
            const double d = ya[r] - yn[r];
            yh[r] = fma(wh, d, yh[r]);
            ye[r] = fma(we, d, ye[r]);
          }
        }
      }
      float err = __builtin_inff();
      double colmax_new = colmax;
      if (rc == SBM_OK) {
        float cs = 0.f;
        if (with_sens) {
#pragma unroll
          for (int i = 0; i < NV; ++i) colmax_new = fmax(colmax_new, fabs(sh.ZN[i * ZS + zl] + zh[i]));
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const double tk = sh.ZN[i * ZS + zl] + zh[i];
            // (the scale clamped in double: an atol below ~1e-38 must not turn into 1 / 0 in single precision)
            const float r = (float)ze[i] * __builtin_amdgcn_rcpf((float)fmax(fma(rtol, fmax(fabs(tk), floor_rel * colmax_new), atol), 1e-30));
            cs = fmaf(r, r, cs);
          }
        }
        float ykl = 0.f;
#pragma unroll
        for (int r = 0; r < RPL; ++r) ykl = fmaxf(ykl, st.has_row[r] ? (float)fabs(yn[r] + yh[r]) : 0.f);
        const float ykmax = sbm_wave_max(sbm_nan_to_inf(ykl));
        float ry2 = 0.f;
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const double yk = yn[r] + yh[r];
          const float ry = st.has_row[r] ? (float)ye[r] * __builtin_amdgcn_rcpf((float)fmax(fma(rtol, fmax(fabs(yk), floor_rel * (double)ykmax), atol), 1e-30)) : 0.f;
          ry2 += sbm_nan_to_inf(ry * ry);
        }
        const float xs = sbm_wave_sumf(ry2);
        const float mx = sbm_wave_max(has_col ? sbm_nan_to_inf(cs) : 0.f);
        err = sqrtf(fmaxf(mx, xs) * (1.0f / NV));
        if (!(err == err)) err = __builtin_inff();
      }
      if (err <= 1.0f) {
        // ---- accept: continue from T_KK ----
#pragma unroll
        for (int r = 0; r < RPL; ++r) { ydot[r] = yh[r] / Hs; yn[r] += yh[r]; }
        if (with_sens) {
#pragma unroll
          for (int i = 0; i < NV; ++i) sh.ZN[i * ZS + zl] += zh[i];
          Stepper::fence();
        }
        colmax = colmax_new;
        t = landing ? target : t + Hs;
        ++n_acc;
        float fac = err > 1e-12f ? 0.9f * __powf(err, expo) : 4.0f;
        fac = fminf(after_reject ? 1.0f : 4.0f, fmaxf(0.2f, fac));
        // a step clipped to land on an output time says little about the step size the solution allows
        if (!landing || fac < 1.0f || Hs * (double)fac > H) H = Hs * (double)fac;
        after_reject = false;
      } else {
        // ---- reject (error too large, Newton failure or non-finite values): a smaller step ----
        ++n_rej;
        float fac = 0.25f;
        if (rc == SBM_OK && err < 3.0e38f) fac = fminf(0.9f, fmaxf(0.1f, 0.9f * __powf(err, expo)));
        H = Hs * (double)fac;
        after_reject = true;
      }
    }
    const bool failed = status != SBM_OK;
    if (Yt && chunk == 0) {
#pragma unroll
      for (int r = 0; r < RPL; ++r)
        if (st.has_row[r]) Yt[(size_t)io * NV + lane + 64 * r] = failed ? __builtin_nan("") : yn[r];
    }
    if (St && has_col) {
#pragma unroll
      for (int i = 0; i < NV; ++i) St[((size_t)io * NV + i) * NK + col] = failed ? __builtin_nan("") : sh.ZN[i * ZS + zl];
    }
  }
  if (lane == 0) {
#ifdef SBM_IEX_COUNT_NEWTON      // developer build: n_reject carries the evaluations of f / J_y / J_p
    n_rej = n_newton;
#endif
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
