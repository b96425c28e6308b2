// sbm_implicit_extrap_seq.hpp -- SBM_IMPLICIT_EXTRAP for CHAIN models, round 4: the Newton iterations of the K
// sequences of a macro step run SIDE BY SIDE, and the wavefronts are persistent.
//
// What sbm_iex_kernel (sbm_implicit_extrap.hpp) waits for.  Its 36 implicit-Euler steps per macro step run one after the
// other, and each is a Newton iteration on 50 state components with ONE component per lane: ~170 instructions per
// iteration of which a row's own arithmetic is ~60 -- the rest is the LDS round trip that hands the iterate round, a
// six-level DPP prefix for the update, a six-level DPP maximum and a scalar branch for the convergence test, every one
// of them waiting for the one before.  At one wavefront per SIMD (456 VGPRs: three copies of a sensitivity column)
// nothing hides that; measured (profiles/r03): state-only passes cost 72 % of a pass with sensitivities per macro step,
// VALU busy 61 %.
//
// The K sequences T_1 .. T_K of a macro step are independent of each other (all start from (y_n, S_n)), and the state
// never depends on the sensitivities.  So, per macro step:
//
//   phase A  the state of ALL sequences.  The 64 lanes form 4 groups of 16 (one DPP row each); group g integrates
//            sequence K - g and then sequence g + 1 -- K + 1 Euler steps per group, the whole harmonic sequence in K + 1
//            slots instead of K (K + 1) / 2 steps in a row.  A lane holds RPG = ceil(NV / 16) CONSECUTIVE state rows: in
//            a slot the fixed costs of a Newton iteration (hand-over, prefix, maximum, branch) are paid once for up to
//            four Euler steps, and the row arithmetic of RPG rows per lane gives the pipeline independent work.  The
//            Newton update of a chain Jacobian is the recurrence x_i = b_i + a_i x_{i-1}: composed along a lane's own
//            rows in registers, along the 16 lanes of a group by the DPP prefix of SbmImplicitStepper::chain_level.
//            Every group follows the stopping rule of SbmImplicitStepper::newton_rate on its own; a group that has
//            finished idles until the slowest has.  For every Euler step the factors of M = I - h J_y and J_p AT THE
//            CONVERGED STATE go to a table in global memory (this workgroup's slice of a scratch buffer: 2 KB per step,
//            written once, read once, L2-resident);
//   phase B  the sensitivity columns, sequence by sequence as before (one column per lane, zs / zh / ze in registers,
//            S_n in LDS): per Euler step the table of that step is copied into the LDS tables sbm_iex_kernel's
//            im_sens_tri reads (loaded one step ahead), then M S_1 = S_0 + h J_p as before.  Nothing in phase B waits
//            for a Newton iteration any more.
//
// Persistent wavefronts: the grid is what the chip holds at once (one wavefront per SIMD); a workgroup takes the next
// trajectory (longest first, a.order) from an atomic counter until none is left -- the launch ends when the work does,
// not when the unluckiest SIMD has finished its fixed share.
//
// Same scheme, same tolerances, same controller as sbm_iex_kernel; results differ from it by the rounding of a
// differently associated Newton update (scheme oracle: oracle/iex_oracle.py).  Models that are not chains, or orders
// K > 8, run sbm_iex_kernel.
#pragma once

#ifndef SBM_SEQ_PF
#define SBM_SEQ_PF 4      // Euler steps a table is loaded ahead of its use (rotated columns)
#endif

// 1 / j for the sequence lengths j = 1 .. 8 (H / j formed as one product, the same in both phases), as a select chain on
// literals: j differs from lane group to lane group in phase A, and a table look-up with a per-lane index is a vector
// memory load -- a microsecond per slot (measured: +4 ms per pass of configs[4])
__device__ __forceinline__ double sbm_iex_rj(int j) {
  double r = 1.0;
  r = j == 2 ? 1.0 / 2 : r;
  r = j == 3 ? 1.0 / 3 : r;
  r = j == 4 ? 1.0 / 4 : r;
  r = j == 5 ? 1.0 / 5 : r;
  r = j == 6 ? 1.0 / 6 : r;
  r = j == 7 ? 1.0 / 7 : r;
  r = j == 8 ? 1.0 / 8 : r;
  return r;
}

template <class M>
struct SbmIexSeqPlan {
  static constexpr int LPG = 16;                                 // lanes per group = one DPP row
  static constexpr int NG = 4;                                   // groups: sequences K - g and g + 1 one after the other
  static constexpr int RPG = (M::NV + LPG - 1) / LPG;            // consecutive rows per lane
  static constexpr int NROWP = LPG * RPG;                        // rows incl. padding
  static constexpr int W = 2 + M::RL_MAXJP;                      // per row: 1 / M_ii, h J_y[i][i-1] / M_ii, J_p[slot ...]
  static constexpr int LW = RPG * W + ((RPG * W) & 1);           // doubles per lane and step (16-byte blocks)
  // rotated columns (ROT): a step's table = RC[64][2] (1 / M_ii, h J_y[i][i-1] / M_ii by row) + A[64] (the row's J_p entry)
  static constexpr int ROT_STEP_DOUBLES = 128 + 64;
  static constexpr bool ROT_OK = M::IM_ROT && M::RL_MAXJP == 1 && (RPG % 2) == 0;
  static constexpr int NRING = 5;                                // Euler steps' tables in LDS at once (prefetch ring)
  static constexpr int KMAX = 8;
  static constexpr int NSTEP = KMAX * (KMAX + 1) / 2;
  static constexpr int STEP_DOUBLES = LPG * LW;                  // one Euler step's table
  static constexpr size_t BLOCK_DOUBLES = (size_t)NSTEP * (STEP_DOUBLES > ROT_STEP_DOUBLES ? STEP_DOUBLES : ROT_STEP_DOUBLES);   // scratch per workgroup
  static constexpr bool OK = M::IM_TRI && M::IM_CHAIN && M::IM_SENS_TRI && !M::IM_DIST && M::RL_MAXJP <= 4 && M::NV <= 64 &&
                             M::RL_MAXJP >= 1;
};

template <class M>
struct SbmIexSeqShared : SbmIexShared<M> {
  using P = SbmIexSeqPlan<M>;
  double YG[P::NG][64];                                // the iterate of every group's sequence, by row
  double TG[P::KMAX][64];                              // T_j (state) of every sequence, by row
  // rotated columns: a ring of NRING step tables (RC[64][2] + A[64] each) that LDS-direct loads fill ahead of their use,
  // + a guard (a lane reads up to row 2 NV - 2 of "its" table: what lies behind a table only has to be finite)
  __attribute__((aligned(16))) double RING[P::ROT_OK ? P::NRING * P::ROT_STEP_DOUBLES + 16 : 2];
};

template <class M>
struct SbmIexSeqFits {
  static constexpr bool value = SbmIexSeqPlan<M>::OK && SbmIexFits<M>::value && sizeof(SbmIexSeqShared<M>) <= 40u * 1024u;
};

// max over the 16 lanes of a DPP row of v >= 0, in every lane of the row (bit patterns of non-negative floats order as integers)
__device__ __forceinline__ float sbm_row16_max(float v) {
  int x = __float_as_int(v);
  x = sbm_dpp_smax<0xb1, 0xf>(x);    // quad_perm:[1,0,3,2]
  x = sbm_dpp_smax<0x4e, 0xf>(x);    // quad_perm:[2,3,0,1]
  x = sbm_dpp_smax<0x124, 0xf>(x);   // row_ror:4
  x = sbm_dpp_smax<0x128, 0xf>(x);   // row_ror:8
  return __int_as_float(x);
}

template <class M, bool ROT>
__global__ void __launch_bounds__(64) sbm_iex_seq_kernel(sbm_kernel_args a, double* __restrict__ scratch, int* __restrict__ counter,
                                                         int n_work, int nch_launch) {
  constexpr int NV = M::NV, NK = M::NK;
  using Pl = SbmIexSeqPlan<M>;
  using Sh = SbmIexSeqShared<M>;
  using Stepper = SbmImplicitStepper<M, Sh>;
  static_assert(Stepper::RPL == 1, "chain models on up to 64 state variables");
  constexpr int RPG = Pl::RPG, W = Pl::W, LW = Pl::LW, ZS = Sh::ZS;
  constexpr int MODE_ITER = 0, MODE_FINAL = 1, MODE_DONE = 2;
  __shared__ Sh sh;
  const int lane = threadIdx.x;
  const int grp = lane >> 4, gl = lane & 15;
  double* const tblock = scratch + (size_t)blockIdx.x * Pl::BLOCK_DOUBLES;
  const bool with_sens = a.S != nullptr;   // wave-uniform

  for (;;) {
    // ---- next piece of work: (trajectory, column chunk) ----
    int work = 0;
    if (lane == 0) work = atomicAdd(counter, 1);
    work = __builtin_amdgcn_readfirstlane(work);
    if (work >= n_work) break;               // (every wavefront gets here: the counter only grows)
    const int wt = work / nch_launch;
    const int chunk = work - wt * nch_launch;
    const int traj = a.order ? a.order[wt] : wt;
    const int col = lane + 64 * chunk;
    const bool has_col = col < NK;
    Stepper st;
    st.setup(&sh, lane, chunk, a.P + (size_t)traj * M::NP);
    Stepper::fence();

    const int goff = a.grid_off ? a.grid_off[traj] : 0;
    const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
    const double* tg = a.t_out + goff;
    double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * NV : nullptr;
    double* St = a.S ? a.S + (size_t)traj * a.n_t * NV * NK : nullptr;
    const double rtol = a.opts.rtol > 0.0 ? a.opts.rtol : 1e-8, atol = a.opts.atol > 0.0 ? a.opts.atol : 1e-11;
    const double nrtol = fmax(1e-5 * rtol, 4e-15);            // (why so far below rtol: sbm_implicit_extrap.hpp)
    int K = a.opts.step_mult;
    if (K <= 0) K = rtol >= 1e-4 ? 4 : (rtol >= 1e-6 ? 6 : 8);
    K = K < 2 ? 2 : (K > Pl::KMAX ? Pl::KMAX : K);            // (the launcher sends K > KMAX to sbm_iex_kernel)
    const float expo = -1.0f / (float)K;
    const double floor_rel = SBM_IEX_FLOOR;
    const long long max_steps = a.opts.max_steps > 0 ? a.opts.max_steps : (a.opts.max_steps < 0 ? -(long long)a.opts.max_steps : 200000LL);
    const int zl = lane < Sh::ZC ? lane : ZS - 1;

    // ROT: register k of this lane's column = row (r0 + k) mod NV, r0 = the row of the column's one J_p entry (rows above
    // it are structurally zero in a chain); idle lanes sit on the zero rows of the coefficient table
    const int r0 = ROT ? (has_col ? M::im_r0(has_col ? col : 0) : 64) : 0;
    const int jq = ROT ? M::im_jpq(has_col ? col : 0) : 0;
    if constexpr (ROT) {
      for (int i = lane; i < Pl::NRING * Pl::ROT_STEP_DOUBLES + 16; i += 64) sh.RING[i] = 0.0;
    }
    double yn[1], ydot[1];
#pragma unroll
    for (int i = 0; i < NV; ++i) sh.ZN[i * ZS + zl] = (!ROT && a.s0 && has_col) ? a.s0[i * NK + col] : 0.0;
    yn[0] = (a.y0 && st.has_row[0]) ? a.y0[lane] : 0.0;
    ydot[0] = 0.0;
    Stepper::fence();

    int status = SBM_OK;
    long long n_acc = 0, n_rej = 0;
#ifdef SBM_SEQ_PROFILE
    long long prof_a = 0, prof_b = 0;
#if defined(SBM_SEQ_PROFILE2) || defined(SBM_SEQ_PROFILE3) || defined(SBM_SEQ_PROFILE4) || defined(SBM_SEQ_PROFILE5) || defined(SBM_SEQ_PROFILE6) || defined(SBM_SEQ_PROFILE7) || defined(SBM_SEQ_PROFILE8)
    long long prof_s = 0;
#endif
    const long long prof_t0 = __builtin_readcyclecounter();
#endif
    double t = a.opts.t0;
    const double t_span = glen > 0 ? tg[glen - 1] - a.opts.t0 : 0.0;
    double H = a.opts.h0 > 0.0 ? a.opts.h0 : 1e-3 * (t_span > 0.0 ? t_span : 1.0);
    double colmax = 0.0;
    bool after_reject = false;
    float cest = __builtin_inff();      // contraction constant of this lane group's Newton iterations (phase A)

    for (int io = 0; io < glen; ++io) {
      const double target = tg[io];
      while (status == SBM_OK && t < target) {
        if (n_acc + n_rej >= max_steps) { status = SBM_MAX_STEPS; break; }
        const double rem = target - t;
        const bool landing = H * 1.0001 >= rem;
        const double Hs = landing ? rem : H;
        if (!(Hs > 1e-14 * fmax(fabs(t), fabs(target)))) { status = SBM_STEP_UNDERFLOW; break; }
        const float ymax = sbm_wave_max(st.has_row[0] ? (float)fabs(yn[0]) : 0.f);
        const double natol = fmax(1e-5 * atol, 4.0e-16 * (double)ymax);

        // =============== phase A: the state of all sequences ===============
#ifdef SBM_SEQ_PROFILE
        const long long tp0 = __builtin_readcyclecounter();
#endif
        int rc = SBM_OK;
        double yh[1] = {0.0}, ye[1] = {0.0};
        {
          // this lane's rows of phase A: row = gl RPG + r.  Fetched per macro step (a few dozen cached loads against 36 Euler
          // steps) and hidden from loop-invariant code motion: kept across phase B they cost registers the columns need.
          int glo = gl;
          asm volatile("" : "+v"(glo));
          const double* Ptraj = a.P + (size_t)traj * M::NP;
          bool hasA[RPG], d0A[RPG];
          int clsA[RPG], yidxA[RPG][M::RL_MAXYS];
          double psA[RPG][M::RL_MAXPS];
#pragma unroll
          for (int r = 0; r < RPG; ++r) {
            const int row = glo * RPG + r;
            hasA[r] = row < NV;
            const int rr = hasA[r] ? row : 0;       // (rows of the padding repeat row 0: finite values nobody reads)
            clsA[r] = M::rl_class(rr);
            d0A[r] = M::im_diagslot(rr) == 0;
#pragma unroll
            for (int q = 0; q < M::RL_MAXYS; ++q) yidxA[r][q] = M::rl_ys(q, rr);
#pragma unroll
            for (int q = 0; q < M::RL_MAXPS; ++q) psA[r][q] = Ptraj[M::rl_ps(q, rr)];
          }
          sh.Y[lane] = yn[0];
          sh.G[lane] = ydot[0];
          Stepper::fence();
          // group g runs sequence K - g, then sequence g + 1 (K + 1 Euler steps in all): the harmonic sequence in K / 2 rows
          const bool grp_on = grp < (K + 1) / 2;
          const int j_first = K - grp;
          const int j_second = (grp_on && grp + 1 < j_first) ? grp + 1 : 0;
          // a lane's rows of one Euler step's table: general layout [row][W]; rotated columns RC[row][2] and A[row]
          auto store_table = [&](double* dst, const double (&tw)[LW]) {
            if constexpr (ROT) {
#pragma unroll
              for (int r = 0; r < RPG; ++r) *reinterpret_cast<double2*>(dst + 2 * r) = double2{tw[r * W], tw[r * W + 1]};
              double* const da = dst - (size_t)gl * RPG * 2 + 128 + (size_t)gl * RPG;
#pragma unroll
              for (int r = 0; r < RPG; r += 2) *reinterpret_cast<double2*>(da + r) = double2{tw[r * W + 2], tw[(r + 1) * W + 2]};
            } else {
#pragma unroll
              for (int q = 0; q < LW; q += 2) *reinterpret_cast<double2*>(dst + q) = double2{tw[q], tw[q + 1]};
            }
          };
          double ya[RPG], yp[RPG], yp2[RPG], yp3[RPG];
#pragma unroll
          for (int r = 0; r < RPG; ++r) ya[r] = yp[r] = yp2[r] = yp3[r] = 0.0;
#ifdef SBM_SEQ_PROFILE7
          asm volatile("" ::: "memory");
          prof_s += __builtin_readcyclecounter() - tp0;
#endif
#pragma unroll 1
          for (int s = 0; s <= K && rc == SBM_OK; ++s) {
            const bool in_first = s < j_first;
            const int jc = in_first ? j_first : j_second;       // the sequence this group works on (0: none)
            const int m = in_first ? s : s - j_first;           // its Euler step
            const bool act = grp_on && jc > 0 && m < jc;
            const double hj = Hs * sbm_iex_rj(jc);        // (the same product in phase B: the two must agree to the bit)
            if (m == 0) {
              // a sequence begins: history from (y_n, slope of the last macro step)
#pragma unroll
              for (int r = 0; r < RPG; ++r) {
                const int row = hasA[r] ? gl * RPG + r : 0;
                ya[r] = sh.Y[row];
                yp[r] = fma(-hj, sh.G[row], ya[r]);
                yp2[r] = yp3[r] = 0.0;
              }
            }
            // predictor: the polynomial through the last 2 / 3 / 4 points of the sequence, as one combination with the
            // coefficients of the group's step number (no select per row)
            const double c0 = m < 2 ? 2.0 : (m == 2 ? 3.0 : 4.0), c1 = m < 2 ? -1.0 : (m == 2 ? -3.0 : -6.0);
            const double c2 = m < 2 ? 0.0 : (m == 2 ? 1.0 : 4.0), c3 = m < 3 ? 0.0 : -1.0;
            double yb[RPG];
#pragma unroll
            for (int r = 0; r < RPG; ++r) yb[r] = fma(c0, ya[r], fma(c1, yp[r], fma(c2, yp2[r], c3 * yp3[r])));
            const double tm = fma((double)(m + 1), hj, t);
            // table of Euler step m of sequence jc, this lane's rows
            double* const trow = ROT ? tblock + (size_t)((jc * (jc - 1)) / 2 + m) * Pl::ROT_STEP_DOUBLES + (size_t)gl * RPG * 2
                                     : tblock + (size_t)((jc * (jc - 1)) / 2 + m) * Pl::STEP_DOUBLES + (size_t)gl * LW;
            int mode = act ? MODE_ITER : MODE_DONE;
            float r_prev = 0.f;
#pragma unroll 1
            for (int it = 0;; ++it) {
#ifdef SBM_SEQ_PROFILE6
              const long long tq6a = __builtin_readcyclecounter();
              asm volatile("" ::: "memory");
#endif
#pragma unroll
              for (int r = 0; r < RPG; ++r) sh.YG[grp][gl * RPG + r] = yb[r];
              Stepper::fence();
              // (ONE straight-line block over the lane's rows: the reciprocals' refinement chains of different rows overlap;
              // the table values wait in registers for the pass in which the group finishes)
              double Ap[RPG], Bp[RPG], tw[LW];
#pragma unroll
              for (int q = 0; q < LW; ++q) tw[q] = 0.0;
#pragma unroll
              for (int r = 0; r < RPG; ++r) {
                double ys[M::RL_MAXYS];
#pragma unroll
                for (int q = 0; q < M::RL_MAXYS; ++q) ys[q] = sh.YG[grp][yidxA[r][q]];
                double f = 0.0, jy[M::RL_MAXJY], jp[M::RL_MAXJP];
#pragma unroll
                for (int q = 0; q < M::RL_MAXJY; ++q) jy[q] = 0.0;
#pragma unroll
                for (int q = 0; q < M::RL_MAXJP; ++q) jp[q] = 0.0;
                M::class_dispatch(clsA[r], tm, ys, psA[r], f, jy, jp);
                // a chain row: the diagonal entry and ONE entry left of it (slot 0 / 1 in either order)
                double jd, off;
                if constexpr (M::RL_MAXJY == 2) {
                  jd = sbm_sel(d0A[r], jy[0], jy[1]);
                  off = sbm_sel(d0A[r], jy[1], jy[0]);
                } else {
                  jd = jy[0];
                  off = 0.0;
                }
                const double rd = sbm_rcp(fma(-hj, jd, 1.0));
                const double ca = off * (hj * rd);
                const double cb = ((yb[r] - ya[r]) - hj * f) * rd;
                // x_r = cb + ca x_{r-1}: composed with the rows above it in this lane
                Bp[r] = r == 0 ? cb : fma(ca, Bp[r > 0 ? r - 1 : 0], cb);
                Ap[r] = r == 0 ? ca : ca * Ap[r > 0 ? r - 1 : 0];
                tw[r * W] = rd;
                tw[r * W + 1] = ca;
#pragma unroll
                for (int q = 0; q < M::RL_MAXJP; ++q) tw[r * W + 2 + q] = jp[q];
              }
              const int mode_in = mode;
#ifdef SBM_SEQ_PROFILE6
              asm volatile("" ::: "memory");
              const long long tq6 = __builtin_readcyclecounter();
              prof_s += tq6 - tq6a;
#endif
#ifdef SBM_SEQ_PROFILE8
              const long long tq8 = __builtin_readcyclecounter();
              asm volatile("" ::: "memory");
#endif
              if (__builtin_amdgcn_ballot_w64(mode == MODE_ITER) == 0ull) {
                // every group still at work only wanted its matrices at the converged state: no update, no test
                if (with_sens && mode == MODE_FINAL) {
                  store_table(trow, tw);
                }
                mode = MODE_DONE;
                break;
              }
              if (mode == MODE_FINAL) mode = MODE_DONE;
              // along the 16 lanes of the group (one DPP row): inclusive prefix of the affine maps, then the value that
              // enters this lane = what leaves the lane before it
              double pa = Ap[RPG - 1], pb = Bp[RPG - 1];
              Stepper::template chain_level<0x111, 0xf>(pa, pb);    // row_shr:1
              Stepper::template chain_level<0x112, 0xf>(pa, pb);    // row_shr:2
              Stepper::template chain_level<0x114, 0xf>(pa, pb);    // row_shr:4
              Stepper::template chain_level<0x118, 0xf>(pa, pb);    // row_shr:8
              const double din = Stepper::template dpp_f64<0x111, 0xf>(pb, 0.0);
              // (update, test and decisions WITHOUT branches and divisions: at one wavefront per SIMD every exec-masked
              // region costs its scalar bookkeeping plus a branch bubble -- round 4's first version spent a dozen of them
              // and two IEEE divisions per pass here)
              float rmax = 0.f;
              const bool upd = mode == MODE_ITER;
#pragma unroll
              for (int r = 0; r < RPG; ++r) {
                const double d = fma(Ap[r], din, Bp[r]);
                yb[r] = upd ? yb[r] - d : yb[r];
                const float q = (float)fabs(d) * __builtin_amdgcn_rcpf((float)fmax(fma(nrtol, fabs(yb[r]), natol), 1e-30));
                const float qi = sbm_nan_to_inf(q);
                rmax = fmaxf(rmax, hasA[r] ? qi : 0.f);                   // (q formed unconditionally: a select, not a branch)
              }
              const float rr = sbm_row16_max(rmax);
              const float rp2 = r_prev * r_prev;
              const bool fin = rr < 3.0e38f;
              const bool conv = rr <= 1.0f;
              const bool rate = (it > 0) & (rr < 0.25f * r_prev) & (rr * rr * rr <= 0.1f * rp2);
              const bool stall = (it >= 2) & (rr >= 0.5f * r_prev) & (rr <= 1.0e3f);
              // One update may do.  Newton contracts quadratically, update_{k+1} ~ c update_k^2 (in tolerances), and c --
              // a property of the equations and the step size -- changes slowly along a trajectory: the second update of
              // an earlier Euler step of this group measured it (cest; refreshed in the first slot of every macro step,
              // where two updates are always made).  If c update_1^2 predicts a second update below a fortieth of the
              // tolerance, the iterate after the FIRST update is the converged state.
              const bool one = (it == 0) & (s > 0) & (cest * rr * rr <= 0.025f);
              const int next_mode = conv ? MODE_DONE : ((rate | stall | one) ? MODE_FINAL : MODE_ITER);
              const bool bad = upd & !fin;
              mode = (upd & fin) ? next_mode : mode;
              {
                const float c = rr * __builtin_amdgcn_rcpf(fmaxf(rp2, 1e-30f));
                const float cnew = cest < 3.0e38f ? fmaxf(c, 0.7f * cest) : c;
                cest = (upd & fin & (it == 1) & (r_prev > 1.0f)) ? cnew : cest;
              }
              r_prev = upd ? rr : r_prev;
              // without sensitivities nobody wants the matrices at the converged state
              mode = (!with_sens & (mode == MODE_FINAL)) ? MODE_DONE : mode;
              if (with_sens && mode_in != MODE_DONE && mode == MODE_DONE) {
                // the group has finished this Euler step: tw holds the matrices of its last evaluation
                store_table(trow, tw);
              }
#ifdef SBM_SEQ_PROFILE8
              asm volatile("" ::: "memory");
              prof_s += __builtin_readcyclecounter() - tq8;
#endif
              if (__builtin_amdgcn_ballot_w64(bad) != 0ull) { rc = SBM_NON_FINITE; break; }
              if (__builtin_amdgcn_ballot_w64(mode != MODE_DONE) == 0ull) break;
              if (it >= 8) { rc = SBM_NEWTON_FAIL; break; }      // eight updates, as newton_rate<8>
            }
            // a sequence that has made its last step leaves T_j; every group moves its history on (a group without work
            // carries values nobody reads: no select per row)
            if (act && m == jc - 1) {
#pragma unroll
              for (int r = 0; r < RPG; ++r) sh.TG[jc - 1][gl * RPG + r] = yb[r];
            }
#pragma unroll
            for (int r = 0; r < RPG; ++r) { yp3[r] = yp2[r]; yp2[r] = yp[r]; yp[r] = ya[r]; ya[r] = yb[r]; }
          }
          if (rc == SBM_OK) {
            // the two extrapolations on the row lanes
            Stepper::fence();
#pragma unroll 1
            for (int j = 1; j <= K; ++j) {
              const double d = sh.TG[j - 1][lane] - yn[0];
              yh[0] = fma(SBM_IEX_W.wh[K][j], d, yh[0]);
              ye[0] = fma(SBM_IEX_W.we[K][j], d, ye[0]);
            }
            Stepper::fence();
          }
        }

        // =============== phase B: the sensitivity columns, sequence by sequence ===============
#ifdef SBM_SEQ_PROFILE
        const long long tp1 = __builtin_readcyclecounter();
#endif
        double zh[NV], ze[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { zh[i] = 0.0; ze[i] = 0.0; }
        if (with_sens && rc == SBM_OK) {
          // phase A's tables were written by other lanes of this wavefront: make them visible to its loads
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          // row lane i: row i of every step's table = W doubles at lane block i / RPG, row i % RPG
          const int srow = st.has_row[0] ? lane : 0;
          const double* const src0 = tblock + (size_t)(srow / RPG) * LW + (size_t)(srow % RPG) * W;
         if constexpr (ROT) {
          // ---- rotated columns: no select, no wave-uniform table walk.  Per Euler step: row lane i puts (rd_i, cc_i) into RT,
          // every column lane fetches its own J_p entry; then z_0 = rd (z_0 + h a), z_k = rd z_k + cc z_{k-1} with the
          // coefficients of row r0 + k read at a per-lane address (consecutive lanes, consecutive 16-byte slots). ----
          // The ring: entry e = k mod NRING holds the table of Euler step k.  Loads are LDS-direct (global_load_lds_dwordx4:
          // lane l's 16 bytes land at M0 + 16 l, no register in between) and issued NRING - 1 steps ahead by inline
          // assembly with the waits written out -- a register ring under the compiler's own s_waitcnt placement
          // collapses to vmcnt(0) at the loop header, i.e. to a distance of one step (measured: 0.4 us of every 0.5 us
          // step spent waiting).  Two loads per step: RC by all lanes, A by lanes 0 .. 31.  Nothing else in this loop may
          // touch vector memory (the counts below assume it; tests/test_seq_kernel_isa.py reads the ISA).
          constexpr int NR = Pl::NRING, ENTRY = Pl::ROT_STEP_DOUBLES;
          const double colmask = has_col ? 1.0 : 0.0;
          const unsigned ring0 = (unsigned)(size_t)(&sh.RING[0]);
          auto issue = [&](int kk) {
            const int kc = kk < Pl::NSTEP ? kk : Pl::NSTEP - 1;
            const double* g = tblock + (size_t)kc * ENTRY + 2 * lane;
            const unsigned dst = ring0 + (unsigned)((kk % NR) * ENTRY * 8);
            // (the instruction offset moves the global AND the LDS address: A follows RC at byte 1024 on both sides)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                         "s_mov_b64 exec, 0xffffffff\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                         "s_mov_b64 exec, -1"
                         :: "v"(g), "s"(dst) : "memory");
          };
#pragma unroll 1
          for (int kk = 0; kk < NR - 1; ++kk) issue(kk);
          const int n_euler = K * (K + 1) / 2;
          int j = 1, m = 0;
          double h = Hs;
          double zs[NV];
#pragma unroll 1
          for (int kstep = 0; kstep < n_euler; ++kstep) {
            if (m == 0) {
#ifdef SBM_SEQ_PROFILE5
              const long long tq5 = __builtin_readcyclecounter();
              asm volatile("" ::: "memory");
#endif
#pragma unroll
              for (int i = 0; i < NV; ++i) zs[i] = sh.ZN[i * ZS + zl];
              h = Hs * sbm_iex_rj(j);
#ifdef SBM_SEQ_PROFILE5
              asm volatile("" ::: "memory");
              prof_s += __builtin_readcyclecounter() - tq5;
#endif
            }
#ifdef SBM_SEQ_PROFILE4
            const long long tq4 = __builtin_readcyclecounter();
            asm volatile("" ::: "memory");
#endif
            issue(kstep + NR - 1);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (NR - 1)) : "memory");
#ifdef SBM_SEQ_PROFILE4
            asm volatile("" ::: "memory");
            prof_s += __builtin_readcyclecounter() - tq4;
#endif
            const double* const ent = &sh.RING[(kstep % NR) * ENTRY];
            const double2* const rt = reinterpret_cast<const double2*>(ent) + r0;
            const double ha = (h * colmask) * ent[128 + (r0 & 63)];       // (no branch round the read: idle lanes multiply by zero)
#ifdef SBM_SEQ_PROFILE2
            const long long tq0 = __builtin_readcyclecounter();
            asm volatile("" ::: "memory");
#endif
            // blocks of BK rows, coefficients two blocks ahead of the arithmetic
            constexpr int BK = 5, NB = (NV + BK - 1) / BK;
            // (a block's rows are read LAST ROW FIRST: LDS answers in order, so the wait for the block's first row -- the last
            // one asked for -- covers the whole block: one s_waitcnt per block instead of one per row, at one instruction
            // per four cycles and wavefront)
            double2 tb[3][BK];
#pragma unroll
            for (int b = 0; b < 2 && b < NB; ++b) {
#pragma unroll
              for (int e = BK - 1; e >= 0; --e) tb[b][e] = rt[(b * BK + e) < NV ? b * BK + e : 0];
              Stepper::fence();
            }
            sbm_static_for<NB>([&](auto bc) {
              constexpr int b = decltype(bc)::value;
              Stepper::fence();
              if constexpr (b + 2 < NB) {
#pragma unroll
                for (int e = BK - 1; e >= 0; --e) tb[(b + 2) % 3][e] = rt[((b + 2) * BK + e) < NV ? (b + 2) * BK + e : 0];
                Stepper::fence();
              }
#pragma unroll
              for (int e = 0; e < BK; ++e) {
                const int k = b * BK + e;
                if (k < NV) {
                  const double2 t = tb[b % 3][e];
                  if (k == 0) zs[0] = t.x * (zs[0] + ha);
                  else zs[k] = fma(t.y, zs[k > 0 ? k - 1 : 0], t.x * zs[k]);
                }
              }
            });
            Stepper::fence();
#ifdef SBM_SEQ_PROFILE2
            asm volatile("" ::: "memory");
            prof_s += __builtin_readcyclecounter() - tq0;
#endif
            if (++m == j) {
#ifdef SBM_SEQ_PROFILE3
              const long long tq1 = __builtin_readcyclecounter();
              asm volatile("" ::: "memory");
#endif
              const double wh = SBM_IEX_W.wh[K][j], we = SBM_IEX_W.we[K][j];
#pragma unroll
              for (int i = 0; i < NV; ++i) {
                zh[i] = fma(wh, zs[i], zh[i]);
                ze[i] = fma(we, zs[i], ze[i]);
              }
              m = 0;
              ++j;
#ifdef SBM_SEQ_PROFILE3
              asm volatile("" ::: "memory");
              prof_s += __builtin_readcyclecounter() - tq1;
#endif
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's last (clamped) loads
         } else {
          // where the sub-diagonal entry of this lane's row goes in the factor table (row 0 has none: spare slot)
          int offpos = Stepper::MFSPARE;
          if constexpr (M::RL_MAXJY == 2) offpos = st.diagslot[0] == 0 ? st.mfpos[0][1] : st.mfpos[0][0];
          // The tables come from L2 / the memory-side cache (~1 us; one Euler step of a column is ~0.7 us): each is loaded
          // PF steps ahead into its own registers.  The steps of all sequences form ONE loop, unrolled PF times, so that
          // stage p of the ring is the same registers in every round (moving a stage on would wait for a load in flight).
          constexpr int PF = 2;
          double wn[PF][W];
#pragma unroll
          for (int p = 0; p < PF; ++p) {
#pragma unroll
            for (int q = 0; q < W; ++q) wn[p][q] = src0[(size_t)(p < Pl::NSTEP ? p : 0) * Pl::STEP_DOUBLES + q];
          }
          const int n_euler = K * (K + 1) / 2;
          int j = 1, m = 0;
          double h = Hs;
          double zs[NV];
#pragma unroll 1
          for (int k0 = 0; k0 < n_euler; k0 += PF) {
            sbm_static_for<PF>([&](auto pc) {
              constexpr int p = decltype(pc)::value;
              const int kstep = k0 + p;
              if (kstep < n_euler) {
                if (m == 0) {
#pragma unroll
                  for (int i = 0; i < NV; ++i) zs[i] = sh.ZN[i * ZS + zl];
                  h = Hs * sbm_iex_rj(j);
                }
                // this step's table into the LDS tables im_sens_tri reads (row lane i: row i); its registers take the
                // table of the step PF ahead
#ifndef SBM_SEQ_NO_STAGE
                if (st.has_row[0]) {
                  sh.MF[st.rdpos[0]] = wn[p][0];
                  sh.MF[offpos] = wn[p][1];
#pragma unroll
                  for (int q = 0; q < M::RL_MAXJP; ++q) sh.A[st.apos[0][q]] = wn[p][2 + q];
                }
                Stepper::fence();
                {
                  const int kn = kstep + PF < Pl::NSTEP ? kstep + PF : Pl::NSTEP - 1;
                  const double* src = src0 + (size_t)kn * Pl::STEP_DOUBLES;
#pragma unroll
                  for (int q = 0; q < W; ++q) wn[p][q] = src[q];
                }
#endif
#ifdef SBM_SEQ_PROFILE2
                const long long tq0 = __builtin_readcyclecounter();
                asm volatile("" ::: "memory");
#endif
#ifndef SBM_SEQ_NO_SENS
                st.sens_euler(h, zs);
#endif
#ifdef SBM_SEQ_PROFILE2
                asm volatile("" ::: "memory");
                prof_s += __builtin_readcyclecounter() - tq0;
#endif
                if (++m == j) {
                  // zh = sum_j wH_j T_j = T_KK itself (the weights add up to one), ze = sum_j (wH_j - wL_j) T_j (they add up to
                  // zero): no S_n in the sums -- with it (sbm_iex_kernel) every sequence ends in 25 dependent LDS round trips
                  // squeezed between the accumulators' AGPR moves, a fifth of phase B.  Price: the rounding of the sums is
                  // 3e3 eps relative to |S| instead of |T_j - S_n| -- 4e-13, against a tolerance of 1e-9.
                  const double wh = SBM_IEX_W.wh[K][j], we = SBM_IEX_W.we[K][j];
#pragma unroll
                  for (int i = 0; i < NV; ++i) {
                    zh[i] = fma(wh, zs[i], zh[i]);
                    ze[i] = fma(we, zs[i], ze[i]);
                  }
                  m = 0;
                  ++j;
                }
              }
            });
          }
         }
        }

#ifdef SBM_SEQ_PROFILE
        const long long tp2 = __builtin_readcyclecounter();
        prof_a += tp1 - tp0;
        prof_b += tp2 - tp1;
#endif
        // =============== error estimate and step-size control (as sbm_iex_kernel) ===============
        float err = __builtin_inff();
        double colmax_new = colmax;
        if (rc == SBM_OK) {
          float cs = 0.f;
          if (with_sens) {
#pragma unroll
            for (int i = 0; i < NV; ++i) colmax_new = fmax(colmax_new, fabs(zh[i]));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              const double tk = zh[i];
              const double sc = fmax(fma(rtol, fmax(fabs(tk), floor_rel * colmax_new), atol), 1e-30);
              const float r = (float)ze[i] * __builtin_amdgcn_rcpf((float)sc);
              cs = fmaf(r, r, cs);
            }
          }
          const float ykmax = sbm_wave_max(sbm_nan_to_inf(st.has_row[0] ? (float)fabs(yn[0] + yh[0]) : 0.f));
          const double yk = yn[0] + yh[0];
          const double scy = fmax(fma(rtol, fmax(fabs(yk), floor_rel * (double)ykmax), atol), 1e-30);
          const float ry = st.has_row[0] ? (float)ye[0] * __builtin_amdgcn_rcpf((float)scy) : 0.f;
          const float xs = sbm_wave_sumf(sbm_nan_to_inf(ry * ry));
          const float mx = sbm_wave_max(has_col ? sbm_nan_to_inf(cs) : 0.f);
          err = sqrtf(fmaxf(mx, xs) * (1.0f / NV));
          if (!(err == err)) err = __builtin_inff();
        }
        if (err <= 1.0f) {
          ydot[0] = yh[0] / Hs;
          yn[0] += yh[0];
          if (with_sens) {
#pragma unroll
            for (int i = 0; i < NV; ++i) sh.ZN[i * ZS + zl] = zh[i];
            Stepper::fence();
          }
          colmax = colmax_new;
          t = landing ? target : t + Hs;
          ++n_acc;
          float fac = err > 1e-12f ? 0.9f * __powf(err, expo) : 4.0f;
          fac = fminf(after_reject ? 1.0f : 4.0f, fmaxf(0.2f, fac));
          if (!landing || fac < 1.0f || Hs * (double)fac > H) H = Hs * (double)fac;
          after_reject = false;
        } else {
          ++n_rej;
          float fac = 0.25f;
          if (rc == SBM_OK && err < 3.0e38f) fac = fminf(0.9f, fmaxf(0.1f, 0.9f * __powf(err, expo)));
          H = Hs * (double)fac;
          after_reject = true;
        }
      }
      const bool failed = status != SBM_OK;
      if (Yt && chunk == 0) {
        if (st.has_row[0]) Yt[(size_t)io * NV + lane] = failed ? __builtin_nan("") : yn[0];
      }
      if (St && has_col) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int row = ROT ? (r0 + i < NV ? r0 + i : r0 + i - NV) : i;
          St[((size_t)io * NV + row) * NK + col] = failed ? __builtin_nan("") : sh.ZN[i * ZS + zl];
        }
      }
    }
#ifdef SBM_SEQ_PROFILE      // developer build: kilocycles of phase A in n_steps, of phase B in n_reject, of the trajectory in status
    n_acc = prof_a >> 10;
    n_rej = prof_b >> 10;
#if defined(SBM_SEQ_PROFILE2) || defined(SBM_SEQ_PROFILE3) || defined(SBM_SEQ_PROFILE4) || defined(SBM_SEQ_PROFILE5) || defined(SBM_SEQ_PROFILE6) || defined(SBM_SEQ_PROFILE7) || defined(SBM_SEQ_PROFILE8)
    n_acc = prof_s >> 10;      // the column step (2) / the accumulation (3) alone instead of phase A
#endif
    status = (int)((__builtin_readcyclecounter() - prof_t0) >> 10);
#endif
    if (lane == 0) {
      const int na = (int)(n_acc > 2000000000LL ? 2000000000LL : n_acc), nr = (int)(n_rej > 2000000000LL ? 2000000000LL : n_rej);
      if (nch_launch > 1) {
        if (a.status) atomicMax(a.status + traj, status);
        if (a.n_steps) atomicMax(a.n_steps + traj, na);
        if (a.n_reject) atomicMax(a.n_reject + traj, nr);
      } else {
        if (a.status) a.status[traj] = status;
        if (a.n_steps) a.n_steps[traj] = na;
        if (a.n_reject) a.n_reject[traj] = nr;
      }
    }
    Stepper::fence();
  }
}
