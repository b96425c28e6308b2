// sbm_sens_mfma.hpp -- the sensitivity right-hand side  dS = J_y S + J_p  on the matrix cores
// (v_mfma_f64_16x16x4_f64), SBM_VARIANT_MFMA.
//
// BASELINE.json asks for the dense Jacobian x S product on MFMA "only if ... dense enough to pay", SURVEY.md section
// 8(d) for a dense variant of the 20-state model to find out.  This is that path: one trajectory per wavefront as in
// the other sensitivity kernels, the state one component per lane, f / J_y / J_p evaluated by the row lanes class by
// class -- and the product as 16x16x4 f64 tiles instead of generated scalar code over the non-zeros:
//
//   * S lives in the accumulator layout of the instruction: tile (rt, ct) = rows [16 rt, 16 rt + 16) x columns
//     [16 ct, 16 ct + 16) is four doubles per lane, register r of lane l holding row 16 rt + (l >> 4) + 4 r, column
//     16 ct + (l & 15).  That layout is at the same time the B-operand layout of the next product: k-step s of a tile
//     wants B[k = 4 s + (l >> 4)][n = l & 15], i.e. register s.  So a Runge-Kutta stage vector goes into the product
//     exactly as it stands -- no transposition, no LDS round trip for S;
//   * J_y is handed over as a dense 16 RT x 16 RT tile image in LDS, stored COLUMN-major (JD[k][i], column stride LDJ ==
//     16 mod 32 doubles): row lane i drops its non-zeros at [column][i] (the rest stays zero), and every lane fetches its
//     A operands A[i = l & 15][k = 4 s + (l >> 4)] from there -- 4 RT^2 loads of 8 bytes per stage, reused for all column
//     tiles.  The banking rules (MI355X_MICROARCH.md, LDS): a ds_read_b64 is served in the lane groups 0-31 / 32-63 on
//     banks (a / 4) mod 64 -- the 32 lanes (l & 15, l >> 4 in {0, 1}) read doubles (l >> 4) LDJ + (l & 15) + const, i.e.
//     {0..15} and {16..31} mod 32: conflict-free; a ds_write_b64 in groups of 16 consecutive lanes on banks (a / 4) mod 32
//     -- the 16 row lanes i write doubles c_s(i) LDJ + i, i.e. i mod 16 whatever their columns: conflict-free.  (Round 3
//     stored the image row-major with a row stride of MP + 2: reads conflict-free, but the row lanes i and i + 8 of a
//     dense J_y -- slot s is the same column in every row -- wrote the same bank pair: a third of the kernel's LDS cycles
//     were conflicts, profiles/r03/dense_pmc_summary.json.)
//   * J_p is the C operand: the accumulators start from the dense [row][column] image the row lanes fill.
//
// Cost per stage: 4 RT^2 CT MFMAs of 64 cycles whatever the sparsity of J_y -- cascade20 (RT = 2, CT = 3): 48 MFMAs =
// 3072 cycles on the matrix pipe for 2 x 20 x 20 x 40 = 32 000 useful flops (a third of the padded tile work).  The
// scalar path costs 2 FMAs per non-zero per column.  On MI355X the f64 matrix rate equals the f64 vector rate
// (MI355X_MICROARCH.md), so the matrix cores can only win by relieving the VALU issue port (the two pipes run
// concurrently) and by saving the operand traffic of the scalar form; DESIGN.md records where the crossover is.
#pragma once

typedef double sbm_v4d __attribute__((ext_vector_type(4)));

template <class M, int CT>
struct SbmMfmaShared {
  static constexpr int RT = (M::NV + 15) / 16;
  static constexpr int MP = 16 * RT;               // padded rows (= padded k range)
  static constexpr int LDJ = (MP % 32 == 16) ? MP : MP + 16;   // COLUMN stride of the J_y image JD[k][i]: == 16 mod 32
  static constexpr int NC = 16 * CT;               // columns of this wavefront's chunk
  static constexpr int LDA = (NC % 32 == 16) ? NC : NC + 16;   // row stride of the J_p image: == 16 mod 32
  double Y[64];
  alignas(16) double JD[MP * LDJ + 64];            // J_y, dense, zero where structurally zero (+ one spare slot per lane:
                                                   //  lanes without a row / slot must not all write ONE address)
  alignas(16) double A[MP * LDA + 64];             // J_p columns of this chunk (+ one spare slot per lane)
};

template <class M, int CT>
struct MfmaSystem {
  using Sh = SbmMfmaShared<M, CT>;
  static constexpr int RT = Sh::RT;
  static constexpr int NV = 4 * RT * CT;           // elements of S per lane
  static constexpr int NVX = NV + 1;               // + this lane's state component
  static constexpr int CPL = 1;
  static constexpr int NCS = CT;                   // one error sum per column tile
  // element e = (rt * CT + ct) * 4 + r
  __device__ __forceinline__ static constexpr int col_of(int, int e) { return (e / 4) % CT; }
  Sh* sh;
  int lane, cbase, cls;
  int yidx[M::RL_MAXYS], jdpos[M::RL_MAXJY], apos[M::RL_MAXJP];
  double ps[M::RL_MAXPS];

  __device__ __forceinline__ static void lds_order() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }
  struct Pending { double ys[M::RL_MAXYS]; };
  struct Token { double f; };

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
    for (int s = 0; s < M::RL_MAXJY; ++s) sh->JD[jdpos[s]] = jy[s];
#pragma unroll
    for (int s = 0; s < M::RL_MAXJP; ++s) sh->A[apos[s]] = jp[s];
    lds_order();
    return k;
  }
  __device__ __forceinline__ void extra_out(const Token& k, double (&dz)[1][NVX]) const { dz[0][NV] = k.f; }
  // dS = J_y S + J_p, tile by tile
  __device__ __forceinline__ void finish(const Token&, double, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    const int lr = lane & 15, lq = lane >> 4;
    double aop[RT][RT][4];       // A operands: J_y[16 rt + lr][16 kt + 4 s + lq]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int kt = 0; kt < RT; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) aop[rt][kt][s] = sh->JD[(16 * kt + 4 * s + lq) * Sh::LDJ + 16 * rt + lr];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        sbm_v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = sh->A[(16 * rt + lq + 4 * r) * Sh::LDA + 16 * ct + lr];
#pragma unroll
        for (int kt = 0; kt < RT; ++kt)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[rt][kt][s], z[0][(kt * CT + ct) * 4 + s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[0][(rt * CT + ct) * 4 + r] = acc[r];
      }
    }
    lds_order();
  }
  __device__ __forceinline__ void rhs(double t, const double (&z)[1][NVX], double (&dz)[1][NVX]) const {
    const Token k = eval(issue(t, z), t);
    extra_out(k, dz);
    finish(k, t, z, dz);
  }
  // max( RMS of the state error, max over columns of the column RMS ); a column's rows sit in the lanes l & 15 = const
  __device__ __forceinline__ float norm(const float (&colsum)[NCS], float xsum) const {
    float m = 0.f;
#pragma unroll
    for (int cc = 0; cc < NCS; ++cc) {
      float v = sbm_nan_to_inf(colsum[cc]);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const bool has_col = cbase + 16 * cc + (lane & 15) < M::NK;
      m = fmaxf(m, has_col ? v : 0.f);
    }
    const float x = (lane < M::NV) ? sbm_nan_to_inf(xsum) : 0.f;
    const float mx = sbm_wave_max(m);
    const float xs = sbm_wave_sumf(x);
    return sqrtf(fmaxf(mx, xs) * (1.0f / M::NV));
  }
  __device__ __forceinline__ double sum(double v) const { return sbm_wave_sum(v); }
};

// columns per wavefront: as many 16-column tiles as keep the lane's share of S at 24 elements or fewer
template <class M>
struct SbmMfmaPlan {
  static constexpr int RT = (M::NV + 15) / 16;
  static constexpr int CT_ALL = (M::NK + 15) / 16;
// Elements of S per lane.  Round 2 planned up to 24 (cascade20: one wavefront per trajectory, 24 elements x 7 stage vectors:
// all 512 registers, 460 B of scratch, one wavefront per SIMD, VALU active 32 %); with at most 12 (16 columns per
// wavefront, three wavefronts per trajectory, each repeating the state evaluation) the kernel needs 254 registers, no
// scratch, and TWO wavefronts share a SIMD -- the matrix pipe of one runs under the Runge-Kutta combinations of the other.
// Measured (4096 vectors, DOPRI45, ms per launch, 24 -> 12): dense20 23.6 -> 13.4, half density 17.2 -> 12.1, quarter 14.9 ->
// 11.5, cascade20 27.1 -> 21.7 (profiles/r03/dense_*).
#ifndef SBM_MFMA_MAX_EL
#define SBM_MFMA_MAX_EL 12
#endif
  static constexpr int CT_FIT = (SBM_MFMA_MAX_EL / (4 * RT)) > 0 ? (SBM_MFMA_MAX_EL / (4 * RT)) : 1;
  static constexpr int CT = CT_ALL < CT_FIT ? CT_ALL : CT_FIT;
  static constexpr int NCH = (M::NK + 16 * CT - 1) / (16 * CT);
  // seven stage vectors of 4 RT CT elements + the A operands: up to 12 elements per lane fit 256 registers
  static constexpr int MIN_WAVES = (4 * RT * CT <= 12) ? 2 : 1;
};

template <class M, int METHOD>
__global__ void __launch_bounds__(64, SbmMfmaPlan<M>::MIN_WAVES) sbm_sens_mfma_kernel(sbm_kernel_args a) {
  constexpr int CT = SbmMfmaPlan<M>::CT, NCH = SbmMfmaPlan<M>::NCH;
  using Sys = MfmaSystem<M, CT>;
  using Sh = SbmMfmaShared<M, CT>;
  constexpr int MNV = M::NV, NK = M::NK, RT = Sys::RT, NE = Sys::NV, NVX = Sys::NVX;
  static_assert(MNV <= 64, "MFMA sensitivity kernel: one state row per lane");
  __shared__ Sh sh;
  if ((int)blockIdx.x >= a.n_traj) return;
  const int traj = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x;
  const int chunk = NCH > 1 ? (int)blockIdx.y : 0;
  const int cbase = chunk * 16 * CT;
  for (int i = lane; i < Sh::MP * Sh::LDJ + 64; i += 64) sh.JD[i] = 0.0;
  for (int i = lane; i < Sh::MP * Sh::LDA + 64; i += 64) sh.A[i] = 0.0;
  sh.Y[lane] = 0.0;

  Sys sys;
  sys.sh = &sh;
  sys.lane = lane;
  sys.cbase = cbase;
  const bool has_row = lane < MNV;
  const int row = has_row ? lane : 0;
  sys.cls = has_row ? M::rl_class(row) : -1;
  const double* P = a.P + (size_t)traj * M::NP;
#pragma unroll
  for (int s = 0; s < M::RL_MAXYS; ++s) sys.yidx[s] = M::rl_ys(s, row);
#pragma unroll
  for (int s = 0; s < M::RL_MAXPS; ++s) sys.ps[s] = P[M::rl_ps(s, row)];
#pragma unroll
  for (int s = 0; s < M::RL_MAXJY; ++s) {
    const int c = M::rl_jycol(s, row);
    sys.jdpos[s] = (has_row && c >= 0) ? c * Sh::LDJ + row : Sh::MP * Sh::LDJ + lane;   // [column][row]; else: this lane's spare slot
  }
#pragma unroll
  for (int s = 0; s < M::RL_MAXJP; ++s) {
    const int lc = M::rl_jpcol(s, row) - cbase;
    sys.apos[s] = (has_row && lc >= 0 && lc < 16 * CT && lc + cbase < NK) ? row * Sh::LDA + lc : Sh::MP * Sh::LDA + lane;
  }
  __syncthreads();

  const int goff = a.grid_off ? a.grid_off[traj] : 0;
  const int glen = a.grid_len ? a.grid_len[traj] : a.n_t;
  const double* tg = a.t_out + goff;
  const int lr = lane & 15, lq = lane >> 4;

  // element (rt, ct, r) of this lane = S[16 rt + lq + 4 r][cbase + 16 ct + lr]
  double z[1][NVX];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int grow = 16 * rt + lq + 4 * r, col = cbase + 16 * ct + lr;
        z[0][(rt * CT + ct) * 4 + r] = (a.s0 && grow < MNV && col < NK) ? a.s0[grow * NK + col] : 0.0;
      }
  z[0][NE] = (a.y0 && has_row) ? a.y0[lane] : 0.0;

  double* Yt = a.Y ? a.Y + (size_t)traj * a.n_t * MNV : nullptr;
  double* St = a.S ? a.S + (size_t)traj * a.n_t * MNV * NK : nullptr;
  auto store = [&](int io, const double (&zz)[1][NVX]) {
    if (Yt && chunk == 0 && has_row) Yt[(size_t)io * MNV + lane] = zz[0][NE];
    if (St) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int grow = 16 * rt + lq + 4 * r, col = cbase + 16 * ct + lr;
            if (grow < MNV && col < NK) St[((size_t)io * MNV + grow) * NK + col] = zz[0][(rt * CT + ct) * 4 + r];
          }
    }
  };

  SbmTrajOut r = sbm_integrate<METHOD>(sys, z, tg, glen, a.opts, store);

  if (lane == 0) {
    if constexpr (NCH > 1) {
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
