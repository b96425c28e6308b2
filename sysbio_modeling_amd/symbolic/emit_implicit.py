"""Sparse direct solver for the Newton matrix of implicit integrators, M = I - gamma * J_y.

Generalises the analytic-ODE-Jacobian path of the reference (``model_jac`` / ``Dfun`` handed to
LSODA, model/ode_model.py:114-120,154-160; symbolic/sympy_tools.py:149-159,219-269): instead of
returning a dense Jacobian to a library solver, the emitter factors the matrix itself.  The
sparsity pattern of J_y is known when the model is generated, so the LU factorisation (natural
order, no pivoting: M is I minus a small multiple of J) is worked out SYMBOLICALLY here -- fill-in
included -- and printed as straight-line code with static indices:

  im_build (gamma, jy[], m[])   m = I - gamma*J_y on the filled pattern
  im_factor(m[])                in-place LU; the diagonal ends up holding the RECIPROCAL pivots
  im_solve (m[], b[])           b <- M^-1 b   (forward, then backward substitution)

For a cascade (J_y lower bidiagonal) this is 2n-1 entries, no fill and no backward pass; a dense
n x n solve would be n^3/3 flops per factorisation.  Every lane of a wavefront runs the same code
on the same (wave-uniform) matrix and its own right-hand side (one sensitivity column each).

General patterns on up to 64 state variables get a form DISTRIBUTED OVER ROWS (IM_DIST): the redundant form above costs
every lane the whole elimination (n^3/3 multiply-adds for a dense pattern) and keeps all IM_NM factors in registers of
every lane (twice IM_NM VGPRs: a dense 20-state model would need 800).  In the distributed form lane i holds row i of
M as a dense register row; at pivot k the lanes below scale their entry and subtract the pivot row, which reaches them
as scalar operands through ``v_readlane`` (static lane, static register) -- per lane one multiply-add per entry of U
instead of one per elimination update (dense: n^2/2 instead of n^3/3).  The finished rows and the reciprocal pivots are
published in LDS (``im_factor_rows``) and the substitutions read them back with wave-uniform addresses
(``im_solve_lds``), as the triangular form does.

Lower-triangular patterns (feed-forward networks) need no elimination at all: the pivots are the
diagonal entries.  For them the emitter adds a DISTRIBUTED form (IM_TRI): row lane i, which has just
evaluated row i of J_y, computes its own reciprocal pivot and scaled off-diagonal entries -- one
reciprocal per lane instead of n per lane -- and publishes them in the table MF; ``im_solve_tri``
then is the forward substitution reading MF (wave-uniform LDS reads).
"""
from __future__ import annotations


def symbolic_lu(n, entries):
    """entries: iterable of (i, j) non-zeros of J_y.  Returns (pattern, ops):
    pattern: sorted list of (i, j) of L+U incl. the diagonal and fill-in;
    ops:     elimination schedule [(k, [(i, [j...])...])]: for pivot k, rows i > k with (i,k) in the
             pattern, and for each the columns j > k with (k,j) in the pattern (targets (i,j))."""
    rows = [set() for _ in range(n)]
    for i, j in entries:
        rows[i].add(j)
    for i in range(n):
        rows[i].add(i)
    ops = []
    for k in range(n):
        upper = sorted(j for j in rows[k] if j > k)
        below = [i for i in range(k + 1, n) if k in rows[i]]
        step = []
        for i in below:
            for j in upper:
                rows[i].add(j)          # fill-in
            step.append((i, upper))
        ops.append((k, step))
    pattern = sorted((i, j) for i in range(n) for j in rows[i])
    return pattern, ops


def lds_leading_dimension(n):
    """Row stride (in doubles) of the n x n factor table in LDS: the smallest LD >= n with LD / 2 ODD.  The lanes read their
    own rows 16 bytes at a time (ds_read_b128: lane groups of 16 on banks (a / 4) mod 64): lane l's address is 2 LD l dwords,
    i.e. slot (LD / 2) l mod 16 of the 256-byte bank row -- distinct for the 16 lanes of a group exactly when LD / 2 is odd.
    (Round 3 rounded up to 2 mod 32 -- 66 for 48 states, 25 KB where 19 do: with the rest of the extrapolation kernel's
    tables that was 55 KB of LDS per wavefront, two wavefronts per CU instead of four.)"""
    if n <= 2:
        return 2
    ld = n + (n & 1)
    if (ld // 2) % 2 == 0:
        ld += 2
    return ld


def emit_distributed(spec, d, pattern, ops):
    """IM_DIST members (see the module docstring).  ``pattern`` / ``ops``: symbolic_lu's."""
    n = spec.n_vars
    ld = lds_leading_dimension(n)
    lower_only = all(j <= i for i, j in pattern)
    # When it pays (measured, 1024 vectors x 512 fixed steps with sensitivities, scripts/dev_implicit_lu.py): the dense
    # 20-state network (400 non-zeros, 2 470 elimination updates) 256 ms redundant -> 47 ms distributed, half density
    # 148 -> 38 ms; cascade20 (bidiagonal + feedback corner: 37 updates) 3.9 -> 5.3 ms -- the distributed form
    # handles dense rows whatever the pattern.  Chosen when the redundant elimination is long or its factors would
    # not fit the register file.
    n_updates = sum(len(upper) for _, step in ops for _, upper in step)
    dist = (not lower_only) and n <= 64 and (n_updates > 8 * n or len(pattern) > 128)
    L = ["  // ---- the same factorisation distributed over rows: lane i factors row i (emit_implicit.py); %d elimination"
         % n_updates,
         "  //      updates in the redundant form ----",
         "  static constexpr bool IM_DIST = %s;" % ("true" if dist else "false"),
         "  static constexpr int IM_LD = %d;     // row stride of the factor table in LDS" % ld]
    if not dist:
        L += ["  __device__ __forceinline__ static void im_factor_rows(double (&)[NV], int, double*) {}",
              "  __device__ __forceinline__ static void im_solve_lds(const double*, const double*, double (&)[NV]) {}"]
        return L
    rows = [[j for (i, j) in pattern if i == r] for r in range(n)]
    L += ["  // m: this lane's row of M = I - gamma J_y (dense, zeros outside the pattern).  On exit: multipliers left of",
          "  // the diagonal, U from the diagonal on; rd[k] = 1 / pivot k (LDS, every lane writes the same value).",
          "  __device__ __forceinline__ static void im_factor_rows(double (&m)[NV], int lane, double* rd) {"]
    for k, step in ops:
        upper = sorted(j for j in rows[k] if j > k)
        L.append("    { const double rp = SBM_LANE_BCAST(SBM_RCP(m[%d]), %d); rd[%d] = rp;" % (k, k, k))
        if step:
            L.append("      const double l = lane > %d ? m[%d] * rp : 0.0; m[%d] = lane > %d ? l : m[%d];" % (k, k, k, k, k))
            for j in upper:
                L.append("      m[%d] = fma(-l, SBM_LANE_BCAST(m[%d], %d), m[%d]);" % (j, j, k, j))
        L.append("    }")
    L += ["  }",
          "  // b <- M^-1 b with the rows im_factor_rows produced, published at mf[row * IM_LD + column]",
          "  __device__ __forceinline__ static void im_solve_lds(const double* mf, const double* rd, double (&b)[NV]) {"]
    cnt = 0
    for i in range(n):
        for c in rows[i]:
            if c < i:
                L.append("    b[%d] = fma(-mf[%d], b[%d], b[%d]);" % (i, i * ld + c, c, i))
                cnt += 1
        if cnt >= 24:
            L.append("    SBM_LDS_FENCE();")
            cnt = 0
    for i in range(n - 1, -1, -1):
        for c in rows[i]:
            if c > i:
                L.append("    b[%d] = fma(-mf[%d], b[%d], b[%d]);" % (i, i * ld + c, c, i))
                cnt += 1
        L.append("    b[%d] *= rd[%d];" % (i, i))
        if cnt >= 24:
            L.append("    SBM_LDS_FENCE();")
            cnt = 0
    L += ["  }"]
    return L


def emit_members(spec, d):
    n = spec.n_vars
    jy_pos = {(r, c): e for e, (r, c, _) in enumerate(d.jy)}
    pattern, ops = symbolic_lu(n, jy_pos.keys())
    idx = {rc: k for k, rc in enumerate(pattern)}
    nm = len(pattern)
    n_fill = sum(1 for rc in pattern if rc not in jy_pos and rc[0] != rc[1])
    lower_only = all(j <= i for i, j in pattern)
    L = ["  // ---- Newton matrix of implicit integrators: M = I - gamma*J_y, sparse LU worked out at generation",
         "  //      time (emit_implicit.py): %d entries (%d of J_y, %d fill-in)%s ----"
         % (nm, len(jy_pos), n_fill, ", lower triangular: no elimination, no backward pass" if lower_only else ""),
         "  static constexpr int IM_NM = %d;" % nm,
         "  __device__ __forceinline__ static void im_build(double gamma, const double* jy, double (&m)[IM_NM]) {",
         "    (void)gamma; (void)jy;"]
    for k, (i, j) in enumerate(pattern):
        if (i, j) in jy_pos:
            e = jy_pos[(i, j)]
            L.append("    m[%d] = %s;" % (k, ("fma(-gamma, jy[%d], 1.0)" % e) if i == j else ("-gamma * jy[%d]" % e)))
        else:
            L.append("    m[%d] = %s;" % (k, "1.0" if i == j else "0.0"))
    L += ["  }",
          "  // in-place LU, natural order; m[diag] <- 1/pivot, strictly lower part <- multipliers",
          "  __device__ __forceinline__ static void im_factor(double (&m)[IM_NM]) {"]
    for k, step in ops:
        L.append("    m[%d] = SBM_RCP(m[%d]);" % (idx[(k, k)], idx[(k, k)]))
        for i, upper in step:
            L.append("    m[%d] *= m[%d];" % (idx[(i, k)], idx[(k, k)]))
            for j in upper:
                L.append("    m[%d] = fma(-m[%d], m[%d], m[%d]);" % (idx[(i, j)], idx[(i, k)], idx[(k, j)], idx[(i, j)]))
    L += ["  }",
          "  // b <- M^-1 b with the factors of im_factor",
          "  __device__ __forceinline__ static void im_solve(const double (&m)[IM_NM], double (&b)[NV]) {"]
    for i in range(n):      # L y = b (unit lower)
        for (r, c) in pattern:
            if r == i and c < i:
                L.append("    b[%d] = fma(-m[%d], b[%d], b[%d]);" % (i, idx[(r, c)], c, i))
    for i in range(n - 1, -1, -1):   # U x = y
        for (r, c) in pattern:
            if r == i and c > i:
                L.append("    b[%d] = fma(-m[%d], b[%d], b[%d]);" % (i, idx[(r, c)], c, i))
        L.append("    b[%d] *= m[%d];" % (i, idx[(i, i)]))
    L += ["  }"]
    # ---- distributed form for lower-triangular patterns ----
    L += ["  static constexpr bool IM_TRI = %s;" % ("true" if lower_only else "false")]
    if lower_only:
        # every row's block [1 / M_ii, scaled entries...] starts at an EVEN index of the (16-byte aligned) table: the
        # substitution reads a row's values in pairs, and a pair that straddles a 16-byte boundary costs a half-rate
        # ds_read2_b64 instead of a ds_read_b128 (measured on stiff50: 209 -> 275 ms per pass of configs[4])
        rstart, pos = [], {}
        k = 0
        for i in range(n):
            k += k & 1
            rstart.append(k)
            k += 1
            for (r, c) in pattern:
                if r == i and c < i:
                    pos[(r, c)] = k
                    k += 1
        n_table = k
        L += ["  // MF layout: per row [1/M_ii, then gamma*J_ij/M_ii for j < i in column order]; b <- M^-1 b",
              "  __device__ __forceinline__ static void im_solve_tri(const double* mf, double (&b)[NV]) {"]
        for i in range(n):
            expr = "mf[%d] * b[%d]" % (rstart[i], i)
            for (r, c) in pattern:
                if r == i and c < i:
                    expr = "fma(mf[%d], b[%d], %s)" % (pos[(r, c)], c, expr)
            L.append("    b[%d] = %s;" % (i, expr))
            if i % 8 == 7 and i + 1 < n:
                # keep the compiler from hoisting every table load to the top (2*IM_NM live registers)
                L.append("    SBM_LDS_FENCE();")
        L += ["  }"]
        # The Newton update of the STATE: the right-hand side is wave-uniform (every lane solves the same system) and a
        # lane keeps only the components of its own rows.  Fused form: g_i is read when row i is due, x_i is handed to
        # its lane (row i lives on lane i mod 64, slot i / 64) and stays in a register only while later rows refer to
        # it -- the plain form keeps all NV values alive up to the final pick, NV register pairs that the
        # extrapolation kernel needs for its running sums (sbm_implicit_extrap.hpp).
        last_use = {}
        for (r, c) in pattern:
            if c < r:
                last_use[c] = max(last_use.get(c, c), r)
        L += ["  // x = M^-1 g for the wave-uniform g (LDS); d[i / 64] of lane i mod 64 <- x_i",
              "  template <int RPL>",
              "  __device__ __forceinline__ static void im_solve_tri_pick(const double* mf, const double* g, int lane, double (&d)[RPL]) {"]
        for i in range(n):
            expr = "mf[%d] * g[%d]" % (rstart[i], i)
            for (r, c) in pattern:
                if r == i and c < i:
                    expr = "fma(mf[%d], x_%d, %s)" % (pos[(r, c)], c, expr)
            L.append("    const double x_%d = %s;" % (i, expr))
            L.append("    d[%d] = SBM_SEL(lane == %d, x_%d, d[%d]);" % (i // 64, i % 64, i, i // 64))
            if i % 8 == 7 and i + 1 < n:
                L.append("    SBM_LDS_FENCE();")
        L += ["  }"]
        # The sensitivity step of the extrapolation kernel, fused: z <- M^-1 (z + hh J_p[:, column of this lane]).  Row by
        # row the J_p pick (independent of the chain) sits next to the substitution (dependent on the previous row), and
        # the table entries of the NEXT block of rows are loaded before the arithmetic of the current one: written as two
        # passes (all picks, then im_solve_tri with its fences) the substitution had three ds_read_b128 in flight and
        # every block of eight rows began by waiting out a full LDS round trip (ISA of stiff50, round 3).
        blocks, cur, width = [], [], 0
        for i in range(n):
            w = 1 + sum(1 for (r, c) in pattern if r == i and c < i)
            w += w & 1
            if cur and width + w > 8:
                blocks.append(cur)
                cur, width = [], 0
            cur.append(i)
            width += w
        blocks.append(cur)
        row_end = lambda i: (rstart[i + 1] if i + 1 < n else n_table)     # noqa: E731

        def loads(b):
            rows_b = blocks[b]
            lo_e, hi_e = rstart[rows_b[0]], row_end(rows_b[-1])
            return ["    double t%d[%d], a%d[%d * IM_JP];  // rows %d..%d" % (b, hi_e - lo_e, b, len(rows_b), rows_b[0], rows_b[-1]),
                    "    _Pragma(\"unroll\") for (int e = 0; e < %d; ++e) t%d[e] = mf[%d + e];" % (hi_e - lo_e, b, lo_e),
                    "    _Pragma(\"unroll\") for (int e = 0; e < %d * RL_MAXJP; ++e) a%d[e] = ja[%d * RL_MAXJP + e];"
                    % (len(rows_b), b, rows_b[0])]

        def pick(b, i):
            # The pick of J_p[i][column of this lane]: SBM_PICK_COL -- a v_cndmask pair under a lane mask that SCALAR
            # instructions make from the literal column index (s_mov / s_cselect: inverse ballot), where `col == c`
            # costs a vector compare per non-zero on top (round 4: 50 of 300 vector instructions per step of stiff50).
            # (tried in round 3: the pick as ONE v_fmac under a one-lane EXEC set from literals by scalar instructions
            # -- no gain: every EXEC write stalls the vector instruction behind it)
            rows_b = blocks[b]
            expr = "0.0"
            for q, (_, c) in reversed(list(enumerate(d.jp_rows[i]))):
                expr = "SBM_PICK_COL(col, %d, a%d[%d * RL_MAXJP + %d], %s)" % (c, b, i - rows_b[0], q, expr)
            return ["    z[%d] = fma(hh, %s, z[%d]);" % (i, expr, i)] if d.jp_rows[i] else []
        L += ["  // z <- M^-1 (z + hh * J_p[:, column of this lane]) with the J_p table `ja` ([row][RL_MAXJP], columns rl_jpcol), the",
              "  // factors `mf`; col = the sensitivity column of this lane",
              "  static constexpr bool IM_SENS_TRI = true;",
              "  static constexpr int IM_JP = RL_MAXJP > 0 ? RL_MAXJP : 1;",
              "  __device__ __forceinline__ static void im_sens_tri(const double* mf, const double* ja, double hh, int col, double (&z)[NV]) {\n    (void)col;"]
        # (two blocks ahead: one block of four rows is ~100 cycles of arithmetic, a ds_read_b128 comes back after ~130 --
        # one block ahead left a quarter of every block waiting at one wavefront per SIMD)
        L += loads(0)
        if len(blocks) > 1:
            L += loads(1)
        for b, rows_b in enumerate(blocks):
            L.append("    SBM_LDS_FENCE();")
            if b + 2 < len(blocks):
                L += loads(b + 2)
            base = rstart[rows_b[0]]
            for i in rows_b:
                L += pick(b, i)
                expr = "t%d[%d] * z[%d]" % (b, rstart[i] - base, i)
                for (r, c) in pattern:
                    if r == i and c < i:
                        expr = "fma(t%d[%d], z[%d], %s)" % (b, pos[(r, c)] - base, c, expr)
                L.append("    z[%d] = %s;" % (i, expr))
        L += ["  }"]
        L += ["  static constexpr int IM_MF = %d;     // entries of the table (rows padded to even starts)" % n_table]
        # a CHAIN (row i refers to row i - 1 only: a cascade): the Newton update of the state is a first-order linear
        # recurrence x_i = b_i + a_i x_{i-1} over the row lanes -- a parallel prefix (sbm_implicit_stepper.hpp)
        chain = n <= 64 and all(c == r or c == r - 1 for (r, c) in pattern)
        L += ["  static constexpr bool IM_CHAIN = %s;" % ("true" if chain else "false")]
        # A chain whose sensitivity columns each have ONE non-zero of J_p (a parameter enters one equation): column c of S
        # is zero above that row r0(c) (a chain hands nothing upwards), so a lane may hold its column ROTATED -- register
        # k = row (r0(c) + k) mod NV -- and the J_p term enters at k = 0 in every lane: the step of a column is
        # z_k = rd[r0 + k] z_k + cc[r0 + k] z_{k-1} with per-lane table addresses and no select at all
        # (sbm_implicit_extrap_seq.hpp).  IM_R0 / IM_JPQ: row and J_p slot of the column's entry.
        col_entries = {}
        for i in range(n):
            for q, (_, c) in enumerate(d.jp_rows[i]):
                col_entries.setdefault(c, []).append((i, q))
        n_cols = (max(col_entries) + 1) if col_entries else 0
        rot = chain and n_cols > 0 and all(len(col_entries.get(c, [])) == 1 for c in range(n_cols))
        L += ["  static constexpr bool IM_ROT = %s;" % ("true" if rot else "false"),
              "  __device__ __forceinline__ static int im_r0(int col) { return SBM_IM_R0[IM_ROT ? col : 0]; }",
              "  __device__ __forceinline__ static int im_jpq(int col) { return SBM_IM_JPQ[IM_ROT ? col : 0]; }"]
        L += emit_distributed(spec, d, pattern, ops)
        return L, dict(tri=True, rstart=rstart, pos=pos, nm=n_table, rot=rot,
                       r0=[col_entries[c][0][0] for c in range(n_cols)] if rot else [0],
                       jpq=[col_entries[c][0][1] for c in range(n_cols)] if rot else [0])
    L += ["  __device__ __forceinline__ static void im_solve_tri(const double*, double (&)[NV]) {}",
          "  template <int RPL>",
          "  __device__ __forceinline__ static void im_solve_tri_pick(const double*, const double*, int, double (&)[RPL]) {}",
          "  static constexpr bool IM_SENS_TRI = false;",
          "  __device__ __forceinline__ static void im_sens_tri(const double*, const double*, double, int, double (&)[NV]) {}",
          "  static constexpr int IM_MF = IM_NM;",
          "  static constexpr bool IM_CHAIN = false;",
          "  static constexpr bool IM_ROT = false;",
          "  __device__ __forceinline__ static int im_r0(int) { return 0; }",
          "  __device__ __forceinline__ static int im_jpq(int) { return 0; }"]
    L += emit_distributed(spec, d, pattern, ops)
    return L, dict(tri=False, nm=nm)


def emit_tables(spec, d, meta):
    """Namespace-scope tables of the distributed triangular form: where row lane i puts its reciprocal
    pivot (IM_RSTART[row]) and class slot s of its row (IM_MFPOS[slot][row]; IM_NM = nowhere), and which
    slot holds the diagonal entry (IM_DIAGSLOT[row], -1: none)."""
    n = spec.n_vars
    max_jy = max([len(x) for x in d.jy_rows] + [1])
    if not meta['tri']:
        return ["__constant__ short SBM_IM_RSTART[1] = {0};", "__constant__ short SBM_IM_DIAGSLOT[1] = {-1};",
                "__constant__ short SBM_IM_MFPOS[1] = {0};", "__constant__ short SBM_IM_R0[1] = {0};",
                "__constant__ short SBM_IM_JPQ[1] = {0};", ""]
    nm = meta['nm']
    diag = [-1] * n
    mfpos = [[nm] * n for _ in range(max_jy)]
    for i in range(n):
        for sidx, (e_idx, c) in enumerate(d.jy_rows[i]):
            if c == i:
                diag[i] = sidx
            else:
                mfpos[sidx][i] = meta['pos'][(i, c)]
    return ["// distributed triangular solve (emit_implicit.py)",
            "__constant__ short SBM_IM_RSTART[%d] = {%s};" % (n, ", ".join(str(v) for v in meta['rstart'])),
            "__constant__ short SBM_IM_DIAGSLOT[%d] = {%s};" % (n, ", ".join(str(v) for v in diag)),
            "__constant__ short SBM_IM_MFPOS[%d] = {%s};" % (max_jy * n, ", ".join(str(v) for sl in mfpos for v in sl)),
            "__constant__ short SBM_IM_R0[%d] = {%s};" % (len(meta['r0']), ", ".join(str(v) for v in meta['r0'])),
            "__constant__ short SBM_IM_JPQ[%d] = {%s};" % (len(meta['jpq']), ", ".join(str(v) for v in meta['jpq'])),
            ""]
