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
        rstart, pos = [], {}
        k = 0
        for i in range(n):
            rstart.append(k)
            k += 1
            for (r, c) in pattern:
                if r == i and c < i:
                    pos[(r, c)] = k
                    k += 1
        assert k == nm
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
        return L, dict(tri=True, rstart=rstart, pos=pos, nm=nm)
    L += ["  __device__ __forceinline__ static void im_solve_tri(const double*, double (&)[NV]) {}"]
    return L, dict(tri=False, nm=nm)


def emit_tables(spec, d, meta):
    """Namespace-scope tables of the distributed triangular form: where row lane i puts its reciprocal
    pivot (IM_RSTART[row]) and class slot s of its row (IM_MFPOS[slot][row]; IM_NM = nowhere), and which
    slot holds the diagonal entry (IM_DIAGSLOT[row], -1: none)."""
    n = spec.n_vars
    max_jy = max([len(x) for x in d.jy_rows] + [1])
    if not meta['tri']:
        return ["__constant__ short SBM_IM_RSTART[1] = {0};", "__constant__ short SBM_IM_DIAGSLOT[1] = {-1};",
                "__constant__ short SBM_IM_MFPOS[1] = {0};", ""]
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
            ""]
