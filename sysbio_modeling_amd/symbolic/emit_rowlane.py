"""Row-lane form of a model: SIMD across isomorphic equations.

The integrator kernels keep one trajectory per wavefront.  What is identical for all
sensitivity columns of a trajectory -- f(y), the non-zeros of J_y and of J_p -- is scalar work
per trajectory; evaluating it redundantly on all 64 lanes is what bounds the per-wave kernel
(rocprofv3: VALU-issue bound, ~40 % of the instructions).  Rate-law networks, however, are
made of a few kinetic forms repeated over many species: rows whose expression bundle
(f_i, dF_i/dy, dF_i/dp) is the same tree up to a renaming of symbols form a CLASS, and a class is
evaluated once, lane i working on row i with per-lane operands.  This module finds the classes
and prints

  * ``__constant__`` tables: class of each row, which state / parameter feeds each operand slot,
    where each produced J_y / J_p entry goes;
  * ``class_dispatch`` -- the class bodies, each run under the exec mask of its rows;
  * ``apply_rowlane`` -- dz_i = sum_m J_y[i,m] z_m + A[i][lane] for a sensitivity column, with
    J_y read (wave-uniform) and A read (one column per lane) from LDS: no per-lane selects.
"""
from __future__ import annotations

from collections import OrderedDict

import sympy
from sympy import Symbol, cse


def _canonical(bundle, var_index, par_index):
    """Rename symbols to placeholders in order of first appearance (depth-first over the bundle).
    Returns (canonical expr tuple, [state indices per Y slot], [param indices per P slot])."""
    ys, ps = [], []
    mapping = {}
    for expr in bundle:
        for node in sympy.preorder_traversal(expr):
            if isinstance(node, Symbol) and node not in mapping:
                if node.name in var_index:
                    mapping[node] = Symbol('YS_%d' % len(ys))
                    ys.append(var_index[node.name])
                elif node.name in par_index:
                    mapping[node] = Symbol('PS_%d' % len(ps))
                    ps.append(par_index[node.name])
    canon = tuple(e.xreplace(mapping) for e in bundle)
    return canon, ys, ps


def find_classes(spec, d):
    """Group rows by the structure of (f_i, J_y entries of the row, J_p entries of the row)."""
    var_index = {v: i for i, v in enumerate(spec.variables)}
    par_index = {p: i for i, p in enumerate(spec.params)}
    classes = OrderedDict()   # key -> dict(canon, rows=[...])
    row_info = []
    for i in range(spec.n_vars):
        bundle = [d.f_c[i]] + [d.jy_c[e] for e, _ in d.jy_rows[i]] + [d.jp_c[e] for e, _ in d.jp_rows[i]]
        canon, ys, ps = _canonical(bundle, var_index, par_index)
        key = (sympy.srepr(canon), len(d.jy_rows[i]), len(d.jp_rows[i]))
        if key not in classes:
            classes[key] = dict(canon=canon, rows=[], n_jy=len(d.jy_rows[i]), n_jp=len(d.jp_rows[i]),
                                n_ys=len(ys), n_ps=len(ps))
        classes[key]['rows'].append(i)
        row_info.append(dict(cls=list(classes.keys()).index(key), ys=ys, ps=ps))
    return list(classes.values()), row_info


def emit_rowlane_tables(spec, d, printer_factory):
    """Namespace-scope ``__constant__`` tables + sizes; returns (lines, meta)."""
    classes, row_info = find_classes(spec, d)
    n = spec.n_vars
    max_ys = max([c['n_ys'] for c in classes] + [1])
    max_ps = max([c['n_ps'] for c in classes] + [1])
    max_jy = max([c['n_jy'] for c in classes] + [1])
    max_jp = max([c['n_jp'] for c in classes] + [1])
    tag = "SBM_RL"

    def table(name, rows_of_slots):
        flat = ", ".join(str(v) for slot in rows_of_slots for v in slot)
        return "__constant__ short %s_%s[%d] = {%s};" % (tag, name, len(rows_of_slots) * n, flat)

    ys_t = [[(row_info[i]['ys'][s] if s < len(row_info[i]['ys']) else 0) for i in range(n)] for s in range(max_ys)]
    ps_t = [[(row_info[i]['ps'][s] if s < len(row_info[i]['ps']) else 0) for i in range(n)] for s in range(max_ps)]
    nj = max(len(d.jy), 1)
    jy_t = [[(d.jy_rows[i][s][0] if s < len(d.jy_rows[i]) else nj) for i in range(n)] for s in range(max_jy)]
    # position in the additive matrix A[NV][64] (+ one spare slot at NV*64 for unused outputs)
    jp_t = [[(i * 64 + d.jp_rows[i][s][1] if s < len(d.jp_rows[i]) else n * 64) for i in range(n)]
            for s in range(max_jp)]
    # the column alone (-1: unused slot), for kernels that cut the columns into chunks (any number of columns)
    jpc_t = [[(d.jp_rows[i][s][1] if s < len(d.jp_rows[i]) else -1) for i in range(n)] for s in range(max_jp)]
    # the column of each J_y slot (-1: unused), for the kernel that hands J_y to the matrix cores as a dense tile
    jyc_t = [[(d.jy_rows[i][s][1] if s < len(d.jy_rows[i]) else -1) for i in range(n)] for s in range(max_jy)]
    L = ["// row-lane tables: [slot][row]",
         "__constant__ short %s_CLASS[%d] = {%s};" % (tag, n, ", ".join(str(r['cls']) for r in row_info)),
         table("YS", ys_t), table("PS", ps_t), table("JYOUT", jy_t), table("APOS", jp_t), table("JPCOL", jpc_t),
         table("JYCOL", jyc_t), ""]
    meta = dict(classes=classes, max_ys=max_ys, max_ps=max_ps, max_jy=max_jy, max_jp=max_jp)
    return L, meta


def emit_rowlane_members(spec, d, meta, make_printer):
    """Member functions of ``struct SbmModel`` for the row-lane kernel."""
    n = spec.n_vars
    classes = meta['classes']
    L = ["  // ---- row-lane form (sbm_sens_rowlane_kernel) ----",
         "  static constexpr int RL_NCLASS = %d;" % len(classes),
         "  static constexpr int RL_MAXYS = %d, RL_MAXPS = %d, RL_MAXJY = %d, RL_MAXJP = %d;"
         % (meta['max_ys'], meta['max_ps'], meta['max_jy'], meta['max_jp']),
         "  static constexpr int RL_LARGEST_CLASS = %d;  // rows evaluated side by side" %
         max(len(c['rows']) for c in classes),
         "  // table accessors (the tables are namespace-scope __constant__ arrays above)",
         "  __device__ __forceinline__ static int rl_class(int row) { return SBM_RL_CLASS[row]; }",
         "  __device__ __forceinline__ static int rl_ys(int slot, int row) { return SBM_RL_YS[slot * NV + row]; }",
         "  __device__ __forceinline__ static int rl_ps(int slot, int row) { return SBM_RL_PS[slot * NV + row]; }",
         "  __device__ __forceinline__ static int rl_jyout(int slot, int row) { return SBM_RL_JYOUT[slot * NV + row]; }",
         "  __device__ __forceinline__ static int rl_apos(int slot, int row) { return SBM_RL_APOS[slot * NV + row]; }",
         "  __device__ __forceinline__ static int rl_jpcol(int slot, int row) { return SBM_RL_JPCOL[slot * NV + row]; }",
         "  __device__ __forceinline__ static int rl_jycol(int slot, int row) { return SBM_RL_JYCOL[slot * NV + row]; }",
         "  // one class body per distinct kinetic form; lane = row, operands per lane",
         "  __device__ __forceinline__ static void class_dispatch(int cls, double t, const double (&ys)[RL_MAXYS],",
         "                                                        const double (&ps)[RL_MAXPS], double& f,",
         "                                                        double (&jy)[RL_MAXJY], double (&jp)[RL_MAXJP]) {",
         "    (void)t; (void)ys; (void)ps;"]
    smap = {'t': 't'}
    for s in range(meta['max_ys']):
        smap['YS_%d' % s] = 'ys[%d]' % s
    for s in range(meta['max_ps']):
        smap['PS_%d' % s] = 'ps[%d]' % s
    pr = make_printer(smap)
    # Branch-free: every class body is evaluated on every lane (a divergent if/else chain would
    # execute all bodies one after the other anyway) and the lane keeps the results of ITS class
    # through by-value selects.  No per-lane control flow is left in the kernel, which matters
    # because other lanes read these registers with v_readlane.  The LAST class is the default of
    # the select chain (n-1 selects per output instead of n): lanes without a row (cls = -1) end up
    # with its values, which the kernels discard (spare LDS slots) or zero (f).
    bodies = []
    for ci, c in enumerate(classes):
        L.append("    // class %d: rows %s" % (ci, ", ".join(str(r) for r in c['rows'])))
        repl, red = cse(list(c['canon']), symbols=sympy.numbered_symbols('c%d_x' % ci), optimizations='basic')
        for sym, e in repl:
            L.append("    const double %s = %s;" % (sym, pr.doprint(e)))
        bodies.append([pr.doprint(e) for e in red])
    last = len(classes) - 1
    for ci in range(last):
        L.append("    const bool is%d = (cls == %d);" % (ci, ci))

    def chain(kind, k):
        """value of output (kind, k): the last class's expression (or 0), overridden class by class"""
        def of(ci):
            c = classes[ci]
            if kind == 'f':
                return bodies[ci][0]
            if kind == 'jy':
                return bodies[ci][1 + k] if k < c['n_jy'] else "0.0"
            return bodies[ci][1 + c['n_jy'] + k] if k < c['n_jp'] else "0.0"
        expr = of(last)
        for ci in range(last - 1, -1, -1):
            expr = "SBM_SEL(is%d, %s, %s)" % (ci, of(ci), expr)
        return expr

    L.append("    f = %s;" % chain('f', 0))
    for k in range(meta['max_jy']):
        L.append("    jy[%d] = %s;" % (k, chain('jy', k)))
    for k in range(meta['max_jp']):
        L.append("    jp[%d] = %s;" % (k, chain('jp', k)))
    # which (row lane, slot) holds J_y non-zero e; entries that depend on parameters only are
    # STATIC: the same for every stage of every step, broadcast once per kernel (rl_static)
    where = {}
    for i in range(n):
        for k, (e_idx, c) in enumerate(d.jy_rows[i]):
            where[e_idx] = (i, k)
    var_names = set(spec.variables)
    static_idx = {}
    for e_idx in sorted(where):
        syms = {str(x) for x in d.jy_c[e_idx].free_symbols}
        if not (syms & var_names) and 't' not in syms:
            static_idx[e_idx] = len(static_idx)
    nst = max(len(static_idx), 1)
    L += ["  }", "",
          "  static constexpr int RL_NSTATIC = %d;   // J_y entries that depend on parameters only" % len(static_idx),
          "  // run once per kernel, after one class_dispatch: broadcast the static entries",
          "  __device__ __forceinline__ static void rl_static(const double (&jy)[RL_MAXJY], double (&sj)[%d]) {" % nst,
          "    (void)jy;"]
    if not static_idx:
        L.append("    sj[0] = 0.0;")
    for e_idx, si in static_idx.items():
        r, k = where[e_idx]
        L.append("    sj[%d] = SBM_LANE_BCAST(jy[%d], %d);" % (si, k, r))
    L += ["  }", "",
          "  // dz = J_y z + acol, one sensitivity column per lane.  A state-dependent J_y[i,m] sits in",
          "  // register jy[slot] of row lane i: SBM_LANE_BCAST (v_readlane, literal lane) turns it into a",
          "  // scalar operand; static entries come from sj; acol[i] = A[i][lane] was fetched from LDS.",
          "  template <int NZ>",
          "  __device__ __forceinline__ static void apply_rowlane(const double (&jy)[RL_MAXJY], const double (&sj)[%d]," % nst,
          "                                                       const double (&acol)[NV], const double (&z)[NZ],",
          "                                                       double (&dz)[NZ]) {",
          "    (void)jy; (void)sj;"]
    for i in range(n):
        expr = "acol[%d]" % i
        for e_idx, c in d.jy_rows[i]:
            if e_idx in static_idx:
                expr = "fma(sj[%d], z[%d], %s)" % (static_idx[e_idx], c, expr)
            else:
                r, k = where[e_idx]
                expr = "fma(SBM_LANE_BCAST(jy[%d], %d), z[%d], %s)" % (k, r, c, expr)
        L.append("    dz[%d] = %s;" % (i, expr))
    L += ["  }", "",
          "  // the same product with the J_y entries read from a table jyl[row * RL_MAXJY + slot] (LDS): the form of the",
          "  // packed kernel, where several trajectories share a wavefront and a lane cannot name its row lane literally",
          "  template <int NZ>",
          "  __device__ __forceinline__ static void apply_lds(const double* jyl, const double (&acol)[NV],",
          "                                                   const double (&z)[NZ], double (&dz)[NZ]) {",
          "    (void)jyl;"]
    for i in range(n):
        expr = "acol[%d]" % i
        for k, (e_idx, c) in enumerate(d.jy_rows[i]):
            expr = "fma(jyl[%d], z[%d], %s)" % (i * meta['max_jy'] + k, c, expr)
        L.append("    dz[%d] = %s;" % (i, expr))
    L += ["  }"]
    return L
