"""Row-group form: the row-lane mapping with the rows of a column split over several lanes.

The row-lane kernel (emit_rowlane.py) keeps one sensitivity column per lane.  A model with
fewer than 64 columns leaves lanes idle (cascade20: 40 of 64), and every lane carries all NV
rows of every Runge-Kutta stage vector, which is what overflows the 256 architectural VGPRs.
Here the lanes of a wavefront form G groups of C lanes; lane (g, c') integrates rows
[g*RPG, (g+1)*RPG) of the CPL columns c', c'+C, ...: RPG*CPL elements per lane instead of NV
(cascade20: G=3, C=20, CPL=2, RPG=7 -> 14 elements instead of 20, 60 of 64 lanes busy).

All lanes run the same instruction stream, so "local row r" must look alike in every group:

  * the J_y terms of local row r are the union, over the groups, of the terms of global row
    g*RPG + r, matched by their cyclic column offset (m - i) mod NV (a feedback from the last
    to the first species is the cyclic neighbour of a sub-diagonal); a group that lacks a term
    reads a zero coefficient;
  * a term whose source row is the same LOCAL row r' in every group is a register operand
    (z[r']); any other term is a HALO term: the source rows are published to LDS once per
    stage (``publish_rowgroup``) and read back through a per-lane offset (table RG_HSRC);
  * coefficients come from the LDS table JYL[row][term] the row lanes fill, read through a
    per-lane base pointer, so the same static offset serves every group.

LDS layout of the tables A (J_p entries) and H (halo rows): indexed by LOCAL row and LANE,
[r][lane][cc] with CPL doubles per lane, so element (row i, column j) sits at
  (i % RPG)*LS + CPL*((i // RPG)*C + j % C) + j // C,      LS = 64*CPL.
Every lane owns private slots: the 64 lanes of a wave access consecutive CPL*8-byte words (no bank
conflicts, one ds_read_b128 per local row for CPL = 2), slots of idle lanes and padded rows are
never written and stay zero.

COLUMN CHUNKS.  The sensitivity columns of a trajectory are coupled only through the state
(S' = J_y(y) S + J_p, column by column), so a trajectory whose columns do not fit one wavefront --
more than 64 parameters, or more elements per lane than the register file holds -- is cut into NCH
chunks of C*CPL columns, each integrated by its own wavefront together with a private copy of the
state (blockIdx.y = chunk).  No data is exchanged between the chunks; each runs its own step-size
control on (state, its columns), so all of them meet the tolerance, on step sequences of their own.

MORE THAN 64 STATE VARIABLES.  The state lives one component per lane; beyond 64 rows a lane takes rows
lane, lane + 64, ... (up to four), evaluating their classes one after the other.  Nothing else changes:
the tables are indexed by global row.

THREE LAYOUTS of the same form are emitted per model, nested structs the kernels are templates over: RG0 (the
throughput split, planned for DOPRI45's seven stage vectors), RG1 (small batches: more, smaller column chunks -- a single
parameter vector leaves the chip empty and extra wavefronts are free), RG2 (DOP853: twelve stage vectors, the planner's
register budgets scaled by 7/12).  RG1 / RG2 are aliases of RG0 where the planner finds nothing different.

``plan`` decides whether the form pays at all (RG_OK); the integrator falls back to the
row-lane kernel (or, beyond 64 columns, to the per-wave kernel) otherwise.
"""
from __future__ import annotations


REG_ELEMS = 15      # elements per lane up to which DOPRI45's stage vectors fit 256 registers (two waves per SIMD)
AGPR_ELEMS = 26     # ... up to which they fit 512 together with the operands (one wave per SIMD, v_accvgpr traffic); beyond: scratch
# DOP853 keeps twelve vectors of a lane's elements alive where DOPRI45 keeps seven: its split (layout RG2) is planned with
# budgets scaled by 7/12.  Measured on cascade20, 4096 vectors (tests/tools/dev_dop853.py): 14 elements per lane (the DOPRI45
# split) 512 registers + 1078 scratch instructions, ~60 ms per pass; 7 elements in two column chunks 262 registers, no
# scratch, ~3 ms.
REG_ELEMS_853 = 8
AGPR_ELEMS_853 = 15
MAX_ROWS_PER_LANE = 4
STATE_COST = 5      # work of the per-chunk state evaluation, in elements per lane


def _best_split(n, ncols, g_min, max_lanes=64):
    """(elems, G, C, CPL, RPG) with the fewest elements per lane for ``ncols`` columns on one wavefront."""
    best = None
    for G in range(g_min, 9 if n <= max_lanes else 17):
        C = max_lanes // G
        if C < 1:
            break
        CPL = -(-ncols // C)
        C = -(-ncols // CPL)       # balance the columns over the CPL slots
        RPG = -(-n // G)
        if G > 1 and (G - 1) * RPG >= n:     # an empty last group: a smaller G does the same
            continue
        elems = RPG * CPL
        if best is None or elems < best[0]:
            best = (elems, G, C, CPL, RPG)
    return best


# Calibration of the cost model below (scripts/dev_plan_time.py, SBM_RG_FORCE_PLAN; kernel ms, DOPRI45):
#   cascade40, 2048 vectors   (4,16,1,10,5) 12.3   (3,20,1,14,4) 12.6   (6,10,1,7,8) 15.9   (2,27,1,20,3) 19.1
#       model cost                           75                76                 96                  120
#   cascade20, 4096 vectors   (3,20,2,7,1)   5.05  (3,20,1,7,2)   6.66  (6,10,1,4,4)  8.85
#       model cost                           19                24                 36
def _cost(elems, nch, rpl=1, reg_elems=REG_ELEMS, agpr_elems=AGPR_ELEMS):
    # the kernel runs two wavefronts per SIMD when elements + state rows per lane <= REG_ELEMS (SBM_RG_MIN_WAVES)
    spill = 1.0 if elems + rpl <= reg_elems else (1.6 if elems <= agpr_elems else 4.0 * elems / agpr_elems)
    return nch * (elems + STATE_COST * rpl) * spill


def plan(n, nk, max_lanes=64, reg_elems=REG_ELEMS, agpr_elems=AGPR_ELEMS):
    """(G, C, CPL, RPG, NCH): lanes (g, c') of NCH column chunks.  None when the row-lane kernel does as
    well (one chunk, and splitting the rows cuts the elements per lane by less than 20 %) or the model has
    more than 4 x 64 state variables."""
    if n < 2 or nk < 1 or n > MAX_ROWS_PER_LANE * max_lanes:
        return None
    import os
    forced = os.environ.get('SBM_RG_FORCE_PLAN')        # developer aid: "G,C,CPL,RPG,NCH" (timing one split against another)
    if forced and reg_elems == REG_ELEMS:
        G, C, CPL, RPG, NCH = (int(v) for v in forced.split(','))
        assert G * C <= max_lanes and G * RPG >= n and C * CPL * NCH >= nk and C * CPL * (NCH - 1) < nk, forced
        return (G, C, CPL, RPG, NCH)
    rpl = -(-n // max_lanes)           # state rows per lane: rows lane, lane + 64, ...
    best = None
    for nch in range(1, 257):
        if nch > nk:
            break
        ncols = -(-nk // nch)
        if nch > 1 and (nch - 1) * ncols >= nk:
            continue                                   # an empty last chunk
        sp = _best_split(n, ncols, 2 if nch == 1 else 1, max_lanes)
        if sp is None:
            continue
        key = (_cost(sp[0], nch, rpl, reg_elems, agpr_elems), nch)
        if best is None or key < best[0]:
            best = (key, sp, nch)
    if best is None:
        return None
    (_, sp, nch) = best
    if nch == 1 and nk <= max_lanes and sp[0] > 0.8 * n and reg_elems == REG_ELEMS:
        return None
    return sp[1:] + (nch,)


# Calibration (scripts/dev_small_batch_time.py, SBM_RG_FORCE_PLAN_SMALL; GPU-busy ms of one Jacobian evaluation of the
# configs[3] project, 8 trajectories of cascade20): 14 elements per lane (throughput split) 1.61, 7 elements 1.45,
# 4 elements 1.01, 3 elements 1.00 / 0.97 / 0.96 with 5 / 8 / 10 chunks -- a plateau: the state evaluation and the step
# control remain.  The small price per chunk below picks the fewest chunks on the plateau.
def plan_latency(n, nk, max_lanes=64, max_chunks=16):
    """The split for SMALL batches (a serial optimiser evaluating one parameter vector at a time: the reference's
    leastsq(project.residuals, x0, Dfun=project.calc_project_jacobian)): the chip is empty, so extra wavefronts are
    free and what counts is the work of ONE wavefront per step -- elements per lane plus the state evaluation.
    More, smaller column chunks (cascade20: four chunks of ten columns, 4 elements per lane instead of 14).
    Returns (G, C, CPL, RPG, NCH) or None."""
    if n < 2 or nk < 1 or n > MAX_ROWS_PER_LANE * max_lanes:
        return None
    import os
    forced = os.environ.get('SBM_RG_FORCE_PLAN_SMALL')   # developer aid, as SBM_RG_FORCE_PLAN
    if forced:
        G, C, CPL, RPG, NCH = (int(v) for v in forced.split(','))
        assert G * C <= max_lanes and G * RPG >= n and C * CPL * NCH >= nk and C * CPL * (NCH - 1) < nk, forced
        return (G, C, CPL, RPG, NCH)
    rpl = -(-n // max_lanes)
    best = None
    for nch in range(1, max_chunks + 1):
        if nch > nk:
            break
        ncols = -(-nk // nch)
        if nch > 1 and (nch - 1) * ncols >= nk:
            continue
        sp = _best_split(n, ncols, 1, max_lanes)
        if sp is None:
            continue
        key = (sp[0] + STATE_COST * rpl + 0.3 * nch, nch)     # a wavefront's work per step; a small price per chunk
        if best is None or key < best[0]:
            best = (key, sp, nch)
    return None if best is None else best[1][1:] + (best[2],)


def latency_plan_or_none(n, nk):
    """The small-batch split when it differs from the throughput split and keeps a lane's share within the
    two-wavefronts-per-SIMD budget; None: the throughput split serves small batches too."""
    p0, p1 = plan(n, nk), plan_latency(n, nk)
    if p0 is None or p1 is None or p1 == p0:
        return None
    if p1[3] * p1[2] + -(-n // 64) > REG_ELEMS or p1[3] * p1[2] >= p0[3] * p0[2]:
        return None
    return p1


def dop853_plan_or_none(n, nk):
    """The split for DOP853's twelve stage vectors, when the form applies at all and the split differs from the
    throughput one; None: DOP853 runs the throughput split (small shares already)."""
    p0 = plan(n, nk)
    if p0 is None:
        return None
    p2 = plan(n, nk, reg_elems=REG_ELEMS_853, agpr_elems=AGPR_ELEMS_853)
    return None if (p2 is None or p2 == p0) else p2


def layout(spec, d, latency=False, dop853=False):
    """Term structure of the row-group form (``latency``: the small-batch split; ``dop853``: the split planned for
    twelve stage vectors).  Returns None when the form does not apply (``latency`` / ``dop853``: or when the throughput
    split serves as well)."""
    n, nk = spec.n_vars, spec.n_sens
    p = dop853_plan_or_none(n, nk) if dop853 else (latency_plan_or_none(n, nk) if latency else plan(n, nk))
    if p is None:
        return None
    G, C, CPL, RPG, NCH = p
    pattern = []                       # per global row: {cyclic offset: (e_idx, m)}
    for i in range(n):
        pattern.append({(m - i) % n: (e_idx, m) for e_idx, m in d.jy_rows[i]})
    terms = []                         # per local row: list of dict(rel, own=r' or None, halo=t or None)
    hsrc = []                          # per halo term: [source global row per group] (n_pad = zero row)
    publish = set()
    n_pad = G * RPG                    # index of the always-zero row of H
    for r in range(RPG):
        rows = [g * RPG + r for g in range(G) if g * RPG + r < n]
        rels = sorted({rel for i in rows for rel in pattern[i]}, key=lambda x: (x != 0, x))
        tl = []
        for rel in rels:
            own = None
            srcs = {}
            for g in range(G):
                i = g * RPG + r
                if i < n and rel in pattern[i]:
                    srcs[g] = pattern[i][rel][1]
            local = {m - g * RPG for g, m in srcs.items()}
            if len(local) == 1:
                rp = next(iter(local))
                if 0 <= rp < RPG and all(m // RPG == g for g, m in srcs.items()):
                    own = rp
            if own is not None:
                tl.append(dict(rel=rel, own=own, halo=None))
            else:
                tl.append(dict(rel=rel, own=None, halo=len(hsrc)))
                hsrc.append([srcs.get(g, n_pad) for g in range(G)])
                for m in srcs.values():
                    publish.add(m % RPG)
        terms.append(tl)
    jys = max([len(t) for t in terms] + [1])
    # where row lane i puts slot s of its class: JYL[i*jys + k]
    max_jy = max([len(x) for x in d.jy_rows] + [1])
    spare = n_pad * jys                # one slot past the table
    jypos = [[spare] * n for _ in range(max_jy)]
    for i in range(n):
        r = i % RPG
        rel_to_k = {t['rel']: k for k, t in enumerate(terms[r])}
        for s, (e_idx, m) in enumerate(d.jy_rows[i]):
            jypos[s][i] = i * jys + rel_to_k[(m - i) % n]
    return dict(G=G, C=C, CPL=CPL, RPG=RPG, NCH=NCH, LS=64 * CPL, terms=terms, hsrc=hsrc,
                publish=sorted(publish), jys=jys, jypos=jypos, max_jy=max_jy, n_pad=n_pad)


def emit_tables(spec, d, tag='RG0', latency=False, dop853=False):
    """Namespace-scope tables of one layout (``tag``: RG0 = throughput split, RG1 = small-batch split, RG2 = DOP853's)."""
    lay = layout(spec, d, latency=latency, dop853=dop853)
    if lay is None:
        return [], None
    n = spec.n_vars
    L = ["// row-group tables, layout %s" % tag,
         "__constant__ short SBM_%s_JYPOS[%d] = {%s};   // [slot][row] -> index into JYL" %
         (tag, lay['max_jy'] * n, ", ".join(str(v) for slot in lay['jypos'] for v in slot))]
    nh = max(len(lay['hsrc']), 1)
    flat = [v for t in lay['hsrc'] for v in t] or [lay['n_pad']] * lay['G']
    L += ["__constant__ short SBM_%s_HSRC[%d] = {%s};   // [halo term][group] -> source row" %
          (tag, nh * lay['G'], ", ".join(str(v) for v in flat)), ""]
    return L, lay


def emit_members(spec, d, lay, tag='RG0', alias_of=None):
    """``struct <tag>`` nested in ``struct SbmModel``: one layout of the row-group form (the kernels are templates
    over it); with ``lay is None`` only RG_OK = false and inert stubs -- or, with ``alias_of``, another name for
    that layout."""
    if lay is None and alias_of is not None:
        return ["  using %s = %s;   // this use runs the throughput split" % (tag, alias_of)]
    body = _emit_struct_body(spec, d, lay, tag)
    return ["  struct %s {" % tag] + body + ["  };"]


def _emit_struct_body(spec, d, lay, tag):
    n = spec.n_vars
    if lay is None:
        return ["  // ---- row-group form: does not pay for this model ----",
                "  static constexpr bool RG_OK = false;",
                "  static constexpr int RG_G = 1, RG_C = 64, RG_CPL = 1, RG_RPG = NV, RG_JYS = 1, RG_NHALO = 0, RG_LS = 64;",
                "  static constexpr int RG_NCH = 1;",
                "  __device__ __forceinline__ static int rg_jypos(int, int) { return 0; }",
                "  __device__ __forceinline__ static int rg_hsrc(int, int) { return 0; }",
                "  __device__ __forceinline__ static int rg_pos(int row, int col) { return row * 64 + col; }",
                "  template <int NZ> __device__ __forceinline__ static void publish_rowgroup(double*, const double (&)[NZ]) {}",
                "  __device__ __forceinline__ static void load_rowgroup(const double*, const double*, double (&)[NV], double (&)[NV]) {}",
                "  template <int NZ> __device__ __forceinline__ static void apply_rowgroup(const double (&)[NV], const double (&)[NV],",
                "      const double*, const int (&)[1], const double (&)[NZ], double (&)[NZ]) {}"]
    G, C, CPL, RPG, jys, LS = lay['G'], lay['C'], lay['CPL'], lay['RPG'], lay['jys'], lay['LS']
    nh = len(lay['hsrc'])
    L = ["  // ---- row-group form (sbm_sens_rowgroup_kernel): lane (g, c') = rows [g*RPG, (g+1)*RPG) of",
         "  //      columns c' + C*cc; element index of (local row r, column slot cc) is r + RPG*cc ----",
         "  static constexpr bool RG_OK = true;",
         "  static constexpr int RG_G = %d, RG_C = %d, RG_CPL = %d, RG_RPG = %d, RG_JYS = %d, RG_NHALO = %d;"
         % (G, C, CPL, RPG, jys, nh),
         "  static constexpr int RG_NCH = %d;  // column chunks (wavefronts) per trajectory, C*CPL columns each" % lay['NCH'],
         "  static constexpr int RG_LS = %d;   // local-row stride of the A / H tables (doubles): [r][lane][cc]" % LS,
         "  // position of (row, column of the chunk) in A / H",
         "  __device__ __forceinline__ static int rg_pos(int row, int col) {",
         "    return (row % RG_RPG) * RG_LS + RG_CPL * ((row / RG_RPG) * RG_C + col % RG_C) + col / RG_C;",
         "  }",
         "  __device__ __forceinline__ static int rg_jypos(int slot, int row) { return SBM_%s_JYPOS[slot * NV + row]; }" % tag,
         "  __device__ __forceinline__ static int rg_hsrc(int term, int group) { return SBM_%s_HSRC[term * RG_G + group]; }" % tag,
         "  // rows other groups read: h_lane = H + CPL*lane",
         "  template <int NZ>",
         "  __device__ __forceinline__ static void publish_rowgroup(double* h_lane, const double (&z)[NZ]) {",
         "    (void)h_lane; (void)z;"]
    for r in lay['publish']:
        for cc in range(CPL):
            L.append("    h_lane[%d] = z[%d];" % (r * LS + cc, r + RPG * cc))
    L += ["  }",
          "  // operands of apply_rowgroup from LDS: a_lane = A + CPL*lane, jy_lane = &JYL[g*RPG][0] (layout",
          "  // JYL[row][RG_JYS]); acol[r + RPG*cc] = this lane's J_p entries, coef[r*RG_JYS + k] = J_y coefficients",
          "  __device__ __forceinline__ static void load_rowgroup(const double* a_lane, const double* jy_lane,",
          "                                                       double (&acol)[%d], double (&coef)[%d]) {" % (RPG * CPL, RPG * jys)]
    for r in range(RPG):
        for k in range(jys):
            L.append("    coef[%d] = jy_lane[%d];" % (r * jys + k, r * jys + k))
    for r in range(RPG):
        for cc in range(CPL):
            L.append("    acol[%d] = a_lane[%d];" % (r + RPG * cc, r * LS + cc))
    L += ["  }",
          "  // dz = J_y z + A for the lane's rows of its columns; h_all = H, hoff[t] = rg_pos(rg_hsrc(t, g), c')",
          "  // or the zero slot RPG*LS (term absent in this lane's group).",
          "  template <int NZ>",
          "  __device__ __forceinline__ static void apply_rowgroup(const double (&acol)[%d], const double (&coef)[%d]," % (RPG * CPL, RPG * jys),
          "                                                        const double* h_all, const int (&hoff)[%d]," % max(nh, 1),
          "                                                        const double (&z)[NZ], double (&dz)[NZ]) {",
          "    (void)h_all; (void)hoff;"]
    for r in range(RPG):
        for k, t in enumerate(lay['terms'][r]):
            if t['halo'] is not None:
                for cc in range(CPL):
                    L.append("    const double h%d_%d_%d = h_all[hoff[%d] + %d];" % (r, k, cc, t['halo'], cc))
    # rows without halo terms first: their operands arrive before the halo reads issued last
    order = sorted(range(RPG), key=lambda r: any(t['halo'] is not None for t in lay['terms'][r]))
    for cc in range(CPL):
        for r in order:
            expr = "acol[%d]" % (r + RPG * cc)
            for k, t in enumerate(lay['terms'][r]):
                src = ("z[%d]" % (t['own'] + RPG * cc)) if t['own'] is not None else ("h%d_%d_%d" % (r, k, cc))
                expr = "fma(coef[%d], %s, %s)" % (r * jys + k, src, expr)
            L.append("    dz[%d] = %s;" % (r + RPG * cc, expr))
    L += ["  }"]
    return L
