"""Source emitters: one symbolic model -> Python, C and HIP right-hand sides.

Replaces the reference's Python-only emitter (symbolic/sympy_tools.py:162-216
``make_ode_model``: one ``yout[i] = (expr)`` line per equation, ``exec``-ed).
All three targets are printed from the SAME common-subexpression-eliminated
expression set so that the GPU kernels, the C oracle RHS and the Python
callables (reference callback contract ``f(y, t, yout, p) -> None``,
tests/test_utils/sens_jittable_model.py:1-38) agree term by term.

Layout contract (reference symbolic/sympy_tools.py:130-146,185-195):
``yout[0:n]`` state derivatives, ``yout[n + i*k + j]`` = d/dt (d y_i / d p_j)
with j running over the non-fixed parameters in ``param_order``.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field

import sympy
from sympy import Symbol, cse
from sympy.printing.c import C99CodePrinter

from . import sympy_tools
from . import emit_rowlane
from . import emit_rowgroup
from . import emit_implicit


_RCP = sympy.Function('SBM_RCP')


_cheapest_memo = {}


def _cheapest(expr):
    """Derivatives of rational rate laws come out of ``diff`` as sums such as
    k/(1+x) - k*x/(1+x)**2; the factored form k/(1+x)**2 is a third of the work.
    Pick the cheapest of raw / factored / simplified (divisions weighted heavily).

    Networks repeat a few kinetic forms over many species, so the same expression comes back with other symbols
    hundreds of times (the dense 20-state network: 440 Jacobian entries, 3.5 minutes of ``simplify``): the work is done
    once per STRUCTURE -- symbols renamed to placeholders in order of first appearance -- and renamed back."""
    def cost(e):
        return sympy.count_ops(e, visual=False) + 8 * len([a for a in sympy.preorder_traversal(e)
                                                           if a.is_Pow and a.exp.is_number and a.exp.is_negative])
    order = []
    for node in sympy.preorder_traversal(expr):
        if isinstance(node, Symbol) and node not in order:
            order.append(node)
    fwd = {sym: Symbol('_q%d' % i) for i, sym in enumerate(order)}
    canon = expr.xreplace(fwd)
    key = sympy.srepr(canon)
    if key not in _cheapest_memo:
        best = canon
        for f in (sympy.factor, sympy.simplify):
            try:
                cand = f(canon)
            except Exception:
                continue
            if cost(cand) < cost(best):
                best = cand
        _cheapest_memo[key] = best
    return _cheapest_memo[key].xreplace({v: k for k, v in fwd.items()})


def _canon_rcp(expr):
    """b**(-n) -> SBM_RCP(b)**n (integer n) so that CSE shares ONE reciprocal per
    distinct denominator; reciprocals dominate the cost of rate-law right-hand sides."""
    def is_neg_pow(e):
        return e.is_Pow and e.exp.is_number and e.exp.is_negative

    def repl(e):
        if e.exp.is_Integer:
            # denominators that differ by a rational factor share one reciprocal: 1 / (y + 1/2) = 2 / (2 y + 1)
            # (stiff50's activation law produces both; a v_rcp_f64 with its two Newton steps is ~35 cycles)
            content, prim = e.base.as_content_primitive()
            if content != 1 and content.is_Rational and content > 0:
                return sympy.Pow(content, e.exp) * sympy.Pow(_RCP(prim), -e.exp)
            return sympy.Pow(_RCP(e.base), -e.exp)
        return _RCP(sympy.Pow(e.base, -e.exp))
    return expr.replace(is_neg_pow, repl)


# ----------------------------------------------------------------------------
# model specification
# ----------------------------------------------------------------------------
@dataclass
class ModelSpec:
    """A model ready for emission.

    variables / params are ordered name lists (the order IS the y / p layout,
    reference ``ordered_params`` / header order symbolic/sympy_tools.py:100-111);
    ``fixed`` names parameters without sensitivity columns; ``equations`` maps
    each variable to d(var)/dt as a SymPy expression in those names and ``t``.
    """
    name: str
    variables: list
    params: list
    equations: OrderedDict
    fixed: list = field(default_factory=list)

    @property
    def n_vars(self):
        return len(self.variables)

    @property
    def n_params(self):
        return len(self.params)

    @property
    def sens_params(self):
        return [p for p in self.params if p not in self.fixed]

    @property
    def n_sens(self):
        return len(self.sens_params)

    @classmethod
    def from_model_dict(cls, model_dict, name='Model'):
        if 'Expanded Equations' not in model_dict:
            sympy_tools.process_model_dict(model_dict)
        params = model_dict['Parameters']
        fixed = [p for p in params if params[p] == 'fixed']
        return cls(name=name, variables=list(model_dict['Variables'].keys()),
                   params=list(params.keys()),
                   equations=OrderedDict(model_dict['Expanded Equations']), fixed=fixed)

    @classmethod
    def from_text(cls, model, name='Model', fixed_params=None):
        md = sympy_tools.parse_model_file(model)
        sympy_tools.process_model_dict(md, fixed_params=fixed_params,
                                       calculate_model_sensitivities=False)
        return cls.from_model_dict(md, name=name)


# ----------------------------------------------------------------------------
# derived form shared by all emitters
# ----------------------------------------------------------------------------
class Derived:
    """f, nnz(J_y), nnz(J_p) after joint CSE."""

    def __init__(self, spec: ModelSpec):
        self.spec = spec
        params = OrderedDict((p, 'fixed' if p in spec.fixed else Symbol(p)) for p in spec.params)
        jy, jp = sympy_tools.derive_sparse_jacobians(spec.equations, params)
        self.jy = jy  # (row, col, expr)
        self.jp = jp  # (row, sens col, expr)
        f_exprs = [_canon_rcp(sympy.sympify(spec.equations[v])) for v in spec.variables]
        all_exprs = f_exprs + [_canon_rcp(_cheapest(e)) for _, _, e in jy] + \
            [_canon_rcp(_cheapest(e)) for _, _, e in jp]
        # joint CSE: subexpressions of f are shared with its derivatives
        self.repl_all, red_all = cse(all_exprs, symbols=sympy.numbered_symbols('x_'),
                                     optimizations='basic')
        n = len(f_exprs)
        # canonical (pre-CSE) forms, re-CSE'd per row block by the cooperative kernel's emitter
        self.f_c = all_exprs[:n]
        self.jy_c = all_exprs[n:n + len(jy)]
        self.jp_c = all_exprs[n + len(jy):]
        self.f_red = red_all[:n]
        self.jy_red = red_all[n:n + len(jy)]
        self.jp_red = red_all[n + len(jy):]
        # state-only form
        self.repl_f, self.f_only = cse(f_exprs, symbols=sympy.numbered_symbols('x_'),
                                       optimizations='basic')
        # per row pattern
        self.jy_rows = [[] for _ in range(spec.n_vars)]
        for e, (r, c, _) in enumerate(jy):
            self.jy_rows[r].append((e, c))
        self.jp_rows = [[] for _ in range(spec.n_vars)]
        for e, (r, c, _) in enumerate(jp):
            self.jp_rows[r].append((e, c))
        self._order_slots_by_structure()

    def _order_slots_by_structure(self):
        """The row-lane class finder (emit_rowlane.find_classes) compares the expression bundles of rows slot by slot,
        and the slots of a row are its J_y entries in COLUMN order: two rows with the same kinetics but their
        entries at different column positions (a densely coupled network: the diagonal entry is the i-th of row i)
        count as different classes.  Sorting a row's entries by expression structure first (symbols blanked), cyclic
        column offset second, makes such rows line up.  Applied only when it lowers the number of classes: the slot
        order of every model where column order already works (the cascades, the reference fixtures) stays as it
        was."""
        from . import emit_rowlane
        n = self.spec.n_vars
        if n < 3:
            return
        names = set(self.spec.variables) | set(self.spec.params)
        blank = {Symbol(nm): Symbol('_') for nm in names}
        before = len(emit_rowlane.find_classes(self.spec, self)[0])
        if before <= 1:
            return
        old = [list(r) for r in self.jy_rows]
        self.jy_rows = [sorted(r, key=lambda ec, i=i: (sympy.srepr(self.jy_c[ec[0]].xreplace(blank)), (ec[1] - i) % n))
                        for i, r in enumerate(old)]
        if len(emit_rowlane.find_classes(self.spec, self)[0]) >= before:
            self.jy_rows = old


class _ExprPrinter(C99CodePrinter):
    """C-family expression printer with explicit reciprocals.

    ``rcp`` is a format string for 1/x: the HIP target maps it to the device
    fast reciprocal (v_rcp_f64 + two Newton steps), C and Python to ``1.0/x``.
    """

    def __init__(self, symbol_map, rcp="(1.0/(%s))", lang='c'):
        super().__init__({'strict': False} if 'strict' in C99CodePrinter._default_settings else {})
        self._smap = symbol_map
        self._rcp = rcp
        self._lang = lang

    def doprint(self, expr, assign_to=None):
        # x**(-n) -> RCP(x**n) BEFORE printing: the stock Mul printer would otherwise
        # collect negative powers into an a/b division
        expr = sympy.sympify(expr).replace(
            lambda e: e.is_Pow and e.exp.is_number and e.exp.is_negative,
            lambda e: _RCP(sympy.Pow(e.base, -e.exp)))
        return super().doprint(expr, assign_to)

    def _print_Function(self, expr):
        if expr.func == _RCP:
            return self._rcp % self._print(expr.args[0])
        return super()._print_Function(expr)

    def _print_Symbol(self, expr):
        return self._smap.get(expr.name, expr.name)

    def _print_Rational(self, expr):
        return "(%d.0/%d.0)" % (expr.p, expr.q)

    def _print_Integer(self, expr):
        return "%d.0" % int(expr.p) if int(expr.p) >= 0 else "(%d.0)" % int(expr.p)

    def _mulpow(self, base, n):
        b = self._print(base)
        if not (base.is_Symbol or base.is_Number):
            b = "(%s)" % b
        return "(" + "*".join([b] * n) + ")"

    def _print_Pow(self, expr):
        base, exp = expr.as_base_exp()
        if exp.is_Integer:
            n = int(exp)
            if 1 <= n <= 4:
                return self._mulpow(base, n)
            if -4 <= n <= -1:
                inner = self._print(base) if n == -1 else self._mulpow(base, -n)
                return self._rcp % inner
        if exp == sympy.Rational(1, 2):
            return "sqrt(%s)" % self._print(base)
        if exp == sympy.Rational(-1, 2):
            return self._rcp % ("sqrt(%s)" % self._print(base))
        return "pow(%s, %s)" % (self._print(base), self._print(exp))


def _symbol_map(spec, y_fmt="y[%d]", p_fmt="p[%d]"):
    m = {}
    for i, v in enumerate(spec.variables):
        m[v] = y_fmt % i
    for i, p in enumerate(spec.params):
        m[p] = p_fmt % i
    m['t'] = 't'
    return m


def _fmt_header(spec, comment):
    c = comment
    lines = [
        "%s generated by sysbio_modeling_amd.symbolic.emit -- do not edit" % c,
        "%s model '%s': %d states, %d parameters (%d with sensitivities)"
        % (c, spec.name, spec.n_vars, spec.n_params, spec.n_sens),
        "%s states: %s" % (c, ", ".join(spec.variables)),
        "%s params: %s" % (c, ", ".join(spec.params)),
    ]
    if spec.fixed:
        lines.append("%s fixed : %s" % (c, ", ".join(spec.fixed)))
    return lines


# ----------------------------------------------------------------------------
# Python
# ----------------------------------------------------------------------------
def emit_python(spec: ModelSpec, derived: Derived = None) -> str:
    """Python module source: ``ordered_params``, ``n_vars``, ``model``, ``sens_model``.

    Same callback contract and yout layout as the reference fixtures
    (tests/test_utils/jittable_model.py:5-26, sens_jittable_model.py:1-38).
    """
    d = derived or Derived(spec)
    n, k = spec.n_vars, spec.n_sens
    pr = _ExprPrinter(_symbol_map(spec), rcp="(1.0/(%s))", lang='py')
    pad = "    "
    L = _fmt_header(spec, "#")
    L += ["from math import exp, log, sqrt, pow, sin, cos, tanh", "",
          "ordered_params = [%s]" % ", ".join("'%s'" % p for p in spec.params),
          "sens_params = [%s]" % ", ".join("'%s'" % p for p in spec.sens_params),
          "n_vars = %d" % n, "n_sens = %d" % k, "", ""]
    L += ["def model(y, t, yout, p):"]
    for s, e in d.repl_f:
        L.append(pad + "%s = %s" % (s, pr.doprint(e)))
    for i, e in enumerate(d.f_only):
        L.append(pad + "yout[%d] = (%s)" % (i, pr.doprint(e)))
    L += ["", "", "def sens_model(y, t, yout, p):"]
    for s, e in d.repl_all:
        L.append(pad + "%s = %s" % (s, pr.doprint(e)))
    for i, e in enumerate(d.f_red):
        L.append(pad + "yout[%d] = (%s)" % (i, pr.doprint(e)))
    for e_idx, e in enumerate(d.jy_red):
        L.append(pad + "jy_%d = %s" % (e_idx, pr.doprint(e)))
    for e_idx, e in enumerate(d.jp_red):
        L.append(pad + "jp_%d = %s" % (e_idx, pr.doprint(e)))
    for i in range(n):
        for j in range(k):
            terms = ["jy_%d*y[%d]" % (e_idx, n + c * k + j) for e_idx, c in d.jy_rows[i]]
            terms += ["jp_%d" % e_idx for e_idx, c in d.jp_rows[i] if c == j]
            L.append(pad + "yout[%d] = (%s)" % (n + i * k + j, " + ".join(terms) if terms else "0.0"))
    L.append("")
    return "\n".join(L)


def emit_python_jacobians(spec: ModelSpec, derived: Derived = None) -> str:
    """Python source of ``model_jac`` and ``sens_model_jac``: the analytic ODE Jacobians the reference's
    ``make_ode_model_jacobian`` prints (symbolic/sympy_tools.py:219-269) and its ``OdeModel`` hands to LSODA as
    ``Dfun`` with ``col_deriv=True`` (model/ode_model.py:114-120,154-160).

    Contract: ``jac(y, t, jacout, p) -> None`` writes the NON-ZERO entries of ``jacout[b, a] = d f_a / d y_b``
    (derivatives down the columns) into a caller-allocated zero matrix, (n, n) for ``model_jac`` and (N, N),
    N = n + n*k, for ``sens_model_jac`` -- as the reference's generated code does (zero entries are comments
    there, :238-241).  With S' = J_y S + J_p the augmented Jacobian has three blocks:

        d f_i / d y_m                      = J_y[i, m]
        d S'[i, j] / d S[l, j]             = J_y[i, l]                               (same column j only)
        d S'[i, j] / d y_m                 = sum_l dJ_y[i, l]/dy_m * S[l, j] + dJ_p[i, j]/dy_m

    The second derivatives are taken on the sparse J_y / J_p entries, not on n*k expanded equations.
    """
    d = derived or Derived(spec)
    n, k = spec.n_vars, spec.n_sens
    pr = _ExprPrinter(_symbol_map(spec), rcp="(1.0/(%s))", lang='py')
    pad = "    "
    ysyms = [Symbol(v) for v in spec.variables]
    # second derivatives: (i, l, m) -> d J_y[i,l] / d y_m ;  (i, j, m) -> d J_p[i,j] / d y_m
    hy, hp = [], []
    for e_idx, (i, l, expr) in enumerate(d.jy):
        for m, ym in enumerate(ysyms):
            if ym in expr.free_symbols:
                dd = sympy.diff(expr, ym)
                if dd != 0:
                    hy.append((i, l, m, _canon_rcp(_cheapest(dd))))
    for e_idx, (i, j, expr) in enumerate(d.jp):
        for m, ym in enumerate(ysyms):
            if ym in expr.free_symbols:
                dd = sympy.diff(expr, ym)
                if dd != 0:
                    hp.append((i, j, m, _canon_rcp(_cheapest(dd))))
    exprs = list(d.jy_c) + [h[3] for h in hy] + [h[3] for h in hp]
    repl, red = cse(exprs, symbols=sympy.numbered_symbols('x_'), optimizations='basic')
    jy_red = red[:len(d.jy)]
    hy_red = red[len(d.jy):len(d.jy) + len(hy)]
    hp_red = red[len(d.jy) + len(hy):]
    L = ["from math import exp, log, sqrt, pow, sin, cos, tanh", "", "",
         "def model_jac(y, t, jacout, p):"]
    repl_y, red_y = cse(list(d.jy_c), symbols=sympy.numbered_symbols('x_'), optimizations='basic')
    for s_, e in repl_y:
        L.append(pad + "%s = %s" % (s_, pr.doprint(e)))
    for (i, m, _), e in zip(d.jy, red_y):
        L.append(pad + "jacout[%d, %d] = (%s)" % (m, i, pr.doprint(e)))
    if not d.jy:
        L.append(pad + "pass")
    L += ["", "", "def sens_model_jac(y, t, jacout, p):"]
    for s_, e in repl:
        L.append(pad + "%s = %s" % (s_, pr.doprint(e)))
    for e_idx, ((i, l, _), e) in enumerate(zip(d.jy, jy_red)):
        L.append(pad + "jy_%d = %s" % (e_idx, pr.doprint(e)))
        L.append(pad + "jacout[%d, %d] = jy_%d" % (l, i, e_idx))
        if k:
            L.append(pad + "for j in range(%d):" % k)
            L.append(pad * 2 + "jacout[%d + j, %d + j] = jy_%d" % (n + l * k, n + i * k, e_idx))
    # lower-left block, grouped by (equation row i, state m)
    by_im = OrderedDict()
    for (i, l, m, _), e in zip(hy, hy_red):
        by_im.setdefault((i, m), dict(hy=[], hp=[]))['hy'].append((l, e))
    for (i, j, m, _), e in zip(hp, hp_red):
        by_im.setdefault((i, m), dict(hy=[], hp=[]))['hp'].append((j, e))
    for q, ((i, m), terms) in enumerate(by_im.items()):
        names = []
        for t_idx, (l, e) in enumerate(terms['hy']):
            L.append(pad + "h_%d_%d = %s" % (q, t_idx, pr.doprint(e)))
            names.append(("h_%d_%d" % (q, t_idx), l))
        if names and k:
            L.append(pad + "for j in range(%d):" % k)
            L.append(pad * 2 + "jacout[%d, %d + j] = %s" % (
                m, n + i * k, " + ".join("%s*y[%d + j]" % (nm, n + l * k) for nm, l in names)))
        elif k and terms['hp']:
            L.append(pad + "for j in range(%d):" % k)
            L.append(pad * 2 + "jacout[%d, %d + j] = 0.0" % (m, n + i * k))
        for j, e in terms['hp']:
            L.append(pad + "jacout[%d, %d] += (%s)" % (m, n + i * k + j, pr.doprint(e)))
    if not d.jy and not by_im:
        L.append(pad + "pass")
    L.append("")
    return "\n".join(L)


# ----------------------------------------------------------------------------
# C (oracle RHS / CPU baseline RHS; plays the role numba plays for the reference)
# ----------------------------------------------------------------------------
def emit_c(spec: ModelSpec, derived: Derived = None) -> str:
    d = derived or Derived(spec)
    n, k = spec.n_vars, spec.n_sens
    pr = _ExprPrinter(_symbol_map(spec), rcp="(1.0/(%s))", lang='c')
    L = _fmt_header(spec, "//")
    L += ["#include <math.h>", "",
          "int sbm_n_vars(void) { return %d; }" % n,
          "int sbm_n_params(void) { return %d; }" % spec.n_params,
          "int sbm_n_sens(void) { return %d; }" % k, ""]
    L += ["void sbm_rhs(const double* y, double t, double* yout, const double* p) {", "  (void)t;"]
    for s, e in d.repl_f:
        L.append("  const double %s = %s;" % (s, pr.doprint(e)))
    for i, e in enumerate(d.f_only):
        L.append("  yout[%d] = %s;" % (i, pr.doprint(e)))
    L += ["}", ""]
    nnz_y, nnz_p = max(len(d.jy), 1), max(len(d.jp), 1)
    L += ["static const int JY_ROW[%d] = {%s};" % (nnz_y, ", ".join(str(r) for r, _, _ in d.jy) or "0"),
          "static const int JY_COL[%d] = {%s};" % (nnz_y, ", ".join(str(c) for _, c, _ in d.jy) or "0"),
          "static const int JP_ROW[%d] = {%s};" % (nnz_p, ", ".join(str(r) for r, _, _ in d.jp) or "0"),
          "static const int JP_COL[%d] = {%s};" % (nnz_p, ", ".join(str(c) for _, c, _ in d.jp) or "0"), ""]
    L += ["// augmented system: yout[n + i*k + j] = J_p[i][j] + sum_m J_y[i][m] * y[n + m*k + j]",
          "void sbm_sens_rhs(const double* y, double t, double* yout, const double* p) {", "  (void)t;",
          "  enum { N = %d, K = %d, NNZ_Y = %d, NNZ_P = %d };" % (n, k, len(d.jy), len(d.jp)),
          "  double jy[%d], jp[%d];" % (nnz_y, nnz_p)]
    for s, e in d.repl_all:
        L.append("  const double %s = %s;" % (s, pr.doprint(e)))
    for i, e in enumerate(d.f_red):
        L.append("  yout[%d] = %s;" % (i, pr.doprint(e)))
    for e_idx, e in enumerate(d.jy_red):
        L.append("  jy[%d] = %s;" % (e_idx, pr.doprint(e)))
    for e_idx, e in enumerate(d.jp_red):
        L.append("  jp[%d] = %s;" % (e_idx, pr.doprint(e)))
    L += ["  for (int i = N; i < N + N * K; ++i) yout[i] = 0.0;",
          "  for (int e = 0; e < NNZ_Y; ++e) {",
          "    const double a = jy[e]; const double* src = y + N + JY_COL[e] * K; double* dst = yout + N + JY_ROW[e] * K;",
          "    for (int j = 0; j < K; ++j) dst[j] += a * src[j];",
          "  }",
          "  for (int e = 0; e < NNZ_P; ++e) yout[N + JP_ROW[e] * K + JP_COL[e]] += jp[e];",
          "}", ""]
    return "\n".join(L)


# ----------------------------------------------------------------------------
# HIP device struct
# ----------------------------------------------------------------------------
def emit_hip(spec: ModelSpec, derived: Derived = None) -> str:
    """Header defining ``struct SbmModel`` consumed by csrc/sbm_integrators.hpp.

    ``eval_f``   : state RHS only (state-only kernels, one trajectory per lane).
    ``eval_jac`` : state RHS + non-zeros of J_y and J_p -- the part that is
                   identical for every sensitivity column of a trajectory.
    ``apply_col``: dz = J_y z + J_p[:, scol] for ONE column held in registers;
                   fully unrolled with static indices so that ``z``/``dz`` stay
                   in VGPRs.
    """
    d = derived or Derived(spec)
    n, k = spec.n_vars, spec.n_sens
    pr = _ExprPrinter(_symbol_map(spec), rcp="SBM_RCP(%s)", lang='hip')
    nnz_y, nnz_p = max(len(d.jy), 1), max(len(d.jp), 1)
    L = _fmt_header(spec, "//")
    rl_tables, rl_meta = emit_rowlane.emit_rowlane_tables(spec, d, None)
    rg_tables, rg_layout = emit_rowgroup.emit_tables(spec, d, tag='RG0')
    rg_tables1, rg_layout1 = emit_rowgroup.emit_tables(spec, d, tag='RG1', latency=True)
    rg_tables2, rg_layout2 = emit_rowgroup.emit_tables(spec, d, tag='RG2', dop853=True)
    rg_tables = rg_tables + rg_tables1 + rg_tables2
    im_members, im_meta = emit_implicit.emit_members(spec, d)
    im_tables = emit_implicit.emit_tables(spec, d, im_meta)
    L += ["#pragma once", ""] + rl_tables + rg_tables + im_tables + [
          "struct SbmModel {",
          "  static constexpr int NV = %d;      // state variables" % n,
          "  static constexpr int NP = %d;      // model parameters (length of p)" % spec.n_params,
          "  static constexpr int NK = %d;      // non-fixed parameters = sensitivity columns" % k,
          "  static constexpr int NNZ_JY = %d;  // non-zeros of df/dy" % len(d.jy),
          "  static constexpr int NNZ_JP = %d;  // non-zeros of df/dp" % len(d.jp),
          "  static constexpr int NJY = %d;" % nnz_y,
          "  static constexpr int NJP = %d;" % nnz_p,
          "  static constexpr const char* NAME = \"%s\";" % spec.name, ""]
    # eval_f
    L += ["  template <class PA>",
          "  __device__ __forceinline__ static void eval_f(double t, const double (&y)[NV], const PA& p,",
          "                                                double (&f)[NV]) {", "    (void)t;"]
    for s, e in d.repl_f:
        L.append("    const double %s = %s;" % (s, pr.doprint(e)))
    for i, e in enumerate(d.f_only):
        L.append("    f[%d] = %s;" % (i, pr.doprint(e)))
    L += ["  }", ""]
    # eval_jac
    L += ["  template <class PA>",
          "  __device__ __forceinline__ static void eval_jac(double t, const double (&y)[NV], const PA& p,",
          "                                                  double (&f)[NV], double (&jy)[NJY], double (&jp)[NJP]) {",
          "    (void)t;"]
    for s, e in d.repl_all:
        L.append("    const double %s = %s;" % (s, pr.doprint(e)))
    for i, e in enumerate(d.f_red):
        L.append("    f[%d] = %s;" % (i, pr.doprint(e)))
    for e_idx, e in enumerate(d.jy_red):
        L.append("    jy[%d] = %s;" % (e_idx, pr.doprint(e)))
    for e_idx, e in enumerate(d.jp_red):
        L.append("    jp[%d] = %s;" % (e_idx, pr.doprint(e)))
    if not d.jy:
        L.append("    jy[0] = 0.0;")
    if not d.jp:
        L.append("    jp[0] = 0.0;")
    L += ["  }", ""]
    # apply_col
    L += ["  // one sensitivity column: dz = J_y z + J_p[:, scol]   (scol < 0: no J_p term)",
          "  __device__ __forceinline__ static void apply_col(const double (&jy)[NJY], const double (&jp)[NJP],",
          "                                                   int scol, const double (&z)[NV], double (&dz)[NV]) {"]
    for i in range(n):
        sel = "0.0"
        for e_idx, c in reversed(d.jp_rows[i]):
            sel = "SBM_PICK(scol, %d, jp[%d], %s)" % (c, e_idx, sel)
        expr = sel
        for e_idx, c in d.jy_rows[i]:
            expr = "fma(jy[%d], z[%d], %s)" % (e_idx, c, expr)
        L.append("    dz[%d] = %s;" % (i, expr))
    L += ["  }", ""]
    # eval_col: the two above fused row by row.  Same arithmetic; every J_y / J_p entry is
    # consumed right after it is computed, which keeps the live set (and the VGPR count of
    # the 7-stage Dormand-Prince kernel) down.  Each CSE temporary is emitted before its first use.
    L += ["  // fused eval_jac + apply_col for ONE column per lane: scol < 0 -> dz = f (state lane),",
          "  // scol in [0, NK) -> dz = J_y z + J_p[:, scol]",
          "  template <class PA>",
          "  __device__ __forceinline__ static void eval_col(double t, const double (&y)[NV], const PA& p, int scol,",
          "                                                  const double (&z)[NV], double (&dz)[NV]) {",
          "    (void)t;", "    const bool state_lane = scol < 0;"]
    temp_expr = OrderedDict((s, e) for s, e in d.repl_all)
    emitted = set()

    def need(expr, out):
        for s in sorted(expr.free_symbols, key=lambda x: x.name):
            if s in temp_expr and s not in emitted and s not in out:
                need(temp_expr[s], out)
                out.append(s)

    for i in range(n):
        exprs = [d.f_red[i]] + [d.jy_red[e] for e, _ in d.jy_rows[i]] + [d.jp_red[e] for e, _ in d.jp_rows[i]]
        todo = []
        for e in exprs:
            need(e, todo)
        for s in todo:
            L.append("    const double %s = %s;" % (s, pr.doprint(temp_expr[s])))
            emitted.add(s)
        sel = "0.0"
        for e_idx, c in reversed(d.jp_rows[i]):
            sel = "SBM_PICK(scol, %d, %s, %s)" % (c, pr.doprint(d.jp_red[e_idx]), sel)
        expr = sel
        for e_idx, c in d.jy_rows[i]:
            expr = "fma(%s, z[%d], %s)" % (pr.doprint(d.jy_red[e_idx]), c, expr)
        L.append("    dz[%d] = SBM_SEL(state_lane, %s, %s);" % (i, pr.doprint(d.f_red[i]), expr))
    L += ["  }", ""]

    # ---- row-lane form: SIMD across isomorphic equations (emit_rowlane.py) ----
    L += emit_rowlane.emit_rowlane_members(spec, d, rl_meta,
                                           lambda smap: _ExprPrinter(smap, rcp="SBM_RCP(%s)", lang='hip'))
    L += [""] + emit_rowgroup.emit_members(spec, d, rg_layout, tag='RG0')
    L += emit_rowgroup.emit_members(spec, d, rg_layout1, tag='RG1', alias_of='RG0')
    L += emit_rowgroup.emit_members(spec, d, rg_layout2, tag='RG2', alias_of='RG0')
    L += [""] + im_members + [
        "  __device__ __forceinline__ static int im_rstart(int row) { return SBM_IM_RSTART[IM_TRI ? row : 0]; }",
        "  __device__ __forceinline__ static int im_diagslot(int row) { return SBM_IM_DIAGSLOT[IM_TRI ? row : 0]; }",
        "  __device__ __forceinline__ static int im_mfpos(int slot, int row) { return SBM_IM_MFPOS[IM_TRI ? slot * NV + row : 0]; }"]
    L += ["};", ""]
    return "\n".join(L)
