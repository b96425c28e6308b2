"""Plain right-hand-side callables -> ModelSpec, read from their SOURCE.

The reference's ``OdeModel`` takes any ``f(y, t, yout, p)`` (model/ode_model.py:27-44) and its tests build it from
the two emitted fixture modules (tests/test_OdeModel.py:12-18: ``OdeModel(jittable_model.model,
sens_jittable_model.sens_model, n_vars, ordered_params)``).  A Python callable cannot run on a GPU, but what the
reference's own generator writes (symbolic/sympy_tools.py:100-111,185-195) is straight-line code of a fixed shape:

    <name> = p[<i>]            parameter i
    <name> = y[<i>]            state variable i (i < n_vars; in a sens_model the rest are sensitivity variables)
    <name> = <expression>      optional intermediate (rate laws, when a user wrote the function by hand)
    yout[<i>] = (<expression>) d y_i / dt

That form is parsed here (``ast``, no execution of the function) into the symbolic equations the HIP emitters start
from; the sensitivity system is then DERIVED, not read -- and checked numerically against the ``sens_model`` that
was handed in, which also tells which parameters carry sensitivity columns.

A hand-written right-hand side may also use (round 4) what unrolls to that form at parse time:

    for i in range(<static ints>): ...        loops with static bounds (nested ones too), unrolled
    y[<int expr>], p[<int expr>], yout[<int expr>]   indices that are integer expressions of loop variables / integer names
    if <comparison of static ints>: ... else: ...     branches decided at parse time (``if i == 0``)
    x += <expr> (and -=, *=, /=)              accumulators

Anything that depends on the DATA -- a branch on y or p, a while loop, a call other than an elementary function -- is
still refused: there is no CPU fallback to hide behind, and a right-hand side that is not smooth has no sensitivities.
"""
from __future__ import annotations

import ast
import hashlib
import inspect
import textwrap
from collections import OrderedDict

import numpy as np
import sympy

from .emit import ModelSpec

_FUNCS = {'exp': sympy.exp, 'log': sympy.log, 'sqrt': sympy.sqrt, 'tanh': sympy.tanh, 'sin': sympy.sin,
          'cos': sympy.cos, 'tan': sympy.tan, 'pow': sympy.Pow, 'power': sympy.Pow, 'abs': sympy.Abs,
          'fabs': sympy.Abs, 'sinh': sympy.sinh, 'cosh': sympy.cosh}


class IngestError(TypeError):
    pass


def _source_of(fn):
    fn = getattr(fn, 'py_func', fn)           # a numba dispatcher keeps the Python function
    fn = getattr(fn, '__wrapped__', fn)
    try:
        src = inspect.getsource(fn)
    except (OSError, TypeError) as e:
        raise IngestError("the source of %r is not available (%s)" % (fn, e))
    return textwrap.dedent(src), fn


def _static_int(node, ienv):
    """Value of an integer expression of literals and the names in ``ienv`` (loop variables, integer constants),
    or None if ``node`` is not one."""
    if isinstance(node, ast.Constant):
        return node.value if isinstance(node.value, int) and not isinstance(node.value, bool) else None
    if isinstance(node, ast.Name):
        return ienv.get(node.id)
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _static_int(node.operand, ienv)
        return None if v is None else (-v if isinstance(node.op, ast.USub) else v)
    if isinstance(node, ast.BinOp):
        a, b = _static_int(node.left, ienv), _static_int(node.right, ienv)
        if a is None or b is None:
            return None
        if isinstance(node.op, ast.Add):
            return a + b
        if isinstance(node.op, ast.Sub):
            return a - b
        if isinstance(node.op, ast.Mult):
            return a * b
        if isinstance(node.op, ast.FloorDiv) and b != 0:
            return a // b
        if isinstance(node.op, ast.Mod) and b != 0:
            return a % b
    return None


def _static_bool(node, ienv):
    """Truth value of a condition over static integers (comparisons, and / or / not), or None."""
    if isinstance(node, ast.Compare):
        vals = [_static_int(n, ienv) for n in [node.left] + list(node.comparators)]
        if any(v is None for v in vals):
            return None
        ops = {ast.Eq: lambda a, b: a == b, ast.NotEq: lambda a, b: a != b, ast.Lt: lambda a, b: a < b,
               ast.LtE: lambda a, b: a <= b, ast.Gt: lambda a, b: a > b, ast.GtE: lambda a, b: a >= b}
        out = True
        for op, a, b in zip(node.ops, vals[:-1], vals[1:]):
            if type(op) not in ops:
                return None
            out = out and ops[type(op)](a, b)
        return out
    if isinstance(node, ast.BoolOp):
        vals = [_static_bool(v, ienv) for v in node.values]
        if any(v is None for v in vals):
            return None
        return all(vals) if isinstance(node.op, ast.And) else any(vals)
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.Not):
        v = _static_bool(node.operand, ienv)
        return None if v is None else (not v)
    return None


def _index_of(node, array, ienv=None):
    """i for ``array[i]`` with a non-negative integer i -- a literal, or an integer expression of the names in
    ``ienv`` -- else None."""
    if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id == array:
        sl = node.slice
        if isinstance(sl, ast.Index):       # python < 3.9
            sl = sl.value
        v = _static_int(sl, ienv or {})
        if v is not None and v >= 0:
            return v
    return None


class _ToSympy(ast.NodeVisitor):
    def __init__(self, names, arrays, ienv=None):
        self.names = names          # local name -> sympy expression
        self.arrays = arrays        # (y symbols, p symbols)
        self.ienv = ienv if ienv is not None else {}     # static integers: loop variables, integer constants

    def visit(self, node):
        m = getattr(self, 'visit_' + type(node).__name__, None)
        if m is None:
            raise IngestError("unsupported syntax in a right-hand side: %s" % ast.dump(node)[:80])
        return m(node)

    def visit_Constant(self, node):
        if isinstance(node.value, bool) or not isinstance(node.value, (int, float)):
            raise IngestError("unsupported constant %r" % (node.value,))
        if isinstance(node.value, int) or float(node.value).is_integer():
            return sympy.Integer(int(node.value))       # 1.0 is 1: keeps sqrt(c + 1.0) and sqrt(c + 1) one expression
        return sympy.Float(node.value)

    def visit_Name(self, node):
        if node.id in self.ienv:
            return sympy.Integer(self.ienv[node.id])
        if node.id in self.names:
            return self.names[node.id]
        if node.id == 't':
            return sympy.Symbol('t')
        raise IngestError("name %r is used before it is assigned" % node.id)

    def visit_Subscript(self, node):
        for arr, table in (('y', self.arrays[0]), ('p', self.arrays[1])):
            i = _index_of(node, arr, self.ienv)
            if i is not None:
                if i >= len(table):
                    raise IngestError("%s[%d] is out of range" % (arr, i))
                return table[i]
        raise IngestError("only y[<int>] and p[<int>] may be indexed (the index: a non-negative integer expression of "
                          "literals and loop variables)")

    def visit_UnaryOp(self, node):
        v = self.visit(node.operand)
        if isinstance(node.op, ast.USub):
            return -v
        if isinstance(node.op, ast.UAdd):
            return v
        raise IngestError("unsupported unary operator")

    def visit_BinOp(self, node):
        a, b = self.visit(node.left), self.visit(node.right)
        if isinstance(node.op, ast.Add):
            return a + b
        if isinstance(node.op, ast.Sub):
            return a - b
        if isinstance(node.op, ast.Mult):
            return a * b
        if isinstance(node.op, ast.Div):
            return a / b
        if isinstance(node.op, ast.Pow):
            return a ** b
        raise IngestError("unsupported binary operator %s" % type(node.op).__name__)

    def visit_Call(self, node):
        f = node.func
        name = f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else None)
        if name not in _FUNCS or node.keywords:
            raise IngestError("unsupported call %r (elementary functions only)" % name)
        return _FUNCS[name](*[self.visit(a) for a in node.args])


def parse_rhs(fn, n_y, n_p):
    """{equation index: sympy expression in the symbols y0.., p0.., t}, parsed from ``fn``'s source."""
    src, fn = _source_of(fn)
    try:
        tree = ast.parse(src)
    except SyntaxError as e:          # e.g. a lambda inside a larger statement: its "source" is a fragment
        raise IngestError("the source of %r does not parse on its own (%s)" % (fn, e))
    fdef = next((n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)), None)
    if fdef is None:
        raise IngestError("no function definition found in the source of %r" % fn)
    args = [a.arg for a in fdef.args.args]
    if len(args) != 4:
        raise IngestError("a right-hand side takes (y, t, yout, p); %s takes %s" % (fdef.name, args))
    y_name, t_name, out_name, p_name = args
    ys = [sympy.Symbol('__y%d' % i) for i in range(n_y)]
    ps = [sympy.Symbol('__p%d' % i) for i in range(n_p)]
    names = {}
    conv = _ToSympy(names, (ys, ps))
    # the argument names need not be y / t / yout / p: rename through the tables the visitor consults
    rename = {y_name: 'y', p_name: 'p', t_name: 't'}

    class _Ren(ast.NodeTransformer):
        def visit_Name(self, node):
            return ast.copy_location(ast.Name(id=rename.get(node.id, node.id), ctx=node.ctx), node)
    eqs = {}
    ienv = conv.ienv
    budget = [200000]          # statements executed while unrolling (a typo in a range() must not hang the parser)

    def run(stmts):
        for stmt in stmts:
            budget[0] -= 1
            if budget[0] < 0:
                raise IngestError("%s unrolls to more than 200000 statements" % fdef.name)
            if isinstance(stmt, ast.Expr) and isinstance(stmt.value, ast.Constant):
                continue                                  # docstring
            if isinstance(stmt, ast.Pass):
                continue
            if isinstance(stmt, ast.Return) and stmt.value is None:
                continue
            if isinstance(stmt, ast.For):
                # for <name> in range(<static ints>): unrolled
                it = stmt.iter
                if not (isinstance(stmt.target, ast.Name) and isinstance(it, ast.Call) and isinstance(it.func, ast.Name)
                        and it.func.id == 'range' and 1 <= len(it.args) <= 3 and not it.keywords and not stmt.orelse):
                    raise IngestError("only `for <name> in range(<static integers>)` loops are understood (line %d of %s)"
                                      % (stmt.lineno, fdef.name))
                bounds = [_static_int(_Ren().visit(a_), ienv) for a_ in it.args]
                if any(b is None for b in bounds):
                    raise IngestError("the bounds of a loop must be integers known when the function is read -- literals, "
                                      "integer names, outer loop variables (line %d of %s)" % (stmt.lineno, fdef.name))
                saved = ienv.get(stmt.target.id)
                for v in range(*bounds):
                    ienv[stmt.target.id] = v
                    run(stmt.body)
                if saved is None:
                    ienv.pop(stmt.target.id, None)
                else:
                    ienv[stmt.target.id] = saved
                continue
            if isinstance(stmt, ast.If):
                cond = _static_bool(_Ren().visit(stmt.test), ienv)
                if cond is None:
                    raise IngestError("a branch must be decided by loop variables / integer constants (`if i == 0:`); one that "
                                      "depends on y, p or t has no place in a differentiable right-hand side (line %d of %s)"
                                      % (stmt.lineno, fdef.name))
                run(stmt.body if cond else stmt.orelse)
                continue
            if isinstance(stmt, ast.AugAssign):
                ops = {ast.Add: lambda a_, b_: a_ + b_, ast.Sub: lambda a_, b_: a_ - b_, ast.Mult: lambda a_, b_: a_ * b_,
                       ast.Div: lambda a_, b_: a_ / b_}
                if type(stmt.op) not in ops:
                    raise IngestError("unsupported augmented assignment (line %d of %s)" % (stmt.lineno, fdef.name))
                value = conv.visit(_Ren().visit(stmt.value))
                tgt = stmt.target
                i = _index_of(tgt, out_name, ienv)
                if i is not None:
                    if i not in eqs:
                        raise IngestError("%s[%d] is updated before it is assigned (line %d of %s)" % (out_name, i, stmt.lineno, fdef.name))
                    eqs[i] = ops[type(stmt.op)](eqs[i], value)
                elif isinstance(tgt, ast.Name) and tgt.id in names:
                    names[tgt.id] = ops[type(stmt.op)](names[tgt.id], value)
                else:
                    raise IngestError("unsupported augmented assignment target (line %d of %s)" % (stmt.lineno, fdef.name))
                continue
            if not isinstance(stmt, ast.Assign) or len(stmt.targets) != 1:
                raise IngestError("only assignments, static loops and static branches are understood in a right-hand side "
                                  "(line %d of %s)" % (stmt.lineno, fdef.name))
            tgt = stmt.targets[0]
            # an integer constant (n = 20) may bound a loop or index an array later on
            if isinstance(tgt, ast.Name):
                iv = _static_int(_Ren().visit(stmt.value), ienv)
                if iv is not None:
                    ienv[tgt.id] = iv
                    names.pop(tgt.id, None)
                    continue
            value = conv.visit(_Ren().visit(stmt.value))
            i = _index_of(tgt, out_name, ienv)
            if i is not None:
                eqs[i] = value
            elif isinstance(tgt, ast.Name):
                names[tgt.id] = value
                ienv.pop(tgt.id, None)
            else:
                raise IngestError("unsupported assignment target (line %d of %s)" % (stmt.lineno, fdef.name))
    run(fdef.body)
    return eqs, ys, ps, src


def _sens_columns(sens_model, n_vars, n_params):
    """k such that ``sens_model`` is a right-hand side of n_vars + n_vars * k equations: the size at which a call
    at a probe point succeeds and writes every output (a generated body indexes y / yout with literals, a hand-written
    one may use slices: either way a wrong size raises or leaves outputs unwritten)."""
    fn = getattr(sens_model, 'py_func', sens_model)
    rng = np.random.default_rng(99)
    p = rng.uniform(0.2, 1.2, n_params)
    for k in range(n_params, -1, -1):
        N = n_vars * (1 + k)
        out = np.full(N, np.nan)
        try:
            fn(rng.uniform(0.2, 1.2, N), 0.37, out, p)
        except (IndexError, ValueError):
            continue
        if np.all(np.isfinite(out)):
            return k
    raise IngestError("sens_model does not evaluate as a system of n_vars + n_vars * k equations for any k <= %d" % n_params)


_CACHE = {}


def spec_from_callables(model, sens_model, n_vars, param_order, name='Model'):
    """ModelSpec of a plain ``model(y, t, yout, p)`` in the reference's emitted form.  ``sens_model`` (optional) fixes
    which parameters have sensitivity columns and is checked against the derived sensitivity system."""
    param_order = list(param_order)
    eqs, ys, ps, src = parse_rhs(model, n_vars, len(param_order))
    if sorted(eqs) != list(range(n_vars)):
        raise IngestError("model writes yout%s, expected yout[0..%d]" % (sorted(eqs), n_vars - 1))
    key = hashlib.sha1((src + repr((n_vars, param_order))).encode()).hexdigest()
    sens_src = _source_of(sens_model)[0] if sens_model is not None else ''
    key += hashlib.sha1(sens_src.encode()).hexdigest()
    if key in _CACHE:
        return _CACHE[key]
    var_names = ['y%d' % i for i in range(n_vars)]
    # keep the caller's parameter names where they are valid identifiers that cannot clash
    par_names = [p if (p.isidentifier() and p not in var_names and p != 't') else 'p%d' % i
                 for i, p in enumerate(param_order)]
    sub = dict(zip(ys, [sympy.Symbol(v) for v in var_names]))
    sub.update(zip(ps, [sympy.Symbol(p) for p in par_names]))
    equations = OrderedDict((var_names[i], eqs[i].subs(sub)) for i in range(n_vars))
    fixed = []
    if sens_model is not None:
        k = _sens_columns(sens_model, n_vars, len(param_order))
        if k < len(param_order):
            fixed = _find_fixed(equations, var_names, par_names, sens_model, n_vars, k)
    spec = ModelSpec(name=name, variables=var_names, params=par_names, equations=equations, fixed=fixed)
    if par_names != param_order:
        spec.param_aliases = dict(zip(par_names, param_order))
    if sens_model is not None:
        _check_sens_model(spec, sens_model)
    _CACHE[key] = spec
    return spec


def _derived_sens_rhs(spec, y, S, p):
    """dS/dt = J_y S + J_p of ``spec`` at a point, dense numpy (checker for the handed-in sens_model)."""
    vs = [sympy.Symbol(v) for v in spec.variables]
    prm = [sympy.Symbol(q) for q in spec.params]
    sp = [sympy.Symbol(q) for q in spec.sens_params]
    f = sympy.Matrix([spec.equations[v] for v in spec.variables])
    at = dict(zip(vs, y))
    at.update(zip(prm, p))
    at[sympy.Symbol('t')] = 0.37
    Jy = np.array(f.jacobian(vs).subs(at).evalf(30), dtype=float)
    Jp = np.array(f.jacobian(sp).subs(at).evalf(30), dtype=float) if sp else np.zeros((len(vs), 0))
    fv = np.array(f.subs(at).evalf(30), dtype=float).ravel()
    return fv, Jy @ S + Jp


def _probe(spec):
    rng = np.random.default_rng(12345)
    n, k = spec.n_vars, spec.n_sens
    return rng.uniform(0.2, 1.2, n), rng.uniform(-0.5, 0.5, (n, k)), rng.uniform(0.2, 1.2, spec.n_params)


def _call_sens(sens_model, spec, y, S, p):
    n, k = spec.n_vars, spec.n_sens
    z = np.concatenate([y, S.ravel()])
    out = np.zeros(n + n * k)
    getattr(sens_model, 'py_func', sens_model)(z, 0.37, out, p)
    return out[:n], out[n:].reshape(n, k)


def _check_sens_model(spec, sens_model):
    y, S, p = _probe(spec)
    try:
        f_user, dS_user = _call_sens(sens_model, spec, y, S, p)
    except Exception as e:   # noqa: BLE001
        raise IngestError("sens_model could not be evaluated at a probe point: %r" % (e,))
    f, dS = _derived_sens_rhs(spec, y, S, p)
    scale = 1.0 + np.abs(dS)
    if not (np.allclose(f_user, f, rtol=1e-9, atol=1e-12) and np.all(np.abs(dS_user - dS) <= 1e-8 * scale)):
        raise IngestError("sens_model does not agree with the sensitivity system derived from model (layout "
                          "n_vars + i * k + j, parameters in param_order without the fixed ones): the GPU "
                          "integrates the derived system, so the two must match")


def _find_fixed(equations, var_names, par_names, sens_model, n_vars, k):
    """Which len(par_names) - k parameters have no sensitivity column?  Those whose removal makes the derived
    system match ``sens_model`` at a probe point (tried in the order a generator would drop them)."""
    import itertools
    for fixed in itertools.combinations(par_names, len(par_names) - k):
        spec = ModelSpec(name='probe', variables=var_names, params=par_names, equations=equations, fixed=list(fixed))
        try:
            _check_sens_model(spec, sens_model)
            return list(fixed)
        except IngestError:
            continue
    raise IngestError("sens_model has %d sensitivity parameters but no choice of %d fixed parameters reproduces it"
                      % (k, len(par_names) - k))
