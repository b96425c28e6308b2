"""'custom' measurement mappings as compiled expressions.

The reference's ``measurement_to_model_map`` accepts ``('custom', (parameters, map_fn, jacobian_map_fn))`` with two
Python callbacks (project/base_project.py:125-128) that receive the whole simulated trajectory.  Callbacks cannot run
in the assembly kernel; what such mappings compute in practice is a pointwise function of a few model variables at
the sampled time (a ratio of species, a saturating readout, a weighted sum).  Here the observable is given as an
EXPRESSION

    ('custom', 'x4 / (x4 + x9)')                    variables by model name, or y[4] / y4 by index
    ('custom', ({'w': 0.3}, 'w * x4 + (1 - w) * x9'))     with named constants

and compiled -- value and symbolic derivatives d g / d y_k -- into the postfix programs the kernel interprets
(include/sbm.h, SBM_OP_*).  The measure's Jacobian rows are sum_k (dg/dy_k) S[var_k, :], which is what the reference's
``jacobian_map_fn`` has to return for a pointwise map (project/utils.py:29-45 is the k = 1, dg/dy = 1 case).
"""
from __future__ import annotations

import re

import numpy as np
import sympy

OP = dict(END=0, VAR=1, CONST=2, ADD=3, SUB=4, MUL=5, DIV=6, NEG=7, POW=8, POWI=9, EXP=10, LOG=11, SQRT=12, TANH=13,
          SIN=14, COS=15, ABS=16, SIGN=17, TIME=18)
MAX_STACK = 16
_UNARY = {sympy.exp: 'EXP', sympy.log: 'LOG', sympy.tanh: 'TANH', sympy.sin: 'SIN', sympy.cos: 'COS',
          sympy.Abs: 'ABS', sympy.sign: 'SIGN'}


class ObservableError(ValueError):
    pass


def parse_observable(mapping_args, var_names):
    """-> (sympy expression in the symbols v0.., list of model variable indices in the order of those symbols)."""
    constants = {}
    expr = mapping_args
    if isinstance(mapping_args, (tuple, list)):
        if len(mapping_args) == 3 and (callable(mapping_args[1]) or callable(mapping_args[2])):
            raise TypeError("'custom' mappings with Python callbacks (parameters, map_fn, jacobian_map_fn) are read by a "
                            "Project (trace_callback_observable needs an experiment to run them on); compile_observable "
                            "takes an expression of the model variables, ('custom', 'x4 / (x4 + x9)') or "
                            "('custom', ({'w': 0.3}, 'w * x4 + (1 - w) * x9'))")
        if len(mapping_args) != 2 or not isinstance(mapping_args[0], dict):
            raise TypeError("'custom' mapping: expected an expression or (constants dict, expression)")
        constants, expr = mapping_args
    t = sympy.Symbol('t', real=True)
    local = {'t': t}
    slots = {}          # model variable index -> symbol
    def sym_of(i):
        if i < 0 or i >= len(var_names):
            raise ObservableError("'custom' mapping refers to variable %d of %d" % (i, len(var_names)))
        if i not in slots:
            slots[i] = sympy.Symbol('__obs_y%d' % i, real=True)
        return slots[i]
    if isinstance(expr, str):
        text = re.sub(r'\by\[(\d+)\]', lambda m: '__obs_y%s' % m.group(1), expr)
        for i, nm in enumerate(var_names):
            local[nm] = sym_of(i) if re.search(r'\b%s\b' % re.escape(nm), text) else sympy.Symbol(nm)
        for m in set(re.findall(r'__obs_y(\d+)', text)):
            local['__obs_y%s' % m] = sym_of(int(m))
        for k, v in constants.items():
            local[k] = sympy.Float(float(v)) if float(v) != int(float(v)) else sympy.Integer(int(float(v)))
        try:
            e = sympy.sympify(text, locals=local)
        except (sympy.SympifyError, SyntaxError, TypeError) as err:
            raise ObservableError("'custom' mapping %r does not parse: %s" % (expr, err))
    else:
        e = sympy.sympify(expr)
        sub = {}
        for s in e.free_symbols:
            nm = str(s)
            if nm in constants:
                sub[s] = sympy.Float(float(constants[nm]))
            elif nm in var_names:
                sub[s] = sym_of(var_names.index(nm))
            elif nm == 't':
                sub[s] = t
        e = e.subs(sub)
    used = sorted(slots)
    known = set(slots[i] for i in used) | {t}
    unknown = [str(s) for s in e.free_symbols if s not in known]
    if unknown:
        raise ObservableError("'custom' mapping uses unknown names %s (model variables: %s)" % (unknown, list(var_names)))
    used = [i for i in used if slots[i] in e.free_symbols]
    if not used:
        raise ObservableError("'custom' mapping does not depend on any model variable")
    return e, used, [slots[i] for i in used]


def _emit(e, syms, consts, code):
    """append the postfix code of ``e``; returns the stack depth it needs"""
    t = sympy.Symbol('t', real=True)
    if e in syms:
        code += [OP['VAR'], syms.index(e)]
        return 1
    if e == t:
        code.append(OP['TIME'])
        return 1
    if e.is_Number or e.is_NumberSymbol:
        v = float(e)
        if v not in consts:
            consts.append(v)
        code += [OP['CONST'], consts.index(v)]
        return 1
    if e.is_Add or e.is_Mul:
        op = OP['ADD'] if e.is_Add else OP['MUL']
        args = list(e.args)
        if e.is_Mul and args[0] == -1 and len(args) >= 2:
            d = _emit(sympy.Mul(*args[1:]), syms, consts, code)
            code.append(OP['NEG'])
            return d
        depth = _emit(args[0], syms, consts, code)
        for a in args[1:]:
            depth = max(depth, 1 + _emit(a, syms, consts, code))
            code.append(op)
        return depth
    if e.is_Pow:
        base, ex = e.args
        if ex == sympy.Rational(1, 2):
            d = _emit(base, syms, consts, code)
            code.append(OP['SQRT'])
            return d
        if ex.is_Integer and abs(int(ex)) <= 64:
            d = _emit(base, syms, consts, code)
            code += [OP['POWI'], int(ex)]
            return d
        d = _emit(base, syms, consts, code)
        d = max(d, 1 + _emit(ex, syms, consts, code))
        code.append(OP['POW'])
        return d
    if e.func in _UNARY and len(e.args) == 1:
        d = _emit(e.args[0], syms, consts, code)
        code.append(OP[_UNARY[e.func]])
        return d
    raise ObservableError("'custom' mapping: %s is not supported in a compiled observable" % e.func)


def compile_observable(mapping_args, var_names):
    """-> dict(variables=[model variable indices], subprograms=[[int, ...], ...] (value, then one per variable),
    constants=[float, ...], expr=sympy expression, symbols=[...])."""
    e, used, syms = parse_observable(mapping_args, list(var_names))
    consts, subs = [], []
    for target in [e] + [sympy.diff(e, s) for s in syms]:
        code = []
        try:
            depth = _emit(sympy.simplify(target) if target is not e else target, syms, consts, code)
        except ObservableError:
            code = []                      # simplify may rewrite sign() as a Piecewise: take the raw derivative
            depth = _emit(target, syms, consts, code)
        if depth > MAX_STACK:
            raise ObservableError("'custom' mapping needs an evaluation stack deeper than %d" % MAX_STACK)
        code.append(OP['END'])
        subs.append(code)
    return dict(variables=used, subprograms=subs, constants=consts, expr=e, symbols=syms)


def program_tables(measures, compiled):
    """Flatten the compiled observables of a project into the arrays of sbm_project_desc.
    measures: measure name per row; compiled: {measure name: compile_observable(...)} for the custom ones.
    Returns dict(n_programs, row_prog, prog_nvars, prog_sub_off, prog_code, prog_const)."""
    names = sorted(compiled)
    index = {nm: i for i, nm in enumerate(names)}
    consts, code, sub_off, nvars = [], [], [0], []
    for nm in names:
        c = compiled[nm]
        remap = {}
        for k, v in enumerate(c['constants']):
            if v not in consts:
                consts.append(v)
            remap[k] = consts.index(v)
        nvars.append(len(c['variables']))
        for sp in c['subprograms']:
            out, pc = [], 0
            while pc < len(sp):
                op = sp[pc]
                out.append(op)
                if op == OP['CONST']:
                    out.append(remap[sp[pc + 1]])
                    pc += 1
                elif op in (OP['VAR'], OP['POWI']):
                    out.append(sp[pc + 1])
                    pc += 1
                pc += 1
            code.extend(out)
            sub_off.append(len(code))
    return dict(n_programs=len(names), row_prog=np.asarray([index.get(nm, -1) for nm in measures], dtype=np.int32),
                prog_nvars=np.asarray(nvars, dtype=np.int32), prog_sub_off=np.asarray(sub_off, dtype=np.int32),
                prog_code=np.asarray(code, dtype=np.int32), prog_const=np.asarray(consts, dtype=np.float64))


# ----------------------------------------------------------------------------------------------------------------
# 'custom' mappings in the REFERENCE's form: (parameters, map_fn, jacobian_map_fn)
# ----------------------------------------------------------------------------------------------------------------
# The reference calls   map_fn(model_sim, model_timepoints, experiment, measurement, parameters, use_experimental_timepoints)
# -> (mapped_sim, mapped_timepoints)   and   jacobian_map_fn(model_jacobian, model_timepoints, experiment, measurement,
# parameters, use_experimental_timepoints) -> mapped_jacobian   (project/base_project.py:125-128,380-383,461-464;
# project/utils.py:10-89 are the two built-in pairs).  Callbacks cannot run in the assembly kernel, but what they
# compute can be READ OFF them once, on the host, when the project is set up: map_fn is run on a TRACED simulation --
# an array whose entries are symbols Y[i, v] (and T[i] for the time grid), carried through indexing, arithmetic and the
# numpy functions a pointwise map uses -- so every returned row is an expression in the symbols of its grid point.  The
# rows must all be the same function of (y at the row's grid point, t): that expression is compiled like the
# expression form above.  jacobian_map_fn is then CHECKED numerically, as OdeModel checks a sens_model it is handed
# (symbolic/ingest.py): on random inputs it has to return sum_k dg/dy_k * (block of variable k), the derivative the
# kernel will use.  The reference hands jacobian_map_fn only the model Jacobian, so a callback written to that contract
# can express constant coefficients (weighted sums); one that also takes ``model_sim`` (keyword) is given the simulation.
_UFUNC = {'exp': sympy.exp, 'log': sympy.log, 'sqrt': sympy.sqrt, 'tanh': sympy.tanh, 'sin': sympy.sin, 'cos': sympy.cos,
          'absolute': sympy.Abs, 'fabs': sympy.Abs, 'sign': sympy.sign, 'negative': lambda a: -a, 'positive': lambda a: a,
          'square': lambda a: a ** 2, 'reciprocal': lambda a: 1 / a,
          'add': lambda a, b: a + b, 'subtract': lambda a, b: a - b, 'multiply': lambda a, b: a * b,
          'true_divide': lambda a, b: a / b, 'divide': lambda a, b: a / b, 'power': lambda a, b: a ** b,
          'float_power': lambda a, b: a ** b}


def _sym_of(x):
    if isinstance(x, _Traced):
        return x.sym
    a = np.asarray(x)
    out = np.empty(a.shape, dtype=object)
    flat = out.reshape(-1)
    for i, v in enumerate(a.reshape(-1)):
        flat[i] = sympy.Float(float(v)) if float(v) != int(float(v)) else sympy.Integer(int(float(v)))
    return out


def _val_of(x):
    return x.val if isinstance(x, _Traced) else np.asarray(x, dtype=float)


class _Traced(object):
    """A numeric array (``val``) travelling with the symbolic expression of every entry (``sym``, object array)."""
    __array_priority__ = 1000

    def __init__(self, val, sym):
        self.val = np.asarray(val, dtype=float)
        self.sym = np.asarray(sym, dtype=object)
        if self.sym.shape != self.val.shape:
            self.sym = np.broadcast_to(self.sym, self.val.shape)

    shape = property(lambda s: s.val.shape)
    ndim = property(lambda s: s.val.ndim)
    size = property(lambda s: s.val.size)
    dtype = property(lambda s: s.val.dtype)
    T = property(lambda s: _Traced(s.val.T, s.sym.T))

    def __len__(self):
        return len(self.val)

    def __array__(self, dtype=None, copy=None):      # index computations (searchsorted, comparisons) see the numbers
        return self.val if dtype is None else self.val.astype(dtype)

    def __getitem__(self, idx):
        return _Traced(self.val[idx], self.sym[idx])

    def __iter__(self):
        for i in range(len(self.val)):
            yield self[i]

    def _bin(self, other, name, swap=False):
        a, b = (other, self) if swap else (self, other)
        return _apply(name, a, b)

    __add__ = lambda s, o: s._bin(o, 'add')
    __radd__ = lambda s, o: s._bin(o, 'add', True)
    __sub__ = lambda s, o: s._bin(o, 'subtract')
    __rsub__ = lambda s, o: s._bin(o, 'subtract', True)
    __mul__ = lambda s, o: s._bin(o, 'multiply')
    __rmul__ = lambda s, o: s._bin(o, 'multiply', True)
    __truediv__ = lambda s, o: s._bin(o, 'true_divide')
    __rtruediv__ = lambda s, o: s._bin(o, 'true_divide', True)
    __pow__ = lambda s, o: s._bin(o, 'power')
    __rpow__ = lambda s, o: s._bin(o, 'power', True)
    __neg__ = lambda s: _apply('negative', s)
    __pos__ = lambda s: s
    __abs__ = lambda s: _apply('absolute', s)

    def sum(self, axis=None, **kw):
        return _Traced(self.val.sum(axis=axis), np.sum(self.sym, axis=axis))

    def reshape(self, *shape):
        return _Traced(self.val.reshape(*shape), self.sym.reshape(*shape))

    def copy(self):
        return _Traced(self.val.copy(), self.sym.copy())

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        name = ufunc.__name__
        if method == '__call__' and name in _UFUNC and not kwargs.get('out'):
            return _apply(name, *inputs)
        if method == 'reduce' and name == 'add':
            return inputs[0].sum(axis=kwargs.get('axis', 0))
        raise ObservableError("'custom' mapping callback uses numpy.%s%s, which a compiled observable cannot express: give "
                              "the observable as an expression" % (name, '' if method == '__call__' else '.' + method))

    def __array_function__(self, func, types, args, kwargs):
        if func is np.searchsorted or func is np.argsort or func is np.shape or func is np.ndim:
            return func(*[_val_of(a) if isinstance(a, _Traced) else a for a in args], **kwargs)
        if func is np.sum:
            return args[0].sum(**{k: v for k, v in kwargs.items() if k == 'axis'}) if len(args) == 1 else args[0].sum(axis=args[1])
        if func is np.zeros_like or func is np.ones_like:
            return func(_val_of(args[0]), **kwargs)
        if func is np.stack or func is np.column_stack or func is np.array:
            pass
        raise ObservableError("'custom' mapping callback uses numpy.%s on the simulation, which a compiled observable cannot "
                              "express: give the observable as an expression" % func.__name__)


def _apply(name, *inputs):
    f = _UFUNC[name]
    vals = [_val_of(x) for x in inputs]
    with np.errstate(all='ignore'):
        val = getattr(np, name)(*vals)
    syms = np.broadcast_arrays(*[_sym_of(x) for x in inputs])
    out = np.empty(np.shape(val), dtype=object)
    flat = out.reshape(-1)
    its = [s.reshape(-1) for s in syms]
    for i in range(flat.size):
        flat[i] = f(*[it[i] for it in its])
    return _Traced(val, out)


def trace_callback_observable(parameters, map_fn, jac_fn, var_names, experiment, measurement, n_exp_params):
    """The observable behind a reference-style ('custom', (parameters, map_fn, jacobian_map_fn)) mapping, as the
    ``mapping_args`` of the expression form: a sympy expression in the model's variable names (and t).  Raises
    ObservableError / TypeError when the callbacks do not describe a pointwise observable (see the comment above)."""
    import inspect
    from . import utils
    n = len(var_names)
    t_end = experiment.get_unique_timepoints()[-1]
    grid = utils.simulation_grid(t_end)
    T = len(grid)
    rng = np.random.default_rng(12345)
    y_val = rng.uniform(0.3, 1.7, (T, n))
    y_sym = np.empty((T, n), dtype=object)
    t_sym = np.empty((T,), dtype=object)
    for i in range(T):
        t_sym[i] = sympy.Symbol('__obs_T%d' % i, real=True)
        for v in range(n):
            y_sym[i, v] = sympy.Symbol('__obs_Y%d_%d' % (i, v), real=True)
    sim, tt = _Traced(y_val, y_sym), _Traced(grid, t_sym)
    try:
        mapped, mapped_t = map_fn(sim, tt, experiment, measurement, parameters, True)
    except ObservableError:
        raise
    except Exception as e:      # noqa: BLE001 -- anything the traced arrays do not support
        raise ObservableError("'custom' mapping callback %s could not be traced (%s: %s): give the observable as an expression, "
                              "('custom', 'x4 / (x4 + x9)')" % (getattr(map_fn, '__name__', map_fn), type(e).__name__, e))
    _, _, tps = measurement.get_nonzero_measurements()
    gi = utils.sample_index(grid, tps)
    if not isinstance(mapped, _Traced):
        raise ObservableError("'custom' mapping callback %s returns values that do not depend on the simulation"
                              % getattr(map_fn, '__name__', map_fn))
    if mapped.shape != (len(gi),):
        raise ObservableError("'custom' mapping callback returns %s values for %d measurement times" % (mapped.shape, len(gi)))
    if not np.allclose(_val_of(mapped_t), grid[gi], rtol=0, atol=0):
        raise ObservableError("'custom' mapping callback does not sample the grid points the reference samples "
                              "(searchsorted(model_timepoints, measurement times), project/utils.py:19-21)")
    names = [sympy.Symbol(nm, real=True) for nm in var_names]
    t = sympy.Symbol('t', real=True)
    forms = []
    for r, i in enumerate(gi):
        ren = {y_sym[i, v]: names[v] for v in range(n)}
        ren[t_sym[i]] = t
        e = sympy.sympify(mapped.sym[r])
        foreign = [s for s in e.free_symbols if s not in ren]
        if foreign:
            raise ObservableError("'custom' mapping callback: row %d depends on other grid points than its own (%s); only "
                                  "pointwise observables can be compiled" % (r, foreign[:3]))
        forms.append(sympy.simplify(e.xreplace(ren)))
    for r in range(1, len(forms)):
        if sympy.simplify(forms[r] - forms[0]) != 0:
            raise ObservableError("'custom' mapping callback computes different functions for different rows (%s vs %s): "
                                  "give the observable as an expression" % (forms[0], forms[r]))
    expr = forms[0]
    used = [v for v in range(n) if names[v] in expr.free_symbols]
    if not used:
        raise ObservableError("'custom' mapping callback does not depend on any model variable")
    # ---- the Jacobian callback against the derived derivative ----
    if jac_fn is not None:
        k = int(n_exp_params)
        S = rng.standard_normal((T, n * k))
        kw = {}
        try:
            if 'model_sim' in inspect.signature(jac_fn).parameters:
                kw['model_sim'] = y_val
        except (TypeError, ValueError):
            pass
        grads = [sympy.diff(expr, names[v]) for v in used]
        constant = all(not g.free_symbols for g in grads)
        if not constant and not kw:
            raise TypeError("'custom' mapping: the observable %s is not linear in the model variables, and the reference hands "
                            "jacobian_map_fn only the model Jacobian (project/base_project.py:461-464) -- its derivative cannot "
                            "be checked against that callback.  Give the observable as an expression, ('custom', '%s'): "
                            "the derivative is then derived" % (expr, expr))
        got = np.asarray(jac_fn(S, grid, experiment, measurement, parameters, True, **kw), dtype=float)
        want = np.zeros((len(gi), k))
        for v, g in zip(used, grads):
            f = sympy.lambdify([names[u] for u in used] + [t], g, 'numpy')
            c = np.broadcast_to(np.asarray(f(*[y_val[gi, u] for u in used], grid[gi]), dtype=float), (len(gi),))
            want += c[:, None] * S[gi, v * k:(v + 1) * k]
        if got.shape != want.shape or not np.allclose(got, want, rtol=1e-9, atol=1e-12):
            raise ValueError("'custom' mapping: jacobian_map_fn disagrees with the derivative of map_fn (observable %s): largest "
                             "difference %.3g" % (expr, float(np.max(np.abs(got - want))) if got.shape == want.shape else np.inf))
    return expr
