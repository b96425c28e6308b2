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
            raise TypeError("'custom' mappings with Python callbacks (parameters, map_fn, jacobian_map_fn) cannot run in "
                            "the assembly kernel: give the observable as an expression of the model variables, "
                            "('custom', 'x4 / (x4 + x9)') or ('custom', ({'w': 0.3}, 'w * x4 + (1 - w) * x9'))")
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
