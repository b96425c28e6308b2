"""Model text -> SymPy -> derived equations.

Host-side step *before* the hot path: it defines the math and the state layout
the HIP kernels integrate.  API mirrors the reference module
(symbolic/sympy_tools.py: ``parse_model_file`` :272, ``process_model_dict`` :325,
``_derive_sensitivity_equations`` :130, ``_derive_jacobian_equations`` :149),
re-written for Python 3 and with one structural change: instead of expanding
and ``simplify``-ing every one of the n*k sensitivity equations
(reference :145), the model is kept in the sparse matrix form

    S' = J_y(y, p) . S + J_p(y, p)          J_y = df/dy (n x n),  J_p = df/dp (n x k)

so that the emitters (``emit.py``) can print J_y/J_p once (the part that is
uniform across sensitivity columns) and a per-column sparse update.  The
augmented-state layout is the reference's: index ``n + i*k + j`` holds
d y_i / d p_j, state-major / parameter-minor, parameters marked ``'fixed'``
skipped (reference :137-139, emit order :185-195).
"""
from __future__ import annotations

import inspect
import re
from collections import OrderedDict

from sympy import Symbol, sympify, diff, cse

_CATEGORIES = OrderedDict([
    ('Parameters', False), ('Variables', False),
    ('Conservation Laws', True), ('Rate Laws', True),
    ('Differential Equations', True), ('Imports', False),
])


def _sympify_chunk(chunk, sympify_rhs=False, local_symbols=None):
    """One ``lhs = rhs`` pair per line; comments and blanks skipped (reference :18-35)."""
    symbols_dict = OrderedDict()
    for line in chunk:
        line = line.replace(" ", "").replace("\t", "")
        if line.startswith("#") or line == "":
            continue
        if "=" not in line:
            continue
        lhs = line[:line.find("=")]
        rhs = line[line.find("=") + 1:]
        if not sympify_rhs:
            symbols_dict[lhs] = Symbol(lhs)
        else:
            symbols_dict[lhs] = sympify(rhs, locals=local_symbols)
    return symbols_dict


def _model_text(model):
    if callable(model) and hasattr(model, 'py_func'):  # numba dispatcher
        model = model.py_func
    if inspect.isfunction(model):
        return inspect.getsource(model)
    if isinstance(model, str):
        if "#*!" in model:
            return model
        with open(model, 'r') as fh:
            return fh.read()
    if hasattr(model, 'read'):
        text = model.read()
        try:
            model.close()
        except Exception:
            pass
        return text
    raise TypeError("model must be a function, a path, model text or a file-like object")


def parse_model_file(model):
    """Split model text on ``#*! <Section> Start`` / ``End`` markers.

    Accepts a python function (its source is read), a path, raw model text or
    a file-like object.  Returns a dict with 'Parameters', 'Variables',
    'Conservation Laws', 'Rate Laws', 'Differential Equations', 'Imports'
    (reference symbolic/sympy_tools.py:272-322).
    """
    model_text = _model_text(model)
    # every name on a lhs is a plain Symbol, also names such as 'S', 'N', 'E'
    # that sympify would otherwise resolve to sympy singletons
    names = re.findall(r'^\s*([A-Za-z_][A-Za-z_0-9]*)\s*=', model_text, flags=re.M)
    local_symbols = {nm: Symbol(nm) for nm in names}
    for nm in list(local_symbols):
        if nm.startswith('d_'):
            local_symbols.pop(nm)

    parsed_model = {}
    for category, sympify_rhs in _CATEGORIES.items():
        start_idx = model_text.find("#*! %s Start" % category)
        end_idx = model_text.find("#*! %s End" % category)
        if start_idx < 0 or end_idx < 0:
            chunk = []
        else:
            chunk = model_text[start_idx:end_idx].split('\n')[1:]
        if category == 'Imports':
            parsed_model[category] = [ln for ln in chunk if ln.strip()] or None
        else:
            parsed_model[category] = _sympify_chunk(chunk, sympify_rhs, local_symbols)
    return parsed_model


def _derive_sensitivity_equations(equations, params):
    """Expanded forward-sensitivity equations, one per (state i, non-fixed param j).

    d/dt sens_i_j = df_i/dp_j + sum_m df_i/dy_m * sens_m_j   (reference :130-146).
    Kept for API parity and for cross-checking the sparse form; the emitters
    use ``derive_sparse_jacobians`` instead.
    """
    sens_eqns = OrderedDict()
    for var_i, f_i in equations.items():
        for par_j in params.keys():
            if params[par_j] == 'fixed':
                continue
            dsens = diff(f_i, Symbol(par_j))
            for var_k in equations.keys():
                sens_kj = Symbol('sens_%s_%s' % (var_k, par_j))
                dsens += diff(f_i, Symbol(var_k)) * sens_kj
            sens_eqns['d_sens_%s_%s' % (var_i, par_j)] = dsens
    return sens_eqns


def _derive_jacobian_equations(equations):
    """d f_i / d y_j for every pair (reference :149-159)."""
    jacobian_equations = OrderedDict()
    for var_i, f_i in equations.items():
        for var_j in equations.keys():
            jacobian_equations[Symbol('d_%s_d_%s' % (var_i, var_j))] = diff(f_i, Symbol(var_j))
    return jacobian_equations


def derive_sparse_jacobians(equations, params):
    """Non-zero entries of J_y and J_p.

    Returns (jy, jp): lists of (row, col, expr); J_p columns are numbered over
    the NON-fixed parameters only, in ``params`` order (= the reference's
    sensitivity column order).
    """
    var_names = list(equations.keys())
    sens_params = [p for p in params if params[p] != 'fixed']
    jy, jp = [], []
    for i, (var_i, f_i) in enumerate(equations.items()):
        free = f_i.free_symbols
        for m, var_m in enumerate(var_names):
            s = Symbol(var_m)
            if s in free:
                d = diff(f_i, s)
                if d != 0:
                    jy.append((i, m, d))
        for j, par_j in enumerate(sens_params):
            s = Symbol(par_j)
            if s in free:
                d = diff(f_i, s)
                if d != 0:
                    jp.append((i, j, d))
    return jy, jp


def process_model_dict(model_dict, fixed_params=None, calculate_model_sensitivities=True,
                       simplify_subexpressions=False, calculate_model_jacobian=False):
    """Substitute rate/conservation laws, derive sensitivities (reference :325-397).

    Adds to ``model_dict``: 'Expanded Equations' (state name -> expr),
    'Sensitivity Equations' (expanded form or None), 'Sparse Jacobians'
    ((jy, jp) triplets), 'Model Jacobian Equations', 'Subexpressions'.
    """
    eqns = model_dict['Differential Equations']
    rate_laws = model_dict.get('Rate Laws') or {}
    cons_laws = model_dict.get('Conservation Laws') or {}
    params = model_dict['Parameters']

    if fixed_params is not None:
        for f_p in fixed_params:
            if f_p not in params:
                raise KeyError('%s not in model parameters' % f_p)
            params[f_p] = 'fixed'

    rl = {Symbol(k): v for k, v in rate_laws.items()}
    cl = {Symbol(k): v for k, v in cons_laws.items()}
    expanded_eqns = OrderedDict()
    for d_var, eqn in eqns.items():
        # rate laws may reference each other / conservation laws: substitute to a fixed point
        e = sympify(eqn)
        for _ in range(8):
            e_new = e.subs(rl).subs(cl)
            if e_new == e:
                break
            e = e_new
        expanded_eqns[d_var[2:]] = e

    variables = model_dict['Variables']
    # the reference fixtures write ``d_y = ...`` for a variable declared as ``_y``
    # (tests/test_utils/simple_model.py:16,21): accept both spellings
    expanded_eqns = OrderedDict(
        (('_' + k) if (k not in variables and ('_' + k) in variables) else k, v)
        for k, v in expanded_eqns.items())
    if list(expanded_eqns.keys()) != list(variables.keys()):
        # equations must be given in variable order: that order IS the state layout
        if set(expanded_eqns.keys()) != set(variables.keys()):
            raise ValueError("Differential equations %s do not match variables %s"
                             % (list(expanded_eqns), list(variables)))
        expanded_eqns = OrderedDict((v, expanded_eqns[v]) for v in variables)
    model_dict['Expanded Equations'] = expanded_eqns

    sens_eqns = None
    if calculate_model_sensitivities:
        sens_eqns = _derive_sensitivity_equations(expanded_eqns, params)
    model_dict['Sensitivity Equations'] = sens_eqns
    model_dict['Sparse Jacobians'] = derive_sparse_jacobians(expanded_eqns, params)

    model_jac_eqns = None
    if calculate_model_jacobian:
        all_eqns = OrderedDict(expanded_eqns)
        if sens_eqns is not None:
            all_eqns.update(sens_eqns)
        model_jac_eqns = _derive_jacobian_equations(all_eqns)
    model_dict['Model Jacobian Equations'] = model_jac_eqns

    subexpressions = None
    if simplify_subexpressions:
        all_vals = list(expanded_eqns.values()) + (list(sens_eqns.values()) if sens_eqns else [])
        repeated, _ = cse(all_vals, optimizations='basic')
        subexpressions = OrderedDict((str(k), v) for k, v in repeated)
    model_dict['Subexpressions'] = subexpressions
    return model_dict
