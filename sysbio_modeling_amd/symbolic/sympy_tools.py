"""Model text -> SymPy -> derived equations.

Host-side step *before* the hot path: it defines the math and the state layout
the HIP kernels integrate.  API mirrors the reference module
(symbolic/sympy_tools.py: ``parse_model_file`` :272, ``process_model_dict`` :325),
with one structural change: instead of expanding
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


_ASSIGNMENT = re.compile(r'^\s*([A-Za-z_][A-Za-z_0-9]*)\s*=(?!=)\s*(.*?)\s*$')


def _section_entries(lines, with_rhs, local_symbols):
    """The ``name = expression`` lines of one model-text section, in order: name -> Symbol(name), or -> the parsed
    right-hand side when ``with_rhs``.  Comment lines, blank lines and lines without an assignment are ignored."""
    entries = OrderedDict()
    for raw in lines:
        text = raw.split('#', 1)[0]
        m = _ASSIGNMENT.match(text)
        if m is None:
            continue
        name, rhs = m.group(1), m.group(2)
        entries[name] = sympify(rhs, locals=local_symbols) if with_rhs else Symbol(name)
    return entries


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
            parsed_model[category] = _section_entries(chunk, sympify_rhs, local_symbols)
    return parsed_model


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


def expanded_sensitivity_equations(equations, params, jy, jp):
    """'d_sens_<var>_<param>' -> expression, state-major / parameter-minor over the non-fixed parameters (the key
    order of the reference's dictionary, symbolic/sympy_tools.py:130-146), built from the sparse Jacobian triplets."""
    var_names = list(equations)
    sens_params = [q for q in params if params[q] != 'fixed']
    rows_y = [[] for _ in var_names]
    for i, m, d in jy:
        rows_y[i].append((m, d))
    entries_p = {(i, j): d for i, j, d in jp}
    out = OrderedDict()
    for i, var_i in enumerate(var_names):
        for j, par_j in enumerate(sens_params):
            total = entries_p.get((i, j), sympify(0))
            for m, d in rows_y[i]:
                total = total + d * Symbol('sens_%s_%s' % (var_names[m], par_j))
            out['d_sens_%s_%s' % (var_i, par_j)] = total
    return out


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

    jy, jp = derive_sparse_jacobians(expanded_eqns, params)
    model_dict['Sparse Jacobians'] = (jy, jp)
    # The expanded forms the reference's dictionary carries are assembled from the sparse triplets (the emitters
    # never use them): d/dt sens_i_j = J_p[i, j] + sum_m J_y[i, m] sens_m_j, and d f_a / d y_b of the augmented system.
    sens_eqns = expanded_sensitivity_equations(expanded_eqns, params, jy, jp) if calculate_model_sensitivities else None
    model_dict['Sensitivity Equations'] = sens_eqns
    model_jac_eqns = None
    if calculate_model_jacobian:
        system = OrderedDict(expanded_eqns)
        if sens_eqns is not None:
            system.update((k[2:], v) for k, v in sens_eqns.items())
        model_jac_eqns = OrderedDict()
        names = list(system)
        for a_name, f_a in system.items():
            present = f_a.free_symbols
            for b_name in names:
                sb = Symbol(b_name)
                model_jac_eqns[Symbol('d_%s_d_%s' % (a_name, b_name))] = diff(f_a, sb) if sb in present else sympify(0)
    model_dict['Model Jacobian Equations'] = model_jac_eqns

    subexpressions = None
    if simplify_subexpressions:
        all_vals = list(expanded_eqns.values()) + (list(sens_eqns.values()) if sens_eqns else [])
        repeated, _ = cse(all_vals, optimizations='basic')
        subexpressions = OrderedDict((str(k), v) for k, v in repeated)
    model_dict['Subexpressions'] = subexpressions
    return model_dict
