"""ORACLE (test infrastructure, not product code): the forward-sensitivity equations in the EXPANDED form the
reference derives them in -- one equation per (state i, non-fixed parameter j),

    d/dt sens_i_j = d f_i / d p_j + sum_m d f_i / d y_m * sens_m_j        (symbolic/sympy_tools.py:130-146)

differentiating every equation with respect to every variable, with no use of the sparsity of the model.  The product
derives the sparse form S' = J_y S + J_p (sysbio_modeling_amd/symbolic/sympy_tools.py::derive_sparse_jacobians) and
builds its own expanded dictionary from that; tests/test_symbolic.py checks the two against each other.
Only tests/ may import this."""
from collections import OrderedDict

from sympy import Symbol, diff


def reference_style_sensitivity_equations(equations, params):
    sens_eqns = OrderedDict()
    for var_i, f_i in equations.items():
        for par_j in params.keys():
            if params[par_j] == 'fixed':                                  # :137-139
                continue
            dsens = diff(f_i, Symbol(par_j))
            for var_k in equations.keys():
                dsens += diff(f_i, Symbol(var_k)) * Symbol('sens_%s_%s' % (var_k, par_j))
            sens_eqns['d_sens_%s_%s' % (var_i, par_j)] = dsens
    return sens_eqns
