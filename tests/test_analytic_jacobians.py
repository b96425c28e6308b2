"""Analytic ODE Jacobians: the reference's ``model_jac`` / ``sens_model_jac`` callbacks.

The reference prints them from symbolic derivatives (symbolic/sympy_tools.py:149-159,219-269) and hands them
to LSODA as ``Dfun`` with ``col_deriv=True`` (model/ode_model.py:114-120,154-160).  Here they come from the
sparse J_y / J_p form (``emit.emit_python_jacobians``).  Pinned three ways: against central differences of the
generated right-hand sides, against tests/golden/jac_path_ref.npz -- the REAL reference ``OdeModel`` run on its
``use_jac`` path with these callbacks (make_golden_jac.py) -- and by the fact that LSODA's results change when
the Jacobian is transposed."""
import os

import numpy as np
import pytest

from oracle import odeint_oracle as oo
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel, zoo_model

HERE = os.path.dirname(os.path.abspath(__file__))


def _model(name):
    if name == 'stiff12':
        return GeneratedModel(models_zoo.stiff_spec(12, name='stiff12'))
    if name == 'motif_fixed':       # conservation law + rate laws substituted, one parameter without a sensitivity column
        from tests.test_gpu_user_models import MOTIF_TEXT
        from sysbio_modeling_amd.symbolic import make_ode_model
        return make_ode_model(MOTIF_TEXT, name='motif', fixed_params=['e_tot'])
    return zoo_model(name)


@pytest.mark.parametrize('name', ['simple', 'michaelis_menten', 'cascade20', 'stiff12', 'motif_fixed'])
def test_callbacks_equal_central_differences(name):
    gm = _model(name)
    n, k = gm.n_vars, gm.n_sens
    N = n + n * k
    rng = np.random.default_rng(3)
    y = rng.uniform(0.1, 1.5, N)
    p = rng.uniform(0.2, 2.0, len(gm.param_order))
    J = np.zeros((N, N))
    gm.sens_model_jac(y, 0.0, J, p)
    Jn = np.zeros((n, n))
    gm.model_jac(y[:n], 0.0, Jn, p)
    fd = np.zeros((N, N))
    for b in range(N):
        h = 1e-6 * max(abs(y[b]), 0.1)
        yp, ym = y.copy(), y.copy()
        yp[b] += h
        ym[b] -= h
        op, om = np.zeros(N), np.zeros(N)
        gm.sens_model(yp, 0.0, op, p)
        gm.sens_model(ym, 0.0, om, p)
        fd[b, :] = (op - om) / (2 * h)                 # row b: derivatives with respect to y_b (col_deriv layout)
    scale = np.abs(fd).max()
    assert np.max(np.abs(J - fd)) <= 1e-8 * scale
    assert np.max(np.abs(Jn - fd[:n, :n])) <= 1e-8 * scale
    # the state block of the augmented Jacobian IS the model Jacobian; the sensitivities do not feed the state
    assert np.allclose(J[:n, :n], Jn, rtol=1e-13, atol=0) and not J[n:, :n].any()
    # a second call overwrites the same entries only (the caller allocates zeros once, ode_model.py:117,157)
    J2 = J.copy()
    gm.sens_model_jac(y, 0.0, J2, p)
    assert np.array_equal(J, J2)


@pytest.mark.parametrize('name', ['michaelis_menten', 'cascade20', 'stiff12', 'stiff50'])
def test_oracle_use_jac_path_equals_the_reference(name):
    """Same SciPy, same callbacks, same call: the oracle's Dfun path reproduces the reference's numbers."""
    g = np.load(os.path.join(HERE, 'golden', 'jac_path_ref.npz'))
    gm = _model(name)
    P, grid, idx = g[name + '_P'], g[name + '_t'], g[name + '_idx']
    for v, p in enumerate(P):
        Y = oo.simulate(gm, p, grid, model_jac=gm.model_jac)[idx]
        assert np.allclose(Y, g[name + '_Y_jac'][v], rtol=1e-12, atol=1e-14)
        if g[name + '_S_jac'].shape[-1]:
            S = oo.calc_jacobian(gm, p, grid, sens_model_jac=gm.sens_model_jac)[idx]
            assert np.allclose(S, g[name + '_S_jac'][v], rtol=1e-12, atol=1e-14)
    # with and without Dfun LSODA agrees to its tolerance -- and on the stiff systems NOT bit for bit: it did
    # use the callback there (on the non-stiff ones it stays on the Adams branch and never calls Dfun)
    d = np.max(np.abs(g[name + '_Y_jac'] - g[name + '_Y_nojac']))
    assert d <= 1e-8
    assert (d > 0) == name.startswith('stiff')


def test_a_transposed_jacobian_changes_the_stiff_result():
    """The index convention matters: handing LSODA the transpose (col_deriv=False layout) makes its BDF Newton
    iteration work with a wrong matrix: more work along another step sequence, or LSODA giving up."""
    gm = _model('stiff12')
    g = np.load(os.path.join(HERE, 'golden', 'jac_path_ref.npz'))
    p, grid, idx = g['stiff12_P'][0], g['stiff12_t'], g['stiff12_idx']

    def transposed(y, t, jacout, pp):
        tmp = np.zeros_like(jacout)
        gm.model_jac(y, t, tmp, pp)
        jacout[:] = tmp.T
    (Y_ok, info_ok) = oo.simulate(gm, p, grid, model_jac=gm.model_jac, full_output=True)
    (Y_t, info_t) = oo.simulate(gm, p, grid, model_jac=transposed, full_output=True)
    assert np.allclose(Y_ok[idx], g['stiff12_Y_jac'][0], rtol=1e-12, atol=1e-14)
    assert info_ok['message'] == 'Integration successful.'
    assert not np.array_equal(Y_ok, Y_t)
    # ... or gives up ("Excess work done on this call (perhaps wrong Dfun type)")
    gave_up = info_t['message'] != 'Integration successful.'
    assert gave_up or info_t['nfe'][-1] > info_ok['nfe'][-1]
