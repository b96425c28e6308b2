"""ORACLE (test infrastructure, not product code) -- integration half of the hot path.

CPU restatement of the reference's ``OdeModel.simulate`` (model/ode_model.py:128-169)
and ``OdeModel.calc_jacobian`` (model/ode_model.py:83-126).  The reference's
algorithm on this path lives in a third-party dependency that is not under
/root/reference: ``scipy.integrate.odeint`` -> ODEPACK LSODA (Fortran).  The
reference pins no SciPy version (no setup.py / requirements); this container
and the GPU box both carry SciPy 1.15.3.  The restatement therefore issues the
reference's exact call

    odeint(func_wrapper, init_conditions, t_sim, Dfun=None, col_deriv=True,
           rtol=1e-10, atol=1e-10)                      (ode_model.py:122-123,167-168)

(or Dfun = the analytic Jacobian callback, the reference's ``use_jac`` path, :114-120,154-160)

on a right-hand side with the reference callback contract f(y, t, yout, p).

PINNED: tests/test_oracle_golden.py checks this module against
  * the reference's own golden array (tests/test_OdeModel.py:25-26) and analytic
    sensitivities (:42-45),
  * tests/golden/*.npz, produced by importing the REAL reference ``OdeModel``
    (tests/golden/make_golden.py, run in the build container).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes

import numpy as np
from scipy.integrate import odeint

RTOL = 1e-10  # model/ode_model.py:123,168
ATOL = 1e-10


def _wrap(fn, n_out, params):
    """func_wrapper of the reference (ode_model.py:109-112,162-165): closure-local yout."""
    yout = np.zeros(n_out)
    p = np.ascontiguousarray(params, dtype=np.float64)

    def func_wrapper(y, t):
        fn(y, t, yout, p)
        return yout
    return func_wrapper


def _wrap_c(cfn, n_out, params):
    """Same wrapper around a compiled C RHS (the role numba plays in the reference,
    ode_model.py:50-51): identical arithmetic to the generated Python callable."""
    yout = np.zeros(n_out)
    p = np.ascontiguousarray(params, dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    yout_p = yout.ctypes.data_as(dp)
    p_p = p.ctypes.data_as(dp)

    def func_wrapper(y, t):
        yc = np.ascontiguousarray(y, dtype=np.float64)
        cfn(yc.ctypes.data_as(dp), float(t), yout_p, p_p)
        return yout
    func_wrapper._keep = (yout, p)
    return func_wrapper


def _wrap_jac(jac, n, params):
    """jac_wrapper of the reference (ode_model.py:114-120,154-160): one zero matrix allocated per call of
    simulate / calc_jacobian, filled in place (the generated code writes its non-zero entries only)."""
    jacout = np.zeros((n, n))
    p = np.ascontiguousarray(params, dtype=np.float64)

    def jac_wrapper(y, t):
        jac(y, t, jacout, p)
        return jacout
    return jac_wrapper


def simulate(model, experiment_params, t_sim, init_conditions=None, n_vars=None, full_output=False,
             use_c=False, model_jac=None):
    """ode_model.py:128-169.  ``model``: callable f(y,t,yout,p) or a GeneratedModel.  ``model_jac``: the
    reference's optional analytic Jacobian callback (``use_jac`` path, :154-160), handed to LSODA as Dfun."""
    gm = None if callable(model) else model
    if gm is not None:
        n_vars = gm.n_vars
    if init_conditions is None:
        init_conditions = np.zeros((n_vars,))                       # :151-152
    if gm is not None and use_c:
        fw = _wrap_c(gm.c_library().sbm_rhs, len(init_conditions), experiment_params)
    else:
        fw = _wrap(gm.model if gm is not None else model, len(init_conditions), experiment_params)
    dfun = _wrap_jac(model_jac, len(init_conditions), experiment_params) if model_jac is not None else None
    return odeint(fw, init_conditions, t_sim, Dfun=dfun, col_deriv=True, rtol=RTOL, atol=ATOL,
                  full_output=full_output)


def calc_jacobian(sens_model, experiment_params, t_sim, init_conditions=None, n_vars=None, n_sens=None,
                  full_output=False, use_c=False, return_states=False, sens_model_jac=None):
    """ode_model.py:83-126: integrate [y; S], return the S part ``sim[:, n_vars:]``.  ``sens_model_jac``:
    optional analytic Jacobian of the augmented system (``use_jac`` path, :114-120)."""
    gm = None if callable(sens_model) else sens_model
    if gm is not None:
        n_vars, n_sens = gm.n_vars, gm.n_sens
    if init_conditions is None:
        init_conditions = np.zeros((n_vars + n_vars * n_sens,))     # base_project.py:512
    if gm is not None and use_c:
        fw = _wrap_c(gm.c_library().sbm_sens_rhs, len(init_conditions), experiment_params)
    else:
        fw = _wrap(gm.sens_model if gm is not None else sens_model, len(init_conditions), experiment_params)
    dfun = _wrap_jac(sens_model_jac, len(init_conditions), experiment_params) if sens_model_jac is not None else None
    out = odeint(fw, init_conditions, t_sim, Dfun=dfun, col_deriv=True, rtol=RTOL, atol=ATOL,
                 full_output=full_output)
    sim, info = (out if full_output else (out, None))
    sens = sim[:, n_vars:]                                          # :125
    res = (sens, sim[:, :n_vars]) if return_states else sens
    return (res, info) if full_output else res


def tight_solution(gm, experiment_params, t_sim, sens=True, use_c=False, atol=1e-16):
    """Independent high-accuracy solution (DOP853, rtol 1e-13) used to tell
    'GPU more accurate than LSODA' from 'GPU wrong' (SURVEY.md section 8(d)).  Returns (len(t_sim), N)
    with N = n (+ n k with ``sens``); t_sim[0] is the initial time, y(t_sim[0]) = 0."""
    from scipy.integrate import solve_ivp
    n, k = gm.n_vars, gm.n_sens
    N = n + n * k if sens else n
    if use_c:
        lib = gm.c_library()
        fw = _wrap_c(lib.sbm_sens_rhs if sens else lib.sbm_rhs, N, experiment_params)
    else:
        fw = _wrap(gm.sens_model if sens else gm.model, N, experiment_params)

    def f(t, y):
        return fw(y, t).copy()
    t_sim = np.asarray(t_sim, dtype=float)
    sol = solve_ivp(f, (float(t_sim[0]), float(t_sim[-1])), np.zeros(N), method='DOP853', t_eval=t_sim,
                    rtol=1e-13, atol=atol)
    return sol.y.T


def tight_stiff_solution_by_columns(gm, experiment_params, t_sim, group=10, rtol=1e-12, atol=1e-15, mxstep=200000):
    """A TIGHT solution of a STIFF model's state + sensitivity system at a fraction of the cost of one LSODA run on all
    n (1 + k) equations (which differences and factors a dense n(1+k)-square Jacobian: 15 - 45 minutes per vector for
    stiff50 at rtol 1e-12).  The columns of S are coupled only through the state -- S_j' = J_y(y) S_j + J_p[:, j] -- so the
    augmented system is integrated in groups of ``group`` columns, each group together with its own copy of the state:
    the SAME odeint call (LSODA, rtol / atol as given, the generated C right-hand side with the other columns held at
    zero), systems of n (1 + group) equations.  Mathematically the same problem; numerically every group controls its
    own steps (as the GPU's column chunks do).  Validated against the full-system tight solutions of
    tests/golden/stiff50_tight.npz (tests/test_oracle_golden.py).  Returns (Y (len(t_sim), n), S (len(t_sim), n k));
    t_sim[0] is the initial time, y = 0 and S = 0 there."""
    n, k = gm.n_vars, gm.n_sens
    N = n + n * k
    cfn = gm.c_library().sbm_sens_rhs
    dp = ctypes.POINTER(ctypes.c_double)
    p = np.ascontiguousarray(experiment_params, dtype=np.float64)
    full = np.zeros(N)
    out = np.zeros(N)
    t_sim = np.asarray(t_sim, dtype=float)
    Y = None
    S = np.zeros((len(t_sim), n, k))
    for start in range(0, k, group):
        cols = np.arange(start, min(start + group, k))
        g = len(cols)
        # positions of the group's entries S[i, j] (layout n + i k + j) in the full augmented vector
        pos = (n + np.arange(n)[:, None] * k + cols[None, :]).ravel()
        red = np.zeros(n + n * g)

        def f(z, t):
            full[:] = 0.0
            full[:n] = z[:n]
            full[pos] = z[n:]
            cfn(full.ctypes.data_as(dp), float(t), out.ctypes.data_as(dp), p.ctypes.data_as(dp))
            red[:n] = out[:n]
            red[n:] = out[pos]
            return red
        sol = odeint(f, np.zeros(n + n * g), t_sim, rtol=rtol, atol=atol, mxstep=mxstep)
        if Y is None:
            Y = sol[:, :n].copy()
        S[:, :, cols] = sol[:, n:].reshape(len(t_sim), n, g)
    return Y, S.reshape(len(t_sim), n * k)
