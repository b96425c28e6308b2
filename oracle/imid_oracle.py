"""ORACLE (test infrastructure, not product code) -- implicit midpoint with forward sensitivities.

The reference has no implicit integrator of its own: stiff systems go to LSODA, which switches to
BDF by itself (model/ode_model.py:122-123,167-168, optional analytic Jacobian ``Dfun`` :114-120).
BASELINE configs[4] asks for an implicit-midpoint GPU integrator; this module restates THAT scheme
on the CPU in dense numpy, step for step as csrc/sbm_integrators.hpp::sbm_imid_kernel runs it, so
the kernel can be checked at the level of the algorithm (tests/test_gpu_implicit.py), while parity
with the reference's results is checked against ``odeint_oracle`` (LSODA) after extrapolation.

  y_{n+1} = y_n + h f(ybar), ybar = (y_n + y_{n+1})/2;  Newton from ybar = y_n + (previous increment)/2
  (rescaled when the step size changes between output intervals; y_n for the first step):
      (I - h/2 J_y(ybar)) delta = ybar - y_n - h/2 f(ybar),  ybar -= delta,
      until max |delta_i| / (atol + rtol |ybar_i|) <= 1 (at most 12 iterations)
  S_{n+1} = 2 Sbar - S_n,  (I - h/2 J_y) Sbar = S_n + h/2 J_p, with J_y, J_p of the last evaluated iterate
  each output interval is cut into step_mult * ceil(dt / h0) equal steps.
  graded=True (SBM_IMPLICIT_MIDPOINT_GRADED): the first BASE step of the trajectory (step_mult steps) is cut
  into 13 groups of step_mult midpoint substeps of sizes hs * 2^-12, 2^-12, 2^-11, ..., 1/2 to resolve an
  initial layer (grids nested across step_mult: what Richardson extrapolation needs).

J_y and J_p are read off the generated sensitivity RHS (S' = J_y S + J_p): no second code path.
Parity: "scheme-level" only -- pinned to LSODA through the convergence tests, not to reference vectors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

MAXIT = 12
GRADE = 12


def jacobians(gm, y, t, p):
    """(f, J_y (n x n), J_p (n x k)) from the generated sens RHS, layout yout[n + i*k + j]."""
    n, k = gm.n_vars, gm.n_sens
    out = np.zeros(n + n * k)
    aug = np.zeros(n + n * k)
    aug[:n] = y
    gm.sens_model(aug, t, out, p)
    f = out[:n].copy()
    Jp = out[n:].reshape(n, k).copy()
    Jy = np.zeros((n, n))
    # S = unit columns, k at a time
    for start in range(0, n, max(k, 1)):
        cols = range(start, min(start + k, n))
        S = np.zeros((n, k))
        for j, m in enumerate(cols):
            S[m, j] = 1.0
        aug[n:] = S.ravel()
        gm.sens_model(aug, t, out, p)
        full = out[n:].reshape(n, k) - Jp
        for j, m in enumerate(cols):
            Jy[:, m] = full[:, j]
    return f, Jy, Jp


def integrate(gm, p, t_out, h0, rtol=1e-10, atol=1e-12, t0=0.0, y0=None, s0=None, with_sens=True, step_mult=1,
              graded=False):
    """Returns (Y (len(t_out), n), S (len(t_out), n*k) or None, n_steps, n_newton)."""
    n, k = gm.n_vars, gm.n_sens
    p = np.asarray(p, dtype=float)
    y = np.zeros(n) if y0 is None else np.array(y0, dtype=float)
    S = np.zeros((n, k)) if s0 is None else np.array(s0, dtype=float).reshape(n, k)
    Y_out = np.zeros((len(t_out), n))
    S_out = np.zeros((len(t_out), n * k))
    t = float(t0)
    n_steps = n_newton = 0
    eye = np.eye(n)
    dy_prev, hs_prev = np.zeros(n), 0.0
    for io, target in enumerate(t_out):
        dt = target - t
        if dt > 0:
            nd = np.ceil(dt / h0 - 1e-9)
            ns = (1 if nd < 1 else int(nd)) * step_mult
            hs = dt / ns
            hh = 0.5 * hs
            t_start = t
            dy_prev = dy_prev * (hs / hs_prev if hs_prev > 0 else 0.0)
            hs_prev = hs
            s = -1
            while s + 1 < ns:
              s += 1
              # graded first base step: step_mult substeps of each size hs * 2^-12, 2^-12, 2^-11, ..., 1/2
              grade_now = graded and n_steps == 0
              if grade_now:
                  sizes = [hs * 2.0 ** -GRADE] + [hs * 2.0 ** -(GRADE - j + 1) for j in range(1, GRADE + 1)]
                  subs = [w for w in sizes for _ in range(step_mult)]
              else:
                  subs = [hs]
              t_sub = t_start + s * hs
              for sub, hsub in enumerate(subs):
                hh = 0.5 * hsub
                tm = t_sub + hh
                t_sub += hsub
                yb = y + 0.5 * dy_prev
                conv = False
                for _ in range(MAXIT):
                    n_newton += 1
                    f, Jy, Jp = jacobians(gm, yb, tm, p)
                    M = eye - hh * Jy
                    delta = np.linalg.solve(M, (yb - y) - hh * f)
                    yb = yb - delta
                    if np.max(np.abs(delta) / (atol + rtol * np.abs(yb))) <= 1.0:
                        conv = True
                        break
                if not conv:
                    raise RuntimeError("Newton did not converge")
                gj = sub // step_mult
                dy_prev = 2.0 * (yb - y) * (2.0 if (len(subs) > 1 and gj > 0 and sub % step_mult == step_mult - 1) else 1.0)
                y = 2.0 * yb - y
                if with_sens:
                    Sb = np.linalg.solve(M, S + hh * Jp)
                    S = 2.0 * Sb - S
              n_steps += step_mult if grade_now else 1
              if grade_now:
                  s += step_mult - 1
            t = target
        Y_out[io] = y
        S_out[io] = S.ravel()
    return Y_out, (S_out if with_sens else None), n_steps, n_newton
