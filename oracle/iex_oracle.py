"""ORACLE (test infrastructure, not product code) -- extrapolated implicit Euler with local step-size control.

The reference has no implicit integrator of its own: stiff systems go to LSODA, which switches to BDF by itself and
controls its local error (model/ode_model.py:122-123,167-168).  The GPU's stiff integrator since round 3 is
csrc/sbm_implicit_extrap.hpp::sbm_iex_kernel (SBM_IMPLICIT_EXTRAP); this module restates THAT scheme on the CPU in
dense numpy, decision for decision as the kernel takes them, so that the kernel can be checked at the level of the
algorithm (tests/test_gpu_implicit.py) while parity with the reference's results is checked against the real
reference's goldens (LSODA) and ``odeint_oracle``.

One macro step of size H from (t, y, S), order K (harmonic sequence 1 .. K):
    T_j = j implicit-Euler steps of size h = H / j:
        Newton on y1 = y + h f(y1):  (I - h J_y(yb)) delta = yb - y - h f(yb),  yb -= delta,
            converged when max |delta_i| / (nrtol |yb_i| + natol) = rr <= 1, or -- two updates known, rr < rr_prev / 4 --
            when the predicted next update rr^2 (rr / rr_prev^2) <= 0.1 (the matrices are then re-evaluated at yb),
            or when the iteration has stalled within 10^3 tolerances of it (rounding); at most 8 iterations;
            nrtol = max(1e-5 rtol, 4e-15), natol = max(1e-5 atol, 4e-16 max |y_n|);
            predictor: polynomial through the last 2 / 3 / 4 points of the sequence (first step: h x slope of the
            last accepted macro step)
        (I - h J_y) S1 = S + h J_p   with J_y, J_p of the last evaluation
    T_KK = sum wH_j T_j,  T_KK - T_K,K-1 = sum wE_j T_j  (weights of polynomial extrapolation to h = 0, accumulated on
    T_j - S_n);  err = max(RMS of the state error, max over columns of the column RMS), every entry against
    atol + rtol max(|T_KK|, 1e-6 x the largest entry of its column so far / of the state now);
    err <= 1: accept T_KK;  H <- H fac,  fac = 0.9 err^(-1/K) within [0.2, 4] ([0.1, 0.9] after a failure of the
    error test, 1/4 after a Newton failure; no growth right after a rejection); steps land on the output times.

J_y and J_p are read off the generated sensitivity right-hand side (S' = J_y S + J_p): no second code path.
Parity: "scheme-level" only -- pinned to the reference through the convergence of the scheme (test_oracle_implicit.py:
against the reference's closed-form fixture and LSODA), not to reference vectors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes

import numpy as np

KMAX = 10
MAXIT = 8
FLOOR = 1e-6


def weights(K):
    """(wH, wE) for the harmonic sequence 1 .. K: T_KK = sum wH_j T_j, T_KK - T_K,K-1 = sum wE_j T_j."""
    wh = np.zeros(K + 1)
    we = np.zeros(K + 1)
    for j in range(1, K + 1):
        h, low = 1.0, (1.0 if j >= 2 else 0.0)
        for i in range(1, K + 1):
            if i == j:
                continue
            h *= j / (j - i)
            if i >= 2 and j >= 2:
                low *= j / (j - i)
        wh[j], we[j] = h, h - low
    return wh, we


def default_order(rtol):
    return 4 if rtol >= 1e-4 else (6 if rtol >= 1e-6 else 8)


def jacobians_from(gm, use_c=True):
    """(y, t, p) -> (f, J_y, J_p) read off the generated sensitivity RHS (layout yout[n + i*k + j]); ``use_c``: the
    compiled C restatement of the generated right-hand side (gm.c_library()), else the generated Python callable."""
    n, k = gm.n_vars, gm.n_sens
    N = n + n * k
    out = np.zeros(N)
    aug = np.zeros(N)
    if use_c:
        cfn = gm.c_library().sbm_sens_rhs
        dp = ctypes.POINTER(ctypes.c_double)

        def call(t, p):
            cfn(aug.ctypes.data_as(dp), float(t), out.ctypes.data_as(dp), p.ctypes.data_as(dp))
    else:
        def call(t, p):
            gm.sens_model(aug, t, out, p)

    def fn(y, t, p):
        aug[:] = 0.0
        aug[:n] = y
        call(t, p)
        f = out[:n].copy()
        Jp = out[n:].reshape(n, k).copy()
        Jy = np.zeros((n, n))
        for start in range(0, n, max(k, 1)):
            cols = range(start, min(start + k, n))
            S = np.zeros((n, k))
            for j, m in enumerate(cols):
                S[m, j] = 1.0
            aug[n:] = S.ravel()
            call(t, p)
            full = out[n:].reshape(n, k) - Jp
            for j, m in enumerate(cols):
                Jy[:, m] = full[:, j]
        return f, Jy, Jp
    return fn


def integrate(gm, p, t_out, rtol=3e-9, atol=3e-12, order=0, t0=0.0, y0=None, s0=None, with_sens=True, h0=0.0,
              max_steps=200000, use_c=True, predictor=None, sums='differences'):
    """Returns (Y (len(t_out), n), S (len(t_out), n*k) or None, info) with info = dict(n_steps, n_reject, n_eval,
    n_euler, status); status as the kernel's: 0 ok, 1 max_steps, 3 step_underflow.

    ``sums``: how the extrapolations of the sensitivities are accumulated -- 'differences' (sum_j w_j (T_j - S_n), added to
    S_n: csrc/sbm_implicit_extrap.hpp::sbm_iex_kernel) or 'values' (sum_j w_j T_j, which IS T_KK since the weights add up to
    one: csrc/sbm_implicit_extrap_seq.hpp, round 4).  The same numbers up to the rounding of the sums.

    ``predictor`` (experiments only, tests/tools/dev_iex_predictor.py; the kernel has no such thing): callable
    (j, m, h, Hs, y_n, ya, default, previous_sequence_states) -> Newton's starting point of step m of sequence j, in place
    of ``default`` (the polynomial through the sequence's own last points)."""
    n, k = gm.n_vars, gm.n_sens
    p = np.ascontiguousarray(p, dtype=np.float64)
    fjac = jacobians_from(gm, use_c)
    K = int(order) if order else default_order(rtol)
    K = min(max(K, 2), KMAX)
    wh, we = weights(K)
    y = np.zeros(n) if y0 is None else np.array(y0, dtype=float)
    S = np.zeros((n, k)) if s0 is None else np.array(s0, dtype=float).reshape(n, k)
    nrtol = max(1e-5 * rtol, 4e-15)
    eye = np.eye(n)
    t_out = np.asarray(t_out, dtype=float)
    t = float(t0)
    span = t_out[-1] - t0 if len(t_out) else 0.0
    H = h0 if h0 > 0 else 1e-3 * (span if span > 0 else 1.0)
    ydot = np.zeros(n)
    colmax = np.zeros(k)
    info = dict(n_steps=0, n_reject=0, n_eval=0, n_euler=0, status=0)
    after_reject = False
    Y_out = np.full((len(t_out), n), np.nan)
    S_out = np.full((len(t_out), n * k), np.nan)

    def newton(tm, h, ya, yb, natol):
        """-> (converged, yb, M, Jp)"""
        r_prev = 0.0
        for it in range(MAXIT):
            f, Jy, Jp = fjac(yb, tm, p)
            info['n_eval'] += 1
            M = eye - h * Jy
            try:
                delta = np.linalg.solve(M, (yb - ya) - h * f)
            except np.linalg.LinAlgError:
                return False, yb, None, None
            yb = yb - delta
            with np.errstate(all='ignore'):
                rr = np.max(np.abs(delta) / (nrtol * np.abs(yb) + natol))
            if not np.isfinite(rr):
                return False, yb, None, None
            if rr <= 1.0:
                return True, yb, M, Jp
            predicted = it > 0 and rr < 0.25 * r_prev and rr * rr * (rr / (r_prev * r_prev)) <= 0.1
            stalled = it >= 2 and rr >= 0.5 * r_prev and rr <= 1.0e3
            if predicted or stalled:
                f, Jy, Jp = fjac(yb, tm, p)
                info['n_eval'] += 1
                return True, yb, eye - h * Jy, Jp
            r_prev = rr
        return False, yb, None, None

    for io, target in enumerate(t_out):
        while info['status'] == 0 and t < target:
            if info['n_steps'] + info['n_reject'] >= max_steps:
                info['status'] = 1
                break
            rem = target - t
            landing = H * 1.0001 >= rem
            Hs = rem if landing else H
            if not Hs > 1e-14 * max(abs(t), abs(target)):
                info['status'] = 3
                break
            natol = max(1e-5 * atol, 4.0e-16 * float(np.float32(np.max(np.abs(y)))))
            yh = np.zeros(n)
            ye = np.zeros(n)
            zh = np.zeros((n, k))
            ze = np.zeros((n, k))
            ok = True
            prev_pts = []
            for j in range(1, K + 1):
                h = Hs / j
                ya, yp, yp2, yp3 = y.copy(), y - h * ydot, np.zeros(n), np.zeros(n)
                Sj = S.copy()
                pts = []
                for m in range(j):
                    if m < 2:
                        yb = 2.0 * ya - yp
                    elif m == 2:
                        yb = 3.0 * (ya - yp) + yp2
                    else:
                        yb = 4.0 * (ya + yp2) - 6.0 * yp - yp3
                    if predictor is not None:
                        yb = predictor(j, m, h, Hs, y, ya, yb, prev_pts)
                    ok, yb, M, Jp = newton(t + (m + 1) * h, h, ya, yb, natol)
                    if not ok:
                        break
                    pts.append(yb)
                    yp3, yp2, yp, ya = yp2, yp, ya, yb
                    info['n_euler'] += 1
                    if with_sens:
                        Sj = np.linalg.solve(M, Sj + h * Jp)
                if not ok:
                    break
                prev_pts = pts
                yh += wh[j] * (ya - y)
                ye += we[j] * (ya - y)
                if with_sens:
                    if sums == 'values':
                        zh += wh[j] * Sj
                        ze += we[j] * Sj
                    else:
                        zh += wh[j] * (Sj - S)
                        ze += we[j] * (Sj - S)
            err = np.inf
            colmax_new = colmax
            if ok:
                with np.errstate(all='ignore'):
                    cs = 0.0
                    if with_sens and k:
                        Tk = zh if sums == 'values' else S + zh
                        colmax_new = np.maximum(colmax, np.max(np.abs(Tk), axis=0))
                        r = ze / (rtol * np.maximum(np.abs(Tk), FLOOR * colmax_new[None, :]) + atol)
                        cs = np.max(np.sum(r * r, axis=0))
                    yk = y + yh
                    ry = ye / (rtol * np.maximum(np.abs(yk), FLOOR * np.max(np.abs(yk))) + atol)
                    err = np.sqrt(max(cs, np.sum(ry * ry)) / n)
                if not np.isfinite(err):
                    err = np.inf
            if err <= 1.0:
                ydot = yh / Hs
                y = y + yh
                if with_sens:
                    S = zh.copy() if sums == 'values' else S + zh
                colmax = colmax_new
                t = target if landing else t + Hs
                info['n_steps'] += 1
                fac = 0.9 * err ** (-1.0 / K) if err > 1e-12 else 4.0
                fac = min(1.0 if after_reject else 4.0, max(0.2, fac))
                if (not landing) or fac < 1.0 or Hs * fac > H:
                    H = Hs * fac
                after_reject = False
            else:
                info['n_reject'] += 1
                fac = 0.25
                if ok and np.isfinite(err):
                    fac = min(0.9, max(0.1, 0.9 * err ** (-1.0 / K)))
                H = Hs * fac
                after_reject = True
        if info['status'] == 0:
            Y_out[io] = y
            S_out[io] = S.ravel()
    return Y_out, (S_out if with_sens else None), info
