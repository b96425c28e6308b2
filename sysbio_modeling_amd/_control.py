"""Host-side control loops around the device integrators: global error control for the implicit
midpoint rule, and the stiff fallback of ``method='auto'``.

The reference never chooses an integrator: ``scipy.integrate.odeint`` is LSODA, which switches between
Adams and BDF formulas by itself and controls its local error (model/ode_model.py:122-123,167-168), so a
stiff model "just works" there.  The device integrators are an explicit adaptive pair (DOPRI45) and a
fixed-step implicit one (implicit midpoint, csrc/sbm_integrators.hpp).  Two loops close the gap:

``controlled_doubling``   The implicit midpoint rule with one Richardson level is run with n, 2n, 4n, ...
    steps (every step halved exactly: ``step_mult``) until two successive extrapolants agree:
    |E(2n) - E(n)| estimates the GLOBAL error of E(n), and E(2n) -- returned -- is 4 to 16 times more
    accurate still.  Vectors leave the loop one by one as they
    converge; only the rest is integrated again.  Cost: 3n steps per level, a geometric series dominated
    by its last term.

``with_stiff_fallback``   DOPRI45 with a step budget first; the vectors that exhaust it (the step size of
    an explicit method on a stiff problem is bounded by stability, not accuracy) or fail otherwise are
    integrated again with ``controlled_doubling``.  The switch is per parameter vector, as LSODA's is per
    trajectory.

Both work on numpy arrays or torch tensors (whatever ``run`` returns).
"""
from __future__ import annotations

import numpy as np

SBM_OK = 0
SBM_TOL_NOT_REACHED = 5          # include/sbm.h: host-side status of the control loop

IMPLICIT_CONTROLLED = ('implicit_controlled', 'implicit_auto', 'stiff')
AUTO = ('auto', 'lsoda_like')


def _is_torch(x):
    return type(x).__module__.startswith('torch')


def _err_per_vector(cur, prev, rtol, atol):
    """max over the entries of a vector's array of |cur - prev| / (rtol * max(|cur|, floor) + atol), with
    floor = 1e-3 * the vector's largest entry (values passing through zero are judged against their
    array's scale, as the parity criterion of SURVEY.md section 8d does).  NaN / inf -> inf."""
    if _is_torch(cur):
        import torch
        V = cur.shape[0]
        c, p = cur.reshape(V, -1), prev.reshape(V, -1)
        if c.shape[1] == 0:
            return torch.zeros(V, dtype=torch.float64, device=cur.device)
        big = c.abs().amax(dim=1, keepdim=True)
        sc = rtol * torch.maximum(c.abs(), 1e-3 * big) + atol
        e = ((c - p).abs() / sc).amax(dim=1)
        return torch.where(torch.isfinite(e), e, torch.full_like(e, float('inf')))
    V = cur.shape[0]
    c, p = cur.reshape(V, -1), prev.reshape(V, -1)
    if c.shape[1] == 0:
        return np.zeros(V)
    with np.errstate(invalid='ignore', over='ignore'):
        big = np.max(np.abs(c), axis=1, keepdims=True)
        sc = rtol * np.maximum(np.abs(c), 1e-3 * big) + atol
        e = np.max(np.abs(c - p) / sc, axis=1)
    return np.where(np.isfinite(e), e, np.inf)


def _to_numpy(x):
    return x.cpu().numpy() if _is_torch(x) else np.asarray(x)


def _index(x, idx):
    if _is_torch(x):
        import torch
        return x[torch.as_tensor(idx, device=x.device, dtype=torch.long)]
    return x[idx]


def _assign(dst, idx, src, sel=None):
    """dst[idx[sel]] = src[sel]"""
    if sel is not None:
        idx = idx[sel]
        src = _index(src, np.flatnonzero(sel))
    if _is_torch(dst):
        import torch
        dst[torch.as_tensor(idx, device=dst.device, dtype=torch.long)] = src
    else:
        dst[idx] = src


def controlled_doubling(run, n_vectors, compare, rtol, atol, max_doublings=9, accept=4.0, trace=None):
    """Global error control by step doubling.

    run(idx, mult) -> (outputs, status, steps): ``outputs`` a dict name -> array with leading axis
        len(idx) (the Richardson-extrapolated results of the vectors ``idx`` with EVERY step of the coarsest
        run cut into ``mult`` = 1, 2, 4, ... equal parts -- the integrators' ``step_mult``; a step COUNT per
        trajectory would not do: fixed-step runs take at least one step per output interval, so doubling a
        count that is below the number of output times changes nothing), ``status`` / ``steps`` integer
        arrays of length len(idx).
    compare : names of the outputs the error estimate is taken over.
    trace : optional list; receives (level, vector indices, estimates in tolerance units) per level.
    accept : the estimate (in units of the tolerance) of the COARSER extrapolant below which the finer one
        is returned; 4 leaves the returned values within tolerance even where stiffness has reduced the
        extrapolant to second order.

    Returns (outputs for all vectors, status (V,), steps spent (V,), levels used (V,)).  A vector whose
    estimate never met the tolerance carries its finest result and status SBM_TOL_NOT_REACHED; one whose
    finest run failed carries that run's status.
    """
    idx = np.arange(n_vectors)
    result = None
    status = np.zeros(n_vectors, dtype=np.int32)
    spent = np.zeros(n_vectors, dtype=np.int64)
    levels = np.zeros(n_vectors, dtype=np.int32)
    if n_vectors == 0:
        out, _, _ = run(idx, 1)
        return out, status, spent, levels
    prev, st_prev, steps = run(idx, 1)
    spent[idx] += _to_numpy(steps).astype(np.int64)
    result = {k: (v.clone() if _is_torch(v) else np.array(v, copy=True)) for k, v in prev.items()}
    status[idx] = _to_numpy(st_prev)
    for lv in range(1, max_doublings + 1):
        cur, st_cur, steps = run(idx, 2 ** lv)
        spent[idx] += _to_numpy(steps).astype(np.int64)
        st_cur = _to_numpy(st_cur).astype(np.int32)
        err = np.zeros(len(idx))
        for k in compare:
            err = np.maximum(err, _to_numpy(_err_per_vector(cur[k], prev[k], rtol, atol)))
        ok = (st_cur == SBM_OK) & (_to_numpy(st_prev) == SBM_OK) & (err <= accept)
        if trace is not None:
            trace.append((lv, idx.copy(), err.copy()))
        # everything still in the loop takes the finer result; the converged ones leave
        for k in result:
            _assign(result[k], idx, cur[k])
        levels[idx] = lv
        status[idx] = np.where(ok, SBM_OK, np.where(st_cur != SBM_OK, st_cur, SBM_TOL_NOT_REACHED))
        if ok.all():
            break
        keep = np.flatnonzero(~ok)
        idx = idx[keep]
        prev = {k: _index(v, keep) for k, v in cur.items()}
        st_prev = st_cur[keep]
    return result, status, spent, levels


def with_stiff_fallback(run_explicit, run_controlled, n_vectors):
    """run_explicit() -> (outputs, status, steps) for all vectors; run_controlled(idx) -> (outputs, status,
    steps, levels) for the subset ``idx`` (numpy index array), or None when the model has no implicit
    integrator (more than 64 state variables).  Returns (outputs, status, steps, stiff) with
    ``stiff`` the boolean mask of the vectors that went through the implicit integrator."""
    out, st, steps = run_explicit()
    st = _to_numpy(st).astype(np.int32).copy()
    steps = _to_numpy(steps).astype(np.int64).copy()
    stiff = st != SBM_OK
    if run_controlled is None:           # no implicit integrator for this model: the failures stand
        return out, st, steps, np.zeros_like(stiff)
    if stiff.any():
        idx = np.flatnonzero(stiff)
        out2, st2, steps2, _ = run_controlled(idx)
        for k in out:
            if _is_torch(out[k]) or isinstance(out[k], np.ndarray):
                _assign(out[k], idx, out2[k])
        st[idx] = st2
        steps[idx] += steps2
    return out, st, steps, stiff
