"""Host-side control loops around the device integrators: global error control for the implicit
midpoint rule, and the stiff fallback of ``method='auto'``.

The reference never chooses an integrator: ``scipy.integrate.odeint`` is LSODA, which switches between
Adams and BDF formulas by itself and controls its local error (model/ode_model.py:122-123,167-168), so a
stiff model "just works" there.  The device integrators are an explicit adaptive pair (DOPRI45) and a
fixed-step implicit one (implicit midpoint, csrc/sbm_integrators.hpp).  Two loops close the gap:

``controlled_romberg``   The implicit midpoint rule is run with n, 2n, 4n, ... steps (every step halved
    exactly: ``step_mult``) and the runs are combined in a Romberg table (the symmetric rule's global error
    expands in h^2) until the once-extrapolated results of two successive rows agree: a third of their
    difference bounds the GLOBAL error of the finer one even where stiffness has reduced the order to two.
    Vectors leave the loop one by one as they converge; only the rest is integrated again.  Every run is
    used once: the cost is a geometric series dominated by its last term.

``with_stiff_fallback``   DOPRI45 with a step budget first; the vectors that exhaust it (the step size of
    an explicit method on a stiff problem is bounded by stability, not accuracy) or fail otherwise are
    integrated again with ``controlled_romberg``.  The switch is per parameter vector, as LSODA's is per
    trajectory.

Both work on numpy arrays or torch tensors (whatever ``run`` returns).
"""
from __future__ import annotations

import numpy as np

SBM_OK = 0
SBM_TOL_NOT_REACHED = 5          # include/sbm.h: host-side status of the control loop

IMPLICIT_CONTROLLED = ('implicit_romberg',)      # the host loop below; 'implicit_controlled' is the in-kernel control now
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


def _where_rows(mask, a, b):
    """rows of a where mask else rows of b (mask: numpy bool over the leading axis)."""
    if _is_torch(a):
        import torch
        m = torch.as_tensor(mask, device=a.device).reshape((-1,) + (1,) * (a.dim() - 1))
        return torch.where(m, a, b)
    return np.where(mask.reshape((-1,) + (1,) * (a.ndim - 1)), a, b)


def controlled_romberg(run, n_vectors, compare, rtol, atol, max_doublings=9, accept=1.0, trace=None):
    """Global error control with a Romberg table: every run is used once.

    run(idx, mult) -> (outputs, status, steps): the RAW (second-order, unextrapolated) results of the vectors
        ``idx`` with every base step cut into ``mult`` parts (a step COUNT per trajectory would not do in place
        of ``mult``: fixed-step runs take at least one step per output interval, so doubling a count below the
        number of output times changes nothing).
    Row k of the table: T[k][0] = run(mult = 2^k), T[k][1] = (4 T[k][0] - T[k-1][0]) / 3 (fourth order: the
    symmetric rule's error expands in h^2), T[k][2] = (16 T[k][1] - T[k-1][1]) / 15 (sixth order where the
    expansion holds).

    Estimate and acceptance.  |T[k][1] - T[k-1][1]| / 3 bounds the error of T[k][1] as long as the error
    falls by at least 4 per halving -- true in the asymptotic regime (16) and on stiff systems, where order
    reduction leaves second order (measured on stiff50: 95 and 260 per halving on the way in).  The sharper
    textbook estimates |T[k][j] - T[k][j-1]| of the higher columns are NOT used: those entries differ by
    (row difference) / (4^j - 1) by construction, so they agree with each other whether or not they are right,
    and on stiff50 the columns beyond the first are 10 - 20 times WORSE than the first until the very end
    (scripts/dev_romberg_table.py).  Returned: T[k][2], which at acceptance differs from T[k][1] by at most a
    fifth of the tolerance and is orders of magnitude better on smooth problems.  A failed run (NaN rows) only
    invalidates the entries built on it: the table recovers two levels later.

    Fast path for smooth solutions.  The higher columns may be trusted once the table itself shows the
    asymptotic regime: the row-to-row differences of the fourth-order column fell by 16 (within [12, 20]) at
    this level AND the one before, and those of the sixth-order column by 64 (within [40, 100]).  Then
    |T[k][3] - T[k][2]| -- the estimate of the sixth-order entry -- decides, against a quarter of the tolerance,
    and the eighth-order entry T[k][3] is returned: two to three levels (4 - 8 times fewer steps) earlier on
    cascade20.  stiff50 never passes the ratio test (its ratios: 177, 99, 5.3, 11.7, 94, 258).

    trace : optional list; receives (level, vector indices, estimates in tolerance units) per level.

    Returns (outputs for all vectors, status (V,), steps spent (V,), levels used (V,)).  A vector whose estimate
    never met the tolerance carries its finest result and status SBM_TOL_NOT_REACHED; one whose finest run
    failed carries that run's status and raw output.
    """
    idx = np.arange(n_vectors)
    status = np.zeros(n_vectors, dtype=np.int32)
    spent = np.zeros(n_vectors, dtype=np.int64)
    levels = np.zeros(n_vectors, dtype=np.int32)
    first, st0, steps = run(idx, 1)
    if n_vectors == 0:
        return first, status, spent, levels
    spent[idx] += _to_numpy(steps).astype(np.int64)
    status[idx] = _to_numpy(st0)
    result = {k: (v.clone() if _is_torch(v) else np.array(v, copy=True)) for k, v in first.items()}
    keys = list(first.keys())
    prev_row = {k: [first[k]] for k in keys}           # T[k-1][0..]
    d1_prev = np.full(n_vectors, np.nan)               # row differences of columns 1 and 2 at the previous level
    d2_prev = np.full(n_vectors, np.nan)
    r1_prev = np.zeros(n_vectors, dtype=bool)          # column-1 ratio test passed at the previous level
    for lv in range(1, max_doublings + 1):
        cur, st_cur, steps = run(idx, 2 ** lv)
        spent[idx] += _to_numpy(steps).astype(np.int64)
        st_cur = _to_numpy(st_cur).astype(np.int32)
        row = {k: [cur[k], (4.0 * cur[k] - prev_row[k][0]) / 3.0] for k in keys}
        est = np.full(len(idx), np.inf)
        d1 = np.full(len(idx), np.nan)
        d2 = np.full(len(idx), np.nan)
        smooth = np.zeros(len(idx), dtype=bool)
        if lv >= 2:
            d1 = np.zeros(len(idx))
            for k in compare:
                d1 = np.maximum(d1, _to_numpy(_err_per_vector(row[k][1], prev_row[k][1], rtol, atol)))
            est = d1 / 3.0
            for k in keys:
                row[k].append((16.0 * row[k][1] - prev_row[k][1]) / 15.0)
        if lv >= 3:
            d2 = np.zeros(len(idx))
            for k in compare:
                d2 = np.maximum(d2, _to_numpy(_err_per_vector(row[k][2], prev_row[k][2], rtol, atol)))
            for k in keys:
                row[k].append((64.0 * row[k][2] - prev_row[k][2]) / 63.0)
        with np.errstate(invalid='ignore', divide='ignore'):
            r1 = (d1_prev / d1 >= 12.0) & (d1_prev / d1 <= 20.0)
            r2 = (d2_prev / d2 >= 40.0) & (d2_prev / d2 <= 100.0)
        if lv >= 4:
            smooth = r1 & r1_prev & r2 & (d2 / 63.0 <= 0.25 * accept)
            est = np.where(smooth, d2 / 63.0 / 0.25, est)      # in units of ``accept``, like the other branch
        # conservative branch returns the sixth-order entry, the fast path the eighth-order one
        pick = {k: (_where_rows(smooth, row[k][3], row[k][2]) if lv >= 4 and smooth.any() else
                    row[k][min(lv, 2)]) for k in keys}
        # a table entry built on a failed run is NaN: fall back to the best finite entry of the row
        for k in keys:
            for j in range(len(row[k]) - 2, -1, -1):
                bad = ~_to_numpy(_finite_rows(pick[k]))
                if not bad.any():
                    break
                pick[k] = _where_rows(bad, row[k][j], pick[k])
        ok = (st_cur == SBM_OK) & (est <= accept)
        if trace is not None:
            trace.append((lv, idx.copy(), est.copy()))
        failed = st_cur != SBM_OK
        for k in keys:
            if failed.any():         # a failed run comes back as the kernel left it (NaN / inf rows), not as NaN algebra
                pick[k] = _where_rows(failed, cur[k], pick[k])
            _assign(result[k], idx, pick[k])
        levels[idx] = lv
        status[idx] = np.where(ok, SBM_OK, np.where(failed, st_cur, SBM_TOL_NOT_REACHED))
        if ok.all():
            break
        keep = np.flatnonzero(~ok)
        idx = idx[keep]
        prev_row = {k: [_index(t, keep) for t in row[k]] for k in keys}
        d1_prev, d2_prev, r1_prev = d1[keep], d2[keep], r1[keep]
    return result, status, spent, levels


def _finite_rows(x):
    """per leading index: are all entries finite?"""
    if _is_torch(x):
        import torch
        return torch.isfinite(x.reshape(x.shape[0], -1)).all(dim=1)
    return np.isfinite(x.reshape(x.shape[0], -1)).all(axis=1)


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
        st[idx] = _to_numpy(st2).astype(np.int32)
        steps[idx] += _to_numpy(steps2).astype(np.int64)
    return out, st, steps, stiff
