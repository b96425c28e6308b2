"""ORACLE side (test infrastructure, not product code): the parity tolerances, stated once.

Integration half.  The reference answer is SciPy ``odeint`` at rtol = atol = 1e-10 (model/ode_model.py:123,168).
Two criteria:

  * against THAT answer:  |gpu - ref| <= 1e-8 |ref| + 5e-9.  LSODA at atol = 1e-10 is itself only good to a few
    1e-9 in ABSOLUTE terms (measured against DOP853 at rtol 1e-13: 1 - 7e-9 on the cascade models), so no correct
    integrator can agree with it more closely on small components; ``parity_err``.
  * against a TIGHT solution (DOP853, rtol 1e-13) -- SURVEY.md section 8(d)'s criterion, with no allowance for
    anybody's noise:  |gpu - ref| <= 1e-8 max(|ref|, 1e-6 column max-abs);  ``survey_err``.

Assembly half.  Residuals, scale factors and the Jacobian are smooth functions of the sampled trajectories s_r and
sensitivities; their tolerances are the FIRST-ORDER PROPAGATION of the trajectory tolerances through the
reference's formulas (squared_loss_function.py:27-106, linear_scale_factor.py:27-42), nothing else:

    B_g   = C / D,   C = sum(s d / sigma^2),  D = sum(s^2 / sigma^2)                 (group g's rows)
    r     = (B s - d) / sigma
    Jm_rj = sum_m S[r, m] p_m   over the model parameters m of the row's experiment that sit in slot j
            (p_m = exp(theta_j): the log-parameter chain rule, base_project.py:450-455,482-485)
    dB_j  = N_j / D - 2 C M_j / D^2,   N_j = sum(Jm_ij d_i / sigma_i^2),  M_j = sum(Jm_ij s_i / sigma_i^2)
    J_rj  = B Jm_rj + s_r dB_j         (reference_compat; divided by sigma_r otherwise)

``project_tolerances`` returns the bound on |delta r|, |delta J| ... for |delta s_r| <= tau_s[r] and
|delta Jm_rj| <= tau_Jm[r, j].  For the LSODA criterion tau_s = 1e-8 |s| + 5e-9 n_vars(row) and
tau_Jm = 1e-8 |Jm| + 5e-9 n_vars(row) (parameters in slot j) exp(theta_j); for the tight criterion the section-8(d)
expression on s and on Jm directly.

Only tests/, __graft_entry__.smoke() and bench.py's checking legs may import this.
"""
from __future__ import annotations

import numpy as np

PARITY_RTOL = 1e-8
PARITY_ATOL = 5e-9     # LSODA's own absolute noise at atol = 1e-10
SURVEY_RTOL = 1e-8
SURVEY_FLOOR = 1e-6    # x column max-abs
# Section 8(d) floors every column at 1e-6 of ITS OWN largest entry -- which says nothing about a column that is
# negligible as a whole: the feedback loop of the cascade models produces sensitivity columns whose largest entry is
# 1e-13 (20 states) to 1e-67 (70 states) of the block's largest, and on those two DOP853 runs at rtol 1e-11 and 1e-13
# already differ by 300 "units" (measured, tests/tools/parity_tight.py).  An entry twelve orders of magnitude below
# the largest entry of the same trajectory's block is numerically zero for everything built from the block in
# double precision (residual Jacobians, J^T J): such entries are judged against 1e-12 of the block's largest instead.
SURVEY_BLOCK_FLOOR = 1e-12


def parity_err(a, b):
    """max |a-b| / (PARITY_ATOL + PARITY_RTOL |b|): <= 1 means within the tolerance against the reference's LSODA."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b) / (PARITY_ATOL + PARITY_RTOL * np.abs(b)))) if a.size else 0.0


def survey_tol(ref, axis=0, block_floor=None):
    """1e-8 max(|ref|, 1e-6 colmax, block_floor * blockmax): colmax over ``axis`` (the time axis of a (T, columns)
    block), blockmax over the whole block.  See SURVEY_BLOCK_FLOOR for the third term."""
    ref = np.asarray(ref, dtype=float)
    if not ref.size:
        return np.zeros_like(ref)
    bf = SURVEY_BLOCK_FLOOR if block_floor is None else block_floor
    colmax = np.max(np.abs(ref), axis=axis, keepdims=True)
    return SURVEY_RTOL * np.maximum(np.maximum(np.abs(ref), SURVEY_FLOOR * colmax), bf * np.max(np.abs(ref)))


def survey_err(a, ref, axis=0, block_floor=None):
    """max |a - ref| / survey_tol(ref): SURVEY.md section 8(d)'s criterion, for TIGHT references.  An entry that is
    exactly zero in the reference (a structurally absent sensitivity) must be exactly zero."""
    a = np.asarray(a, dtype=float)
    ref = np.asarray(ref, dtype=float)
    if not a.size:
        return 0.0
    return tol_ratio(a, ref, survey_tol(ref, axis, block_floor))


def tol_ratio(a, ref, tol):
    """max |a - ref| / tol with 0 / 0 = 0: an entry that is structurally zero (tolerance 0) must be exactly zero."""
    d = np.abs(np.asarray(a, dtype=float) - np.asarray(ref, dtype=float))
    if not d.size:
        return 0.0
    with np.errstate(invalid='ignore', divide='ignore'):
        e = np.where(d == 0.0, 0.0, d / tol)
    return float(np.max(e))


# ---------------------------------------------------------------------------
# assembly half
# ---------------------------------------------------------------------------
def _row_counts(a):
    """per row: number of model variables summed into it ('direct' 1, 'sum' several)."""
    off = np.asarray(a['row_var_off'])
    return (off[1:] - off[:-1]).astype(float)


def _slot_counts(a, q):
    """[E][q]: how many sensitivity-carrying model parameters of experiment e sit in project slot j."""
    pmap = np.asarray(a['pmap'])
    sens_col = np.asarray(a['sens_col'])
    E = pmap.shape[0]
    cnt = np.zeros((E, q))
    for e in range(E):
        for m in range(pmap.shape[1]):
            if pmap[e, m] >= 0 and sens_col[m] >= 0:
                cnt[e, pmap[e, m]] += 1.0
    return cnt


def lsoda_taus(a, theta, sims, Jm=None):
    """Trajectory-level tolerances against the reference's LSODA, per row (and per Jacobian entry)."""
    nv = _row_counts(a)
    tau_s = PARITY_RTOL * np.abs(sims) + PARITY_ATOL * nv
    if Jm is None:
        return tau_s, None
    q = Jm.shape[1]
    cnt = _slot_counts(a, q)[np.asarray(a['row_exp'])]              # (R, q)
    tau_Jm = PARITY_RTOL * np.abs(Jm) + PARITY_ATOL * nv[:, None] * cnt * np.exp(np.asarray(theta))[None, :]
    return tau_s, tau_Jm


def tight_taus(a, sims, Jm=None):
    """Section 8(d) on the sampled values themselves; a 'column' = one measurement's rows (one experiment, one
    measured variable, its time points) -- for the Jacobian, per project parameter."""
    exp = np.asarray(a['row_exp'])
    off = np.asarray(a['row_var_off'])
    var = np.asarray(a['row_vars'])
    key = [(int(exp[r]), tuple(var[off[r]:off[r + 1]])) for r in range(len(exp))]
    tau_s = np.zeros(len(exp))
    tau_Jm = np.zeros_like(Jm) if Jm is not None else None
    for k in set(key):
        sel = np.array([kk == k for kk in key])
        tau_s[sel] = survey_tol(sims[sel])
        if Jm is not None:
            tau_Jm[sel] = survey_tol(Jm[sel], axis=0)
    return tau_s, tau_Jm


def project_tolerances(a, sims, B, tau_s, Jm=None, tau_Jm=None, reference_compat=None, n_prior_rows=None,
                       sf_prior=None):
    """First-order bounds for the square loss with linear scale factors.

    a : descriptor arrays of the project (row_sf, row_data, row_sigma, ...); sims (R,); B (G,) scale factors;
    tau_s (R,), tau_Jm (R, q): the trajectory-level tolerances.  n_prior_rows: parameter-prior rows after the R
    measurement rows (plain arithmetic on theta: rounding only); sf_prior: (group index, sigma) per SF-prior row;
    both, and reference_compat, default to what the descriptor arrays say.
    Returns a dict: 'sims', 'sf' (G,), 'residuals' (R [+ SF-prior rows]), and with Jm: 'model_jacobian',
    'sf_gradient' (G, q), 'jacobian' (R [+ SF-prior rows], q)."""
    if reference_compat is None:
        reference_compat = bool(a.get('reference_compat', 1))
    if n_prior_rows is None:
        n_prior_rows = len(a.get('prior_idx', ()))
    if sf_prior is None:
        sf_prior = list(zip([int(g) for g in a.get('sf_prior_group', ())], [float(x) for x in a.get('sf_prior_sigma', ())]))
    grp = np.asarray(a['row_sf'])
    d = np.asarray(a['row_data'], dtype=float)
    sg = np.asarray(a['row_sigma'], dtype=float)
    R = len(sims)
    G = len(B)
    if int(a.get('loss_type', 0)) == 1:
        return _log_loss_tolerances(grp, d, sg, sims, B, tau_s, Jm, tau_Jm, reference_compat, n_prior_rows, sf_prior)
    w = 1.0 / sg ** 2
    Brow = np.where(grp >= 0, np.asarray(B, dtype=float)[np.maximum(grp, 0)], 1.0) if G else np.ones(R)
    tau_B = np.zeros(G)
    s_dB = np.zeros(R)        # |s_r| * tau_B[group of r]
    out = {'sims': tau_s}
    q = Jm.shape[1] if Jm is not None else 0
    tau_dB = np.zeros((G, q))
    dB = np.zeros((G, q))
    for g in range(G):
        sel = grp == g
        s, dd, ww = sims[sel], d[sel], w[sel]
        C, D = np.sum(s * dd * ww), np.sum(s * s * ww)
        dBds = (dd - 2.0 * (C / D) * s) * ww / D
        tau_B[g] = np.sum(np.abs(dBds) * tau_s[sel])
        if Jm is not None:
            Jg, tJ = Jm[sel], tau_Jm[sel]
            N = Jg.T @ (dd * ww)
            M = Jg.T @ (s * ww)
            dB[g] = N / D - 2.0 * C * M / D ** 2
            a_i = dd * ww / D - 2.0 * C * s * ww / D ** 2                                  # d dB_j / d Jm_ij
            b_ij = (-2.0 * (s * ww)[:, None] * N[None, :] / D ** 2 - 2.0 * (dd * ww)[:, None] * M[None, :] / D ** 2
                    - 2.0 * C * Jg * ww[:, None] / D ** 2 + 8.0 * C * (s * ww)[:, None] * M[None, :] / D ** 3)
            tau_dB[g] = np.abs(a_i) @ tJ + (np.abs(b_ij) * tau_s[sel][:, None]).sum(axis=0)
    has = grp >= 0
    gi = np.maximum(grp, 0)
    if G:
        s_dB = np.where(has, np.abs(sims) * tau_B[gi], 0.0)
    out['sf'] = tau_B
    tol_r = (np.abs(Brow) * tau_s + s_dB) / sg
    extra_r = [1e-13] * int(n_prior_rows)            # (theta - mean) / sigma: exact up to rounding
    extra_J = [np.full(q, 1e-13)] * int(n_prior_rows)
    for (g, sigma_p) in (sf_prior or []):
        extra_r.append(tau_B[g] / (abs(B[g]) * sigma_p))
        if Jm is not None:
            t = tau_dB[g] / abs(B[g]) + np.abs(dB[g]) * tau_B[g] / B[g] ** 2
            extra_J.append(t if reference_compat else t / sigma_p)     # (dB / B)[/ sigma]: linear_scale_factor.py:44-53
    out['residuals'] = np.concatenate([tol_r, np.asarray(extra_r)]) if extra_r else tol_r
    if Jm is not None:
        out['model_jacobian'] = tau_Jm
        out['sf_gradient'] = tau_dB
        tol_J = np.abs(Brow)[:, None] * tau_Jm
        if G:
            tol_J = tol_J + np.where(has[:, None], tau_B[gi][:, None] * np.abs(Jm) + tau_s[:, None] * np.abs(dB[gi])
                                     + np.abs(sims)[:, None] * tau_dB[gi], 0.0)
        if not reference_compat:
            tol_J = tol_J / sg[:, None]
        out['jacobian'] = np.vstack([tol_J, np.asarray(extra_J)]) if extra_J else tol_J
    return out


def _log_loss_tolerances(grp, d, sg, sims, B, tau_s, Jm, tau_Jm, reference_compat, n_prior_rows, sf_prior):
    """The same propagation for the log-square loss (log_squared_loss_function.py:23-98, log_scale_factor.py:17-36):
        log B_g = (sum(log d / e^2) - sum(log s / e^2)) / W,  e = sigma / d,  W = sum(1 / e^2)
        r = (log(B s) - log d) / sigma,   J_rj = Jm_rj / s_r + (dB_j / B),   dB_j / B = -(1/W) sum(Jm_ij / (s_i e_i^2))
    """
    R, G = len(sims), len(B)
    q = Jm.shape[1] if Jm is not None else 0
    rel = tau_s / np.abs(sims)                                     # |delta log s|
    tau_logB = np.zeros(G)
    tau_dlogB = np.zeros((G, q))
    dlogB = np.zeros((G, q))
    for g in range(G):
        sel = grp == g
        e2 = (sg[sel] / d[sel]) ** 2
        W = np.sum(1.0 / e2)
        tau_logB[g] = np.sum(rel[sel] / e2) / W
        if Jm is not None:
            s = sims[sel]
            dlogB[g] = -(Jm[sel] / (s * e2)[:, None]).sum(axis=0) / W
            tau_dlogB[g] = ((tau_Jm[sel] / np.abs(s)[:, None] + np.abs(Jm[sel]) * (tau_s[sel] / s ** 2)[:, None])
                            / e2[:, None]).sum(axis=0) / W
    has = grp >= 0
    gi = np.maximum(grp, 0)
    tol_r = (rel + (np.where(has, tau_logB[gi], 0.0) if G else 0.0)) / sg
    out = {'sims': tau_s, 'sf': np.abs(B) * tau_logB}
    extra_r = [1e-13] * int(n_prior_rows)
    extra_J = [np.full(q, 1e-13)] * int(n_prior_rows)
    for (g, sigma_p) in (sf_prior or []):
        extra_r.append(tau_logB[g] / sigma_p)
        if Jm is not None:
            extra_J.append(tau_dlogB[g] if reference_compat else tau_dlogB[g] / sigma_p)
    out['residuals'] = np.concatenate([tol_r, np.asarray(extra_r)]) if extra_r else tol_r
    if Jm is not None:
        out['model_jacobian'] = tau_Jm
        out['sf_gradient'] = np.abs(B)[:, None] * tau_dlogB + np.abs(dlogB) * (np.abs(B) * tau_logB)[:, None]
        tol_J = tau_Jm / np.abs(sims)[:, None] + np.abs(Jm) * (tau_s / sims ** 2)[:, None]
        if G:
            tol_J = tol_J + np.where(has[:, None], tau_dlogB[gi], 0.0)
        if not reference_compat:
            tol_J = tol_J / sg[:, None]
        out['jacobian'] = np.vstack([tol_J, np.asarray(extra_J)]) if extra_J else tol_J
    return out
