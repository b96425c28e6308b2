"""The implicit-midpoint restatement (oracle/imid_oracle.py) is pinned to the reference integrator
(SciPy odeint / LSODA through oracle/odeint_oracle.py): second-order convergence towards it, and
agreement after Richardson extrapolation.  CPU only."""
import math

import numpy as np
import pytest

from oracle import imid_oracle, odeint_oracle as oo
from tests import reference_cases as rc


@pytest.fixture(scope='module')
def mm(zoo):
    return zoo('michaelis_menten')


def test_imid_converges_to_lsoda_with_order_two(mm):
    t = np.linspace(0, 100, 1000)
    idx = [250, 999]
    p = rc.MM_PARAMS * np.array([40.0, 30.0, 5.0, 3.0, 20.0])      # faster kinetics: visible curvature
    Yr = oo.simulate(mm, p, t)[idx]
    Sr = oo.calc_jacobian(mm, p, t)[idx]
    errs = []
    runs = []
    for mult in (1, 2, 4):
        Y, S, ns, nn = imid_oracle.integrate(mm, p, t[idx], h0=0.5, step_mult=mult)
        assert ns == 201 * mult             # two output intervals, 51 + 150 steps at mult 1
        runs.append((Y, S))
        errs.append((np.max(np.abs(Y - Yr)), np.max(np.abs(S - Sr) / (np.abs(Sr) + 1e-3))))
    for a, b in zip(errs[:-1], errs[1:]):
        assert 3.3 < a[0] / b[0] < 4.7 and 3.3 < a[1] / b[1] < 4.7       # h -> h/2: error / 4
    # (4 y_2n - y_n) / 3 cancels the h^2 term
    Yx = (4 * runs[2][0] - runs[1][0]) / 3
    Sx = (4 * runs[2][1] - runs[1][1]) / 3
    assert np.max(np.abs(Yx - Yr)) < errs[2][0] / 8
    assert np.max(np.abs(Sx - Sr) / (np.abs(Sr) + 1e-3)) < errs[2][1] / 8


def test_jacobians_read_off_the_sens_rhs(zoo):
    gm = zoo('cascade20')
    rng = np.random.default_rng(0)
    y = rng.uniform(0.1, 1.0, 20)
    p = rng.uniform(0.1, 1.0, 40)
    f, Jy, Jp = imid_oracle.jacobians(gm, y, 0.0, p)
    out = np.zeros(20)
    gm.model(y, 0.0, out, p)
    assert np.allclose(f, out)
    for m in (0, 7, 19):
        e = np.zeros(20); e[m] = 1e-6
        fp, fm = np.zeros(20), np.zeros(20)
        gm.model(y + e, 0.0, fp, p); gm.model(y - e, 0.0, fm, p)
        assert np.allclose(Jy[:, m], (fp - fm) / 2e-6, atol=1e-7)
    for j in (0, 25):
        e = np.zeros(40); e[j] = 1e-6
        fp, fm = np.zeros(20), np.zeros(20)
        gm.model(y, 0.0, fp, p + e); gm.model(y, 0.0, fm, p - e)
        assert np.allclose(Jp[:, j], (fp - fm) / 2e-6, atol=1e-7)


def test_oracle_equals_real_reference_on_stiff50(zoo, golden):
    """stiff50_ref.npz was produced by the REAL reference OdeModel (make_golden_stiff.py); the oracle
    issues the same odeint call.  State trajectories only here (the 2550-equation sensitivity run takes
    40 s per vector: it is what the golden file is for)."""
    from sysbio_modeling_amd import models_zoo
    gm = zoo('stiff50')
    g = golden('stiff50_ref.npz')
    assert np.array_equal(g['P'], models_zoo.stiff_ensemble(4096)[1][:3])
    for v in range(3):
        Y = oo.simulate(gm, g['P'][v], g['t'], use_c=True)[g['idx']]
        assert np.allclose(Y, g['Y'][v], rtol=1e-12, atol=1e-14)
    assert g['S'].shape == (3, 16, 2500) and np.all(np.isfinite(g['S']))
    # feed-forward cascade: species i does not depend on the rates of species downstream of it
    S4 = g['S'].reshape(3, 16, 50, 50)
    iu = np.triu_indices(50, k=1)
    assert np.abs(S4[:, :, iu[0], iu[1]]).max() < 1e-12


# ----------------------------------------------------------------------------------------------------------------
# extrapolated implicit Euler with local step control (oracle/iex_oracle.py: the scheme of csrc/sbm_implicit_extrap.hpp)
# ----------------------------------------------------------------------------------------------------------------
def test_iex_weights_are_polynomial_extrapolation_to_step_zero():
    """T_KK = sum wH_j T_j extrapolates polynomials of degree < K in h = 1/j to h = 0 exactly; T_K,K-1 those of
    degree < K - 1 through T_2 .. T_K."""
    from oracle import iex_oracle
    for K in (2, 4, 6, 8, 10):
        wh, we = iex_oracle.weights(K)
        j = np.arange(1, K + 1, dtype=float)
        assert abs(wh[1:].sum() - 1.0) < 1e-9 and abs(we[1:].sum()) < 1e-9
        for deg in range(1, K):
            assert abs(np.dot(wh[1:], (1.0 / j) ** deg)) < 1e-7 * np.abs(wh).max()
        wl = wh - we
        assert wl[1] == 0.0
        for deg in range(1, K - 1):
            assert abs(np.dot(wl[1:], (1.0 / j) ** deg)) < 1e-7 * np.abs(wl).max()
        # the first neglected power is seen by the lower-order combination only: that difference is the estimate
        assert abs(abs(np.dot(we[1:], (1.0 / j) ** (K - 1))) * math.factorial(K) - 1.0) < 1e-6     # = +-1 / K!


def test_iex_reproduces_the_reference_fixture_and_lsoda(zoo, mm):
    """The scheme against (a) the closed form of the reference's one-state fixture (tests/test_OdeModel.py:31-52:
    y = k_synt / k_deg (1 - exp(-k_deg t)) and its parameter derivatives) and (b) the reference's odeint call on the
    Michaelis-Menten fixture; tolerance proportionality: ten times tighter, about ten times closer."""
    from oracle import iex_oracle
    gm = zoo('simple')
    k_deg, k_synt = rc.SIMPLE_PARAMS if hasattr(rc, 'SIMPLE_PARAMS') else (0.001, 0.01)
    t = np.linspace(10.0, 100.0, 10)
    Y, S, info = iex_oracle.integrate(gm, np.array([k_deg, k_synt]), t, rtol=1e-9, atol=1e-12, use_c=False)
    e = np.exp(-k_deg * t)
    assert info['status'] == 0 and np.allclose(Y[:, 0], k_synt / k_deg * (1 - e), rtol=2e-9)
    dk_deg = k_synt * (t * e / k_deg - (1 - e) / k_deg ** 2)
    assert np.allclose(S[:, 0], dk_deg, rtol=1e-8) and np.allclose(S[:, 1], (1 - e) / k_deg, rtol=2e-9)
    tt = np.linspace(0, 100, 1000)
    idx = [250, 999]
    p = rc.MM_PARAMS * np.array([40.0, 30.0, 5.0, 3.0, 20.0])
    Yr = oo.simulate(mm, p, tt)[idx]
    Sr = oo.calc_jacobian(mm, p, tt)[idx]
    errs = []
    for rtol in (1e-6, 1e-7, 1e-8):
        Y, S, info = iex_oracle.integrate(mm, p, tt[idx], rtol=rtol, atol=1e-3 * rtol, order=6, use_c=False)
        assert info['status'] == 0 and info['n_euler'] >= info['n_steps'] * 21
        errs.append(max(np.max(np.abs(Y - Yr) / (np.abs(Yr) + 1e-3)), np.max(np.abs(S - Sr) / (np.abs(Sr) + 1e-3))))
    assert errs[0] < 1e-5 and errs[1] < 0.4 * errs[0] and errs[2] < 0.4 * errs[1], errs
    assert errs[2] < 2e-8


def test_iex_on_a_stiff_golden_vector_state_only(zoo, golden):
    """stiff50 (rates spanning 10^6), one vector of the real reference's golden, state only (the dense numpy solves
    of the full sensitivity system take minutes here; the GPU tests compare kernel and oracle on it)."""
    from oracle import iex_oracle
    from oracle.tolerances import parity_err
    gm = zoo('stiff50')
    g, gt = golden('stiff50_ref.npz'), golden('stiff50_tight.npz')
    t_out = g['t'][g['idx']][:4]
    Y, _, info = iex_oracle.integrate(gm, g['P'][0], t_out, rtol=3e-9, atol=3e-12, with_sens=False)
    assert info['status'] == 0 and info['n_steps'] < 60
    assert parity_err(Y, g['Y'][0][:4]) <= 1.0 and parity_err(Y, gt['Y'][0][:4]) <= 0.5
    # about three evaluations of f / J_y / J_p per Euler step (quadratically convergent Newton from a cubic predictor)
    assert 2.0 < info['n_eval'] / info['n_euler'] < 3.6


def test_wide_stiff_golden_is_the_oracles_call_and_column_groups_give_the_tight_solution(zoo, golden):
    """stiff50_wide_ref.npz: 32 more vectors of the ensemble through the REAL reference OdeModel
    (make_golden_stiff_wide.py) -- the oracle's odeint call reproduces their state rows (the 2550-equation sensitivity
    runs are what the file is for).  And the tight solutions that arbitrate: the augmented system integrated by column
    groups (odeint_oracle.tight_stiff_solution_by_columns) equals the full-system LSODA run at the same tolerances
    (stiff50_tight.npz: 15 - 45 minutes per vector) -- checked on half the columns of one vector."""
    from sysbio_modeling_amd import models_zoo
    from oracle.tolerances import parity_err
    gm = zoo('stiff50')
    w, wt = golden('stiff50_wide_ref.npz'), golden('stiff50_wide_tight.npz')
    P = models_zoo.stiff_ensemble(4096)[1]
    assert len(w['P']) == 32 and np.array_equal(w['P'], P[w['index']]) and len(np.unique(w['index'])) == 32
    assert w['index'].min() >= 3 and w['index'].max() == 4095 and np.array_equal(wt['P'], w['P'])
    for j in (0, 13, 31):
        Y = oo.simulate(gm, w['P'][j], w['t'], use_c=True)[w['idx']]
        assert np.allclose(Y, w['Y'][j], rtol=1e-12, atol=1e-14)
    # the reference's LSODA results sit about one parity unit from the tight ones, as on the first three vectors
    worst = max(parity_err(w['S'][j], wt['S'][j]) for j in range(32))
    assert 0.2 < worst < 3.0 and max(parity_err(w['Y'][j], wt['Y'][j]) for j in range(32)) < 1.0
    g, gt = golden('stiff50_ref.npz'), golden('stiff50_tight.npz')
    t = np.concatenate([[0.0], g['t'][g['idx']][:6]])
    Yc, Sc = oo.tight_stiff_solution_by_columns(gm, g['P'][1], t, group=25)
    assert parity_err(Yc[1:], gt['Y'][1][:6]) < 1e-2 and parity_err(Sc[1:], gt['S'][1][:6]) < 1e-2


def test_iex_sums_of_values_and_of_differences_are_the_same_scheme(zoo, golden):
    """The two accumulations of the extrapolated sensitivities -- sum_j w_j (T_j - S_n) added to S_n (sbm_iex_kernel) and
    sum_j w_j T_j (sbm_iex_seq_kernel: the weights add up to one) -- are one scheme: on a stiff golden vector the same steps
    and the same rows up to the rounding of the sums (3e3 eps relative to |S| per macro step)."""
    from oracle import iex_oracle
    g = golden('stiff50_ref.npz')
    gm = zoo('stiff50')
    t = g['t'][g['idx']][:3]
    kw = dict(rtol=3e-8, atol=3e-11, order=6)
    Ya, Sa, ia = iex_oracle.integrate(gm, g['P'][0], t, sums='differences', **kw)
    Yb, Sb, ib = iex_oracle.integrate(gm, g['P'][0], t, sums='values', **kw)
    assert ia['status'] == 0 and ib['status'] == 0 and abs(ia['n_steps'] - ib['n_steps']) <= 1 and ia['n_reject'] == ib['n_reject']
    assert np.max(np.abs(Ya - Yb) / (3e-8 * np.abs(Ya) + 3e-11)) <= 0.05
    assert np.max(np.abs(Sa - Sb) / (3e-8 * np.maximum(np.abs(Sa), 1e-6 * np.abs(Sa).max(axis=0)) + 3e-11)) <= 0.05
