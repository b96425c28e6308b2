"""Host-side control loops (sysbio_modeling_amd/_control.py) on synthetic integrators: no GPU involved.

``run`` stands in for the device call: an "extrapolated result" with a known error constant per vector,
E(n) = exact + c_v / n^4, optionally failing below some n (a Newton failure on too coarse a grid)."""
import numpy as np
import pytest

from sysbio_modeling_amd import _control


def _fake_run(exact, c, order=4, fail_below=None, calls=None, n0=16):
    def run(idx, mult):
        n = n0 * mult
        if calls is not None:
            calls.append((len(idx), n))
        vals = exact[idx] + c[idx][:, None] / float(n) ** order
        st = np.zeros(len(idx), dtype=np.int32)
        if fail_below is not None:
            bad = n < fail_below[idx]
            st[bad] = 4
            vals[bad] = np.nan
        return {'y': vals, 'aux': vals * 2.0}, st, np.full(len(idx), 3 * n, dtype=np.int32)
    return run


def test_vectors_leave_the_loop_as_they_converge():
    exact = np.array([[1.0, 2.0, 0.0], [3.0, -1.0, 0.5], [0.2, 0.1, 0.3]])
    c = np.array([1e1, 1e4, 1e7])                       # easy, medium, hard
    calls = []
    out, st, spent, levels = _control.controlled_doubling(_fake_run(exact, c, calls=calls), 3, ['y'], 1e-9, 1e-12,
                                                          max_doublings=12)
    assert st.tolist() == [0, 0, 0]
    assert levels[0] < levels[1] < levels[2]
    # the returned values are the finer of the two compared runs: error well inside the tolerance
    assert np.all(np.abs(out['y'] - exact) <= 1e-9 * np.maximum(np.abs(exact), 1e-3 * np.abs(exact).max(axis=1, keepdims=True)) + 1e-12)
    assert np.allclose(out['aux'], 2.0 * out['y'])      # every output follows, not only the compared one
    # fewer vectors per call as the loop goes on, steps doubling
    sizes = [n_vec for n_vec, _ in calls]
    assert sizes[0] == 3 and sizes[-1] == 1 and sizes == sorted(sizes, reverse=True)
    assert [n for _, n in calls] == [16 * 2 ** k for k in range(len(calls))]
    assert spent[2] == sum(3 * n for _, n in calls) and spent[0] < spent[1] < spent[2]


def test_failed_coarse_runs_do_not_stop_the_loop_and_unreachable_tolerances_are_flagged():
    exact = np.ones((2, 4))
    c = np.array([1e-3, 1e-3])
    fail_below = np.array([0, 128])                     # vector 1: Newton fails on the two coarsest grids
    out, st, spent, levels = _control.controlled_doubling(_fake_run(exact, c, fail_below=fail_below, n0=32), 2, ['y'],
                                                          1e-9, 1e-12, max_doublings=8)
    assert st.tolist() == [0, 0] and levels.tolist() == [1, 3]    # 32|64 agree; 128|256 the first valid pair
    assert np.all(np.isfinite(out['y']))
    out, st, _, levels = _control.controlled_doubling(_fake_run(exact, np.array([1e12, 1e12])), 2, ['y'], 1e-12, 1e-15,
                                                      max_doublings=3)
    assert st.tolist() == [_control.SBM_TOL_NOT_REACHED] * 2 and levels.tolist() == [3, 3]
    assert np.allclose(out['y'], exact + 1e12 / (16 * 8) ** 4)        # the finest result is what comes back
    # a vector that still fails on the finest grid keeps that status
    out, st, _, _ = _control.controlled_doubling(_fake_run(exact, c, fail_below=np.array([0, 10 ** 9])), 2, ['y'],
                                                 1e-9, 1e-12, max_doublings=4)
    assert st.tolist() == [0, 4]


def test_stiff_fallback_touches_only_failed_vectors():
    V = 5
    explicit = {'y': np.arange(V * 2, dtype=float).reshape(V, 2), 'z': np.ones((V, 1))}
    st = np.array([0, 1, 0, 3, 0], dtype=np.int32)
    seen = []

    def controlled(idx):
        seen.append(idx.tolist())
        return ({'y': -np.ones((len(idx), 2)), 'z': np.zeros((len(idx), 1))}, np.zeros(len(idx), dtype=np.int32),
                np.full(len(idx), 1000), np.ones(len(idx), dtype=np.int32))
    out, st2, steps, stiff = _control.with_stiff_fallback(
        lambda: ({k: v.copy() for k, v in explicit.items()}, st, np.full(V, 7)), controlled, V)
    assert seen == [[1, 3]] and stiff.tolist() == [False, True, False, True, False]
    assert st2.tolist() == [0] * V and steps.tolist() == [7, 1007, 7, 1007, 7]
    assert np.array_equal(out['y'][[0, 2, 4]], explicit['y'][[0, 2, 4]]) and np.all(out['y'][[1, 3]] == -1.0)
    # nothing failed: the controlled integrator is never called
    out, st2, _, stiff = _control.with_stiff_fallback(lambda: (explicit, np.zeros(V, dtype=np.int32), np.full(V, 7)),
                                                     lambda idx: pytest.fail("not expected"), V)
    assert not stiff.any()
    # a model without an implicit integrator: the failures stand
    out, st2, _, stiff = _control.with_stiff_fallback(lambda: (explicit, st, np.full(V, 7)), None, V)
    assert st2.tolist() == st.tolist() and not stiff.any()


def test_control_loops_on_torch_tensors():
    torch = pytest.importorskip('torch')
    exact = np.array([[1.0, 2.0], [3.0, 4.0]])
    c = np.array([1e3, 1e7])
    base = _fake_run(exact, c)

    def run(idx, n):
        o, st, ns = base(idx, n)
        return {k: torch.from_numpy(v) for k, v in o.items()}, torch.from_numpy(st), torch.from_numpy(ns)
    out, st, _, levels = _control.controlled_doubling(run, 2, ['y'], 1e-9, 1e-12, max_doublings=10)
    assert st.tolist() == [0, 0] and levels[0] < levels[1] and isinstance(out['y'], torch.Tensor)
    assert np.allclose(out['y'].numpy(), exact, rtol=1e-9)
