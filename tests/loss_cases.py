"""Hand-built loss-function cases of the reference's tests/test_Loss_Functions.py, shared by the
oracle pin (test_oracle_golden.py) and the GPU frame-level parity tests (test_gpu_loss_frames.py)."""
import numpy as np

from oracle.project_oracle import ProjectOracle


class RowsOnlyOracle(ProjectOracle):
    """ProjectOracle with the simulate step replaced by given sims / Jacobian, to reach the
    loss-function identities the reference tests on hand-built frames."""

    def __init__(self, rows, sims, J, sf_groups, q):
        self._rows, self._sims, self._J = rows, np.asarray(sims, float), J
        self.sf_groups = [[g] if isinstance(g, str) else sorted(g) for g in sf_groups]
        self.n_project_params = q
        self.parameter_priors = {}
        self.sf_priors = {}
        self.compat = True
        self.scale_factors = [1.0] * len(self.sf_groups)

    def rows(self):
        return self._rows

    def simulate_rows(self, theta, with_jacobian=False):
        return self._sims, None, (self._J if with_jacobian else None)

    def _prior_rows(self, theta):
        return []


def lin_square_rows(scale_lin=1.0, scale_sq=1.0, noise=None):
    t = np.linspace(0, 100, 101)                                        # test_Loss_Functions.py:21-24
    sims = np.concatenate([2 * t, t ** 2])
    data = np.concatenate([2 * t * scale_lin, t ** 2 * scale_sq])
    if noise is not None:
        data = data - noise
    rows = [(0, 'Lin', d, 1.0, tt) for d, tt in zip(data[:101], t)] + \
           [(1, 'Square', d, 1.0, tt) for d, tt in zip(data[101:], t)]
    return rows, sims
