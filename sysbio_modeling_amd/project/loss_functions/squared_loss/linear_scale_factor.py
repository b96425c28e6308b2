"""Linear scale factor B = sum(s d / sigma^2) / sum(s^2 / sigma^2).

The arithmetic of the reference's ``update_sf`` / ``update_sf_gradient``
(project/loss_functions/squared_loss/linear_scale_factor.py:27-42) runs on the
device inside csrc/sbm_core.hip::k_assemble; this object only carries the latest
values back to the caller (``Project.scale_factors[measure].sf``).
"""
import numpy as np

from ..abstract_scale_factor import ScaleFactorABC


def scale_factor_entropy(sim_dot_sim, sim_dot_exp, log_prior, log_prior_sigma, temperature=1.0):
    """log of  integral du exp(-a_k/(2T) (B(u) - B*)^2 - (u + log B* - prior)^2 / (2 sigma^2)),  B(u) = e^u B*,
    B* = b_k / a_k: the SloppyCell scale-factor entropy the reference evaluates per group
    (linear_scale_factor.py:13-18,63-81; a_k = sum s^2/sigma^2, b_k = sum s d/sigma^2).  Host-side
    quadrature, as in the reference: it feeds the serial sampler's free energy, not the fitting loop."""
    import scipy.integrate
    b_best = sim_dot_exp / sim_dot_sim
    log_b = np.log(b_best)

    def integrand(u):
        with np.errstate(over='ignore'):      # u -> +inf: exp(-inf) = 0 is the intended value
            b = np.exp(u) * b_best
            return np.exp(-sim_dot_sim / (2.0 * temperature) * (b - b_best) ** 2
                          - (u + log_b - log_prior) ** 2 / (2.0 * log_prior_sigma ** 2))
    ans, _ = scipy.integrate.quad(integrand, -np.inf, np.inf, limit=1000)
    return np.log(ans)


class LinearScaleFactor(ScaleFactorABC):
    def __init__(self, log_prior=None, log_prior_sigma=None):
        super(LinearScaleFactor, self).__init__(log_prior, log_prior_sigma)
        self._sf = 1.0

    @property
    def sf(self):
        return self._sf

    @property
    def gradient(self):
        return None if self._sf_gradient is None else np.array(self._sf_gradient, copy=True)

    def calc_sf_prior_residual(self):
        """(log B - log_prior) / sigma  (reference :55-61)."""
        if self.log_prior is None:
            return None
        return (np.log(self._sf) - self.log_prior) / self.log_prior_sigma

    def calc_sf_prior_gradient(self):
        """d log B / d theta = (dB/dtheta) / B  (reference :44-53)."""
        if self.log_prior is None or self._sf_gradient is None:
            return None
        return self._sf_gradient / self._sf
