"""Geometric-mean scale factor of the log-square loss.

log B = sum(w (log d - log s)) / sum(w),  w = (d / sigma)^2
(reference project/loss_functions/squared_loss/log_scale_factor.py:17-36; evaluated on the
device in csrc/sbm_core.hip::k_assemble, this object carries the last values)."""
from .linear_scale_factor import LinearScaleFactor


class LogScaleFactor(LinearScaleFactor):
    def __init__(self, log_prior=None, log_prior_sigma=None):
        super(LogScaleFactor, self).__init__(log_prior, log_prior_sigma)
        self._sf = 0  # reference :12
