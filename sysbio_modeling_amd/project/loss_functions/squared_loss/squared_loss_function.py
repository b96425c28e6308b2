"""Weighted squared loss with optional linear scale factors.

Configuration object for the loss the assembly kernel evaluates
(csrc/sbm_core.hip::k_assemble):  r = (B s - d) / sigma,  J = B dS + s (x) dB
(reference project/loss_functions/squared_loss/squared_loss_function.py:27-80,
abstract_loss_function.py:22-120).  It holds the scale-factor groups, their
priors and the last values computed on the device.
"""
from copy import deepcopy

from ...utils import OrderedHashDict
from .linear_scale_factor import LinearScaleFactor


class SquareLossFunction(object):
    loss_type = 0                    # SBM_LOSS_SQUARE (include/sbm.h)
    normalize_sigma_by_mean = False  # NormalizedSquareLossFunction sets this

    def __init__(self, sf_groups=None, sf_type=LinearScaleFactor):
        self._scale_factors = OrderedHashDict()
        if sf_groups is not None:
            if isinstance(sf_groups, (str, frozenset)):
                sf_groups = [sf_groups]
            for measure_group in sf_groups:
                self._scale_factors[measure_group] = sf_type()

    @property
    def scale_factors(self):  # noqa: F811 -- instance view shadows the class marker
        return deepcopy(self._scale_factors)

    @property
    def groups(self):
        """Scale-factor groups as lists of measure names, in definition order."""
        return [[k] if isinstance(k, str) else sorted(k) for k in self._scale_factors.keys()]

    def group_index(self, measure_name):
        for gi, g in enumerate(self._scale_factors.keys()):
            if g == measure_name or (not isinstance(g, str) and measure_name in g):
                return gi
        return -1

    def set_scale_factor_priors(self, measure_name, log_scale_factor_prior, log_sigma_scale_factor):
        try:
            sf = self._scale_factors[measure_name]
        except KeyError:
            raise KeyError("%s not present as a scale factor" % (measure_name,))
        sf.log_prior = log_scale_factor_prior
        sf.log_prior_sigma = log_sigma_scale_factor

    def _store(self, sf_values, sf_gradients=None):
        for gi, sf in enumerate(self._scale_factors.values()):
            sf._sf = float(sf_values[gi])
            if sf_gradients is not None:
                sf._sf_gradient = sf_gradients[gi].copy()
