"""Scale-factor record (reference project/loss_functions/abstract_scale_factor.py:6-42)."""


class ScaleFactorABC(object):
    def __init__(self, log_prior=None, log_prior_sigma=None):
        self._sf = None
        self._sf_gradient = None
        self.log_prior = log_prior
        self.log_prior_sigma = log_prior_sigma

    def __repr__(self):
        out = "SF value: %.5e\n" % self._sf
        if self.log_prior is not None:
            out += "SF log prior: %.4f\n" % self.log_prior
            out += "SF log prior sigma: %.4f\n" % self.log_prior_sigma
        return out
