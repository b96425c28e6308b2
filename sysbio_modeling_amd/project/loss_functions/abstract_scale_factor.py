"""What every scale factor carries between two device evaluations.

The reference keeps the arithmetic on these objects (project/loss_functions/abstract_scale_factor.py:6-42
declares ``update_sf`` / ``update_sf_gradient`` / prior hooks for subclasses to fill in); here the
arithmetic runs inside csrc/sbm_core.hip::k_assemble and the object is a plain record: the value B and
its gradient dB/dtheta as last written back by ``SquareLossFunction._store``, and the optional Gaussian
prior on log B that ``Project.set_scale_factor_log_prior`` attaches.
"""


class ScaleFactorABC(object):
    __slots__ = ('_sf', '_sf_gradient', 'log_prior', 'log_prior_sigma')

    def __init__(self, log_prior=None, log_prior_sigma=None):
        if (log_prior is None) != (log_prior_sigma is None):
            raise ValueError("a scale-factor prior needs both its mean and its sigma (log units)")
        self.log_prior, self.log_prior_sigma = log_prior, log_prior_sigma
        self._sf_gradient = None
        self._sf = None

    def has_prior(self):
        return self.log_prior is not None

    def __repr__(self):
        text = ["SF value: %s" % ("not evaluated yet" if self._sf is None else "%.5e" % self._sf)]
        if self.has_prior():
            text.append("SF log prior: %.4f +- %.4f" % (self.log_prior, self.log_prior_sigma))
        return "\n".join(text) + "\n"
