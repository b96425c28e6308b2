"""What a Project expects of a model object (reference model/abstract_model.py:4-24): a name, the
parameter order that fixes the layout of p, the number of state variables, and the two integrations
``simulate`` / ``calc_jacobian``.  ``OdeModel`` (ode_model.py) is the implementation; this base only
pins the attribute names the reference's callers read."""


class ModelABC(object):
    def __init__(self, model, n_vars, param_order, model_name):
        self.model_name = model_name
        self.param_order = list(param_order)
        self._n_vars = int(n_vars)
        self._model = model

    @property
    def n_vars(self):
        return self._n_vars

    def get_n_vars(self):           # the reference exposes both spellings (:18-21)
        return self._n_vars

    def simulate(self, *args, **kwargs):
        raise NotImplementedError("%s does not integrate" % type(self).__name__)

    def calc_jacobian(self, *args, **kwargs):
        raise NotImplementedError("%s has no sensitivity integration" % type(self).__name__)
