"""Host-side helpers of the project layer.

``OrderedHashDict`` keeps the key rules of the reference's container for
scale-factor groups (project/utils.py:98-158, pinned by tests/test_Project_Utils.py).
The reference's four sampling functions (project/utils.py:10-89) have no Python
counterpart here: 'direct' / 'sum' sampling runs inside the fused assembly kernel
(csrc/sbm_core.hip::k_assemble); ``sample_index`` is the one piece of them the
host needs -- which grid points the kernels must land on.
"""
from collections import OrderedDict

import numpy as np

N_SIM_POINTS = 1000  # project/base_project.py:419,510


def simulation_grid(t_end):
    """The reference's output grid for one experiment (project/base_project.py:418-419)."""
    return np.linspace(0, t_end, N_SIM_POINTS)


def sample_index(model_timepoints, measure_timepoints):
    """First grid point at or after each measurement time: the reference samples
    the trajectory there and does NOT interpolate (project/utils.py:18-21)."""
    return np.searchsorted(model_timepoints, measure_timepoints)


class OrderedHashDict(OrderedDict):
    """Ordered dict keyed by strings or frozensets of strings; a string can be
    looked up directly or through the frozenset that contains it, and may appear
    only once across all keys."""

    def _find(self, key):
        if OrderedDict.__contains__(self, key):
            return key
        if isinstance(key, str):
            for k in OrderedDict.keys(self):
                if not isinstance(k, str) and key in k:
                    return k
        raise KeyError("%s not in dictionary or in any of the groups in the dictionary" % (key,))

    def __getitem__(self, key):
        return OrderedDict.__getitem__(self, self._find(key))

    def __contains__(self, key):
        try:
            self._find(key)
            return True
        except (KeyError, TypeError):
            return False

    def __setitem__(self, key, value):
        if not isinstance(key, (str, frozenset)):
            raise TypeError("Keys can only be strings, or frozen sets of strings")
        if isinstance(key, frozenset):
            if not OrderedDict.__contains__(self, key):
                for member in key:
                    if not isinstance(member, str):
                        raise TypeError("Every element within the frozenset has to be a string")
                    if member in self:
                        raise KeyError("%s already in dict in a hashgroup" % (key,))
            OrderedDict.__setitem__(self, key, value)
        else:
            if key in self and not OrderedDict.__contains__(self, key):
                # present, but only as a member of a group: must be updated through the group
                raise KeyError("%s already in dict in a hashgroup" % key)
            OrderedDict.__setitem__(self, key, value)


def exp_param_transform(project_param_vector):
    """theta -> p for log-space optimisation (project/utils.py:161-182)."""
    return np.exp(project_param_vector)


def exp_param_transform_derivative(project_param_vector):
    """d p / d theta of the transform above (project/utils.py:185-204)."""
    return np.exp(project_param_vector)
