"""Base of the measurement containers.

Same validation as the reference (measurement/abstract_measurement.py:8-24): without error bars every
point weighs the same (sigma = 1), a zero sigma is refused (the residual divides by it), and values and
error bars must pair up.  The arrays end up, flattened over all experiments, in ``sbm_project_desc``'s
``row_data`` / ``row_sigma`` (include/sbm.h).
"""
import numpy as np


def _column(x, what):
    a = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if a.ndim != 1:
        raise ValueError("%s must be one-dimensional" % what)
    return a


class MeasurementABC(object):
    def __init__(self, variable_name, measurement_value, measurement_std=None):
        values = _column(measurement_value, 'measurement values')
        std = np.ones_like(values) if measurement_std is None else _column(measurement_std, 'measurement std')
        if std.shape != values.shape:
            raise ValueError('Length of Standard Deviation Array Not Equal to Length of Measurements')
        if not np.all(std != 0):
            raise ValueError('Standard deviation of measurement cannot be 0')
        self.variable_name, self.values, self.std = variable_name, values, std
