"""Measurement base type (reference measurement/abstract_measurement.py:8-24)."""
from abc import ABCMeta, abstractmethod

import numpy as np


class MeasurementABC(metaclass=ABCMeta):
    @abstractmethod
    def __init__(self, variable_name, measurement_value, measurement_std=None):
        measurement_value = np.asarray(measurement_value, dtype=float)
        if measurement_std is None:
            # unweighted fit: sigma = 1 for every point (reference :12-13)
            measurement_std = np.ones_like(measurement_value)
        measurement_std = np.asarray(measurement_std, dtype=float)
        if np.any(measurement_std == 0):
            raise ValueError('Standard deviation of measurement cannot be 0')
        if len(measurement_value) != len(measurement_std):
            raise ValueError('Length of Standard Deviation Array Not Equal to Length of Measurements')
        self.variable_name = variable_name
        self.values = measurement_value
        self.std = measurement_std
