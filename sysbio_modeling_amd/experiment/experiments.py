"""Experiment container (reference experiment/experiments.py:4-140)."""
import numpy as np


class Experiment(object):
    """A named set of measurements with the conditions they were taken under.

    name : str, must start with a letter or digit (names sort the residual rows)
    measurements : one measurement or an iterable of them (kept sorted by variable name)
    fixed_parameters : {param_name: value} parameters not optimised in this experiment
    experiment_settings : {setting: value} conditions that 'Shared' parameters depend on
    """

    def __init__(self, name, measurements, fixed_parameters=None, experiment_settings=None):
        if not name[0].isalnum():
            raise ValueError("Experiment names must start with a letter or number")
        self.name = name
        self.fixed_parameters = fixed_parameters
        self.settings = dict(experiment_settings) if experiment_settings is not None else {}
        self.initial_conditions = {}  # never read by the reference either (:40)
        self._measurements = []
        if hasattr(measurements, '__iter__'):
            for measurement in measurements:
                self.add_measurement(measurement)
        else:
            self.add_measurement(measurements)
        # filled by Project: OrderedDict model parameter name -> index in the project vector
        self.param_global_vector_idx = None

    @property
    def measurements(self):
        return self._measurements

    def drop_timepoint_zero(self, variable=None):
        for measurement in self._measurements:
            if variable is None or measurement.variable_name == variable:
                measurement.drop_timepoint_zero()

    def get_unique_timepoints(self, include_zero=False):
        """Sorted union of the timepoints of all measurements."""
        unique_timepoints = np.unique(np.concatenate([m.timepoints for m in self._measurements]))
        if not include_zero:
            unique_timepoints = unique_timepoints[unique_timepoints != 0]
        return unique_timepoints

    def get_variable_measurements(self, variable_name):
        for measurement in self._measurements:
            if measurement.variable_name == variable_name:
                return measurement
        raise KeyError('%s not in measurements' % variable_name)

    def add_measurement(self, measurement):
        for existing in self._measurements:
            if existing.variable_name == measurement.variable_name:
                raise KeyError('%s already has timeseries data associated with this experiment' % self.name)
        self._measurements.append(measurement)
        self._measurements.sort(key=lambda m: m.variable_name)
