"""Time-course data container (reference measurement/timecourse_measurement.py:6-35;
plotting :37-45 is out of scope)."""
import numpy as np

from .abstract_measurement import MeasurementABC


class TimecourseMeasurement(MeasurementABC):
    """Values of one measured variable at given times, with optional standard deviations."""

    def __init__(self, variable_name, measurement_value, measurement_time, measurement_std=None):
        super(TimecourseMeasurement, self).__init__(variable_name, measurement_value, measurement_std)
        measurement_time = np.asarray(measurement_time, dtype=float)
        if len(self.values) != len(measurement_time):
            raise ValueError('Length of Standard Deviation Array Not Equal to Length of Timepoints')
        self.timepoints = measurement_time

    def drop_timepoint_zero(self):
        keep = self.timepoints != 0
        self.values = self.values[keep]
        self.std = self.std[keep]
        self.timepoints = self.timepoints[keep]

    def get_nonzero_measurements(self):
        """(values, std, timepoints) without the t = 0 points: the model starts from its
        initial condition there whatever the parameters are."""
        keep = self.timepoints != 0
        return self.values[keep], self.std[keep], self.timepoints[keep]
