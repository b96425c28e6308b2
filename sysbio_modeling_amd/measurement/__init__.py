from .abstract_measurement import MeasurementABC
from .timecourse_measurement import TimecourseMeasurement

__all__ = ['MeasurementABC', 'TimecourseMeasurement']
