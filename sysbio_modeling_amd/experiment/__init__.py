from .experiments import Experiment

__all__ = ['Experiment']
