from .squared_loss import SquareLossFunction, LinearScaleFactor

__all__ = ['SquareLossFunction', 'LinearScaleFactor']
