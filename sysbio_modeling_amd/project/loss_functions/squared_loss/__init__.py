from .linear_scale_factor import LinearScaleFactor
from .squared_loss_function import SquareLossFunction

__all__ = ['SquareLossFunction', 'LinearScaleFactor']
