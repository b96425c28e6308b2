from .linear_scale_factor import LinearScaleFactor
from .log_scale_factor import LogScaleFactor
from .squared_loss_function import SquareLossFunction
from .log_squared_loss_function import LogSquareLossFunction
from .normalized_squared_loss_function import NormalizedSquareLossFunction

__all__ = ['SquareLossFunction', 'LogSquareLossFunction', 'NormalizedSquareLossFunction',
           'LinearScaleFactor', 'LogScaleFactor']
