from .squared_loss import (SquareLossFunction, LogSquareLossFunction, NormalizedSquareLossFunction,
                           LinearScaleFactor, LogScaleFactor)

__all__ = ['SquareLossFunction', 'LogSquareLossFunction', 'NormalizedSquareLossFunction',
           'LinearScaleFactor', 'LogScaleFactor']
