"""r = (log(B sim) - log(data)) / sigma  (reference log_squared_loss_function.py:9-98).
Configuration object: the arithmetic runs in csrc/sbm_core.hip::k_assemble (SBM_LOSS_LOG_SQUARE)."""
from .squared_loss_function import SquareLossFunction
from .log_scale_factor import LogScaleFactor


class LogSquareLossFunction(SquareLossFunction):
    loss_type = 1  # SBM_LOSS_LOG_SQUARE

    def __init__(self, sf_groups=None):
        super(LogSquareLossFunction, self).__init__(sf_groups, LogScaleFactor)
