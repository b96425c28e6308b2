"""Square loss with sigma multiplied by the measurement mean (a relative loss), reference
normalized_squared_loss_function.py:6-46.  Pure preprocessing of sigma: Project multiplies
row_sigma by row_data when it flattens the rows, the kernel sees SBM_LOSS_SQUARE."""
from .squared_loss_function import SquareLossFunction


class NormalizedSquareLossFunction(SquareLossFunction):
    normalize_sigma_by_mean = True
