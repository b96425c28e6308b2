from .abstract_model import ModelABC
from .ode_model import OdeModel

__all__ = ['ModelABC', 'OdeModel']
