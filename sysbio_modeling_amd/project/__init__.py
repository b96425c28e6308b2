from .base_project import Project
from . import utils

__all__ = ['Project', 'utils']
