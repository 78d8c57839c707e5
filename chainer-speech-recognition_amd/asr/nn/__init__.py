from .nn import *  # noqa: F401,F403
from . import nn as _nn
from ..link import Chain, Link, Parameter, initializers  # noqa: F401
