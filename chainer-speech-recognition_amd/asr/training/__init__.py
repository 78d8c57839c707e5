"""asr.training of the reference (asr/training/__init__.py): the run-time environment of a training process and its epoch counter."""
from .environment import Environment
from .iteration import Iteration

__all__ = ["Environment", "Iteration"]
