"""Import-path parity with the reference (``dbgsom.BaseSom``)."""
from .base import BaseSom  # noqa: F401
from .schedule import exponential_decay, linear_decay  # noqa: F401
