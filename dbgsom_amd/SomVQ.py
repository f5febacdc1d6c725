"""Import-path parity with the reference (``dbgsom.SomVQ.SomVQ``)."""
from .som_vq import SomVQ  # noqa: F401
