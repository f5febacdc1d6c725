"""Import-path parity with the reference (``dbgsom.SomClassifier.SomClassifier``)."""
from .som_classifier import SomClassifier  # noqa: F401
