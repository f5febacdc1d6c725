"""MI355X-native batch-SOM training core with the DBGSOM estimator surface."""
from .backend import EpochResult, HipBackend, HotPathBackend  # noqa: F401

__all__ = ["HipBackend", "HotPathBackend", "EpochResult"]
