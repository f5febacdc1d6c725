"""MI355X-native batch-SOM training core with the DBGSOM estimator surface.

    from dbgsom_amd import SomVQ, SomClassifier
"""
from .backend import EpochResult, HipBackend, HotPathBackend  # noqa: F401
from .som_classifier import SomClassifier  # noqa: F401
from .som_vq import SomVQ  # noqa: F401

__all__ = ["SomVQ", "SomClassifier", "HipBackend", "HotPathBackend", "EpochResult"]
