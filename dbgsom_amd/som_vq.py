"""Unsupervised estimator: vector quantisation / clustering with a growing SOM.

Mirrors ``dbgsom/SomVQ.py`` of the reference (:16-152): ``fit`` / ``predict`` / ``fit_predict``
(from ``ClusterMixin``) / ``labels_``; a sample's label is the index of its best matching unit.
"""
from __future__ import annotations

import numpy as np
from sklearn.base import ClusterMixin, TransformerMixin
from sklearn.utils import check_array
from sklearn.utils.validation import check_is_fitted

from .base import BaseSom


class SomVQ(BaseSom, ClusterMixin, TransformerMixin):
    """Directed batch growing SOM used as a vector quantiser (see ``BaseSom`` for parameters)."""

    def _check_input_data(self, X, y=None):
        # float32 is kept as float32 (the device stores it as such); anything else -> float64
        # (the finite check rides on the device's column sums when it can: BaseSom._assert_finite_from_moments)
        self._finite_deferred = self._finite_check_on_device()
        X = check_array(array=X, ensure_min_samples=4, dtype=[np.float64, np.float32],
                        **self._finite_kw(not self._finite_deferred))
        return X, None  # any y is ignored

    def _label_prototypes(self, X, y=None) -> None:
        self._lattice.write_attributes({"label": np.arange(len(self._lattice))})

    def predict(self, X) -> np.ndarray:
        """Index of the closest prototype for every sample (SomVQ.py:130-148)."""
        check_is_fitted(self)
        if not self._is_resident(X):
            # integer / half input is converted like the reference's engine does (sklearn's
            # NearestNeighbors); float32 stays float32
            X = check_array(X, dtype=[np.float64, np.float32])
        _, labels = self._get_winning_neurons(X, n_bmu=1)
        return labels

    def _fit(self, X) -> None:
        self.labels_ = self.predict(X)
