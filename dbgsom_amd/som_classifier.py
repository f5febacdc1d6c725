"""Supervised estimator: growing SOM with per-prototype class statistics.

Mirrors ``dbgsom/SomClassifier.py`` of the reference (:19-220).  The training path is the same
accelerated hot path; prototype labelling uses one more BMU pass; ``predict`` /
``predict_proba`` go through ``transform`` (LARS sparse coding on the host) exactly as the
reference does -- that part is outside the accelerated path.
"""
from __future__ import annotations

from statistics import mode

import numpy as np
from sklearn.base import ClassifierMixin, TransformerMixin
from sklearn.utils import check_array, check_X_y
from sklearn.utils.validation import check_is_fitted

from .base import BaseSom


class SomClassifier(BaseSom, TransformerMixin, ClassifierMixin):
    """Directed batch growing SOM classifier (see ``BaseSom`` for parameters)."""

    def _check_input_data(self, X, y):
        # (X's finite check rides on the device's column sums when it can; y is checked here as ever)
        self._finite_deferred = self._finite_check_on_device()
        X, y = check_X_y(X=X, y=y, ensure_min_samples=4, dtype=[np.float64, np.float32],
                         **self._finite_kw(not self._finite_deferred))
        return X, y

    def _label_prototypes(self, X, y) -> None:
        """Majority label and class frequencies of every prototype's Voronoi set
        (SomClassifier.py:130-152); a dead prototype gets label -1."""
        _, winners = self._get_winning_neurons(X, n_bmu=1)
        m, n_classes = len(self.neurons_), self.classes_.shape[0]
        hits = self._node_stats["hit_count"]
        labels = np.empty(m, dtype=np.int64)
        probs = np.zeros((m, n_classes))
        for j in range(m):
            members = y[winners == j]
            if len(members) == 0:
                labels[j] = -1
                probs[j, -1] = 0 / hits[j] if hits[j] > 0 else 1
                continue
            labels[j] = mode(members)
            ids, counts = np.unique(members, return_counts=True)
            probs[j, ids] = counts / hits[j] if hits[j] > 0 else 1
        self._lattice.write_attributes({"label": labels, "probabilities": probs})

    def predict(self, X) -> np.ndarray:
        check_is_fitted(self)
        X = check_array(X, dtype=[np.float64, np.float32])
        return self.classes_[np.argmax(self.predict_proba(X=X), axis=1)]

    def predict_proba(self, X) -> np.ndarray:
        """Class probabilities: sparse code over the prototypes times the prototypes' class
        frequencies, rows normalised (SomClassifier.py:178-220)."""
        check_is_fitted(self)
        X = check_array(X, dtype=[np.float64, np.float32])
        if self.vertical_growth:
            _, winners = self._get_winning_neurons(X, n_bmu=1)
            rows = []
            for sample, w in zip(X, winners):
                attrs = self.som_.nodes[self.neurons_[w]]
                if "som" in attrs:
                    # (a child map knows the classes of its Voronoi set only, by this map's class CODES --
                    #  what _grow_vertical hands it as y: back into this map's columns)
                    child = attrs["som"]
                    row = np.zeros(self.classes_.shape[0])
                    row[np.asarray(child.classes_, dtype=np.int64)] = child.predict_proba(sample[None, :])[0]
                    rows.append(row)
                else:
                    rows.append(attrs["probabilities"])
            return np.array(rows)
        code = self.transform(X)
        raw = code @ self._extract_values_from_graph("probabilities")
        return raw / raw.sum(axis=1)[np.newaxis].T
