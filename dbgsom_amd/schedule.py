"""Neighbourhood bandwidth schedule (host scalar math feeding the smoothing kernel).

Restates ``BaseSom._calculate_current_sigma`` (reference dbgsom/BaseSom.py:863-902) and the two
decay laws (:1001-1025).  Defaults: ``sigma_start = 0.2 sqrt(M)``, ``sigma_end = max(0.7,
0.05 sqrt(M))`` with M the CURRENT number of neurons; decay only in the coarse phase, evaluated
at the stretched time ``epoch / coarse_training_frac``; constant ``sigma_end`` in the fine phase.
"""
from __future__ import annotations

from math import exp, sqrt


def linear_decay(sigma_start, sigma_end, max_iter, current_iter, learning_rate=None):
    frac = current_iter / max_iter
    return sigma_start * (1 - frac) + sigma_end * frac


def exponential_decay(sigma_start, sigma_end, max_iter, current_iter, learning_rate):
    return sigma_end + (sigma_start - sigma_end) * exp(-learning_rate * current_iter)


_DECAY = {"linear": linear_decay, "exponential": exponential_decay}


def current_sigma(*, epoch, n_neurons, n_iter, phase, decay_function, learning_rate,
                  coarse_training_frac, sigma_start=None, sigma_end=None):
    s0 = 0.2 * sqrt(n_neurons) if sigma_start is None else sigma_start
    s1 = max(0.7, 0.05 * sqrt(n_neurons)) if sigma_end is None else sigma_end
    if phase != "coarse":
        return s1
    return _DECAY[decay_function](sigma_start=s0, sigma_end=s1, max_iter=n_iter,
                                  current_iter=epoch / coarse_training_frac,
                                  learning_rate=learning_rate)
