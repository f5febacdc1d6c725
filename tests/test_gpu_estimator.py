"""Whole fits on the MI355X through the default (HIP) backend against the reference's recorded
results -- the drop-in check for SomVQ / SomClassifier fit / predict / fit_predict."""
import numpy as np
import pytest

from tests import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gi.FIT_CASES)
def test_fit_on_gpu_matches_reference(name):
    from dbgsom_amd import SomClassifier, SomVQ
    from dbgsom_amd.backend import HipBackend

    g = gi.load(name)
    X, y = gi.case_X(name)
    cls = SomClassifier if name in gi.CLF_CASES else SomVQ
    est = cls(**gi.EST_KWARGS[name])  # backend=None -> HipBackend
    est.fit(X, y) if y is not None else est.fit(X)
    assert isinstance(est._engine(), HipBackend)
    assert est.n_iter_ == int(g["final_n_iter"])
    assert [tuple(n) for n in g["final_neurons"]] == est.neurons_
    np.testing.assert_allclose(est.weights_, g["final_weights"], rtol=1e-5)  # north_star bound
    np.testing.assert_allclose(est.weights_, g["final_weights"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(est.quantization_error_, float(g["final_qe"]), rtol=1e-10)
    assert est.topographic_error_ == float(g["final_te"])
    if name not in gi.CLF_CASES:
        assert np.array_equal(est.labels_, g["final_labels"])   # BMU indices bit-exact
        assert np.array_equal(est.predict(X), g["final_labels"])
        assert np.array_equal(cls(**gi.EST_KWARGS[name]).fit_predict(X), g["final_labels"])
    else:
        assert np.array_equal(est.predict(X), g["final_predict"])
        assert est.score(X, y) == float(g["final_score"])


def test_known_answers_digits_gpu():
    from dbgsom_amd import SomVQ

    X, _ = gi.case_X("digits_f64")
    est = SomVQ(random_state=0).fit(X)
    assert est.n_iter_ == 112 and len(est.neurons_) == 25
    np.testing.assert_allclose(est.quantization_error_, 23.80011502226172, rtol=1e-12)
    assert est.topographic_error_ == 0.05008347245409015
    np.testing.assert_allclose(est.weights_.sum(), 7793.246057345110, rtol=1e-11)
    assert est.labels_[:10].tolist() == [23, 16, 6, 18, 19, 14, 17, 2, 21, 5]
