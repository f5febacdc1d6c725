"""Import shim for the *reference* package (run only in the build container).

TEST/FIXTURE TOOLING -- never imported by the product, never shipped to the GPU box
as a dependency (``/root/reference`` does not exist there).

The reference (``/root/reference/dbgsom``) imports ``numba`` and ``seaborn.objects`` and
``sklearn.base.check_X_y``; none of those resolve in this image, and the reference answers an
ImportError with ``sys.exit()`` (``dbgsom/BaseSom.py:34-36``).  The shim registers inert stand-ins
for the two *plotting/JIT* modules (identity ``njit`` decorator, ``prange = range``), so the two
``numba`` functions run as the plain serial NumPy/Python they are written in -- which is the
parity target (serial semantics of the error sum).  No reference source is copied.

Usage (note the env var: the reference tree must stay free of ``__pycache__``):

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _identity_decorator(*dargs, **dkwargs):
    # supports both ``@njit`` and ``@njit(parallel=True, ...)``
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn

    return wrap


def install() -> None:
    """Make ``import dbgsom.SomVQ`` work against ``/root/reference``."""
    if os.environ.get("PYTHONDONTWRITEBYTECODE") != "1":
        raise RuntimeError(
            "export PYTHONDONTWRITEBYTECODE=1 before importing the reference "
            "(keeps /root/reference free of __pycache__)"
        )
    sys.dont_write_bytecode = True
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"{REFERENCE_ROOT} is not present (GPU box?)")

    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")
        nb.njit = _identity_decorator
        nb.jit = _identity_decorator
        nb.prange = range
        sys.modules["numba"] = nb
    if "seaborn" not in sys.modules:
        sb = types.ModuleType("seaborn")
        so = types.ModuleType("seaborn.objects")
        sb.objects = so
        sys.modules["seaborn"] = sb
        sys.modules["seaborn.objects"] = so

    import sklearn.base
    import sklearn.utils
    import sklearn.utils.validation

    for name in ("check_X_y", "check_array"):
        if not hasattr(sklearn.base, name):
            setattr(sklearn.base, name, getattr(sklearn.utils, name))
    if not hasattr(sklearn.base, "check_is_fitted"):
        sklearn.base.check_is_fitted = sklearn.utils.validation.check_is_fitted

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def versions() -> dict:
    import networkx
    import numpy
    import scipy
    import sklearn

    return {
        "python": sys.version.split()[0],
        "numpy": numpy.__version__,
        "scipy": scipy.__version__,
        "scikit-learn": sklearn.__version__,
        "networkx": networkx.__version__,
        "reference": "SandroMartens/DBGSOM @ 2024_10_08",
    }
