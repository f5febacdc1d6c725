"""CPU: the C-ABI library loads and exports every symbol include/dbgsom_hip.h declares
(no compute calls without a GPU); argument errors come back as status codes."""
import ctypes
import os
import re

import numpy as np
import pytest

from dbgsom_amd import _native


def _declared():
    text = open(_native.HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dbgsom_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported_and_bound():
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    lib = _native.load()
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_native.SIGNATURES) == set(names)
    assert lib.dbgsom_abi_version() == _native.ABI_VERSION == 4


def test_argument_errors_are_status_codes_not_exceptions():
    lib = _native.load()
    # k = 3 is invalid; rejected before any device work, so this runs without a GPU
    rc = lib.dbgsom_bmu(None, 0, 10, 4, 4, None, None, 5, None, 3, 0, None, None, None)
    assert rc == -1
    assert b"k must be 1 or 2" in lib.dbgsom_last_error()
    rc = lib.dbgsom_bmu(None, 7, 10, 4, 4, None, None, 5, None, 1, 0, None, None, None)
    assert rc == -1 and b"x_dtype" in lib.dbgsom_last_error()
    assert lib.dbgsom_accumulate_workspace_bytes(0, 0, 0) == 0
    assert lib.dbgsom_accumulate_workspace_bytes(1000, 16, 10) > 1000 * 4
    with pytest.raises(ValueError):
        _native.call("dbgsom_smooth", None, 10, 4, None, 1.0, 9, None, None, None, None, 0, None)


def test_hip_backend_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dbgsom_amd.backend import HipBackend

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HipBackend()
