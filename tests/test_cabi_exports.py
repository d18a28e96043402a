"""CPU test: libarachne_amd.so loads and exports every symbol include/arachne_amd.h declares (no compute without a GPU),
and refuses to open a context when no GPU is present instead of falling back to anything."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as ge
from arachne_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ge.build_product()
    return ctypes.CDLL(api.LIB_PATH)


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "arachne_amd.h")).read()
    names = set(re.findall(r"\b(arx_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 14
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_backend_is_hip(lib):
    lib.arx_backend.restype = ctypes.c_char_p
    assert lib.arx_backend() == b"hip:gfx950"


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.ArachneError, match="no HIP device|CPU fallback"):
        api.Reference("/tmp/whatever")
