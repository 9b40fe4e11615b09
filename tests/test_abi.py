"""The C-ABI boundary: libovtk_amd.so (the HIP build) loads on a GPU-less machine and exports every function
include/ovtk_amd.h declares; the ctypes mirror lists the same names; compute entry points fail loudly (never fall
back to a CPU path) when no HIP device is present."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from openvino_tokenizers_amd import _lib as L

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "ovtk_amd.h"


def declared_functions():
    txt = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(ovtk_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def hip_build():
    subprocess.run(["make", "-C", str(ROOT / "openvino_tokenizers_amd" / "csrc"), "-s"], check=True)
    return L.load()


def test_header_and_ctypes_list_agree():
    assert declared_functions() == sorted(L.EXPORTS)


def test_library_exports_every_declared_symbol(hip_build):
    for name in declared_functions():
        assert hasattr(hip_build, name), f"{name} is declared in include/ovtk_amd.h but not exported"
    assert hip_build.ovtk_abi_version() == 1002


def test_no_cpu_fallback_without_a_device(hip_build):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert hip_build.ovtk_device_name() is None
    h = C.c_void_p()
    p = L.RegexSplitParams(b"\\s+", 3, b"remove", 0, -1, 0)
    pat = b"'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+"
    p = L.RegexSplitParams(pat, len(pat), b"isolate", 0, -1, 0)
    assert hip_build.ovtk_regex_split_create(C.byref(p), C.byref(h)) == L.E_HIP
    assert b"no CPU execution path" in hip_build.ovtk_last_error()
    b = np.zeros(1, np.int32)
    assert hip_build.ovtk_fuze_ragged(b.ctypes.data, b.ctypes.data, C.c_int64(1), b.ctypes.data, b.ctypes.data, C.c_int64(1),
                                      b.ctypes.data, b.ctypes.data, L.MEM_HOST, 0, None) == L.E_HIP
