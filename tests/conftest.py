"""Test tiers
  -m "not gpu"  CPU: oracle vs the reference's known answers / HF golden vectors, host logic, C-ABI exports,
                and the kernel *logic* through the SIMT-emulator build of the same sources (tests/emu).
  -m gpu        the parity tests proper: libovtk_amd.so on an MI355X through the C ABI vs the oracle.
"""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel sources compiled against tests/emu (CPU fibers).  Test infrastructure only."""
    import fcntl
    from openvino_tokenizers_amd import _lib as L
    (ROOT / "tests" / "emu" / "build").mkdir(parents=True, exist_ok=True)
    with open(ROOT / "tests" / "emu" / "build" / ".lock", "w") as lock:   # pytest -n: one worker builds, the others wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-C", str(ROOT / "openvino_tokenizers_amd" / "csrc"), "-s", "emu"], check=True)
    return L.load(ROOT / "tests" / "emu" / "build" / "libovtk_emu.so")


@pytest.fixture(scope="session")
def hip_lib():
    import torch
    from openvino_tokenizers_amd import _lib as L
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    lib = L.load()
    assert lib.ovtk_device_name() is not None
    return lib


class Backend:
    """Which library runs the op + where the data tensors live."""

    def __init__(self, name, lib):
        self.name, self.lib = name, lib

    def data(self, arrays):
        if self.name == "hip-device":
            import torch
            return [torch.as_tensor(a, device="cuda") for a in arrays]
        return list(arrays)

    @staticmethod
    def host(x):
        import numpy as np
        return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


@pytest.fixture(params=["emu", pytest.param("hip-host", marks=pytest.mark.gpu),
                        pytest.param("hip-device", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "emu":
        return Backend("emu", request.getfixturevalue("emu_lib"))
    return Backend(request.param, request.getfixturevalue("hip_lib"))


@pytest.fixture(params=[pytest.param("hip-device", marks=pytest.mark.gpu)])
def gpu_backend(request):
    return Backend(request.param, request.getfixturevalue("hip_lib"))
