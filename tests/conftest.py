import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sgl-cpu-tests_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    """Returns (tensors, meta); fp8 tensors stored as uint8 (`__fp8` suffix) are viewed back."""
    import torch
    from safetensors import safe_open
    tensors = {}
    with safe_open(os.path.join(GOLDEN, name + ".safetensors"), framework="pt") as f:
        meta = f.metadata()
        for k in f.keys():
            t = f.get_tensor(k)
            if k.endswith("__fp8"):
                tensors[k[:-5]] = t.view(torch.float8_e4m3fn)
            else:
                tensors[k] = t
    return tensors, meta


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture
def knob(monkeypatch):
    """Sets SGLK_* developer knobs for ONE test: `knob(SGLK_MOE_TILE_M=96, SGLK_PERSIST=None)` (None = unset).

    The library reads its environment once (sgl-cpu-tests_amd/csrc/knobs.h); sglk_reload_env() makes it look again, here
    after every change and once more when the test is over, so no knob leaks into the next test."""
    from sgl_kernel import _lib

    def set_knobs(**kw):
        for k, v in kw.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        _lib.lib().sglk_reload_env()

    yield set_knobs
    monkeypatch.undo()
    _lib.lib().sglk_reload_env()
