import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- the library reads its RFLU_* tuning variables once per handle (rflu_create); tests switch them between calls through
# monkeypatch: let every setenv / delenv / undo make the default handles read the environment again (rflu_reload_tuning)
def _reload_library_tuning():
    mod = sys.modules.get("recursivefactorization.jl_amd._ffi")
    if mod is not None and getattr(mod, "_lib", None) is not None:
        try:
            mod.reload_tuning()
        except Exception:
            pass


def _wrap_monkeypatch(name):
    orig = getattr(pytest.MonkeyPatch, name)

    def wrapped(self, *a, **k):
        r = orig(self, *a, **k)
        _reload_library_tuning()
        return r

    setattr(pytest.MonkeyPatch, name, wrapped)


for _n in ("setenv", "delenv", "undo"):
    _wrap_monkeypatch(_n)
