import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name + ".npz")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(golden_path(name), allow_pickle=False))
        return cache[name]
    return load


@pytest.fixture
def seld_env(monkeypatch):
    """Set / unset SELD_* switches of libseld_hip.so for one test.  The library reads its environment once, so every
    change is followed by `seld_env_reload`; the original environment is restored (and re-read) afterwards."""
    import importlib
    L = importlib.import_module("sound-event-localization-and-detection_amd._lib")

    class Env:
        def set(self, name, value):
            monkeypatch.setenv(name, value)
            L.reload_env()

        def unset(self, name):
            monkeypatch.delenv(name, raising=False)
            L.reload_env()
    yield Env()
    monkeypatch.undo()
    L.reload_env()
