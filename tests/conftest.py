"""pytest configuration: markers, import paths, shared helpers."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "torch-interpol_amd")
for p in (PKG_ROOT, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE_DIR = "/root/reference/interpol"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def load_reference():
    """Import the reference package under the alias `interpol_ref` (it never
    shadows the product package `interpol`).  Build container only."""
    if "interpol_ref" in sys.modules:
        return sys.modules["interpol_ref"]
    if not os.path.isdir(REFERENCE_DIR):
        pytest.skip("reference not available on this machine")
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(
        "interpol_ref", os.path.join(REFERENCE_DIR, "__init__.py"),
        submodule_search_locations=[REFERENCE_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["interpol_ref"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def reference():
    return load_reference()
