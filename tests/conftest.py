"""pytest configuration: marker registration + import paths.

The product modules are flat (reference style: ``PYTHONPATH=source``), living in
``navierstokes-with-fenics_amd/``; the oracle is test infrastructure under ``oracle/``.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "navierstokes-with-fenics_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _run_in_tmp_dir(tmp_path, monkeypatch):
    """problem drivers write results/ under the current directory (as the reference does)"""
    monkeypatch.chdir(tmp_path)
