"""Shared pytest configuration.

Markers
-------
``gpu``  tests that need a real MI355X (run by the driver with ``-m gpu``).
Everything else runs on CPU (``-m "not gpu"``) in a few minutes.

Roles
-----
* ``oracle``            -- CPU restatement + compiled reference (the checker).
* ``pico_tree_amd``     -- the product; GPU tests call it through the C ABI.
* ``tests/cpp/libptk_emu.so`` -- the product's kernel source compiled for the host
  (lane-by-lane emulation) so the CPU tier can check kernel logic.
"""

from __future__ import annotations

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X GPU")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Compile libptk.so, the oracle and the emulator once per session."""
    import __graft_entry__

    __graft_entry__.build()


def have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not have_gpu():
        pytest.skip("no GPU visible")
    return 0


@pytest.fixture(autouse=True)
def _no_test_knobs_left_behind():
    """The test hooks of the library (pico_tree_amd.set_test_knobs -> PTK_TEST_KNOBS) never outlive the test that set them."""
    yield
    os.environ.pop("PTK_TEST_KNOBS", None)
