import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def emu_mod():
    from emu import emu
    emu.lib()
    return emu


def make_holder(desc_dict):
    from micro_raytracer_amd import _abi, scene
    r = scene.load_render(desc_dict)
    return r, _abi.build_desc(r)
