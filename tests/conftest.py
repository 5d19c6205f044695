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
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


# ---- math modes ------------------------------------------------------------------------
# The device's default mode is OUTFIT_MATH_GLIBC; the oracle's default is correctly rounded (its KAT tests choose
# what they need). Every -m gpu test compares the two, so for those the oracle's restated-glibc switch follows the
# device's process-wide default; tests that use the `device_math_mode` fixture run once per mode.
def pytest_generate_tests(metafunc):
    if "device_math_mode" in metafunc.fixturenames:
        metafunc.parametrize("device_math_mode", ["glibc", "cr"], indirect=True)


@pytest.fixture(autouse=True)
def _oracle_follows_the_device_math_mode(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import oracle_lib as O
    import outfit_amd
    with O.math_flags(libm=0, glibc=1 if outfit_amd.get_math_mode() == outfit_amd.MATH_GLIBC else 0):
        yield


@pytest.fixture
def device_math_mode(request):
    """Sets the device's process-wide default mode and the oracle's matching switch for one test."""
    import oracle_lib as O
    import outfit_amd
    mode = outfit_amd.MATH_GLIBC if request.param == "glibc" else outfit_amd.MATH_CORRECTLY_ROUNDED
    with outfit_amd.math_mode(mode), O.math_flags(libm=0, glibc=1 if request.param == "glibc" else 0):
        yield request.param
