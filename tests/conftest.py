import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_runtest_setup(item):
    """Every GPU parity test is checked against the REAL reference (oracle/_ref/libsqoa_ref.so: /root/reference/seqoia.h
    compiled by oracle/Makefile in the build container, travelling with the snapshot).  pyoracle.best() would fall back to
    the restatement if it had not travelled; that must be a failure, not luck."""
    if item.get_closest_marker("gpu") is not None:
        from oracle import pyoracle as po

        assert po.best().kind == "reference", "oracle/_ref/libsqoa_ref.so is missing: GPU parity must run against the compiled reference"


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat_encode():
    return load_golden("kat_encode.json")


@pytest.fixture(scope="session")
def kat_decode():
    return load_golden("kat_decode.json")


@pytest.fixture(scope="session")
def synth_digests():
    return load_golden("synth_digests.json")
