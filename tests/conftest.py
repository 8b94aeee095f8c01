import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Build liboracle.so / libsdfgrid.so when missing (they are git-ignored; on the GPU box the prebuilt
    files travel with the snapshot)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "sdf-viewer_amd", "libsdfgrid.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sdf-viewer_amd", "csrc")], stdout=subprocess.DEVNULL)
    if not (os.path.exists(os.path.join(ROOT, "sdf-viewer_amd", "libsdfviewer_host_test.so")) and
            os.path.exists(os.path.join(ROOT, "sdf-viewer_amd", "libsdfdemo_provider.so"))):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "sdf-viewer_amd", "host")], stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("sdf-viewer_amd")


@pytest.fixture(scope="session")
def host():
    import host_binding
    return host_binding


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding


@pytest.fixture(scope="session")
def gyroid_provider(tmp_path_factory):
    """tests/c/gyroid_provider.c built as a shared object: a HOST-ONLY SDF behind include/sdf_provider.h's per-point ABI."""
    out = tmp_path_factory.mktemp("gyroid") / "libgyroid_provider.so"
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
                           "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "gyroid_provider.c"), "-o", str(out), "-lm"])
    return str(out)
