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


def _build_provider(tmp_path_factory, name, *defines, source="gyroid_provider.c"):
    out = tmp_path_factory.mktemp(name) / f"lib{name}.so"
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
                           "-Wextra", "-Werror", *defines, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", source), "-o", str(out), "-lm"])
    return str(out)


_build_gyroid = _build_provider


@pytest.fixture(scope="session")
def failing_provider(tmp_path_factory):
    """tests/c/failing_provider.c: a provider whose calls fail the ways wasm/native.rs guards against (NULL results, unknown tags,
    ragged lengths)."""
    return _build_provider(tmp_path_factory, "failing_provider", source="failing_provider.c")


@pytest.fixture(scope="session")
def gyroid_provider(tmp_path_factory):
    """tests/c/gyroid_provider.c built as a shared object: a HOST-ONLY SDF behind include/sdf_provider.h's per-point ABI."""
    return _build_gyroid(tmp_path_factory, "gyroid_provider")


@pytest.fixture(scope="session")
def gyroid_provider_batch(tmp_path_factory):
    """... the same SDF from a library that ALSO exports the optional `sample_batch` (batched sampling, sdf_provider.h)."""
    return _build_gyroid(tmp_path_factory, "gyroid_provider_batch", "-DGYROID_BATCH")
