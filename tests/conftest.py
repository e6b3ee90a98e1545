import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ray-tracing-practice_amd")
sys.path.insert(0, ROOT)
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # a GPU test that hangs must not hold the box until the outer limit: hard per-test limit
    # (thread method: the process is ended even while it sits inside a HIP call)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(pytest.mark.timeout(240, method="thread"))
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def run_child(cmd, timeout, **kw):
    """A child process that uses the GPU, ended WITH ITS WHOLE GROUP when it overruns: the per-test limit above ends this
    process with os._exit, and children left behind would keep holding the card.  `timeout` stays below that limit."""
    import signal
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        raise AssertionError(f"{cmd[0]} … did not finish within {timeout} s\n{out[-2000:]}\n{err[-2000:]}")
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    """Make sure the in-tree libraries exist (they are git-ignored build products)."""
    needed = [os.path.join(PKG, "librtp_amd.so"), os.path.join(PKG, "librtp_amd_dev.so"), os.path.join(PKG, "librtp_host.so"),
              os.path.join(ROOT, "oracle", "librt_oracle.so")]
    if not all(os.path.exists(p) for p in needed):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(scope="session")
def test_config_text():
    with open(os.path.join(ROOT, "tests", "golden", "test_config.txt")) as f:
        return f.read()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "survey_pins.json")) as f:
        return json.load(f)
