import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and they were not deselected."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# A line per finished test in gpurun_out/pytest_progress.log (flushed; created on demand, ignored when the directory cannot be written):
# `-q` prints one unflushed-looking dot per test, and a GPU box's watchdog takes a run that writes nothing for seven minutes to be hung - the
# multi-rank bench tests alone run for a minute each.  The file also says which test a killed run was in.
_progress = {"fh": None, "t0": None}


def pytest_runtest_logstart(nodeid, location):
    import time

    _progress["t0"] = time.time()
    try:
        if _progress["fh"] is None:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            _progress["fh"] = open(os.path.join(d, "pytest_progress.log"), "a")
        _progress["fh"].write(f"start {nodeid}\n")
        _progress["fh"].flush()
    except OSError:
        _progress["fh"] = False


def pytest_runtest_logfinish(nodeid, location):
    import time

    fh = _progress["fh"]
    if fh:
        try:
            fh.write(f"done  {nodeid}  {time.time() - (_progress['t0'] or time.time()):.1f} s\n")
            fh.flush()
        except OSError:
            pass
