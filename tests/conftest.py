import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# measured parity errors of the -m gpu run (tests call record_error; written to gpurun_out/parity_errors.json at session end so the
# tolerances in tests/test_gpu_parity.py can be audited against what was actually measured on the MI355X)
PARITY_ERRORS = {}


def record_error(test: str, what: str, err: float, scale: float, tol: float):
    e = PARITY_ERRORS.setdefault(test, {"worst_rel": 0.0, "worst": None, "n": 0})
    rel = err / max(scale, 1e-30)
    e["n"] += 1
    if rel >= e["worst_rel"]:
        e["worst_rel"], e["worst"] = rel, {"what": what, "err": err, "scale": scale, "tol": tol}


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_ERRORS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(PARITY_ERRORS, f, indent=1, sort_keys=True)
    except OSError:
        pass
