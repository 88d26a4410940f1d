import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def record_parity(test: str, **fields) -> None:
    """Append one line to gpurun_out/parity_records.jsonl (kept by the suite: the mismatch counts of the argmax-id checks are
    recorded, not only printed).  The round's records are copied to profiles/."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_records.jsonl"), "a") as fh:
            fh.write(json.dumps(dict(test=test, **fields)) + "\n")
    except OSError:
        pass
