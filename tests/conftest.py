import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with gzip.open(os.path.join(ROOT, "tests", "golden", "golden.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def streams():
    """name -> np.uint8 stream, generated once per session (tests/synth.py)."""
    import synth

    makers = {
        "modes1": lambda: synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")),
        "uniform": synth.case_uniform,
        "coarse": synth.case_coarse,
        "edges": synth.case_edges,
        "edges_smear": lambda: synth.case_edges(seed=23, smear16=6),
        "frames": synth.case_frames,
        "smear": synth.case_smear,
        "lowsnr": synth.case_lowsnr,
        "noise": synth.case_noise,
        "saturated": synth.case_saturated,
    }
    cache = {}

    class Lazy(dict):
        def __missing__(self, k):
            cache[k] = makers[k]()
            self[k] = cache[k]
            return cache[k]

        def names(self):
            return list(makers)

    return Lazy()
