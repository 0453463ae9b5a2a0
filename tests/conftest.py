import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_finish(session):
    """A GPU session that will reach the full-oracle tests starts the 1 GB oracle runs now, in background threads:
    they take ~3.5 minutes on one core each and are ready when tests/test_gpu_fullsize.py gets to them."""
    if any(item.name.endswith("_full_oracle") for item in session.items):
        try:
            import torch
            if not torch.cuda.is_available():
                return
        except ImportError:
            return
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import _full_oracle
        _full_oracle.start(keys=tuple(sorted({item.name.split("_")[1] for item in session.items
                                              if item.name.endswith("_full_oracle")} & set(_full_oracle.CONFIGS))))


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fasta():
    z = np.load(os.path.join(GOLDEN_DIR, "fasta_fixtures.npz"))
    return {k: z[k].tobytes() for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc
