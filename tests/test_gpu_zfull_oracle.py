"""The COMPLETE suffix and LCP arrays of the 1 GB configs (BASELINE config 3, config 5, the high-LCP text, and configs 3 / 5 on
round 1's inputs) against
the oracle's, element by element, on a real MI355X (run with -m gpu).  The oracle needs ~110 s for a 1 GB suffix array
and 80-95 s for the quadratic LCP array on one core: tests/conftest.py starts the three runs in background threads when
the session is collected (tests/_full_oracle.py), this module sorts last, and the comparisons cost ~3.5 minutes together
instead of ~3.5 minutes each.  reference: src/table.rs:388-574 (sais), :348-361 (lcp_lens_quadratic)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = os.path.join(ROOT, "tests", "golden", "fullsize_pins.json")
N = 1_000_000_000


@pytest.fixture(scope="module")
def eng():
    import suffix_amd
    e = suffix_amd.default_engine()
    e.require_device()
    assert e.path.endswith("libsuffix_hip.so")
    return e


def _sha(t):
    return hashlib.sha256(memoryview(t.cpu().numpy())).hexdigest()


def _pin(key, n):
    with open(PINS) as f:
        return json.load(f)[key][str(n)]


def _full_oracle_compare(eng, key):
    """The COMPLETE suffix array and LCP array of a 1 GB config against the oracle's, element by element (the oracle
    ran in a background thread since the start of the session: tests/_full_oracle.py)."""
    import _full_oracle
    from suffix_amd import device as sdev
    ref = _full_oracle.get(key, N)
    host = ref["text"]
    pin = _pin(key, N)
    assert hashlib.sha256(memoryview(host)).hexdigest() == pin["sha256_text"]
    text = torch.from_numpy(host).to(torch.device("cuda", 0))
    sa, lcp = sdev.build_sa_lcp(text)                 # SuffixTable::new + lcp_lens as the engine's one call
    torch.cuda.synchronize()
    sa_h = sa.cpu().numpy().view(np.uint32)
    assert np.array_equal(sa_h, ref["sa"]), f"{key}: suffix array differs from the oracle's"
    del sa_h
    lcp_h = lcp.cpu().numpy().view(np.uint32)
    assert np.array_equal(lcp_h, ref["lcp"]), f"{key}: LCP array differs from the oracle's ({ref['lcp_routine']})"
    assert _sha(sa) == pin["sha256_sa"] and _sha(lcp) == pin["sha256_lcp"]       # (the pins are these arrays)
    print(f"full oracle {key}: sais {ref['sa_seconds']} s, {ref['lcp_routine']} {ref['lcp_seconds']} s, all 10^9 entries equal")
    del text, sa, lcp, lcp_h
    _full_oracle.release(key, N)
    torch.cuda.empty_cache()


def test_c3_full_oracle(eng):
    _full_oracle_compare(eng, "c3")


def test_c5_full_oracle(eng):
    _full_oracle_compare(eng, "c5")


def test_dup_full_oracle(eng):
    """The high-LCP config (mean LCP 275, max 2556): rank rounds, the Phi / PLCP route of lcp_lens."""
    _full_oracle_compare(eng, "dup")




def test_c3r1_full_oracle(eng):
    """Config 3 on round 1's input: the pin of bench.py's `c3r1` record, re-derived against the oracle in this run."""
    _full_oracle_compare(eng, "c3r1")


def test_c5r1_full_oracle(eng):
    """Config 5 on round 1's input (no refinement round at all: the direct pass finishes what the 64-bit keys leave tied)."""
    _full_oracle_compare(eng, "c5r1")
