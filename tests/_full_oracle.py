"""Full-size oracle runs for the GPU suite (TEST INFRASTRUCTURE).  The oracle (oracle/: the C restatement of the
reference) needs ~110 s for the suffix array of 1 GB and 80-95 s for the quadratic LCP array, on ONE core each -- the
GPU box has hundreds.  start() launches one thread per full-size config at the beginning of a `-m gpu` session (ctypes
releases the GIL for the whole call), so that the complete arrays are ready by the time the full-size tests reach them
and the suite pays ~3.5 minutes once instead of ~3.5 minutes per config."""
import threading
import time

import numpy as np

_jobs = {}
_lock = threading.Lock()

# config key -> (generator name in tests/_gen.py, LCP routine of the oracle)
CONFIGS = {
    "c3": ("english_like", "lcp_quadratic"),       # the reference's own routine (src/table.rs:348-361)
    "c5": ("utf8_mixed", "lcp_quadratic"),
    # mean LCP 275: the quadratic routine would need half an hour; Kasai's array is the same array (test_oracle.py checks
    # the two routines against each other on every parity text)
    "dup": ("near_duplicates", "lcp_kasai"),
    # configs 3 / 5 on ROUND 1's inputs (tests/_gen_r1.py: the numpy generators, 17 + 27 s): their pins dated from a round-1 run
    # until round 5 put the complete comparison into every -m gpu session as well
    "c3r1": ("_gen_r1:english_like", "lcp_quadratic"),
    "c5r1": ("_gen_r1:utf8_mixed", "lcp_quadratic"),
}


def _run(key, n, out):
    import _gen
    import oracle
    gen, lcp_fn = CONFIGS[key]
    try:
        t0 = time.time()
        if ":" in gen:
            import importlib
            mod, fn = gen.split(":")
            host = getattr(importlib.import_module(mod), fn)(n)
        else:
            host = getattr(_gen, gen)(n)
        out["text"] = host
        sa = oracle.sais(host)
        out["sa_seconds"] = round(time.time() - t0, 1)
        out["sa"] = sa
        t1 = time.time()
        out["lcp"] = getattr(oracle, lcp_fn)(host, sa)
        out["lcp_seconds"] = round(time.time() - t1, 1)
        out["lcp_routine"] = lcp_fn
    except BaseException as e:                       # surfaced by get()
        out["error"] = repr(e)


def start(n=1_000_000_000, keys=("c3", "c5", "dup", "c3r1", "c5r1")):
    import oracle
    oracle.build()
    with _lock:
        for key in keys:
            if (key, n) in _jobs:
                continue
            out = {}
            th = threading.Thread(target=_run, args=(key, n, out), name=f"full-oracle-{key}", daemon=True)
            th.start()
            _jobs[(key, n)] = (th, out)


def get(key, n=1_000_000_000, timeout=1500):
    """Complete oracle arrays of a config: dict(text, sa, lcp, sa_seconds, lcp_seconds, lcp_routine)."""
    start(n, (key,))
    th, out = _jobs[(key, n)]
    th.join(timeout)
    if th.is_alive():
        raise TimeoutError(f"oracle run of {key} still going after {timeout} s")
    if "error" in out:
        raise RuntimeError(out["error"])
    return out


def release(key, n=1_000_000_000):
    with _lock:
        _jobs.pop((key, n), None)
