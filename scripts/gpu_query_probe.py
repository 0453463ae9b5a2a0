#!/usr/bin/env python3
"""Where the time of a query batch goes (development): config 5's text, several query sets, the undirected search
against the resident index (B+tree of prefix keys / bucket directory).  Kernel times from the engine's profiler."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import _gen, suffix_amd
import _devlib
from suffix_amd import device as sdev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
eng = _devlib.engine(); eng.require_device()
dev = torch.device("cuda", 0)
host = _gen.utf8_mixed(n)
text = torch.from_numpy(host).to(dev)
sa = sdev.build_sa(text)
torch.cuda.synchronize()
t0 = time.perf_counter(); ix = sdev.DeviceIndex(text, sa); torch.cuda.synchronize()
print(json.dumps({"index_build_ms": round((time.perf_counter() - t0) * 1e3, 1)}), flush=True)
nq = 1_000_000
qb, off = _gen.queries(host, nq)
rng = np.random.default_rng(5)
sets = {"survey_8d": (qb, off)}
lens = off[1:] - off[:-1]
# the same queries cut to <= 8 bytes at a code-point boundary is fiddly: use plain byte prefixes of the text instead
st = rng.integers(0, n - 64, nq)
def substrings(length):
    idx = (st[:, None] + np.arange(length)[None, :]).reshape(-1)
    return host[idx].copy(), np.arange(0, (nq + 1) * length, length, dtype=np.int64)
sets["text_bytes_len6"] = substrings(6)
sets["text_bytes_len12"] = substrings(12)
sets["text_bytes_len32"] = substrings(32)
rb = rng.integers(0, 256, nq * 12, dtype=np.uint8)
sets["random_bytes_len12"] = (rb, np.arange(0, (nq + 1) * 12, 12, dtype=np.int64))
only = os.environ.get("PROBE_SETS")
if only:
    sets = {k: v for k, v in sets.items() if k in only.split(",")}
for name, (b, o) in sets.items():
    d_b, d_o = torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o).to(dev)
    rec = {"set": name}
    legs = (("undirected", lambda: sdev.query_batch(text, sa, d_b, d_o)), ("index", lambda: ix.query(d_b, d_o)))
    if os.environ.get("PROBE_INDEX_ONLY"):
        legs = legs[1:]
    for label, fn in legs:
        fn(); torch.cuda.synchronize()
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
        k = {x["name"]: round(x["total_ms"], 3) for x in eng.profile_report()}; eng.profile(False)
        rec[label] = {"wall_ms": round(wall * 1e3, 3), "kernel_ms": k, "hits": round(float(r[2].float().mean()), 3),
                      "mean_matches": round(float((r[1] - r[0]).float().mean()), 1)}
    print(json.dumps(rec), flush=True)
