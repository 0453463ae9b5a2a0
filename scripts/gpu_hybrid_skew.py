#!/usr/bin/env python3
"""Hybrid initial sort on skewed 4-letter texts (100 MB): route taken and time, hybrid on / off (development)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, suffix_amd
    from suffix_amd import device as sdev
    import _devlib
    eng = _devlib.engine(); eng.require_device()
    n = 100_000_000
    rng = np.random.default_rng(3)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    for name, p in (("uniform", [0.25] * 4), ("mild 32/18", [0.32, 0.18, 0.18, 0.32]), ("medium 36/14", [0.36, 0.14, 0.14, 0.36]),
                    ("strong 42/8", [0.42, 0.08, 0.08, 0.42])):
        text = torch.from_numpy(letters[rng.choice(4, size=n, p=p)]).cuda()
        ws = sdev.sa_workspace(n, text.device); sa = torch.empty(n, dtype=torch.int32, device="cuda")
        sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): sdev.build_sa(text, out=sa, workspace=ws)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        eng.profile(True); eng.profile_reset(); sdev.build_sa(text, out=sa, workspace=ws); torch.cuda.synchronize()
        k = {r["name"]: round(r["total_ms"], 3) for r in eng.profile_report()}; eng.profile(False)
        print(json.dumps({"text": name, "hybrid": os.environ.get("SFX_HYBRID", "1"), "ms": round(ms, 3), "lds": k.get("bucket_sort_ties", k.get("bucket_sort_lds")),
                          "oversize": [k.get("oversize_gather"), k.get("oversize_return")], "radix_scatter_u32": k.get("radix_scatter_u32"),
                          "active_after_initial": eng.build_stats()["active_after_initial"]}), flush=True)
        del text, ws, sa
else:
    for h in ("1", "0"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SFX_HYBRID=h))
