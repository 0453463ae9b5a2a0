"""Reads bench.py's stdout: DETAIL lines (full records) + the final compact JSON line.
    final, details = load(sys.stdin)      # details: {"headline": {...}, "c3": {...}, ...}
`legacy(final, details)` rebuilds the single big record the round-3 scripts were written against."""
import json


def load(stream):
    final, details = None, {}
    for line in stream:
        line = line.strip()
        if line.startswith("DETAIL "):
            rec = json.loads(line[7:])
            details[rec.pop("detail")] = rec
        elif line.startswith("{"):
            final = json.loads(line)
    return final, details


def legacy(final, details):
    out = dict(final)
    head = details.get("headline", {})
    for k in ("roofline", "cpu_baseline", "lcp", "verification"):
        if k in head:
            out[k] = head[k]
    out["config"] = head.get("config", out.get("config"))
    out["configs"] = [v for k, v in details.items() if k != "headline"]
    return out
