#!/usr/bin/env python3
"""Regenerate tests/golden/*.  Run in the build container only (it reads the two
FASTA fixtures from /root/reference/tests/, which does not exist on the GPU box).

Outputs
  fasta_fixtures.npz   the reference's two bench inputs
                       (/root/reference/tests/AP009048_10000.fasta, _100000.fasta;
                       used by tests/bench.rs:27-60) stored as uint8 arrays, so that
                       BASELINE config 1 can run on the GPU box.
  golden.json          expected values:
                         * sha256 of SA / LCP (little-endian u32) of both fixtures,
                           computed HERE from the *definition* (sorted byte suffixes,
                           direct LCP) -- independent of any SA-IS code -- and equal to
                           the values recorded in SURVEY.md section 8c;
                         * every known-answer literal of the reference's tests
                           (tests/tests.rs:22-70, :100-168, :181-213), its doc-tests
                           (src/lib.rs:16-23, src/table.rs:191-196, :217-222, :272-278)
                           and examples (examples/basic.rs:7).
"""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests"


def definitional(text: bytes):
    n = len(text)
    sa = sorted(range(n), key=lambda i: text[i:])
    lcp = [0] * n
    for r in range(1, n):
        a, b, k = sa[r - 1], sa[r], 0
        while a + k < n and b + k < n and text[a + k] == text[b + k]:
            k += 1
        lcp[r] = k
    return np.array(sa, dtype="<u4"), np.array(lcp, dtype="<u4")


def main():
    fixtures = {}
    golden = {"fixtures": {}, "sa_literals": {}, "search": []}
    for name in ("AP009048_10000", "AP009048_100000"):
        raw = open(os.path.join(REF, name + ".fasta"), "rb").read()
        fixtures[name] = np.frombuffer(raw, dtype=np.uint8)
        sa, lcp = definitional(raw)
        golden["fixtures"][name] = {
            "len": len(raw),
            "sha256_text": hashlib.sha256(raw).hexdigest(),
            "sha256_sa": hashlib.sha256(sa.tobytes()).hexdigest(),
            "sha256_lcp": hashlib.sha256(lcp.tobytes()).hexdigest(),
            "sa_head": sa[:6].tolist(), "sa_tail": sa[-4:].tolist(),
            "max_lcp": int(lcp.max()),
        }
    np.savez_compressed(os.path.join(HERE, "fasta_fixtures.npz"), **fixtures)

    # tests/tests.rs:22-70 -- new(x) == new_naive(x); expected = the definition
    for s in ["apple", "banana", "mississippi", "tgtgtgtgcaccg", "", "a", "ab", "aa",
              "\x00", "☃abc☃", "poëzie"]:
        sa, lcp = definitional(s.encode("utf-8"))
        golden["sa_literals"][s] = {"sa": sa.tolist(), "lcp": lcp.tolist()}

    # search known answers: (text, query, positions in SA order, contains)
    S = golden["search"]
    S += [["", "", [], False], ["", "a", [], False], ["", "ab", [], False],      # :100-119
          ["a", "", [], False], ["a", "b", [], False], ["a", "a", [0], True],    # :121-140
          ["ab", "b", [1], True], ["aa", "a", [1, 0], True],                     # :142-154
          ["zzzzzaazzzzz", "a", [5, 6], True],                                   # :156-161
          ["zzzzabczzzzzabczzzzzz", "abc", [4, 12], True],                       # :163-168
          ["az", "mnomnomnomnomnomnomno", [], False],                            # :181-186
          ["zz", "mnomnomnomnomnomnomno", [], False],                            # :188-193
          ["aa", "mnomnomnomnomnomnomno", [], False],                            # :195-200
          ["The quick brown fox was very quick.", "quick", [4, 29], True],       # :202-206
          ["☃abc☃", "☃", [6, 0], True],                           # :208-213
          ["the quick brown fox was quick.", "quick", [4, 24], True],            # src/lib.rs:19
          ["the quick brown fox was quick.", "faux", [], False],                 # src/lib.rs:23
          ["The quick brown fox.", "quick", [4], True]]                          # table.rs:194
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1, ensure_ascii=True, sort_keys=True)
    print(json.dumps(golden["fixtures"], indent=1))


if __name__ == "__main__":
    main()
