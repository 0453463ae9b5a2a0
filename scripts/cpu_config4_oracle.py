"""BASELINE config 4 pinned to the ORACLE ("u32-oracle cross-check since n < 2^32").

CPU only -- needs about 24 GB of memory and tens of minutes on one core; no GPU, no /root/reference.  Generates the config-4
text (4 * 10^9 bytes uniform DNA, splitmix64 seed 0x5AF1C5 + 4: tests/_gen.dna_fast), runs oracle.sais (the C restatement of
/root/reference/src/table.rs:388-574; u32 positions fit since n < 2^32, :380) on the whole text once, and records

    sha256 of the text, sha256 of the complete suffix array (little-endian u32),
    sha256 of every 2^28-entry chunk of it (to localise a mismatch), the first / last entries, the run time

into tests/golden/fullsize_pins.json under "c4".  tests/_config4.rehearse compares the GPU array with these pins.

    python scripts/cpu_config4_oracle.py [n]        # default n = 4 000 000 000
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHUNK = 1 << 28


def main():
    import _gen
    import oracle
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000_000
    seed = 0x5AF1C5 + 4
    t0 = time.time()
    text = _gen.dna_fast(n, seed=seed)
    sha_text = hashlib.sha256(memoryview(text)).hexdigest()
    print("text", n, sha_text, round(time.time() - t0, 1), flush=True)
    t0 = time.time()
    sa = oracle.sais(text)
    sais_s = time.time() - t0
    print("oracle.sais seconds", round(sais_s, 1), flush=True)
    assert sa.dtype == np.uint32 and sa.size == n
    h = hashlib.sha256()
    chunks = []
    for lo in range(0, n, CHUNK):
        mv = memoryview(sa[lo:lo + CHUNK])
        h.update(mv)
        chunks.append(hashlib.sha256(mv).hexdigest())
    rec = {
        "sha256_text": sha_text,
        "sha256_sa": h.hexdigest(),
        "sha256_sa_chunks_2p28": chunks,
        "sa_first": [int(x) for x in sa[:6]],
        "sa_last": [int(x) for x in sa[-4:]],
        "oracle_sais_seconds": round(sais_s, 1),
        "source": "scripts/cpu_config4_oracle.py: oracle.sais (C restatement of src/table.rs:388-574) over the complete "
                  "text on one CPU core, u32 positions (n < 2^32, src/table.rs:380); seed 0x5AF1C5+4, tests/_gen.dna_fast",
    }
    print(json.dumps(rec), flush=True)
    pins_path = os.path.join(ROOT, "tests", "golden", "fullsize_pins.json")
    pins = json.load(open(pins_path))
    pins.setdefault("c4", {})[str(n)] = rec
    json.dump(pins, open(pins_path, "w"), indent=1)
    print("pinned into", pins_path, flush=True)


if __name__ == "__main__":
    main()
