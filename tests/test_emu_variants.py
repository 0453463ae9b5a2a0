"""Non-default code paths of the engine, on the CPU emulator: the tuning / test-hook
environment variables are read once per process, so every variant runs in a subprocess.
  SFX_RADIX_SWEEP / _NW / _KPT / _RANK   radix schedules, tile geometries, ranking methods
  SFX_MAX_GRID                           multi-tile chunks per workgroup on small inputs
  SFX_PARTITION_MIN                      partitioned (cache-confined) rank / Phi scatters
Every run compares SA and LCP with the oracle on a few texts that exercise the path."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np
import _gen, oracle
from suffix_amd import Engine, SuffixTable
oracle.build()
eng = Engine(os.path.join({here!r}, "emu", "libsuffix_emu.so"))
texts = [_gen.dna(40000, seed=9).tobytes(),                      # E64 path, several tiles
         _gen.english_like(15000, seed=3).tobytes(),             # 64-bit keys, rank rounds
         b"AAAAAAAAAAAAAAAAAAAAAAAC" * 600,                      # long repeats: many rounds
         _gen.dna(9000, seed=3).tobytes() + b"A" * 3000]
for t in texts:
    st = SuffixTable(t, engine=eng)
    exp = oracle.sais(t)
    assert np.array_equal(st.table(), exp), ("SA", len(t))
    assert np.array_equal(st.lcp_lens(), oracle.lcp_quadratic(t, exp)), ("LCP", len(t))
print("OK")
"""

VARIANTS = {
    "chunked-multi-tile": {"SFX_RADIX_SWEEP": "0", "SFX_MAX_GRID": "2"},
    "one-sweep-4-waves-kpt16-ballot": {"SFX_RADIX_NW": "4", "SFX_RADIX_KPT": "16", "SFX_RADIX_RANK": "0"},
    "one-sweep-8-waves-kpt8": {"SFX_RADIX_NW": "8", "SFX_RADIX_KPT": "8", "SFX_MAX_GRID": "3"},
    "partitioned-scatter": {"SFX_PARTITION_MIN": "1"},
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_engine_variant_on_emulator(tmp_path, name):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "variant.py"
    script.write_text(SCRIPT.format(root=ROOT, here=HERE))
    env = dict(os.environ, **VARIANTS[name])
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
