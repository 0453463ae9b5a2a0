"""world_size 2 and 3 CPU tests of the range-partitioned (multi-GPU) construction:
torch.distributed over gloo, compute through the emulator build of the product
kernels (tensors live in host memory).  The slices of the two ranks must
concatenate to the oracle's suffix array."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
import numpy as np, torch, torch.distributed as dist
import suffix_amd, _gen
from suffix_amd import dist as sdist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
eng = suffix_amd.Engine(os.path.join({here!r}, "emu", "libsuffix_emu.so"))
kind = os.environ["SFX_CASE"]
m = {{"periodic": 1200, "unary": 600}}.get(kind, 6000)
if kind == "ragged":                                       # shards of different lengths (and not multiples of a word)
    m = [5000, 6001, 4377][rank]
if kind == "tiny":                                         # a shard below 64 bytes: the text travels as raw bytes
    m = [700, 33, 912][rank]
if kind == "dna":
    full = _gen.dna(m * world, seed=99)
elif kind == "text":
    full = _gen.english_like(m * world, seed=7)
elif kind == "unary":
    full = np.frombuffer(b"a" * (m * world), dtype=np.uint8)      # one key bin: every rank but one gets an empty slice
elif kind == "ragged":
    full = _gen.dna(5000 + 6001 + 4377, seed=98)
elif kind == "tiny":
    full = _gen.dna(700 + 33 + 912, seed=97)
else:
    full = np.frombuffer((b"ab" * (m * world // 2)), dtype=np.uint8)
if kind in ("ragged", "tiny"):
    lo = sum(([5000, 6001, 4377] if kind == "ragged" else [700, 33, 912])[:rank])
    shard = torch.from_numpy(np.ascontiguousarray(full[lo:lo + m]).copy())
else:
    shard = torch.from_numpy(np.ascontiguousarray(full[rank * m:(rank + 1) * m]).copy())
timings = {{}}
part, offset, n, text = sdist.build_sa_partitioned(shard, engine=eng, top_bits=10, return_text=True, timings=timings)
assert "range_build" in timings and "key_hist" in timings, timings
# every shard has >= 64 bytes: the text travels as packed symbol codes, ragged shards included
if kind == "tiny":
    assert timings["info"]["text_exchange"] == "raw bytes", timings
else:
    assert timings["info"]["text_exchange"].startswith("packed words") and (kind != "ragged" or "ragged" in timings["info"]["text_exchange"]), timings
if kind in ("periodic", "unary"):                          # repeats longer than text refinement can settle inside a slice
    assert "fallback" in timings["info"], timings
else:
    assert "fallback" not in timings["info"], timings
    assert all(isinstance(v, (int, float)) for k, v in timings.items() if k != "info"), timings      # phases are numbers
np.save(os.path.join(os.environ["SFX_OUT"], f"part{{rank}}.npy"), part.numpy().view(np.uint32))
np.save(os.path.join(os.environ["SFX_OUT"], f"off{{rank}}.npy"), np.array([offset, n]))
# the partitioned index in use: per-slice LCP (one suffix index exchanged per rank) and
# queries answered by every rank on its slice + one all-reduce
lcp = sdist.build_lcp_partitioned(text, part, engine=eng)
np.save(os.path.join(os.environ["SFX_OUT"], f"lcp{{rank}}.npy"), lcp.numpy().view(np.uint32))
fb = full.tobytes()
qm = 4000 if kind == "ragged" else (600 if kind == "tiny" else m)
qs = [fb[100:106], fb[-7:], b"zzzz", fb[qm // 2:qm // 2 + 2], fb[qm - 2:qm + 3], fb[5:6]]
qb = torch.from_numpy(np.frombuffer(b"".join(qs), dtype=np.uint8).copy())
qoff = torch.tensor(np.concatenate([[0], np.cumsum([len(q) for q in qs])]), dtype=torch.int64)
gs, ge = sdist.positions_partitioned(text, part, offset, qb, qoff, engine=eng)
np.save(os.path.join(os.environ["SFX_OUT"], f"q{{rank}}.npy"), np.stack([gs.numpy(), ge.numpy()]))
# u64 indices (BASELINE config 4)
p64, o64, _n = sdist.build_sa_partitioned(shard, engine=eng, top_bits=10, index_dtype=torch.int64)
assert p64.dtype == torch.int64 and o64 == offset and np.array_equal(p64.numpy().astype(np.uint32), part.numpy().view(np.uint32))
ok, how = sdist.verify_partitioned(shard, part, n, engine=eng)
assert ok, how
bad = part.clone(); 
if bad.numel() > 1:
    bad[0], bad[1] = part[1].clone(), part[0].clone()          # one swapped pair must be caught (on every rank)
ok2, how2 = sdist.verify_partitioned(shard, bad, n, engine=eng)
assert not ok2, how2
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.parametrize("case,world", [("dna", 2), ("text", 2), ("periodic", 2), ("unary", 2), ("dna", 3), ("ragged", 3), ("tiny", 3)])
def test_partitioned_build_ranks(tmp_path, oracle, case, world):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, here=HERE))
    env = dict(os.environ, SFX_CASE=case, SFX_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                           str(script)], env=env, timeout=600)
    sys.path.insert(0, HERE)
    import _gen
    m = {"periodic": 1200, "unary": 600}.get(case, 6000)
    if case == "dna":
        full = _gen.dna(m * world, seed=99)
    elif case == "text":
        full = _gen.english_like(m * world, seed=7)
    elif case == "unary":
        full = np.frombuffer(b"a" * (m * world), dtype=np.uint8)
    elif case == "ragged":
        full = _gen.dna(5000 + 6001 + 4377, seed=98)
    elif case == "tiny":
        full = _gen.dna(700 + 33 + 912, seed=97)
    else:
        full = np.frombuffer((b"ab" * (m * world // 2)), dtype=np.uint8)
    exp = oracle.sais(full.tobytes())
    parts = [np.load(tmp_path / f"part{r}.npy") for r in range(world)]
    offs = [np.load(tmp_path / f"off{r}.npy") for r in range(world)]
    assert int(offs[0][0]) == 0
    for r in range(1, world):
        assert int(offs[r][0]) == sum(p.size for p in parts[:r])
    assert int(offs[0][1]) == full.size
    assert np.array_equal(np.concatenate(parts), exp)
    text = full.tobytes()
    lcps = [np.load(tmp_path / f"lcp{r}.npy") for r in range(world)]
    assert np.array_equal(np.concatenate(lcps), oracle.lcp_quadratic(text, exp))
    qm = 4000 if case == "ragged" else (600 if case == "tiny" else m)
    qs = [text[100:106], text[-7:], b"zzzz", text[qm // 2:qm // 2 + 2], text[qm - 2:qm + 3], text[5:6]]
    for r in range(world):                                   # every rank holds the same global answer
        q = np.load(tmp_path / f"q{r}.npy")
        for k, query in enumerate(qs):
            es, ee = oracle.positions(text, exp, query)
            assert (int(q[0][k]), int(q[1][k])) == (es, ee), (case, r, query)
