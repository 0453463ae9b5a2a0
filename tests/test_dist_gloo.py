"""world_size-2 CPU test of the range-partitioned (multi-GPU) construction:
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
m = 6000
if kind == "dna":
    full = _gen.dna(m * world, seed=99)
elif kind == "text":
    full = _gen.english_like(m * world, seed=7)
else:
    full = np.frombuffer((b"ab" * (m * world // 2)), dtype=np.uint8)
shard = torch.from_numpy(np.ascontiguousarray(full[rank * m:(rank + 1) * m]).copy())
part, offset, n = sdist.build_sa_partitioned(shard, engine=eng, top_bits=10)
np.save(os.path.join(os.environ["SFX_OUT"], f"part{{rank}}.npy"), part.numpy().view(np.uint32))
np.save(os.path.join(os.environ["SFX_OUT"], f"off{{rank}}.npy"), np.array([offset, n]))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("case", ["dna", "text", "periodic"])
def test_partitioned_build_two_ranks(tmp_path, oracle, case):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(HERE, "emu")])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, here=HERE))
    env = dict(os.environ, SFX_CASE=case, SFX_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                           str(script)], env=env, timeout=600)
    sys.path.insert(0, HERE)
    import _gen
    m, world = 6000, 2
    if case == "dna":
        full = _gen.dna(m * world, seed=99)
    elif case == "text":
        full = _gen.english_like(m * world, seed=7)
    else:
        full = np.frombuffer((b"ab" * (m * world // 2)), dtype=np.uint8)
    exp = oracle.sais(full.tobytes())
    parts = [np.load(tmp_path / f"part{r}.npy") for r in range(world)]
    offs = [np.load(tmp_path / f"off{r}.npy") for r in range(world)]
    assert int(offs[0][0]) == 0 and int(offs[1][0]) == parts[0].size
    assert int(offs[0][1]) == m * world
    assert np.array_equal(np.concatenate(parts), exp)
