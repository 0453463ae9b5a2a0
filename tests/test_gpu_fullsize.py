"""BASELINE.json configs 3 and 5 AT THEIR STATED SIZE (1 GB) on a real MI355X, plus the two
size-gated code paths the small parity cases cannot reach (run with -m gpu):

  * config 3 / config 5: SA, LCP (and the 10^6 positions() queries) through the C ABI, sha256 of the
    complete arrays compared with tests/golden/fullsize_pins.json; config 5's query answers are compared with
    the oracle here, all 10^6 of them;
  * (tests/test_gpu_zfull_oracle.py, last in the session: the COMPLETE 10^9-entry SA and LCP arrays of configs 3, 5
    and the high-LCP text against the oracle's, element by element)
  * n >= 2^27: rank rounds whose rank-array updates go through the partitioned scatter
    (scatter_pairs_u32), complete SA and LCP compared with the oracle;
  * n >= 2^30: the chunked radix schedule (one-sweep status words no longer fit), property gate.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401

import _gen

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


def _build(host):
    from suffix_amd import device as sdev
    dev = torch.device("cuda", 0)
    text = torch.from_numpy(host).to(dev)
    sa = sdev.build_sa(text)
    lcp = sdev.build_lcp(text, sa)
    torch.cuda.synchronize()
    return text, sa, lcp


def test_config3_1gb_english_sa_lcp(eng):
    host = _gen.english_like(N)
    pin = _pin("c3", N)
    assert hashlib.sha256(memoryview(host)).hexdigest() == pin["sha256_text"]
    text, sa, lcp = _build(host)
    st = eng.build_stats()
    assert st["n"] == N and st["tile_sorted"] > 0
    assert _sha(sa) == pin["sha256_sa"]
    assert _sha(lcp) == pin["sha256_lcp"]
    del sa, lcp
    # the same two arrays from the fused entry point (LCP read off the keys of the initial sort and the text rounds)
    from suffix_amd import device as sdev
    sa2, lcp2 = sdev.build_sa_lcp(text)
    torch.cuda.synchronize()
    assert _sha(sa2) == pin["sha256_sa"] and _sha(lcp2) == pin["sha256_lcp"]
    del text, sa2, lcp2
    torch.cuda.empty_cache()


def test_config5_1gb_utf8_sa_lcp_queries(eng, oracle):
    from suffix_amd import device as sdev
    host = _gen.utf8_mixed(N)
    pin = _pin("c5", N)
    assert hashlib.sha256(memoryview(host)).hexdigest() == pin["sha256_text"]
    text, sa, lcp = _build(host)
    assert _sha(sa) == pin["sha256_sa"]
    assert _sha(lcp) == pin["sha256_lcp"]
    del lcp
    qb, off = _gen.queries(host, 1_000_000)
    dev = text.device
    d_qb, d_off = torch.from_numpy(qb).to(dev), torch.from_numpy(off).to(dev)
    s0, e0, f0, a0 = sdev.query_batch(text, sa, d_qb, d_off)            # undirected binary search
    ix = sdev.DeviceIndex(text, sa)                                     # resident index with its bucket directory
    s, e, f, a = ix.query(d_qb, d_off)
    torch.cuda.synchronize()
    assert torch.equal(s, s0) and torch.equal(e, e0) and torch.equal(f, f0)
    assert hashlib.sha256(memoryview(torch.stack([s, e]).cpu().numpy())).hexdigest() == pin["sha256_start_end"]
    ix.close()
    sa_h = sa.cpu().numpy().view(np.uint32)
    es, ee = oracle.positions_batch(host, sa_h, qb, off)                 # every one of the 10^6 queries
    assert np.array_equal(s.cpu().numpy().view(np.uint32), es)
    assert np.array_equal(e.cpu().numpy().view(np.uint32), ee)
    found = f.cpu().numpy().astype(bool)
    assert np.array_equal(found, ee > es)
    hit = np.flatnonzero(found)[:2000]                                   # any_position: a real occurrence
    anyp = a.cpu().numpy().view(np.uint32)
    for k in hit.tolist():
        q = qb[off[k]:off[k + 1]]
        p = int(anyp[k])
        assert np.array_equal(host[p:p + q.size], q)
    assert 0.4 < found.mean() < 0.8
    del text, sa
    torch.cuda.empty_cache()


def test_rank_rounds_through_partitioned_scatter_2p27(eng, oracle):
    """150 MB of near-duplicate documents: mean LCP in the hundreds forces rank rounds, and n >= 2^27
    sends every rank-array update through scatter_pairs_u32; complete SA and LCP vs the oracle."""
    n = 150_000_000
    host = _gen.near_duplicates(n, ndocs=4, every=300)
    text, sa, lcp = _build(host)
    st = eng.build_stats()
    assert st["rank_rounds"] >= 2 and n >= (1 << 27), st
    exp = oracle.sais(host)
    assert np.array_equal(sa.cpu().numpy().view(np.uint32), exp)
    assert np.array_equal(lcp.cpu().numpy().view(np.uint32), oracle.lcp_kasai(host, exp))
    del text, sa, lcp
    torch.cuda.empty_cache()


def test_chunked_radix_schedule_2p30(eng):
    """1.1 * 10^9 B of DNA: m >= 2^30 elements per pass, the chunked schedule takes over."""
    sys.path.insert(0, ROOT)
    import bench
    from suffix_amd import device as sdev
    n = 1_100_000_000
    host = _gen.dna_fast(n, seed=0x5AF1C5 + 9)
    dev = torch.device("cuda", 0)
    text = torch.from_numpy(host).to(dev)
    eng.profile(True); eng.profile_reset()
    sa = sdev.build_sa(text)
    torch.cuda.synchronize()
    names = {r["name"] for r in eng.profile_report()}
    eng.profile(False)
    assert "radix_hist" in names, names                                  # (only the chunked schedule launches it)
    ok, how = bench.verify_sa_chunked(torch, sdev, text, sa)
    assert ok, how
    del text, sa
    torch.cuda.empty_cache()


def test_chunked_radix_schedule_compressed_keys_2p30(eng):
    """1.1 * 10^9 B of English-like text: compressed 64-bit keys (k_ht_keys) through the chunked schedule of the
    key/value passes (m >= 2^30: no one-sweep status words), deep rounds on 6 * 10^8 tied suffixes; the size-independent
    property gate (permutation, every adjacent pair in order, sampled LCP bytes)."""
    sys.path.insert(0, ROOT)
    import bench
    from suffix_amd import device as sdev
    n = 1_100_000_000
    host = _gen.english_like(n)
    dev = torch.device("cuda", 0)
    text = torch.from_numpy(host).to(dev)
    eng.profile(True); eng.profile_reset()
    sa = sdev.build_sa(text)
    torch.cuda.synchronize()
    names = {r["name"] for r in eng.profile_report()}
    eng.profile(False)
    assert "radix_hist" in names and "ht_keys" in names and "deep_wave" in names, names
    ok, how = bench.verify_sa_chunked(torch, sdev, text, sa)
    assert ok, how
    del text, sa
    torch.cuda.empty_cache()


def test_config4_virtual_ranks(eng):
    """BASELINE config 4 at its stated size on one GPU: 4 * 10^9 B of DNA, the four ranges of plan_ranges built one after
    another by sfx_build_sa_range_packed_u32_dev (what each of the 4 ranks would run), every slice equal to its stretch of
    one single-GPU build of the same text (itself gated: permutation + every adjacent pair in order), sizes summing to n,
    sha256 of the concatenation equal, u64 widening checked; per-rank milliseconds go to gpurun_out/ (copied to
    profiles/r5_config4_virtual.jsonl).  Round 5: the complete array's sha256 is pinned to an oracle run.  tests/_config4.py."""
    import _config4
    n = 4_000_000_000
    free, _total = torch.cuda.mem_get_info()
    if free < 240e9:
        pytest.skip("needs ~230 GB of free HBM for the single-GPU comparison build")
    recs = _config4.rehearse(n, world=4)
    assert len(recs) == 5 and all(r["equals_single_gpu_slice"] for r in recs[:4])
    # (round 5) the array is the ORACLE's: sha256 of the complete suffix array = the pin of scripts/cpu_config4_oracle.py
    # (oracle.sais over all 4 * 10^9 bytes, u32 positions: tests/golden/fullsize_pins.json "c4")
    assert recs[4]["sa_equals_oracle_pin"] is True, recs[4]
    assert max(r["largest_position"] for r in recs[:4]) >= (1 << 31)          # positions beyond 2^31 were placed
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r5_config4_virtual.jsonl"), "w") as f:
            for r in recs:
                f.write(json.dumps(r) + "\n")
    except OSError:
        pass


def _bench_over_rccl(extra, size="20000000", share_gpu=False):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if share_gpu:
        env["SFX_BENCH_SHARE_GPU"] = "1"                 # both ranks on cuda:0, gloo instead of RCCL (bench.py's rehearsal hook)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--size", size, "--configs", "", "--cpu-sample", "0"] + extra,
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["verified"] is True, rec
    return rec


@pytest.mark.parametrize("case", ["even-packed", "ragged-packed", "replicated-fallback", "strong"])
def test_two_gpu_bench_over_rccl(case):
    """bench.py --gpus 2 under torch.distributed.run with the nccl (= RCCL) backend, one rank per GPU: the partitioned build's
    first contact with RCCL, one sub-case per branch of suffix_amd/dist.py -- shards of equal length (packed exchange on the
    global word grid), ragged shards (lengths that are no multiple of the symbols per word: the straddling words are assembled
    from the halos), and a periodic text (the range build reports SFX_ERR_NEEDS_RANKS, the ranks agree with one all-reduce and
    every rank builds the whole array: the replicated fallback).  The gate of every sub-case is verify_partitioned (permutation
    + every adjacent pair, slice boundaries included).  Skipped on 1-GPU boxes (the driver's multi-GPU node runs it); the same
    three shapes run over gloo in tests/test_dist_gloo.py."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    if case == "even-packed":
        rec = _bench_over_rccl([])
        ph = rec["config"]["partitioned_phases_ms"]
        assert ph["range_build"] > 0 and "fallback" not in ph, ph
        assert rec["config"]["text_bytes_total"] == 40000000
    elif case == "ragged-packed":
        rec = _bench_over_rccl(["--ragged"])
        ph = rec["config"]["partitioned_phases_ms"]
        assert ph["range_build"] > 0 and "fallback" not in ph, ph
        assert rec["config"]["text_bytes_total"] == 2 * 20000000 - 1 - 4098
        assert "packed" in str(ph.get("text_exchange", "packed")), ph
    elif case == "strong":
        # --total-size: ONE text cut into N shards (the strong series of BASELINE config 4, here at 40 MB)
        rec = _bench_over_rccl(["--total-size", "40000001"])
        assert rec["scaling"] == "strong" and rec["config"]["text_bytes_total"] == 40000001, rec
    else:
        rec = _bench_over_rccl(["--input", "periodic"], size="3000000")
        ph = rec["config"]["partitioned_phases_ms"]
        assert "fallback" in ph, ph


@pytest.mark.parametrize("case", ["even-packed", "ragged-packed", "replicated-fallback", "strong"])
def test_two_rank_bench_rehearsal_on_one_gpu(case):
    """The same three sub-cases as test_two_gpu_bench_over_rccl on the 1-GPU boxes of this pool: both ranks on cuda:0, gloo in
    place of RCCL (which refuses two ranks on one device) -- bench.py's N > 1 code, its --ragged / --input flags and every
    branch of suffix_amd/dist.py on device memory.  Not a multi-GPU measurement."""
    if case == "even-packed":
        rec = _bench_over_rccl([], size="3000000", share_gpu=True)
        assert "fallback" not in rec["config"]["partitioned_phases_ms"] and rec["config"]["text_bytes_total"] == 6000000
    elif case == "ragged-packed":
        rec = _bench_over_rccl(["--ragged"], size="3000000", share_gpu=True)
        ph = rec["config"]["partitioned_phases_ms"]
        assert "fallback" not in ph and rec["config"]["text_bytes_total"] == 6000000 - 1 - 4098, rec["config"]
        assert "packed" in str(ph.get("text_exchange", "")), ph
    elif case == "strong":
        # the strong series: the same 6 000 001-byte text at N = 2 (two shards of one stream) and at N = 1 (bench.py alone) --
        # scaling "strong", total work fixed
        rec = _bench_over_rccl(["--total-size", "6000001"], share_gpu=True)
        assert rec["scaling"] == "strong" and rec["config"]["text_bytes_total"] == 6000001, rec
        assert "ONE text cut into 2" in rec["config"]["workload"], rec["config"]
        import subprocess
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--total-size", "6000001",
                              "--no-microbench"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
        one = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert one["scaling"] == "strong" and one["n_gpus"] == 1 and one["verified"] is True, one
        assert one["config"]["text_bytes_total"] == 6000001 and one["cpu_baseline"] is None, one
    else:
        rec = _bench_over_rccl(["--input", "periodic"], size="1000000", share_gpu=True)
        assert "fallback" in rec["config"]["partitioned_phases_ms"], rec["config"]
