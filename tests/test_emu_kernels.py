"""CPU tests of the KERNEL LOGIC: the product's .hip sources compiled unchanged
against the fiber emulator (tests/emu), driven through the same C ABI and the
same SuffixTable host mirror, compared bit-exactly with the oracle.  Sizes are
small because every work-item is a fiber on one CPU core."""
import os
import subprocess

import numpy as np
import pytest

import _cases
import _gen
from suffix_amd import Engine

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")


@pytest.fixture(scope="session")
def emu():
    subprocess.check_call(["make", "-s", "-j8", "-C", EMU_DIR])
    return Engine(os.path.join(EMU_DIR, "libsuffix_emu.so"))


def test_literals(emu, oracle, golden):
    _cases.literals(emu, oracle, golden)


def test_search_known_answers(emu, golden):
    _cases.search_known_answers(emu, golden)


def test_parts_roundtrip(emu):
    _cases.parts_roundtrip(emu)


def test_fasta_10k(emu, oracle, golden, fasta):
    _cases.fasta_fixture(emu, oracle, golden, fasta, "AP009048_10000")


def test_single_workgroup_build(emu, oracle):
    _cases.tiny_build(emu, oracle)


def test_random_small(emu, oracle):
    _cases.random_small(emu, oracle, iters=60, max_len=90, seed=101)


def test_unicode(emu, oracle):
    _cases.unicode_strings(emu, oracle, iters=40, seed=5)


def test_structured(emu, oracle):
    _cases.structured(emu, oracle, scale=3)


def test_structured_deep_ties(emu, oracle):
    # overlap-free / Fibonacci words: many small buckets whose members tie beyond the direct
    # comparison depth, next to members that resolve between them (rank consistency)
    import _gen
    _cases.check_text(emu, oracle, _gen.thue_morse(4000).tobytes())
    _cases.check_text(emu, oracle, _gen.fibonacci_string(17))
    _cases.check_text(emu, oracle, (b"ab" * 700 + b"c") * 3)


def test_generated(emu, oracle):
    _cases.generated(emu, oracle, n_dna=9000, n_text=6000)


def test_planted_repeats_switch_text_to_rank_rounds(emu, oracle):
    _cases.planted_repeats(emu, oracle, 7000)


def test_small_buckets_direct_ordering(emu, oracle):
    _cases.small_buckets(emu, oracle)


def test_index_directory_queries(emu, oracle):
    _cases.directory_queries(emu, oracle)


def test_suffix_tree_topology_and_doc_lookup(emu, oracle):
    _cases.suffix_tree_topology(emu, oracle)


def test_suffix_tree_open_list_overflow(emu, oracle):
    """A monotone LCP array (a^9000 b ...): every boundary's search to the right leaves its tile, more of them than the
    list of open searches holds (n / 16 + 4096) -- the tiles finish the overflow themselves and the second launch must
    not read a slot that was reserved but never written (found by the AddressSanitizer run)."""
    _cases.suffix_tree_at_scale(emu, oracle, b"a" * 9000 + b"b" + _gen.dna(3000, seed=2).tobytes(), device="cpu")


def test_key_width_from_symbol_counts(emu, oracle):
    """A large alphabet used unevenly (natural-language text: 4.2 bits of entropy in 7-bit symbols; here Zipf bytes) takes 64-bit keys --
    compressed ones -- from 2^16 bytes on, although 4 symbols x log2(sigma) would cover log2(n) + 1 bits; the same
    alphabet used evenly keeps 32-bit keys (choose_key, sfx_sa.hip)."""
    from suffix_amd import SuffixTable
    zipf = np.minimum(np.random.default_rng(21).zipf(1.3, 66000), 150).astype(np.uint8).tobytes()     # ~3 bits of entropy in 8-bit symbols
    for text, bits in ((zipf, 64), (_gen.uniform_bytes(66000, 150, 4, base=40).tobytes(), 32)):
        emu.profile(True); emu.profile_reset()
        st = SuffixTable(text, engine=emu)
        sa = st.table()
        names = {r["name"] for r in emu.profile_report()}
        emu.profile(False)
        stats = emu.build_stats()
        assert stats["key_bits"] == bits and ("ht_keys" in names) == (bits == 64), (stats, sorted(names))
        assert np.array_equal(sa, oracle.sais(text))


def test_random_medium_sweep(emu, oracle):
    _cases.random_medium_sweep(emu, oracle, iters=7, max_len=4000, seed=5)


def test_fused_sa_lcp(emu, oracle):
    _cases.fused_lcp_tails(emu, oracle, iters=30)


def test_multi_tile_and_multi_block(emu, oracle):
    # > 4096-key radix tiles, several persistent workgroups, u32 and u64 initial keys
    import _gen
    _cases.check_text(emu, oracle, _gen.dna(13000, seed=77).tobytes())
    _cases.check_text(emu, oracle, _gen.uniform_bytes(9000, 200, 78).tobytes())


def test_error_paths(emu):
    import ctypes
    import numpy as np
    lib = emu.lib
    assert lib.sfx_build_sa_u32(None, 5, None) == 1                      # SFX_ERR_ARG
    assert lib.sfx_build_sa_u32(None, 0, None) == 0                      # n == 0 is fine
    assert lib.sfx_build_sa_u32(None, 1 << 32, None) == 2                # > u32::MAX (:380)
    t = np.frombuffer(b"banana", dtype=np.uint8)
    sa = np.zeros(6, dtype=np.uint32)
    ws = np.zeros(16, dtype=np.uint8)
    rc = lib.sfx_build_sa_u32_dev(t.ctypes.data, 6, sa.ctypes.data, ws.ctypes.data, 16, None)
    assert rc == 5                                                       # SFX_ERR_WORKSPACE
    assert b"workspace" in lib.sfx_strerror(5)
    with pytest.raises(OverflowError):
        emu.check(2, "x")


def test_build_stats_and_profile(emu):
    import _gen
    from suffix_amd import SuffixTable
    emu.profile(True)
    emu.profile_reset()
    with _cases.general_build(emu):
        SuffixTable(_gen.dna(5000).tobytes(), engine=emu)
    st = emu.build_stats()
    assert st["n"] == 5000 and st["sigma"] == 4 and st["bits_per_symbol"] == 2
    assert st["key_bits"] == 32 and st["symbols_per_key"] == 16
    rep = {r["name"]: r for r in emu.profile_report()}
    emu.profile(False)
    assert rep["radix_scatter_text_u32"]["launches"] >= 1 and rep["radix_scatter_u32"]["launches"] >= 3
    assert rep["radix_scatter_u32"]["algo_bytes"] > 0 and "pack_text" in rep


@pytest.mark.parametrize("nranges", [1, 3, 11])
def test_range_build_virtual_ranks(emu, oracle, nranges):
    import _gen
    _cases.range_slices(emu, oracle, _gen.dna(3001, seed=8).tobytes(), nranges, packed=(nranges != 3))
    _cases.range_slices(emu, oracle, _gen.english_like(2503).tobytes(), nranges, packed=(nranges == 3))
    _cases.range_slices(emu, oracle, (b"ab" * 150 + b"b"), nranges)
    rng = np.random.default_rng(5)
    _cases.range_slices(emu, oracle, rng.integers(0, 256, 1500, dtype=np.uint8).tobytes(), nranges, packed=True)
    _cases.range_slices(emu, oracle, (rng.integers(0, 11, 1500, dtype=np.uint8) + 65).tobytes(), nranges)
