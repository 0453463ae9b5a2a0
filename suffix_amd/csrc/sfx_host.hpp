// sfx_host.hpp -- host-side plumbing shared by the translation units of
// libsuffix_hip.so: status codes, HIP error capture, a bump allocator over the
// caller's device workspace, and the optional per-kernel event profiler.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/suffix_hip.h"
#include "sfx_device.hpp"

namespace sfx {

// ---- error capture -----------------------------------------------------------
void note_hip_error(hipError_t e, const char* what, const char* file, int line);

#define SFX_HIP(expr)                                                   \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) {                                         \
            ::sfx::note_hip_error(e_, #expr, __FILE__, __LINE__);       \
            return SFX_ERR_HIP;                                         \
        }                                                               \
    } while (0)

#define SFX_TRY(expr)                       \
    do {                                    \
        int s_ = (expr);                    \
        if (s_ != SFX_OK) return s_;        \
    } while (0)

// ---- device workspace carving --------------------------------------------------
// All device scratch comes from one caller-provided allocation (torch tensor in
// bench.py, hipMalloc in the host-pointer entry points): nothing is allocated
// inside a build, so the build is a pure kernel/memcpy sequence on one stream.
constexpr uint64_t kArenaAlign = 256;       // every carve starts on a multiple of it; the gap behind a carve belongs to nobody
struct Arena {
    char* base = nullptr;
    uint64_t size = 0, used = 0;
    bool overflow = false;
    Arena() {}
    Arena(void* p, uint64_t bytes) : base((char*)p), size(bytes) {}
    template <class T> T* take(uint64_t count)
    {
        uint64_t bytes = (count * sizeof(T) + kArenaAlign - 1) & ~(kArenaAlign - 1);
        if (used + bytes > size) { overflow = true; return nullptr; }
        T* p = (T*)(base + used);
        used += bytes;
        return p;
    }
};
// same arithmetic without memory, for *_workspace_bytes()
struct ArenaSizer {
    uint64_t used = 0;
    template <class T> void take(uint64_t count) { used += (count * sizeof(T) + kArenaAlign - 1) & ~(kArenaAlign - 1); }
};

// ---- development hooks -----------------------------------------------------------
// The SFX_* environment knobs (small tiles for the emulator, A/B switches for lab runs) exist only in builds
// made with -DSFX_DEV_HOOKS: the kernel-logic emulator (tests/emu) and libsuffix_hip_dev.so (`make dev`).
// The shipped libsuffix_hip.so reads no environment variable: its routes depend on (alphabet, n) alone.
#ifdef SFX_DEV_HOOKS
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif

// ---- launch geometry -----------------------------------------------------------
// Streaming kernels use a fixed-size grid of persistent workgroups, each owning a
// contiguous chunk (256 CUs x 8 resident 256-thread workgroups = 2048), so
// per-workgroup partials stay tiny and chunk carries need one small scan.
constexpr unsigned kMaxGrid = 2048;
constexpr int kRadixBits = 8;             // key bits consumed per radix pass
constexpr int kRadix = 1 << kRadixBits;   // buckets per pass

struct Chunking {
    unsigned blocks;        // workgroups launched
    uint64_t tiles;         // total tiles
    uint64_t tiles_per_block;
};
// test hook: SFX_MAX_GRID=<k> caps the persistent grid so that small inputs still give
// every workgroup a multi-tile chunk (carries across tiles get exercised on the emulator)
inline unsigned grid_cap()
{
    static const unsigned cap = [] {
        const char* e = dev_env("SFX_MAX_GRID");
        int v = e ? atoi(e) : 0;
        return (v >= 1 && v <= (int)kMaxGrid) ? (unsigned)v : kMaxGrid;
    }();
    return cap;
}
inline Chunking make_chunking(uint64_t items, uint64_t tile, unsigned max_blocks = kMaxGrid)
{
    Chunking c;
    if (max_blocks > grid_cap()) max_blocks = grid_cap();
    c.tiles = (items + tile - 1) / tile;
    if (c.tiles == 0) c.tiles = 1;
    c.tiles_per_block = (c.tiles + max_blocks - 1) / max_blocks;
    c.blocks = (unsigned)((c.tiles + c.tiles_per_block - 1) / c.tiles_per_block);
    return c;
}

// ---- profiler ---------------------------------------------------------------------
bool profile_on();
void profile_begin(const char* name, hipStream_t st, double algo_bytes);
void profile_end(hipStream_t st);

// Launch wrapper: every kernel in the library goes through this.
#define SFX_LAUNCH(name, algo_bytes, kernel, grid, block, stream, ...)                  \
    do {                                                                                \
        if (::sfx::profile_on()) ::sfx::profile_begin(name, stream, (double)(algo_bytes)); \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__);    \
        if (::sfx::profile_on()) ::sfx::profile_end(stream);                            \
        SFX_HIP(hipGetLastError());                                                     \
    } while (0)

// Small device -> host read-backs (round totals, alphabet bins, counters): every build waits on three or four of them with
// an idle GPU.  A one-wave kernel posts the words into a per-thread page of pinned host memory that the device writes
// directly (fine-grained: hipHostMalloc) followed by a sequence word, and the host polls that word: 11.0 us per (kernel +
// read-back) against 15.2 for an asynchronous copy into pinned memory + hipStreamSynchronize and 39.5 for a copy into
// pageable memory (lab/sync_probe.hip, MI355X).  The stream is queried every few thousand polls: a stream that has drained
// (or failed) without the word arriving ends the wait with its error.
#ifndef SFX_EMULATED
namespace detail {
constexpr unsigned kPostWords = 64;                                  // 256 bytes per read-back
static __global__ void __launch_bounds__(64) k_post_words(const uint32_t* __restrict__ src, unsigned words, volatile uint32_t* host,
                                                   uint32_t seq)
{
    const unsigned lane = threadIdx.x;
    if (lane < words) host[lane] = src[lane];
    __threadfence_system();
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) host[kPostWords] = seq;                           // (its own 64-byte block, after the payload)
}
}  // namespace detail
#endif
inline int read_back(void* dst, const void* d_src, size_t bytes, hipStream_t st)
{
    // One pinned allocation per thread, two disjoint areas: [0, kStage) stages the copies of the fallback below (and of every
    // read above 256 bytes: the 2 KiB of byte bins), [kStage, kStage + 512) is the posted area -- payload + sequence word.  They
    // must not overlap: a staged copy that left a stale value where the sequence word lives would end a later poll before its
    // kernel has run (the bins of a text with spaces wrote 1 over the word, the first post polled for 1).
    // Portable + coherent + mapped: kernels of whichever device is current later write the page, and the host sees the words
    // without a flush.
    constexpr size_t kStage = 4096, kPostArea = 512;
    thread_local void* stage = nullptr;
    thread_local bool tried = false;
    if (!tried) {
        tried = true;
        if (hipHostMalloc(&stage, kStage + kPostArea, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            stage = nullptr;
            (void)hipGetLastError();
        }
        if (stage) std::memset(stage, 0, kStage + kPostArea);
    }
#ifndef SFX_EMULATED
    thread_local uint32_t seq = 0;
    if (stage && bytes > 0 && bytes <= detail::kPostWords * 4 && (bytes & 3u) == 0 && (reinterpret_cast<uintptr_t>(d_src) & 3u) == 0) {
        static_assert((detail::kPostWords + 16) * 4 <= kPostArea, "payload + the sequence word's own 64-byte block");
        volatile uint32_t* const host = reinterpret_cast<volatile uint32_t*>(static_cast<char*>(stage) + kStage);
        if (++seq == 0) seq = 1;
        hipLaunchKernelGGL(detail::k_post_words, dim3(1), dim3(64), 0, st, static_cast<const uint32_t*>(d_src), (unsigned)(bytes / 4), host, seq);
        SFX_HIP(hipGetLastError());
        for (unsigned spins = 1;; spins++) {
            if (host[detail::kPostWords] == seq) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();                                  // (neutral on every latency measured: profiles/r5_post_page_ab.txt)
#endif
            if ((spins & 0xFFFu) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipErrorNotReady) continue;
                SFX_HIP(q);                                          // (a failed stream)
                if (host[detail::kPostWords] == seq) break;
                SFX_HIP(hipStreamSynchronize(st));                   // drained without the word (cannot happen): take the copy below
                goto by_copy;
            }
        }
        for (size_t i = 0; i < bytes / 4; i++) static_cast<uint32_t*>(dst)[i] = host[i];
        return SFX_OK;
    }
by_copy:
#endif
    void* via = (stage && bytes <= kStage) ? stage : dst;
    SFX_HIP(hipMemcpyAsync(via, d_src, bytes, hipMemcpyDeviceToHost, st));
    SFX_HIP(hipStreamSynchronize(st));
    if (via != dst) std::memcpy(dst, via, bytes);
    return SFX_OK;
}

// ---- cross-TU entry points ----------------------------------------------------------
struct BuildStats;   // = sfx_build_stats

// LSD radix sorts (sfx_radix.hip), 8 bits per pass over element bits [bit_lo, bit_hi).
// `scratch` is radix_scratch_words(m) u32 of device memory.
//  - E64: one 64-bit word per suffix, (32-bit key << 32) | suffix.  e0/e1 ping-pong.
//    With `text` the first pass computes element i from the packed text (e0 holds nothing).
//    With `split_v` the last pass writes the suffix halves to split_v and the key halves to
//    a u32 array carved from whichever of e0/e1 it does not read (*split_k_out);
//    otherwise *result_in_1 tells which buffer holds the sorted elements.
//  - KV: u64 keys + u32 values, (k0,v0)/(k1,v1) ping-pong; with `text` the first pass
//    reads (packed_key64(text, i), i) instead of (k0, v0).
uint64_t radix_scratch_words(uint64_t m);
//    `ties` (with split_v; round 6): a caller that only needs to know WHICH elements share their whole key with a neighbour, not
//    the sorted keys.  When the hybrid route runs and no sub-bucket is oversized, its LDS sort (k_bucket_sort<.., true>) writes no
//    keys (*split_k_out = nullptr, unless want_keys) and leaves one bit mask over the m slots of split_v (bit r & 31 of word r / 32 = slot r, zero
//    words behind): tmask = the element shares its key with a neighbour.  lmask, hmask: two more arrays of the same size for the
//    caller (lmask zeroed).  produced = false: the sorted keys are in *split_k_out as always.  The masks stay valid until e0 /
//    e1 are written again.
struct TieRecords {
    bool produced;
    const uint32_t* tmask;
    uint32_t* lmask;
    uint32_t* hmask;
    bool want_keys;                // in: leave the sorted 32-bit keys in *split_k_out as well (the fused LCP reads them once)
};
//    A producer that had the keys in registers anyway (the range filter) may have counted the
//    digits itself: `hist_blocks` workgroups' counts at radix_partial(scratch)[(pass * 256 +
//    digit) * hist_blocks + workgroup]; only honoured when radix_e64_presort_hist() said so.
int radix_sort_e64(uint64_t* e0, uint64_t* e1, uint64_t m, int bit_lo, int bit_hi, uint32_t* scratch,
                   hipStream_t st, int* result_in_1, sfx_build_stats* stats, const PackedText* text,
                   uint32_t* split_v, uint32_t** split_k_out, unsigned hist_blocks = 0, int elem_bits = 0, TieRecords* ties = nullptr);
// > 0: an E64 sort of m elements on bits [bit_lo, bit_hi) takes digit counts from its producer,
// from at most this many workgroups
unsigned radix_e64_presort_hist(uint64_t m, int bit_lo, int bit_hi);
bool radix_e64_hybrid_expected(uint64_t m, int key_bits);
inline uint32_t* radix_partial(uint32_t* scratch) { return scratch; }
// kv12_cap: the caller's statement that (k0, v0) and (k1, v1) are two DISJOINT regions each carved as "kv12_cap keys, then
// (behind at most one arena alignment gap that belongs to nobody) kv12_cap values" (carve_sa) -- such a pair also serves as one
// array of up to kv12_cap 12-byte (key, suffix) elements for the middle passes.  0 = separate arrays, never reinterpreted.
int radix_sort_kv64(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, uint64_t m, int bit_lo,
                    int bit_hi, uint32_t* scratch, hipStream_t st, int* result_in_1,
                    sfx_build_stats* stats, const PackedText* text, uint32_t* last_v = nullptr, uint64_t kv12_cap = 0);
// the same over the compressed keys (k_ht_keys) of all m = text.n suffixes; ht = the code table on the device
int radix_sort_ht64(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, uint64_t m, uint32_t* scratch, hipStream_t st,
                    int* result_in_1, sfx_build_stats* stats, const PackedText& text, const uint32_t* ht, uint32_t* last_v = nullptr,
                    uint64_t kv12_cap = 0, int ctx_sigma = 0);         // ctx_sigma > 0: context codes (k_ht_keys_ctx), ht holds their tables
// Segmented sort of the large buckets of a refinement round (sfx_radix.hip).  Scratch:
//   segs      8 B per segment, filled by the caller          tiles    32 B per tile (<= nlarge / tile + nseg)
//   tilehist  4 KiB per tile of a multi-tile segment (<= 2 * nlarge / tile)
//   segexcl   4 KiB per multi-tile segment (<= nlarge / tile)   status   1 KiB per such tile
//   counters  16 u32
struct LcpEmit;
struct SegSort {
    void* segs;
    void* tiles;
    uint32_t* tilehist;
    uint32_t* segexcl;
    uint32_t* status;
    uint64_t status_words;
    uint32_t* counters;
};
uint32_t seg_tile_elems(bool kv);
// segments of at most skip_upto members are left out of the table (someone else sorts them)
int segmented_layout(const SegSort& q, uint32_t nseg, bool kv, hipStream_t st, uint32_t skip_upto = 0);
// one entry of the tile table (32 bytes): list positions [begin, begin + count) of one segment
struct SegTileHost { uint32_t begin, count, seg_start, info, mseg, pad[3]; };
int segmented_sort_e64(uint64_t* A, uint64_t* B, const SegSort& q, uint32_t nseg, uint64_t nlarge, uint32_t* V,
                       uint8_t* F8, hipStream_t st, sfx_build_stats* stats, const LcpEmit& emit, uint16_t* Hd = nullptr,
                       uint32_t wsym = 0);
// target[idx] = val for m (idx << 32 | val) pairs, idx < n: one partitioning pass on the top
// bits of idx, then a scatter whose writes stay inside a cache-sized window of `target`.
// `tmp` is m u64 of scratch, `radix_scratch` as for the sorts.  Pays for 4n >> Infinity Cache.
int scatter_pairs_u32(uint64_t* pairs, uint64_t* tmp, uint64_t m, uint64_t n, uint32_t* target,
                      uint32_t* radix_scratch, hipStream_t st, sfx_build_stats* stats, unsigned hist_blocks = 0);
// the producer of the pairs may count the digits of the partitioning passes itself: bits [*lo_out, *nb_out) of the
// suffix index, 8 per pass, into radix_scratch[(pass * 256 + digit) * workgroups + workgroup]; returns how many
// workgroups at most (0: do not, the sort counts for itself)
unsigned scatter_pairs_presort_hist(uint64_t m, uint64_t n, int* lo_out, int* nb_out);
// from 2^27 entries (a 512 MB target) up; SFX_PARTITION_MIN=<entries> is a test hook
inline uint64_t partitioned_scatter_min()
{
    static const uint64_t v = [] {
        const char* e = dev_env("SFX_PARTITION_MIN");
        long long x = e ? atoll(e) : 0;
        return x > 0 ? (uint64_t)x : (1ull << 27);
    }();
    return v;
}
inline int radix_pass_count(int bit_lo, int bit_hi) { return (bit_hi - bit_lo + 7) / 8; }

// ---- refinement rounds on 64-bit (key2, suffix) elements (sfx_tile.hip) -----------------------
// One round over the active list (m elements): V[p] = suffix, G[p] = list position of the head of
// p's bucket; key2 of a suffix = its next symbols (text round) or the rank of the suffix h symbols on
// (rank round).  On return V holds, at the same list positions, the suffixes of every bucket
// ordered by key2; F the head / singleton bits and part_* the per-chunk partials that
// k_groups_scan + k_groups_apply consume (chunks of kFlagChunkTile elements, make_chunking).
constexpr int kFlagChunkTile = 8192;
// Fused LCP (sfx_build_sa_lcp_u32_dev): when a round puts two neighbours of a bucket into different
// classes, their common prefix is h + the equal leading symbols of their key2 values (text rounds) --
// known right there, no text access.  Pairs split by rank rounds or by the direct pass only get a lower
// bound: 0x80000000 | h, finished on the text at the end (k_lcp_pending).  lcp == nullptr: off.
struct LcpEmit {
    uint32_t* lcp;
    const uint32_t* S;          // SA slot of every list position
    uint32_t h;                 // symbols the members of a bucket share
    uint32_t n;
    int rank_mode;
    int field_bits;             // width of the symbol field of a 32-bit text key2
    uint32_t inv_bits;          // ceil(65536 / bits)
    int field_bits64;           // ... of a 64-bit text key2 (flag in bit 63)
};
constexpr uint32_t kLcpBoundFlag = 0x80000000u;
// LCP of a class head (suffix sb, key kb) with the member of the preceding class that stands in front of it NOW
// (suffix sa, key ka).  All members of that class share ka, so the symbols the keys have in common are the same
// whichever of them ends up in front of sb -- except that the key of a suffix ending inside it is zero-padded and
// may imitate the smallest symbol: when that is what sa is (n - sa < common prefix of the keys), the value depends
// on which member the class's own resolution puts last, and the pair is left pending (decided on the text once
// the suffix array is final).  sb's own length caps the value for good.
// (depth: symbols the two suffixes' bucket shares -- L.h in a rank round, the bucket's own in a deep text round)
__device__ __forceinline__ uint32_t lcp_from_key2_at(const LcpEmit& L, uint32_t depth, uint32_t ka, uint32_t kb, uint32_t sa, uint32_t sb)
{
    if (L.rank_mode) return kLcpBoundFlag | L.h;
    const uint32_t la = L.n - sa, lb = L.n - sb;
    if (!(ka & kb & 0x80000000u)) return la < lb ? la : lb;   // one of them ends before offset h (a class of its own): the shorter is a prefix
    const uint32_t x = ka ^ kb;                          // (!= 0: different classes)
    const uint32_t lz = (uint32_t)__clz((int)x) - (32u - (uint32_t)L.field_bits);
    const uint32_t v = depth + ((lz * L.inv_bits) >> 16);
    if (v > la) return kLcpBoundFlag | depth;
    return v < lb ? v : lb;
}
__device__ __forceinline__ uint32_t lcp_from_key2(const LcpEmit& L, uint32_t ka, uint32_t kb, uint32_t sa, uint32_t sb)
{
    return lcp_from_key2_at(L, L.h, ka, kb, sa, sb);
}
constexpr unsigned kDeepSlotWords = 1024 * 8;
// flag bytes of a round (F8): 1 = first of its (bucket, key2) class, 2 = class of one, 4 = placed by the deep kernel,
// kRankKept = member of the class that starts where its old bucket started (its head slot, hence its rank, is unchanged)
constexpr unsigned kRankKept = 8u;
struct TileRound {
    LcpEmit emit;
    const uint32_t* G;
    uint32_t* V;
    uint8_t* F8;                  // m bytes (+ 8), scratch
    uint16_t* Hd;                 // deep text rounds: symbols the members of p's bucket share, per list position (in / out)
    uint32_t wsym;                // ... and what a split by the large-bucket path adds to it (text_round_symbols)
    uint32_t h;                   // symbols every bucket of the list shares at least (the round's h)
    uint16_t* F;
    uint32_t* part_head; uint32_t* part_keep; uint32_t* part_ghead;
    uint32_t* block_counts;       // kMaxGrid
    uint32_t* part_pairs;         // rank rounds: per-chunk count of the members whose rank changes (kMaxGrid; nullptr: not wanted)
    uint32_t* totals;
    unsigned long long* counters; // 4
    unsigned long long* deep_slots; // kDeepSlotWords: per-wave-class counter lines of the deep text rounds
    uint64_t* EA; uint64_t* EB;   // large buckets: m u64 each (key2 << 32 | suffix at the list positions, ping-pong;
                                  // 64-bit text keys: the key arrays, values ping-pong between V and V_other)
    uint32_t* V_other;
    SegSort seg;                  // scratch of the segmented sort
};
LcpEmit make_lcp_emit(uint32_t* lcp, const uint32_t* S, const PackedText& pt, uint64_t h, bool rank_mode);
// deep text round (sfx_tile.hip): buckets of a few hundred members are finished inside one wave, each from its own
// depth r.Hd (deep_text_symbols(pt) symbols per gather); larger ones are split by their next text_round_symbols(pt) symbols
int deep_text_symbols(const PackedText& pt);
int text_key64_symbols(const PackedText& pt);       // symbols of a 64-bit text key (flag in bit 63)
inline int text_round_symbols(const PackedText& pt) { return pt.kbits == 32 ? pt.spw - 1 : pt.spw; }
int deep_round_text(const PackedText& pt, const TileRound& r, uint64_t m, hipStream_t st, sfx_build_stats* stats);
// (needs n - 1 + h < 2^32: key2 = rank + h)
int tile_round_rank(const uint32_t* isa, uint64_t n, uint64_t h, const TileRound& r, uint64_t m, hipStream_t st,
                    sfx_build_stats* stats);
__global__ void k_scan_block_counts(uint32_t* __restrict__ counts, unsigned nb, uint32_t* __restrict__ out_total);

uint64_t sa_workspace_bytes(uint64_t n);
// texts of up to tiny_limit() bytes: one workgroup, one launch (sfx_tiny.hip); *done = false: not a text for it
uint64_t tiny_max_default();
uint64_t tiny_limit();
void tiny_set_limit(uint64_t n);
int tiny_build_sa_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* ws, hipStream_t st, bool* done);
int build_sa_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* ws, uint64_t ws_bytes,
                     hipStream_t st);
int pack_small_alphabet(const uint8_t* d_text, uint64_t n, int max_bits, void* small, uint32_t* d_packed,
                        hipStream_t st, PackedText* pt, bool* packed);
uint64_t sa_lcp_workspace_bytes(uint64_t n);
int build_sa_lcp_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, uint32_t* d_lcp, void* ws,
                         uint64_t ws_bytes, hipStream_t st);
// lcp[r] == 0xFFFFFFFF marks the pairs still to compare (they share their first h0 symbols unless one
// of them ends earlier); *done = false if a pair reached the direct cap (caller recomputes the array)
int lcp_finish_pending_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint32_t* d_lcp, uint64_t h0,
                           void* ws, uint64_t ws_bytes, hipStream_t st, bool* done);
uint64_t lcp_workspace_bytes(uint64_t n);
int build_lcp_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint32_t* d_lcp,
                      void* ws, uint64_t ws_bytes, hipStream_t st);
int query_batch_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint64_t sa_len,
                    const uint8_t* d_q, const uint64_t* d_qoff, uint64_t nq, uint32_t* d_start,
                    uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, hipStream_t st);
// bucket directory of the resident index (sfx_query.hip)
int dir_shape(uint64_t n, int bits, int* k_out, int* dbits_out, uint64_t* entries_out);
uint64_t dir_scratch_words(uint64_t entries);
int dir_build_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, const uint16_t* host_lut256, int bits, int k,
                  int dbits, uint64_t entries, uint16_t* d_lut256, uint32_t* d_dir, uint32_t* d_scratch, hipStream_t st,
                  uint64_t* bad_out);
int query_batch_dir_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, const uint32_t* d_dir,
                        const uint16_t* d_lut256, int bits, int k, int dbits, const uint8_t* d_q, const uint64_t* d_qoff, uint64_t nq,
                        uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, hipStream_t st);
uint64_t key_tree_words(uint64_t n);
int key_tree_build_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint64_t* d_tree, uint64_t* level_offsets_out,
                       int* levels_out, hipStream_t st);
int query_batch_tree_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, const uint64_t* d_tree,
                         const uint64_t* level_offsets, int levels, const uint8_t* d_q, const uint64_t* d_qoff, uint64_t nq,
                         uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, hipStream_t st,
                         void* scratch, bool ordered, const uint32_t* d_dir, const uint16_t* d_lut256, int bits, int k, int dbits);
uint64_t query_scratch_bytes(uint64_t nq, bool ordered);
uint64_t query_two_phase_min();
int byte_presence_host(const uint8_t* d_text, uint64_t n, void* d_small4k, unsigned long long* host_bins256,
                       hipStream_t st);
int build_lcp_range_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa_part, uint64_t count,
                            uint32_t prev_suffix, uint32_t* d_lcp_part, hipStream_t st);
int widen_u32_to_u64_dev(const uint32_t* d_in, uint64_t count, uint64_t* d_out, hipStream_t st);
int byte_histogram_dev(const uint8_t* d_text, uint64_t begin, uint64_t end, uint64_t* d_bins,
                       hipStream_t st);
int key_histogram_dev(const uint8_t* d_text, uint64_t n, uint64_t begin, uint64_t end,
                      const uint64_t* d_byte_bins, int top_bits, uint64_t* d_bins, hipStream_t st);
uint64_t sa_range_workspace_bytes(uint64_t n, uint64_t max_count);
int build_sa_range_u32_dev(const uint8_t* d_text, uint64_t n, const uint64_t* d_byte_bins,
                           int top_bits, uint32_t bin_lo, uint32_t bin_hi, uint64_t capacity,
                           uint32_t* d_sa_part, uint64_t* count_out, void* ws, uint64_t ws_bytes,
                           hipStream_t st, const uint32_t* d_packed_in = nullptr);
int pack_text_dev(const uint8_t* d_text, uint64_t count, const uint64_t* d_byte_bins, uint8_t* d_lut256,
                  uint32_t* d_words, uint64_t n_words, hipStream_t st);

// suffix-tree topology / generalized suffix array (sfx_tree.hip)
uint64_t lcp_intervals_workspace_bytes(uint64_t n);
int lcp_intervals_dev(const uint32_t* d_lcp, uint64_t n, uint32_t* d_lb, uint32_t* d_rb, uint32_t* d_node, uint32_t* d_parent,
                      uint32_t* d_leaf_parent, void* ws, uint64_t ws_bytes, hipStream_t st);
int doc_lookup_dev(const uint32_t* d_pos, uint64_t count, const uint64_t* d_starts, uint64_t ndocs, uint32_t* d_doc,
                   uint32_t* d_offset, hipStream_t st);

sfx_build_stats& tls_build_stats();

}  // namespace sfx
