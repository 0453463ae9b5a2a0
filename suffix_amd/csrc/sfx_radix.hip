// sfx_radix.hip -- device-wide LSD radix sort, 8 key bits (256 buckets) per pass.
//
// This is the "bucket" engine of the suffix sorter: where the reference keeps
// per-symbol bucket head/tail pointers in `Bins` (src/table.rs:671-750) and
// scatters one suffix at a time (head_insert/tail_insert :723-736), the GPU
// engine distributes whole arrays of suffixes 256 buckets at a time.
//
// Element layouts
//   E64   one 64-bit word per suffix: (32-bit key << 32) | suffix index.  Used
//         whenever the key fits 32 bits (DNA: 16 symbols).  One 8-byte load, one
//         8-byte LDS staging slot and one 8-byte store per element and pass; a
//         bucket's share of a 4096-element tile leaves as one 128-byte run.
//         The last pass may "split": keys -> a u32 array, suffixes -> the SA.
//   KV    64-bit key array + 32-bit suffix array (composite keys of the
//         refinement rounds, 2*spw-symbol keys of large alphabets).
// The first pass of an initial sort is "text-fed": element i is computed from
// the packed text, the unsorted key array is never materialised.
//
// Two schedules for a pass (same tile engine):
//   one-sweep  (default, m < 2^30)  digit totals of ALL passes come from one
//         up-front histogram kernel; a pass is a single kernel in which tiles are
//         handed out by an atomic ticket and each tile obtains its bucket offsets
//         by decoupled look-back over a [tile][256] status array (agent-scope
//         relaxed atomics; flag and value share one 32-bit word, so no fence is
//         needed).  Traffic per pass and element: read + write, nothing else.
//   chunked    every workgroup owns one contiguous chunk; per pass a histogram
//         kernel counts the chunk's digits, a scan kernel turns the
//         [256][workgroups] matrix into offsets, the scatter kernel walks the
//         chunk.  No inter-workgroup communication inside a launch.
//
// Tile engine (k_radix_pass): wave w owns 64*KPT consecutive elements, 64 per
// round, so (wave, round, lane) order is memory order and the ranking is stable:
//   rank    each round, lanes with equal digits find each other through a 64-bit
//           match mask in LDS (ds_or_b64 of the lane bit, read back, lowest lane
//           of each digit clears the mask and advances the wave's digit count) --
//           ~10 VALU + 5 LDS instructions per key instead of the ~50 VALU of an
//           8-ballot match (the ballot form is kept as a variant);
//   scan    thread d sums digit d over the 4 waves, block scan -> tile-local
//           bucket starts; bucket heads from look-back (one-sweep) or from the
//           running cursor kept in thread d's register (chunked);
//   reorder elements go to their tile-local slot in LDS, are read back in slot
//           order and leave as one contiguous run per bucket.
// No MFMA anywhere: pure scan/scatter, HBM-bound by design.
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>

#include "sfx_host.hpp"

namespace sfx {

constexpr uint32_t kStatusAgg = 1u << 30;        // tile aggregate published
constexpr uint32_t kStatusPrefix = 2u << 30;     // inclusive prefix published
constexpr uint32_t kStatusValue = (1u << 30) - 1u;
constexpr int kMaxPasses = 8;
constexpr unsigned kHistAllGrid = 1024;          // workgroups of the up-front histogram

// ---- element sources / sinks ---------------------------------------------------------
struct SrcE64 {
    static constexpr bool kHasVal = false;
    static constexpr bool kFromText = false;
    const uint64_t* in;
    __device__ __forceinline__ uint64_t key(uint64_t i) const { return in[i]; }
    __device__ __forceinline__ uint32_t val(uint64_t) const { return 0u; }
};
struct SrcText32 {
    static constexpr bool kHasVal = false;
    static constexpr bool kFromText = true;
    PackedText t;
    __device__ __forceinline__ uint64_t key(uint64_t i) const
    {
        return ((uint64_t)packed_key32(t, i) << 32) | (uint64_t)(uint32_t)i;
    }
    __device__ __forceinline__ uint32_t val(uint64_t) const { return 0u; }
};
// The same with `extra` more key bits in the element where the suffix index leaves room (round 6; the hybrid route, m <= 2^28:
// a suffix index needs 28 bits): element = key of 32 + extra bits << (32 - extra) | suffix.  The order of whole elements is
// the order of (longer key, suffix); the top 32 bits are the 32-bit key as before.  Two more symbols of DNA in the key: a
// sixteenth of the ties (k_bucket_sort<.., true>, sbits).
struct SrcText36 {
    static constexpr bool kHasVal = false;
    static constexpr bool kFromText = true;
    PackedText t;
    int extra;                      // whole symbols' worth of bits, <= 4
    bool wide;                      // more than one symbol: the window of two packed words does not always hold them
    __device__ __forceinline__ uint64_t key(uint64_t i) const
    {
        uint64_t k64;
        if (wide) {
            k64 = packed_key64(t, i);                                    // 2 * kbits = 64 bits of symbols (kbits == 32 here), a 12-byte load
        } else {
            // (one symbol more than the 32-bit key: the two words packed_key32 loads hold it at any offset)
            const uint64_t q = packed_word_index(t, i);
            const unsigned off = packed_word_offset(t, i, q);
            uint64_t pair;
            __builtin_memcpy(&pair, t.words + q, 8);
            const uint64_t both = ((pair & 0xFFFFFFFFull) << 32) | (pair >> 32);
            k64 = both << (off * (unsigned)t.bits);
        }
        return ((k64 >> (32 - extra)) << (32 - extra)) | (uint64_t)(uint32_t)i;
    }
    __device__ __forceinline__ uint32_t val(uint64_t) const { return 0u; }
};
struct SrcKV {
    static constexpr bool kHasVal = true;
    static constexpr bool kFromText = false;
    const uint64_t* k;
    const uint32_t* v;
    __device__ __forceinline__ uint64_t key(uint64_t i) const { return k[i]; }
    __device__ __forceinline__ uint32_t val(uint64_t i) const { return v[i]; }
};
struct SrcKeyIota {                 // keys in an array, value = index (the compressed keys of k_ht_keys)
    static constexpr bool kHasVal = true;
    static constexpr bool kFromText = false;
    const uint64_t* k;
    __device__ __forceinline__ uint64_t key(uint64_t i) const { return k[i]; }
    __device__ __forceinline__ uint32_t val(uint64_t i) const { return (uint32_t)i; }
};
struct SrcText64 {
    static constexpr bool kHasVal = true;
    static constexpr bool kFromText = true;
    PackedText t;
    __device__ __forceinline__ uint64_t key(uint64_t i) const { return packed_key64(t, i); }
    __device__ __forceinline__ uint32_t val(uint64_t i) const { return (uint32_t)i; }
};
struct DstE64 {
    uint64_t* out;
    __device__ __forceinline__ void store(uint32_t d, uint64_t key, uint32_t) const { out[d] = key; }
};
struct DstSplit32 {
    uint32_t* k;
    uint32_t* v;
    __device__ __forceinline__ void store(uint32_t d, uint64_t key, uint32_t) const
    {
        k[d] = (uint32_t)(key >> 32);
        v[d] = (uint32_t)key;
    }
};
struct DstKV {
    uint64_t* k;
    uint32_t* v;
    __device__ __forceinline__ void store(uint32_t d, uint64_t key, uint32_t val) const
    {
        k[d] = key;
        v[d] = val;
    }
};

// KV12: key and suffix of one element side by side, 12 bytes (4-byte aligned).  The passes BETWEEN the first and the last pass
// of a 64-bit-key sort move their elements in this form: a bucket's share of a tile then leaves as ONE run of 12 x count bytes
// instead of an 8 x count-byte key run and a 4 x count-byte value run.  What a pass pays for on the store side is the partial
// 64-byte block at either end of every run (~46 ps against ~14 ps for a whole block), and one run has two
// ends where two runs have four: a 12288-element tile over 256 buckets writes 576-byte runs (8 whole + 2 partial blocks: 204 ps
// per 48 elements) instead of 384 + 192 bytes (7 whole + 4 partial: 282 ps).
struct KV12 { uint32_t lo, hi, v; };
static_assert(sizeof(KV12) == 12 && alignof(KV12) == 4, "three words, word-aligned");
struct SrcKV12 {
    static constexpr bool kHasVal = true;
    static constexpr bool kFromText = false;
    const KV12* in;
    __device__ __forceinline__ uint64_t key(uint64_t i) const { return ((uint64_t)in[i].hi << 32) | in[i].lo; }
    __device__ __forceinline__ uint32_t val(uint64_t i) const { return in[i].v; }
    // key and value of element i with ONE 12-byte load (a wave's 64 consecutive elements: 768 contiguous bytes per instruction)
    __device__ __forceinline__ void fetch(uint64_t i, uint64_t& k, uint32_t& v) const
    {
        const KV12 e = in[i];
        k = ((uint64_t)e.hi << 32) | e.lo;
        v = e.v;
    }
};
struct DstKV12 {
    KV12* out;
    __device__ __forceinline__ void store(uint32_t d, uint64_t key, uint32_t val) const
    {
        KV12 e;
        e.lo = (uint32_t)key;
        e.hi = (uint32_t)(key >> 32);
        e.v = val;
        out[d] = e;
    }
};

// element i of a source: its own one-load form where it has one (SrcKV12), key() + val() otherwise
template <class Src, class = void> struct SrcHasFetch { static constexpr bool value = false; };
template <class Src> struct SrcHasFetch<Src, decltype((void)&Src::fetch)> { static constexpr bool value = true; };
template <class Src>
__device__ __forceinline__ void src_fetch(const Src& src, uint64_t i, uint64_t& k, uint32_t& v)
{
    if constexpr (SrcHasFetch<Src>::value) {
        src.fetch(i, k, v);
    } else {
        k = src.key(i);
        v = Src::kHasVal ? src.val(i) : 0u;
    }
}

__device__ __forceinline__ unsigned digit_of(uint64_t key, int shift, unsigned mask)
{
    return (unsigned)(key >> shift) & mask;
}

// ---- histograms ------------------------------------------------------------------------
// chunked schedule: digit counts of one pass for this workgroup's chunk -> column of
// the [256][workgroups] matrix.
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_radix_hist_chunk(Src src, uint64_t m, int shift, unsigned mask, uint64_t chunk,
                   uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[kWavesPerBlock][kRadix];
    const unsigned tid = threadIdx.x, w = wave_id();
    for (unsigned i = tid; i < kWavesPerBlock * kRadix; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    uint64_t begin = (uint64_t)blockIdx.x * chunk;
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    // 8 independent loads per thread in flight before the first histogram update
    uint64_t i = begin + tid;
    for (; i + 7 * kBlock < end; i += 8 * kBlock) {
        uint64_t k[8];
#pragma unroll
        for (int j = 0; j < 8; j++) k[j] = src.key(i + (uint64_t)j * kBlock);
#pragma unroll
        for (int j = 0; j < 8; j++) atomicAdd(&h[w][digit_of(k[j], shift, mask)], 1u);
    }
    for (; i < end; i += kBlock) atomicAdd(&h[w][digit_of(src.key(i), shift, mask)], 1u);
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; k++) c += h[k][tid];
    hist[(uint64_t)tid * gridDim.x + blockIdx.x] = c;
}

// one-sweep schedule: digit counts of ALL passes in one read of the keys.
// partial[(pass*256 + digit) * workgroups + workgroup]
template <class Src>
__global__ void __launch_bounds__(kBlock)
k_radix_hist_all(Src src, uint64_t m, int bit_lo, int bit_hi, int npass, uint64_t chunk,
                 uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[kWavesPerBlock][kMaxPasses][kRadix];     // 32 KiB
    const unsigned tid = threadIdx.x, w = wave_id();
    for (unsigned i = tid; i < kWavesPerBlock * kMaxPasses * kRadix; i += kBlock) (&h[0][0][0])[i] = 0;
    __syncthreads();
    uint64_t begin = (uint64_t)blockIdx.x * chunk;
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint64_t i = begin + tid;
    // (whole waves only: the loop condition is the same for all 64 lanes, the wave-wide test below needs it)
    for (; i - lane_id() + (kWave - 1) + 3 * kBlock < end; i += 4 * kBlock) {
        uint64_t k[4];
#pragma unroll
        for (int j = 0; j < 4; j++) k[j] = src.key(i + (uint64_t)j * kBlock);
        for (int p = 0; p < npass; p++) {
            const int shift = bit_lo + p * kRadixBits;
            const int nb = bit_hi - shift < kRadixBits ? bit_hi - shift : kRadixBits;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // the high digits of sorted or bucket-ordered keys are the same across a wave:
                // one add of 64 instead of 64 atomics on one LDS address
                const unsigned d = digit_of(k[j], shift, (1u << nb) - 1u);
                if (!Src::kFromText && __all(d == __shfl(d, 0))) {        // (text-fed keys are not ordered: no test)
                    if (lane_id() == 0) atomicAdd(&h[w][p][d], (uint32_t)kWave);
                } else {
                    atomicAdd(&h[w][p][d], 1u);
                }
            }
        }
    }
    for (; i < end; i += kBlock) {
        const uint64_t key = src.key(i);
        for (int p = 0; p < npass; p++) {
            const int shift = bit_lo + p * kRadixBits;
            const int nb = bit_hi - shift < kRadixBits ? bit_hi - shift : kRadixBits;
            atomicAdd(&h[w][p][digit_of(key, shift, (1u << nb) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int p = 0; p < npass; p++) {
        uint32_t c = 0;
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; k++) c += h[k][p][tid];
        partial[((uint64_t)p * kRadix + tid) * gridDim.x + blockIdx.x] = c;
    }
}

// Text-fed initial sort with 8 % bits == 0 and 32-bit keys: digit p of the key of suffix i is
// the 8-bit window of the packed symbol stream at symbol (i + shift_p), shift_p = (3 - p) *
// (8 / bits).  So all four digit histograms are ONE histogram W of the 8-bit windows at
// symbols [0, n + 3 * 8/bits), shifted: H_p = W minus the <= 3*8/bits windows before
// shift_p and the same number past n + shift_p (k_window_fix).  One LDS atomic per position
// instead of four, and 16 windows come out of two packed words.
__global__ void __launch_bounds__(kBlock)
k_window_hist(PackedText t, uint64_t npos, uint64_t words_per_block, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[kWavesPerBlock][kRadix];
    const unsigned tid = threadIdx.x, w = wave_id();
    for (unsigned i = tid; i < kWavesPerBlock * kRadix; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint64_t nwords = (npos + (uint64_t)t.spw - 1) / (uint64_t)t.spw;
    const uint64_t qb = (uint64_t)blockIdx.x * words_per_block;
    const uint64_t qe = dmin<uint64_t>(nwords, qb + words_per_block);
    for (uint64_t q = qb + tid; q < qe; q += kBlock) {
        const uint64_t comb = ((uint64_t)t.words[q] << 32) | (uint64_t)t.words[q + 1];
        const uint64_t j0 = q * (uint64_t)t.spw;
        for (int o = 0; o < t.spw; o++)
            if (j0 + (uint64_t)o < npos) atomicAdd(&h[w][(unsigned)(comb >> (56 - o * t.bits)) & 255u], 1u);
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; k++) c += h[k][tid];
    partial[(uint64_t)tid * gridDim.x + blockIdx.x] = c;
}
// totals[0..256) = W on entry; totals[p*256 + d] = H_p[d] on exit (p = 0 is the lowest digit)
__global__ void __launch_bounds__(kBlock)
k_window_fix(PackedText t, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t sub[4][kRadix];
    const unsigned tid = threadIdx.x;
    const uint32_t wcount = totals[tid];
    for (int p = 0; p < 4; p++) sub[p][tid] = 0;
    __syncthreads();
    if (tid == 0) {
        const int spd = 8 / t.bits;
        auto window = [&](uint64_t j) -> unsigned {
            const uint64_t q = j / (uint64_t)t.spw;
            const int o = (int)(j - q * (uint64_t)t.spw);
            const uint64_t comb = ((uint64_t)t.words[q] << 32) | (uint64_t)t.words[q + 1];
            return (unsigned)(comb >> (56 - o * t.bits)) & 255u;
        };
        for (int p = 0; p < 4; p++) {
            const uint64_t shift = (uint64_t)(3 - p) * spd;
            for (uint64_t j = 0; j < shift; j++) sub[p][window(j)]++;
            for (uint64_t j = t.n + shift; j < t.n + 3ull * spd; j++) sub[p][window(j)]++;
        }
    }
    __syncthreads();
    for (int p = 0; p < 4; p++) totals[p * kRadix + tid] = wcount - sub[p][tid];
}

// The same W from the hybrid route's histogram of 16-bit key prefixes when that route gives way: totals[256 + x] holds
// the count of the 8-bit windows x at the positions [0, n); the 3 * 8/bits positions past the end see only padding.
__global__ void __launch_bounds__(kBlock)
k_window_from_hist16(uint32_t* __restrict__ totals, uint32_t tail_positions)
{
    totals[threadIdx.x] = totals[kRadix + threadIdx.x] + (threadIdx.x == 0 ? tail_positions : 0u);
}

// one workgroup per row: exclusive scan of the row in place, row total -> row_total.
__global__ void __launch_bounds__(kBlock)
k_radix_scan(uint32_t* __restrict__ hist, unsigned nblocks, uint32_t* __restrict__ row_total)
{
    __shared__ uint32_t part[kWavesPerBlock];
    uint32_t* row = hist + (uint64_t)blockIdx.x * nblocks;
    uint32_t carry = 0;
    for (unsigned base = 0; base < nblocks; base += kBlock) {
        unsigned i = base + threadIdx.x;
        uint32_t v = (i < nblocks) ? row[i] : 0u;
        uint32_t total;
        uint32_t ex = block_scan_add_excl(v, part, total);
        if (i < nblocks) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

// ---- tile engine ---------------------------------------------------------------------
// The match masks of the ranking (NW x 256 u64) live in the first 2 KiB x NW of `stage`:
// ranking is over before anything is staged, and every wave clears its own row before it ranks.
template <int KPT, bool HAS_VAL, int NW, bool RANK_ATOMIC>
struct RadixSmem {
    uint32_t cnt[NW][kRadix];                           // per-wave digit counts, then tile-local bases
    uint32_t off[kRadix];                               // global bucket head minus tile-local bucket start
    uint32_t part[2][NW];
    uint32_t ticket;
    uint64_t stage[NW * kWave * KPT];
    uint32_t stage_v[HAS_VAL ? NW * kWave * KPT : 1];
};

// decoupled look-back for digit `tid` of tile `tile_no`: publishes the tile's count,
// returns the number of elements with this digit in all earlier tiles.
#ifndef SFX_LOOKAHEAD
#define SFX_LOOKAHEAD 4
#endif
constexpr int kLookAhead = SFX_LOOKAHEAD;
struct LookBack {
    uint32_t sv[kLookAhead];     // status words of the kLookAhead nearest predecessors, in flight
};
// Publish this tile's count for digit `tid` and START the look-back: the status words of the
// next kLookAhead predecessors are fetched together (independent uncached loads) and not
// waited for -- the caller does its LDS staging in between.
// `first`: the tile opens its sequence (tile 0 of a sort; the first tile of a segment of a segmented
// sort): it publishes a prefix at once, which is also where the look-back of its successors ends.
__device__ __forceinline__ void lookback_begin(uint32_t* status, uint32_t tile_no, unsigned tid, uint32_t agg, LookBack& lb,
                                               bool first)
{
    uint32_t* mine = status + (uint64_t)tile_no * kRadix + tid;
    __hip_atomic_store(mine, (first ? kStatusPrefix : kStatusAgg) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < kLookAhead; u++) {
        const int64_t t = (int64_t)tile_no - 1 - u;
        lb.sv[u] = t >= 0 ? __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : kStatusPrefix;                                    // before tile 0: empty prefix
    }
}
// Finish: number of elements with this digit in all earlier tiles; publishes the inclusive prefix.
__device__ __forceinline__ uint32_t lookback_finish(uint32_t* status, uint32_t tile_no, unsigned tid, uint32_t agg, LookBack& lb,
                                                    bool first)
{
    if (first) return 0u;
    uint32_t* mine = status + (uint64_t)tile_no * kRadix + tid;
    uint32_t excl = 0;
    int64_t j = (int64_t)tile_no - 1;
    for (;;) {
#pragma unroll
        for (int u = 0; u < kLookAhead; u++) {
            const int64_t t = j - u;
            while ((lb.sv[u] >> 30) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                lb.sv[u] = __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            excl += lb.sv[u] & kStatusValue;
            if ((lb.sv[u] >> 30) == 2u) {
                __hip_atomic_store(mine, kStatusPrefix | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return excl;
            }
        }
        j -= kLookAhead;
#pragma unroll
        for (int u = 0; u < kLookAhead; u++) {
            const int64_t t = j - u;
            lb.sv[u] = t >= 0 ? __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : kStatusPrefix;
        }
    }
}

// NW = waves per workgroup (4 or 8): thread d < 256 owns bucket d; a 512-thread workgroup
// sorts 8192-element tiles, i.e. 256-byte runs per bucket.
#ifndef SFX_RADIX_MIN_WAVES
#define SFX_RADIX_MIN_WAVES 1
#endif
// SEG (segmented one-sweep, refinement rounds): the elements of many independent segments (the large
// buckets of the active list) are sorted segment by segment in ONE launch.  Tiles come from a table
// (a segment is cut into consecutive tiles, none straddles two segments); the head of digit d is
// seg_start + (digit offsets inside the segment) + (look-back over the EARLIER TILES OF THE SAME SEGMENT).
// A segment of one tile needs neither table nor look-back: its digit offsets are the tile's own scan.
struct SegTile {
    uint32_t begin, count;      // list positions [begin, begin + count)
    uint32_t seg_start;         // list position of the segment's first element
    uint32_t info;              // bit 0: first tile of its segment, bit 1: the segment's only tile; bits 2..: dense index
                                // among the tiles of multi-tile segments (status words, per-tile digit counts)
    uint32_t mseg;              // multi-tile segments: index of the segment's digit-offset table
    uint32_t pad[3];
};
struct SegArgs {
    const SegTile* tiles;
    const uint32_t* ntiles;     // device-side count (made by k_seg_layout: no host round trip)
    const uint32_t* segexcl;    // [mseg][npass][256]: elements of the segment with a smaller digit, per pass
    int pass, npass;
};

template <class Src, class Dst, int KPT, bool ONESWEEP, bool RANK_ATOMIC, int NW, bool SEG = false>
__global__ void __launch_bounds__(NW * kWave, SFX_RADIX_MIN_WAVES)
k_radix_pass(Src src, Dst dst, uint64_t m, int shift, unsigned mask, uint64_t chunk,
             const uint32_t* __restrict__ hist, const uint32_t* __restrict__ digit_total,
             uint32_t* __restrict__ status, uint32_t* __restrict__ ticket, SegArgs seg = SegArgs{nullptr, nullptr, nullptr, 0, 0})
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    __shared__ RadixSmem<KPT, HAS_VAL, NW, RANK_ATOMIC> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;                   // thread d owns bucket d
    unsigned par = 0;

    static_assert(kWave * KPT >= kRadix, "the match masks must fit the staging buffer");
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0u;
    }
    // one-sweep: global start of bucket `tid`; chunked: this workgroup's running head of bucket `tid`
    // (segmented: set per tile)
    uint32_t my_head = SEG ? 0u : block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    static_assert(!SEG || ONESWEEP, "segments are handed out by ticket");
    const uint32_t seg_ntiles = SEG ? *seg.ntiles : 0u;
    if (!ONESWEEP && owner) my_head += hist[(uint64_t)tid * gridDim.x + blockIdx.x];

    uint64_t next = (uint64_t)blockIdx.x * chunk;
    uint64_t limit = m;
    if (!ONESWEEP) limit = dmin<uint64_t>(m, next + chunk);
    __syncthreads();

    // load: wave-striped, 64 consecutive elements per round; padding sorts last in the tile.
    // Chunked schedule: the NEXT tile's loads are issued before this tile is processed, so a
    // workgroup always has a tile of HBM reads in flight (bandwidth = bytes in flight / latency).
    uint64_t nkey[KPT];
    uint32_t nval[HAS_VAL ? KPT : 1];
    auto load_tile = [&](uint64_t tile) {
        const unsigned nv = (unsigned)dmin<uint64_t>(kTile, limit - tile);
        unsigned first = w * (kWave * KPT) + lane;                  // (opaque: see k_partition)
        SFX_OPAQUE_VGPR(first);
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = first + r * kWave;
            uint64_t k = ~0ull;
            uint32_t v = 0u;
            if (idx < nv) src_fetch(src, tile + idx, k, v);
            nkey[r] = k;
            if (HAS_VAL) nval[r] = v;
        }
    };
    if (!ONESWEEP && next < limit) load_tile(next);

    for (;;) {
        uint64_t tile;
        uint32_t tile_no = 0;
        bool seq_first = false, seg_single = false;
        uint32_t seg_start = 0;
        if (ONESWEEP) {
            if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
            __syncthreads();
            tile_no = s.ticket;
            tile = (uint64_t)tile_no * kTile;
            seq_first = tile_no == 0;
            if (SEG) {
                if (tile_no >= seg_ntiles) break;
                const SegTile d = seg.tiles[tile_no];
                tile = d.begin;
                limit = (uint64_t)d.begin + d.count;
                seq_first = d.info & 1u;
                seg_single = d.info & 2u;
                seg_start = d.seg_start;
                if (owner && !seg_single) my_head = d.seg_start + seg.segexcl[((uint64_t)d.mseg * seg.npass + seg.pass) * kRadix + tid];
                tile_no = d.info >> 2;                              // status words are indexed by the dense multi-tile index
            }
        } else {
            tile = next;
            next += kTile;
        }
        if (tile >= limit) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, limit - tile);
        if (ONESWEEP) load_tile(tile);

        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            key[r] = nkey[r];
            if (HAS_VAL) val[r] = nval[r];
        }
        if (!ONESWEEP && next < limit) load_tile(next);
        if (RANK_ATOMIC) {                                  // this wave's match masks (aliased onto the stage)
#pragma unroll
            for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
            wave_sync();
        }
        // (segmented: a bucket rarely fills its last tile -- rounds that hold nothing but padding are not ranked)
        const unsigned ranked = SEG ? (nvalid / (kWave * KPT)) * (kWave * KPT) + ((nvalid % (kWave * KPT) + kWave - 1) / kWave) * kWave
                                    : (unsigned)kTile;
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            if (!SEG || w * (kWave * KPT) + r * kWave < nvalid)
                pos[r] = rank_round<RANK_ATOMIC>(digit_of(key[r], shift, mask), my_flags, s.cnt[w], mybit);
            else
                pos[r] = 0;
        }
        __syncthreads();

        // thread d: bucket d's size in this tile -> tile-local start, per-wave bases, global head
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
        {
            uint32_t c[NW], tile_count = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                c[k] = owner ? s.cnt[k][tid] : 0u;
                tile_count += c[k];
            }
            const uint32_t ex = block_scan_excl_1b<NW>(tile_count, s.part, par);
            if (owner) {
                uint32_t run = ex;
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    s.cnt[k][tid] = run;
                    run += c[k];
                }
                // padding elements all carry the largest digit (== mask)
                real_count = tile_count - ((tid == mask) ? (uint32_t)(ranked - nvalid) : 0u);
                tile_ex = ex;
                if (ONESWEEP) {
                    if (!(SEG && seg_single))
                        lookback_begin(status, tile_no, tid, real_count, lb, seq_first);     // loads in flight during the staging
                } else {
                    s.off[tid] = my_head - ex;
                    my_head += real_count;
                }
            }
        }
        __syncthreads();

        // reorder through LDS: every bucket's elements become one contiguous run
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            if (SEG && w * (kWave * KPT) + r * kWave >= nvalid) continue;
            const unsigned p = pos[r] + s.cnt[w][digit_of(key[r], shift, mask)];
            s.stage[p] = key[r];
            if (HAS_VAL) s.stage_v[p] = val[r];
        }
        if (ONESWEEP && owner) {
            if (SEG && seg_single) s.off[tid] = seg_start;            // digit offsets = the tile's own scan
            else s.off[tid] = my_head + lookback_finish(status, tile_no, tid, real_count, lb, seq_first) - tile_ex;
        }
        __syncthreads();
        // batches of 8 slots: within a batch all LDS reads of a kind are in flight together;
        // more than 8 at once only costs registers (16-element threads spilled)
        constexpr int kOut = (KPT % 8 == 0) ? 8 : ((KPT % 4 == 0) ? 4 : KPT);
        unsigned t = tid;                                           // (opaque: see k_partition)
        SFX_OPAQUE_VGPR(t);
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) {
                key[r] = s.stage[r * kThreads + t];
                if (HAS_VAL) val[r] = s.stage_v[r * kThreads + t];
            }
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) pos[r] = s.off[digit_of(key[r], shift, mask)] + (r * kThreads + t);
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++)
                if ((unsigned)(r * kThreads) + t < nvalid) dst.store(pos[r], key[r], HAS_VAL ? val[r] : 0u);
        }
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0u;
        }
        __syncthreads();
    }
}

// ---- the one-sweep pass of the dense sorts, round 4 ------------------------------------------------------------------------
// Same tile engine and look-back protocol as k_radix_pass<..., ONESWEEP = true>, without what the chunked and the segmented
// schedules need (no second key array, no per-tile descriptors) and with 16-bit counts: a tile is < 65536 elements, per-wave
// counts and tile-local bucket starts fit 16 bits, and the 8 KB that saves -- together with the registers the leaner loop
// frees -- let an E64 tile grow from 11 to 16 elements per thread (16384 elements: 512-byte runs per bucket, a third fewer
// look-backs) and a KV tile from 9 to 10.  lab/radix_lab2.hip, 100 M elements, ms per pass: E64 11 / 12 / 14 / 16 / 17 per
// thread = 0.471 / 0.449 / 0.441 / 0.423 / 0.451 (17 spills); KV 9 / 10 / 11 / 12 = 0.616 / 0.595 / 0.602 / 0.70 (round 3's
// kernel at 9: 0.70).  What the lab's phase timers say bounds a pass (profiles/r4_radix_lab.txt): ranking 25 %, waiting for
// the look-back 26 % (it grows with the number of workgroups in flight), waiting for the tile's loads 16 %, scan 12 %; the
// stores are asynchronous and cost what the memory side charges for the two partial 64-byte blocks at the ends of every run
// (~50 ps each, whoever completes the block later; the lane -> address mapping does not matter: lab/store_probe.hip).
template <int KPT, bool HAS_VAL, int NW>
struct SweepSmem {
    uint64_t stage[NW * kWave * KPT];                   // (the match masks of the ranking alias its first NW x 2 KiB)
    uint32_t stage_v[HAS_VAL ? NW * kWave * KPT : 1];
    uint16_t cnt[NW][kRadix];                           // per-wave digit counts, then per-wave tile-local bases
    uint32_t off[kRadix];                               // global bucket head minus tile-local bucket start
    uint32_t part[2][NW];
    uint32_t ticket;
};
template <class Src, class Dst, int KPT, int NW>
__global__ void __launch_bounds__(NW * kWave, 1)
k_radix_sweep(Src src, Dst dst, uint64_t m, int shift, unsigned mask, const uint32_t* __restrict__ digit_total,
              uint32_t* __restrict__ status, uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    static_assert(kWave * KPT >= kRadix, "the match masks must fit the staging buffer");
    static_assert(kTile < 65536, "16-bit tile positions");
    __shared__ SweepSmem<KPT, HAS_VAL, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;                   // thread d owns bucket d
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);   // global start of bucket `tid`
    __syncthreads();
    for (;;) {
        if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile_no = s.ticket;
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        // load: wave-striped, 64 consecutive elements per round; padding sorts last in the tile
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos[KPT];
        unsigned first = w * (kWave * KPT) + lane;                  // (opaque, like t below)
        SFX_OPAQUE_VGPR(first);
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = first + r * kWave;
            uint64_t k = ~0ull;
            uint32_t v = 0u;
            if (idx < nvalid) src_fetch(src, tile + idx, k, v);
            key[r] = k;
            if (HAS_VAL) val[r] = v;
        }
#pragma unroll
        for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] = rank_round16(digit_of(key[r], shift, mask), my_flags, s.cnt[w], mybit);
        __syncthreads();
        // thread d: bucket d's size in this tile -> tile-local start, per-wave bases, look-back
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
        {
            uint32_t c[NW], tile_count = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                c[k] = owner ? s.cnt[k][tid] : 0u;
                tile_count += c[k];
            }
            const uint32_t ex = block_scan_excl_1b<NW>(tile_count, s.part, par);
            if (owner) {
                uint32_t run = ex;
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                // padding elements all carry the largest digit (== mask)
                real_count = tile_count - ((tid == mask) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);       // loads in flight during the staging
            }
        }
        __syncthreads();
        // reorder through LDS: every bucket's elements become one contiguous run
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned p = pos[r] + s.cnt[w][digit_of(key[r], shift, mask)];
            s.stage[p] = key[r];
            if (HAS_VAL) s.stage_v[p] = val[r];
        }
        if (owner) s.off[tid] = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0) - tile_ex;
        __syncthreads();
        // small batches: all LDS reads of a batch are in flight together; batches of 1 .. KPT tie as long as nothing spills
        constexpr int kOut = (KPT % 4 == 0) ? 4 : ((KPT % 3 == 0) ? 3 : ((KPT % 2 == 0) ? 2 : 1));
        // (the thread index, made opaque: the compiler otherwise computes the LDS addresses of this loop once, before the tile
        // loop, has no registers to keep them in and reloads them from scratch in every tile -- see k_partition)
        unsigned t = tid;
        SFX_OPAQUE_VGPR(t);
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) {
                key[r] = s.stage[r * kThreads + t];
                if (HAS_VAL) val[r] = s.stage_v[r * kThreads + t];
            }
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) pos[r] = s.off[digit_of(key[r], shift, mask)] + (r * kThreads + t);
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++)
                if ((unsigned)(r * kThreads) + t < nvalid) dst.store(pos[r], key[r], HAS_VAL ? val[r] : 0u);
        }
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
    }
}

// ---- the one-sweep pass with TWO workgroups per CU (round 5) ------------------------------------------------------------------
// k_radix_sweep keeps one 12288-element KV tile per CU: its elements in registers, its staging copy in 147 KB of LDS.  One
// workgroup per CU means nothing overlaps a tile's phases -- while it ranks, scans and waits for its look-back the CU asks the
// memory system for nothing (a tile takes 18.8 us, of which its 295 KB need 12 at the CU's fair share of HBM: DESIGN.md section
// 9).  Here a tile is held by 512 threads and staged through LDS in two halves of sorted places, so that a workgroup needs half
// the LDS of its tile and TWO of them share a CU: one loads and ranks while the other writes.  Registers: 16 waves per CU
// either way, 128 per thread.  Same tickets, same look-back, same stable ranking order (wave, round, lane) as k_radix_sweep;
// only the elements' way out differs: a sorted place p of the tile leaves in half p / (tile / 2), and a run that straddles the
// middle is written in two pieces.
// Measured (profiles/r5_duo_ab.jsonl, the eight KV passes of config 3): k_radix_sweep 47.1 ms; two workgroups of 512 threads x 14
// elements (7168-element tiles, 336-byte runs, 128 registers, no scratch) 44.9; x 16 (8192-element tiles: 44 bytes of scratch)
// 50.0.  The whole 12288-element tile (24 per thread) needs 236 bytes of scratch under the 128 registers that two workgroups
// per CU leave a thread, 10- and 12-wave workgroups spill at 14 per thread: 14 x 8 is the geometry.
constexpr int kDuoKPT = 14, kDuoNW = 8;
constexpr int kDuoKPTE64 = 16;                                     // 8-byte elements: 8192-element tiles (18 and 20 per thread spill)
template <int KPT, int NW, bool HAS_VAL>
struct DuoSmem {
    static constexpr int kHalf = NW * kWave * KPT / 2;
    uint64_t stage[kHalf];                              // (the match masks of the ranking alias its first NW x 2 KiB)
    uint32_t stage_v[HAS_VAL ? kHalf : 1];
    uint16_t cnt[NW][kRadix];
    uint32_t off[kRadix];
    uint32_t part[2][NW];
    uint32_t ticket;
};
template <class Src, class Dst, int KPT, int NW>
__global__ void __launch_bounds__(NW * kWave) SFX_WAVES_PER_EU(NW / 2, NW / 2)      // (two workgroups per CU: 2 NW / 4 waves per SIMD)
k_radix_sweep_duo(Src src, Dst dst, uint64_t m, int shift, unsigned mask, const uint32_t* __restrict__ digit_total,
                  uint32_t* __restrict__ status, uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    constexpr int kHalf = kTile / 2;
    static_assert(kThreads >= kRadix && KPT % 2 == 0 && kHalf % kThreads == 0, "an owner thread per bucket; whole output rounds per half");
    static_assert(kHalf * 8 >= NW * kRadix * 8, "the match masks must fit the staging buffer");
    static_assert(kTile < 65536, "16-bit tile positions");
    __shared__ DuoSmem<KPT, NW, HAS_VAL> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    __syncthreads();
    for (;;) {
        if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile_no = SFX_WAVE_UNIFORM(s.ticket);
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos2[KPT / 2];                              // sorted places, two 16-bit values per register
        unsigned first = w * (kWave * KPT) + lane;
        SFX_OPAQUE_VGPR(first);
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = first + r * kWave;
            uint64_t k = ~0ull;
            uint32_t v = 0u;
            if (idx < nvalid) src_fetch(src, tile + idx, k, v);
            key[r] = k;
            if (HAS_VAL) val[r] = v;
        }
#pragma unroll
        for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r += 2) {
            const uint32_t a = rank_round16(digit_of(key[r], shift, mask), my_flags, s.cnt[w], mybit);
            const uint32_t b = rank_round16(digit_of(key[r + 1], shift, mask), my_flags, s.cnt[w], mybit);
            pos2[r / 2] = a | (b << 16);
        }
        __syncthreads();
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
        {
            // (the bucket's column of the count table through an opaque index: its NW addresses are otherwise computed once,
            // before the tile loop, and kept -- in scratch, for want of registers)
            unsigned col = tid;
            SFX_OPAQUE_VGPR(col);
            uint32_t c[NW], tile_count = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                c[k] = owner ? s.cnt[k][col] : 0u;
                tile_count += c[k];
            }
            const uint32_t ex = block_scan_excl_1b<NW>(tile_count, s.part, par);
            if (owner) {
                uint32_t run = ex;
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    s.cnt[k][col] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == mask) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
        }
        __syncthreads();
        // sorted place of every element in the tile (16 bits), then the two halves through LDS
#pragma unroll
        for (int r = 0; r < KPT; r += 2)
            pos2[r / 2] += (uint32_t)s.cnt[w][digit_of(key[r], shift, mask)] | ((uint32_t)s.cnt[w][digit_of(key[r + 1], shift, mask)] << 16);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            __syncthreads();                                    // (h = 0: every thread has read its bases; h = 1: the first half has left)
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned p = (r & 1) ? (pos2[r / 2] >> 16) : (pos2[r / 2] & 0xFFFFu);
                if ((p >= (unsigned)kHalf) == (h == 1)) {
                    s.stage[p - (unsigned)(h * kHalf)] = key[r];
                    if (HAS_VAL) s.stage_v[p - (unsigned)(h * kHalf)] = val[r];
                }
            }
            // (the look-back's loads were in flight during the staging of the first half)
            if (h == 0 && owner) s.off[tid] = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0) - tile_ex;
            __syncthreads();
            unsigned t = tid;
            SFX_OPAQUE_VGPR(t);
            constexpr int kRounds = kHalf / kThreads;               // 12
            constexpr int kOut = (kRounds % 4 == 0) ? 4 : ((kRounds % 3 == 0) ? 3 : ((kRounds % 2 == 0) ? 2 : 1));
#pragma unroll
            for (int j0 = 0; j0 < kRounds; j0 += kOut) {
                uint64_t ok[kOut];
                uint32_t ov[kOut], od[kOut];
#pragma unroll
                for (int j = 0; j < kOut; j++) {
                    ok[j] = s.stage[(j0 + j) * kThreads + t];
                    ov[j] = HAS_VAL ? s.stage_v[(j0 + j) * kThreads + t] : 0u;
                }
#pragma unroll
                for (int j = 0; j < kOut; j++) od[j] = s.off[digit_of(ok[j], shift, mask)] + (unsigned)(h * kHalf + (j0 + j) * kThreads) + t;
#pragma unroll
                for (int j = 0; j < kOut; j++)
                    if ((unsigned)(h * kHalf + (j0 + j) * kThreads) + t < nvalid) dst.store(od[j], ok[j], ov[j]);
            }
        }
        if (owner) {
            unsigned col = tid;
            SFX_OPAQUE_VGPR(col);
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][col] = 0;
        }
        __syncthreads();
    }
}

// ---- partition passes: when the order inside a bucket does not matter ------------------------------------------------------
// The two device-wide passes of the hybrid route only have to bring the elements of every sub-bucket (top 16 key bits)
// together: the LDS sort that follows orders each sub-bucket by the whole 64-bit element, whatever order it arrives in.
// Without stability a pass needs neither tickets nor a look-back nor the match-mask ranking (half of a one-sweep tile's time:
// DESIGN.md section 9): an element's place inside its tile's bucket is the return value of ONE LDS atomic (all of a thread's
// atomics in flight together), and the tile's run of bucket d goes wherever a returning global atomic on bucket d's cursor
// says -- the cursors start at the bucket starts, which the 65536-bin histogram of the route has.  MSD order, so that both
// passes are free: the first splits by the top 8 bits (256 cursors, one per 128-byte line: a tile's 256 atomics would
// otherwise queue up on eight lines), the second splits every top-8 bucket by the next 8 bits (cursor = the sub-bucket's own
// start: 65536 of them, a few dozen atomics each).  Tiles of the second pass never straddle two top-8 buckets: tile v of the
// pass is tile (v - first tile of b) of bucket b, found by bisecting the running tile counts.
constexpr unsigned kCursorPad = 16;                                  // words between two cursors of the first pass
constexpr unsigned kPartClasses = 8;                                 // stretches of the input of the first pass = XCDs
#ifndef SFX_PART_ABL
#define SFX_PART_ABL 0                                               // lab/partition_lab.hip: 1 no LDS atomics, 2 no global atomics, 4 no stores, 8 no loads
#endif
template <int KPT, int NW>
struct PartSmem {
    uint64_t stage[NW * kWave * KPT];                                // (the match masks of the ranking alias its first NW x 2 KiB)
    uint16_t cnt[NW][kRadix];                                        // per-wave digit counts, then per-wave tile-local bases
    uint32_t off[kRadix];                                            // global start of the tile's run of bucket d minus its tile-local start
    uint32_t tiles_before[kRadix];                                   // SUB: tiles of the top-8 buckets of b's class (b mod 8) before b
    uint32_t class_tiles[8];                                         // SUB: tiles of each class
    uint32_t part[2][NW];
};
// SUB = false: the elements src.key(0 .. m), bucket = bits [shift, shift + 8); the input is cut into kPartClasses stretches of
//              class_len elements, tile v = 8 u + x is tile u of stretch x, cursor[(x * 256 + d) * kCursorPad] -- as in the second
//              pass, the workgroups of one XCD then append to cursors of their own, and the seams of their runs meet in its L2.
// SUB = true:  src = the output of the first pass; bstart16[b << 8] = start of top-8 bucket b in it (bstart16[65536] = m);
//              bucket = bits [shift, shift + 8) inside top-8 bucket b, cursor[(b << 8) | d].
// (Ranking: the match masks of the one-sweep pass.  One returning LDS atomic per element was the first version -- the LDS
// retires about one of them per clock and CU: 8.4 us per 16384-element tile against 4.6 for the masks, lab/partition_lab.hip.)
template <class Src, int KPT, int NW, bool SUB>
__global__ void __launch_bounds__(NW * kWave) SFX_WAVES_PER_EU(4, 4)      // (16 waves per CU: one workgroup of 16 or two of 8)
k_partition(Src src, uint64_t* __restrict__ out, uint64_t m, int shift, uint32_t* __restrict__ cursor,
            const uint32_t* __restrict__ bstart16, uint64_t class_len)
{
    constexpr int kThreads = NW * kWave;
    constexpr uint32_t kTile = kThreads * KPT;
    static_assert(kThreads >= kRadix, "thread d owns bucket d");
    static_assert(kWave * KPT >= kRadix, "the match masks must fit the staging buffer");
    static_assert(kTile < 65536, "16-bit tile positions");
    __shared__ PartSmem<KPT, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    uint64_t ntiles;
    {
        // (the longest stretch is the first; the last may be short or empty)
        const uint64_t first = dmin<uint64_t>(m, class_len);
        ntiles = ((first + kTile - 1) / kTile) * kPartClasses;
    }
    if (SUB) {
        // Tile v = 8 u + x is tile u of class x = the top-8 buckets b with b mod 8 == x, in order.  Workgroup j takes v = j,
        // j + gridDim, ...: with a grid that is a multiple of 8 all its tiles are of class j mod 8 -- and workgroups are dealt to
        // the 8 XCDs round robin, so all tiles of a top-8 bucket are sorted on ONE XCD: the runs that consecutive tiles append to
        // a sub-bucket meet in that XCD's L2, and the 64-byte blocks at their seams leave it whole (the store model of DESIGN.md
        // section 9: a partial block costs three whole ones on the memory side).
        if (owner) s.tiles_before[tid] = (bstart16[(tid + 1u) << 8] - bstart16[tid << 8] + kTile - 1u) / kTile;
        __syncthreads();
        if (tid < 8u) {
            uint32_t run = 0;
            for (unsigned b = tid; b < (unsigned)kRadix; b += 8u) {
                const uint32_t tb = s.tiles_before[b];
                s.tiles_before[b] = run;
                run += tb;
            }
            s.class_tiles[tid] = run;
        }
        __syncthreads();
        uint32_t most = 0;
#pragma unroll
        for (int x = 0; x < 8; x++) most = dmax(most, s.class_tiles[x]);
        ntiles = (uint64_t)most * 8u;
    }
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    __syncthreads();
    // where tile v lies: its first element, its length, its top-8 bucket (SUB)
    auto locate = [&](uint64_t v, uint64_t& begin, unsigned& nvalid, unsigned& top) {
        begin = v * kTile;
        nvalid = 0;
        top = 0;
        if (v >= ntiles) return;
        if (SUB) {
            const unsigned x = (unsigned)(v & 7u);
            const uint32_t u = (uint32_t)(v >> 3);
            if (u >= s.class_tiles[x]) return;                       // (a class with fewer tiles than the largest one)
            unsigned lo = 0, hi = kRadix / 8;                        // largest i with tiles_before[x + 8 i] <= u (uniform)
            while (hi - lo > 1u) {
                const unsigned mid = (lo + hi) / 2u;
                if (s.tiles_before[x + 8u * mid] <= u) lo = mid; else hi = mid;
            }
            top = x + 8u * lo;
            const uint32_t b0 = bstart16[top << 8], b1 = bstart16[(top + 1u) << 8];
            begin = (uint64_t)b0 + (uint64_t)(u - s.tiles_before[top]) * kTile;
            nvalid = (unsigned)dmin<uint64_t>(kTile, (uint64_t)b1 - begin);
        } else {
            const unsigned x = (unsigned)(v % kPartClasses);
            const uint64_t p0 = dmin<uint64_t>(m, (uint64_t)x * class_len), p1 = dmin<uint64_t>(m, p0 + class_len);
            begin = p0 + (v / kPartClasses) * kTile;
            top = x;
            if (begin < p1) nvalid = (unsigned)dmin<uint64_t>(kTile, p1 - begin);
        }
    };
    // (wave-striped loads, 64 consecutive elements per round; the padding of a short tile carries the largest digit and is
    // ranked behind the real elements of its wave's last rounds: it stays out of the counts and is never stored.  Requesting
    // the next tile's elements before this one leaves was measured: the 32 registers it takes spill into the ranking loop.)
    for (uint64_t v = blockIdx.x; v < ntiles; v += gridDim.x) {
        uint64_t begin;
        unsigned nvalid, top;
        locate(v, begin, nvalid, top);
        if (nvalid == 0u) continue;                                  // (SUB: this class has fewer tiles than the largest)
        uint64_t key[KPT];
        uint32_t pos[KPT];
        unsigned first = w * (kWave * KPT) + lane;                   // (opaque, like t below)
        SFX_OPAQUE_VGPR(first);
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = first + r * kWave;
            key[r] = idx < nvalid ? src.key(begin + idx) : ~0ull;
        }
#pragma unroll
        for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] = (SFX_PART_ABL & 1) ? 0u : rank_round16(digit_of(key[r], shift, 255u), my_flags, s.cnt[w], mybit);
        __syncthreads();
        uint32_t real_count = 0, tile_ex = 0, mine = 0;
        {
            uint32_t c[NW], tile_count = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                c[k] = owner ? s.cnt[k][tid] : 0u;
                tile_count += c[k];
            }
            const uint32_t ex = block_scan_excl_1b<NW>(tile_count, s.part, par);
            if (owner) {
                uint32_t run = ex;
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                // (the reservation is in flight while the tile is staged)
                if (SFX_PART_ABL & 2) mine = (uint32_t)begin + ex;
                else if (real_count) mine = atomicAdd(&cursor[SUB ? ((top << 8) | tid) : (top * kRadix + tid) * kCursorPad], real_count);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++) s.stage[pos[r] + s.cnt[w][digit_of(key[r], shift, 255u)]] = key[r];
        if (owner) s.off[tid] = mine - tile_ex;
        __syncthreads();
        constexpr int kOut = (KPT % 4 == 0) ? 4 : ((KPT % 2 == 0) ? 2 : 1);
        // (the thread index, made opaque: the compiler otherwise computes the sixteen LDS addresses of this loop once, before the
        // tile loop, finds no registers to keep them in and reloads them from scratch in every tile)
        unsigned t = tid;
        SFX_OPAQUE_VGPR(t);
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) key[r] = s.stage[r * kThreads + t];
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) pos[r] = s.off[digit_of(key[r], shift, 255u)] + ((unsigned)r * kThreads + t);
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++)
                if (!(SFX_PART_ABL & 4) && (unsigned)r * kThreads + t < nvalid) out[(SFX_PART_ABL ? pos[r] % m : pos[r])] = key[r];
        }
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
    }
}
// the cursors of both passes from the sub-bucket starts: second pass = the sub-bucket starts themselves; first pass = one per
// (stretch of the input x, top-8 bucket b): the bucket's start + what the stretches before x put into it
__global__ void __launch_bounds__(kBlock)
k_partition_cursors(const uint32_t* __restrict__ bstart16, const uint32_t* __restrict__ class_top8, uint32_t* __restrict__ cursor16,
                    uint32_t* __restrict__ cursor8)
{
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    if (i < (unsigned)(1 << 16)) {
        const uint32_t b = bstart16[i];
        cursor16[i] = b;
        if ((i & 255u) == 0u) {
            uint32_t run = b;
            for (unsigned x = 0; x < kPartClasses; x++) {
                cursor8[(x * kRadix + (i >> 8)) * kCursorPad] = run;
                run += class_top8[x * kRadix + (i >> 8)];
            }
        }
    }
}

// ---- hybrid initial sort: two device-wide passes on the top 16 key bits, the rest in LDS ----------
// An LSD sort moves every element once per 8 key bits through the CU write path (0.41 of HBM peak, §9 of
// DESIGN.md).  When the text is large enough that the 65536 sub-buckets of the top 16 key bits hold a few
// thousand suffixes each, and none more than an LDS tile, two passes suffice: the top two digits are sorted
// device-wide (LSD order: bits [hi-16, hi-8), then [hi-8, hi) -- after them the array is ordered by its top
// 16 bits, stably), then every sub-bucket is sorted by its remaining low bits inside LDS by one workgroup and
// written out sequentially (k_bucket_sort): the same stable LSD ranking, 2 digits deep, with no scatter.
// Sub-bucket boundaries and the digit totals of the two passes come from one 65536-bin histogram of the top
// 16 bits of every suffix's key, counted from the packed text in two sweeps of 32768 LDS counters.
constexpr int kH16Bins = 1 << 16;
constexpr int kH16Words = kH16Bins / 2;                              // two 16-bit counters per LDS word
constexpr int kH16Threads = 1024;
// One sweep over the workgroup's stretch of the text with 16-bit counters (128 KiB of LDS).  A counter can only
// wrap when a sub-bucket holds >= 65536 suffixes of this stretch alone -- far beyond what the LDS sort accepts;
// a wrap changes the sum of all counters by -65535 (low half: its carry lands in the high half) or -65536 (high
// half), never by 0 in any combination, so the host detects it from the total (!= m) and takes the other route.
__global__ void __launch_bounds__(kH16Threads)
k_hist16_text(PackedText t, int drop, uint64_t words_per_block, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[kH16Words];                                 // 128 KiB
    const unsigned tid = threadIdx.x;
    const uint64_t nwords = (t.n + (uint64_t)t.spw - 1) / (uint64_t)t.spw;
    const uint64_t qb = (uint64_t)blockIdx.x * words_per_block;
    const uint64_t qe = dmin<uint64_t>(nwords, qb + words_per_block);
    const uint64_t mask = (1ull << t.kbits) - 1ull;
    for (unsigned i = tid; i < (unsigned)kH16Words; i += kH16Threads) h[i] = 0;
    __syncthreads();
    for (uint64_t q = qb + tid; q < qe; q += kH16Threads) {
        // the spw keys that start in word q, from the two words they span (as packed_key32)
        const uint64_t both = ((uint64_t)t.words[q] << t.kbits) | (uint64_t)t.words[q + 1];
        const uint64_t j0 = q * (uint64_t)t.spw;
        for (int o = 0; o < t.spw; o++) {
            if (j0 + (uint64_t)o >= t.n) break;
            const uint32_t key = (uint32_t)((both >> ((unsigned)(t.spw - o) * (unsigned)t.bits)) & mask);
            const uint32_t top = key >> drop;
            atomicAdd(&h[top >> 1], (top & 1u) ? 65536u : 1u);
        }
    }
    __syncthreads();
    uint32_t* out = partial + (uint64_t)blockIdx.x * kH16Words;
    for (unsigned i = tid; i < (unsigned)kH16Words; i += kH16Threads) out[i] = h[i];
}
// The same counts from (key << 32 | suffix) elements that already exist (a slice of the partitioned build): bits
// [shift, shift + 16) of the element.
__global__ void __launch_bounds__(kH16Threads)
k_hist16_e64(const uint64_t* __restrict__ E, uint64_t m, int shift, uint64_t per_block, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t h[kH16Words];                                 // 128 KiB
    const unsigned tid = threadIdx.x;
    const uint64_t qb = (uint64_t)blockIdx.x * per_block;
    const uint64_t qe = dmin<uint64_t>(m, qb + per_block);
    for (unsigned i = tid; i < (unsigned)kH16Words; i += kH16Threads) h[i] = 0;
    __syncthreads();
    // (one workgroup per CU: eight elements per thread in flight, or the sweep waits on its own loads)
    constexpr int kFly = 8;
    for (uint64_t q0 = qb; q0 < qe; q0 += (uint64_t)kH16Threads * kFly) {
        uint64_t e[kFly];
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            const uint64_t q = q0 + (uint64_t)k * kH16Threads + tid;
            e[k] = q < qe ? E[q] : ~0ull;
        }
#pragma unroll
        for (int k = 0; k < kFly; k++) {
            if (q0 + (uint64_t)k * kH16Threads + tid < qe) {
                const uint32_t top = (uint32_t)(e[k] >> shift) & 0xFFFFu;
                atomicAdd(&h[top >> 1], (top & 1u) ? 65536u : 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* out = partial + (uint64_t)blockIdx.x * kH16Words;
    for (unsigned i = tid; i < (unsigned)kH16Words; i += kH16Threads) out[i] = h[i];
}
// bins[b] += the counts of a slice of the workgroups' partial histograms: workgroup x + 64 y takes the 1024 bins
// from 1024 x (2 KB of every partial, one 8-byte load per thread) and the partials y, y + split, ...
constexpr int kH16ReduceBins = 1024;
constexpr unsigned kH16ReduceSplit = 8;
// (The rows are split among the kH16ReduceSplit = 8 block groups in contiguous stretches of `rows_per_class`: group x sums the
// workgroups that counted the x-th stretch of the text, and a wave's 256 bins are one top-8 bucket -- so the sum over a wave is
// the number of suffixes of that stretch in that bucket: class_top8[x * 256 + bucket], which the first partition pass turns
// into one cursor per (stretch, bucket), k_partition.)
__global__ void __launch_bounds__(kBlock)
k_hist16_reduce(const uint32_t* __restrict__ partial, unsigned nblocks, uint32_t* __restrict__ bins, unsigned rows_per_class,
                uint32_t* __restrict__ class_top8)
{
    const unsigned bx = blockIdx.x % (kH16Bins / kH16ReduceBins), by = blockIdx.x / (kH16Bins / kH16ReduceBins);
    const unsigned w0 = bx * (kH16ReduceBins / 2) + threadIdx.x * 2;              // this thread's two counter words = 4 bins
    uint32_t c[4] = {0, 0, 0, 0};
    const unsigned g1 = dmin(nblocks, (by + 1u) * rows_per_class);
    for (unsigned g = by * rows_per_class; g < g1; g++) {
        const uint2 v = *reinterpret_cast<const uint2*>(partial + (uint64_t)g * kH16Words + w0);
        c[0] += v.x & 0xFFFFu;
        c[1] += v.x >> 16;
        c[2] += v.y & 0xFFFFu;
        c[3] += v.y >> 16;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (c[k]) atomicAdd(&bins[2 * w0 + k], c[k]);
    uint32_t all = c[0] + c[1] + c[2] + c[3];
    for (int d = 32; d >= 1; d >>= 1) all += __shfl_xor(all, d);
    static_assert(kWave * 4 == kRadix, "a wave's bins are one top-8 bucket");
    if (lane_id() == 0) class_top8[by * kRadix + (2u * w0) / (unsigned)kRadix] = all;
}
// totals_lo[d] = sum over the high digits of bin (j, d), totals_hi[j] = sum over the low digits; bins[b] -> first
// position of sub-bucket b (in place), bins[65536] = m.  stat_out (zeroed by the caller) = {largest bin, sum of all
// bins, sub-buckets of more than `cap` suffixes, their suffixes, suffixes in sub-buckets of more than `fast`}.
struct OversizeEntry { uint32_t bin, start, size, off; };
constexpr uint32_t kOversizeMax = 16384;
__global__ void __launch_bounds__(kH16Threads)
k_hist16_scan(uint32_t* __restrict__ bins, uint32_t* __restrict__ totals_lo, uint32_t* __restrict__ totals_hi,
              uint32_t* __restrict__ stat_out, uint32_t cap, uint32_t fast)
{
    __shared__ uint32_t part[kH16Threads / kWave];
    __shared__ uint32_t pmax[kH16Threads / kWave];
    __shared__ uint32_t lo[4][kRadix];
    __shared__ uint32_t tsum[kH16Threads];
    constexpr int kPer = kH16Bins / kH16Threads;                       // 64 consecutive bins per thread: a quarter of a high digit
    const unsigned tid = threadIdx.x;
    {
        // thread (q, d): the bins (j, d) with j = q mod 4
        const unsigned d = tid & 255u, q = tid >> 8;
        uint32_t t = 0;
        for (unsigned j = q; j < (unsigned)kRadix; j += 4) t += bins[j * kRadix + d];
        lo[q][d] = t;
    }
    uint32_t v[kPer], sum = 0, most = 0;
    {
        uint32_t slow = 0, nover = 0, sover = 0;
        for (int j = 0; j < kPer; j++) {
            v[j] = bins[tid * kPer + j];
            sum += v[j];
            most = dmax(most, v[j]);
            if (v[j] > fast) slow += v[j];
            if (v[j] > cap) { nover++; sover += v[j]; }
        }
        if (slow) atomicAdd(&stat_out[4], slow);
        if (nover) { atomicAdd(&stat_out[2], nover); atomicAdd(&stat_out[3], sover); }
    }
    tsum[tid] = sum;
    for (int d = 32; d >= 1; d >>= 1) most = dmax(most, __shfl_xor(most, d));
    // exclusive scan of one value per thread over 16 waves
    const uint32_t incl = wave_scan_add(sum);
    if (lane_id() == 63) part[wave_id()] = incl;
    if (lane_id() == 0) pmax[wave_id()] = most;
    __syncthreads();
    if (tid < (unsigned)kRadix) {
        totals_lo[tid] = lo[0][tid] + lo[1][tid] + lo[2][tid] + lo[3][tid];
        totals_hi[tid] = tsum[4 * tid] + tsum[4 * tid + 1] + tsum[4 * tid + 2] + tsum[4 * tid + 3];
    }
    uint32_t base = 0;
    for (unsigned k = 0; k < wave_id(); k++) base += part[k];
    uint32_t run = base + incl - sum;
    for (int j = 0; j < kPer; j++) { bins[tid * kPer + j] = run; run += v[j]; }
    if (tid == kH16Threads - 1) { bins[kH16Bins] = run; stat_out[1] = run; }
    if (tid == 0) {
        uint32_t mx = 0;
        for (unsigned k = 0; k < (unsigned)(kH16Threads / kWave); k++) mx = dmax(mx, pmax[k]);
        stat_out[0] = mx;
    }
}
// (only when there are any:) the list of the sub-buckets of more than `cap` suffixes, in bin order, from the starts
__global__ void __launch_bounds__(kH16Threads)
k_hist16_oversize(const uint32_t* __restrict__ bstart, uint32_t cap, OversizeEntry* __restrict__ over)
{
    __shared__ uint64_t pover[kH16Threads / kWave];
    constexpr int kPer = kH16Bins / kH16Threads;
    const unsigned tid = threadIdx.x;
    uint64_t ov = 0;                                                  // listed sub-buckets << 32 | their suffixes
    for (int j = 0; j < kPer; j++) {
        const uint32_t b = tid * kPer + j, v = bstart[b + 1] - bstart[b];
        if (v > cap) ov += (1ull << 32) | (uint64_t)v;
    }
    const uint64_t incl = wave_scan_add(ov);
    if (lane_id() == 63) pover[wave_id()] = incl;
    __syncthreads();
    uint64_t run = incl - ov;
    for (unsigned k = 0; k < wave_id(); k++) run += pover[k];
    if (ov == 0) return;
    for (int j = 0; j < kPer; j++) {
        const uint32_t b = tid * kPer + j, start = bstart[b], v = bstart[b + 1] - start;
        if (v > cap) {
            const uint32_t at = (uint32_t)(run >> 32);
            if (at < kOversizeMax) over[at] = OversizeEntry{b, start, v, (uint32_t)run};
            run += (1ull << 32) | (uint64_t)v;
        }
    }
}
// The listed sub-buckets, gathered into one array with (list index, low key bits) as the key -- sorted by the
// device-wide sort as ONE array, they come back in place, sub-bucket by sub-bucket, in key order.
__global__ void __launch_bounds__(kBlock)
k_oversize_gather(const uint64_t* __restrict__ X, const OversizeEntry* __restrict__ over, uint32_t nover, int low_bits,
                  uint64_t* __restrict__ T)
{
    const uint32_t lmask = (uint32_t)((1ull << low_bits) - 1ull);
    for (uint32_t j = blockIdx.x; j < nover; j += gridDim.x) {
        const OversizeEntry e = over[j];
        for (uint32_t i = threadIdx.x; i < e.size; i += kBlock) {
            const uint64_t x = X[(uint64_t)e.start + i];
            const uint32_t key = (j << low_bits) | ((uint32_t)(x >> 32) & lmask);
            T[(uint64_t)e.off + i] = ((uint64_t)key << 32) | (x & 0xFFFFFFFFull);
        }
    }
}
__global__ void __launch_bounds__(kBlock)
k_oversize_return(const uint64_t* __restrict__ T, const OversizeEntry* __restrict__ over, uint32_t nover, int low_bits,
                  uint32_t* __restrict__ K, uint32_t* __restrict__ V)
{
    const uint32_t lmask = (uint32_t)((1ull << low_bits) - 1ull);
    for (uint32_t j = blockIdx.x; j < nover; j += gridDim.x) {
        const OversizeEntry e = over[j];
        for (uint32_t i = threadIdx.x; i < e.size; i += kBlock) {
            const uint64_t x = T[(uint64_t)e.off + i];
            K[(uint64_t)e.start + i] = (e.bin << low_bits) | ((uint32_t)(x >> 32) & lmask);
            V[(uint64_t)e.start + i] = (uint32_t)x;
        }
    }
}

// One workgroup per sub-bucket [bstart[b], bstart[b + 1]) of X (sorted by its top 16 key bits) whose size lies in
// (lo, hi] -- three launches cover three size classes with three geometries: LSD sort by key
// bits [0, low_bits) in LDS -- digit A = bits [0, 8), digit B = bits [8, low_bits) -- then keys and suffixes
// leave as two sequential runs.  A bucket of `size` elements is spread evenly: wave w owns the 64 * kpt
// consecutive elements from w * 64 * kpt, kpt = ceil(size / (64 NW)), so (wave, round, lane) order is memory
// order and the ranking is stable.  Padding (~0) carries the largest digit and stays behind the real elements.
//
// Fast path (`pairs`, taken when no group below holds more than kPairLimit elements -- always, on keys that are
// spread evenly): the elements are GROUPED by the top 10 of their low key bits with one returning LDS atomic each
// (the order inside a group does not matter yet), and an element's final place is its group's start plus the
// number of group members that are smaller -- a couple of LDS reads against a group of 1-4.  An element is a whole
// 64-bit word (key << 32 | suffix), all different, so "smaller" is a total order and ties in the key come out in
// suffix order, exactly as the stable LSD rounds leave them.  No match masks, no wave-level round trips.
constexpr uint32_t kPairLimit = 32;
constexpr int kGroupBits = 10;
constexpr int kGroups = 1 << kGroupBits;
// (The 256 x 8 geometry asks for six workgroups per CU -- 80 registers instead of the 88 the compiler takes unasked, five
// workgroups: the sort waits on LDS round trips 62 % of its wave cycles (scripts/gpu_sq_dna.sh), and one more workgroup to
// switch to is worth 10 %, 0.40 -> 0.36 ms on 100 MB of DNA; seven (72 registers) gives it back, 0.41.)
// TIES (round 6): the sort also says which of its elements share their whole key with a neighbour -- the members of the buckets
// the refinement has to go on with -- so that nobody reads the sorted keys again (k_groups_reduce / k_groups_apply re-read 4 + 8
// bytes per suffix that this kernel had in LDS a moment before: 0.19 of the headline's 1.50 ms).  Keys are not written at all.
// What leaves instead is one bit per SLOT of the array, a device-wide bit mask (bit r = slot r): `tied`, the element shares its
// key with another.  Equal keys are neighbours in the sorted order, so the mask -- with the array, which holds the suffixes --
// says where the stretches of equal keys lie.  On the fast path the thread that places an element has seen every member of its
// group, hence every element with its key: one fire-and-forget LDS atomic, no barrier of its own; after the LSD rounds equal
// keys are neighbours in the staging buffer.  The LDS mask is kept in slot alignment (bit (begin & 31) + place), so its words are
// words of the global array: the inner ones are stored, the first and the last -- shared with the neighbouring sub-buckets --
// or-ed in (the array is zeroed before the launch).  k_tie_direct (sfx_sa.hip) orders the stretches on the text where they
// are; what it cannot finish goes through k_tie_heads / k_tie_list into the first active list.
template <int WORDS, bool ON>
struct TieSmem {
    uint32_t tmask[WORDS + 1];                                      // (+ 1: the sub-bucket starts anywhere inside its first word)
};
template <int WORDS>
struct TieSmem<WORDS, false> {};
template <int NW, int KPT, bool TIES = false, bool TIE_KEYS = false>
__global__ void __launch_bounds__(NW * kWave) SFX_WAVES_PER_EU(NW == 4 && KPT == 8 ? 6 : 1, 8)
k_bucket_sort(const uint64_t* __restrict__ X, const uint32_t* __restrict__ bstart, uint32_t nbuckets, int low_bits,
              uint32_t lo, uint32_t hi, uint32_t* __restrict__ K, uint32_t* __restrict__ V, uint32_t* __restrict__ GT = nullptr,
              int sbits_arg = 32)
{
    constexpr int kThreads = NW * kWave;
    constexpr uint32_t kCap = kThreads * KPT;
    constexpr int kMaskWords = (int)(kCap / 32u);
    // sbits: the low bits of an element that hold the suffix index; the key lies above them, its low_bits low bits are what is
    // sorted here (TIES: 28 when the elements carry 36 key bits, SrcText36; the keys-out form is always 32 + 32)
    const int sbits = TIES ? sbits_arg : 32;
    const uint64_t smask = (1ull << sbits) - 1ull;
    static_assert(kWave * KPT >= kRadix, "the match masks must fit the staging buffer");
    static_assert(NW * kRadix >= kGroups && kGroups % (NW * kWave) == 0, "the group counts of the fast path live in cnt");
    static_assert(kMaskWords < kThreads, "one thread per mask word");
    __shared__ struct {
        uint32_t cnt[NW][kRadix];                                   // LSD rounds: per-wave digit counts; fast path: the group counts
        uint16_t gstart[kGroups];                                   // (sub-buckets hold at most 4096 elements)
        uint32_t part[2][NW];
        uint32_t big;
        uint64_t stage[NW * kWave * KPT];
        TieSmem<kMaskWords, TIES> tie;
    } s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    if (tid == 0) s.big = 0u;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    unsigned par = 0;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0u;
    }
    if constexpr (TIES) {
        if (tid <= (unsigned)kMaskWords) s.tie.tmask[tid] = 0u;
    }
    __syncthreads();
    // Two sub-buckets ahead: the bounds of bucket b + 2 G and the elements of bucket b + G are requested before
    // bucket b is sorted, so that a workgroup always has a bucket of HBM reads in flight.
    auto bounds = [&](uint32_t b, uint32_t& begin, uint32_t& size) {
        begin = 0; size = 0;
        if (b < nbuckets) { begin = bstart[b]; size = bstart[b + 1] - begin; }
        if (size <= lo || size > hi || size > kCap) size = 0;                    // (another size class, or sorted device-wide: k_oversize_*)
    };
    uint64_t nkey[KPT];
    auto fetch = [&](uint32_t begin, uint32_t size) {
        const unsigned per = ((size + kThreads - 1) / kThreads) * kWave;
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * per + r * kWave + lane;
            nkey[r] = ((unsigned)(r * kWave) < per && idx < size) ? X[(uint64_t)begin + idx] : ~0ull;
        }
    };
    uint32_t b = blockIdx.x, begin, size, begin1, size1;
    bounds(b, begin, size);
    bounds(b + gridDim.x, begin1, size1);
    fetch(begin, size);
    for (; b < nbuckets; b += gridDim.x) {
        uint64_t key[KPT];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) key[r] = nkey[r];
        uint32_t begin2, size2;
        bounds(b + 2 * gridDim.x, begin2, size2);
        fetch(begin1, size1);
        const unsigned kpt = (size + kThreads - 1) / kThreads;                  // rounds in use, <= KPT
        const unsigned per = kpt * kWave;
        bool pairs = false;
        // TIES: the sub-bucket's mask words leave for the global arrays (and the LDS copies are cleared for the next sub-bucket);
        // the caller has a barrier between the last atomic on the masks and this
        const uint32_t tshift = begin & 31u;                                    // place p = bit tshift + p of the LDS masks
        auto tie_masks_out = [&]() {
          if constexpr (TIES) {
            const uint32_t words = (tshift + size + 31u) >> 5;
            if (tid < words) {
                const uint64_t at = (uint64_t)(begin >> 5) + tid;
                const uint32_t mt = s.tie.tmask[tid];
                if (tid == 0 || tid + 1u == words) {
                    if (mt) atomicOr(&GT[at], mt);
                } else {
                    GT[at] = mt;
                }
                s.tie.tmask[tid] = 0u;
            }
          }
        };
        if (size > 1) {
            // group by the top bits of the low key part: place inside the group from a returning atomic, group starts by a scan
            const int gbits = low_bits < kGroupBits ? low_bits : kGroupBits;
            const int gshift = sbits + low_bits - gbits;
            const unsigned gmask = (1u << gbits) - 1u;
            uint32_t* const gcount = &s.cnt[0][0];
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)r < kpt && w * per + r * kWave + lane < size) pos[r] = atomicAdd(&gcount[digit_of(key[r], gshift, gmask)], 1u);
            __syncthreads();
            constexpr int kPerThread = kGroups / kThreads;                      // consecutive groups per thread
            uint32_t c[kPerThread], sum = 0, most = 0;
#pragma unroll
            for (int k = 0; k < kPerThread; k++) {
                c[k] = gcount[tid * kPerThread + k];
                sum += c[k];
                most = dmax(most, c[k]);
            }
            uint32_t run = block_scan_excl_1b<NW>(sum, s.part, par);
#pragma unroll
            for (int k = 0; k < kPerThread; k++) {
                s.gstart[tid * kPerThread + k] = (uint16_t)run;
                run += c[k];
            }
            if (most > kPairLimit) s.big = 1u;
            __syncthreads();
            pairs = s.big == 0u;
            if (pairs) {
#pragma unroll
                for (int r = 0; r < KPT; r++)
                    if ((unsigned)r < kpt && w * per + r * kWave + lane < size)
                        s.stage[s.gstart[digit_of(key[r], gshift, gmask)] + pos[r]] = key[r];
                __syncthreads();
                if constexpr (TIES) {
                    for (unsigned q = tid; q < size; q += kThreads) {
                        const uint64_t e = s.stage[q];
                        const unsigned d = digit_of(e, gshift, gmask);
                        const unsigned gb = s.gstart[d], ge = gb + gcount[d];
                        unsigned rank = 0, same = 0;
                        for (unsigned j = gb; j < ge; j++) {
                            const uint64_t x = s.stage[j];
                            rank += x < e ? 1u : 0u;
                            same += ((x ^ e) >> sbits) == 0ull ? 1u : 0u;          // (counts e itself)
                        }
                        const unsigned place = gb + rank;
                        V[(uint64_t)begin + place] = (uint32_t)(e & smask);
                        if constexpr (TIE_KEYS) K[(uint64_t)begin + place] = (uint32_t)(e >> 32);  // (the fused LCP: the 32-bit key, whatever sbits is)
                        if (same > 1u) {
                            const unsigned bitp = tshift + place;
                            atomicOr(&s.tie.tmask[bitp >> 5], 1u << (bitp & 31u));
                        }
                    }
                } else {
                for (unsigned q = tid; q < size; q += kThreads) {
                    const uint64_t e = s.stage[q];
                    const unsigned d = digit_of(e, gshift, gmask);
                    const unsigned gb = s.gstart[d], ge = gb + gcount[d];
                    unsigned rank = 0;
                    for (unsigned j = gb; j < ge; j++) rank += s.stage[j] < e ? 1u : 0u;
                    K[(uint64_t)begin + gb + rank] = (uint32_t)(e >> 32);
                    V[(uint64_t)begin + gb + rank] = (uint32_t)e;
                }
                }
            }
            __syncthreads();                                                    // (stage and the counts are read to the end)
#pragma unroll
            for (int k = 0; k < kPerThread; k++) gcount[tid * kPerThread + k] = 0u;
            if (tid == 0) s.big = 0u;
            if (pairs) tie_masks_out();
            __syncthreads();
        }
        for (int pass = 0; pass < 3 && size > 1 && !pairs; pass++) {           // (low_bits <= 16, or 20 with the longer keys: 3 rounds)
            const int shift = sbits + 8 * pass;
            const int nb = low_bits - 8 * pass < 8 ? low_bits - 8 * pass : 8;
            if (nb <= 0) break;
            const unsigned mask = (1u << nb) - 1u;
#pragma unroll
            for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
            wave_sync();
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)r < kpt) pos[r] = rank_round<true>(digit_of(key[r], shift, mask), my_flags, s.cnt[w], mybit);
            __syncthreads();
            {
                uint32_t c[NW], tile_count = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    c[k] = owner ? s.cnt[k][tid] : 0u;
                    tile_count += c[k];
                }
                const uint32_t ex = block_scan_excl_1b<NW>(tile_count, s.part, par);
                if (owner) {
                    uint32_t run = ex;
#pragma unroll
                    for (int k = 0; k < NW; k++) {
                        s.cnt[k][tid] = run;
                        run += c[k];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)r < kpt) s.stage[pos[r] + s.cnt[w][digit_of(key[r], shift, mask)]] = key[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)r < kpt) key[r] = s.stage[w * per + r * kWave + lane];
            if (owner) {
#pragma unroll
                for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0u;
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * per + r * kWave + lane;
            if ((unsigned)r < kpt && idx < size && !pairs) {
                V[(uint64_t)begin + idx] = (uint32_t)(key[r] & smask);
                if constexpr (TIES) {
                    if constexpr (TIE_KEYS) K[(uint64_t)begin + idx] = (uint32_t)(key[r] >> 32);
                    // (after the LSD rounds the sub-bucket lies sorted in the staging buffer: a run of equal keys is a run of neighbours)
                    if (size > 1u) {
                        const uint64_t kk = key[r] >> sbits;
                        const bool eq_prev = idx > 0u && (s.stage[idx - 1u] >> sbits) == kk;
                        const bool eq_next = idx + 1u < size && (s.stage[idx + 1u] >> sbits) == kk;
                        if (eq_prev || eq_next) {
                            const unsigned bitp = tshift + idx;
                            atomicOr(&s.tie.tmask[bitp >> 5], 1u << (bitp & 31u));
                        }
                    }
                } else {
                    K[(uint64_t)begin + idx] = (uint32_t)(key[r] >> 32);
                }
            }
        }
        if constexpr (TIES) {
            if (size > 1u && !pairs) {                                          // (block-uniform; the fast path wrote its masks before it let go of the groups)
                __syncthreads();
                tie_masks_out();
            }
        }
        begin = begin1; size = size1;
        begin1 = begin2; size1 = size2;
    }
}

// ---- host side -------------------------------------------------------------------------
// Tuning knobs (development only; read once per process):
//   SFX_RADIX_SWEEP  1 = one-sweep (default), 0 = chunked
//   SFX_RADIX_KPT    elements per thread of the E64 passes: 11 (default), 16 or 8
//   SFX_RADIX_KPT_TEXT  ... of the text-fed pass: 16 (default), 11 or 8
//   SFX_RADIX_RANK   1 = LDS match masks (default), 0 = 8-ballot match
//   SFX_RADIX_NW     waves per workgroup: 4, 8 or 16 (default); tile = 64 * NW * KPT elements
struct RadixTuning { int sweep, kpt, rank, nw, kpt_text, kpt_kv, kv12, duo, duo_e64; };
static RadixTuning radix_tuning()
{
    static const RadixTuning t = [] {
        RadixTuning r = {1, 16, 1, 16, 16, 12, 1, 1, 0}; // measured best on MI355X (round 4, lab/radix_lab2.hip): 1024-thread
                                                // workgroups, 16384-element E64 tiles (512-byte runs), 12288-element KV tiles
        if (const char* e = dev_env("SFX_RADIX_SWEEP")) r.sweep = atoi(e) ? 1 : 0;
        if (const char* e = dev_env("SFX_RADIX_KPT")) r.kpt = (atoi(e) >= 8 && atoi(e) <= 16) ? atoi(e) : 8;
        if (const char* e = dev_env("SFX_RADIX_RANK")) r.rank = atoi(e) ? 1 : 0;
        if (const char* e = dev_env("SFX_RADIX_KPT_TEXT")) r.kpt_text = (atoi(e) >= 8 && atoi(e) <= 16) ? atoi(e) : 16;
        if (const char* e = dev_env("SFX_RADIX_KPT_KV")) r.kpt_kv = (atoi(e) >= 8 && atoi(e) <= 12) ? atoi(e) : 8;
        if (const char* e = dev_env("SFX_RADIX_KV12")) r.kv12 = atoi(e) ? 1 : 0;         // 0: (key array, value array) in every pass
        if (const char* e = dev_env("SFX_RADIX_DUO_E64")) r.duo_e64 = atoi(e) ? 1 : 0;   // 1: the E64 -> E64 one-sweep passes by k_radix_sweep_duo (development)
        if (const char* e = dev_env("SFX_RADIX_DUO")) r.duo = atoi(e) ? 1 : 0;           // 0: the KV passes by k_radix_sweep (one workgroup per CU)
        if (const char* e = dev_env("SFX_RADIX_NW")) r.nw = atoi(e) == 16 ? 16 : (atoi(e) == 8 ? 8 : 4);
        return r;
    }();
    return t;
}

uint64_t radix_scratch_words(uint64_t m)
{
    // [0, partial): max(one-sweep partials 8*256*1024, chunked matrix 256*2048)
    // totals 8*256, tickets 64, status m/8 + 256 (tiles of >= 2048 elements)
    return (uint64_t)kMaxPasses * kRadix * kHistAllGrid + (uint64_t)kMaxPasses * kRadix + 64 + m / 8 + 2 * kRadix;
}

struct RadixScratch {
    uint32_t* partial;      // one-sweep: [npass*256][grid] ; chunked: [256][grid]
    uint32_t* totals;       // [npass][256]
    uint32_t* tickets;      // [kMaxPasses]
    uint32_t* status;       // [tiles][256]
    RadixScratch(uint32_t* base, uint64_t)
    {
        partial = base;
        totals = partial + (uint64_t)kMaxPasses * kRadix * kHistAllGrid;
        tickets = totals + (uint64_t)kMaxPasses * kRadix;
        status = tickets + 64;
    }
};

template <class Src, class Dst, int KPT, bool ONESWEEP, bool RANK_ATOMIC, int NW>
static int launch_pass(const char* name, double algo_bytes, const Src& src, const Dst& dst, uint64_t m, int shift,
                       unsigned mask, const RadixScratch& scr, int pass, hipStream_t st)
{
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    if constexpr (ONESWEEP) {
        const uint64_t tiles = (m + kTile - 1) / kTile;
        const unsigned grid = (unsigned)dmin<uint64_t>(tiles, kMaxGrid);
        // (the status words are zeroed once, for the tile count of the kernel that runs: the two-workgroup kernels have smaller tiles)
        auto zero_status = [&](uint64_t t) { return hipMemsetAsync(scr.status, 0, t * kRadix * sizeof(uint32_t), st); };
        if constexpr (RANK_ATOMIC && NW == 16 && KPT == 12 && Src::kHasVal && !Src::kFromText && !std::is_same<Src, SrcKeyIota>::value) {
            // (7168-element tiles held by 512 threads, two workgroups per CU: k_radix_sweep_duo.  Not the first pass of a compressed-key
            // sort, which reads 8 bytes per element where the others read 12: 5.31 ms with k_radix_sweep, 5.57 with this)
            if (radix_tuning().duo) {
                const uint64_t dtile = (uint64_t)kDuoKPT * kDuoNW * kWave;
                const uint64_t dtiles = (m + dtile - 1) / dtile;
                const unsigned dgrid = (unsigned)dmin<uint64_t>(dtiles, kMaxGrid);
                SFX_HIP(zero_status(dtiles));
                SFX_LAUNCH(name, algo_bytes, (k_radix_sweep_duo<Src, Dst, kDuoKPT, kDuoNW>), dgrid, kDuoNW * kWave, st, src, dst, m, shift,
                           mask, (const uint32_t*)(scr.totals + pass * kRadix), scr.status, scr.tickets + pass);
                return SFX_OK;
            }
        }
        if constexpr (RANK_ATOMIC && NW == 16 && KPT == 16 && std::is_same<Src, SrcE64>::value && std::is_same<Dst, DstE64>::value) {
            // (8-byte elements in, 8-byte elements out: 8192-element tiles held by 512 threads, two workgroups per CU -- development
            // only, SFX_RADIX_DUO_E64=1.  Measured on 10^9 bytes each (profiles/r5_duo_e64_ab.jsonl): the rank-update partition passes
            // of the near-duplicate documents 70.8 -> 67.0 ms, but the E64 passes of 1 GB of DNA 12.6 -> 13.2 and those of config 5
            // 23.2 -> 25.2: 8 bytes per element leave a 8192-element tile 256-byte runs, and that costs what the overlap gives)
            if (radix_tuning().duo_e64) {
                const uint64_t dtile = (uint64_t)kDuoKPTE64 * kDuoNW * kWave;
                const uint64_t dtiles = (m + dtile - 1) / dtile;
                const unsigned dgrid = (unsigned)dmin<uint64_t>(dtiles, kMaxGrid);
                SFX_HIP(zero_status(dtiles));
                SFX_LAUNCH(name, algo_bytes, (k_radix_sweep_duo<Src, Dst, kDuoKPTE64, kDuoNW>), dgrid, kDuoNW * kWave, st, src, dst, m,
                           shift, mask, (const uint32_t*)(scr.totals + pass * kRadix), scr.status, scr.tickets + pass);
                return SFX_OK;
            }
        }
        SFX_HIP(zero_status(tiles));
        if constexpr (RANK_ATOMIC && NW == 16 && kTile < 65536) {
            SFX_LAUNCH(name, algo_bytes, (k_radix_sweep<Src, Dst, KPT, NW>), grid, kThreads, st, src, dst, m, shift, mask,
                       (const uint32_t*)(scr.totals + pass * kRadix), scr.status, scr.tickets + pass);
        } else {
            SFX_LAUNCH(name, algo_bytes, (k_radix_pass<Src, Dst, KPT, true, RANK_ATOMIC, NW>), grid, kThreads, st, src, dst, m,
                       shift, mask, (uint64_t)0, (const uint32_t*)nullptr, (const uint32_t*)(scr.totals + pass * kRadix),
                       scr.status, scr.tickets + pass);
        }
    } else {
        Chunking ch = make_chunking(m, kTile);
        const uint64_t chunk = ch.tiles_per_block * kTile;
        SFX_LAUNCH("radix_hist", (double)m * 8.0, (k_radix_hist_chunk<Src>), ch.blocks, kBlock, st, src, m, shift,
                   mask, chunk, scr.partial);
        SFX_LAUNCH("radix_scan", (double)kRadix * ch.blocks * 8, k_radix_scan, kRadix, kBlock, st, scr.partial,
                   ch.blocks, scr.totals);
        SFX_LAUNCH(name, algo_bytes, (k_radix_pass<Src, Dst, KPT, false, RANK_ATOMIC, NW>), ch.blocks, kThreads, st, src,
                   dst, m, shift, mask, chunk, (const uint32_t*)scr.partial, (const uint32_t*)scr.totals,
                   (uint32_t*)nullptr, (uint32_t*)nullptr);
    }
    return SFX_OK;
}

template <class Src, class Dst>
static int run_pass(const char* name, double algo_bytes, const Src& src, const Dst& dst, uint64_t m, int shift,
                    unsigned mask, const RadixScratch& scr, int pass, bool sweep, hipStream_t st)
{
    const RadixTuning t = radix_tuning();
#define SFX_PASS(KPT, SW, RK, NW) launch_pass<Src, Dst, KPT, SW, RK, NW>(name, algo_bytes, src, dst, m, shift, mask, scr, pass, st)
#define SFX_PASS_NW(KPT, NW)                                                                     \
    do {                                                                                         \
        if (sweep) return t.rank ? SFX_PASS(KPT, true, true, NW) : SFX_PASS(KPT, true, false, NW);  \
        return t.rank ? SFX_PASS(KPT, false, true, NW) : SFX_PASS(KPT, false, false, NW);           \
    } while (0)
    if (t.nw == 16) {
        if constexpr (!Src::kHasVal) {                           // (KV elements: 8 per thread, LDS)
            // elements per thread swept from 8 to 16 on hardware (profiles/r1c_radix_variants.txt):
            // 11 wins for the E64 passes, 16 for the text-fed pass; the other sizes are not built
            // (the chunked schedule keeps the next tile's elements in registers as well: 16 per thread spill there -- 2 * 10^9 bytes
            // of DNA 107.9 ms at 16, 84.1 at 11, 88.7 at 8)
            const int kk = !sweep ? (t.kpt > 11 ? 11 : t.kpt) : (Src::kFromText ? t.kpt_text : t.kpt);
            if (kk == 16) SFX_PASS_NW(16, 16);
            if (kk == 11) SFX_PASS_NW(11, 16);
        }
        if constexpr (Src::kHasVal && !Src::kFromText) {
            // KV passes (12-byte elements): 10 per thread was the largest tile without spills in k_radix_sweep until its output
            // loop stopped leaving sixteen hoisted LDS addresses in scratch (SFX_OPAQUE_VGPR); 12 is what the LDS holds.  Config 3's
            // eight passes: 53.8 / 52.2 / 51.9 ms at 10 / 11 / 12.
            // (round 3's kernel: 9; 8 / 9 / 10 measured 106.2 / 102.0 / 105.1 ms over the 23 KV passes of config 3 then)
            // (11 and 12: the one-sweep kernel only -- the tile of the other schedules' kernel does not fit the LDS)
            if (t.kpt_kv == 12 && sweep && t.rank) return SFX_PASS(12, true, true, 16);
            if (t.kpt_kv == 11 && sweep && t.rank) return SFX_PASS(11, true, true, 16);
            // (chunked schedule, m >= 2^30: 8 per thread -- the prefetched tile's registers again; 1.5 * 10^9 bytes of English-like
            // text 204.7 / 186.4 / 185.9 ms at 10 / 9 / 8)
            static const int kv_chunked = [] { const char* e = dev_env("SFX_RADIX_KPT_KV_CHUNKED"); return e ? atoi(e) : 8; }();
            if (sweep && t.kpt_kv >= 10) SFX_PASS_NW(10, 16);
            if (!sweep && kv_chunked >= 10) SFX_PASS_NW(10, 16);
            if (sweep ? t.kpt_kv == 9 : kv_chunked == 9) SFX_PASS_NW(9, 16);
        }
        SFX_PASS_NW(8, 16);
    }
    if (t.nw == 8) { if (t.kpt == 16) SFX_PASS_NW(16, 8); SFX_PASS_NW(8, 8); }
    if (t.kpt == 16) SFX_PASS_NW(16, 4);
    SFX_PASS_NW(8, 4);
#undef SFX_PASS_NW
#undef SFX_PASS
}

// one-sweep preparation: digit totals of every pass + zeroed tickets
template <class Src>
static int prepare_sweep(const char* name, double algo_bytes, const Src& src, uint64_t m, int bit_lo, int bit_hi,
                         int npass, const RadixScratch& scr, hipStream_t st)
{
    Chunking ch = make_chunking(m, kBlock * 8, kHistAllGrid);
    const uint64_t chunk = ch.tiles_per_block * kBlock * 8;
    SFX_HIP(hipMemsetAsync(scr.tickets, 0, 64 * sizeof(uint32_t), st));
    SFX_LAUNCH(name, algo_bytes, (k_radix_hist_all<Src>), ch.blocks, kBlock, st, src, m, bit_lo, bit_hi, npass, chunk,
               scr.partial);
    SFX_LAUNCH("radix_scan", (double)npass * kRadix * ch.blocks * 8, k_radix_scan, npass * kRadix, kBlock, st,
               scr.partial, ch.blocks, scr.totals);
    return SFX_OK;
}

// the window form applies to a full text-fed build whose digits are whole symbols
static bool window_hist_applies(const PackedText* text, uint64_t m, int bit_lo, int bit_hi)
{
    return text && m == text->n && text->kbits == 32 && (8 % text->bits) == 0 && bit_lo == 32 && bit_hi == 64;
}
static int prepare_sweep_windows(const PackedText& t, const RadixScratch& scr, hipStream_t st)
{
    const uint64_t npos = t.n + 3ull * (8 / t.bits);
    const uint64_t nwords = (npos + t.spw - 1) / t.spw;
    Chunking ch = make_chunking(nwords, kBlock * 4, kHistAllGrid);
    const uint64_t wpb = ch.tiles_per_block * kBlock * 4;
    SFX_HIP(hipMemsetAsync(scr.tickets, 0, 64 * sizeof(uint32_t), st));
    SFX_LAUNCH("radix_hist_all_text_u32", (double)t.n * t.bits / 8.0, k_window_hist, ch.blocks, kBlock, st, t, npos, wpb,
               scr.partial);
    SFX_LAUNCH("radix_scan", (double)kRadix * ch.blocks * 8, k_radix_scan, kRadix, kBlock, st, scr.partial, ch.blocks,
               scr.totals);
    SFX_LAUNCH("radix_window_fix", 0.0, k_window_fix, 1, kBlock, st, t, scr.totals);
    return SFX_OK;
}

static bool use_sweep(uint64_t m, int npass)
{
    return radix_tuning().sweep && m < (1ull << 30) && npass <= kMaxPasses;
}

// The hybrid route of a text-fed E64 sort with split output (see k_bucket_sort).  Applies to texts between
// 2^25 and 2^28 suffixes whose key has at least 24 bits (below, a sub-bucket is too small to keep a workgroup
// busy: 20 MB of DNA 0.63 against 0.59 ms, 34 MB 0.82 against 0.88, 50 MB 1.04 against 1.23, 100 MB 1.78 against
// 2.24, 200 MB 3.88 against 4.63).  Sub-buckets of more than 16384 suffixes (skewed composition, repeats) do not
// fit the LDS: they are gathered into one array keyed by (list index, low key bits), sorted by the ordinary
// device-wide passes as ONE array and copied back.  The route is for evenly spread keys with a few outliers: when
// more than 1/64 of the suffixes sit in sub-buckets above 4096 (or a 16-bit counter wrapped) it gives way (returns
// 0 in *done) and the text keeps the four-pass sort.
//   SFX_HYBRID=0 switches it off; SFX_HYBRID_MIN=<suffixes> (tests) moves the lower bound; SFX_HYBRID_CAP=<elements>
//   (tests) lowers the size from which a sub-bucket counts as oversized.
constexpr int kBucketNW = 16, kBucketKPT = 16;                // the largest geometry: sub-buckets of up to 16384 suffixes
// elem_bits > 0 (a slice of the partitioned build): the elements exist already, in e0, with keys below 2^elem_bits
// (k_range_filter stores key - first key of the range): the sub-bucket of an element is its key >> (elem_bits - 16), the
// histogram is counted from the elements (k_hist16_e64), both device-wide passes read elements (e0 -> e1 -> e0) and the
// LDS sort reads e0; when the route gives way nothing has been touched but e1 and the scratch.
// will a slice of m explicit elements whose keys fit key_bits bits take the hybrid route (hybrid_sort_e64_text's own entry test)?
// Then its sub-bucket histogram supplies the digit totals and the producer of the elements need not count digits.
bool radix_e64_hybrid_expected(uint64_t m, int key_bits)
{
    static const int enabled = [] { const char* e = dev_env("SFX_HYBRID"); return e ? atoi(e) : 1; }();
    static const uint64_t min_m = [] { const char* e = dev_env("SFX_HYBRID_MIN"); return e ? (uint64_t)strtoull(e, nullptr, 10) : (1ull << 25); }();
    return enabled && key_bits >= 24 && m >= min_m && m <= (1ull << 28);
}
static int hybrid_sort_e64_text(uint64_t* e0, uint64_t* e1, uint64_t m, int bit_lo, int bit_hi, const RadixScratch& scr,
                                hipStream_t st, sfx_build_stats* stats, const PackedText& text, uint32_t* split_v,
                                uint32_t** split_k_out, bool* done, bool* windows_ready, int elem_bits = 0, TieRecords* ties = nullptr)
{
    if (ties) ties->produced = false;
    *done = false;
    *windows_ready = false;
    const bool from_elems = elem_bits > 0;
    static const int enabled = [] { const char* e = dev_env("SFX_HYBRID"); return e ? atoi(e) : 1; }();
    static const uint64_t min_m = [] { const char* e = dev_env("SFX_HYBRID_MIN"); return e ? (uint64_t)strtoull(e, nullptr, 10) : (1ull << 25); }();
    static const uint32_t cap = [] {
        const char* e = dev_env("SFX_HYBRID_CAP");
        const uint32_t full = kBucketNW * kWave * kBucketKPT;
        const uint32_t v = e ? (uint32_t)atoi(e) : full;
        return v >= 1 && v < full ? v : full;
    }();
    const int key_bits = from_elems ? elem_bits : bit_hi - bit_lo;
    if (!enabled || bit_lo != 32 || key_bits < 24 || key_bits > bit_hi - bit_lo || m < min_m || m > (1ull << 28)) return SFX_OK;
    if (!from_elems && (key_bits != text.kbits || m != text.n)) return SFX_OK;
    const int low_bits = key_bits - 16;
    const int top_hi = bit_lo + key_bits;                      // the sub-bucket = element bits [top_hi - 16, top_hi)
    // counts / sub-bucket starts, statistics and the oversize list sit at the END of the histogram scratch: the
    // device-wide sort of the oversized sub-buckets (<= 4 passes) uses its first half
    constexpr uint64_t kReserve = 1u << 18;
    uint32_t* bins = scr.partial + ((uint64_t)kMaxPasses * kRadix * kHistAllGrid - kReserve);     // 65537 u32
    uint32_t* stat = bins + kH16Bins + 64;                     // largest sub-bucket, sum of all, oversized ones, their suffixes
    OversizeEntry* over = reinterpret_cast<OversizeEntry*>(bins + kH16Bins + 128);
    static_assert(kH16Bins + 128 + 4 * kOversizeMax <= kReserve, "the reserve holds bins, statistics and the oversize list");
    uint32_t* partial = reinterpret_cast<uint32_t*>(e1);       // [workgroups][65536]: e1 is idle until the second pass
    const uint64_t nwords = from_elems ? m : (m + (uint64_t)text.spw - 1) / (uint64_t)text.spw;    // (units of the histogram sweep)
    const uint64_t room = m * sizeof(uint64_t) / (kH16Words * sizeof(uint32_t));     // partial histograms that fit e1
    if (room == 0) return SFX_OK;
    Chunking ch = make_chunking(nwords, kH16Threads, (unsigned)dmin<uint64_t>(room, 256));
    SFX_HIP(hipMemsetAsync(scr.tickets, 0, 64 * sizeof(uint32_t), st));
    SFX_HIP(hipMemsetAsync(bins, 0, (kH16Bins + 128) * sizeof(uint32_t), st));             // (the counts and the statistics)
    if (from_elems)
        SFX_LAUNCH("radix_hist16_elems", (double)m * 8.0, k_hist16_e64, ch.blocks, kH16Threads, st, (const uint64_t*)e0, m, top_hi - 16,
                   ch.tiles_per_block * kH16Threads, partial);
    else
        SFX_LAUNCH("radix_hist16_text", (double)m * text.bits / 8.0, k_hist16_text, ch.blocks, kH16Threads, st, text, low_bits,
                   ch.tiles_per_block * kH16Threads, partial);
    // (the rows of the partial counts in kPartClasses contiguous stretches: k_partition's first pass wants the top-8 counts of each)
    static_assert(kH16ReduceSplit == kPartClasses, "one block group of the reduction per stretch");
    const unsigned rows_per_class = (ch.blocks + kPartClasses - 1) / kPartClasses;
    const uint64_t class_len = (uint64_t)rows_per_class * ch.tiles_per_block * kH16Threads * (from_elems ? 1u : (uint64_t)text.spw);
    uint32_t* cursor16 = bins + kH16Bins + 128 + 4 * kOversizeMax;     // (behind the oversize list)
    uint32_t* class_top8 = cursor16 + kH16Bins;
    uint32_t* cursor8 = class_top8 + kPartClasses * kRadix;
    static_assert(kH16Bins + 128 + 4 * kOversizeMax + kH16Bins + kPartClasses * kRadix * (1 + kCursorPad) <= kReserve, "the reserve holds the cursors too");
    SFX_LAUNCH("radix_hist16_reduce", (double)ch.blocks * kH16Words * 4, k_hist16_reduce, (kH16Bins / kH16ReduceBins) * kH16ReduceSplit, kBlock, st,
               (const uint32_t*)partial, ch.blocks, bins, rows_per_class, class_top8);
    SFX_LAUNCH("radix_hist16_scan", (double)kH16Bins * 8, k_hist16_scan, 1, kH16Threads, st, bins, scr.totals, scr.totals + kRadix,
               stat, cap, dmin(cap, 4096u));
    uint32_t host_stat[5] = {0, 0, 0, 0, 0};
    SFX_TRY(read_back(host_stat, stat, sizeof(host_stat), st));
    const uint32_t host_max = host_stat[0], nover = host_stat[2];
    const uint64_t nlarge = host_stat[3], nslow = host_stat[4];
    // A 16-bit counter wrapped, or more than 1/64 of the text sits in sub-buckets of more than 4096 suffixes: four
    // passes.  (Measured on 100 MB over {A, C, G, T} drawn 32/18/18/32 %: 15 % of the suffixes in sub-buckets of 4097 ..
    // 16384 -- one workgroup per CU sorts them at 50 ps per suffix, five times the price of two device-wide passes --
    // and 3.3 ms against 2.5; the device-wide sort of the oversized ones is latency-bound, ~0.3 ms however few they are.
    // The route pays on evenly spread keys and tolerates a few outliers: repeats in an otherwise random-like text.)
    const bool give_way = nover > kOversizeMax || nslow * 64 > m ||
                          (nover && (m + 1) / 2 + 32 + 2 * (nlarge + 32) > m);   // (tiny inputs of the tests: no room behind the keys)
    if ((uint64_t)host_stat[1] != m) return SFX_OK;
    if (give_way) {
        // the four-pass sort takes its digit totals from this histogram when its digits are whole symbols
        // (prepare_sweep_windows would count the 8-bit windows again)
        if (!from_elems && text.kbits == 32 && (8 % text.bits) == 0 && bit_hi == 64) {
            SFX_HIP(hipMemsetAsync(scr.tickets, 0, 64 * sizeof(uint32_t), st));
            SFX_LAUNCH("radix_window_from_hist16", 0.0, k_window_from_hist16, 1, kBlock, st, scr.totals, (uint32_t)(3 * (8 / text.bits)));
            SFX_LAUNCH("radix_window_fix", 0.0, k_window_fix, 1, kBlock, st, text, scr.totals);
            *windows_ready = true;
        }
        return SFX_OK;
    }
    const bool sweep = true;
    static const int partition = [] { const char* e = dev_env("SFX_HYBRID_PARTITION"); return e ? atoi(e) : 1; }();
    // With no oversized sub-bucket the LDS sort can name the tied elements itself (k_bucket_sort<.., true>): no sorted keys are
    // written, the caller orders the ties from the mask (TieRecords, sfx_host.hpp).  SFX_HYBRID_TIES=0 (development): the
    // sorted keys, as rounds 3-5.
    static const int ties_on = [] { const char* e = dev_env("SFX_HYBRID_TIES"); return e ? atoi(e) : 1; }();
    const bool tie_mode = ties && ties_on && nover == 0;
    // ... and then nobody reads a 32-bit key out of an element again: a text-fed sort lets the key grow into the four bits that the
    // suffix index (m <= 2^28) leaves free -- two more symbols of DNA, a sixteenth of the ties (SrcText36).  SFX_HYBRID_KEY36=0
    // (development): 32 + 32 bits.
    // One symbol more comes out of the two packed words the 32-bit key is read from; two symbols of DNA (SFX_HYBRID_KEY36=2) need
    // a third word now and then -- a 12-byte load per element: the text-fed pass 0.347 -> 0.381 ms for 0.048 -> 0.033 in k_tie_direct.
    static const int key36_on = [] { const char* e = dev_env("SFX_HYBRID_KEY36"); return e ? atoi(e) : 1; }();
    int extra = 0;
    if (tie_mode && key36_on && partition && !from_elems && text.kbits == 32 && text.bits <= 4 && m <= (1ull << 28))
        extra = key36_on >= 2 ? (4 / text.bits) * text.bits : text.bits;
    const bool wide_key = extra > text.bits;
    const int sbits = 32 - extra;
    if (partition) {
        // two partition passes (k_partition: no order inside a sub-bucket, none needed): top 8 bits, then the next 8 inside
        // every top-8 bucket; cursors behind the oversize list
        constexpr int kPKpt = 16, kPNw = 16;
        SFX_LAUNCH("partition_cursors", (double)kH16Bins * 8, k_partition_cursors, kH16Bins / kBlock, kBlock, st, (const uint32_t*)bins,
                   (const uint32_t*)class_top8, cursor16, cursor8);
        const uint64_t tile = (uint64_t)kPKpt * kPNw * kWave;
        // (one workgroup per CU, each striding over the tiles: 128 KB of LDS leave room for no second one)
        static const unsigned cus = [] {
            const char* e = dev_env("SFX_PARTITION_GRID");
            if (e && atoi(e) > 0) return (unsigned)atoi(e);
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            return (unsigned)n;
        }();
        const unsigned grid1 = (unsigned)dmin<uint64_t>((m + tile - 1) / tile, dmin(cus, grid_cap()));
        const unsigned grid2 = (unsigned)dmin<uint64_t>((m + tile - 1) / tile + kRadix, dmin(cus, grid_cap()));
        // A full build's two passes run as TWO workgroups of 8 waves per CU (round 5): 8192-element tiles, 72 KB of LDS each, the
        // same 128 registers per thread -- one workgroup loads and ranks while the other writes (what k_radix_sweep_duo does for the
        // stable passes; here the whole tile fits the LDS and the kernel is the same template).  Headline, 50 steps each
        // (profiles/r5_partition_duo_bench.jsonl): 1.566 / 1.577 ms with one workgroup of 16 waves, 1.523 / 1.523 with two of 8 (the
        // text-fed pass 0.400 -> 0.353 ms, the element-fed one 0.400 -> 0.373); the runs of consecutive tiles meet in one XCD's L2
        // either way, so the shorter run of a smaller tile costs little here.  SFX_PARTITION_WAVES=16 / 8 / 4 (development).
        static const int pwaves = [] { const char* e = dev_env("SFX_PARTITION_WAVES"); const int v = e ? atoi(e) : 8; return (v == 16 || v == 4) ? v : 8; }();
        if (pwaves != 16) {
#define SFX_PARTITION_PAIR(DNW)                                                                                                         \
            do {                                                                                                                            \
                const uint64_t dtile = (uint64_t)kPKpt * (DNW) * kWave;                                                                     \
                const unsigned per_cu = 16u / (DNW);                                                                                        \
                const unsigned g1 = (unsigned)dmin<uint64_t>((m + dtile - 1) / dtile, dmin(per_cu * cus, grid_cap()));                     \
                const unsigned g2 = (unsigned)dmin<uint64_t>((m + dtile - 1) / dtile + kRadix, dmin(per_cu * cus, grid_cap()));            \
                if (from_elems) {                                                                                                           \
                    SFX_LAUNCH("radix_scatter_u32", (double)m * 16.0, (k_partition<SrcE64, kPKpt, DNW, false>), g1, (DNW) * kWave, st,      \
                               SrcE64{e0}, e1, m, top_hi - 8, cursor8, (const uint32_t*)bins, class_len);                                   \
                    SFX_LAUNCH("radix_scatter_u32", (double)m * 16.0, (k_partition<SrcE64, kPKpt, DNW, true>), g2, (DNW) * kWave, st,       \
                               SrcE64{e1}, e0, m, top_hi - 16, cursor16, (const uint32_t*)bins, class_len);                                 \
                    uint64_t* t = e0; e0 = e1; e1 = t;     /* (from here on: e1 = the array grouped by its top 16 bits, e0 = free) */      \
                } else {                                                                                                                    \
                    if (extra)                                                                                                              \
                        SFX_LAUNCH("radix_scatter_text_u32", (double)m * (text.bits / 8.0 + 8.0), (k_partition<SrcText36, kPKpt, DNW, false>), \
                                   g1, (DNW) * kWave, st, SrcText36{text, extra, wide_key}, e0, m, top_hi - 8, cursor8, (const uint32_t*)bins, class_len); \
                    else                                                                                                                    \
                    SFX_LAUNCH("radix_scatter_text_u32", (double)m * (text.bits / 8.0 + 8.0), (k_partition<SrcText32, kPKpt, DNW, false>),  \
                               g1, (DNW) * kWave, st, SrcText32{text}, e0, m, top_hi - 8, cursor8, (const uint32_t*)bins, class_len);        \
                    SFX_LAUNCH("radix_scatter_u32", (double)m * 16.0, (k_partition<SrcE64, kPKpt, DNW, true>), g2, (DNW) * kWave, st,       \
                               SrcE64{e0}, e1, m, top_hi - 16, cursor16, (const uint32_t*)bins, class_len);                                 \
                }                                                                                                                           \
            } while (0)
            if (pwaves == 4) SFX_PARTITION_PAIR(4);
            else SFX_PARTITION_PAIR(8);
#undef SFX_PARTITION_PAIR
        } else
        if (from_elems) {
            SFX_LAUNCH("radix_scatter_u32", (double)m * 16.0, (k_partition<SrcE64, kPKpt, kPNw, false>), grid1, kPNw * kWave, st, SrcE64{e0}, e1,
                       m, top_hi - 8, cursor8, (const uint32_t*)bins, class_len);
            SFX_LAUNCH("radix_scatter_u32", (double)m * 16.0, (k_partition<SrcE64, kPKpt, kPNw, true>), grid2, kPNw * kWave, st, SrcE64{e1}, e0,
                       m, top_hi - 16, cursor16, (const uint32_t*)bins, class_len);
            uint64_t* t = e0; e0 = e1; e1 = t;                 // (from here on: e1 = the array grouped by its top 16 bits, e0 = free)
        } else {
            if (extra)
                SFX_LAUNCH("radix_scatter_text_u32", (double)m * (text.bits / 8.0 + 8.0), (k_partition<SrcText36, kPKpt, kPNw, false>), grid1,
                           kPNw * kWave, st, SrcText36{text, extra, wide_key}, e0, m, top_hi - 8, cursor8, (const uint32_t*)bins, class_len);
            else
            SFX_LAUNCH("radix_scatter_text_u32", (double)m * (text.bits / 8.0 + 8.0), (k_partition<SrcText32, kPKpt, kPNw, false>), grid1,
                       kPNw * kWave, st, SrcText32{text}, e0, m, top_hi - 8, cursor8, (const uint32_t*)bins, class_len);
            SFX_LAUNCH("radix_scatter_u32", (double)m * 16.0, (k_partition<SrcE64, kPKpt, kPNw, true>), grid2, kPNw * kWave, st, SrcE64{e0}, e1,
                       m, top_hi - 16, cursor16, (const uint32_t*)bins, class_len);
        }
    } else if (from_elems) {
        SFX_TRY(run_pass("radix_scatter_u32", (double)m * 16.0, SrcE64{e0}, DstE64{e1}, m, top_hi - 16, 255u, scr, 0, sweep, st));
        SFX_TRY(run_pass("radix_scatter_u32", (double)m * 16.0, SrcE64{e1}, DstE64{e0}, m, top_hi - 8, 255u, scr, 1, sweep, st));
        uint64_t* t = e0; e0 = e1; e1 = t;                     // (from here on: e1 = the array sorted by its top 16 bits, e0 = free)
    } else {
        SrcText32 tsrc = {text};
        SFX_TRY(run_pass("radix_scatter_text_u32", (double)m * (text.bits / 8.0 + 8.0), tsrc, DstE64{e0}, m, top_hi - 16, 255u, scr, 0,
                         sweep, st));
        SFX_TRY(run_pass("radix_scatter_u32", (double)m * 16.0, SrcE64{e0}, DstE64{e1}, m, top_hi - 8, 255u, scr, 1, sweep, st));
    }
    uint32_t* split_k = reinterpret_cast<uint32_t*>(e0);       // (the first half of e0; the oversized sub-buckets are sorted in the second)
    const uint64_t* over_sorted = nullptr;
    if (nover) {
        SFX_LAUNCH("radix_hist16_oversize", (double)kH16Bins * 8, k_hist16_oversize, 1, kH16Threads, st, (const uint32_t*)bins, cap, over);
        uint64_t* T = e0 + (((m + 1) / 2 + 31) & ~uint64_t(31));       // (behind the m u32 keys of split_k)
        uint64_t* T2 = T + ((nlarge + 31) & ~uint64_t(31));
        const unsigned g = (unsigned)dmin<uint64_t>(nover, kMaxGrid);
        SFX_LAUNCH("oversize_gather", (double)nlarge * 16, k_oversize_gather, g, kBlock, st, (const uint64_t*)e1, (const OversizeEntry*)over,
                   nover, low_bits, T);
        // (sorted on the key bits alone: the members of one key come out in the order the partition passes left them in, which
        // depends on atomic timing -- unlike k_bucket_sort, which orders by the whole element.  Nothing downstream reads an order
        // into a run of equal keys: k_groups_reduce compares keys only, the direct pass and the rounds order a bucket's members
        // from the text / the ranks whatever order they arrive in, the fused LCP leaves such pairs pending.  ADVICE round 4.)
        int in1 = 0;
        SFX_TRY(radix_sort_e64(T, T2, nlarge, 32, 32 + low_bits + bits_for(nover > 1 ? nover - 1 : 1), scr.partial, st, &in1, stats,
                               nullptr, nullptr, nullptr, 0));
        over_sorted = in1 ? T2 : T;
    }
    // three size classes, three geometries: 256 threads x 8 (8 workgroups per CU) is the fastest -- measured on 100 MB
    // of DNA with the LSD rounds 0.55 ms against 0.63 (256 x 16) and 0.80 (512 x 8), 0.39 with the grouped all-pairs
    // path -- and takes the sub-buckets of up to 2048 suffixes; 256 x 16 those up to 4096; 1024 x 16 those up to 16384
    // (what the LDS holds).  SFX_HYBRID_GEOM=1 / 2 (tests) gives everything to the second / third.
    static const int force_geom = [] { const char* e = dev_env("SFX_HYBRID_GEOM"); return e ? atoi(e) : -1; }();
    const uint32_t top = dmin(host_max, cap);                  // the largest sub-bucket the LDS sort takes
    const uint32_t c1 = force_geom >= 1 ? 0u : dmin(2048u, cap), c2 = force_geom == 2 ? c1 : dmin(4096u, cap);
    const unsigned grid = (unsigned)dmin<uint64_t>(kH16Bins, (uint64_t)grid_cap() * 2);
    // the mask (m / 32 words + zero words behind them) and two more arrays of the same size for the caller -- in e0, which nobody
    // needs once the elements are in e1
    const uint64_t mask_words = ((m + 31) / 32 + 64) & ~uint64_t(31);
    const bool tie_keys = tie_mode && ties->want_keys;         // (the sorted 32-bit keys too: the first half of e0, the masks behind them)
    uint32_t* const gt = reinterpret_cast<uint32_t*>(e0) + (tie_keys ? ((m + 63) & ~uint64_t(63)) : 0);
    if (tie_keys && (((m + 63) & ~uint64_t(63)) + 3 * mask_words) * sizeof(uint32_t) > m * sizeof(uint64_t)) return SFX_ERR_INTERNAL;
    uint32_t* const gl = gt + mask_words;
    uint32_t* const gh = gl + mask_words;
    if (tie_mode) SFX_HIP(hipMemsetAsync(gt, 0, 2 * mask_words * sizeof(uint32_t), st));  // (the third array is the caller's to fill)
#define SFX_BUCKET_SORT(NW, KPT, LO, HI, GRID)                                                                              \
    do {                                                                                                                    \
        if (tie_mode && tie_keys)                                                                                           \
            SFX_LAUNCH("bucket_sort_ties_keys", (double)m * 16.125, (k_bucket_sort<NW, KPT, true, true>), GRID, NW * kWave, st, (const uint64_t*)e1, \
                       (const uint32_t*)bins, (uint32_t)kH16Bins, low_bits + extra, (uint32_t)(LO), (uint32_t)(HI), split_k, split_v, gt, sbits); \
        else if (tie_mode)                                                                                                  \
            SFX_LAUNCH("bucket_sort_ties", (double)m * 12.125, (k_bucket_sort<NW, KPT, true>), GRID, NW * kWave, st, (const uint64_t*)e1, \
                       (const uint32_t*)bins, (uint32_t)kH16Bins, low_bits + extra, (uint32_t)(LO), (uint32_t)(HI), (uint32_t*)nullptr, split_v, \
                       gt, sbits);                                                                                          \
        else                                                                                                                \
            SFX_LAUNCH("bucket_sort_lds", (double)m * 16.0, (k_bucket_sort<NW, KPT>), GRID, NW * kWave, st, (const uint64_t*)e1, \
                       (const uint32_t*)bins, (uint32_t)kH16Bins, low_bits, (uint32_t)(LO), (uint32_t)(HI), split_k, split_v); \
    } while (0)
    if (c1 > 0) SFX_BUCKET_SORT(4, 8, 0u, c1, grid);
    if (c2 > c1 && top > c1) SFX_BUCKET_SORT(4, 16, c1, c2, grid);
    if (cap > c2 && top > c2) SFX_BUCKET_SORT(16, 16, c2, cap, dmin(grid, grid_cap()));
#undef SFX_BUCKET_SORT
    if (tie_mode) {
        ties->produced = true;
        ties->tmask = gt;
        ties->lmask = gl;
        ties->hmask = gh;
    }
    if (nover) {
        const unsigned g = (unsigned)dmin<uint64_t>(nover, kMaxGrid);
        SFX_LAUNCH("oversize_return", (double)nlarge * 16, k_oversize_return, g, kBlock, st, over_sorted, (const OversizeEntry*)over, nover,
                   low_bits, split_k, split_v);
    }
    if (split_k_out) *split_k_out = (tie_mode && !tie_keys) ? (uint32_t*)nullptr : split_k;
    if (stats) { stats->radix_passes += 2; stats->elements_sorted += 2 * m; }
    *done = true;
    return SFX_OK;
}

// E64 sort on element bits [bit_lo, bit_hi).  With `text` the first pass computes element i
// from the packed text (e0 need not hold anything).  With `split_v` the last pass writes
// the suffix halves to split_v and the key halves to a u32 array carved from whichever of
// e0/e1 it does not read (returned in *split_k_out); otherwise *result_in_1 tells which
// buffer holds the sorted elements.
unsigned radix_e64_presort_hist(uint64_t m, int bit_lo, int bit_hi)
{
    if (m == 0 || bit_hi <= bit_lo || m > 0xFFFFFFFFull) return 0;
    return use_sweep(m, radix_pass_count(bit_lo, bit_hi)) ? kHistAllGrid : 0u;
}
// the bits of the suffix index that scatter_pairs_u32 partitions by, and how many producer workgroups may count
// them (0: the sort counts for itself)
static int scatter_part_bits()
{
    static const int v = [] { const char* e = dev_env("SFX_PARTITION_BITS"); int x = e ? atoi(e) : 24; return x >= 8 && x <= 24 ? x : 24; }();
    return v;
}
unsigned scatter_pairs_presort_hist(uint64_t m, uint64_t n, int* lo_out, int* nb_out)
{
    const int nb = bits_for(n > 1 ? n - 1 : 1);
    const int lo = nb > scatter_part_bits() ? nb - scatter_part_bits() : 0;
    *lo_out = lo;
    *nb_out = nb;
    if (m == 0 || m > 0xFFFFFFFFull) return 0;
    const int npass = radix_pass_count(32 + lo, 32 + nb);
    if (!use_sweep(m, npass)) return 0;
    return (unsigned)((uint64_t)kMaxPasses * kHistAllGrid / (unsigned)npass);   // [npass][256][workgroups] fits the partials
}

int radix_sort_e64(uint64_t* e0, uint64_t* e1, uint64_t m, int bit_lo, int bit_hi, uint32_t* scratch, hipStream_t st,
                   int* result_in_1, sfx_build_stats* stats, const PackedText* text, uint32_t* split_v,
                   uint32_t** split_k_out, unsigned hist_blocks, int elem_bits, TieRecords* ties)
{
    if (ties) ties->produced = false;
    // elem_bits (with split_v, without text): the keys of the elements in e0 are all below 2^elem_bits -- a slice of the
    // partitioned build; lets the hybrid route take it
    *result_in_1 = 0;
    if (split_k_out) *split_k_out = (uint32_t*)e1;
    if (m == 0) return SFX_OK;
    if (bit_hi <= bit_lo) return (text || split_v) ? SFX_ERR_INTERNAL : SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    const int npass = radix_pass_count(bit_lo, bit_hi);
    const bool sweep = use_sweep(m, npass);
    RadixScratch scr(scratch, m);
    bool windows_ready = false;                                  // the digit totals of all passes are in place already
    if (sweep && text && split_v && npass >= 3) {
        bool done = false;
        SFX_TRY(hybrid_sort_e64_text(e0, e1, m, bit_lo, bit_hi, scr, st, stats, *text, split_v, split_k_out, &done, &windows_ready, 0, ties));
        if (done) return SFX_OK;
    }
    if (sweep && !text && split_v && npass >= 3 && elem_bits > 0) {
        bool done = false, unused = false;
        SFX_TRY(hybrid_sort_e64_text(e0, e1, m, bit_lo, bit_hi, scr, st, stats, PackedText{nullptr, 0, 0, 1, 0, 1.0}, split_v, split_k_out,
                                     &done, &unused, elem_bits, ties));
        if (done) return SFX_OK;
    }
    SrcText32 tsrc = {text ? *text : PackedText{nullptr, 0, 0, 1, 0, 1.0}};
    if (sweep && hist_blocks && !text) {
        // the producer of e0 counted the digits (radix_e64_presort_hist): only the row sums are left
        if ((uint64_t)hist_blocks * (unsigned)npass > (uint64_t)kMaxPasses * kHistAllGrid) return SFX_ERR_INTERNAL;
        SFX_HIP(hipMemsetAsync(scr.tickets, 0, 64 * sizeof(uint32_t), st));
        SFX_LAUNCH("radix_scan", (double)npass * kRadix * hist_blocks * 8, k_radix_scan, npass * kRadix, kBlock, st,
                   scr.partial, hist_blocks, scr.totals);
    } else if (sweep && windows_ready) {
        // (hybrid_sort_e64_text gave way and left the totals)
    } else if (sweep) {
        if (window_hist_applies(text, m, bit_lo, bit_hi)) SFX_TRY(prepare_sweep_windows(*text, scr, st));
        else if (text) SFX_TRY(prepare_sweep("radix_hist_all_text_u32", (double)m * text->bits / 8.0, tsrc, m, bit_lo, bit_hi, npass, scr, st));
        else SFX_TRY(prepare_sweep("radix_hist_all_u32", (double)m * 8.0, SrcE64{e0}, m, bit_lo, bit_hi, npass, scr, st));
    }
    uint64_t* cur = text ? nullptr : e0;        // the text-fed pass reads no element buffer
    uint64_t* nxt = text ? e0 : e1;
    for (int p = 0; p < npass; p++) {
        const int shift = bit_lo + p * kRadixBits;
        const int nb = bit_hi - shift < kRadixBits ? bit_hi - shift : kRadixBits;
        const unsigned mask = (1u << nb) - 1u;
        const bool first_text = text && p == 0;
        const bool last_split = split_v && p == npass - 1;
        const double in_bytes = first_text ? text->bits / 8.0 : 8.0;
        const char* name = first_text ? "radix_scatter_text_u32" : "radix_scatter_u32";
        const double algo = (double)m * (in_bytes + 8.0);
        uint32_t* split_k = (uint32_t*)nxt;
        if (first_text && last_split) {
            SFX_TRY(run_pass(name, algo, tsrc, DstSplit32{split_k, split_v}, m, shift, mask, scr, p, sweep, st));
        } else if (first_text) {
            SFX_TRY(run_pass(name, algo, tsrc, DstE64{nxt}, m, shift, mask, scr, p, sweep, st));
        } else if (last_split) {
            SFX_TRY(run_pass(name, algo, SrcE64{cur}, DstSplit32{split_k, split_v}, m, shift, mask, scr, p, sweep, st));
        } else {
            SFX_TRY(run_pass(name, algo, SrcE64{cur}, DstE64{nxt}, m, shift, mask, scr, p, sweep, st));
        }
        if (last_split) {
            if (split_k_out) *split_k_out = split_k;
        } else {
            uint64_t* other = (nxt == e0) ? e1 : e0;
            cur = nxt;
            nxt = other;
        }
        if (stats) { stats->radix_passes++; stats->elements_sorted += m; }
    }
    *result_in_1 = (cur == e1) ? 1 : 0;
    return SFX_OK;
}

// ---- compressed keys of a whole text ---------------------------------------------------------------
// key of position i = the codes of symbols i, i + 1, ... (at most kHtMaxSym of them) cut to 64 bits (sfx_device.hpp).
// A workgroup takes tiles of kHtTile positions and first lays the codes of the tile's symbols (+ the kHtPad beyond it) end to
// end as ONE BIT STRING in LDS: a thread concatenates the codes of kHtRun consecutive symbols in registers (at most 96 bits), a
// block-wide scan of the run lengths gives every run its bit offset, the runs are OR-ed into the zeroed string (four LDS
// atomics per run) and the offset of every symbol is kept (16 bits).  The key of position i is then the 64-bit WINDOW of
// the string at symbol i's offset, cut where symbol i + kHtMaxSym starts: three LDS words and two funnel shifts per key, any
// thread for any position -- so the keys leave the registers in coalesced order, and the digit counts of all eight radix
// passes are taken on the way out (the role of k_radix_hist_all).  (Rounds 4-6 kept a rolling 128-bit buffer per thread over 8
// consecutive positions: ~90 VALU instructions per key in a divergent refill loop with an LDS round trip per symbol, the
// kernel 90 % VALU-busy at 2.5x its HBM time.)
constexpr int kHtRun = 8;
constexpr int kHtPad = 16;                                   // >= kHtMaxSym: the symbols beyond the tile its last keys hold
constexpr int kHtSpan = kBlock * kHtRun;                     // 2048 symbols laid out per tile: one run per thread
constexpr int kHtTile = kHtSpan - kHtPad;                    // 2032 positions (16 256 bytes of keys: whole 128-byte lines)
constexpr int kHtStreamWords = kHtSpan * kHtMaxLen / 32 + 4; // the bit string (+ the window's two words of slack)
static_assert(kHtPad >= (int)kHtMaxSym && kHtRun * kHtMaxLen <= 96, "a run fits 96 bits");
// the digits of all passes of one key into the tile's counters: with all 64 bits sorted, eight byte extractions from the two
// halves (a loop over a run-time pass count shifts 64 bits by a variable eight times)
__device__ __forceinline__ void ht_count_digits(uint32_t (*h)[kRadix], uint64_t key, int npass)
{
    if (kHtKeyBits == 64 && npass == 8) {
        const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            atomicAdd(&h[p][(klo >> (8 * p)) & 255u], 1u);
            atomicAdd(&h[p + 4][(khi >> (8 * p)) & 255u], 1u);
        }
    } else {
        for (int p = 0; p < npass; p++) atomicAdd(&h[p][(unsigned)(key >> (64 - kHtKeyBits + 8 * p)) & 255u], 1u);
    }
}
// exclusive block-wide prefix sum of v (s_wsum: one word per wave; one barrier)
__device__ __forceinline__ unsigned ht_block_scan(unsigned v, uint32_t* s_wsum, unsigned tid)
{
    const unsigned lane = tid & (kWave - 1);
    unsigned incl = v;
#pragma unroll
    for (unsigned d = 1; d < (unsigned)kWave; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == (unsigned)kWave - 1u) s_wsum[tid / kWave] = incl;
    __syncthreads();
    unsigned add = 0;
#pragma unroll
    for (unsigned w = 0; w < (unsigned)(kBlock / kWave); w++)
        if (w < tid / kWave) add += s_wsum[w];
    return add + incl - v;
}
// bits [S, S + 96) of the LDS bit string (big-endian inside its 32-bit words) |= a run, left-aligned in hi:lo (<= 96 bits)
__device__ __forceinline__ void ht_stream_or(uint32_t* stream, unsigned S, uint64_t hi, uint64_t lo)
{
    const unsigned w = S >> 5, sh = S & 31u;
    const uint32_t x0 = (uint32_t)(hi >> 32), x1 = (uint32_t)hi, x2 = (uint32_t)(lo >> 32);
    atomicOr(&stream[w], x0 >> sh);
    atomicOr(&stream[w + 1], (uint32_t)((((uint64_t)x0 << 32) | x1) >> sh));
    atomicOr(&stream[w + 2], (uint32_t)((((uint64_t)x1 << 32) | x2) >> sh));
    atomicOr(&stream[w + 3], (uint32_t)(((uint64_t)x2 << 32) >> sh));
}
// the 64 bits of the string from bit o on
__device__ __forceinline__ uint64_t ht_stream_window(const uint32_t* stream, unsigned o)
{
    const unsigned w = o >> 5, sh = o & 31u;
    const uint64_t w0 = stream[w], w1 = stream[w + 1], w2 = stream[w + 2];
    return (((w0 << 32) | w1) << sh & 0xFFFFFFFF00000000ull) | ((((w1 << 32) | w2) << sh) >> 32);
}
// one code (c: left-aligned in 64 bits, len bits) behind the pos bits a run holds so far
__device__ __forceinline__ void ht_run_append(uint64_t& hi, uint64_t& lo, unsigned pos, uint64_t c, unsigned len)
{
    if (pos < 64u) {
        hi |= c >> pos;
        if (pos + len > 64u) lo |= c << (64u - pos);          // (pos >= 52 here: the shift is < 64)
    } else {
        lo |= c >> (pos - 64u);
    }
}
__global__ void __launch_bounds__(kBlock)
k_ht_keys(PackedText t, const uint32_t* __restrict__ ent, uint64_t m, uint64_t tiles_per_block, int npass,
          uint64_t* __restrict__ K, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t s_tab[256];
    __shared__ uint32_t s_words[kHtSpan / 4 + 4];                    // the span's packed words (at least 4 symbols per word)
    __shared__ uint32_t s_stream[kHtStreamWords];
    __shared__ __attribute__((aligned(16))) uint16_t s_off[kHtSpan];
    __shared__ uint32_t s_wsum[kBlock / kWave];
    __shared__ uint32_t h[kMaxPasses][kRadix];                       // (one copy: the key bytes are spread evenly)
    const unsigned tid = threadIdx.x;
    s_tab[tid] = ent[tid];
    for (unsigned i = tid; i < kMaxPasses * kRadix; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    const unsigned bits = (unsigned)t.bits;
    const uint32_t smask = (1u << bits) - 1u;
    const float inv_spw = 1.0f / (float)t.spw;
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_block;
    for (uint64_t tile = tile0; tile < tile0 + tiles_per_block; tile++) {
        const uint64_t base = tile * kHtTile;
        if (base >= m) break;
        // the span's packed words, two or three independent loads per thread; the bit string zeroed
        const uint64_t q0 = packed_word_index(t, base);
        const uint64_t qlast = (t.n + (uint64_t)t.spw - 1) / (uint64_t)t.spw + 2;         // (three zero words behind the text)
        const unsigned nw = (unsigned)(packed_word_index(t, base + kHtSpan - 1) - q0) + 1u;
#pragma unroll
        for (int k = 0; k < (int)((kHtSpan / 4 + 4 + kBlock - 1) / kBlock); k++) {
            const unsigned w = tid + (unsigned)k * kBlock;
            if (w < nw) s_words[w] = q0 + w <= qlast ? t.words[q0 + w] : 0u;
        }
        for (unsigned w = tid; w < (unsigned)kHtStreamWords; w += kBlock) s_stream[w] = 0u;
        __syncthreads();
        {
            // the run of symbols [8 tid, 8 tid + 8): word and offset of the first in single precision (exact for the < 2^12
            // symbols of a span), a step per symbol; positions past the text read as the padding's zero symbol
            const unsigned off0 = (unsigned)(base - q0 * (uint64_t)t.spw);
            const unsigned lim = (unsigned)dmin<uint64_t>(t.n - base, 0xFFFFFFu);
            const unsigned i0 = tid * (unsigned)kHtRun;
            unsigned wq = (unsigned)(((float)(off0 + i0) + 0.5f) * inv_spw);
            unsigned off = off0 + i0 - wq * (unsigned)t.spw;
            uint64_t hi = 0, lo = 0;
            unsigned pos = 0;
            uint32_t offs[kHtRun];
#pragma unroll
            for (int k = 0; k < kHtRun; k++) {
                const uint32_t e = s_tab[i0 + (unsigned)k < lim ? (s_words[wq] >> (((unsigned)t.spw - 1u - off) * bits)) & smask : 0u];
                if (++off == (unsigned)t.spw) { off = 0; wq++; }
                const unsigned len = e & 31u;
                offs[k] = pos;
                ht_run_append(hi, lo, pos, (uint64_t)(e & ~31u) << 32, len);
                pos += len;
            }
            const unsigned S = ht_block_scan(pos, s_wsum, tid);
            uint4 o;
            o.x = (S + offs[0]) | ((S + offs[1]) << 16);
            o.y = (S + offs[2]) | ((S + offs[3]) << 16);
            o.z = (S + offs[4]) | ((S + offs[5]) << 16);
            o.w = (S + offs[6]) | ((S + offs[7]) << 16);
            __builtin_memcpy(&s_off[i0], &o, sizeof(o));                 // (16-byte aligned: one LDS store)
            ht_stream_or(s_stream, S, hi, lo);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kHtRun; k++) {
            const unsigned i = tid + (unsigned)k * kBlock;
            if (i < (unsigned)kHtTile && base + i < m) {
                const unsigned o = s_off[i], nb = (unsigned)s_off[i + kHtMaxSym] - o;      // (symbol i + kHtMaxSym stays out)
                uint64_t key = ht_stream_window(s_stream, o);
                if (nb < 64u) key &= ~0ull << (64u - nb);
                if (kHtKeyBits < 64) key &= ~((1ull << (64 - kHtKeyBits)) - 1ull);       // (the bits that are sorted)
                K[base + i] = key;
                if (partial) ht_count_digits(h, key, npass);
            }
        }
        __syncthreads();
    }
    if (partial)
        for (int p = 0; p < npass; p++) partial[((uint64_t)p * kRadix + tid) * gridDim.x + blockIdx.x] = h[p][tid];
}

// ---- compressed keys over CONTEXTS (round 6) ---------------------------------------------------------------------------
// An order-preserving prefix code per class of the PRECEDING symbol: the key of position i = the order-0 code of symbol i, then
// for j = i + 1, i + 2, ... the code of symbol j in the table of class(symbol j - 1).  Equal prefixes of two suffixes have equal
// contexts, so the concatenation is still order-preserving and prefix-free; and the stream from i + 1 on does not depend on i,
// so k_ht_keys' bit string survives: it holds the stream, the key is the first symbol's order-0 code and the window behind
// symbol i.  Mixed-script UTF-8 (config 5): 5.0 instead of 6.4 bits per symbol with 16 classes.  How many WHOLE code words a
// key holds cannot be read off it with the order-0 end-mask table any more, so the kernel counts them where it knows them -- a
// second bit string with one bit per code-word END, the population count of ITS window -- and stores the count in the key's
// LOW 4 BITS (kHtCtxCountBits; the code string takes the top 60): equal 60-bit prefixes of a prefix code hold the same code
// words, so order and tie classes are what the 60 bits alone give, and a bucket's depth is key & 15 (k_groups_apply).
// tab: [0, 256) order-0 entries, then at kHtCtxOff: class of every dense symbol (256 bytes), then 16 x 256 class entries.
constexpr unsigned kHtCtxMaxSym = 15;                         // symbols a context key is made from at most (the count's range)
static_assert(kHtPad >= (int)kHtCtxMaxSym, "the stream symbols of the tile's last key");
__global__ void __launch_bounds__(kBlock)
k_ht_keys_ctx(PackedText t, const uint32_t* __restrict__ tab, int sigma, uint64_t m, uint64_t tiles_per_block, int npass,
              uint64_t* __restrict__ K, uint32_t* __restrict__ partial)
{
    __shared__ uint32_t s_tab0[256];
    __shared__ uint8_t s_cls[256];
    __shared__ uint32_t s_tab1[kHtCtxClasses * kHtCtxSigmaMax];
    __shared__ uint32_t s_words[kHtSpan / 4 + 8];
    __shared__ __attribute__((aligned(8))) uint8_t s_sym[kHtSpan];
    __shared__ uint32_t s_stream[kHtStreamWords];
    __shared__ uint32_t s_ends[kHtStreamWords];
    __shared__ __attribute__((aligned(16))) uint16_t s_off[kHtSpan];
    __shared__ uint32_t s_wsum[kBlock / kWave];
    __shared__ uint32_t h[kMaxPasses][kRadix];
    const unsigned tid = threadIdx.x;
    s_tab0[tid] = tab[tid];
    s_cls[tid] = reinterpret_cast<const uint8_t*>(tab + kHtCtxOff)[tid];
    for (unsigned i = tid; i < (unsigned)(kHtCtxClasses * sigma); i += kBlock)
        s_tab1[i] = tab[kHtCtxOff + 64 + (i / (unsigned)sigma) * 256u + (i % (unsigned)sigma)];
    for (unsigned i = tid; i < kMaxPasses * kRadix; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    const unsigned bits = (unsigned)t.bits;
    const uint32_t smask = (1u << bits) - 1u;
    const float inv_spw = 1.0f / (float)t.spw;
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_block;
    for (uint64_t tile = tile0; tile < tile0 + tiles_per_block; tile++) {
        const uint64_t base = tile * kHtTile;
        if (base >= m) break;
        // the span's packed words (from the symbol BEFORE the tile on: the first stream code needs its context)
        const uint64_t first = base ? base - 1 : 0;
        const uint64_t q0 = packed_word_index(t, first);
        const uint64_t qlast = (t.n + (uint64_t)t.spw - 1) / (uint64_t)t.spw + 2;         // (three zero words behind the text)
        const unsigned nw = (unsigned)(packed_word_index(t, base + kHtSpan - 1) - q0) + 1u;
#pragma unroll
        for (int k = 0; k < (int)((kHtSpan / 4 + 8 + kBlock - 1) / kBlock); k++) {
            const unsigned w = tid + (unsigned)k * kBlock;
            if (w < nw) s_words[w] = q0 + w <= qlast ? t.words[q0 + w] : 0u;
        }
        for (unsigned w = tid; w < (unsigned)kHtStreamWords; w += kBlock) { s_stream[w] = 0u; s_ends[w] = 0u; }
        __syncthreads();
        {
            // the run of span symbols [8 tid, 8 tid + 8), from the symbol before it on (its first context; before position 0:
            // the class of symbol 0).  Span symbol j = position base + j; li = its index among the symbols of s_words
            const unsigned off0 = (unsigned)(first - q0 * (uint64_t)t.spw);
            const unsigned lead = base ? 1u : 0u;            // span symbol 0 sits at li = off0 + lead
            const unsigned lim = (unsigned)dmin<uint64_t>(t.n - base, 0xFFFFFFu);      // span symbols inside the text
            const unsigned i0 = tid * (unsigned)kHtRun;
            unsigned prev = 0;
            unsigned wq = 0, off = 0;
            if (i0 + lead > 0u) {
                const unsigned li = off0 + i0 + lead - 1u;   // the symbol before the run
                wq = (unsigned)(((float)li + 0.5f) * inv_spw);
                off = li - wq * (unsigned)t.spw;
                prev = i0 <= lim ? (s_words[wq] >> (((unsigned)t.spw - 1u - off) * bits)) & smask : 0u;   // (position base + i0 - 1 < n)
                if (++off == (unsigned)t.spw) { off = 0; wq++; }
            }
            uint64_t hi = 0, lo = 0, ehi = 0, elo = 0;
            unsigned pos = 0;
            uint32_t offs[kHtRun];
            uint64_t syms = 0;
#pragma unroll
            for (int k = 0; k < kHtRun; k++) {
                const unsigned sym = i0 + (unsigned)k < lim ? (s_words[wq] >> (((unsigned)t.spw - 1u - off) * bits)) & smask : 0u;
                if (++off == (unsigned)t.spw) { off = 0; wq++; }
                const uint32_t e = s_tab1[(unsigned)s_cls[prev] * (unsigned)sigma + sym];   // symbol j after symbol j - 1
                prev = sym;
                syms |= (uint64_t)sym << (8 * k);
                const unsigned len = e & 31u;
                offs[k] = pos;
                ht_run_append(hi, lo, pos, (uint64_t)(e & ~31u) << 32, len);
                pos += len;
                ht_run_append(ehi, elo, pos - 1u, 1ull << 63, 1u);                         // (one bit where the code word ends)
            }
            *reinterpret_cast<uint64_t*>(&s_sym[i0]) = syms;
            const unsigned S = ht_block_scan(pos, s_wsum, tid);
            uint4 o;
            o.x = (S + offs[0]) | ((S + offs[1]) << 16);
            o.y = (S + offs[2]) | ((S + offs[3]) << 16);
            o.z = (S + offs[4]) | ((S + offs[5]) << 16);
            o.w = (S + offs[6]) | ((S + offs[7]) << 16);
            __builtin_memcpy(&s_off[i0], &o, sizeof(o));                 // (16-byte aligned: one LDS store)
            ht_stream_or(s_stream, S, hi, lo);
            ht_stream_or(s_ends, S, ehi, elo);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kHtRun; k++) {
            const unsigned i = tid + (unsigned)k * kBlock;
            if (i < (unsigned)kHtTile && base + i < m) {
                // the stream behind symbol i: symbols i + 1 .. i + kHtCtxMaxSym - 1
                const unsigned o = s_off[i + 1u], nb = (unsigned)s_off[i + kHtCtxMaxSym] - o;
                uint64_t win = ht_stream_window(s_stream, o);
                if (nb < 64u) win = nb ? win & (~0ull << (64u - nb)) : 0ull;
                const uint32_t e0 = s_tab0[s_sym[i]];
                const unsigned len0 = e0 & 31u;
                const unsigned room = 64u - (unsigned)kHtCtxCountBits - len0;     // stream bits that fit behind the first code
                const unsigned have = nb < room ? nb : room;                     // (fewer only where the symbol cap cut the stream)
                const uint64_t code = ((uint64_t)(e0 & ~31u) << 32) | (win >> len0);
                unsigned cnt = 1u + (have ? (unsigned)__popcll(ht_stream_window(s_ends, o) >> (64u - have)) : 0u);
                if (cnt > kHtCtxMaxSym) cnt = kHtCtxMaxSym;
                const uint64_t key = (code & ~((1ull << kHtCtxCountBits) - 1ull)) | (uint64_t)cnt;
                K[base + i] = key;
                if (partial) ht_count_digits(h, key, npass);
            }
        }
        __syncthreads();
    }
    if (partial)
        for (int p = 0; p < npass; p++) partial[((uint64_t)p * kRadix + tid) * gridDim.x + blockIdx.x] = h[p][tid];
}

// ---- 64-bit keys: the passes between the first and the last move 12-byte (key, suffix) elements (KV12 above) -----------------
// (k, v) can serve as ONE array of m <= cap 12-byte elements when the caller carved it as cap keys followed by cap values
// (kv12_cap of radix_sort_kv64 / radix_sort_ht64: a statement of the caller, not something inferred from pointer distances --
// ADVICE round 5).  What is checked here is that the statement holds: v behind k's cap keys with at most one arena alignment
// gap, and the two regions [k0, v0 + cap) and [k1, v1 + cap) disjoint.
static bool kv12_pair(const uint64_t* k0, const uint32_t* v0, const uint64_t* k1, const uint32_t* v1, uint64_t m, uint64_t cap)
{
    if (cap == 0 || m > cap) return false;
    auto region = [cap](const uint64_t* k, const uint32_t* v, uintptr_t* lo, uintptr_t* hi) {
        const uintptr_t ke = reinterpret_cast<uintptr_t>(k + cap), vb = reinterpret_cast<uintptr_t>(v);
        *lo = reinterpret_cast<uintptr_t>(k);
        *hi = reinterpret_cast<uintptr_t>(v + cap);
        return vb >= ke && vb - ke < kArenaAlign;
    };
    uintptr_t lo0, hi0, lo1, hi1;
    if (!region(k0, v0, &lo0, &hi0) || !region(k1, v1, &lo1, &hi1)) return false;
    return hi0 <= lo1 || hi1 <= lo0;
}
static void kv_trace(uint64_t m, int npass, bool e12)                  // SFX_TRACE=1 (development)
{
    static const bool trace = [] { const char* e = dev_env("SFX_TRACE"); return e && atoi(e) != 0; }();
    if (trace) fprintf(stderr, "[sfx] 64-bit-key sort: m=%llu passes=%d elements=%s\n", (unsigned long long)m, npass, e12 ? "kv12" : "k+v");
}
template <class Src>
static int kv_pass_out(const char* name, double algo, const Src& src, bool out12, KV12* out_e, uint64_t* out_k, uint32_t* out_v,
                       uint64_t m, int shift, unsigned mask, const RadixScratch& scr, int pass, bool sweep, hipStream_t st)
{
    if (out12) return run_pass(name, algo, src, DstKV12{out_e}, m, shift, mask, scr, pass, sweep, st);
    return run_pass(name, algo, src, DstKV{out_k, out_v}, m, shift, mask, scr, pass, sweep, st);
}

// Sort of all m = text.n suffixes by their compressed 64-bit keys (eight passes); (k0, v0) / (k1, v1) as
// radix_sort_kv64, the keys are made here.
// last_v (both sorts): the suffixes of the LAST pass go there instead of into v0 / v1 (the caller's SA: every suffix
// lands in its slot without a copy); the keys still end in k0 / k1 as *result_in_1 says.
int radix_sort_ht64(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, uint64_t m, uint32_t* scratch, hipStream_t st,
                    int* result_in_1, sfx_build_stats* stats, const PackedText& text, const uint32_t* ht, uint32_t* last_v,
                    uint64_t kv12_cap, int ctx_sigma)
{
    *result_in_1 = 0;
    if (m == 0) return SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    const int npass = kHtKeyBits / 8;                          // (the low 64 - kHtKeyBits bits of the keys are zero)
    const int bit0 = 64 - kHtKeyBits;
    const bool sweep = use_sweep(m, npass);
    RadixScratch scr(scratch, m);
    {
        Chunking ch = make_chunking(m, kHtTile, kHistAllGrid);
        SFX_HIP(hipMemsetAsync(scr.tickets, 0, 64 * sizeof(uint32_t), st));
        if (ctx_sigma > 0) {
            if (ctx_sigma > kHtCtxSigmaMax || kHtKeyBits != 64) return SFX_ERR_INTERNAL;
            SFX_LAUNCH("ht_keys", (double)m * (text.bits / 8.0 + 8.0), k_ht_keys_ctx, ch.blocks, kBlock, st, text, ht, ctx_sigma, m, ch.tiles_per_block,
                       npass, k0, sweep ? scr.partial : (uint32_t*)nullptr);
        } else
        SFX_LAUNCH("ht_keys", (double)m * (text.bits / 8.0 + 8.0), k_ht_keys, ch.blocks, kBlock, st, text, ht, m, ch.tiles_per_block, npass, k0,
                   sweep ? scr.partial : (uint32_t*)nullptr);
        if (sweep)
            SFX_LAUNCH("radix_scan", (double)npass * kRadix * ch.blocks * 8, k_radix_scan, npass * kRadix, kBlock, st, scr.partial,
                       ch.blocks, scr.totals);
    }
    // passes 1 .. npass - 2 read and write 12-byte elements, the first writes them, the last reads them (and leaves the sorted
    // keys as an array of their own, the suffixes in last_v / the value array of that side): region "0" = k0 + v0, "1" = k1 + v1
    const bool e12 = radix_tuning().kv12 && npass >= 2 && kv12_pair(k0, v0, k1, v1, m, kv12_cap);
    kv_trace(m, npass, e12);
    uint64_t* kin = k0; uint32_t* vin = v0;
    uint64_t* kout = k1; uint32_t* vout = v1;
    int flips = 0;
    for (int p = 0; p < npass; p++) {
        const double algo = (double)m * ((p == 0 ? 8.0 : 12.0) + 12.0);
        uint32_t* vdst = (last_v && p == npass - 1) ? last_v : vout;
        const bool out12 = e12 && p < npass - 1;
        KV12* const ein = reinterpret_cast<KV12*>(kin);
        KV12* const eout = reinterpret_cast<KV12*>(kout);
        const int shift = bit0 + 8 * p;
        if (p == 0) SFX_TRY(kv_pass_out("radix_scatter_u64", algo, SrcKeyIota{kin}, out12, eout, kout, vdst, m, shift, 255u, scr, p, sweep, st));
        else if (e12) SFX_TRY(kv_pass_out("radix_scatter_u64", algo, SrcKV12{ein}, out12, eout, kout, vdst, m, shift, 255u, scr, p, sweep, st));
        else SFX_TRY(kv_pass_out("radix_scatter_u64", algo, SrcKV{kin, vin}, false, eout, kout, vdst, m, shift, 255u, scr, p, sweep, st));
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        flips ^= 1;
        if (stats) { stats->radix_passes++; stats->elements_sorted += m; }
    }
    *result_in_1 = flips;
    return SFX_OK;
}

int radix_sort_kv64(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, uint64_t m, int bit_lo, int bit_hi,
                    uint32_t* scratch, hipStream_t st, int* result_in_1, sfx_build_stats* stats,
                    const PackedText* text, uint32_t* last_v, uint64_t kv12_cap)
{
    *result_in_1 = 0;
    if (m == 0 || bit_hi <= bit_lo) return last_v ? SFX_ERR_INTERNAL : SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    const int npass = radix_pass_count(bit_lo, bit_hi);
    const bool sweep = use_sweep(m, npass);
    RadixScratch scr(scratch, m);
    SrcText64 tsrc = {text ? *text : PackedText{nullptr, 0, 0, 1, 0, 1.0}};
    if (sweep) {
        if (text) SFX_TRY(prepare_sweep("radix_hist_all_text_u64", (double)m * text->bits / 8.0, tsrc, m, bit_lo, bit_hi, npass, scr, st));
        else SFX_TRY(prepare_sweep("radix_hist_all_u64", (double)m * 8.0, SrcKV{k0, v0}, m, bit_lo, bit_hi, npass, scr, st));
    }
    // (12-byte elements between the first and the last pass: see radix_sort_ht64)
    const bool e12 = radix_tuning().kv12 && npass >= 2 && kv12_pair(k0, v0, k1, v1, m, kv12_cap);
    kv_trace(m, npass, e12);
    uint64_t* kin = k0; uint32_t* vin = v0;
    uint64_t* kout = k1; uint32_t* vout = v1;
    int flips = 0;
    for (int p = 0; p < npass; p++) {
        const int shift = bit_lo + p * kRadixBits;
        const int nb = bit_hi - shift < kRadixBits ? bit_hi - shift : kRadixBits;
        const unsigned mask = (1u << nb) - 1u;
        const bool first_text = text && p == 0;
        const double in_bytes = first_text ? text->bits / 8.0 : 12.0;
        const double algo = (double)m * (in_bytes + 12.0);
        uint32_t* vdst = (last_v && p == npass - 1) ? last_v : vout;
        const bool out12 = e12 && p < npass - 1;
        KV12* const ein = reinterpret_cast<KV12*>(kin);
        KV12* const eout = reinterpret_cast<KV12*>(kout);
        if (first_text) {
            SFX_TRY(kv_pass_out("radix_scatter_text_u64", algo, tsrc, out12, eout, kout, vdst, m, shift, mask, scr, p, sweep, st));
        } else if (e12 && p > 0) {
            SFX_TRY(kv_pass_out("radix_scatter_u64", algo, SrcKV12{ein}, out12, eout, kout, vdst, m, shift, mask, scr, p, sweep, st));
        } else {
            SFX_TRY(kv_pass_out("radix_scatter_u64", algo, SrcKV{kin, vin}, out12, eout, kout, vdst, m, shift, mask, scr, p, sweep, st));
        }
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        flips ^= 1;
        if (stats) { stats->radix_passes++; stats->elements_sorted += m; }
    }
    *result_in_1 = flips;
    return SFX_OK;
}

// ---- segmented sort of the large buckets of a refinement round ---------------------------------
// segs[k] = (start, size) of bucket k (list positions, any order).  The elements E[p] = key2 << 32 |
// suffix of those positions are sorted by key2 inside every bucket: four one-sweep passes over 16 bytes
// per element -- against eight passes over 24 bytes for the composite (bucket id, key2) key of round 1,
// and no extraction or write-back of a sub-list.
constexpr int kSegKPT = 11, kSegNW = 16;                  // the E64 geometry: 11264-element tiles
constexpr int kSegSmallKPT = 16, kSegSmallNW = 4;         // SFX_SEG_SMALL=1 (tests): 4096-element tiles
static bool seg_small()
{
    static const bool v = [] { const char* e = dev_env("SFX_SEG_SMALL"); return e && atoi(e) != 0; }();
    return v;
}
// counters: [0] tiles, [1] tiles of multi-tile segments, [2] multi-tile segments
__global__ void __launch_bounds__(kBlock)
k_seg_layout(const uint2* __restrict__ segs, uint32_t nseg, uint32_t tile_elems, SegTile* __restrict__ tiles,
             uint32_t* __restrict__ counters, uint32_t skip_upto)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= nseg) return;
    const uint32_t start = segs[k].x, size = segs[k].y;
    const uint32_t nt = (size + tile_elems - 1) / tile_elems;
    if (size <= skip_upto) return;                                    // (sorted inside LDS by k_seg_single, sfx_tile.hip)
    const uint32_t t0 = atomicAdd(&counters[0], nt);
    if (nt == 1) {
        tiles[t0] = SegTile{start, size, start, 3u, 0u, {1u, size, 0u}};
        return;
    }
    const uint32_t mt0 = atomicAdd(&counters[1], nt);
    const uint32_t ms = atomicAdd(&counters[2], 1u);
    for (uint32_t j = 0; j < nt; j++) {
        const uint32_t b = j * tile_elems;
        tiles[t0 + j] = SegTile{start + b, dmin(tile_elems, size - b), start, (j == 0 ? 1u : 0u) | ((mt0 + j) << 2), ms,
                                {nt, size, 0u}};
    }
}
// digit counts of every pass for every tile of a multi-tile segment: tilehist[mt][npass][256]
// (digit p of an element = bits [shift0 + 8p, shift0 + 8p + 8) of its 64-bit key word)
constexpr int kSegMaxPasses = 8;
__global__ void __launch_bounds__(kBlock)
k_seg_hist(const uint64_t* __restrict__ E, const SegTile* __restrict__ tiles, const uint32_t* __restrict__ ntiles,
           uint32_t* __restrict__ tilehist, int npass, int shift0)
{
    __shared__ uint32_t h[kSegMaxPasses][kRadix];
    const unsigned tid = threadIdx.x;
    const uint32_t nt = *ntiles;
    for (uint32_t t = blockIdx.x; t < nt; t += gridDim.x) {
        const SegTile d = tiles[t];
        if (d.info & 2u) continue;
        for (int p = 0; p < npass; p++) h[p][tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < d.count; i += kBlock) {
            const uint64_t key = E[(uint64_t)d.begin + i] >> shift0;
            for (int p = 0; p < npass; p++) atomicAdd(&h[p][(unsigned)(key >> (8 * p)) & 255u], 1u);
        }
        __syncthreads();
        uint32_t* out = tilehist + (uint64_t)(d.info >> 2) * npass * kRadix;
        for (int p = 0; p < npass; p++) out[p * kRadix + tid] = h[p][tid];
        __syncthreads();
    }
}
// per multi-tile segment and pass: elements of the segment with a smaller digit
__global__ void __launch_bounds__(kBlock)
k_seg_scan(const SegTile* __restrict__ tiles, const uint32_t* __restrict__ ntiles, const uint32_t* __restrict__ tilehist,
           uint32_t* __restrict__ segexcl, int npass)
{
    __shared__ uint32_t part[kWavesPerBlock];
    const unsigned tid = threadIdx.x;
    const uint32_t nt = *ntiles;
    for (uint32_t t = blockIdx.x; t < nt; t += gridDim.x) {
        const SegTile d = tiles[t];
        if ((d.info & 3u) != 1u) continue;                         // first tile of a multi-tile segment
        const uint32_t mt0 = d.info >> 2, cnt = d.pad[0];
        for (int p = 0; p < npass; p++) {
            uint32_t tot = 0;
            for (uint32_t j = 0; j < cnt; j++) tot += tilehist[((uint64_t)(mt0 + j) * npass + p) * kRadix + tid];
            uint32_t total;
            const uint32_t ex = block_scan_add_excl(tot, part, total);
            segexcl[((uint64_t)d.mseg * npass + p) * kRadix + tid] = ex;
        }
    }
}
// suffixes back to the list, head / singleton flags from the sorted key2 values
__global__ void __launch_bounds__(kBlock)
k_seg_finish(const uint64_t* __restrict__ E, const SegTile* __restrict__ tiles, const uint32_t* __restrict__ ntiles,
             uint32_t* __restrict__ V, uint8_t* __restrict__ F8, LcpEmit emit, uint16_t* __restrict__ Hd, uint32_t wsym)
{
    const uint32_t nt = *ntiles;
    for (uint32_t t = blockIdx.x; t < nt; t += gridDim.x) {
        const SegTile d = tiles[t];
        const uint64_t seg_end = (uint64_t)d.seg_start + d.pad[1];
        const uint32_t depth = d.pad[2];                              // (the bucket's depth, left by k_seg_gather)
        for (uint32_t i = threadIdx.x; i < d.count; i += kBlock) {
            const uint64_t p = (uint64_t)d.begin + i;
            const uint64_t e = E[p];
            const uint32_t key = (uint32_t)(e >> 32);
            const bool head = p == d.seg_start || (uint32_t)(E[p - 1] >> 32) != key;
            const bool last = p + 1 == seg_end || (uint32_t)(E[p + 1] >> 32) != key;
            V[p] = (uint32_t)e;
            F8[p] = (uint8_t)((head ? 1u : 0u) | ((head && last) ? 2u : 0u));
            if (Hd) Hd[p] = (uint16_t)(depth + wsym);
            if (emit.lcp && head && p != d.seg_start) {               // split from its predecessor in this round
                const uint64_t ep = E[p - 1];
                emit.lcp[emit.S[p]] = lcp_from_key2_at(emit, depth, (uint32_t)(ep >> 32), key, (uint32_t)ep, (uint32_t)e);
            }
        }
    }
}

// tile of the segmented sort: E64 elements (rank rounds) or 64-bit keys + values (text rounds)
uint32_t seg_tile_elems(bool kv)
{
    if (seg_small()) return kv ? 8 * kWave * 9 : kSegSmallNW * kWave * kSegSmallKPT;       // (4608 / 4096 elements)
    return kv ? 16 * kWave * 9 : kSegNW * kWave * kSegKPT;
}

template <int KPT, int NW>
static int seg_passes(uint64_t* A, uint64_t* B, const SegSort& q, uint32_t* status, uint64_t status_words, hipStream_t st,
                      double algo)
{
    for (int p = 0; p < 4; p++) {
        SFX_HIP(hipMemsetAsync(status, 0, status_words * sizeof(uint32_t), st));
        SegArgs sa = {reinterpret_cast<const SegTile*>(q.tiles), q.counters, q.segexcl, p, 4};
        const uint64_t* src = (p & 1) ? B : A;
        uint64_t* dst = (p & 1) ? A : B;
        SFX_LAUNCH("seg_radix_pass", algo, (k_radix_pass<SrcE64, DstE64, KPT, true, true, NW, true>), kMaxGrid / 4, NW * kWave, st,
                   SrcE64{src}, DstE64{dst}, (uint64_t)0, 32 + 8 * p, 255u, (uint64_t)0, (const uint32_t*)nullptr,
                   (const uint32_t*)nullptr, status, q.counters + 4 + p, sa);
    }
    return SFX_OK;
}
// tile table of the segments q.segs[0, nseg) (device side; the tile count stays on the device)
int segmented_layout(const SegSort& q, uint32_t nseg, bool kv, hipStream_t st, uint32_t skip_upto)
{
    if (nseg == 0) return SFX_OK;
    SFX_HIP(hipMemsetAsync(q.counters, 0, 16 * sizeof(uint32_t), st));
    SFX_LAUNCH("seg_layout", (double)nseg * 24, k_seg_layout, (nseg + kBlock - 1) / kBlock, kBlock, st,
               reinterpret_cast<const uint2*>(q.segs), nseg, seg_tile_elems(kv), reinterpret_cast<SegTile*>(q.tiles), q.counters,
               skip_upto);
    return SFX_OK;
}

// E (= A) holds key2 << 32 | suffix at the positions of the nseg segments (anything elsewhere is left
// alone; segmented_layout has run); on return V and F8 are written for those positions.  nlarge = sum of
// the segment sizes.
int segmented_sort_e64(uint64_t* A, uint64_t* B, const SegSort& q, uint32_t nseg, uint64_t nlarge, uint32_t* V,
                       uint8_t* F8, hipStream_t st, sfx_build_stats* stats, const LcpEmit& emit, uint16_t* Hd, uint32_t wsym)
{
    if (nseg == 0) return SFX_OK;
    const uint32_t te = seg_tile_elems(false);
    const unsigned grid = (unsigned)dmin<uint64_t>(nlarge / te + nseg, kMaxGrid);
    SFX_LAUNCH("seg_hist", (double)nlarge * 8, k_seg_hist, grid, kBlock, st, A, reinterpret_cast<const SegTile*>(q.tiles),
               q.counters, q.tilehist, 4, 32);
    SFX_LAUNCH("seg_scan", 0.0, k_seg_scan, grid, kBlock, st, reinterpret_cast<const SegTile*>(q.tiles), q.counters,
               q.tilehist, q.segexcl, 4);
    const uint64_t status_words = (2 * (nlarge / te) + 2) * kRadix;
    if (status_words > q.status_words) return SFX_ERR_WORKSPACE;
    if (seg_small()) SFX_TRY((seg_passes<kSegSmallKPT, kSegSmallNW>(A, B, q, q.status, status_words, st, (double)nlarge * 16)));
    else SFX_TRY((seg_passes<kSegKPT, kSegNW>(A, B, q, q.status, status_words, st, (double)nlarge * 16)));
    SFX_LAUNCH("seg_finish", (double)nlarge * 13, k_seg_finish, grid, kBlock, st, A, reinterpret_cast<const SegTile*>(q.tiles),
               q.counters, V, F8, emit, Hd, wsym);
    if (stats) { stats->radix_passes += 4; stats->elements_sorted += 4 * nlarge; }
    return SFX_OK;
}

// ---- cache-confined scatter ------------------------------------------------------------
// target[idx] = val for m (idx << 32 | val) pairs whose idx values are spread over [0, n).
// A random 4-byte write into a multi-GB array costs a 128-byte read-modify-write in HBM
// (22-25 G writes/s measured on the rank and Phi scatters at n = 4*10^8 ... 10^9, whatever the
// array size beyond the L2).  Three radix passes on the top 24 bits of idx first confine each
// stretch of the stream to a 256-byte window of the target, so the writes of a wave merge into
// whole lines; the passes cost less than the scatter saves (26 vs 44 ms per 10^9 pairs), the
// fourth pass of a full sort would not.
__global__ void __launch_bounds__(kBlock)
k_scatter_pairs(const uint64_t* __restrict__ pairs, uint64_t m, uint32_t* __restrict__ target)
{
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i0 < m; i0 += U * stride) {
        uint64_t e[U];
#pragma unroll
        for (int u = 0; u < U; u++) e[u] = (i0 + u * stride < m) ? pairs[i0 + u * stride] : 0ull;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i0 + u * stride < m) target[e[u] >> 32] = (uint32_t)e[u];
    }
}

int scatter_pairs_u32(uint64_t* pairs, uint64_t* tmp, uint64_t m, uint64_t n, uint32_t* target,
                      uint32_t* radix_scratch, hipStream_t st, sfx_build_stats* stats, unsigned hist_blocks)
{
    if (m == 0) return SFX_OK;
    const int nb = bits_for(n > 1 ? n - 1 : 1);
    // measured at n = 10^9 (ms, sort + scatter): direct scatter 44; 8 bits 50; 12 bits 52; 16 bits 37;
    // 20 bits (3 passes, 4 KB windows) 34; 24 bits (still 3 passes, 256-byte windows: the writes
    // coalesce into whole lines) 26
    const int part_bits = scatter_part_bits();
    const int lo = nb > part_bits ? nb - part_bits : 0;
    int in1 = 0;
    // (hist_blocks: the producer of the pairs counted the digits, scatter_pairs_presort_hist)
    SFX_TRY(radix_sort_e64(pairs, tmp, m, 32 + lo, 32 + nb, radix_scratch, st, &in1, stats, nullptr, nullptr, nullptr, hist_blocks));
    const uint64_t* src = in1 ? tmp : pairs;
    unsigned grid = (unsigned)dmin<uint64_t>((m + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("scatter_pairs", (double)m * 12, k_scatter_pairs, grid, kBlock, st, src, m, target);
    return SFX_OK;
}

}  // namespace sfx
