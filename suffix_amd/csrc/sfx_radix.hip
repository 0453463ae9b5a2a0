// sfx_radix.hip -- device-wide LSD radix sort of (key, u32 value) pairs.
//
// This is the "bucket" engine of the suffix sorter: where the reference keeps
// per-symbol bucket head/tail pointers in `Bins` (src/table.rs:671-750) and
// scatters one suffix at a time (head_insert/tail_insert :723-736), the GPU
// engine distributes whole arrays of suffixes 8 key bits (256 buckets) per pass:
//
//   k_radix_hist     each persistent workgroup histograms its contiguous chunk
//                    of keys (16-byte loads) into LDS, one private histogram per
//                    wave, and writes one column of the [256][blocks] matrix;
//   k_radix_scan     one workgroup per digit turns its row into exclusive
//                    offsets and records the digit total (bucket sizes, cf.
//                    Bins::find_sizes :686-704);
//   k_radix_scatter  the same chunking; bucket heads (cf. find_head_pointers
//                    :706-712) live in LDS; every tile of 256*KPT keys is ranked
//                    with wave64 ballots (8 ballots -> match mask -> popcount
//                    rank), reordered through LDS so that each bucket's elements
//                    leave as one contiguous run, then written out; heads advance
//                    by the tile's bucket sizes.  Stable, so passes compose
//                    LSD-first.  Keys and values are staged through the SAME LDS
//                    buffer one after the other, which halves the footprint and
//                    lets a tile hold 8192 keys (32 per bucket on average = 128-B
//                    runs) at 4 workgroups per CU.
//
// HBM traffic per pass and element: read key (hist) + read key,value + write
// key,value  =  3*sizeof(Key) + 8 bytes.  No MFMA anywhere: pure scan/scatter.
#include <stdlib.h>

#include "sfx_host.hpp"

namespace sfx {

// --------------------------------------------------------------------------------------
template <class KeyT, bool FROM_TEXT>
__global__ void __launch_bounds__(kBlock)
k_radix_hist(const KeyT* __restrict__ keys, PackedText src, uint64_t m, int shift, unsigned mask,
             uint64_t chunk, uint32_t* __restrict__ hist)
{
    constexpr int kVec = 16 / sizeof(KeyT);                    // keys per 16-byte load
    struct alignas(16) Vec { KeyT v[kVec]; };
    __shared__ uint32_t h[kWavesPerBlock][kRadix];
    const unsigned tid = threadIdx.x, w = wave_id();
    for (unsigned i = tid; i < kWavesPerBlock * kRadix; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    uint64_t begin = (uint64_t)blockIdx.x * chunk;             // chunk is a multiple of the tile
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    if (FROM_TEXT) {
        for (uint64_t i = begin + tid; i < end; i += kBlock)
            atomicAdd(&h[w][(unsigned)(packed_key<KeyT>(src, i) >> shift) & mask], 1u);
    } else {
        uint64_t vec_end = begin + ((end > begin ? end - begin : 0) / kVec) * kVec;
        for (uint64_t i = begin + (uint64_t)tid * kVec; i < vec_end; i += (uint64_t)kBlock * kVec) {
            Vec q = *reinterpret_cast<const Vec*>(keys + i);
#pragma unroll
            for (int j = 0; j < kVec; j++) atomicAdd(&h[w][(unsigned)(q.v[j] >> shift) & mask], 1u);
        }
        for (uint64_t i = vec_end + tid; i < end; i += kBlock)
            atomicAdd(&h[w][(unsigned)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; k++) c += h[k][tid];
    hist[(uint64_t)tid * gridDim.x + blockIdx.x] = c;
}

// grid = 256 workgroups, one per digit: exclusive scan of that digit's row.
__global__ void __launch_bounds__(kBlock)
k_radix_scan(uint32_t* __restrict__ hist, unsigned nblocks, uint32_t* __restrict__ digit_total)
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
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}

// --------------------------------------------------------------------------------------
template <class KeyT, int KPT, int WPS, bool FROM_TEXT>   // WPS: waves/SIMD the registers must allow
__global__ void __launch_bounds__(kBlock, WPS)
k_radix_scatter(const KeyT* __restrict__ kin, const uint32_t* __restrict__ vin, PackedText src,
                KeyT* __restrict__ kout, uint32_t* __restrict__ vout, uint64_t m, int shift,
                unsigned mask, uint64_t chunk, const uint32_t* __restrict__ hist,
                const uint32_t* __restrict__ digit_total)
{
    constexpr int kTile = kBlock * KPT;
    __shared__ uint32_t cnt[kWavesPerBlock][kRadix];   // per-wave bucket counts, then bases
    __shared__ uint32_t dstart[kRadix];                // tile-local first slot of each bucket
    __shared__ uint32_t cursor[kRadix];                // this workgroup's global bucket heads
    __shared__ uint32_t part[kWavesPerBlock];
    __shared__ KeyT stage[kTile];                      // keys, then (as u32) values
    uint32_t* stage32 = reinterpret_cast<uint32_t*>(stage);

    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lane_lt = (1ull << lane) - 1ull;

    {   // bucket heads: exclusive scan of the digit totals + this workgroup's row offset
        uint32_t total;
        uint32_t ex = block_scan_add_excl(digit_total[tid], part, total);
        cursor[tid] = ex + hist[(uint64_t)tid * gridDim.x + blockIdx.x];
    }
    __syncthreads();

    uint64_t begin = (uint64_t)blockIdx.x * chunk;
    uint64_t end = begin + chunk;
    if (end > m) end = m;

    for (uint64_t tile = begin; tile < end; tile += kTile) {
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, end - tile);
        for (unsigned i = tid; i < kWavesPerBlock * kRadix; i += kBlock) (&cnt[0][0])[i] = 0;
        __syncthreads();

        KeyT key[KPT];
        uint32_t pos[KPT];          // rank within (wave, bucket), then tile-local slot
        // wave-striped: wave w owns tile slots [w*64*KPT, (w+1)*64*KPT), 64 consecutive per round
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            if (FROM_TEXT) key[r] = (idx < nvalid) ? packed_key<KeyT>(src, tile + idx) : ~KeyT(0);
            else key[r] = (idx < nvalid) ? kin[tile + idx] : ~KeyT(0);   // padding sorts last in the tile
        }
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            unsigned d = (unsigned)(key[r] >> shift) & mask;
            uint64_t peers = ~0ull;                         // lanes holding the same digit
#pragma unroll
            for (int b = 0; b < kRadixBits; b++) {
                bool bit = (d >> b) & 1u;
                uint64_t vote = __ballot(bit);
                peers &= bit ? vote : ~vote;
            }
            uint32_t pre = cnt[w][d];
            wave_sync();
            unsigned below = (unsigned)__popcll(peers & lane_lt);
            if (below == 0) cnt[w][d] = pre + (uint32_t)__popcll(peers);
            wave_sync();
            pos[r] = pre + below;
        }
        // values: issued now, consumed after the key phase
        uint32_t val[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            if (FROM_TEXT) val[r] = (uint32_t)(tile + idx);
            else val[r] = (idx < nvalid) ? vin[tile + idx] : 0u;
        }
        __syncthreads();

        // bucket sizes of this tile -> tile-local bucket starts and per-wave bases
        uint32_t c0 = cnt[0][tid], c1 = cnt[1][tid], c2 = cnt[2][tid], c3 = cnt[3][tid];
        uint32_t tile_count = c0 + c1 + c2 + c3, total;
        uint32_t ex = block_scan_add_excl(tile_count, part, total);
        dstart[tid] = ex;
        cnt[0][tid] = ex;
        cnt[1][tid] = ex + c0;
        cnt[2][tid] = ex + c0 + c1;
        cnt[3][tid] = ex + c0 + c1 + c2;
        __syncthreads();

        // key phase: LDS reorder, then each bucket's keys leave as one contiguous run
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            unsigned d = (unsigned)(key[r] >> shift) & mask;
            pos[r] += cnt[w][d];
            stage[pos[r]] = key[r];
        }
        __syncthreads();
        uint32_t dest[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            unsigned p = r * kBlock + tid;
            KeyT k = stage[p];
            unsigned d = (unsigned)(k >> shift) & mask;
            dest[r] = cursor[d] + (p - dstart[d]);
            if (p < nvalid) kout[dest[r]] = k;
        }
        __syncthreads();
        // value phase through the same buffer
#pragma unroll
        for (int r = 0; r < KPT; r++) stage32[pos[r]] = val[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            unsigned p = r * kBlock + tid;
            if (p < nvalid) vout[dest[r]] = stage32[p];
        }
        __syncthreads();
        cursor[tid] += tile_count;
        // the next iteration's first barrier (after zeroing cnt) orders this update
    }
}

// --------------------------------------------------------------------------------------
// Tuning knob (development only): SFX_RADIX_VARIANT picks the scatter geometry.
//   0 (default)  = 3
//   1  8 keys/thread, 6 waves/SIMD     2  16 keys/thread, 4 waves/SIMD
//   3  16 keys/thread, 3 waves/SIMD    4  32 keys/thread, 2 waves/SIMD
static int radix_variant(size_t key_bytes, uint64_t m)
{
    const char* e = getenv("SFX_RADIX_VARIANT");
    int forced = e ? atoi(e) : 0;
    if (forced >= 1 && forced <= 4) return forced;
    (void)key_bytes; (void)m;
    return 3;
}

template <class KeyT, int KPT, int WPS>
static int radix_sort_impl(KeyT* k0, uint32_t* v0, KeyT* k1, uint32_t* v1, uint64_t m, int bit_lo,
                           int bit_hi, uint32_t* hist, hipStream_t st, int* result_in_1,
                           sfx_build_stats* stats, const PackedText* src)
{
    constexpr int kTile = kBlock * KPT;
    PackedText none = {nullptr, 0, 0, 1, 0, 1.0};
    Chunking ch = make_chunking(m, kTile);
    const uint64_t chunk = ch.tiles_per_block * kTile;
    uint32_t* digit_total = hist + (uint64_t)kRadix * kMaxGrid;
    KeyT* kin = k0; uint32_t* vin = v0;
    KeyT* kout = k1; uint32_t* vout = v1;
    int flips = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += kRadixBits) {
        int nb = bit_hi - shift < kRadixBits ? bit_hi - shift : kRadixBits;
        unsigned mask = (1u << nb) - 1u;
        const bool from_text = src && shift == bit_lo;
        // algorithmic bytes: a text-fed pass reads bits/8 bytes per element instead of key (+value)
        const double in_key = from_text ? src->bits / 8.0 : (double)sizeof(KeyT);
        if (from_text) {
            SFX_LAUNCH(sizeof(KeyT) == 4 ? "radix_hist_text_u32" : "radix_hist_text_u64", (double)m * in_key,
                       (k_radix_hist<KeyT, true>), ch.blocks, kBlock, st, kin, *src, m, shift, mask,
                       chunk, hist);
        } else {
            SFX_LAUNCH(sizeof(KeyT) == 4 ? "radix_hist_u32" : "radix_hist_u64", (double)m * in_key,
                       (k_radix_hist<KeyT, false>), ch.blocks, kBlock, st, kin, none, m, shift, mask,
                       chunk, hist);
        }
        SFX_LAUNCH("radix_scan", (double)kRadix * ch.blocks * 8, k_radix_scan, kRadix, kBlock, st,
                   hist, ch.blocks, digit_total);
        if (from_text) {
            SFX_LAUNCH(sizeof(KeyT) == 4 ? "radix_scatter_text_u32" : "radix_scatter_text_u64",
                       (double)m * (in_key + sizeof(KeyT) + 4), (k_radix_scatter<KeyT, KPT, WPS, true>),
                       ch.blocks, kBlock, st, kin, vin, *src, kout, vout, m, shift, mask, chunk, hist,
                       digit_total);
        } else {
            SFX_LAUNCH(sizeof(KeyT) == 4 ? "radix_scatter_u32" : "radix_scatter_u64",
                       2.0 * (double)m * (sizeof(KeyT) + 4), (k_radix_scatter<KeyT, KPT, WPS, false>),
                       ch.blocks, kBlock, st, kin, vin, none, kout, vout, m, shift, mask, chunk, hist,
                       digit_total);
        }
        KeyT* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        flips ^= 1;
        if (stats) { stats->radix_passes++; stats->elements_sorted += m; }
    }
    *result_in_1 = flips;
    return SFX_OK;
}

template <class KeyT>
int radix_sort_pairs(KeyT* k0, uint32_t* v0, KeyT* k1, uint32_t* v1, uint64_t m, int bit_lo,
                     int bit_hi, uint32_t* hist, hipStream_t st, int* result_in_1,
                     sfx_build_stats* stats, const PackedText* src)
{
    *result_in_1 = 0;
    if (m == 0 || bit_hi <= bit_lo) return SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    switch (radix_variant(sizeof(KeyT), m)) {
    case 1: return radix_sort_impl<KeyT, 8, 6>(k0, v0, k1, v1, m, bit_lo, bit_hi, hist, st, result_in_1, stats, src);
    case 2: return radix_sort_impl<KeyT, 16, 4>(k0, v0, k1, v1, m, bit_lo, bit_hi, hist, st, result_in_1, stats, src);
    case 4: return radix_sort_impl<KeyT, 32, 2>(k0, v0, k1, v1, m, bit_lo, bit_hi, hist, st, result_in_1, stats, src);
    default: return radix_sort_impl<KeyT, 16, 3>(k0, v0, k1, v1, m, bit_lo, bit_hi, hist, st, result_in_1, stats, src);
    }
}

template int radix_sort_pairs<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint64_t, int,
                                        int, uint32_t*, hipStream_t, int*, sfx_build_stats*,
                                        const PackedText*);
template int radix_sort_pairs<uint64_t>(uint64_t*, uint32_t*, uint64_t*, uint32_t*, uint64_t, int,
                                        int, uint32_t*, hipStream_t, int*, sfx_build_stats*,
                                        const PackedText*);

}  // namespace sfx
