// sfx_radix.hip -- device-wide LSD radix sort of (key, u32 value) pairs.
//
// This is the "bucket" engine of the suffix sorter: where the reference keeps
// per-symbol bucket head/tail pointers in `Bins` (src/table.rs:671-750) and
// scatters one suffix at a time (head_insert/tail_insert :723-736), the GPU
// engine distributes whole arrays of suffixes 8 key bits (256 buckets) per pass:
//
//   k_radix_hist     each persistent workgroup histograms its contiguous chunk
//                    of keys into LDS (one private histogram per wave) and
//                    writes one column of the [256][blocks] count matrix;
//   k_radix_scan     one workgroup per digit turns its row into exclusive
//                    offsets and records the digit total (bucket sizes, cf.
//                    Bins::find_sizes :686-704);
//   k_radix_scatter  the same chunking; bucket heads (cf. find_head_pointers
//                    :706-712) live in LDS; every 4096-key tile is ranked with
//                    wave64 ballots (8 ballots -> match mask -> popcount rank),
//                    reordered through LDS so that each bucket's keys leave as
//                    one contiguous run, then written out; heads advance by the
//                    tile's bucket sizes.  Stable, so passes compose LSD-first.
//
// HBM traffic per pass and element: read key (hist) + read key,value + write
// key,value  =  3*sizeof(Key) + 8 bytes.  No MFMA anywhere: pure scan/scatter.
#include "sfx_host.hpp"

namespace sfx {

constexpr int kKeysPerThread = 16;
constexpr int kRadixTile = kBlock * kKeysPerThread;           // 4096 keys per tile

template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_radix_hist(const KeyT* __restrict__ keys, uint64_t m, int shift, unsigned mask,
             uint64_t tiles_per_block, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[kWavesPerBlock][kRadix];
    const unsigned tid = threadIdx.x, w = wave_id();
    for (unsigned i = tid; i < kWavesPerBlock * kRadix; i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    uint64_t begin = (uint64_t)blockIdx.x * tiles_per_block * kRadixTile;
    uint64_t end = begin + tiles_per_block * kRadixTile;
    if (end > m) end = m;
    for (uint64_t i = begin + tid; i < end; i += kBlock) {
        unsigned d = (unsigned)(keys[i] >> shift) & mask;
        atomicAdd(&h[w][d], 1u);
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

template <class KeyT>
__global__ void __launch_bounds__(kBlock)
k_radix_scatter(const KeyT* __restrict__ kin, const uint32_t* __restrict__ vin,
                KeyT* __restrict__ kout, uint32_t* __restrict__ vout, uint64_t m, int shift,
                unsigned mask, uint64_t tiles_per_block, const uint32_t* __restrict__ hist,
                const uint32_t* __restrict__ digit_total)
{
    __shared__ uint32_t cnt[kWavesPerBlock][kRadix];   // per-wave bucket counts, then bases
    __shared__ uint32_t dstart[kRadix];                // tile-local first slot of each bucket
    __shared__ uint32_t cursor[kRadix];                // this workgroup's global bucket heads
    __shared__ uint32_t part[kWavesPerBlock];
    __shared__ KeyT skey[kRadixTile];
    __shared__ uint32_t sval[kRadixTile];

    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint64_t lane_lt = (1ull << lane) - 1ull;

    {   // bucket heads: exclusive scan of the digit totals + this workgroup's row offset
        uint32_t total;
        uint32_t ex = block_scan_add_excl(digit_total[tid], part, total);
        cursor[tid] = ex + hist[(uint64_t)tid * gridDim.x + blockIdx.x];
    }
    __syncthreads();

    uint64_t begin = (uint64_t)blockIdx.x * tiles_per_block * kRadixTile;
    uint64_t end = begin + tiles_per_block * kRadixTile;
    if (end > m) end = m;

    for (uint64_t tile = begin; tile < end; tile += kRadixTile) {
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kRadixTile, end - tile);
        for (unsigned i = tid; i < kWavesPerBlock * kRadix; i += kBlock) (&cnt[0][0])[i] = 0;
        __syncthreads();

        KeyT key[kKeysPerThread];
        uint32_t val[kKeysPerThread];
        uint32_t rnk[kKeysPerThread];
        // wave-striped: wave w owns tile slots [w*1024, (w+1)*1024), 64 consecutive per round
#pragma unroll
        for (int r = 0; r < kKeysPerThread; r++) {
            unsigned idx = w * (kWave * kKeysPerThread) + r * kWave + lane;
            bool valid = idx < nvalid;
            key[r] = valid ? kin[tile + idx] : ~KeyT(0);   // padding sorts last within the tile
            val[r] = valid ? vin[tile + idx] : 0u;
        }
#pragma unroll
        for (int r = 0; r < kKeysPerThread; r++) {
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
            rnk[r] = pre + below;
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

#pragma unroll
        for (int r = 0; r < kKeysPerThread; r++) {
            unsigned d = (unsigned)(key[r] >> shift) & mask;
            unsigned p = cnt[w][d] + rnk[r];
            skey[p] = key[r];
            sval[p] = val[r];
        }
        __syncthreads();

#pragma unroll
        for (int r = 0; r < kKeysPerThread; r++) {
            unsigned p = r * kBlock + tid;
            if (p < nvalid) {
                KeyT k = skey[p];
                unsigned d = (unsigned)(k >> shift) & mask;
                uint32_t g = cursor[d] + (p - dstart[d]);
                kout[g] = k;
                vout[g] = sval[p];
            }
        }
        __syncthreads();
        cursor[tid] += tile_count;
        // next iteration's first barrier (after zeroing cnt) orders this update
    }
}

template <class KeyT>
int radix_sort_pairs(KeyT* k0, uint32_t* v0, KeyT* k1, uint32_t* v1, uint64_t m, int bit_lo,
                     int bit_hi, uint32_t* hist, hipStream_t st, int* result_in_1,
                     sfx_build_stats* stats)
{
    *result_in_1 = 0;
    if (m == 0 || bit_hi <= bit_lo) return SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    Chunking ch = make_chunking(m, kRadixTile);
    uint32_t* digit_total = hist + (uint64_t)kRadix * kMaxGrid;
    KeyT* kin = k0; uint32_t* vin = v0;
    KeyT* kout = k1; uint32_t* vout = v1;
    int flips = 0;
    for (int shift = bit_lo; shift < bit_hi; shift += kRadixBits) {
        int nb = bit_hi - shift < kRadixBits ? bit_hi - shift : kRadixBits;
        unsigned mask = (1u << nb) - 1u;
        SFX_LAUNCH(sizeof(KeyT) == 4 ? "radix_hist_u32" : "radix_hist_u64", (double)m * sizeof(KeyT), (k_radix_hist<KeyT>), ch.blocks, kBlock, st,
                   kin, m, shift, mask, ch.tiles_per_block, hist);
        SFX_LAUNCH("radix_scan", (double)kRadix * ch.blocks * 8, k_radix_scan, kRadix, kBlock, st,
                   hist, ch.blocks, digit_total);
        SFX_LAUNCH(sizeof(KeyT) == 4 ? "radix_scatter_u32" : "radix_scatter_u64",
                   2.0 * (double)m * (sizeof(KeyT) + 4), (k_radix_scatter<KeyT>),
                   ch.blocks, kBlock, st, kin, vin, kout, vout, m, shift, mask,
                   ch.tiles_per_block, hist, digit_total);
        KeyT* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
        flips ^= 1;
        if (stats) { stats->radix_passes++; stats->elements_sorted += m; }
    }
    *result_in_1 = flips;
    return SFX_OK;
}

template int radix_sort_pairs<uint32_t>(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint64_t, int,
                                        int, uint32_t*, hipStream_t, int*, sfx_build_stats*);
template int radix_sort_pairs<uint64_t>(uint64_t*, uint32_t*, uint64_t*, uint32_t*, uint64_t, int,
                                        int, uint32_t*, hipStream_t, int*, sfx_build_stats*);

}  // namespace sfx
