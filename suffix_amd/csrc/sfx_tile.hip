// sfx_tile.hip -- refinement rounds without device-wide composite sorts.
//
// After the initial k-symbol sort the unresolved suffixes form an "active list" in
// which every bucket (suffixes sharing their first h symbols) is a contiguous run; a
// refinement round has to sort each bucket by key2 = the next symbols (text round) or
// the rank of the suffix h symbols on (rank round) -- the "recursive sort" of the
// reference (src/table.rs:496-500) flattened into rounds.  Round 1 of this engine did
// that with a device-wide LSD sort of the composite (bucket id, key2): 8 passes whose
// high half only keeps every element where it already is.  Here:
//
//   k_compose_*_e64   one 64-bit element per active suffix: (key2 << 32) | suffix
//   k_tile_sort       buckets of at most kTmax members are sorted ENTIRELY IN LDS: a
//                     workgroup owns the buckets whose head lies in its stretch of kT
//                     list positions, loads the kT + kTmax window that is guaranteed to
//                     contain them, sorts (local bucket, key2) with an LDS radix sort and
//                     writes the suffixes back in place together with one flag byte per
//                     element (bucket head / singleton): one read and one write of the list
//   large buckets     (> kTmax members: a few per cent of a natural-language text, all of a
//                     unary one) are extracted, sorted by (bucket id, key2) with the
//                     device-wide radix sort and written back to their positions
//   k_flags_reduce    flag bytes -> the 2-bit-per-element words and per-chunk partials that
//                     k_groups_scan / k_groups_apply (sfx_sa.hip) consume
#include "sfx_host.hpp"

namespace sfx {

// ---- composite elements of a round ---------------------------------------------------
// text round: key2 = 1 << 31 | the next wsym symbols (big-endian), or n-1-i (< h) when the
// suffix has no symbol left at offset h ("shorter sorts first", :422-425)
__global__ void __launch_bounds__(kBlock)
k_compose_text_e64(const uint32_t* __restrict__ suf, uint64_t m, PackedText src, uint64_t h, int drop_bits,
                   uint64_t* __restrict__ E)
{
    constexpr int U = 4;                                   // independent gathers in flight per thread
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q0 < m; q0 += U * stride) {
        uint64_t i[U];
        uint32_t tk[U];
#pragma unroll
        for (int u = 0; u < U; u++) i[u] = (q0 + u * stride < m) ? suf[q0 + u * stride] : 0;
#pragma unroll
        for (int u = 0; u < U; u++)
            tk[u] = (q0 + u * stride < m && i[u] + h < src.n) ? packed_key32(src, i[u] + h) >> drop_bits : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t q = q0 + u * stride;
            if (q < m) {
                const uint32_t key2 = (i[u] + h < src.n) ? (0x80000000u | tk[u]) : (uint32_t)(src.n - 1 - i[u]);
                E[q] = ((uint64_t)key2 << 32) | i[u];
            }
        }
    }
}
// rank round: key2 = rank of suffix i+h (+h), or n-1-i; the caller guarantees n-1+h < 2^32
__global__ void __launch_bounds__(kBlock)
k_compose_rank_e64(const uint32_t* __restrict__ suf, uint64_t m, const uint32_t* __restrict__ isa, uint64_t n,
                   uint64_t h, uint64_t* __restrict__ E)
{
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q0 < m; q0 += U * stride) {
        uint64_t i[U];
        uint32_t rk[U];
#pragma unroll
        for (int u = 0; u < U; u++) i[u] = (q0 + u * stride < m) ? suf[q0 + u * stride] : 0;
#pragma unroll
        for (int u = 0; u < U; u++) rk[u] = (q0 + u * stride < m && i[u] + h < n) ? isa[i[u] + h] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t q = q0 + u * stride;
            if (q < m) {
                const uint32_t key2 = (i[u] + h < n) ? (uint32_t)((uint64_t)rk[u] + h) : (uint32_t)(n - 1 - i[u]);
                E[q] = ((uint64_t)key2 << 32) | i[u];
            }
        }
    }
}

// ---- LDS bucket sort --------------------------------------------------------------------
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }

template <int NW, int KPT>
struct TileSmem {
    static constexpr int kWin = NW * kWave * KPT;
    static constexpr bool kAliasMasks = KPT >= 4;           // a wave's share of `stage` holds its 256 match masks
    uint64_t stage[kWin];                                   // bucket-id window, then the elements being sorted
    uint64_t masks[kAliasMasks ? 1 : NW * kRadixDev];
    uint32_t sufwin[kWin];                                  // suffix of every window position
    uint16_t posmap[kWin];                                  // window offset of the j-th owned element
    uint32_t cnt[NW][kRadixDev];
    uint32_t part[2][NW];
};

// E: (key2 << 32 | suffix) per list position, G: bucket id = list position of the bucket's head.
// Sorts every bucket of <= kTmax members whose head lies in [blockIdx * kT, (blockIdx+1) * kT) by
// key2 (stable), writes the suffixes to V at the same list positions and F8 = 1 (first of its
// (bucket, key2) class) | 2 (class of one).  owned_total += elements handled.
template <int NW, int KPT>
__global__ void __launch_bounds__(NW * kWave)
k_tile_sort(const uint64_t* __restrict__ E, const uint32_t* __restrict__ G, uint64_t m,
            uint32_t* __restrict__ V, uint8_t* __restrict__ F8, unsigned long long* __restrict__ owned_total)
{
    constexpr int kThreads = NW * kWave;
    constexpr int kWin = kThreads * KPT;
    constexpr int kT = kWin / 2;
    constexpr int kTmax = kWin - kT;
    constexpr int kIdxBits = ilog2_c(kWin);
    constexpr int kGidBits = kIdxBits - 1;                  // local bucket id < kT
    static_assert((1 << kIdxBits) == kWin, "window must be a power of two");
    static_assert(kThreads >= kRadixDev, "thread d owns digit d");
    static_assert(kIdxBits + kGidBits <= 32, "element = key2 | local bucket | window offset");
    __shared__ TileSmem<NW, KPT> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const uint64_t base = (uint64_t)blockIdx.x * kT;
    uint32_t* gwin = reinterpret_cast<uint32_t*>(s.stage);
    for (unsigned i = tid; i < (unsigned)kWin; i += kThreads) gwin[i] = (base + i < m) ? G[base + i] : 0xFFFFFFFFu;
    __syncthreads();

    // ownership of the thread's KPT consecutive window positions
    uint64_t e[KPT];
    uint32_t lg[KPT];
    unsigned own = 0;
    const unsigned i0 = tid * KPT;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        const uint64_t p = base + i0 + j;
        e[j] = p < m ? E[p] : 0ull;
        const uint32_t g = gwin[i0 + j];
        bool o = p < m && (uint64_t)g >= base && (uint64_t)g < base + kT;
        if (o) {                                            // a bucket is large iff position head + kTmax is still in it
            const uint64_t far = (uint64_t)g + kTmax;       // (<= base + kWin - 1: inside the window)
            if (far < m && gwin[(unsigned)(far - base)] == g) o = false;
        }
        lg[j] = o ? (uint32_t)((uint64_t)g - base) : 0u;
        own |= (o ? 1u : 0u) << j;
    }
    const uint32_t cnt = (uint32_t)__popc(own);
    const uint32_t incl = wave_scan_add(cnt);
    if (lane == 63) s.part[0][w] = incl;
    __syncthreads();                                        // (every read of gwin is behind this barrier)
    uint32_t before = 0, total = 0;
#pragma unroll
    for (unsigned k = 0; k < (unsigned)NW; k++) {
        const uint32_t q = s.part[0][k];
        if (k < w) before += q;
        total += q;
    }
    if (total == 0) return;
    {
        uint32_t at = before + incl - cnt;
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            s.sufwin[i0 + j] = (uint32_t)e[j];
            if ((own >> j) & 1u) {
                s.stage[at] = (e[j] & 0xFFFFFFFF00000000ull) | ((uint64_t)lg[j] << kIdxBits) | (uint64_t)(i0 + j);
                s.posmap[at] = (uint16_t)(i0 + j);
                at++;
            }
        }
        for (unsigned i = total + tid; i < (unsigned)kWin; i += kThreads) s.stage[i] = ~0ull;   // padding sorts last
    }
    __syncthreads();

    // LSD radix sort of stage[0, total) on key2 (bits 32..63), then on the local bucket id: stable,
    // so every bucket ends up where it was, ordered by key2
    unsigned long long* const my_flags =
        TileSmem<NW, KPT>::kAliasMasks ? reinterpret_cast<unsigned long long*>(s.stage) + w * kRadixDev
                                       : reinterpret_cast<unsigned long long*>(s.masks) + w * kRadixDev;
    unsigned par = 1;
    auto pass = [&](int shift, int nbits) {
        const unsigned mask = (1u << nbits) - 1u;
        uint64_t key[KPT];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) key[r] = s.stage[w * (kWave * KPT) + r * kWave + lane];
        __syncthreads();                                    // all keys are in registers: stage may hold the masks
#pragma unroll
        for (int k = 0; k < kRadixDev / kWave; k++) {
            my_flags[k * kWave + lane] = 0ull;
            s.cnt[w][k * kWave + lane] = 0u;
        }
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            pos[r] = 0;
            if (w * (kWave * KPT) + r * kWave < total)      // (rounds that hold nothing but padding stay where they are)
                pos[r] = rank_round<true>((unsigned)(key[r] >> shift) & mask, my_flags, s.cnt[w], mybit);
        }
        __syncthreads();
        {
            const bool owner = tid < (unsigned)kRadixDev;
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
            if (w * (kWave * KPT) + r * kWave < total)
                s.stage[pos[r] + s.cnt[w][(unsigned)(key[r] >> shift) & mask]] = key[r];
        __syncthreads();
    };
    for (int sh = 32; sh < 64; sh += 8) pass(sh, 8);
    for (int sh = kIdxBits; sh < kIdxBits + kGidBits; sh += 8) pass(sh, dmin(8, kIdxBits + kGidBits - sh));

    for (unsigned i = tid; i < total; i += kThreads) {
        const uint64_t key = s.stage[i];
        const uint64_t cls = key >> kIdxBits;               // (key2, local bucket)
        const bool head = i == 0 || (s.stage[i - 1] >> kIdxBits) != cls;
        const bool last = i + 1 == total || (s.stage[i + 1] >> kIdxBits) != cls;
        const uint64_t p = base + s.posmap[i];
        V[p] = s.sufwin[(unsigned)key & (unsigned)(kWin - 1)];
        F8[p] = (uint8_t)((head ? 1u : 0u) | ((head && last) ? 2u : 0u));
    }
    if (tid == 0) atomicAdd(owned_total, (unsigned long long)total);
}

// ---- large buckets ----------------------------------------------------------------------
// stream compaction of the members of buckets with more than tmax members (order kept):
// phase 0 counts per workgroup, phase 1 emits KL = (bucket id << 32 | key2), VL = suffix,
// P = list position.
__global__ void __launch_bounds__(kBlock)
k_large_extract(const uint64_t* __restrict__ E, const uint32_t* __restrict__ G, uint64_t m, uint64_t tmax,
                uint64_t chunk, int phase, uint32_t* __restrict__ block_counts, uint64_t* __restrict__ KL,
                uint32_t* __restrict__ VL, uint32_t* __restrict__ P)
{
    __shared__ uint32_t part[kWavesPerBlock];
    const unsigned tid = threadIdx.x;
    uint64_t begin = (uint64_t)blockIdx.x * chunk;
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint64_t running = (phase == 1) ? (uint64_t)block_counts[blockIdx.x] : 0ull;
    for (uint64_t b0 = begin; b0 < end; b0 += kBlock) {
        const uint64_t p = b0 + tid;
        uint32_t g = 0;
        bool large = false;
        if (p < end) {
            g = G[p];
            const uint64_t far = (uint64_t)g + tmax;
            large = far < m && G[far] == g;
        }
        uint32_t total;
        const uint32_t ex = block_scan_add_excl<uint32_t>(large ? 1u : 0u, part, total);
        if (phase == 1 && large) {
            const uint64_t e = E[p];
            KL[running + ex] = ((uint64_t)g << 32) | (e >> 32);
            VL[running + ex] = (uint32_t)e;
            P[running + ex] = (uint32_t)p;
        }
        running += total;
    }
    if (phase == 0 && tid == 0) block_counts[blockIdx.x] = (uint32_t)running;
}
// the sorted sub-list goes back to the positions it came from (P is increasing and the sort is
// by bucket first, so sorted element j belongs at P[j])
__global__ void __launch_bounds__(kBlock)
k_large_writeback(const uint64_t* __restrict__ K, const uint32_t* __restrict__ Vs, const uint32_t* __restrict__ P,
                  uint64_t cnt, uint32_t* __restrict__ V, uint8_t* __restrict__ F8)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < cnt; j += stride) {
        const uint64_t k = K[j];
        const bool head = j == 0 || K[j - 1] != k;
        const bool last = j + 1 == cnt || K[j + 1] != k;
        const uint32_t p = P[j];
        V[p] = Vs[j];
        F8[p] = (uint8_t)((head ? 1u : 0u) | ((head && last) ? 2u : 0u));
    }
}

// ---- flag bytes -> flag words + partials (the role of k_groups_reduce after a key sort) ----
__global__ void __launch_bounds__(kBlock)
k_flags_reduce(const uint8_t* __restrict__ F8, uint64_t m, uint64_t chunk, uint32_t* __restrict__ part_head,
               uint32_t* __restrict__ part_keep, uint32_t* __restrict__ part_ghead, uint16_t* __restrict__ flags_out)
{
    __shared__ uint32_t red[3][kWavesPerBlock];
    const unsigned tid = threadIdx.x;
    uint64_t begin = (uint64_t)blockIdx.x * chunk;               // chunk: a multiple of 8 elements
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint32_t last_head = 0, keep = 0, ghead = 0;
    for (uint64_t i0 = begin + (uint64_t)tid * 8; i0 < end; i0 += (uint64_t)kBlock * 8) {
        const uint64_t f = *reinterpret_cast<const uint64_t*>(F8 + i0);      // (F8 is padded to a multiple of 8)
        const unsigned valid = (i0 + 8 <= m) ? 0xFFu : ((1u << (unsigned)(m - i0)) - 1u);
        // bit j of the result = bit 0 (resp. 1) of byte j
        const unsigned head = (unsigned)(((f & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) & valid;
        const unsigned single = (unsigned)((((f >> 1) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) & valid;
        flags_out[i0 / 8] = (uint16_t)(head | (single << 8));
        if (head) last_head = (uint32_t)i0 + (32u - (unsigned)__clz((int)head));
        keep += (uint32_t)__popc(valid & ~single);
        ghead += (uint32_t)__popc(head & ~single);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        last_head = dmax(last_head, __shfl_xor(last_head, d));
        keep += __shfl_xor(keep, d);
        ghead += __shfl_xor(ghead, d);
    }
    if (lane_id() == 0) {
        red[0][wave_id()] = last_head;
        red[1][wave_id()] = keep;
        red[2][wave_id()] = ghead;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0, b = 0, c = 0;
        for (int k = 0; k < kWavesPerBlock; k++) { a = dmax(a, red[0][k]); b += red[1][k]; c += red[2][k]; }
        part_head[blockIdx.x] = a;
        part_keep[blockIdx.x] = b;
        part_ghead[blockIdx.x] = c;
    }
}

// ---- host side -----------------------------------------------------------------------------
// SFX_TILE_SMALL=1 is a test hook: 256-thread workgroups with 256-element windows, so that small
// inputs on the emulator cross tile boundaries and reach the large-bucket path
static bool tile_small()
{
    static const bool v = [] { const char* e = getenv("SFX_TILE_SMALL"); return e && atoi(e) != 0; }();
    return v;
}
constexpr int kTileNW = 16, kTileKPT = 8;                   // 8192-element windows, buckets of <= 4096 in LDS

int compose_text_e64(const uint32_t* V, uint64_t m, const PackedText& pt, uint64_t h, uint64_t* E, hipStream_t st)
{
    const unsigned grid = (unsigned)dmin<uint64_t>((m + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("compose_text_e64", (double)m * 16, k_compose_text_e64, grid, kBlock, st, V, m, pt, h,
               pt.kbits == 32 ? pt.bits : 0, E);
    return SFX_OK;
}
int compose_rank_e64(const uint32_t* V, uint64_t m, const uint32_t* isa, uint64_t n, uint64_t h, uint64_t* E,
                     hipStream_t st)
{
    const unsigned grid = (unsigned)dmin<uint64_t>((m + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("compose_rank_e64", (double)m * 16, k_compose_rank_e64, grid, kBlock, st, V, m, isa, n, h, E);
    return SFX_OK;
}

int tile_round(const TileRound& r, uint64_t m, hipStream_t st, sfx_build_stats* stats)
{
    if (m == 0) return SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    SFX_HIP(hipMemsetAsync(r.counters, 0, 2 * sizeof(unsigned long long), st));
    uint64_t tmax;
    if (tile_small()) {
        constexpr int kWin = 4 * kWave * 1;
        tmax = kWin - kWin / 2;
        const unsigned grid = (unsigned)((m + kWin / 2 - 1) / (kWin / 2));
        SFX_LAUNCH("tile_sort", (double)m * 17, (k_tile_sort<4, 1>), grid, 4 * kWave, st, r.E, r.G, m, r.V, r.F8, r.counters);
    } else {
        constexpr int kWin = kTileNW * kWave * kTileKPT;
        tmax = kWin - kWin / 2;
        const uint64_t tiles = (m + kWin / 2 - 1) / (kWin / 2);
        if (tiles > 0x7FFFFFFFull) return SFX_ERR_TOO_LARGE;
        SFX_LAUNCH("tile_sort", (double)m * 17, (k_tile_sort<kTileNW, kTileKPT>), (unsigned)tiles, kTileNW * kWave, st, r.E,
                   r.G, m, r.V, r.F8, r.counters);
    }
    unsigned long long owned = 0;
    SFX_TRY(read_back(&owned, r.counters, sizeof(owned), st));
    if (owned > m) return SFX_ERR_INTERNAL;
    const uint64_t nlarge = m - owned;
    if (stats) stats->tile_sorted += owned;
    if (nlarge > 0) {
        Chunking ch = make_chunking(m, 1024);
        const uint64_t chunk = ch.tiles_per_block * 1024;
        SFX_LAUNCH("large_count", (double)m * 4, k_large_extract, ch.blocks, kBlock, st, r.E, r.G, m, tmax, chunk, 0,
                   r.block_counts, r.KL0, r.VL0, r.P);
        SFX_LAUNCH("large_scan", 0.0, k_scan_block_counts, 1, kBlock, st, r.block_counts, ch.blocks, r.totals);
        SFX_LAUNCH("large_extract", (double)m * 4 + (double)nlarge * 24, k_large_extract, ch.blocks, kBlock, st, r.E,
                   r.G, m, tmax, chunk, 1, r.block_counts, r.KL0, r.VL0, r.P);
        int in1 = 0;
        SFX_TRY(radix_sort_kv64(r.KL0, r.VL0, r.KL1, r.VL1, nlarge, 0, 32 + bits_for(m - 1), r.radix_scratch, st, &in1,
                                stats, nullptr));
        const unsigned grid = (unsigned)dmin<uint64_t>((nlarge + kBlock - 1) / kBlock, kMaxGrid);
        SFX_LAUNCH("large_writeback", (double)nlarge * 21, k_large_writeback, grid, kBlock, st, in1 ? r.KL1 : r.KL0,
                   in1 ? r.VL1 : r.VL0, r.P, nlarge, r.V, r.F8);
        if (stats) stats->large_sorted += nlarge;
    }
    Chunking ch = make_chunking(m, kFlagChunkTile);
    SFX_LAUNCH("flags_reduce", (double)m * 1.25, k_flags_reduce, ch.blocks, kBlock, st, r.F8, m,
               ch.tiles_per_block * kFlagChunkTile, r.part_head, r.part_keep, r.part_ghead, r.F);
    return SFX_OK;
}

}  // namespace sfx
