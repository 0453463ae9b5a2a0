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
//   k_tile_sort       buckets of at most kTmax members are sorted ENTIRELY IN LDS: a
//                     workgroup owns the buckets whose head lies in its stretch of kT
//                     list positions, loads the kT + kTmax window that is guaranteed to
//                     contain them, gathers key2 of the members, orders them (all-pairs
//                     ranking for buckets of <= 32, LDS radix sort on (local bucket, key2)
//                     for the rest) and writes the suffixes back in place together with one
//                     flag byte per element (bucket head / singleton): one read and one
//                     write of the list, one gather per member
//   large buckets     (> kTmax members) are announced by the tile that holds their last member and
//                     sorted where they are by the SEGMENTED one-sweep radix sort of
//                     sfx_radix.hip: four 16-byte passes on key2 alone, tiles never straddle
//                     two buckets, look-back only over the earlier tiles of the same bucket
//   k_flags_reduce    flag bytes -> the 2-bit-per-element words and per-chunk partials that
//                     k_groups_scan / k_groups_apply (sfx_sa.hip) consume
#include "sfx_host.hpp"

namespace sfx {

// ---- key2 of a suffix ------------------------------------------------------------------------
// text round: key2 = 1 << 31 | the next wsym symbols (big-endian), or n-1-i (< h) when the suffix
// has no symbol left at offset h ("shorter sorts first", :422-425)
struct TextKey {
    PackedText t;
    uint64_t h;
    int drop_bits;              // 32-bit packed words give up their last symbol to make room for the flag
    // (depth: symbols the members of the suffix's bucket share -- the deep text rounds keep one per bucket)
    __device__ __forceinline__ uint32_t at(uint32_t i, uint64_t depth) const
    {
        const uint64_t p = (uint64_t)i + depth;
        if (p >= t.n) return (uint32_t)(t.n - 1 - (uint64_t)i);
        return 0x80000000u | (packed_key32(t, p) >> drop_bits);
    }
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return at(i, h); }
    __device__ __forceinline__ uint32_t depth_default() const { return (uint32_t)h; }
};
// rank round: key2 = rank of suffix i+h (+h), or n-1-i; the caller guarantees n-1+h < 2^32
struct RankKey {
    const uint32_t* isa;
    uint64_t n, h;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const
    {
        const uint64_t p = (uint64_t)i + h;
        if (p >= n) return (uint32_t)(n - 1 - (uint64_t)i);
        return (uint32_t)((uint64_t)isa[p] + h);
    }
    __device__ __forceinline__ uint32_t at(uint32_t i, uint64_t) const { return (*this)(i); }     // (rank rounds: one h for all)
    __device__ __forceinline__ uint32_t depth_default() const { return (uint32_t)h; }
};

// ---- LDS bucket sort --------------------------------------------------------------------
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }
// buckets up to this size are ranked by all-pairs comparison.  (64 since the kernel only serves rank rounds: 1 GB of
// near-duplicate documents, buckets of ~55 members, tile_sort 139 -> 129 ms; mixed-script UTF-8 unchanged.  32 was the
// better choice for text keys in round 2.)
constexpr int kPairMaxDefault = 64;

template <int NW, int KPT>
struct TileSmem {
    static constexpr int kWin = NW * kWave * KPT;
    static constexpr bool kAliasMasks = KPT >= 4;           // a wave's share of `stage` holds its 256 match masks
    uint64_t stage[kWin];                                   // bucket-id window, then the elements being sorted
    uint64_t masks[kAliasMasks ? 1 : NW * kRadixDev];
    uint32_t sufwin[kWin];                                  // suffix of every window position
    uint16_t posmap[kWin];                                  // window offset that slot j of `stage` writes to
    uint16_t hslot[kWin / 2];                               // slot of the head of the bucket with local id g
    uint8_t blabel[kWin / 2];                               // big buckets: dense label (rank among the tile's big buckets); small: members - 1
    uint16_t bslot[256];                                    // ... and the first slot of the big bucket with that label
    uint32_t cnt[NW][kRadixDev];
    uint32_t part[2][NW];
    uint64_t part64[NW];
};

// V: suffix per list position, G: bucket id = list position of the bucket's head.
// Sorts every bucket of <= kTmax members whose head lies in [blockIdx * kT, (blockIdx+1) * kT) by
// key2 = keyfn(suffix) (stable), writes the suffixes back to V at the same list positions and
// F8 = 1 (first of its (bucket, key2) class) | 2 (class of one).  owned_total += elements handled.
// Buckets of <= kPairMax members (half of all elements of a natural-language text, nearly all in the
// late rounds) are ranked by comparing every member with every other one; the rest goes through an LDS
// radix sort: four passes on key2, then ONE pass on the bucket's dense label (its rank among the
// tile's big buckets, < 256) -- after the key2 passes the labels are interleaved at random, which is
// what the match-mask ranking likes; the raw bucket ids would cost two passes on clustered digits.
template <int NW, int KPT, int kPairMax, class KeyFn>
__global__ void __launch_bounds__(NW * kWave)
k_tile_sort(KeyFn keyfn, const uint32_t* __restrict__ G, uint64_t m, uint32_t* __restrict__ V,
            uint8_t* __restrict__ F8, unsigned long long* __restrict__ owned_total, uint2* __restrict__ segs, LcpEmit emit)
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
    static_assert(kPairMax < kTmax, "the size test looks kPairMax positions past the head");
    static_assert(kWin / (kPairMax + 1) < 256, "dense labels of the big buckets fit one radix digit");
    constexpr int kLabelBits = 8;
    __shared__ TileSmem<NW, KPT> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const uint64_t base = (uint64_t)blockIdx.x * kT;
    uint32_t* gwin = reinterpret_cast<uint32_t*>(s.stage);
    for (unsigned i = tid; i < (unsigned)kWin; i += kThreads) gwin[i] = (base + i < m) ? G[base + i] : 0xFFFFFFFFu;
    __syncthreads();

    // a bucket of more than kTmax members is announced by the tile in whose home range its LAST member
    // lies (one report per bucket): (start, size) into the segment list, owned_total[1] counts them
    for (unsigned i = tid; i < (unsigned)kT; i += kThreads) {
        const uint64_t p = base + i;
        if (p >= m) break;
        const uint32_t g = gwin[i];
        if ((p + 1 == m || gwin[i + 1] != g) && p - g + 1 > (uint64_t)kTmax) {
            const unsigned long long k = atomicAdd(&owned_total[1], 1ull);
            segs[k] = uint2{g, (uint32_t)(p - g + 1)};
        }
    }
    // ownership and size class of the thread's KPT consecutive window positions
    uint32_t suf[KPT], key2[KPT], lg[KPT];
    unsigned own = 0, big = 0, last = 0;                    // (last: the bucket ends at this position)
    const unsigned i0 = tid * KPT;
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        const uint64_t p = base + i0 + j;
        suf[j] = p < m ? V[p] : 0u;
        const uint32_t g = gwin[i0 + j];
        last |= ((i0 + j + 1 >= (unsigned)kWin || gwin[i0 + j + 1] != g) ? 1u : 0u) << j;
        bool o = p < m && (uint64_t)g >= base && (uint64_t)g < base + kT;
        bool bg = false;
        if (o) {                                            // position head + k is in the bucket iff it has > k members
            const unsigned lh = (unsigned)((uint64_t)g - base);
            const uint64_t far = (uint64_t)g + kTmax;       // (<= base + kWin - 1: inside the window)
            if (far < m && gwin[lh + kTmax] == g) o = false;
            else bg = (uint64_t)g + kPairMax < m && gwin[lh + kPairMax] == g;
            lg[j] = lh;
        } else {
            lg[j] = 0u;
        }
        own |= (o ? 1u : 0u) << j;
        big |= ((o && bg) ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < KPT; j++) key2[j] = ((own >> j) & 1u) ? keyfn(suf[j]) : 0u;   // gathers of all owned positions in flight together
    // exclusive prefix of (small members, big members, big heads) counts, packed 16 + 16 + 16 bits
    unsigned bighead = 0;
#pragma unroll
    for (int j = 0; j < KPT; j++) bighead |= (((big >> j) & 1u) && lg[j] == i0 + j) ? (1u << j) : 0u;
    const uint64_t cnt = (uint64_t)__popc(own & ~big) | ((uint64_t)__popc(big) << 16) | ((uint64_t)__popc(bighead) << 32);
    const uint64_t incl = wave_scan_add(cnt);
    if (lane == 63) s.part64[w] = incl;
    __syncthreads();                                        // (every read of gwin is behind this barrier)
    uint64_t before = 0, total = 0;
#pragma unroll
    for (unsigned k = 0; k < (unsigned)NW; k++) {
        const uint64_t q = s.part64[k];
        if (k < w) before += q;
        total += q;
    }
    const unsigned ns = (unsigned)total & 0xFFFFu, nb = (unsigned)(total >> 16) & 0xFFFFu;   // small region [0, ns), big region [ns, ns + nb)
    if (ns + nb == 0) return;
    {
        const uint64_t ex = before + incl - cnt;
        unsigned at_s = (unsigned)ex & 0xFFFFu, at_b = ns + ((unsigned)(ex >> 16) & 0xFFFFu), lab = (unsigned)(ex >> 32);
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            s.sufwin[i0 + j] = suf[j];
            if ((bighead >> j) & 1u) s.blabel[lg[j]] = (uint8_t)lab++;
            // a small bucket's members stand in consecutive slots, in window order: its last member knows how many they are
            if (((own & ~big & last) >> j) & 1u) s.blabel[lg[j]] = (uint8_t)(i0 + j - lg[j]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; j++) {
            if ((own >> j) & 1u) {
                const bool bg = (big >> j) & 1u;
                const unsigned at = bg ? at_b++ : at_s++;
                // small: (key2, local bucket id, offset) -- big: (key2, dense label, offset)
                const unsigned mid = bg ? (unsigned)s.blabel[lg[j]] : lg[j];
                s.stage[at] = ((uint64_t)key2[j] << 32) | ((uint64_t)mid << kIdxBits) | (uint64_t)(i0 + j);
                s.posmap[at] = (uint16_t)(i0 + j);
                if (!bg && lg[j] == i0 + j) s.hslot[lg[j]] = (uint16_t)at;
                if (bg && lg[j] == i0 + j) s.bslot[mid] = (uint16_t)at;
            }
        }
    }
    __syncthreads();

    // small buckets: slot of an element = bucket start + number of members that sort before it
    {
        constexpr int kPer = (kWin + kThreads - 1) / kThreads;
        uint64_t mine[kPer];
        unsigned dest[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const unsigned j = tid + (unsigned)k * kThreads;
            dest[k] = 0xFFFFFFFFu;
            if (j < ns) {
                const uint64_t key = s.stage[j];
                const unsigned lgj = (unsigned)(key >> kIdxBits) & (unsigned)(kT - 1);
                const unsigned b0 = s.hslot[lgj], b1 = b0 + 1u + (unsigned)s.blabel[lgj];
                unsigned r = 0;
                // same bucket: order by key2, then by window offset (stable).  (The loop ran to the first element of
                // another bucket in round 3 -- a field extraction, a compare and a branch per member and member: on near-
                // duplicate documents, buckets of ~55, 125 ms of tile_sort against 112 now; mixed-script UTF-8 21.8 -> 20.1.)
                // (Two / four / eight members per turn with their LDS reads in flight together: 110 / 117 / 117 ms against 112 --
                // at 630 M rank gathers per launch, 14 ms, the kernel is within 15 % of what the gathers alone take.)
                for (unsigned t = b0; t < b1; t++) r += s.stage[t] < key ? 1u : 0u;
                mine[k] = key;
                dest[k] = b0 + r;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if (dest[k] != 0xFFFFFFFFu) s.stage[dest[k]] = mine[k];
    }
    __syncthreads();

    // big buckets: LSD radix sort of stage[ns, ns + nb) on key2 (bits 32..63), then on the local bucket
    // id: stable, so every bucket ends up where it was, ordered by key2
    if (nb > 0) {
        unsigned long long* const my_flags =
            TileSmem<NW, KPT>::kAliasMasks ? reinterpret_cast<unsigned long long*>(s.stage) + w * kRadixDev
                                           : reinterpret_cast<unsigned long long*>(s.masks) + w * kRadixDev;
        // the match masks live in stage[0, NW * 256): when they would overlap live elements of the small
        // region... they may: the masks are only written between the two barriers that bracket the ranking,
        // when every element of BOTH regions that a thread needs is in its registers -- so the small region
        // is saved to registers as well and restored afterwards
        unsigned par = 1;
        auto pass = [&](int shift, int nbits) {
            const unsigned mask = (1u << nbits) - 1u;
            uint64_t key[KPT];
            uint32_t pos[KPT];
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned q = w * (kWave * KPT) + r * kWave + lane;
                key[r] = q < nb ? s.stage[ns + q] : ~0ull;
            }
            // keep what the masks are about to overwrite: slots [0, NW * 256) of the small region
            constexpr int kSave = TileSmem<NW, KPT>::kAliasMasks ? (NW * kRadixDev + kThreads - 1) / kThreads : 0;
            uint64_t saved[kSave > 0 ? kSave : 1];
#pragma unroll
            for (int k = 0; k < kSave; k++) {
                const unsigned q = tid + (unsigned)k * kThreads;
                saved[k] = (q < (unsigned)(NW * kRadixDev) && q < ns) ? s.stage[q] : 0ull;
            }
            __syncthreads();                                // all keys are in registers: stage may hold the masks
#pragma unroll
            for (int k = 0; k < kRadixDev / kWave; k++) {
                my_flags[k * kWave + lane] = 0ull;
                s.cnt[w][k * kWave + lane] = 0u;
            }
            wave_sync();
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                pos[r] = 0;
                if (w * (kWave * KPT) + r * kWave < nb)     // (rounds that hold nothing but padding are skipped)
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
            for (int k = 0; k < kSave; k++) {
                const unsigned q = tid + (unsigned)k * kThreads;
                if (q < (unsigned)(NW * kRadixDev) && q < ns) s.stage[q] = saved[k];
            }
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned q = w * (kWave * KPT) + r * kWave + lane;
                if (q < nb) s.stage[ns + pos[r] + s.cnt[w][(unsigned)(key[r] >> shift) & mask]] = key[r];
            }
            __syncthreads();
        };
        for (int sh = 32; sh < 64; sh += 8) pass(sh, 8);
        pass(kIdxBits, kLabelBits);
    }

    const unsigned tot = ns + nb;
    for (unsigned i = tid; i < tot; i += kThreads) {
        const uint64_t key = s.stage[i];
        const uint64_t cls = key >> kIdxBits;               // (key2, local bucket)
        const bool head = i == 0 || i == ns || (s.stage[i - 1] >> kIdxBits) != cls;
        const bool last = i + 1 == tot || i + 1 == ns || (s.stage[i + 1] >> kIdxBits) != cls;
        const uint64_t p = base + s.posmap[i];
        const uint32_t sfx = s.sufwin[(unsigned)key & (unsigned)(kWin - 1)];
        V[p] = sfx;
        // 8: in the class that starts where the bucket starts -- its head, hence its members' rank, stays (kRankKept)
        const unsigned mid = (unsigned)(key >> kIdxBits) & ((1u << (32 - kIdxBits)) - 1u);
        const unsigned first = i < ns ? (unsigned)s.hslot[mid & (unsigned)(kT - 1)] : (unsigned)s.bslot[mid & 255u];
        const bool kept_rank = (s.stage[first] >> kIdxBits) == cls;
        F8[p] = (uint8_t)((head ? 1u : 0u) | ((head && last) ? 2u : 0u) | (kept_rank ? kRankKept : 0u));
        if (emit.lcp && head && i != 0 && i != ns) {
            // a class head that is not the first member of its bucket: split from its predecessor by this round
            const uint64_t kp = s.stage[i - 1];
            constexpr uint64_t kMid = ((1ull << (32 - kIdxBits)) - 1ull) << kIdxBits;       // bucket id / label field
            if ((kp & kMid) == (key & kMid))
                emit.lcp[emit.S[p]] = lcp_from_key2(emit, (uint32_t)(kp >> 32), (uint32_t)(key >> 32),
                                                    s.sufwin[(unsigned)kp & (unsigned)(kWin - 1)], sfx);
        }
    }
    if (tid == 0) atomicAdd(owned_total, (unsigned long long)tot);
}

// ---- large buckets ----------------------------------------------------------------------
// key2 << 32 | suffix at the list positions of the large buckets (the tile table of the segmented sort
// says where they are): the one gather per member that the LDS path does inside k_tile_sort
template <class KeyFn>
__global__ void __launch_bounds__(kBlock)
k_seg_gather(KeyFn keyfn, const uint32_t* __restrict__ V, SegTileHost* __restrict__ tiles,
             const uint32_t* __restrict__ ntiles, uint64_t* __restrict__ E, const uint16_t* __restrict__ Hd)
{
    const uint32_t nt = *ntiles;
    for (uint32_t t = blockIdx.x; t < nt; t += gridDim.x) {
        const uint32_t begin = tiles[t].begin, count = tiles[t].count;
        // the depth of the tile's bucket: kept in the tile table for k_seg_finish, which rewrites Hd
        const uint32_t depth = Hd ? (uint32_t)Hd[tiles[t].seg_start] : keyfn.depth_default();
        if (threadIdx.x == 0) tiles[t].pad[2] = depth;
        constexpr int U = 4;
        for (uint32_t i0 = threadIdx.x; i0 < count; i0 += U * kBlock) {
            uint32_t sfx[U], k2[U];
#pragma unroll
            for (int u = 0; u < U; u++) sfx[u] = (i0 + u * kBlock < count) ? V[(uint64_t)begin + i0 + u * kBlock] : 0u;
#pragma unroll
            for (int u = 0; u < U; u++) k2[u] = (i0 + u * kBlock < count) ? keyfn.at(sfx[u], depth) : 0u;
#pragma unroll
            for (int u = 0; u < U; u++)
                if (i0 + u * kBlock < count) E[(uint64_t)begin + i0 + u * kBlock] = ((uint64_t)k2[u] << 32) | (uint64_t)sfx[u];
        }
    }
}

// ---- large buckets that fit one tile: sorted inside LDS ------------------------------------------
// A bucket of 1025 .. 11264 members went through the segmented sort like any other: key2 gathered into an
// 8-byte element (one write), four device-wide passes of 16 bytes, flags and suffixes read back -- ~85 bytes per
// member and round although the whole bucket fits a workgroup's LDS.  Here one workgroup takes the bucket:
// suffixes in, key2 gathered, four stable LSD rounds on key2 in LDS (the tile engine of the radix pass: match-mask
// ranking, block scan of the digit counts, reorder through the staging buffer), suffixes + flag bytes (+ fused LCP)
// out: 13 bytes per member and round plus the gather.  Members are spread evenly over the waves (wave w owns
// the 64 * kpt consecutive members from w * 64 * kpt, kpt = ceil(size / threads)), so (wave, round, lane) order is
// list order and the sort is stable.  segs: the (start, size) list of the round; a workgroup takes the segments
// whose size lies in (lo, hi].
template <int NW, int KPT, class KeyFn>
__global__ void __launch_bounds__(NW * kWave)
k_seg_single(KeyFn keyfn, const uint2* __restrict__ segs, uint32_t nseg, uint32_t lo, uint32_t hi, uint32_t* __restrict__ V,
             uint8_t* __restrict__ F8, LcpEmit emit, uint16_t* __restrict__ Hd, uint32_t wsym)
{
    constexpr int kThreads = NW * kWave;
    static_assert(kWave * KPT >= kRadixDev, "the match masks must fit the staging buffer");
    static_assert(kThreads >= kRadixDev, "thread d owns digit d");
    __shared__ struct {
        uint32_t cnt[NW][kRadixDev];
        uint32_t part[2][NW];
        uint64_t stage[NW * kWave * KPT + 2];                         // (+ sentinel slots on either side of the sorted bucket)
    } s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadixDev;
    uint64_t* const stage = s.stage + 1;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(stage) + w * kRadixDev;
    unsigned par = 0;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0u;
    }
    __syncthreads();
    for (uint32_t g = blockIdx.x; g < nseg; g += gridDim.x) {
        const uint32_t begin = segs[g].x, size = segs[g].y;
        if (size <= lo || size > hi) continue;                        // (uniform: the whole workgroup skips)
        const unsigned kpt = (size + kThreads - 1) / kThreads;        // rounds in use, <= KPT
        const unsigned per = kpt * kWave;
        const uint32_t depth = Hd ? (uint32_t)Hd[begin] : keyfn.depth_default();    // (read before the barriers below, rewritten after them)
        uint64_t key[KPT];
        uint32_t pos[KPT];
        {
            uint32_t suf[KPT];
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned idx = w * per + r * kWave + lane;
                suf[r] = ((unsigned)r < kpt && idx < size) ? V[(uint64_t)begin + idx] : 0u;
            }
#pragma unroll
            for (int r = 0; r < KPT; r++) {                           // (the gathers of all rounds in flight together)
                const unsigned idx = w * per + r * kWave + lane;
                key[r] = ((unsigned)r < kpt && idx < size) ? (((uint64_t)keyfn.at(suf[r], depth) << 32) | (uint64_t)suf[r]) : ~0ull;
            }
        }
        for (int shift = 32; shift < 64; shift += 8) {
#pragma unroll
            for (int k = 0; k < kRadixDev / kWave; k++) my_flags[k * kWave + lane] = 0ull;
            wave_sync();
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)r < kpt) pos[r] = rank_round<true>((unsigned)(key[r] >> shift) & 255u, my_flags, s.cnt[w], mybit);
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
                if ((unsigned)r < kpt) stage[pos[r] + s.cnt[w][(unsigned)(key[r] >> shift) & 255u]] = key[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)r < kpt) key[r] = stage[w * per + r * kWave + lane];
            if (owner) {
#pragma unroll
                for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0u;
            }
            __syncthreads();                                          // (the padding, key2 = ~0, is behind the members: slots >= size)
        }
        // stage[0, size) = the bucket in key2 order; flags and fused LCP from the neighbours in LDS
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * per + r * kWave + lane;
            if ((unsigned)r < kpt && idx < size) {
                const uint64_t e = key[r];
                const uint32_t k2 = (uint32_t)(e >> 32);
                const bool head = idx == 0 || (uint32_t)(stage[idx - 1] >> 32) != k2;
                const bool last = idx + 1 == size || (uint32_t)(stage[idx + 1] >> 32) != k2;
                const uint64_t p = (uint64_t)begin + idx;
                V[p] = (uint32_t)e;
                const bool kept_rank = (uint32_t)(stage[0] >> 32) == k2;           // (the bucket's first class: same head, same rank)
                F8[p] = (uint8_t)((head ? 1u : 0u) | ((head && last) ? 2u : 0u) | (kept_rank ? kRankKept : 0u));
                if (Hd) Hd[p] = (uint16_t)(depth + wsym);
                if (emit.lcp && head && idx != 0) {                   // split from its predecessor in this round
                    const uint64_t ep = stage[idx - 1];
                    emit.lcp[emit.S[p]] = lcp_from_key2_at(emit, depth, (uint32_t)(ep >> 32), k2, (uint32_t)ep, (uint32_t)e);
                }
            }
        }
        __syncthreads();                                              // (stage is read to the end before the next bucket's masks)
    }
}

// ---- deep text rounds: small buckets finished by one wave each ---------------------------------
// A text round of k_tile_sort moves every unresolved suffix through a list pass (read G / V, write V / F8, flags ->
// scan -> apply) for each 4 symbols it looks at: 2.2 G member-rounds for 0.9 G unresolved suffixes of 1 GB of
// English-like text.  But what one round leaves of a bucket of a few hundred members is a handful of tiny
// sub-buckets that the NEXT rounds could order without ever leaving the CU.  k_deep_wave does that: a WAVE (no
// workgroup barrier anywhere) owns the buckets whose head lies in its stretch of H = W / 2 list positions and that
// have at most W - H members (the W-position window then contains them), and finishes them:
//   repeat:  every member of a still-tied sub-bucket gathers the next 64 bits of symbols (ONE random line for 8-9
//            symbols instead of one per 4), is ranked against the other members of its sub-bucket by direct
//            comparison (lt = keys below, eq = equal keys: its new place, the extent of its new sub-bucket and
//            whether it is now alone all come out of the same loop), and moves there
//   until nothing is tied, the iterations stop resolving (a repeat: left to the rank rounds), or the cap is hit.
// Sub-buckets only ever split, so all state is per list position: suffix, start and end of the sub-bucket it
// lies in (LDS, 16 bytes per position with the key).  Members that stay tied leave as ordinary buckets of the
// active list with Hd = the symbols their sub-bucket is now known to share, which is where the next round picks
// them up (a per-bucket depth: they do not have to agree with the buckets that took the large path).
// Buckets above W - H members are announced to the large-bucket path exactly as k_tile_sort does.
struct DeepTextKey {
    PackedText t;
    int shift;                  // 2 * kbits - wsym * bits: symbols of the 2-word window that do not fit beside the flag
    int wsym;                   // symbols per key
    __device__ __forceinline__ uint64_t operator()(uint32_t i, uint32_t depth) const
    {
        const uint64_t p = (uint64_t)i + depth;
        if (p >= t.n) return t.n - 1 - (uint64_t)i;
        return (1ull << 63) | (packed_key64(t, p) >> shift);
    }
};
constexpr uint32_t kDeepMaxDepth = 65000;                   // Hd is 16 bits

// LCP of a class head (suffix sb, key kb) with a member of the class in front of it (suffix sa, key ka) when the
// members of their bucket share `depth` symbols (the 64-bit form of lcp_from_key2_at)
__device__ __forceinline__ uint32_t lcp_deep(const LcpEmit& L, uint32_t depth, uint64_t ka, uint64_t kb, uint32_t sa, uint32_t sb)
{
    const uint32_t la = L.n - sa, lb = L.n - sb;
    if (!((ka & kb) >> 63)) return la < lb ? la : lb;
    const uint64_t x = ka ^ kb;
    const uint32_t lz = (uint32_t)__clzll((long long)x) - (64u - (uint32_t)L.field_bits64);
    const uint32_t v = depth + ((lz * L.inv_bits) >> 16);
    if (v > la) return kLcpBoundFlag | depth;
    return v < lb ? v : lb;
}

// One element of the wave's window during a sort: (sub-bucket tag, the 64 gathered key bits, the slot it came from), ordered by
// the three in turn -- a strict total order: ties cannot make the two sides of a compare-exchange disagree.  For the network the
// 96 bits travel as hi = tag << 48 | key >> 16 and lo = (key & 0xFFFF) << 16 | origin: one 64-bit and one 32-bit compare per
// compare-exchange instead of three compares with their selects (the kernel is VALU-bound: SQ counters, 75 % of the SIMD cycles).
struct DeepElem { uint64_t hi; uint32_t lo; };
__device__ __forceinline__ DeepElem deep_pack(uint64_t key, uint32_t pk)
{
    return DeepElem{((uint64_t)(pk >> 16) << 48) | (key >> 16), ((uint32_t)(key & 0xFFFFull) << 16) | (pk & 0xFFFFu)};
}
__device__ __forceinline__ void deep_unpack(const DeepElem& x, uint64_t& key, uint32_t& pk)
{
    key = (x.hi << 16) | (uint64_t)(x.lo >> 16);
    pk = ((uint32_t)(x.hi >> 48) << 16) | (x.lo & 0xFFFFu);
}
__device__ __forceinline__ bool deep_less(const DeepElem& a, const DeepElem& b)
{
    return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo;
}
// Bitonic sort of the 64 * E elements of a wave, E consecutive ones per lane (element i = lane * E + e): the
// compare-exchanges at distance < E stay inside a lane's registers, the others swap with lane ^ (distance / E).
// No LDS, no divergence, the same instructions per element whatever the bucket sizes are.
template <int E>
__device__ __forceinline__ void deep_bitonic(uint64_t (&key)[E], uint32_t (&pk)[E])
{
    const unsigned lane = lane_id();
    DeepElem x[E];
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = deep_pack(key[e], pk[e]);
#pragma unroll
    for (int k = 2; k <= kWave * E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                const unsigned lm = (unsigned)(j / E);
                const bool asc = (lane & (unsigned)(k / E)) == 0u;          // (k >= 2 * E here: the direction is the lane's)
                const bool keep_min = asc == ((lane & lm) == 0u);
#pragma unroll
                for (int e = 0; e < E; e++) {
                    DeepElem o;
                    o.hi = __shfl_xor(x[e].hi, (int)lm);
                    o.lo = __shfl_xor(x[e].lo, (int)lm);
                    const bool mine_less = deep_less(x[e], o);
                    if (mine_less != keep_min) x[e] = o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const int f = e ^ j;
                    if (f > e) {
                        const bool asc = k >= E ? ((lane * (unsigned)E) & (unsigned)k) == 0u : (e & k) == 0;
                        if (deep_less(x[f], x[e]) == asc) {
                            const DeepElem t = x[e]; x[e] = x[f]; x[f] = t;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; e++) deep_unpack(x[e], key[e], pk[e]);
}

// counters[1] += buckets announced to the large path (the returned value is the bucket's place in `segs`); the members
// owned and the gathers made are summed per wave into one of kDeepSlots counter lines (slots[slot * 8 + 0 / 1]: one
// device-wide counter for millions of waves is a queue at one L2 channel -- measured: it was the whole kernel time)
constexpr unsigned kDeepSlots = 1024;
__global__ void __launch_bounds__(kBlock)
k_deep_totals(const unsigned long long* __restrict__ slots, unsigned long long* __restrict__ counters, int set_owned)
{
    __shared__ unsigned long long part[2][kWavesPerBlock];
    unsigned long long a = 0, b = 0;
    for (unsigned i = threadIdx.x; i < kDeepSlots; i += kBlock) { a += slots[i * 8u]; b += slots[i * 8u + 1u]; }
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if (lane_id() == 0) { part[0][wave_id()] = a; part[1][wave_id()] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long x = 0, y = 0;
        for (int w = 0; w < kWavesPerBlock; w++) { x += part[0][w]; y += part[1][w]; }
        if (set_owned) counters[0] = x;                    // (first pass of a round: what is left is the large path's)
        counters[3] += y;                                  // gathers of the whole build (zeroed by its first round)
    }
}
template <int W>
struct DeepSmem {
    int32_t st[W];                                          // set-up: start (window position) of every position's bucket, -1 = not here
    uint32_t suf[W];                                        // slot j of the compacted tied members: suffix,
    uint16_t pos[W];                                        //   window position the slot stands for,
    uint16_t tag[W];                                        //   slot of the first member of its sub-bucket,
    uint16_t hd[W];                                         //   depth of its bucket when the wave took it
};

// One iteration over the cnt <= 64 * E tied members of a wave (slots [0, cnt), sub-buckets contiguous): gather the next
// key of every member, sort the slots by (sub-bucket, key) in registers, find the new sub-buckets, write out the members
// that are now alone (final) and compact the others to the front.  Returns the number still tied.
template <int E, int W, bool EMIT>
__device__ __forceinline__ unsigned deep_step(const DeepTextKey& keyfn, DeepSmem<W>& s, unsigned cnt, uint32_t it, uint64_t wbase,
                                              uint32_t* __restrict__ V, uint8_t* __restrict__ F8, const LcpEmit& emit)
{
    const unsigned lane = lane_id();
    const unsigned j0 = lane * (unsigned)E;
    uint64_t key[E];
    uint32_t pk[E], suf[E];
    uint16_t pos[E], hd[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const unsigned j = j0 + (unsigned)e;
        const bool valid = j < cnt;
        suf[e] = valid ? s.suf[j] : 0u;
        pos[e] = valid ? s.pos[j] : (uint16_t)0;
        hd[e] = valid ? s.hd[j] : (uint16_t)0;
        pk[e] = ((valid ? (uint32_t)s.tag[j] : 0xFFFFu) << 16) | j;     // (empty slots sort behind everything)
    }
#pragma unroll
    for (int e = 0; e < E; e++) key[e] = j0 + (unsigned)e < cnt ? keyfn(suf[e], (uint32_t)hd[e] + it * (uint32_t)keyfn.wsym) : ~0ull;
    // (Ordering by all-pairs comparison inside the sub-buckets when none of the wave's has more than 8 / 16 / 32 members -- keys
    // through LDS, ~11 instructions per member and step against the network's 15 per member and stage -- was measured on config 3:
    // 22.3 / 21.9 / 23.5 ms against 17.8.  The 3 KB of LDS per wave it needs cost a quarter of the occupancy (20.3 ms with the
    // network alone and the larger LDS footprint), and the kernel lives on both: 75 % VALU-busy AND waiting on its gathers.)
    deep_bitonic<E>(key, pk);
    wave_sync();
#pragma unroll
    for (int e = 0; e < E; e++) suf[e] = s.suf[pk[e] & 0xFFFFu];        // the suffixes follow their elements
    // new sub-buckets: a slot starts one where the old sub-bucket or the key changes
    const uint64_t pkey = __shfl_up(key[E - 1], 1u);
    const uint32_t ppk = __shfl_up(pk[E - 1], 1u);
    const uint32_t psuf = __shfl_up(suf[E - 1], 1u);
    unsigned nhead = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint64_t kp = e ? key[e - 1] : pkey;
        const uint32_t pp = e ? pk[e - 1] : ppk;
        const bool other_bucket = (e == 0 && lane == 0u) || (pp >> 16) != (pk[e] >> 16);
        const bool hnew = other_bucket || kp != key[e] || j0 + (unsigned)e >= cnt;
        nhead |= (hnew ? 1u : 0u) << e;
        if (EMIT && emit.lcp && hnew && !other_bucket && j0 + (unsigned)e < cnt)    // split from the member in front of it by this key
            emit.lcp[emit.S[wbase + pos[e]]] = lcp_deep(emit, (uint32_t)hd[e] + it * (uint32_t)keyfn.wsym, kp, key[e],
                                                        e ? suf[e - 1] : psuf, suf[e]);
    }
    uint32_t ntag[E];
    {   // first slot of every slot's sub-bucket: running maximum of the head slots
        uint32_t run = 0, loc[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
            if ((nhead >> e) & 1u) run = j0 + (unsigned)e + 1u;         // (slot + 1: 0 = none in this lane)
            loc[e] = run;
        }
        const uint32_t incl = wave_scan_max(run);
        uint32_t before = __shfl_up(incl, 1u);
        if (lane == 0u) before = 0u;
#pragma unroll
        for (int e = 0; e < E; e++) ntag[e] = (loc[e] ? loc[e] : before) - 1u;
    }
    const unsigned nl = __shfl_down(nhead, 1u) & 1u;
    const unsigned single = nhead & ((nhead >> 1) | ((lane == 63u ? 1u : nl) << (E - 1)));
    unsigned keep = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (j0 + (unsigned)e < cnt) {
            if ((single >> e) & 1u) {
                const uint64_t p = wbase + pos[e];
                V[p] = suf[e];
                F8[p] = (uint8_t)7;                                     // first of its class | alone | finished here
            } else {
                keep |= 1u << e;
            }
        }
    }
    // the members still tied move to the front: whole sub-buckets, in order
    const unsigned mine = (unsigned)__popc(keep);
    const unsigned incl = wave_scan_add(mine);
    const unsigned total = __shfl(incl, 63);
    unsigned at = incl - mine;
    wave_sync();                                                        // (every read of the old slots is done)
#pragma unroll
    for (int e = 0; e < E; e++) {
        if ((keep >> e) & 1u) {
            const unsigned j = j0 + (unsigned)e;
            s.suf[at] = suf[e];
            s.pos[at] = pos[e];
            s.hd[at] = hd[e];
            s.tag[at] = (uint16_t)(at - (j - ntag[e]));                 // (its sub-bucket moves as a whole)
            at++;
        }
    }
    wave_sync();
    return total;
}

// The buckets of the active list (G), each from its own depth Hd; buckets above W - H members are announced to the
// large path.  (A second pass of the same kernel over the classes the large path makes of its buckets was measured:
// it finishes most of them a round earlier, but the large buckets stay large for as many levels as before and the
// extra scan of the list costs what the shorter lists save -- 222 against 177 ms on 1 GB of English-like text.)
// (the kernel waits on its gathers: at 8 waves per SIMD -- 64 registers -- config 3's deep rounds took 19.4 ms in round 3, at the
// 6 the compiler picks by itself 21.6; with round 4's packed sort elements 64 registers spill three words, and 7 waves -- 72
// registers, no spill -- is the better point: 17.3 against 17.8 ms)
template <int KPT, bool EMIT>
__global__ void __launch_bounds__(kBlock) SFX_WAVES_PER_EU(KPT <= 4 ? 7 : 4, 8)
k_deep_wave(DeepTextKey keyfn, const uint32_t* __restrict__ G, uint64_t m, uint32_t* __restrict__ V, uint8_t* __restrict__ F8,
            uint16_t* __restrict__ Hd, unsigned long long* __restrict__ counters,
            unsigned long long* __restrict__ slots, uint2* __restrict__ segs, LcpEmit emit, int max_iter, uint32_t max_depth)
{
    constexpr int W = kWave * KPT, H = W / 2, kOwn = W - H;
    static_assert((KPT & (KPT - 1)) == 0 && KPT >= 2 && KPT <= 8 && W <= 32768, "bitonic network up to 8 per lane; positions fit 15 bits");
    __shared__ DeepSmem<W> smem[kWavesPerBlock];
    DeepSmem<W>& s = smem[wave_id()];
    const unsigned lane = lane_id();
    const uint64_t wbase = ((uint64_t)blockIdx.x * kWavesPerBlock + wave_id()) * (uint64_t)H;
    if (wbase >= m) return;                                 // (the whole wave)
    const unsigned c0 = lane * (unsigned)KPT;               // the lane's positions: c0 .. c0 + KPT - 1
    const uint64_t p0 = wbase + c0;

    int32_t start[KPT];
    unsigned headm = 0, elig = 0;                           // bit e: position c0 + e starts a bucket / may be taken
    uint32_t g[KPT];
    if (p0 + KPT <= m) {
#pragma unroll
        for (int q = 0; q < KPT / 2; q++) {
            const uint2 v = *reinterpret_cast<const uint2*>(G + p0 + 2 * q);
            g[2 * q] = v.x;
            g[2 * q + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < KPT; e++) g[e] = p0 + e < m ? G[p0 + e] : 0u;
    }
#pragma unroll
    for (int e = 0; e < KPT; e++) {
        const bool in = p0 + e < m;
        headm |= ((!in || (uint64_t)g[e] == p0 + e) ? 1u : 0u) << e;        // (the end of the list ends the last bucket)
        elig |= (in ? 1u : 0u) << e;
        start[e] = (in && (uint64_t)g[e] >= wbase) ? (int32_t)((uint64_t)g[e] - wbase) : -1;
    }
#pragma unroll
    for (int e = 0; e < KPT; e++) s.st[c0 + e] = ((elig >> e) & 1u) ? start[e] : -2;
    wave_sync();
    // the head flag of the position after this lane's last one (window end: counts as a head)
    const unsigned nxt_lane_head = __shfl_down(headm, 1u) & 1u;
    const unsigned nexth = (headm >> 1) | ((lane == 63u ? 1u : nxt_lane_head) << (KPT - 1));   // bit e: position c0 + e + 1 is a head
    unsigned own = 0, act = 0;
#pragma unroll
    for (int e = 0; e < KPT; e++) {
        const unsigned c = c0 + (unsigned)e;
        const uint64_t p = p0 + e;
        // large buckets: announced by the wave in whose home their last member lies
        if (c < (unsigned)H && p < m && ((nexth >> e) & 1u) && p - (uint64_t)g[e] + 1 > (uint64_t)kOwn) {
            const unsigned long long k = atomicAdd(&counters[1], 1ull);
            segs[k] = uint2{g[e], (uint32_t)(p - (uint64_t)g[e] + 1)};
        }
        if (((elig >> e) & 1u) && start[e] >= 0 && start[e] < H) {
            const unsigned sc = (unsigned)start[e];
            const bool longer = wbase + sc + kOwn < m && s.st[sc + kOwn] == start[e];      // (sc + kOwn < W)
            if (!longer) {
                own |= 1u << e;
                if (!(((headm >> e) & 1u) && ((nexth >> e) & 1u))) act |= 1u << e;          // not alone in its bucket
                else F8[p] = (uint8_t)7;                                   // (a bucket of one: nothing to do)
            }
        }
    }
    // the tied members move to slots [0, cnt): whole buckets, in list order
    unsigned cnt;
    {
        const unsigned mine = (unsigned)__popc(act);
        const unsigned incl = wave_scan_add(mine);
        cnt = __shfl(incl, 63);
        unsigned at = incl - mine;
#pragma unroll
        for (int e = 0; e < KPT; e++) {
            if ((act >> e) & 1u) {
                const unsigned c = c0 + (unsigned)e;
                s.suf[at] = V[p0 + e];
                s.pos[at] = (uint16_t)c;
                s.hd[at] = Hd[p0 + e];
                s.tag[at] = (uint16_t)(at - (c - (unsigned)start[e]));                      // (its bucket moves as a whole)
                at++;
            }
        }
    }
    wave_sync();
    auto wave_sum = [](unsigned v) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        return v;
    };
    unsigned n_gather = 0;
    uint32_t it = 0;
    while (cnt > 0 && (int)it < max_iter) {
        {   // the depth must stay representable: the buckets that another key would take past the 16 bits of Hd leave now, as
            // buckets of the list with the depth they have reached (the rank rounds' job) -- the wave's other buckets go on
            // (until round 5 one such bucket ended the whole wave's round: ADVICE round 3)
            const uint32_t next_add = (it + 1u) * (uint32_t)keyfn.wsym;
            bool over = false;
            for (unsigned j = lane; j < cnt; j += kWave) over |= (uint32_t)s.hd[j] + next_add > max_depth;
            if (__any(over)) {
                unsigned kept = 0;                                      // slots [0, kept): the members that stay, whole sub-buckets, in order
                for (unsigned j0 = 0; j0 < cnt; j0 += kWave) {          // (a chunk writes below what later chunks still have to read)
                    const unsigned j = j0 + lane;
                    const bool valid = j < cnt;
                    const uint32_t suf = valid ? s.suf[j] : 0u;
                    const uint16_t pos = valid ? s.pos[j] : (uint16_t)0, hd = valid ? s.hd[j] : (uint16_t)0, tag = valid ? s.tag[j] : (uint16_t)0;
                    const bool out = valid && (uint32_t)hd + next_add > max_depth;
                    const bool stay = valid && !out;
                    if (out) {
                        const uint64_t p = wbase + pos;
                        V[p] = suf;
                        F8[p] = (uint8_t)((tag == j ? 1u : 0u) | 4u);
                        Hd[p] = (uint16_t)((uint32_t)hd + it * (uint32_t)keyfn.wsym);
                    }
                    const unsigned long long bal = __ballot(stay);
                    const unsigned at = kept + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                    wave_sync();
                    if (stay) {
                        s.suf[at] = suf;
                        s.pos[at] = pos;
                        s.hd[at] = hd;
                        s.tag[at] = (uint16_t)(at - (j - tag));          // (a sub-bucket stays or leaves as a whole: one depth)
                    }
                    kept += (unsigned)__popcll(bal);
                    wave_sync();
                }
                cnt = kept;
                if (cnt == 0) break;
            }
        }
        n_gather += cnt;
        unsigned left;
        if (cnt <= (unsigned)kWave) left = deep_step<1, W, EMIT>(keyfn, s, cnt, it, wbase, V, F8, emit);
        else if (KPT >= 2 && cnt <= 2u * kWave) left = deep_step<(KPT >= 2 ? 2 : 1), W, EMIT>(keyfn, s, cnt, it, wbase, V, F8, emit);
        else if (KPT >= 4 && cnt <= 4u * kWave) left = deep_step<(KPT >= 4 ? 4 : 1), W, EMIT>(keyfn, s, cnt, it, wbase, V, F8, emit);
        else left = deep_step<KPT, W, EMIT>(keyfn, s, cnt, it, wbase, V, F8, emit);
        it++;
        const unsigned resolved = cnt - left;
        cnt = left;
        // a wave most of whose members stay tied is looking at a repeat: the rank rounds' job
        if (resolved * 8u < left && left * 4u >= (unsigned)H) break;
    }
    // what is still tied leaves as buckets of the list, with the depth reached
    for (unsigned j = lane; j < cnt; j += kWave) {
        const uint64_t p = wbase + s.pos[j];
        V[p] = s.suf[j];
        F8[p] = (uint8_t)((s.tag[j] == j ? 1u : 0u) | 4u);
        Hd[p] = (uint16_t)((uint32_t)s.hd[j] + it * (uint32_t)keyfn.wsym);
    }
    const unsigned n_own = wave_sum((unsigned)__popc(own));
    if (lane == 0) {
        unsigned long long* slot = slots + (size_t)((blockIdx.x * (unsigned)kWavesPerBlock + wave_id()) % kDeepSlots) * 8u;
        if (n_own) atomicAdd(&slot[0], (unsigned long long)n_own);
        if (n_gather) atomicAdd(&slot[1], (unsigned long long)n_gather);
    }
}

// ---- flag bytes -> flag words + partials (the role of k_groups_reduce after a key sort) ----
__global__ void __launch_bounds__(kBlock)
k_flags_reduce(const uint8_t* __restrict__ F8, uint64_t m, uint64_t chunk, uint32_t* __restrict__ part_head,
               uint32_t* __restrict__ part_keep, uint32_t* __restrict__ part_ghead, uint16_t* __restrict__ flags_out,
               uint32_t* __restrict__ part_pairs)
{
    // part_pairs (rank rounds): elements of the chunk whose rank changes (flag kRankKept clear)
    __shared__ uint32_t red[4][kWavesPerBlock];
    const unsigned tid = threadIdx.x;
    uint64_t begin = (uint64_t)blockIdx.x * chunk;               // chunk: a multiple of 8 elements
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint32_t last_head = 0, keep = 0, ghead = 0, pairs = 0;
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
        const unsigned same = (unsigned)((((f >> 3) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
        pairs += (uint32_t)__popc(valid & ~same);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        last_head = dmax(last_head, __shfl_xor(last_head, d));
        keep += __shfl_xor(keep, d);
        ghead += __shfl_xor(ghead, d);
        pairs += __shfl_xor(pairs, d);
    }
    if (lane_id() == 0) {
        red[0][wave_id()] = last_head;
        red[1][wave_id()] = keep;
        red[2][wave_id()] = ghead;
        red[3][wave_id()] = pairs;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0, b = 0, c = 0, e = 0;
        for (int k = 0; k < kWavesPerBlock; k++) { a = dmax(a, red[0][k]); b += red[1][k]; c += red[2][k]; e += red[3][k]; }
        part_head[blockIdx.x] = a;
        part_keep[blockIdx.x] = b;
        part_ghead[blockIdx.x] = c;
        if (part_pairs) part_pairs[blockIdx.x] = e;
    }
}

// ---- host side -----------------------------------------------------------------------------
// SFX_TILE_SMALL=1 is a test hook: 256-thread workgroups with 256-element windows, so that small
// inputs on the emulator cross tile boundaries and reach the large-bucket path
static bool tile_small()
{
    static const bool v = [] { const char* e = dev_env("SFX_TILE_SMALL"); return e && atoi(e) != 0; }();
    return v;
}
// SFX_TILE_GEOM (development): 0 = 1024 threads x 8 (8192-element windows, one workgroup per CU),
// 1 = 1024 x 4 (4096, two per CU), 2 = 512 x 8 (4096, two per CU), 3 = 512 x 4 (2048, four per CU: the
// default); SFX_TILE_PAIR = 32 (default) | 64: all-pairs threshold.  Measured on 1 GB of English-like
// text (tile_sort ms, profiles/r2_tile_geometry_sweep.jsonl): 66 / 76 / 50 / 34 at pair 64, 62 / 72 / 46 / 33
// at pair 32 -- small workgroups hide the gather latency better and leave fewer members to the radix passes
static int tile_geom()
{
    static const int v = [] { const char* e = dev_env("SFX_TILE_GEOM"); int x = e ? atoi(e) : 3; return x >= 0 && x <= 5 ? x : 3; }();
    return v;
}
static int tile_pair()
{
    static const int v = [] { const char* e = dev_env("SFX_TILE_PAIR"); int x = e ? atoi(e) : kPairMaxDefault; return x == 64 ? 64 : 32; }();
    return v;
}

template <int NW, int KPT, int PM, class KeyFn>
static int launch_tile(const KeyFn& keyfn, const TileRound& r, uint64_t m, hipStream_t st, uint64_t* tmax)
{
    constexpr int kWin = NW * kWave * KPT;
    *tmax = kWin - kWin / 2;
    const uint64_t tiles = (m + kWin / 2 - 1) / (kWin / 2);
    if (tiles > 0x7FFFFFFFull) return SFX_ERR_TOO_LARGE;
    // read V + G, gather key2 (one sector), write V + F8
    SFX_LAUNCH("tile_sort", (double)m * (4 + 4 + 4 + 4 + 1), (k_tile_sort<NW, KPT, PM, KeyFn>), (unsigned)tiles, NW * kWave, st,
               keyfn, r.G, m, r.V, r.F8, r.counters, reinterpret_cast<uint2*>(r.seg.segs), r.emit);
    return SFX_OK;
}

// buckets above the LDS paths' size (announced in r.seg.segs by the tile / deep kernel): those that fit one workgroup's
// LDS are sorted by k_seg_single, the rest by the segmented device-wide sort; V and F8 are written at their positions
template <class KeyFn>
static int large_phase(const KeyFn& keyfn, const TileRound& r, uint32_t nseg, uint64_t nlarge, hipStream_t st,
                       sfx_build_stats* stats)
{
    if (nseg == 0) return SFX_OK;
    // buckets that fit one tile of the segmented sort are sorted inside LDS by one workgroup each (SFX_SEG_SINGLE=0:
    // development, everything through the segmented passes)
    static const bool singles = [] { const char* e = dev_env("SFX_SEG_SINGLE"); return !e || atoi(e) != 0; }();
    const uint32_t te = seg_tile_elems(false);
    // three size classes: 256 threads x 16, 1024 x 11 (one tile of the segmented sort), 1024 x 16 (the LDS holds no more);
    // with the small tiles of the tests (4096) only the first
    const uint32_t top = !singles ? 0u : (te > 4096u ? 16384u : 4096u);
    if (singles) {
        const uint2* segs = reinterpret_cast<const uint2*>(r.seg.segs);
        const unsigned grid = (unsigned)dmin<uint64_t>(nseg, (uint64_t)grid_cap() * 2);
        // (most large buckets are small ones: up to 1024 members a workgroup needs 12 KB of LDS instead of 37 -- thirteen
        // of them share a CU instead of four, and the kernel waits on its gathers: config 3 13.0 -> 11.5 ms.  512 threads x 8
        // for 1025 .. 4096 members: 11.9)
        // (algorithmic bytes: the host knows the members of all size classes together, not of each -- the first launch
        // declares all of them, 13 bytes per member (suffix in, suffix + flag byte out, 4-byte key gathered), the others none,
        // so the sum over the launches is right; round 3 declared the total on every launch, four times too much)
        SFX_LAUNCH("seg_single_lds", (double)nlarge * 13, (k_seg_single<4, 4, KeyFn>), grid, 4 * kWave, st, keyfn, segs, nseg, 0u, 1024u,
                   r.V, r.F8, r.emit, r.Hd, r.wsym);
        SFX_LAUNCH("seg_single_lds", 0.0, (k_seg_single<4, 16, KeyFn>), grid, 4 * kWave, st, keyfn, segs, nseg, 1024u, 4096u,
                   r.V, r.F8, r.emit, r.Hd, r.wsym);
        if (top > 4096u) {
            SFX_LAUNCH("seg_single_lds", 0.0, (k_seg_single<16, 11, KeyFn>), dmin(grid, grid_cap()), 16 * kWave, st,
                       keyfn, segs, nseg, 4096u, 11264u, r.V, r.F8, r.emit, r.Hd, r.wsym);
            SFX_LAUNCH("seg_single_lds", 0.0, (k_seg_single<16, 16, KeyFn>), dmin(grid, grid_cap()), 16 * kWave, st,
                       keyfn, segs, nseg, 11264u, top, r.V, r.F8, r.emit, r.Hd, r.wsym);
        }
    }
    SFX_TRY(segmented_layout(r.seg, nseg, false, st, top));
    const unsigned grid = (unsigned)dmin<uint64_t>(nlarge / te + nseg, kMaxGrid);
    SFX_LAUNCH("seg_gather", (double)nlarge * 16, (k_seg_gather<KeyFn>), grid, kBlock, st, keyfn, (const uint32_t*)r.V,
               reinterpret_cast<SegTileHost*>(r.seg.tiles), (const uint32_t*)r.seg.counters, r.EA, (const uint16_t*)r.Hd);
    SFX_TRY(segmented_sort_e64(r.EA, r.EB, r.seg, nseg, nlarge, r.V, r.F8, st, stats, r.emit, r.Hd, r.wsym));
    if (stats) stats->large_sorted += nlarge;
    return SFX_OK;
}
static int flags_phase(const TileRound& r, uint64_t m, hipStream_t st)
{
    Chunking ch = make_chunking(m, kFlagChunkTile);
    SFX_LAUNCH("flags_reduce", (double)m * 1.25, k_flags_reduce, ch.blocks, kBlock, st, r.F8, m,
               ch.tiles_per_block * kFlagChunkTile, r.part_head, r.part_keep, r.part_ghead, r.F, r.part_pairs);
    return SFX_OK;
}

template <class KeyFn>
static int tile_round_impl(const KeyFn& keyfn, const TileRound& r, uint64_t m, hipStream_t st, sfx_build_stats* stats)
{
    if (m == 0) return SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    SFX_HIP(hipMemsetAsync(r.counters, 0, 3 * sizeof(unsigned long long), st));     // ([3]: gathers of the deep rounds so far)
    uint64_t tmax = 0;
    if (tile_small()) {
        SFX_TRY((launch_tile<4, 1, 32, KeyFn>(keyfn, r, m, st, &tmax)));
    } else {
        const int g = tile_geom(), pm = tile_pair();
#define SFX_TILE(NW, KPT) (pm == 32 ? launch_tile<NW, KPT, 32, KeyFn>(keyfn, r, m, st, &tmax) : launch_tile<NW, KPT, 64, KeyFn>(keyfn, r, m, st, &tmax))
        if (g == 1) SFX_TRY(SFX_TILE(16, 4));
        else if (g == 2) SFX_TRY(SFX_TILE(8, 8));
        else if (g == 3) SFX_TRY(SFX_TILE(8, 4));
        else if (g == 4) SFX_TRY(SFX_TILE(4, 4));
        else if (g == 5) SFX_TRY(SFX_TILE(4, 2));
        else SFX_TRY(SFX_TILE(16, 8));
#undef SFX_TILE
    }
    unsigned long long host[2] = {0, 0};                      // elements sorted in LDS, buckets left to the segmented sort
    SFX_TRY(read_back(host, r.counters, sizeof(host), st));
    if (host[0] > m || host[1] > m / (tmax + 1)) return SFX_ERR_INTERNAL;
    const uint64_t nlarge = m - host[0];
    const uint32_t nseg = (uint32_t)host[1];
    if ((nlarge == 0) != (nseg == 0)) return SFX_ERR_INTERNAL;
    if (stats) stats->tile_sorted += host[0];
    SFX_TRY(large_phase(keyfn, r, nseg, nlarge, st, stats));
    return flags_phase(r, m, st);
}

// ---- a deep text round -----------------------------------------------------------------------
// SFX_DEEP_KPT (development): window positions per lane of k_deep_wave, 4 / 8 (default) / 16; the emulator's small-tile
// hook selects 2 (128-position windows) so that small inputs reach the large path and the residue rules
static int deep_kpt()
{
    static const int v = [] { const char* e = dev_env("SFX_DEEP_KPT"); int x = e ? atoi(e) : 4; return (x == 4 || x == 8) ? x : 4; }();
    return v;
}
static int deep_max_iter()
{
    static const int v = [] { const char* e = dev_env("SFX_DEEP_ITERS"); int x = e ? atoi(e) : 24; return x >= 1 && x <= 4096 ? x : 24; }();
    return v;
}
// SFX_DEEP_MAX_DEPTH (tests): a lower bound on the depth from which a bucket leaves the deep kernel for good (the 16-bit
// limit of Hd otherwise), so that small inputs have buckets that leave beside buckets that go on in the same wave
static uint32_t deep_max_depth()
{
    static const uint32_t v = [] { const char* e = dev_env("SFX_DEEP_MAX_DEPTH"); int x = e ? atoi(e) : 0; return x >= 1 && x < (int)kDeepMaxDepth ? (uint32_t)x : kDeepMaxDepth; }();
    return v;
}
int deep_text_symbols(const PackedText& pt) { return text_key64_symbols(pt); }

template <int KPT>
static int launch_deep(const DeepTextKey& keyfn, const TileRound& r, uint64_t m, hipStream_t st, uint64_t* own_max)
{
    constexpr int W = kWave * KPT, H = W / 2;
    *own_max = W - H;
    const uint64_t blocks = (m + (uint64_t)H * kWavesPerBlock - 1) / ((uint64_t)H * kWavesPerBlock);
    if (blocks > 0x7FFFFFFFull) return SFX_ERR_TOO_LARGE;
    SFX_HIP(hipMemsetAsync(r.deep_slots, 0, (size_t)kDeepSlotWords * sizeof(unsigned long long), st));
    // read G (+ V + Hd of what it takes), write V + F8 (+ Hd of what stays tied); the gathers are counted by the kernel
    const double algo = (double)m * (4 + 4);
    // A bucket that leaves without another key must still be deeper than the h the NEXT round assumes of every bucket
    // (h + r.wsym, refine): the limit never lies below h + r.wsym + one key.  (The 16-bit limit itself is far above that: the
    // build switches to ranks before h + 2 wsym reaches 60000.)
    uint32_t max_depth = deep_max_depth();
    const uint64_t floor_depth = (uint64_t)r.h + r.wsym + (uint64_t)keyfn.wsym;
    if (floor_depth > kDeepMaxDepth) return SFX_ERR_NEEDS_RANKS;      // (refine switches to ranks long before: h + 2 wsym > 60000)
    if (max_depth < floor_depth) max_depth = (uint32_t)floor_depth;
    if (r.emit.lcp)
        SFX_LAUNCH("deep_wave", algo, (k_deep_wave<KPT, true>), (unsigned)blocks, kBlock, st, keyfn, r.G, m, r.V, r.F8, r.Hd,
                   r.counters, r.deep_slots, reinterpret_cast<uint2*>(r.seg.segs), r.emit, deep_max_iter(), max_depth);
    else
        SFX_LAUNCH("deep_wave", algo, (k_deep_wave<KPT, false>), (unsigned)blocks, kBlock, st, keyfn, r.G, m, r.V, r.F8, r.Hd,
                   r.counters, r.deep_slots, reinterpret_cast<uint2*>(r.seg.segs), r.emit, deep_max_iter(), max_depth);
    SFX_LAUNCH("deep_totals", 0.0, k_deep_totals, 1, kBlock, st, (const unsigned long long*)r.deep_slots, r.counters, 1);
    return SFX_OK;
}

// One text round over the active list.  Buckets of up to W / 2 members are FINISHED by k_deep_wave (every bucket carries
// its own depth r.Hd; what stays tied leaves with a larger one).  Larger buckets are split by their next
// text_round_symbols symbols at THEIR depth (32-bit key2, LDS / segmented sorts; their classes leave with that many more).
// r.counters[3] accumulates the gathers.
int deep_round_text(const PackedText& pt, const TileRound& r, uint64_t m, hipStream_t st, sfx_build_stats* stats)
{
    if (m == 0) return SFX_OK;
    if (m > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    const int wsym = text_key64_symbols(pt);
    const DeepTextKey keyfn = {pt, 2 * pt.kbits - wsym * pt.bits, wsym};
    SFX_HIP(hipMemsetAsync(r.counters, 0, 3 * sizeof(unsigned long long), st));
    uint64_t own_max = 0;
    if (tile_small()) SFX_TRY(launch_deep<2>(keyfn, r, m, st, &own_max));
    else if (deep_kpt() == 8) SFX_TRY(launch_deep<8>(keyfn, r, m, st, &own_max));
    else SFX_TRY(launch_deep<4>(keyfn, r, m, st, &own_max));
    unsigned long long host[2] = {0, 0};                      // members finished or refined by the waves, large buckets
    SFX_TRY(read_back(host, r.counters, sizeof(host), st));
    if (host[0] > m || host[1] > m / (own_max + 1)) return SFX_ERR_INTERNAL;
    const uint64_t nlarge = m - host[0];
    const uint32_t nseg = (uint32_t)host[1];
    if ((nlarge == 0) != (nseg == 0)) return SFX_ERR_INTERNAL;
    if (stats) stats->tile_sorted += host[0];
    SFX_TRY(large_phase(TextKey{pt, 0, pt.kbits == 32 ? pt.bits : 0}, r, nseg, nlarge, st, stats));
    return flags_phase(r, m, st);
}

int text_key64_symbols(const PackedText& pt);
LcpEmit make_lcp_emit(uint32_t* lcp, const uint32_t* S, const PackedText& pt, uint64_t h, bool rank_mode)
{
    LcpEmit e;
    e.lcp = lcp;
    e.S = S;
    e.h = (uint32_t)h;
    e.n = (uint32_t)pt.n;
    e.rank_mode = rank_mode ? 1 : 0;
    e.field_bits = pt.kbits == 32 ? 32 - pt.bits : pt.kbits;       // (see TextKey: a 32-bit word gives up its last symbol)
    e.inv_bits = (65536u + (unsigned)pt.bits - 1u) / (unsigned)pt.bits;
    e.field_bits64 = text_key64_symbols(pt) * pt.bits;
    return e;
}

int text_key64_symbols(const PackedText& pt)
{
    const int two = 2 * pt.spw, fit = 63 / pt.bits;
    return two < fit ? two : fit;
}

int tile_round_rank(const uint32_t* isa, uint64_t n, uint64_t h, const TileRound& r, uint64_t m, hipStream_t st,
                    sfx_build_stats* stats)
{
    return tile_round_impl(RankKey{isa, n, h}, r, m, st, stats);
}

}  // namespace sfx
