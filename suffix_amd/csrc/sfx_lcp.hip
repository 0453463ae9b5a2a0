// sfx_lcp.hip -- LCP array on the device (replaces lcp_lens ->
// lcp_lens_quadratic, /root/reference/src/table.rs:130-138, :348-365).
//
// The reference compares every adjacent pair of suffixes from scratch
// (quadratic on repetitive text) and builds an inverse suffix array it never
// uses (:131-134).  The engine computes the same array with the Phi/PLCP
// formulation of the linear algorithm the reference keeps commented out
// (:314-346), which is exact on bytes:
//   k_phi_scatter  phi[sa[r]] = sa[r-1]            (the predecessor in SA order)
//   k_plcp         PLCP in TEXT order: each thread walks 32 consecutive text
//                  positions carrying h; REDUCIBLE positions (text[i-1] ==
//                  text[phi[i]-1]) take PLCP[i-1] - 1 without a comparison, runs
//                  that start inside a reducible stretch get their base from a
//                  block scan, so only irreducible values are ever compared
//                  (O(n log n) symbols in total, whatever the text); phi/PLCP
//                  tiles are staged through LDS so global traffic is coalesced
//   k_lcp_gather   lcp[r] = PLCP[sa[r]]
// Algorithmic bytes per text byte: 8 (phi) + 4+4+2 (plcp) + 4+4+4 (gather) = 30.
//
// Three n-sized random-access passes are the price of linearity.  Where the text does not
// need it -- a sample of adjacent pairs says the mean LCP is a few dozen bytes -- the
// reference's own formulation is cheaper on this machine: k_lcp_windows compares every
// suffix with its predecessor directly (:348-361), one 16-byte gather per suffix (the
// predecessor's window comes from the neighbouring lane), capped at kDirectCap bytes; if any
// pair reaches the cap the Phi/PLCP path recomputes the array, so the worst case stays linear.
#include "sfx_host.hpp"

namespace sfx {

constexpr uint32_t kNoPhi = 0xFFFFFFFFu;
constexpr int kRun = 32;                         // consecutive text positions per thread
constexpr int kPlcpTile = kBlock * kRun;         // 8192 positions per workgroup step

// (entries of sa that are not text positions -- an unchecked from_parts table, src/table.rs:105-119 -- are
// counted in counters[2] and skipped by every kernel here: the call then fails with SFX_ERR_ARG)
__global__ void __launch_bounds__(kBlock)
k_phi_scatter(const uint32_t* __restrict__ sa, uint64_t n, uint32_t* __restrict__ phi, unsigned long long* __restrict__ counters)
{
    constexpr int U = 4;                     // independent scatters in flight per thread
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r0 < n; r0 += U * stride) {
        uint32_t cur[U], prev[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t r = r0 + u * stride;
            cur[u] = r < n ? sa[r] : 0u;
            prev[u] = (r < n && r) ? sa[r - 1] : kNoPhi;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (r0 + u * stride < n) {
                if (cur[u] < n && (prev[u] < n || prev[u] == kNoPhi)) phi[cur[u]] = prev[u];
                else atomicAdd(&counters[2], 1ull);
            }
        }
    }
}

// large texts: (position, predecessor) pairs in SA order for scatter_pairs_u32
__global__ void __launch_bounds__(kBlock)
k_phi_pairs(const uint32_t* __restrict__ sa, uint64_t n, uint64_t* __restrict__ pairs, unsigned long long* __restrict__ counters)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
        uint32_t cur = sa[r], prev = r ? sa[r - 1] : kNoPhi;
        if (cur >= n || (prev >= n && prev != kNoPhi)) {             // keep the scatter in bounds; the call fails anyway
            atomicAdd(&counters[2], 1ull);
            cur = (uint32_t)r;
            prev = kNoPhi;
        }
        pairs[r] = ((uint64_t)cur << 32) | (uint64_t)prev;
    }
}

// length of the common prefix of text[a..] and text[b..] beyond the first h bytes, 8 bytes
// per step where both windows are inside the text (unaligned 8-byte loads), bytes at the end
__device__ __forceinline__ uint64_t extend_match(const uint8_t* __restrict__ text, uint64_t n, uint64_t a,
                                                 uint64_t b, uint64_t h)
{
    while (a + h + 8 <= n && b + h + 8 <= n) {
        uint64_t x, y;
        __builtin_memcpy(&x, text + a + h, 8);
        __builtin_memcpy(&y, text + b + h, 8);
        const uint64_t d = x ^ y;
        if (d) return h + (uint64_t)(__ffsll((long long)d) - 1) / 8;      // little-endian: lowest differing byte
        h += 8;
    }
    while (a + h < n && b + h < n && text[a + h] == text[b + h]) h++;
    return h;
}

// PLCP in text order.  PLCP[i] is REDUCIBLE when text[i-1] == text[phi[i]-1]: then PLCP[i] = PLCP[i-1] - 1
// exactly, no comparison needed (Karkkainen, Manzini, Puglisi: the irreducible values sum to O(n log n),
// whatever the text).  A thread walks kRun consecutive positions carrying h; a reducible position whose
// predecessor's value is not known yet (the run started inside a reducible stretch) stays relative until the
// block scan below hands every run the value just before it.  Only the first position of a workgroup's chunk
// is always compared from scratch, so the worst case is one long comparison per chunk (<= kMaxGrid of them),
// not one per run: a unary text costs n reads, not n^2 / 512 comparisons.
constexpr uint32_t kPlcpRel = 0xFFFFFFFFu;
// (flag << 32 | v): flag 1 = "the value after this stretch is v"; flag 0 = "it is (value before) - v"
__device__ __forceinline__ uint64_t plcp_combine(uint64_t a, uint64_t b)
{
    if (b >> 32) return b;
    const uint32_t av = (uint32_t)a, bv = (uint32_t)b;
    return (a >> 32) ? ((1ull << 32) | (uint64_t)(av - bv)) : (uint64_t)(av + bv);
}
__global__ void __launch_bounds__(kBlock)
k_plcp(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ phi_plcp,
       uint64_t tiles_per_block)
{
    __shared__ uint32_t s[kBlock * (kRun + 1)];          // +1 pad: conflict-free per-thread rows
    __shared__ uint64_t wsum[kWavesPerBlock];
    __shared__ uint32_t tile_carry;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    uint64_t begin = (uint64_t)blockIdx.x * tiles_per_block * kPlcpTile;
    uint64_t end = begin + tiles_per_block * kPlcpTile;
    if (end > n) end = n;
    uint32_t carry = 0;                                  // PLCP of the position just before the tile (known from the 2nd tile on)
    for (uint64_t tile = begin; tile < end; tile += kPlcpTile) {
        for (unsigned j = tid; j < (unsigned)kPlcpTile; j += kBlock) {
            uint64_t g = tile + j;
            s[(j / kRun) * (kRun + 1) + (j % kRun)] = (g < end) ? phi_plcp[g] : kNoPhi;
        }
        __syncthreads();
        // the thread's run: values, or kPlcpRel while the run is still relative to what precedes it
        bool known = false;
        uint64_t h = 0;
        int first_known = kRun;
        uint32_t steps = 0;                              // positions walked (the relative decrement of an all-relative run)
        for (int k = 0; k < kRun; k++) {
            const uint64_t i = tile + (uint64_t)tid * kRun + k;
            if (i >= end) break;
            steps++;
            uint32_t j = s[tid * (kRun + 1) + k];
            if (j >= n) j = kNoPhi;                      // (also: a slot an invalid table never wrote)
            uint32_t val;
            if (j == kNoPhi) {                           // first suffix of the array: no predecessor
                h = 0; known = true; val = 0;
            } else if (i > 0 && j > 0 && i != begin && text[i - 1] == text[j - 1]) {
                if (known) { h--; val = (uint32_t)h; }   // reducible: PLCP[i] = PLCP[i-1] - 1
                else val = kPlcpRel;
            } else {                                     // irreducible (or the chunk's first position): compare
                h = extend_match(text, n, i, (uint64_t)j, known && h ? h - 1 : 0);
                known = true;
                val = (uint32_t)h;
            }
            if (known && first_known == kRun) first_known = k;
            s[tid * (kRun + 1) + k] = val;
        }
        // exclusive scan of the runs' summaries -> the value just before every run
        const uint64_t mine = steps == 0 ? 0ull : (known ? ((1ull << 32) | (uint64_t)(uint32_t)h) : (uint64_t)steps);
        uint64_t incl = mine;
#pragma unroll
        for (unsigned d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d);
            if (lane >= d) incl = plcp_combine(o, incl);
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint64_t before = (1ull << 32) | (uint64_t)carry;            // what precedes the tile is always absolute
        for (unsigned k = 0; k < w; k++) before = plcp_combine(before, wsum[k]);
        uint64_t prev = __shfl_up(incl, 1u);
        if (lane == 0) prev = 0ull;                                  // (identity: relative, decrement 0)
        const uint64_t excl = plcp_combine(before, prev);            // absolute: `before` is
        const uint32_t base = (uint32_t)excl;                        // PLCP of the position before my run
        for (int k = 0; k < first_known && k < (int)steps; k++) s[tid * (kRun + 1) + k] = base - (uint32_t)(k + 1);
        if (tid == kBlock - 1) {
            uint64_t all = before;
            for (unsigned k = w; k < (unsigned)kWavesPerBlock; k++) all = plcp_combine(all, wsum[k]);
            tile_carry = (uint32_t)all;
        }
        __syncthreads();
        carry = tile_carry;
        for (unsigned j = tid; j < (unsigned)kPlcpTile; j += kBlock) {
            uint64_t g = tile + j;
            if (g < end) phi_plcp[g] = s[(j / kRun) * (kRun + 1) + (j % kRun)];
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock)
k_lcp_gather(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ plcp, uint64_t n,
             uint32_t* __restrict__ lcp)
{
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r0 < n; r0 += U * stride) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (r0 + u * stride < n) ? sa[r0 + u * stride] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (r0 + u * stride < n && v[u] < n) ? plcp[v[u]] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (r0 + u * stride < n) lcp[r0 + u * stride] = v[u];
    }
}

// first 8 * NWORDS bytes of suffix s, zero-padded past the end of the text
template <int NWORDS>
__device__ __forceinline__ void load_window(const uint8_t* __restrict__ text, uint64_t n, uint64_t s, uint64_t (&w)[NWORDS])
{
    if (s + 8 * NWORDS <= n) {
#pragma unroll
        for (int k = 0; k < NWORDS; k++) __builtin_memcpy(&w[k], text + s + 8 * k, 8);
    } else {
#pragma unroll
        for (int k = 0; k < NWORDS; k++) {
            uint64_t x = 0;
            for (unsigned b = 0; b < 8 && s + 8u * (unsigned)k + b < n; b++) x |= (uint64_t)text[s + 8u * (unsigned)k + b] << (8 * b);
            w[k] = x;
        }
    }
}
template <int NWORDS>
__device__ __forceinline__ unsigned match_windows(const uint64_t (&a)[NWORDS], const uint64_t (&b)[NWORDS])
{
    unsigned l = 8u * NWORDS;
#pragma unroll
    for (int k = NWORDS - 1; k >= 0; k--) {
        const uint64_t d = a[k] ^ b[k];
        if (d) l = 8u * (unsigned)k + (unsigned)(__ffsll((long long)d) - 1) / 8u;
    }
    return l;
}
// extend_match that gives up at `cap` bytes (returns a value >= cap then)
__device__ __forceinline__ uint64_t extend_match_capped(const uint8_t* __restrict__ text, uint64_t n, uint64_t a,
                                                        uint64_t b, uint64_t h, uint64_t cap)
{
    while (h < cap && a + h + 8 <= n && b + h + 8 <= n) {
        uint64_t x, y;
        __builtin_memcpy(&x, text + a + h, 8);
        __builtin_memcpy(&y, text + b + h, 8);
        const uint64_t d = x ^ y;
        if (d) return h + (uint64_t)(__ffsll((long long)d) - 1) / 8;
        h += 8;
    }
    while (h < cap && a + h < n && b + h < n && text[a + h] == text[b + h]) h++;
    return h;
}

constexpr uint32_t kDirectCap = 1024;            // bytes compared directly before a pair is handed to Phi/PLCP
constexpr uint32_t kSampleCap = 4096;
constexpr uint32_t kSamples = 1u << 16;          // adjacent pairs sampled to choose the path
constexpr uint64_t kSampleMeanMax = 64;          // direct path if the sampled mean LCP is at most this many bytes
constexpr uint64_t kDirectMinN = 1ull << 20;     // below: the three launches of the Phi path, no host round trip

// counters[0] += LCP of `samples` evenly spaced adjacent pairs (capped)
__global__ void __launch_bounds__(kBlock)
k_lcp_sample(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint32_t samples,
             uint64_t every, unsigned long long* __restrict__ counters)
{
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t l = 0;
    if (j < samples) {
        const uint64_t r = 1 + j * every;
        if (r < n) {
            const uint64_t a = sa[r - 1], b = sa[r];                 // (an invalid table is reported by k_sa_range_check)
            if (a < n && b < n) l = extend_match_capped(text, n, a, b, 0, kSampleCap);
        }
    }
    for (int d = 32; d >= 1; d >>= 1) l += __shfl_xor(l, d);
    if (lane_id() == 0 && l) atomicAdd(&counters[0], (unsigned long long)l);
}

// lcp[r] for every r; counters[1] counts the pairs that reached the cap (their lcp[r] is not final).
// NWORDS: 8-byte words of a suffix's window.  The window is what a pair is decided on without leaving the wave's lock step:
// a pair that agrees on all of it walks on alone (extend_match_capped: a dependent load-compare loop in one lane while the
// wave waits).  With 16 bytes about a third of the pairs of natural-language text walk on (mean LCP 13.7), with 32 a few per
// cent -- and the second half of a window lies in the line the first half fetched, or the next one.
template <int NWORDS>
__global__ void __launch_bounds__(kBlock)
k_lcp_windows(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa,
              uint32_t* __restrict__ lcp, unsigned long long* __restrict__ counters)
{
    constexpr unsigned kBytes = 8u * NWORDS;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const unsigned lane = lane_id();
    uint32_t capped = 0;
    // (whole waves stay in the loop: the predecessor's window is taken from the lane below)
    for (uint64_t r0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); r0 < n; r0 += stride) {
        const uint64_t r = r0 + lane;
        const bool live = r < n;
        uint64_t cur = live ? (uint64_t)sa[r] : 0;
        if (cur >= n) { atomicAdd(&counters[2], 1ull); cur = 0; }
        uint64_t c[NWORDS], p[NWORDS];
#pragma unroll
        for (int k = 0; k < NWORDS; k++) c[k] = 0;
        if (live) load_window<NWORDS>(text, n, cur, c);
        uint64_t prev = __shfl_up(cur, 1u);
#pragma unroll
        for (int k = 0; k < NWORDS; k++) p[k] = __shfl_up(c[k], 1u);
        if (lane == 0 && live && r > 0) {
            prev = (uint64_t)sa[r - 1];
            if (prev >= n) prev = 0;                                 // (counted by the lane that owns r - 1)
            load_window<NWORDS>(text, n, prev, p);
        }
        if (!live) continue;
        uint64_t l = 0;
        if (r > 0) {
            const uint64_t room = n - (cur > prev ? cur : prev);          // bytes the shorter suffix has
            l = match_windows<NWORDS>(p, c);
            if (l > room) l = room;
            if (l == kBytes) l = extend_match_capped(text, n, prev, cur, kBytes, kDirectCap);
            if (l >= kDirectCap) capped++;
        }
        lcp[r] = (uint32_t)l;
    }
    if (__any(capped != 0)) {
        for (int d = 32; d >= 1; d >>= 1) capped += __shfl_xor(capped, d);
        if (lane == 0) atomicAdd(&counters[1], (unsigned long long)capped);
    }
}

// The same on packed symbol codes (alphabets of <= 4 bits: DNA packs 32 symbols into the 64-bit
// window, and the packed text of a 1 GB genome is 250 MB -- Infinity-Cache sized -- where the raw
// text makes every gather an HBM line fetch).  Lengths are symbols = bytes of the text.
__global__ void __launch_bounds__(kBlock)
k_lcp_windows_packed(PackedText t, const uint32_t* __restrict__ sa, uint32_t* __restrict__ lcp,
                     unsigned long long* __restrict__ counters)
{
    const uint64_t n = t.n;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const unsigned lane = lane_id();
    const unsigned W = 2u * (unsigned)t.spw;                       // symbols per window
    const unsigned pad = 64u - 2u * (unsigned)t.kbits;             // unused high bits of a window
    const unsigned inv_bits = (65536u + (unsigned)t.bits - 1u) / (unsigned)t.bits;   // x / bits for x < 64, bits <= 4
    auto common = [&](uint64_t a, uint64_t b) -> unsigned {        // equal leading symbols of two windows
        const uint64_t x = a ^ b;
        return x ? (((unsigned)__clzll((long long)x) - pad) * inv_bits) >> 16 : W;
    };
    uint32_t capped = 0;
    for (uint64_t r0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); r0 < n; r0 += stride) {
        const uint64_t r = r0 + lane;
        const bool live = r < n;
        uint64_t cur = live ? (uint64_t)sa[r] : 0;
        if (cur >= n) { atomicAdd(&counters[2], 1ull); cur = 0; }
        const uint64_t kc = live ? packed_key64(t, cur) : 0;
        uint64_t prev = __shfl_up(cur, 1u), kp = __shfl_up(kc, 1u);
        if (lane == 0 && live && r > 0) {
            prev = (uint64_t)sa[r - 1];
            if (prev >= n) prev = 0;
            kp = packed_key64(t, prev);
        }
        if (!live) continue;
        uint64_t l = 0;
        if (r > 0) {
            const uint64_t room = n - (cur > prev ? cur : prev);
            l = common(kp, kc);
            if (l == W) {
                while (l < room && l < kDirectCap) {
                    const unsigned m = common(packed_key64(t, prev + l), packed_key64(t, cur + l));
                    l += m;
                    if (m < W) break;
                }
            }
            if (l > room) l = room;
            if (l >= kDirectCap) capped++;
        }
        lcp[r] = (uint32_t)l;
    }
    if (__any(capped != 0)) {
        for (int d = 32; d >= 1; d >>= 1) capped += __shfl_xor(capped, d);
        if (lane == 0) atomicAdd(&counters[1], (unsigned long long)capped);
    }
}

// ---- fused SA + LCP: the pairs the initial sort could not tell apart ------------------------------
// lcp[r] = kLcpBoundFlag | b: the pair (r - 1, r) shares at least b symbols (unless one of them ends earlier)
__global__ void __launch_bounds__(kBlock)
k_lcp_pending(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint32_t* __restrict__ lcp,
              unsigned long long* __restrict__ counters)
{
    // A lane reads eight entries at a time (32 bytes, the wave 2 KB in a row) and then works off the pending ones among them: a
    // few per cent of the entries are pending, and with one entry per lane and iteration nearly every iteration of a wave had a
    // lane in the dependent chain array -> text -> compare while 63 waited (round 6: 0.28 -> 0.1 ms per 10^8 on uniform DNA).
    constexpr unsigned kPer = 8;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock * kPer;
    uint32_t capped = 0;
    for (uint64_t r0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * kPer; r0 < n; r0 += stride) {
        uint32_t v[kPer];
        if (r0 + kPer <= n) {
            const uint4 x = *reinterpret_cast<const uint4*>(lcp + r0), y = *reinterpret_cast<const uint4*>(lcp + r0 + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
        } else {
#pragma unroll
            for (unsigned k = 0; k < kPer; k++) v[k] = r0 + k < n ? lcp[r0 + k] : 0u;
        }
        unsigned pend = 0;
#pragma unroll
        for (unsigned k = 0; k < kPer; k++) pend |= ((v[k] & kLcpBoundFlag) ? 1u : 0u) << k;
        while (pend) {
            const unsigned k = (unsigned)__ffs((int)pend) - 1u;
            pend &= pend - 1u;
            uint32_t vk = v[0];
#pragma unroll
            for (unsigned j = 1; j < kPer; j++) vk = k == j ? v[j] : vk;
            const uint64_t r = r0 + k;
            const uint64_t h0 = vk & ~kLcpBoundFlag;
            const uint64_t a = sa[r - 1], b = sa[r];                     // (r = 0 is never pending)
            const uint64_t h = (a + h0 <= n && b + h0 <= n) ? h0 : 0;    // equal keys = equal symbols, padding aside
            const uint64_t l = extend_match_capped(text, n, a, b, h, h + kDirectCap);
            if (l >= h + kDirectCap) capped++;
            lcp[r] = (uint32_t)l;
        }
    }
    if (capped) atomicAdd(&counters[1], (unsigned long long)capped);
}
// The keys of suffixes shorter than the key are zero-padded, so a key comparison may overstate their
// common prefix with a neighbour.  There are fewer than h0 such suffixes: find each one's rank by
// binary search and redo its two LCP entries on the text.
__device__ __forceinline__ bool suffix_less(const uint8_t* __restrict__ text, uint64_t n, uint64_t a, uint64_t b)
{
    if (a == b) return false;
    const uint64_t c = extend_match(text, n, a, b, 0);
    if (a + c >= n) return true;                                     // a is a proper prefix of b: shorter first
    if (b + c >= n) return false;
    return text[a + c] < text[b + c];
}
__global__ void __launch_bounds__(kBlock)
k_lcp_tail_fix(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint32_t* __restrict__ lcp,
               uint64_t h0)
{
    // One WAVE per short suffix i = n - k: its rank by a 64-ary search -- every step the lanes probe 64 ranks spread over what is
    // left of [lo, hi) and a ballot says where "the suffix at that rank is less than suffix i" ends (the predicate is monotone
    // over the ranks): 5 steps of two dependent loads instead of the 27 of a lane's own binary search (0.07 -> 0.02 ms).
    const unsigned lane = lane_id();
    const uint64_t k = (uint64_t)blockIdx.x * kWavesPerBlock + wave_id() + 1;
    if (k >= h0 || k > n) return;                                     // (the whole wave)
    const uint64_t i = n - k;
    uint64_t lo = 0, hi = n;                                          // first rank whose suffix is not less than suffix i: in [lo, hi]
    while (lo < hi) {
        const uint64_t span = hi - lo;
        uint64_t probe;
        if (span <= kWave) probe = lo + lane;                         // every rank of [lo, hi)
        else probe = lo + (span * (uint64_t)(lane + 1u)) / (uint64_t)(kWave + 1u);   // 64 ranks strictly inside, ascending
        const bool less = probe < hi && suffix_less(text, n, (uint64_t)sa[probe], i);
        const unsigned long long bal = __ballot(less);
        const unsigned nless = (unsigned)__popcll(bal);               // (monotone: the lanes that say "less" are the first nless)
        if (span <= kWave) { lo = lo + nless; hi = lo; break; }
        const uint64_t below = nless ? lo + (span * (uint64_t)nless) / (uint64_t)(kWave + 1u) : lo;          // last probe that was less (or lo)
        const uint64_t above = nless < kWave ? lo + (span * (uint64_t)(nless + 1u)) / (uint64_t)(kWave + 1u) : hi;   // first probe that was not
        lo = nless ? below + 1 : lo;
        hi = above;
    }
    const uint64_t r = lo;
    if (lane != 0) return;
    if (r >= n || sa[r] != (uint32_t)i) return;                      // (cannot happen on a valid suffix array)
    lcp[r] = r ? (uint32_t)extend_match(text, n, (uint64_t)sa[r - 1], i, 0) : 0u;
    if (r + 1 < n) lcp[r + 1] = (uint32_t)extend_match(text, n, i, (uint64_t)sa[r + 1], 0);
}

int lcp_finish_pending_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint32_t* d_lcp, uint64_t h0,
                           void* ws, uint64_t ws_bytes, hipStream_t st, bool* done)
{
    *done = false;
    if (!ws || ws_bytes < 64) return SFX_ERR_WORKSPACE;
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(ws);
    unsigned long long host[2] = {0, 0};
    SFX_HIP(hipMemsetAsync(counters, 0, sizeof(host), st));
    const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("lcp_pending", (double)n * 4, k_lcp_pending, grid, kBlock, st, d_text, n, d_sa, d_lcp, counters);
    SFX_LAUNCH("lcp_tail_fix", 0.0, k_lcp_tail_fix, (unsigned)((h0 + kWavesPerBlock - 1) / kWavesPerBlock), kBlock, st, d_text, n, d_sa,
               d_lcp, h0);
    SFX_TRY(read_back(host, counters, sizeof(host), st));
    *done = host[1] == 0;
    return SFX_OK;
}

// LCP of one contiguous SLICE of the suffix array (range-partitioned index): every rank
// compares each suffix of its slice with its predecessor directly, 8 bytes per step --
// the reference's own formulation (lcp_lens_quadratic :348-361), which is the right one
// for the low-LCP texts the partitioned build is meant for, needs no inverse permutation
// and no data from other ranks except the last suffix of the previous slice.
__global__ void __launch_bounds__(kBlock)
k_lcp_direct(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint64_t count,
             uint32_t prev_suffix, uint32_t* __restrict__ lcp)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < count; r += stride) {
        const uint32_t cur = sa[r];
        const uint32_t prev = r ? sa[r - 1] : prev_suffix;
        lcp[r] = (prev == kNoPhi) ? 0u : (uint32_t)extend_match(text, n, (uint64_t)prev, (uint64_t)cur, 0);
    }
}

int build_lcp_range_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa_part, uint64_t count,
                            uint32_t prev_suffix, uint32_t* d_lcp_part, hipStream_t st)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (count == 0) return SFX_OK;
    if (!d_text || !d_sa_part || !d_lcp_part || count > n) return SFX_ERR_ARG;
    unsigned grid = (unsigned)dmin<uint64_t>((count + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("lcp_direct", (double)count * 12, k_lcp_direct, grid, kBlock, st, d_text, n, d_sa_part, count,
               prev_suffix, d_lcp_part);
    return SFX_OK;
}

// u32 -> u64 index arrays (BASELINE config 4 asks for u64 indices; positions fit u32, :380)
__global__ void __launch_bounds__(kBlock)
k_widen(const uint32_t* __restrict__ in, uint64_t count, uint64_t* __restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) out[i] = in[i];
}
int widen_u32_to_u64_dev(const uint32_t* d_in, uint64_t count, uint64_t* d_out, hipStream_t st)
{
    if (count == 0) return SFX_OK;
    if (!d_in || !d_out) return SFX_ERR_ARG;
    unsigned grid = (unsigned)dmin<uint64_t>((count + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("widen_u64", (double)count * 12, k_widen, grid, kBlock, st, d_in, count, d_out);
    return SFX_OK;
}

// SFX_LCP_DIRECT_MIN=<n> is a test hook (the direct path from n bytes up; default 2^20)
static uint64_t direct_lcp_min()
{
    static const uint64_t v = [] {
        const char* e = dev_env("SFX_LCP_DIRECT_MIN");
        long long x = e ? atoll(e) : 0;
        return x >= 8 ? (uint64_t)x : kDirectMinN;              // (the counters borrow 16 bytes of the 4n Phi array)
    }();
    return v;
}

// phi / PLCP (4n); for large texts also the pair buffers and radix scratch of the
// partitioned Phi scatter
uint64_t lcp_workspace_bytes(uint64_t n)
{
    ArenaSizer a;
    a.take<uint32_t>(n);
    a.take<uint8_t>(4096);                       // direct path: alphabet scratch + packed text (sigma <= 16)
    a.take<uint32_t>(n / 8 + 8);
    if (n >= partitioned_scatter_min()) {
        a.take<uint64_t>(n);
        a.take<uint64_t>(n);
        a.take<uint32_t>(radix_scratch_words(n));
    }
    return a.used + 256;
}

int build_lcp_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint32_t* d_lcp,
                      void* ws, uint64_t ws_bytes, hipStream_t st)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!d_text || !d_sa || !d_lcp) return SFX_ERR_ARG;
    if (!ws || ws_bytes < lcp_workspace_bytes(n)) return SFX_ERR_WORKSPACE;
    Arena ar(ws, ws_bytes);
    uint32_t* phi = ar.take<uint32_t>(n);
    uint8_t* small = ar.take<uint8_t>(4096);
    uint32_t* packed_words = ar.take<uint32_t>(n / 8 + 8);
    if (ar.overflow) return SFX_ERR_WORKSPACE;
    unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(small + 3072);  // (past the alphabet bins and LUT)
    unsigned long long host[3] = {0, 0, 0};
    SFX_HIP(hipMemsetAsync(counters, 0, sizeof(host), st));
    if (n >= direct_lcp_min()) {
        // low-LCP text (by a sample of adjacent pairs): compare directly, fall through to the
        // linear path only if some pair reached the cap
        const uint32_t samples = (uint32_t)dmin<uint64_t>(kSamples, n - 1);
        const uint64_t every = (n - 1) / samples;
        SFX_LAUNCH("lcp_sample", (double)samples * 24, k_lcp_sample, (samples + kBlock - 1) / kBlock, kBlock, st, d_text,
                   n, d_sa, samples, every, counters);
        SFX_TRY(read_back(host, counters, sizeof(host), st));
        if (host[0] <= kSampleMeanMax * samples) {
            PackedText pt;
            bool packed = false;
            SFX_TRY(pack_small_alphabet(d_text, n, 4, small, packed_words, st, &pt, &packed));
            if (packed)
                SFX_LAUNCH("lcp_windows_packed", (double)n * 24, k_lcp_windows_packed, grid, kBlock, st, pt, d_sa, d_lcp,
                           counters);
            else {
                // window of 32 bytes where the sample says a 16-byte window would leave most pairs to walk on alone: sampled mean
                // LCP >= 16 bytes.  Measured on 10^9 bytes (round 5, profiles/r5_lcp_window_ab.jsonl): mixed-script UTF-8 (mean
                // 21.7) 37.0 -> 33.0 ms, English-like text (mean 13.7) 28.9 -> 29.6 ms -- below the threshold the second half of
                // the window is two more load instructions for nothing.  (SFX_LCP_WINDOW=2 / 4, development: force 16 / 32 bytes)
                static const int forced = [] { const char* e = dev_env("SFX_LCP_WINDOW"); return e ? atoi(e) : 0; }();
                const bool wide = forced ? forced == 4 : host[0] >= 16ull * samples;
                if (wide)
                    SFX_LAUNCH("lcp_windows", (double)n * 24, k_lcp_windows<4>, grid, kBlock, st, d_text, n, d_sa, d_lcp, counters);
                else
                    SFX_LAUNCH("lcp_windows", (double)n * 24, k_lcp_windows<2>, grid, kBlock, st, d_text, n, d_sa, d_lcp, counters);
            }
            SFX_TRY(read_back(host, counters, sizeof(host), st));
            if (host[2]) return SFX_ERR_ARG;
            if (host[1] == 0) return SFX_OK;
        }
    }
    if (n >= partitioned_scatter_min()) {
        uint64_t* pairs = ar.take<uint64_t>(n);
        uint64_t* tmp = ar.take<uint64_t>(n);
        uint32_t* scratch = ar.take<uint32_t>(radix_scratch_words(n));
        if (ar.overflow) return SFX_ERR_WORKSPACE;
        SFX_LAUNCH("phi_pairs", (double)n * 12, k_phi_pairs, grid, kBlock, st, d_sa, n, pairs, counters);
        SFX_TRY(scatter_pairs_u32(pairs, tmp, n, n, phi, scratch, st, nullptr));
    } else {
        SFX_LAUNCH("phi_scatter", (double)n * 8, k_phi_scatter, grid, kBlock, st, d_sa, n, phi, counters);
    }
    Chunking ch = make_chunking(n, kPlcpTile);
    SFX_LAUNCH("plcp", (double)n * 10, k_plcp, ch.blocks, kBlock, st, d_text, n, phi,
               ch.tiles_per_block);
    SFX_LAUNCH("lcp_gather", (double)n * 12, k_lcp_gather, grid, kBlock, st, d_sa, phi, n, d_lcp);
    SFX_TRY(read_back(host, counters, sizeof(host), st));            // (an invalid table: every kernel above skipped it)
    if (host[2]) return SFX_ERR_ARG;
    return SFX_OK;
}

}  // namespace sfx
