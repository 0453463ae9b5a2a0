// sfx_query.hip -- batched positions() / contains() / any_position()
// (/root/reference/src/table.rs:223-293, binary_search :900-914).
//
// One query per lane.  positions(q) is the half-open SA interval
//   start = #suffixes < q            (first suffix with q <= suffix,   :244-246)
//   end   = start + #suffixes having q as a prefix                     (:247-250)
// The reference's range early-outs (:228-235) are pure optimisations and never
// change the result, so the kernel only keeps the empty-text / empty-query case.
// contains(q) == (end > start); any_position(q) returns table[start] -- the
// reference documents the choice as arbitrary (:261-262).
// Algorithmic bytes per query: 2*ceil(log2 n) probes x (4 B SA entry + compared bytes).
#include "sfx_host.hpp"

namespace sfx {

// Three-way comparison of query q[0..m) with the first min(m, n - s) bytes of the suffix at s,
// 8 bytes per step (unaligned 8-byte loads; little-endian, so the first differing byte is the
// lowest non-zero byte of the XOR).  <0: q smaller, >0: q larger, 0: equal on those bytes.
__device__ __forceinline__ int compare_query(const uint8_t* __restrict__ q, uint64_t m,
                                             const uint8_t* __restrict__ text, uint64_t n, uint64_t s)
{
    const uint64_t len = n - s, lim = m < len ? m : len;
    uint64_t k = 0;
    while (k + 8 <= lim) {
        uint64_t x, y;
        __builtin_memcpy(&x, q + k, 8);
        __builtin_memcpy(&y, text + s + k, 8);
        const uint64_t d = x ^ y;
        if (d) {
            const unsigned sh = (unsigned)(__ffsll((long long)d) - 1) & ~7u;
            return (int)((x >> sh) & 0xFFu) - (int)((y >> sh) & 0xFFu);
        }
        k += 8;
    }
    while (k < lim) {
        const int a = q[k], b = text[s + k];
        if (a != b) return a - b;
        k++;
    }
    return 0;
}
// query <= suffix ?  (Rust slice Ord: lexicographic, a proper prefix is smaller)
__device__ __forceinline__ bool query_le_suffix(const uint8_t* __restrict__ q, uint64_t m,
                                                const uint8_t* __restrict__ text, uint64_t n,
                                                uint64_t s)
{
    const int c = compare_query(q, m, text, n, s);
    return c ? c < 0 : m <= n - s;
}
__device__ __forceinline__ bool suffix_starts_with(const uint8_t* __restrict__ q, uint64_t m,
                                                   const uint8_t* __restrict__ text, uint64_t n,
                                                   uint64_t s)
{
    return n - s >= m && compare_query(q, m, text, n, s) == 0;
}

__global__ void __launch_bounds__(kBlock)
k_query_batch(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint64_t sa_len,
              const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, uint64_t nq,
              uint32_t* __restrict__ start_out, uint32_t* __restrict__ end_out,
              uint8_t* __restrict__ found_out, uint32_t* __restrict__ any_out)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < nq; k += stride) {
        const uint8_t* q = qbytes + qoff[k];
        uint64_t m = qoff[k + 1] - qoff[k];
        uint64_t start = 0, end = 0;
        if (sa_len != 0 && m != 0) {                                // :228-229
            uint64_t lo = 0, hi = sa_len;                           // :244-246, :900-914
            while (lo < hi) {
                uint64_t mid = (lo + hi) >> 1;
                if (query_le_suffix(q, m, text, n, sa[mid])) hi = mid; else lo = mid + 1;
            }
            start = lo;
            // :247-250 is a binary search of "starts with q" over table[start..]; the predicate is
            // monotone there (true on a prefix of it), so galloping from `start` finds the same end
            // in ~2 log2(#matches) probes instead of log2(n) -- most queries match a few suffixes
            uint64_t cnt = 0;                                       // matches known so far
            if (start < sa_len && suffix_starts_with(q, m, text, n, sa[start])) {
                cnt = 1;
                uint64_t step = 1;
                while (start + cnt - 1 + step < sa_len &&
                       suffix_starts_with(q, m, text, n, sa[start + cnt - 1 + step])) {
                    cnt += step;
                    step <<= 1;
                }
                // the end lies in (start + cnt - 1, min(sa_len, start + cnt - 1 + step)]
                lo = start + cnt;
                hi = dmin<uint64_t>(sa_len, start + cnt - 1 + step);
                while (lo < hi) {
                    uint64_t mid = (lo + hi) >> 1;
                    if (!suffix_starts_with(q, m, text, n, sa[mid])) hi = mid; else lo = mid + 1;
                }
                cnt = lo - start;
            }
            end = start + cnt;
        }
        bool found = end > start;
        if (!found) start = end = 0;
        if (start_out) start_out[k] = (uint32_t)start;
        if (end_out) end_out[k] = (uint32_t)end;
        if (found_out) found_out[k] = found ? 1 : 0;
        if (any_out) any_out[k] = found ? sa[start] : 0xFFFFFFFFu;
    }
}

// sa_len == n: the whole suffix array.  sa_len < n: a contiguous SLICE of it (one rank of
// the range-partitioned index); start/end are then positions inside the slice.
int query_batch_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint64_t sa_len, const uint8_t* d_q,
                    const uint64_t* d_qoff, uint64_t nq, uint32_t* d_start, uint32_t* d_end,
                    uint8_t* d_found, uint32_t* d_any, hipStream_t st)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (sa_len > n) return SFX_ERR_ARG;
    if (nq == 0) return SFX_OK;
    if (!d_qoff || (sa_len && (!d_text || !d_sa))) return SFX_ERR_ARG;
    unsigned grid = (unsigned)dmin<uint64_t>((nq + kBlock - 1) / kBlock, kMaxGrid);
    double probes = 2.0 * bits_for(sa_len ? sa_len : 1);
    SFX_LAUNCH("query_batch", (double)nq * probes * 12.0, k_query_batch, grid, kBlock, st, d_text, n,
               d_sa, sa_len, d_q, d_qoff, nq, d_start, d_end, d_found, d_any);
    return SFX_OK;
}

}  // namespace sfx
