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
//                  positions carrying h (PLCP[i+1] >= PLCP[i]-1), restarting
//                  from h=0 only at the start of its run; phi/PLCP tiles are
//                  staged through LDS so global traffic is coalesced
//   k_lcp_gather   lcp[r] = PLCP[sa[r]]
// Algorithmic bytes per text byte: 8 (phi) + 4+4+2 (plcp) + 4+4+4 (gather) = 30.
#include "sfx_host.hpp"

namespace sfx {

constexpr uint32_t kNoPhi = 0xFFFFFFFFu;
constexpr int kRun = 32;                         // consecutive text positions per thread
constexpr int kPlcpTile = kBlock * kRun;         // 8192 positions per workgroup step

__global__ void __launch_bounds__(kBlock)
k_phi_scatter(const uint32_t* __restrict__ sa, uint64_t n, uint32_t* __restrict__ phi)
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
        for (int u = 0; u < U; u++)
            if (r0 + u * stride < n) phi[cur[u]] = prev[u];
    }
}

// large texts: (position, predecessor) pairs in SA order for scatter_pairs_u32
__global__ void __launch_bounds__(kBlock)
k_phi_pairs(const uint32_t* __restrict__ sa, uint64_t n, uint64_t* __restrict__ pairs)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride)
        pairs[r] = ((uint64_t)sa[r] << 32) | (uint64_t)(r ? sa[r - 1] : kNoPhi);
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

__global__ void __launch_bounds__(kBlock)
k_plcp(const uint8_t* __restrict__ text, uint64_t n, uint32_t* __restrict__ phi_plcp,
       uint64_t tiles_per_block)
{
    __shared__ uint32_t s[kBlock * (kRun + 1)];          // +1 pad: conflict-free per-thread rows
    const unsigned tid = threadIdx.x;
    uint64_t begin = (uint64_t)blockIdx.x * tiles_per_block * kPlcpTile;
    uint64_t end = begin + tiles_per_block * kPlcpTile;
    if (end > n) end = n;
    for (uint64_t tile = begin; tile < end; tile += kPlcpTile) {
        for (unsigned j = tid; j < (unsigned)kPlcpTile; j += kBlock) {
            uint64_t g = tile + j;
            s[(j / kRun) * (kRun + 1) + (j % kRun)] = (g < end) ? phi_plcp[g] : kNoPhi;
        }
        __syncthreads();
        uint64_t h = 0;
        for (int k = 0; k < kRun; k++) {
            uint64_t i = tile + (uint64_t)tid * kRun + k;
            if (i >= end) break;
            uint32_t j = s[tid * (kRun + 1) + k];
            if (j == kNoPhi) {
                h = 0;
            } else {
                h = extend_match(text, n, i, (uint64_t)j, h);
            }
            s[tid * (kRun + 1) + k] = (uint32_t)h;
            if (h) h--;
        }
        __syncthreads();
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
        for (int u = 0; u < U; u++) v[u] = (r0 + u * stride < n) ? plcp[v[u]] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (r0 + u * stride < n) lcp[r0 + u * stride] = v[u];
    }
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

// phi / PLCP (4n); for large texts also the pair buffers and radix scratch of the
// partitioned Phi scatter
uint64_t lcp_workspace_bytes(uint64_t n)
{
    ArenaSizer a;
    a.take<uint32_t>(n);
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
    if (ar.overflow) return SFX_ERR_WORKSPACE;
    unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    if (n >= partitioned_scatter_min()) {
        uint64_t* pairs = ar.take<uint64_t>(n);
        uint64_t* tmp = ar.take<uint64_t>(n);
        uint32_t* scratch = ar.take<uint32_t>(radix_scratch_words(n));
        if (ar.overflow) return SFX_ERR_WORKSPACE;
        SFX_LAUNCH("phi_pairs", (double)n * 12, k_phi_pairs, grid, kBlock, st, d_sa, n, pairs);
        SFX_TRY(scatter_pairs_u32(pairs, tmp, n, n, phi, scratch, st, nullptr));
    } else {
        SFX_LAUNCH("phi_scatter", (double)n * 8, k_phi_scatter, grid, kBlock, st, d_sa, n, phi);
    }
    Chunking ch = make_chunking(n, kPlcpTile);
    SFX_LAUNCH("plcp", (double)n * 10, k_plcp, ch.blocks, kBlock, st, d_text, n, phi,
               ch.tiles_per_block);
    SFX_LAUNCH("lcp_gather", (double)n * 12, k_lcp_gather, grid, kBlock, st, d_sa, phi, n, d_lcp);
    return SFX_OK;
}

}  // namespace sfx
