// sfx_tree.hip -- suffix-tree topology as flat arrays, and the generalized-suffix-array lookup
// (/root/reference/suffix_tree/src/lib.rs:392-505 `to_suffix_tree`,
// /root/reference/README.md:60-74).
//
// The reference builds its suffix tree by one left-to-right sweep over (SA, LCP) with a stack of
// ancestors: every internal node is an *lcp-interval* -- a maximal range of ranks [lb, rb] whose
// suffixes share `depth` symbols, depth = min lcp[lb+1..rb] > max(lcp[lb], lcp[rb+1]).  The sweep is
// serial; the same tree falls out of two all-nearest-smaller-value problems on the LCP array:
//   for a boundary p (between ranks p-1 and p, value lcp[p]):
//     lb[p] = the nearest q < p with lcp[q] < lcp[p]          (q = 0 at the latest: lcp[0] = 0 = the root)
//     rb[p] = (the nearest q > p with lcp[q] < lcp[p]) - 1    (n - 1 at the latest)
//   -> [lb, rb] at depth lcp[p] is the node that boundary p belongs to; its id is its LEFTMOST boundary
//      with that value, node[p] = the first q > lb[p] with lcp[q] <= lcp[p];
//   parent = the node of whichever of the two delimiting boundaries lb[p] / rb[p] + 1 is deeper;
//   a leaf (rank r) hangs under the node of the deeper of its two boundaries r and r + 1.
// Boundaries with lcp 0 belong to the root (id 0, [0, n-1], depth 0).  Searches run over a pyramid of
// block minima (64-ary), so a nearest-smaller query costs O(64 log_64 n) reads in the worst case and a
// handful for the nearby answers that dominate.
#include "sfx_host.hpp"

namespace sfx {

constexpr int kPyrFan = 64;
constexpr int kPyrMaxLevels = 6;                 // 64^6 > 2^32
constexpr uint32_t kNoNode = 0xFFFFFFFFu;

struct Pyramid {
    const uint32_t* lvl[kPyrMaxLevels];          // lvl[0] = the LCP array itself
    uint64_t len[kPyrMaxLevels];
    int levels;
};

__global__ void __launch_bounds__(kBlock)
k_pyr_reduce(const uint32_t* __restrict__ in, uint64_t n_in, uint32_t* __restrict__ out, uint64_t n_out, int first_is_zero)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n_out; j += stride) {
        uint32_t m = 0xFFFFFFFFu;
        const uint64_t b = j * kPyrFan, e = dmin<uint64_t>(b + kPyrFan, n_in);
        for (uint64_t i = b; i < e; i++) m = dmin(m, in[i]);
        out[j] = (j == 0 && first_is_zero) ? 0u : m;             // (boundary 0 has depth 0 whatever lcp[0] holds)
    }
}

// nearest q < p with lcp[q] < v (v > 0, so q = 0 qualifies at the latest)
__device__ __forceinline__ uint64_t prev_smaller(const Pyramid& py, uint64_t p, uint32_t v)
{
    int l = 0;
    int64_t idx = (int64_t)p - 1;
    for (;;) {                                               // climb: scan to the left inside the current block
        bool found = false;
        for (;;) {
            if ((l == 0 && idx == 0) || py.lvl[l][idx] < v) { found = true; break; }     // (boundary 0: depth 0 by definition)
            if (idx % kPyrFan == 0) break;
            idx--;
        }
        if (found) break;
        idx = idx / kPyrFan - 1;                             // the block to the left, one level up
        l++;
    }
    while (l > 0) {                                          // descend: the rightmost child below v
        l--;
        int64_t c = dmin<int64_t>(idx * kPyrFan + kPyrFan - 1, (int64_t)py.len[l] - 1);
        while (!(l == 0 && c == 0) && py.lvl[l][c] >= v) c--;
        idx = c;
    }
    return (uint64_t)idx;
}
// nearest q > p with lcp[q] < v, or n if there is none; with `or_equal`, lcp[q] <= v instead
__device__ __forceinline__ uint64_t next_smaller(const Pyramid& py, uint64_t p, uint32_t v, bool or_equal, uint64_t n)
{
    auto hit = [&](uint32_t x) { return or_equal ? x <= v : x < v; };
    int l = 0;
    uint64_t idx = p + 1;
    for (;;) {
        if (idx >= py.len[l]) return n;
        bool found = false;
        for (;;) {
            if (hit(py.lvl[l][idx])) { found = true; break; }
            if (idx % kPyrFan == kPyrFan - 1 || idx + 1 >= py.len[l]) break;
            idx++;
        }
        if (found) break;
        idx = idx / kPyrFan + 1;
        l++;
        if (l >= py.levels) return n;
    }
    while (l > 0) {
        l--;
        uint64_t c = idx * kPyrFan;
        while (!hit(py.lvl[l][c])) c++;
        idx = c;
    }
    return idx;
}

// per boundary p in [0, n): lb, rb, node id (kNoNode for p = 0 and for boundaries of the root other than its id 0)
__global__ void __launch_bounds__(kBlock)
k_lcp_intervals(Pyramid py, uint64_t n, uint32_t* __restrict__ lb, uint32_t* __restrict__ rb, uint32_t* __restrict__ node)
{
    const uint32_t* lcp = py.lvl[0];
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
        const uint32_t v = lcp[p];
        if (p == 0 || v == 0) {                              // the root: every boundary of depth 0
            lb[p] = 0;
            rb[p] = (uint32_t)(n - 1);
            node[p] = 0;
            continue;
        }
        const uint64_t l = prev_smaller(py, p, v);
        const uint64_t r = next_smaller(py, p, v, false, n);
        lb[p] = (uint32_t)l;
        rb[p] = (uint32_t)(r - 1);
        node[p] = (uint32_t)next_smaller(py, l, v, true, n);  // leftmost boundary of the interval with its depth (<= p)
    }
}
// parent of the node a boundary belongs to, and the parent of every leaf
__global__ void __launch_bounds__(kBlock)
k_tree_parents(const uint32_t* __restrict__ lcp, uint64_t n, const uint32_t* __restrict__ lb, const uint32_t* __restrict__ rb,
               const uint32_t* __restrict__ node, uint32_t* __restrict__ parent, uint32_t* __restrict__ leaf_parent)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
        // leaf of rank p: under the deeper of boundaries p and p + 1
        const uint32_t dl = lcp[p], dr = p + 1 < n ? lcp[p + 1] : 0u;
        leaf_parent[p] = dl >= dr ? node[p] : node[p + 1];
        if (node[p] == 0) { parent[p] = kNoNode; continue; }                     // (the root has no parent: every boundary of depth 0)
        const uint64_t l = lb[p], r = (uint64_t)rb[p] + 1;                        // the two delimiting boundaries
        const uint32_t vl = lcp[l], vr = r < n ? lcp[r] : 0u;
        parent[p] = vl >= vr ? node[l] : node[r];
    }
}

// generalized suffix array (README.md:60-74): documents concatenated with a separator; the document of
// a text position = the last start <= position (binary search over the sorted starts)
__global__ void __launch_bounds__(kBlock)
k_doc_lookup(const uint32_t* __restrict__ pos, uint64_t count, const uint64_t* __restrict__ starts, uint64_t ndocs,
             uint32_t* __restrict__ doc, uint32_t* __restrict__ offset)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        const uint64_t p = pos[i];
        uint64_t lo = 0, hi = ndocs;                         // first start > p
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (starts[mid] <= p) lo = mid + 1; else hi = mid;
        }
        const uint64_t d = lo ? lo - 1 : 0;
        if (doc) doc[i] = (uint32_t)d;
        if (offset) offset[i] = (uint32_t)(p - starts[d]);
    }
}

static uint64_t pyramid_words(uint64_t n)
{
    uint64_t words = 0, len = n;
    for (int l = 1; l < kPyrMaxLevels && len > 1; l++) {
        len = (len + kPyrFan - 1) / kPyrFan;
        words += (len + 63) & ~uint64_t(63);
    }
    return words;
}
uint64_t lcp_intervals_workspace_bytes(uint64_t n) { return pyramid_words(n) * sizeof(uint32_t) + 256; }

int lcp_intervals_dev(const uint32_t* d_lcp, uint64_t n, uint32_t* d_lb, uint32_t* d_rb, uint32_t* d_node, uint32_t* d_parent,
                      uint32_t* d_leaf_parent, void* ws, uint64_t ws_bytes, hipStream_t st)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!d_lcp || !d_lb || !d_rb || !d_node || !d_parent || !d_leaf_parent) return SFX_ERR_ARG;
    if (!ws || ws_bytes < lcp_intervals_workspace_bytes(n)) return SFX_ERR_WORKSPACE;
    Pyramid py;
    py.lvl[0] = d_lcp;
    py.len[0] = n;
    py.levels = 1;
    uint32_t* w = reinterpret_cast<uint32_t*>(ws);
    uint64_t len = n;
    while (py.levels < kPyrMaxLevels && len > 1) {
        const uint64_t out_len = (len + kPyrFan - 1) / kPyrFan;
        const unsigned grid = (unsigned)dmin<uint64_t>((out_len + kBlock - 1) / kBlock, kMaxGrid);
        SFX_LAUNCH("tree_pyramid", (double)len * 4, k_pyr_reduce, grid, kBlock, st, py.lvl[py.levels - 1], len, w, out_len, 1);
        py.lvl[py.levels] = w;
        py.len[py.levels] = out_len;
        py.levels++;
        w += (out_len + 63) & ~uint64_t(63);
        len = out_len;
    }
    for (int l = py.levels; l < kPyrMaxLevels; l++) { py.lvl[l] = nullptr; py.len[l] = 0; }
    const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("tree_intervals", (double)n * 16, k_lcp_intervals, grid, kBlock, st, py, n, d_lb, d_rb, d_node);
    SFX_LAUNCH("tree_parents", (double)n * 28, k_tree_parents, grid, kBlock, st, d_lcp, n, (const uint32_t*)d_lb,
               (const uint32_t*)d_rb, (const uint32_t*)d_node, d_parent, d_leaf_parent);
    return SFX_OK;
}

int doc_lookup_dev(const uint32_t* d_pos, uint64_t count, const uint64_t* d_starts, uint64_t ndocs, uint32_t* d_doc,
                   uint32_t* d_offset, hipStream_t st)
{
    if (count == 0) return SFX_OK;
    if (!d_pos || !d_starts || ndocs == 0 || (!d_doc && !d_offset)) return SFX_ERR_ARG;
    const unsigned grid = (unsigned)dmin<uint64_t>((count + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("doc_lookup", (double)count * 12, k_doc_lookup, grid, kBlock, st, d_pos, count, d_starts, ndocs, d_doc, d_offset);
    return SFX_OK;
}

}  // namespace sfx
