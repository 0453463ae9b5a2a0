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

// per boundary p in [0, n): lb, rb, node id (0 for the root: p = 0 and every boundary of depth 0).
// The three nearest-smaller searches of a boundary nearly always end inside its neighbourhood: a workgroup stages
// kIvTile values in LDS under a binary min-tree (tr[1] = the tile's minimum, tr[kIvTile + i] = value i) and answers
// them there -- up the tree until a sibling holds a smaller value, down to the nearest one: <= 2 log2(kIvTile) LDS reads,
// the same for every lane (a linear scan was measured first: one lane with a distant answer holds up its wave, 170 ms
// per 10^9 boundaries against round 2's 126-133 through the global pyramid alone).  A search that leaves the tile
// continues in the global min-pyramid from the tile's edge.
constexpr int kIvTile = 2048;
// nearest position < i (tile coordinates) with a value < v, or -1
__device__ __forceinline__ int iv_prev_smaller(const uint32_t* tr, int i, uint32_t v)
{
    unsigned k = (unsigned)(kIvTile + i);
    for (;;) {
        if (k <= 1u) return -1;
        if ((k & 1u) && tr[k - 1u] < v) { k--; break; }
        k >>= 1;
    }
    while (k < (unsigned)kIvTile) {
        k = 2u * k + 1u;
        if (tr[k] >= v) k--;
    }
    return (int)k - kIvTile;
}
// nearest position > i with a value < v (or <= v with or_equal), or kIvTile
__device__ __forceinline__ int iv_next_smaller(const uint32_t* tr, int i, uint32_t v, bool or_equal)
{
    auto hit = [&](uint32_t x) { return or_equal ? x <= v : x < v; };
    unsigned k = (unsigned)(kIvTile + i);
    for (;;) {
        if (k <= 1u) return kIvTile;
        if (!(k & 1u) && hit(tr[k + 1u])) { k++; break; }
        k >>= 1;
    }
    while (k < (unsigned)kIvTile) {
        k = 2u * k;
        if (!hit(tr[k])) k++;
    }
    return (int)k - kIvTile;
}
// A search that leaves the tile goes on in the global pyramid -- dozens of dependent loads, for which the other 63
// lanes of the wave would wait.  Such boundaries (a fraction of a per cent: the shallow ones, whose intervals are huge)
// are therefore LISTED (per tile in LDS, one device-wide reservation per tile) and finished by a second, dense launch,
// one listed boundary per lane.  bit 32 / 33 of a list entry: the left / right search is open.
constexpr uint64_t kIvLeftOpen = 1ull << 32, kIvRightOpen = 1ull << 33;
__device__ __forceinline__ void iv_finish(const Pyramid& py, uint64_t n, uint64_t e, uint32_t* __restrict__ lb, uint32_t* __restrict__ rb,
                                          uint32_t* __restrict__ node)
{
    const uint64_t p = e & 0xFFFFFFFFull;
    const uint64_t base = p / kIvTile * kIvTile;
    const uint32_t v = py.lvl[0][p];
    if (e & kIvLeftOpen) {
        const uint64_t l = prev_smaller(py, base, v);                    // (everything in [base, p) is >= v)
        lb[p] = (uint32_t)l;
        node[p] = (uint32_t)next_smaller(py, l, v, true, n);             // leftmost boundary of the interval with its depth (<= p)
    }
    if (e & kIvRightOpen) rb[p] = (uint32_t)(next_smaller(py, base + kIvTile - 1, v, false, n) - 1);
}
__global__ void __launch_bounds__(kBlock)
k_lcp_intervals(Pyramid py, uint64_t n, uint64_t tiles_per_block, uint32_t* __restrict__ lb, uint32_t* __restrict__ rb,
                uint32_t* __restrict__ node, uint64_t* __restrict__ open_list, uint64_t open_cap,
                unsigned long long* __restrict__ open_count)
{
    __shared__ uint32_t tr[2 * kIvTile];
    __shared__ uint64_t esc[kIvTile];
    __shared__ uint32_t n_esc;
    __shared__ unsigned long long esc_base;
    const uint32_t* lcp = py.lvl[0];
    const uint64_t tile0 = (uint64_t)blockIdx.x * tiles_per_block;
    for (uint64_t tile = tile0; tile < tile0 + tiles_per_block; tile++) {
        const uint64_t base = tile * kIvTile;
        if (base >= n) break;
        if (threadIdx.x == 0) n_esc = 0;
        for (unsigned i = threadIdx.x; i < (unsigned)kIvTile; i += kBlock) {
            const uint64_t g = base + i;
            tr[kIvTile + i] = (g == 0 || g >= n) ? 0u : lcp[g];          // (boundary 0 and the end of the array: depth 0)
        }
        __syncthreads();
        for (unsigned w = kIvTile / 2; w >= 1; w >>= 1) {                // level by level: w nodes, ids [w, 2w)
            for (unsigned k = w + threadIdx.x; k < 2 * w; k += kBlock) tr[k] = dmin(tr[2 * k], tr[2 * k + 1]);
            __syncthreads();
        }
        for (unsigned i = threadIdx.x; i < (unsigned)kIvTile; i += kBlock) {
            const uint64_t p = base + i;
            if (p >= n) break;
            const uint32_t v = tr[kIvTile + i];
            if (p == 0 || v == 0) {                           // the root: every boundary of depth 0
                lb[p] = 0;
                rb[p] = (uint32_t)(n - 1);
                node[p] = 0;
                continue;
            }
            uint64_t open = 0;
            const int jl = iv_prev_smaller(tr, (int)i, v);
            if (jl >= 0) {
                lb[p] = (uint32_t)(base + (uint64_t)jl);
                // leftmost boundary of the interval with its depth: the first value <= v after l (everything between is >= v)
                node[p] = (uint32_t)(base + (uint64_t)iv_next_smaller(tr, jl, v, true));     // (<= i: position i qualifies)
            } else {
                open |= kIvLeftOpen;
            }
            const int jr = iv_next_smaller(tr, (int)i, v, false);
            if (jr < kIvTile) rb[p] = (uint32_t)(dmin<uint64_t>(base + (uint64_t)jr, n) - 1);
            else if (base + kIvTile >= n) rb[p] = (uint32_t)(n - 1);
            else open |= kIvRightOpen;
            if (open) esc[atomicAdd(&n_esc, 1u)] = open | p;
        }
        __syncthreads();
        const uint32_t cnt = n_esc;
        if (cnt) {
            if (threadIdx.x == 0) esc_base = atomicAdd(open_count, (unsigned long long)cnt);
            __syncthreads();
            const unsigned long long at = esc_base;
            // (what does not fit the list -- a monotone LCP array -- is finished here, slowly; every slot below the
            // capacity that was reserved is written, so the second launch reads no gap)
            for (unsigned k = threadIdx.x; k < cnt; k += kBlock) {
                if (at + k < open_cap) open_list[at + k] = esc[k];
                else iv_finish(py, n, esc[k], lb, rb, node);
            }
        }
        __syncthreads();
    }
}
// the listed boundaries, one per lane
__global__ void __launch_bounds__(kBlock)
k_lcp_intervals_open(Pyramid py, uint64_t n, const uint64_t* __restrict__ open_list, uint64_t open_cap,
                     const unsigned long long* __restrict__ open_count, uint32_t* __restrict__ lb, uint32_t* __restrict__ rb,
                     uint32_t* __restrict__ node)
{
    // (entries reserved beyond the capacity were finished by their tiles; reservations never leave gaps below it)
    unsigned long long cnt = *open_count;
    if (cnt > open_cap) cnt = open_cap;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < cnt; k += stride) iv_finish(py, n, open_list[k], lb, rb, node);
}
// parent of the node a boundary belongs to, and the parent of every leaf
__global__ void __launch_bounds__(kBlock)
k_tree_parents(const uint32_t* __restrict__ lcp, uint64_t n, const uint32_t* __restrict__ lb, const uint32_t* __restrict__ rb,
               const uint32_t* __restrict__ node, uint32_t* __restrict__ parent, uint32_t* __restrict__ leaf_parent)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
        // leaf of rank p: under the deeper of boundaries p and p + 1.  Boundary 0 has depth 0 whatever d_lcp[0] holds
        // ("not looked at", include/suffix_hip.h: k_pyr_reduce, prev_smaller and k_lcp_intervals ignore it too)
        const uint32_t dl = p ? lcp[p] : 0u, dr = p + 1 < n ? lcp[p + 1] : 0u;
        leaf_parent[p] = dl >= dr ? node[p] : node[p + 1];
        if (node[p] == 0) { parent[p] = kNoNode; continue; }                     // (the root has no parent: every boundary of depth 0)
        const uint64_t l = lb[p], r = (uint64_t)rb[p] + 1;                        // the two delimiting boundaries
        const uint32_t vl = l ? lcp[l] : 0u, vr = r < n ? lcp[r] : 0u;
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
// (+ the list of boundaries whose searches leave their tile: n / 16 entries, and its counter)
static uint64_t open_list_cap(uint64_t n) { return n / 16 + 4096; }
uint64_t lcp_intervals_workspace_bytes(uint64_t n)
{
    return ((pyramid_words(n) * sizeof(uint32_t) + 255) & ~uint64_t(255)) + 256 + open_list_cap(n) * sizeof(uint64_t) + 256;
}

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
    {
        char* tail = reinterpret_cast<char*>(ws) + ((pyramid_words(n) * sizeof(uint32_t) + 255) & ~uint64_t(255));
        unsigned long long* open_count = reinterpret_cast<unsigned long long*>(tail);
        uint64_t* open_list = reinterpret_cast<uint64_t*>(tail + 256);
        const uint64_t cap = open_list_cap(n);
        SFX_HIP(hipMemsetAsync(open_count, 0, sizeof(unsigned long long), st));
        Chunking ch = make_chunking(n, kIvTile, 4 * kMaxGrid);
        SFX_LAUNCH("tree_intervals", (double)n * 16, k_lcp_intervals, ch.blocks, kBlock, st, py, n, ch.tiles_per_block, d_lb, d_rb, d_node,
                   open_list, cap, open_count);
        SFX_LAUNCH("tree_intervals_open", 0.0, k_lcp_intervals_open, grid, kBlock, st, py, n, (const uint64_t*)open_list, cap,
                   (const unsigned long long*)open_count, d_lb, d_rb, d_node);
    }
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
