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
#include <stdlib.h>

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

// ---- bucket directory of the resident index ---------------------------------
// dir[c] = first rank whose suffix's first `dbits` bits of dense symbol codes (`bits` bits per symbol,
// zero-padded past the end of the text) are >= c, for every dbits-bit prefix c, plus dir[2^dbits] = n:
// the bucket structure the build's initial sort works with, kept next to the suffix array -- about one
// bucket per four suffixes (dbits = log2 n - 2, at most 28: 1 GiB of directory for 10^9 suffixes; HBM is
// what this machine has plenty of).  A query looks its own first dbits bits up -- two adjacent reads
// instead of the ~ dbits top levels of the binary search, and no probe at all for queries of <= kfull
// symbols or with a byte the text does not contain.  k = symbols a key touches (the last one partly).
struct DirParams {
    const uint32_t* dir;
    const uint16_t* lut;        // byte -> symbol code + 1, 0 = byte does not occur in the text
    int bits, k, dbits;
};
__device__ __forceinline__ uint32_t dir_code_of_suffix(const uint8_t* __restrict__ text, uint64_t n, uint64_t s,
                                                       const uint16_t* __restrict__ lut, int bits, int k, int dbits)
{
    uint64_t c = 0;
    for (int j = 0; j < k; j++) c = (c << bits) | (s + j < n ? (uint64_t)lut[text[s + j]] - 1u : 0u);
    return (uint32_t)(c >> (k * bits - dbits));
}
// dir[code of rank r] = r wherever the code changes (dir pre-filled with 0xFFFFFFFF); bad[0] counts
// suffix-array entries >= n (from_parts hands the engine an unchecked table, :105-119)
__global__ void __launch_bounds__(kBlock)
k_dir_mark(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, const uint16_t* __restrict__ lut,
           int bits, int k, int dbits, uint32_t* __restrict__ dir, unsigned long long* __restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const unsigned lane = lane_id();
    for (uint64_t r0 = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); r0 < n; r0 += stride) {
        const uint64_t r = r0 + lane;
        const bool live = r < n;
        uint64_t s = live ? (uint64_t)sa[r] : 0;
        const bool ok = s < n;
        if (live && !ok) atomicAdd(bad, 1ull);
        if (!ok) s = 0;
        const uint32_t c = live ? dir_code_of_suffix(text, n, s, lut, bits, k, dbits) : 0u;
        uint32_t cp = __shfl_up(c, 1u);
        if (lane == 0 && live && r > 0) {
            const uint64_t sp = sa[r - 1];
            cp = sp < n ? dir_code_of_suffix(text, n, sp, lut, bits, k, dbits) : 0u;
        }
        if (live && (r == 0 || c != cp)) dir[c] = (uint32_t)r;
    }
}
// empty buckets take the start of the next non-empty one: suffix-min over the directory, three steps
constexpr int kDirRun = 16;
constexpr int kDirTile = kBlock * kDirRun;
__global__ void __launch_bounds__(kBlock)
k_dir_block_min(const uint32_t* __restrict__ dir, uint64_t entries, uint32_t* __restrict__ bmin)
{
    __shared__ uint32_t red[kWavesPerBlock];
    const uint64_t base = (uint64_t)blockIdx.x * kDirTile;
    uint32_t v = 0xFFFFFFFFu;
    for (int j = 0; j < kDirRun; j++) {
        const uint64_t i = base + (uint64_t)j * kBlock + threadIdx.x;
        if (i < entries) v = dmin(v, dir[i]);
    }
    for (int d = 32; d >= 1; d >>= 1) v = dmin(v, __shfl_xor(v, d));
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = red[0];
        for (int w = 1; w < kWavesPerBlock; w++) a = dmin(a, red[w]);
        bmin[blockIdx.x] = a;
    }
}
__global__ void k_dir_scan_mins(uint32_t* __restrict__ bmin, uint64_t nb)      // one thread: nb <= 65537
{
    if (threadIdx.x || blockIdx.x) return;
    uint32_t run = 0xFFFFFFFFu;
    for (uint64_t b = nb; b-- > 0;) {
        const uint32_t mine = bmin[b];
        bmin[b] = run;                                               // min over the blocks to the right
        run = dmin(run, mine);
    }
}
__global__ void __launch_bounds__(kBlock)
k_dir_fill(uint32_t* __restrict__ dir, uint64_t entries, const uint32_t* __restrict__ bmin)
{
    __shared__ uint32_t tmin[kBlock];
    const uint64_t i0 = (uint64_t)blockIdx.x * kDirTile + (uint64_t)threadIdx.x * kDirRun;
    uint32_t v[kDirRun], run = 0xFFFFFFFFu;
    for (int j = kDirRun - 1; j >= 0; j--) {
        v[j] = i0 + j < entries ? dir[i0 + j] : 0xFFFFFFFFu;
        run = dmin(run, v[j]);
    }
    tmin[threadIdx.x] = run;
    __syncthreads();
    uint32_t right = bmin[blockIdx.x];
    for (unsigned t = threadIdx.x + 1; t < (unsigned)kBlock; t++) right = dmin(right, tmin[t]);   // (256 LDS reads: one-time index build)
    for (int j = kDirRun - 1; j >= 0; j--) {
        right = dmin(right, v[j]);
        if (i0 + j < entries) dir[i0 + j] = right;
    }
}

__global__ void __launch_bounds__(kBlock)
k_query_batch_dir(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, DirParams dp,
                  const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, uint64_t nq,
                  uint32_t* __restrict__ start_out, uint32_t* __restrict__ end_out,
                  uint8_t* __restrict__ found_out, uint32_t* __restrict__ any_out)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t qi = (uint64_t)blockIdx.x * kBlock + threadIdx.x; qi < nq; qi += stride) {
        const uint8_t* q = qbytes + qoff[qi];
        const uint64_t m = qoff[qi + 1] - qoff[qi];
        uint64_t start = 0, end = 0;
        if (n != 0 && m != 0) {                                       // :228-229
            const int kfull = dp.dbits / dp.bits;                     // symbols wholly inside a directory key
            const int L = m < (uint64_t)dp.k ? (int)m : dp.k;
            uint64_t c = 0;
            bool absent = false;
            for (int j = 0; j < L; j++) {
                const uint32_t sym = dp.lut[q[j]];
                absent |= sym == 0u;
                c = (c << dp.bits) | (uint64_t)(sym - 1u);
            }
            if (!absent) {
                // the first min(L * bits, dbits) bits of the query's code stream select the bucket range
                const int have = L * dp.bits;
                uint64_t c_lo, c_hi;
                if (have >= dp.dbits) { c_lo = c >> (have - dp.dbits); c_hi = c_lo + 1; }
                else { c_lo = c << (dp.dbits - have); c_hi = (c + 1) << (dp.dbits - have); }
                uint64_t lo = dp.dir[c_lo], hi = dp.dir[c_hi];
                if (m <= (uint64_t)kfull) {
                    // every suffix in these buckets starts with q, except suffixes shorter than q whose
                    // zero padding imitates q's tail: they are the first entries of the first bucket
                    while (lo < hi && n - (uint64_t)sa[lo] < m) lo++;
                    start = lo;
                    end = hi;
                } else {
                    const uint64_t top = hi;
                    while (lo < hi) {                                 // :244-246 inside the bucket
                        const uint64_t mid = (lo + hi) >> 1;
                        if (query_le_suffix(q, m, text, n, sa[mid])) hi = mid; else lo = mid + 1;
                    }
                    start = lo;
                    uint64_t cnt = 0;                                 // :247-250 as in k_query_batch, bounded by the bucket
                    if (start < top && suffix_starts_with(q, m, text, n, sa[start])) {
                        // gallop while the interval may still be short (<= 15 matches: 4 probes), then bisect what is
                        // left of the range -- queries drawn from a natural-language text match 10^5 suffixes on
                        // average, where galloping all the way costs 2 log2(#matches) probes and bisecting log2(range)
                        cnt = 1;
                        uint64_t step = 1;
                        bool more = true;
                        while (step <= 8) {
                            more = start + cnt - 1 + step < top && suffix_starts_with(q, m, text, n, sa[start + cnt - 1 + step]);
                            if (!more) break;
                            cnt += step;
                            step <<= 1;
                        }
                        lo = start + cnt;
                        hi = more ? top : dmin<uint64_t>(top, start + cnt - 1 + step);
                        while (lo < hi) {
                            const uint64_t mid = (lo + hi) >> 1;
                            if (!suffix_starts_with(q, m, text, n, sa[mid])) hi = mid; else lo = mid + 1;
                        }
                        cnt = lo - start;
                    }
                    end = start + cnt;
                }
            }
        }
        const bool found = end > start;
        if (!found) start = end = 0;
        if (start_out) start_out[qi] = (uint32_t)start;
        if (end_out) end_out[qi] = (uint32_t)end;
        if (found_out) found_out[qi] = found ? 1 : 0;
        if (any_out) any_out[qi] = found ? sa[start] : 0xFFFFFFFFu;
    }
}

// directory key width: about n / 4 buckets, at most 2^28 (1 GiB of u32); k = symbols a key touches
int dir_shape(uint64_t n, int bits, int* k_out, int* dbits_out, uint64_t* entries_out)
{
    int dbits = bits_for(n) - 2;
    if (dbits > 28) dbits = 28;
    if (dbits < bits) dbits = bits;
    *dbits_out = dbits;
    *k_out = (dbits + bits - 1) / bits;
    *entries_out = (1ull << dbits) + 1;
    return SFX_OK;
}
uint64_t dir_scratch_words(uint64_t entries) { return (entries + kDirTile - 1) / kDirTile + 8; }

// d_lut256 (256 u16: byte -> code + 1) and d_dir (entries u32) are filled; d_scratch: dir_scratch_words(entries) u32
// + 8 bytes.  *bad_out = suffix-array entries >= n (the table is not usable then).
int dir_build_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, const uint16_t* host_lut256, int bits, int k,
                  int dbits, uint64_t entries, uint16_t* d_lut256, uint32_t* d_dir, uint32_t* d_scratch, hipStream_t st,
                  uint64_t* bad_out)
{
    unsigned long long* bad = reinterpret_cast<unsigned long long*>(d_scratch);
    uint32_t* bmin = d_scratch + 2;
    SFX_HIP(hipMemcpyAsync(d_lut256, host_lut256, 256 * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    SFX_HIP(hipMemsetAsync(d_dir, 0xFF, entries * sizeof(uint32_t), st));
    SFX_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned long long), st));
    const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("dir_mark", (double)n * 12, k_dir_mark, grid, kBlock, st, d_text, n, d_sa, d_lut256, bits, k, dbits, d_dir, bad);
    const uint32_t n32 = (uint32_t)n;
    SFX_HIP(hipMemcpyAsync(d_dir + (entries - 1), &n32, sizeof(n32), hipMemcpyHostToDevice, st));
    const uint64_t nb = (entries + kDirTile - 1) / kDirTile;
    SFX_LAUNCH("dir_block_min", (double)entries * 4, k_dir_block_min, (unsigned)nb, kBlock, st, d_dir, entries, bmin);
    SFX_LAUNCH("dir_scan_mins", 0.0, k_dir_scan_mins, 1, 64, st, bmin, nb);
    SFX_LAUNCH("dir_fill", (double)entries * 8, k_dir_fill, (unsigned)nb, kBlock, st, d_dir, entries, bmin);
    unsigned long long host_bad = 0;
    SFX_TRY(read_back(&host_bad, bad, sizeof(host_bad), st));
    *bad_out = host_bad;
    return SFX_OK;
}

int query_batch_dir_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, const uint32_t* d_dir,
                        const uint16_t* d_lut256, int bits, int k, int dbits, const uint8_t* d_q, const uint64_t* d_qoff, uint64_t nq,
                        uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, hipStream_t st)
{
    if (nq == 0) return SFX_OK;
    if (!d_qoff || (n && (!d_text || !d_sa || !d_dir || !d_lut256))) return SFX_ERR_ARG;
    const unsigned grid = (unsigned)dmin<uint64_t>((nq + kBlock - 1) / kBlock, kMaxGrid);
    // one directory read, then 2 * log2(bucket) probes of 12 bytes: ~ half of the undirected search
    const double probes = 1.0 + (double)dmax(2, 2 * (bits_for(n ? n : 1) - dbits));
    DirParams dp = {d_dir, d_lut256, bits, k, dbits};
    SFX_LAUNCH("query_batch_dir", (double)nq * probes * 12.0, k_query_batch_dir, grid, kBlock, st, d_text, n, d_sa, dp,
               d_q, d_qoff, nq, d_start, d_end, d_found, d_any);
    return SFX_OK;
}

// ---- prefix-key B+tree of the resident index ------------------------------------------------------
// What a query pays for in a binary search over the suffix array is not the ~30 probes but the last ~12 of
// them: the top levels' SA entries and text lines are shared by all queries and sit in L2, the bottom
// levels are two random 128-byte lines per probe (SA entry, then text).  The index therefore keeps, for
// every rank, the first 16 bytes of its suffix as two big-endian integers (zero-padded past the end of the
// text) -- the leaves of a static 16-ary B+tree whose inner levels hold the last key of every block of 16.
// One node = adjacent 128-byte lines: the 16 first words, then the 16 second words; a search reads the first
// line of one node per level (8 levels at n = 10^9, the top five cached) and the second line only where the
// first words tie with the query's (inside a run of suffixes that share 8 bytes: the bottom levels of a query
// longer than that).  The text is touched only when the query is longer than the key AND shares all of it
// with several suffixes.  Order: for zero-padded keys A, B of byte strings a, b, A < B implies a < b (a proper
// prefix sorts first, padding is the smallest byte), so the ranks whose key lies in [q padded with 0x00, q
// padded with 0xFF] contain every suffix that starts with q, and for |q| <= 16 nothing else except suffixes
// shorter than q, which come first in that range.  16 n bytes of HBM + 7 % for the inner levels (17 GB at
// n = 10^9: HBM is what this machine has plenty of).
constexpr int kTreeFan = 16;                           // (the shifts by 4 below are log2 of it)
// a key = the first 16 bytes of a suffix, as two big-endian words.  (Three words -- 24 bytes, 26 GB at n = 10^9 --
// are one constant away and were measured on config 5's query set: 0.42 + 0.18 ms against 0.35 + 0.27 with two,
// 3 % in all for 8 n more bytes and 13 ms more index build: not taken.)
constexpr int kTreeKeyWords = 2;
constexpr int kTreeNodeWords = kTreeKeyWords * kTreeFan; // [16 first words][16 second words]
constexpr int kTreeMaxLevels = 9;                        // 16^8 = 2^32
struct KeyTree {
    const uint64_t* lvl[kTreeMaxLevels];                 // lvl[0] = leaves, each level padded with ~0 to whole nodes
    uint64_t len[kTreeMaxLevels];                        // real entries per level
    int levels;
    uint64_t n;
};
__device__ __forceinline__ uint64_t be64_of_suffix(const uint8_t* __restrict__ text, uint64_t n, uint64_t s)
{
    uint64_t k = 0;
    if (s + 8 <= n) {
        __builtin_memcpy(&k, text + s, 8);
        return __builtin_bswap64(k);
    }
    for (unsigned j = 0; j < 8; j++) k = (k << 8) | (s + j < n ? (uint64_t)text[s + j] : 0ull);
    return k;
}
// entry e of a level lives in node e / 16, slot e % 16
__device__ __forceinline__ uint64_t tree_slot(uint64_t e) { return (e / kTreeFan) * kTreeNodeWords + (e % kTreeFan); }
__global__ void __launch_bounds__(kBlock)
k_tree_leaves(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, uint64_t* __restrict__ leaves)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) {
        const uint64_t sfx = sa[r];
        const uint64_t w = tree_slot(r);
#pragma unroll
        for (int k = 0; k < kTreeKeyWords; k++)                          // (an invalid table is refused elsewhere)
            leaves[w + k * kTreeFan] = sfx < n ? be64_of_suffix(text, n, sfx + 8u * k) : ~0ull;
    }
}
__global__ void __launch_bounds__(kBlock)
k_tree_level(const uint64_t* __restrict__ below, uint64_t len_below, uint64_t* __restrict__ out, uint64_t len_out)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < len_out; j += stride) {
        const uint64_t src = tree_slot(dmin<uint64_t>(j * kTreeFan + kTreeFan - 1, len_below - 1));   // last key of block j
        const uint64_t dst = tree_slot(j);
#pragma unroll
        for (int k = 0; k < kTreeKeyWords; k++) out[dst + k * kTreeFan] = below[src + k * kTreeFan];
    }
}
// ---- the descents of a wave's 64 queries, node lines fetched cooperatively ---------------------------
// A lane that reads its own node needs 8 loads of 16 bytes, and every one of them is 64 different lines to the
// wave: 8 x 64 tag lookups per level in the CU's L1, which is what a batch of short queries was bound by
// (profiles/r2_query_pmc.txt: 153 L1 accesses per query at 60 % of one per cycle, 10 HBM lines per query).
// Here lanes 8 j' .. 8 j' + 7 of load i fetch the line of query 8 i + j' together -- one line per 8 lanes, 64
// lookups per level instead of 512 -- into an LDS row of the wave; then every lane reads its own row back.
// Rows are 144 bytes apart (9 x 16): a wave's 16-byte reads of 64 rows fall on distinct banks.
constexpr int kCoopRow = 9;
struct CoopSmem {
    ulonglong2 rows[kWavesPerBlock][kWave][kCoopRow];
    uint64_t addr[kWavesPerBlock][kWave];
};
// np: this lane's node line (16 u64), or nullptr; kk: its 16 keys (zeros for nullptr).  Every lane of the wave calls.
__device__ __forceinline__ void coop_load_line(CoopSmem& s, const uint64_t* np, ulonglong2 (&kk)[kTreeFan / 2])
{
    const unsigned w = wave_id(), lane = lane_id();
    s.addr[w][lane] = reinterpret_cast<uint64_t>(np);
    wave_sync();
    ulonglong2 v[kTreeFan / 2];
#pragma unroll
    for (int i = 0; i < kTreeFan / 2; i++) {
        const uint64_t a = s.addr[w][8 * i + (lane >> 3)];
        v[i] = a ? reinterpret_cast<const ulonglong2*>(a)[lane & 7u] : ulonglong2{0ull, 0ull};
    }
#pragma unroll
    for (int i = 0; i < kTreeFan / 2; i++) s.rows[w][8 * i + (lane >> 3)][lane & 7u] = v[i];
    wave_sync();
#pragma unroll
    for (int k = 0; k < kTreeFan / 2; k++) kk[k] = s.rows[w][lane][k];
    wave_sync();                                              // (the next call overwrites the rows)
}
// one key word of a node against q, over the slots [base, base + ties) that tie on the words before it:
// how many of them are below q, how many equal
__device__ __forceinline__ void count_word(const ulonglong2 (&kk)[kTreeFan / 2], uint64_t q, unsigned base, unsigned ties,
                                           unsigned& below, unsigned& equal)
{
    below = 0;
    equal = 0;
#pragma unroll
    for (int i = 0; i < kTreeFan / 2; i++) {
        const unsigned s0 = 2u * i - base;                            // (unsigned: slots before the range wrap)
        const unsigned in0 = s0 < ties, in1 = s0 + 1u < ties;
        below += (in0 & (kk[i].x < q)) + (in1 & (kk[i].y < q));
        equal += (in0 & (kk[i].x == q)) + (in1 & (kk[i].y == q));
    }
}
// One side of a descent inside a node: after word w the keys below the search key are the `base` first slots plus
// whatever the tying slots [base, base + ties) decide on the next word.  #keys < q = base at the end; #keys <= q =
// base + ties.  A side is settled when nothing ties, or when the rest of its search key is all 0x00 (lower end:
// no key is below it) / all 0xFF (upper end: every key is <= it).
struct NodeSide {
    unsigned base, ties;
    __device__ __forceinline__ void start() { base = 0; ties = kTreeFan; }
    __device__ __forceinline__ void word(const ulonglong2 (&kk)[kTreeFan / 2], uint64_t q)
    {
        unsigned below, equal;
        count_word(kk, q, base, ties, below, equal);
        base += below;
        ties = equal;
    }
};
// lo = first rank whose key is >= klo, hi = first rank whose key is > khi (klo <= khi, kTreeKeyWords words each), for
// the query of every lane of the wave (`act`: this lane has one).  A lane's two descents start at node `pos` of
// level `top` (the root: levels - 1, 0), whose ranks must include both answers (or end at them), and share every
// node until their paths part -- for a query that matches a handful of suffixes, at the leaf.  Per level the wave
// makes one cooperative fetch per key word for the lower (and shared) descents, one per word for the parted upper
// ones if there are any; words after the first only while some lane still has slots that tie.
__device__ __forceinline__ void tree_bounds_wave(CoopSmem& s, const KeyTree& t, const uint64_t (&klo)[kTreeKeyWords],
                                                 const uint64_t (&khi)[kTreeKeyWords], bool act, int top, uint64_t pos,
                                                 uint64_t& lo, uint64_t& hi)
{
    uint64_t plo = pos, phi = pos;
    bool lo_out = !act, hi_out = !act;                                // beyond the last key of a level: the answer is n
    int lmax = act ? top : -1;
    for (int d = 32; d >= 1; d >>= 1) lmax = dmax(lmax, __shfl_xor(lmax, d));
    // bit w: the words w.. of klo are all 0x00 / of khi all 0xFF
    unsigned rest0 = 0, restf = 0;
    {
        bool z = true, f = true;
#pragma unroll
        for (int w = kTreeKeyWords - 1; w >= 0; w--) {
            z = z && klo[w] == 0ull;
            f = f && khi[w] == ~0ull;
            rest0 |= (z ? 1u : 0u) << w;
            restf |= (f ? 1u : 0u) << w;
        }
    }
    ulonglong2 kk[kTreeFan / 2];
    for (int l = lmax; l >= 0; l--) {
        const bool on = act && l <= top;                              // this lane's descents have started
        const bool shared = on && !lo_out && !hi_out && plo == phi;
        const bool act_a = on && !lo_out;
        const uint64_t* np_a = act_a ? t.lvl[l] + plo * kTreeNodeWords : nullptr;
        NodeSide a, b;
        a.start();
        b.start();
#pragma unroll
        for (int w = 0; w < kTreeKeyWords; w++) {
            // (word 0 is always read; a later one only by the lanes whose side is not settled yet)
            const bool more_a = act_a && (w == 0 || (a.ties && !((rest0 >> w) & 1u)));
            const bool more_b = shared && (w == 0 || (b.ties && !((restf >> w) & 1u)));
            if (w > 0 && !__any(more_a || more_b)) break;
            coop_load_line(s, (more_a || more_b) ? np_a + w * kTreeFan : nullptr, kk);
            if (more_a) a.word(kk, klo[w]);
            if (more_b) b.word(kk, khi[w]);
        }
        const bool act_b = on && !hi_out && !shared;                  // the upper descent on its own path
        if (__any(act_b)) {
            const uint64_t* np_b = act_b ? t.lvl[l] + phi * kTreeNodeWords : nullptr;
#pragma unroll
            for (int w = 0; w < kTreeKeyWords; w++) {
                const bool more_b = act_b && (w == 0 || (b.ties && !((restf >> w) & 1u)));
                if (w > 0 && !__any(more_b)) break;
                coop_load_line(s, more_b ? np_b + w * kTreeFan : nullptr, kk);
                if (more_b) b.word(kk, khi[w]);
            }
        }
        if (act_a) { plo = plo * kTreeFan + a.base; lo_out = plo >= t.len[l]; }
        if (on && !hi_out) { phi = phi * kTreeFan + b.base + b.ties; hi_out = phi >= t.len[l]; }
    }
    lo = lo_out ? t.n : plo;
    hi = hi_out ? t.n : phi;
}

// A query longer than the tree's keys, inside [lo, hi) = the ranks that share its first kTreeKeyBytes (16) bytes:
// ONE bisection until a probe lands on a suffix that starts with q (or the range is empty: no match), then
// the two ends are searched on either side of that rank -- a bounded gallop (the interval is usually short
// against the range), then a bisection of what is left.  About log2(range) + 2 log2(#matches) probes of two
// lines (SA entry, text) where separate searches for start (:244-246) and end (:247-250) take 2 log2(range).
// Comparisons skip the bytes the keys have settled; the query's next 32 bytes are held in registers as
// big-endian words (a probe then reads the table entry and the text, never the query).
constexpr uint64_t kTreeKeyBytes = 8 * kTreeKeyWords;
constexpr int kQueryWords = 4;
// 8 bytes of p[0..len) at offset k as a big-endian integer, zero-padded past len
__device__ __forceinline__ uint64_t be64_at(const uint8_t* __restrict__ p, uint64_t k, uint64_t len)
{
    uint64_t v = 0;
    if (k + 8 <= len) {
        __builtin_memcpy(&v, p + k, 8);
        return __builtin_bswap64(v);
    }
    for (unsigned j = 0; j < 8; j++) v = (v << 8) | (k + j < len ? (uint64_t)p[k + j] : 0ull);
    return v;
}
struct LongQueryKey {
    uint64_t w[kQueryWords];                                          // bytes [16 + 8 i, 24 + 8 i) of the query
    const uint8_t* q;
    uint64_t m;
};
__device__ __forceinline__ LongQueryKey long_query_key(const uint8_t* __restrict__ q, uint64_t m)
{
    LongQueryKey k;
#pragma unroll
    for (int i = 0; i < kQueryWords; i++) k.w[i] = be64_at(q, kTreeKeyBytes + 8u * i, m);
    k.q = q;
    k.m = m;
    return k;
}
// three-way comparison of q with the suffix at s on bytes [16, min(m, n - s)): <0 q smaller, >0 q larger, 0 equal there
__device__ __forceinline__ int compare_beyond_keys(const LongQueryKey& key, const uint8_t* __restrict__ text, uint64_t n, uint64_t s)
{
    const uint64_t len = n - s, lim = key.m < len ? key.m : len;
    const uint8_t* tp = text + s;
#pragma unroll
    for (int i = 0; i < kQueryWords; i++) {
        const uint64_t k = kTreeKeyBytes + 8u * i;
        if (k >= lim) return 0;
        uint64_t x = key.w[i], y = be64_at(tp, k, len);
        if (lim - k < 8) {
            const uint64_t mask = ~0ull << (8u * (8u - (unsigned)(lim - k)));
            x &= mask;
            y &= mask;
        }
        if (x != y) return x < y ? -1 : 1;
    }
    for (uint64_t k = kTreeKeyBytes + 8u * kQueryWords; k < lim; k += 8) {      // (queries of more than 48 bytes)
        uint64_t x = be64_at(key.q, k, key.m), y = be64_at(tp, k, len);
        if (lim - k < 8) {
            const uint64_t mask = ~0ull << (8u * (8u - (unsigned)(lim - k)));
            x &= mask;
            y &= mask;
        }
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
// The three searches as ONE loop with one probe site (phase 0: the bisection for a match, 1 / 2: gallop and
// bisection for the start, 3 / 4: the same for the end): lanes in different phases still probe together.
__device__ __forceinline__ void long_query_interval(const uint8_t* __restrict__ q, uint64_t m, const uint8_t* __restrict__ text,
                                                    uint64_t n, const uint32_t* __restrict__ sa, uint64_t lo, uint64_t hi,
                                                    uint64_t& start, uint64_t& end)
{
    const LongQueryKey key = long_query_key(q, m);
    start = end = 0;
    uint64_t a = lo, b = hi, hit = 0, top = hi, step = 1;
    int phase = 0;
    for (;;) {
        uint64_t r = 0;
        bool probe = false;
        while (!probe && phase < 5) {                                 // settle which rank to probe next
            if (phase == 0) {                                         // ranks < a are smaller than q, ranks >= b larger and no match
                if (a < b) { r = (a + b) >> 1; probe = true; } else phase = 6;              // (6: no match at all)
            } else if (phase == 1) {                                  // the first match lies in [a, b], b is a match
                if (step <= 8 && b > a) { r = b - a > step ? b - step : a; probe = true; } else phase = 2;
            } else if (phase == 2) {
                if (a < b) { r = (a + b) >> 1; probe = true; }
                else { start = b; a = hit + 1; b = top; step = 1; phase = 3; }
            } else if (phase == 3) {                                  // the end lies in [a, b], everything in [hit, a) matches
                if (step <= 8 && a < b) { r = b - a > step ? a + step - 1 : b - 1; probe = true; } else phase = 4;
            } else {
                if (a < b) { r = (a + b) >> 1; probe = true; } else { end = a; phase = 5; }
            }
        }
        if (!probe) break;
        const uint64_t s = sa[r];
        const int c = compare_beyond_keys(key, text, n, s);
        const bool match = c == 0 && n - s >= m;                      // (a suffix that is a proper prefix of q sorts before q)
        if (phase == 0) {
            if (c < 0) b = r;
            else if (!match) a = r + 1;
            else { hit = r; top = b; b = r; step = 1; phase = 1; }
        } else if (phase == 1) {
            if (match) { b = r; step <<= 1; } else { a = r + 1; phase = 2; }
        } else if (phase == 2) {
            if (match) b = r; else a = r + 1;
        } else if (phase == 3) {
            if (match) { a = r + 1; step <<= 1; } else { b = r; phase = 4; }
        } else {
            if (match) a = r + 1; else b = r;
        }
    }
    if (phase == 6) start = end = 0;
}
__device__ __forceinline__ void query_write(uint64_t qi, uint64_t start, uint64_t end, const uint32_t* __restrict__ sa,
                                            uint32_t* __restrict__ start_out, uint32_t* __restrict__ end_out,
                                            uint8_t* __restrict__ found_out, uint32_t* __restrict__ any_out)
{
    const bool found = end > start;
    if (!found) start = end = 0;
    if (start_out) start_out[qi] = (uint32_t)start;
    if (end_out) end_out[qi] = (uint32_t)end;
    if (found_out) found_out[qi] = found ? 1 : 0;
    if (any_out) any_out[qi] = found ? sa[start] : 0xFFFFFFFFu;
}

// Phase 1: the two descents of every query; queries of <= 16 bytes (and misses) are answered here.  The others
// cost ten times as many lines; with `work` they are appended to a list (query, lo, hi) for phase 2, so that a
// wave never idles 60 lanes while 4 of them bisect on the text; without it they are finished in place.
struct LongQuery { uint32_t qi, lo, hi; };
// (4 workgroups per CU is what its 38 KB of LDS allow; asking for 6 or 8 -- fewer registers -- measured the same, 8 spills)
__global__ void __launch_bounds__(kBlock, 4)
k_query_batch_tree(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa, KeyTree tree,
                   const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, uint64_t nq,
                   uint32_t* __restrict__ start_out, uint32_t* __restrict__ end_out,
                   uint8_t* __restrict__ found_out, uint32_t* __restrict__ any_out, const uint32_t* __restrict__ order,
                   LongQuery* __restrict__ work, uint32_t* __restrict__ work_count, DirParams dp)
{
    // `order` (optional): the queries sorted by their first 8 bytes -- neighbouring lanes then walk the same
    // tree nodes and, inside a range of suffixes sharing those bytes, probe the same SA entries and text lines
    __shared__ CoopSmem coop;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const unsigned lane = lane_id();
    for (uint64_t base = (uint64_t)blockIdx.x * kBlock + (threadIdx.x & ~63u); base < nq; base += stride) {
        const uint64_t slot = base + lane;
        const bool live = slot < nq;
        const uint64_t qi = live ? (order ? (uint64_t)order[slot] : slot) : 0;
        const uint8_t* q = qbytes + (live ? qoff[qi] : 0);
        const uint64_t m = live ? qoff[qi + 1] - qoff[qi] : 0;
        uint64_t start = 0, end = 0;
        bool later = false;
        uint64_t lo = 0, hi = 0;
        uint64_t klo[kTreeKeyWords], khi[kTreeKeyWords];
#pragma unroll
        for (int w = 0; w < kTreeKeyWords; w++) klo[w] = khi[w] = 0;
        int top = tree.levels - 1;
        uint64_t node = 0;
        bool none = true;                                             // no descent: empty text / query, or the directory says no
        if (n != 0 && m != 0) {                                       // :228-229
            // the query's first 16 bytes as the two ends of its key range: padded with 0x00 and with 0xFF
            none = false;
#pragma unroll
            for (int w = 0; w < kTreeKeyWords; w++) {
                klo[w] = be64_at(q, 8u * w, m);
                const uint64_t have = m > 8u * w ? m - 8u * w : 0;    // bytes of the query in this word
                khi[w] = klo[w] | (have >= 8 ? 0ull : (have == 0 ? ~0ull : ~0ull >> (8u * (unsigned)have)));
            }
            // The bucket directory first (dp.dir, optional): the ranks [d_lo, d_hi) whose first symbols have q's code
            // prefix contain both answers, so the descents start at the lowest node that spans them instead of the
            // root.
            if (dp.dir) {
                const int L = m < (uint64_t)dp.k ? (int)m : dp.k;
                uint64_t c = 0;
                for (int j = 0; j < L; j++) {                         // (L <= 16: the bytes are in the key)
                    const uint32_t sym = dp.lut[(unsigned)(klo[j >> 3] >> (56 - 8 * (j & 7))) & 0xFFu];
                    none |= sym == 0u;                                // a byte the text does not contain
                    c = (c << dp.bits) | (uint64_t)(sym - 1u);
                }
                if (!none) {
                    const int have = L * dp.bits;
                    uint64_t c_lo, c_hi;
                    if (have >= dp.dbits) { c_lo = c >> (have - dp.dbits); c_hi = c_lo + 1; }
                    else { c_lo = c << (dp.dbits - have); c_hi = (c + 1) << (dp.dbits - have); }
                    const uint64_t d_lo = dp.dir[c_lo], d_hi = dp.dir[c_hi];
                    none = d_lo >= d_hi;
                    for (int l = 0; l < tree.levels - 1; l++) {       // the lowest node [p F^(l+1), (p+1) F^(l+1)] with both ends inside
                        const int sh = 4 * (l + 1);
                        const uint64_t p = d_lo >> sh;
                        if (d_hi <= ((p + 1) << sh)) { top = l; node = p; break; }
                    }
                }
            }
        }
        tree_bounds_wave(coop, tree, klo, khi, !none, top, node, lo, hi);          // (every lane of the wave takes part)
        if (!none) {
            if (lo < hi) {
                if (m <= kTreeKeyBytes) {
                    // every rank in [lo, hi) starts with q, except suffixes shorter than q whose padding imitates
                    // q's zero bytes (q ends in 0x00 then): they are the first entries of the range
                    if (q[m - 1] == 0)
                        while (lo < hi && n - (uint64_t)sa[lo] < m) lo++;
                    start = lo;
                    end = hi;
                } else if (work) {
                    later = true;
                } else {
                    long_query_interval(q, m, text, n, sa, lo, hi, start, end);
                }
            }
        }
        if (work) {                                                   // (uniform: every lane of the wave is here)
            const unsigned long long votes = __ballot(later);
            if (votes) {
                const int leader = __ffsll((long long)votes) - 1;
                uint32_t at = 0;
                if ((int)lane == leader) at = atomicAdd(work_count, (uint32_t)__popcll(votes));
                at = __shfl(at, leader);
                if (later) work[at + lanes_below(votes)] = LongQuery{(uint32_t)qi, (uint32_t)lo, (uint32_t)hi};
            }
        }
        if (live && !later) query_write(qi, start, end, sa, start_out, end_out, found_out, any_out);
    }
}
// Phase 2: the listed queries, one per lane, every lane busy.
__global__ void __launch_bounds__(kBlock, 8)
k_query_tree_long(const uint8_t* __restrict__ text, uint64_t n, const uint32_t* __restrict__ sa,
                  const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff,
                  const LongQuery* __restrict__ work, const uint32_t* __restrict__ work_count,
                  uint32_t* __restrict__ start_out, uint32_t* __restrict__ end_out,
                  uint8_t* __restrict__ found_out, uint32_t* __restrict__ any_out)
{
    const uint64_t count = *work_count;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) {
        const LongQuery w = work[k];
        const uint8_t* q = qbytes + qoff[w.qi];
        const uint64_t m = qoff[w.qi + 1] - qoff[w.qi];
        uint64_t start, end;
        long_query_interval(q, m, text, n, sa, w.lo, w.hi, start, end);
        query_write(w.qi, start, end, sa, start_out, end_out, found_out, any_out);
    }
}

// (first 8 bytes of query k, big-endian, zero-padded) for the ordering above
__global__ void __launch_bounds__(kBlock)
k_query_keys(const uint8_t* __restrict__ qbytes, const uint64_t* __restrict__ qoff, uint64_t nq, uint64_t* __restrict__ keys,
             uint32_t* __restrict__ idx)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < nq; k += stride) {
        const uint8_t* q = qbytes + qoff[k];
        const uint64_t m = qoff[k + 1] - qoff[k];
        uint64_t key = 0;
        for (unsigned j = 0; j < 8; j++) key = (key << 8) | (j < m ? (uint64_t)q[j] : 0ull);
        keys[k] = key;
        idx[k] = (uint32_t)k;
    }
}

// words (u64) of one allocation that holds all levels, each padded to whole nodes plus one spare node
uint64_t key_tree_words(uint64_t n)
{
    uint64_t words = 0, len = n;
    for (int l = 0; l < kTreeMaxLevels; l++) {
        words += ((len + kTreeFan - 1) / kTreeFan + 1) * kTreeNodeWords;
        if (len <= (uint64_t)kTreeFan) break;
        len = (len + kTreeFan - 1) / kTreeFan;
    }
    return words;
}
// d_tree: key_tree_words(n) u64.  level_offsets_out[l] = word offset of level l, *levels_out = count.
int key_tree_build_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint64_t* d_tree, uint64_t* level_offsets_out,
                       int* levels_out, hipStream_t st)
{
    SFX_HIP(hipMemsetAsync(d_tree, 0xFF, key_tree_words(n) * sizeof(uint64_t), st));       // padding keys = max
    const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("tree_leaves", (double)n * (4 + 8 + 8 * kTreeKeyWords), k_tree_leaves, grid, kBlock, st, d_text, n, d_sa, d_tree);
    uint64_t off = 0, len = n;
    int l = 0;
    level_offsets_out[0] = 0;
    while (len > (uint64_t)kTreeFan && l + 1 < kTreeMaxLevels) {
        const uint64_t next_off = off + ((len + kTreeFan - 1) / kTreeFan + 1) * kTreeNodeWords;
        const uint64_t next_len = (len + kTreeFan - 1) / kTreeFan;
        const unsigned g = (unsigned)dmin<uint64_t>((next_len + kBlock - 1) / kBlock, kMaxGrid);
        SFX_LAUNCH("tree_level", (double)next_len * 16 * kTreeKeyWords, k_tree_level, g, kBlock, st, (const uint64_t*)(d_tree + off), len,
                   d_tree + next_off, next_len);
        off = next_off;
        len = next_len;
        level_offsets_out[++l] = off;
    }
    *levels_out = l + 1;
    return SFX_OK;
}
// scratch of a batch: the list of phase 2 (a counter, then nq entries); with `ordered`, behind it the query
// ordering: 2 * nq u64 + 2 * nq u32 + radix_scratch_words(nq) u32
// smaller batches are finished in place (one launch); SFX_QUERY_PHASE_MIN (tests) moves the threshold
uint64_t query_two_phase_min()
{
    static const uint64_t v = [] {
        const char* e = dev_env("SFX_QUERY_PHASE_MIN");
        return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)4096;
    }();
    return v;
}
static uint64_t query_work_bytes(uint64_t nq) { return (256 + nq * sizeof(LongQuery) + 255) & ~uint64_t(255); }
uint64_t query_scratch_bytes(uint64_t nq, bool ordered)
{
    uint64_t b = query_work_bytes(nq);
    if (ordered) b += 2 * nq * sizeof(uint64_t) + 2 * nq * sizeof(uint32_t) + radix_scratch_words(nq) * sizeof(uint32_t) + 1024;
    return b;
}
int query_batch_tree_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, const uint64_t* d_tree,
                         const uint64_t* level_offsets, int levels, const uint8_t* d_q, const uint64_t* d_qoff, uint64_t nq,
                         uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, hipStream_t st, void* scratch,
                         bool ordered, const uint32_t* d_dir, const uint16_t* d_lut256, int bits, int k, int dbits)
{
    if (nq == 0) return SFX_OK;
    if (!d_qoff || !d_tree || levels < 1 || levels > kTreeMaxLevels) return SFX_ERR_ARG;
    KeyTree t;
    uint64_t len = n;
    for (int l = 0; l < kTreeMaxLevels; l++) {
        t.lvl[l] = l < levels ? d_tree + level_offsets[l] : nullptr;
        t.len[l] = l < levels ? len : 0;
        len = (len + kTreeFan - 1) / kTreeFan;
    }
    t.levels = levels;
    t.n = n;
    const unsigned grid = (unsigned)dmin<uint64_t>((nq + kBlock - 1) / kBlock, kMaxGrid);
    // the directory narrows the descents when its key lies inside the tree's 16 bytes (always, except for 1-bit symbols)
    static const bool want_dir = [] { const char* e = dev_env("SFX_TREE_DIR"); return !e || atoi(e) != 0; }();
    DirParams dp = {want_dir && d_lut256 && k <= (int)kTreeKeyBytes ? d_dir : nullptr, d_lut256, bits, k, dbits};
    const uint32_t* order = nullptr;
    const bool two_phase = scratch && nq >= query_two_phase_min() && nq <= 0xFFFFFFFFull;
    uint32_t* work_count = two_phase ? reinterpret_cast<uint32_t*>(scratch) : nullptr;
    LongQuery* work = two_phase ? reinterpret_cast<LongQuery*>(reinterpret_cast<char*>(scratch) + 256) : nullptr;
    if (two_phase) SFX_HIP(hipMemsetAsync(work_count, 0, sizeof(uint32_t), st));
    if (two_phase && ordered) {
        // sort (first 8 bytes, query number): 8 passes over 12-byte elements of a small array
        char* w = reinterpret_cast<char*>(scratch) + query_work_bytes(nq);
        uint64_t* k0 = reinterpret_cast<uint64_t*>(w);
        uint64_t* k1 = k0 + nq;
        uint32_t* v0 = reinterpret_cast<uint32_t*>(k1 + nq);
        uint32_t* v1 = v0 + nq;
        uint32_t* scr = v1 + nq;
        scr = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(scr) + 255) & ~uintptr_t(255));
        SFX_LAUNCH("query_keys", (double)nq * 28, k_query_keys, grid, kBlock, st, d_q, d_qoff, nq, k0, v0);
        int in1 = 0;
        SFX_TRY(radix_sort_kv64(k0, v0, k1, v1, nq, 0, 64, scr, st, &in1, nullptr, nullptr));
        order = in1 ? v1 : v0;
    }
    // two descents of one 128-byte line per level (a second one at the bottom levels of queries longer than 8
    // bytes), a few probes of 2 lines beyond 16 bytes
    SFX_LAUNCH("query_batch_tree", (double)nq * (2.0 * (levels + 3) * 128 + 2 * 256), k_query_batch_tree, grid, kBlock, st, d_text, n,
               d_sa, t, d_q, d_qoff, nq, d_start, d_end, d_found, d_any, order, work, work_count, dp);
    if (two_phase)
        SFX_LAUNCH("query_tree_long", (double)nq * 0.4 * 40 * 256, k_query_tree_long, grid, kBlock, st, d_text, n, d_sa, d_q,
                   d_qoff, (const LongQuery*)work, (const uint32_t*)work_count, d_start, d_end, d_found, d_any);
    return SFX_OK;
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
