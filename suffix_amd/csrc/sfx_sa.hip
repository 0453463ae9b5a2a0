// sfx_sa.hip -- suffix-array construction on the device (replaces sais_table /
// sais, /root/reference/src/table.rs:378-574).
//
// The suffix array of a text is unique (all suffixes differ; order = bytewise
// lexicographic with a proper prefix first, naive_table :367-376), so the engine
// is free in HOW it sorts; only the final u32 array must be identical.  SA-IS's
// L-then-S induction (:419-448, :544-573) is a serial pointer chase through the
// buckets, so the MI355X engine keeps the reference's *bucket* structure but
// replaces induction by whole-array bucket refinement, every step a streaming
// scan / histogram / scatter kernel:
//
//   1. alphabet:   byte histogram (cf. Bins::find_sizes :686-704) -> dense symbol
//                  codes of `bits` bits each (sigma = 4 -> 2 bits); the text is
//                  packed once, 2^k symbols per 32-bit word (PackedText).
//   2. initial buckets: every suffix gets a key = its first k symbols (k = 16
//                  for DNA in 32 bits), read straight out of the packed text by
//                  the first pass of one LSD radix sort (sfx_radix.hip): all
//                  suffixes land in k-symbol bucket order -- the bucket sort of
//                  the reference's level 0 plus its first recursion levels at once.
//   3. bucket ranks ("naming", cf. :465-482): adjacent-compare flags + device
//                  scan find the bucket heads; buckets of size 1 are final.
//   4. refinement rounds (the "recursive sort", cf. :496-500): suffixes still
//                  sharing a bucket are compacted; each gets the composite key
//                  (dense bucket id, key2); radix sort; new flags/scan split the
//                  buckets.  key2 is either the NEXT k symbols (text round, h += k;
//                  needs nothing but the packed text) or the RANK of the suffix h
//                  symbols further on (rank round, h doubles; needs ISA[suffix] =
//                  head slot of its bucket).  When the initial sort leaves few
//                  suffixes unresolved the first round is a text round and the
//                  n-element ISA scatter is skipped unless a later round needs it.
//                  Terminates when every bucket is a singleton (<= log2 n rank rounds).
//
// Short suffixes: keys are zero-padded past the end of the text; a suffix whose
// first h symbols run off the end ("consumed") gets key2 = n-1-i, which is
// smaller than every real key2 and decreasing in i, i.e. "shorter first",
// exactly how the reference's virtual sentinel orders them (:422-425).
//
// The same kernels serve the range-partitioned (multi-GPU) build, where a rank
// sorts only the suffixes whose leading key bits fall in its bucket range and
// refines with text rounds only (no ranks of foreign suffixes needed).
#include <math.h>
#include <stdio.h>
#include <thread>
#include <vector>
#include <chrono>
#include <string.h>

#include "sfx_host.hpp"

namespace sfx {

constexpr int kPackWords = kBlock;       // packed words per workgroup step in k_pack_text
constexpr int kMaxTopBits = 14;          // 16384 u32 bins = 64 KiB of LDS

// ---------------------------------------------------------------------------------
// 1. alphabet and packed text
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_byte_hist(const uint8_t* __restrict__ text, uint64_t begin, uint64_t end,
            unsigned long long* __restrict__ bins)
{
    // 16 sub-counters per byte value, picked by lane: small alphabets (DNA: 4 values) would
    // otherwise serialise a whole wave on 4 LDS addresses
    constexpr unsigned kCols = 16;
    __shared__ uint32_t h[256 * kCols];
    const unsigned tid = threadIdx.x, col = tid & (kCols - 1u);
    for (unsigned i = tid; i < 256 * kCols; i += kBlock) h[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    // head up to the first 16-byte boundary, 16-byte vectors, tail
    uint64_t vb = begin + ((16u - (unsigned)((reinterpret_cast<uintptr_t>(text) + begin) & 15u)) & 15u);
    if (vb > end) vb = end;
    const uint64_t nvec = (end - vb) / 16;
    const uint4* t16 = reinterpret_cast<const uint4*>(text + vb);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + tid; i < nvec; i += stride) {
        const uint4 v = t16[i];
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&h[(wds[k] & 255u) * kCols + col], 1u);
            atomicAdd(&h[((wds[k] >> 8) & 255u) * kCols + col], 1u);
            atomicAdd(&h[((wds[k] >> 16) & 255u) * kCols + col], 1u);
            atomicAdd(&h[(wds[k] >> 24) * kCols + col], 1u);
        }
    }
    if (blockIdx.x == 0) {
        for (uint64_t i = begin + tid; i < vb; i += kBlock) atomicAdd(&h[text[i] * kCols + col], 1u);
        for (uint64_t i = vb + nvec * 16 + tid; i < end; i += kBlock) atomicAdd(&h[text[i] * kCols + col], 1u);
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (unsigned k = 0; k < kCols; k++) c += h[tid * kCols + ((k + tid) & (kCols - 1u))];
    if (c) atomicAdd(&bins[tid], (unsigned long long)c);
}

// Which byte values occur?  The single-GPU build only needs the alphabet, not the counts
// (the sorted `alphas` of Bins::find_sizes :700): plain LDS stores of a flag, 16 bytes of
// text per thread and step -- no atomics, so 4-symbol DNA does not serialise on 4 counters.
// bins[c] = 1 for every byte value c that occurs (bins zeroed by the caller).
__global__ void __launch_bounds__(kBlock)
k_byte_presence(const uint8_t* __restrict__ text, uint64_t n, unsigned long long* __restrict__ bins)
{
    __shared__ uint32_t present[256];
    const unsigned tid = threadIdx.x;
    present[tid] = 0;
    __syncthreads();
    const uint64_t n16 = n / 16;
    const uint4* t16 = reinterpret_cast<const uint4*>(text);        // text comes 16-byte aligned or is handled below
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    if ((reinterpret_cast<uintptr_t>(text) & 15u) == 0) {
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + tid; i < n16; i += stride) {
            const uint4 v = t16[i];
            const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                present[wds[k] & 255u] = 1u;
                present[(wds[k] >> 8) & 255u] = 1u;
                present[(wds[k] >> 16) & 255u] = 1u;
                present[wds[k] >> 24] = 1u;
            }
        }
        for (uint64_t i = n16 * 16 + (uint64_t)blockIdx.x * kBlock + tid; i < n; i += stride) present[text[i]] = 1u;
    } else {
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + tid; i < n; i += stride) present[text[i]] = 1u;
    }
    __syncthreads();
    if (present[tid]) bins[tid] = 1ull;
}

// dense symbol codes from the 256 global byte counts (one workgroup, thread = byte value)
// counts of the (previous symbol, symbol) pairs of the text in dense symbol codes (make_lut's), for the context codes of
// k_ht_keys_ctx: out[prev * sigma + cur] (u64).  One workgroup per CU, sigma^2 LDS counters (sigma <= kHtCtxSigmaMax), 16 bytes
// per thread and step; the pair that starts a thread's 16 bytes takes the byte before them.
__global__ void __launch_bounds__(1024)
k_bigram_hist(const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ lut, int sigma, unsigned long long* __restrict__ out)
{
    __shared__ uint32_t h[kHtCtxSigmaMax * kHtCtxSigmaMax];
    __shared__ uint8_t s_lut[256];
    const unsigned tid = threadIdx.x;
    if (tid < 256u) s_lut[tid] = lut[tid];
    for (unsigned i = tid; i < (unsigned)(sigma * sigma); i += 1024u) h[i] = 0u;
    __syncthreads();
    const uint64_t nvec = n / 16, stride = (uint64_t)gridDim.x * 1024u;
    const bool aligned = (reinterpret_cast<uintptr_t>(text) & 15u) == 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024u + tid; i < nvec; i += stride) {
        uint8_t b[16];
        if (aligned) {
            const uint4 v = reinterpret_cast<const uint4*>(text)[i];
            __builtin_memcpy(b, &v, 16);
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) b[k] = text[i * 16 + k];
        }
        unsigned prev = i ? (unsigned)s_lut[text[i * 16 - 1]] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned cur = (unsigned)s_lut[b[k]];
            if (prev != 0xFFFFFFFFu) atomicAdd(&h[prev * (unsigned)sigma + cur], 1u);
            prev = cur;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {                                  // the tail behind the last whole 16 bytes
        for (uint64_t p = dmax<uint64_t>(nvec * 16, 1); p < n; p++)
            atomicAdd(&h[(unsigned)s_lut[text[p - 1]] * (unsigned)sigma + (unsigned)s_lut[text[p]]], 1u);
    }
    __syncthreads();
    for (unsigned i = tid; i < (unsigned)(sigma * sigma); i += 1024u)
        if (h[i]) atomicAdd(&out[i], (unsigned long long)h[i]);
}

__global__ void __launch_bounds__(kBlock)
k_make_lut(const unsigned long long* __restrict__ bins, uint8_t* __restrict__ lut)
{
    __shared__ uint32_t part[kWavesPerBlock];
    uint32_t present = bins[threadIdx.x] ? 1u : 0u, total;
    uint32_t code = block_scan_add_excl(present, part, total);
    lut[threadIdx.x] = (uint8_t)code;
}

// words[j] = symbols of positions [j*spw, (j+1)*spw), big-endian, `bits` bits each,
// 0 past the end; plus 3 trailing zero words (see PackedText).  A tile is
// kPackWords words = kPackWords*spw text positions.
__global__ void __launch_bounds__(kBlock)
k_pack_text(const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ lut, int bits,
            int spw, uint64_t tiles_per_block, uint64_t n_words_total, uint32_t* __restrict__ words)
{
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t codes[kPackWords * 32];
    const unsigned tid = threadIdx.x;
    const unsigned tile_pos = (unsigned)kPackWords * (unsigned)spw;
    s_lut[tid] = lut[tid];
    __syncthreads();
    uint64_t wbegin = (uint64_t)blockIdx.x * tiles_per_block * kPackWords;
    uint64_t wend = wbegin + tiles_per_block * kPackWords;
    if (wend > n_words_total) wend = n_words_total;
    for (uint64_t wt = wbegin; wt < wend; wt += kPackWords) {
        const uint64_t tile = wt * (uint64_t)spw;
        for (unsigned j = tid; j < tile_pos; j += kBlock) {
            uint64_t g = tile + j;
            codes[j] = (g < n) ? s_lut[text[g]] : (uint8_t)0;
        }
        __syncthreads();
        uint64_t gw = wt + tid;                         // kPackWords == kBlock: one word per thread
        if (gw < wend) {
            uint32_t w = 0;
            for (int s = 0; s < spw; s++) w = (w << bits) | codes[tid * (unsigned)spw + s];
            words[gw] = w;
        }
        __syncthreads();
    }
}

// Fast path of the above for bits in {8, 4, 2, 1} (spw = 32/bits exactly) and a 16-byte aligned
// text: a thread makes one word from spw consecutive bytes fetched with one or two wide
// loads -- consecutive threads read consecutive bytes, no LDS staging, only the LUT lives
// in LDS.  The last word (< spw bytes left) and the zero tail take the guarded path.
// (round 5: also 7-bit codes, four to a word -- natural-language ASCII: the LDS-staged general kernel took 1.2 ms per 10^9)
template <int SPW, int BITS = 32 / SPW>
__global__ void __launch_bounds__(kBlock)
k_pack_text_pow2(const uint8_t* __restrict__ text, uint64_t n, const uint8_t* __restrict__ lut,
                 uint64_t n_words_total, uint32_t* __restrict__ words)
{
    static_assert(BITS * SPW <= 32, "a word holds SPW codes of BITS bits in its low bits");
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < n_words_total; q += stride) {
        const uint64_t p0 = q * SPW;
        uint32_t w = 0;
        if (p0 + SPW <= n) {
            uint8_t raw[SPW];
            if (SPW == 4) {
                *reinterpret_cast<uint32_t*>(raw) = *reinterpret_cast<const uint32_t*>(text + p0);
            } else if (SPW == 8) {
                *reinterpret_cast<uint2*>(raw) = *reinterpret_cast<const uint2*>(text + p0);
            } else {
#pragma unroll
                for (int k = 0; k < SPW / 16; k++)
                    *reinterpret_cast<uint4*>(raw + 16 * k) = *reinterpret_cast<const uint4*>(text + p0 + 16 * k);
            }
#pragma unroll
            for (int k = 0; k < SPW; k++) w = (w << BITS) | s_lut[raw[k]];
        } else {
            for (int k = 0; k < SPW; k++) w = (w << BITS) | (p0 + k < n ? (uint32_t)s_lut[text[p0 + k]] : 0u);
        }
        words[q] = w;
    }
}

// ---------------------------------------------------------------------------------
// 2. partitioned build: bucket-boundary histogram and range filter
// ---------------------------------------------------------------------------------
// Counts the top `top_bits` bits of the key of every suffix starting in [begin, end).
// Straight from the raw text: the top bits of a key are its first nsym = ceil(top_bits/bits)
// symbols, so a thread rolls a window of nsym symbol codes over 16 consecutive positions
// (symbol codes from the global byte counts, rebuilt per workgroup: no packed text, no
// scratch, nothing to allocate).  Positions past the end of the text read as code 0, the
// same zero padding the packed keys have.  Bins are privatised in LDS (2^top_bits u32
// <= 64 KiB) and flushed once per workgroup.
constexpr int kKeyHistRun = 16;
__global__ void __launch_bounds__(kBlock)
k_key_hist_raw(const uint8_t* __restrict__ text, uint64_t n, uint64_t begin, uint64_t end,
               const unsigned long long* __restrict__ byte_bins, int bits, int nsym, int top_bits,
               uint64_t chunk, unsigned long long* __restrict__ bins)
{
    __shared__ uint32_t h[1 << kMaxTopBits];
    __shared__ uint8_t lut[256];
    __shared__ uint32_t part[kWavesPerBlock];
    const unsigned tid = threadIdx.x, nbins = 1u << top_bits;
    {
        uint32_t present = byte_bins[tid] ? 1u : 0u, total;
        lut[tid] = (uint8_t)block_scan_add_excl(present, part, total);
    }
    for (unsigned i = tid; i < nbins; i += kBlock) h[i] = 0;
    __syncthreads();
    const uint64_t cb = begin + (uint64_t)blockIdx.x * chunk;
    const uint64_t ce = dmin<uint64_t>(cb + chunk, end);
    const uint32_t wmask = (1u << (nsym * bits)) - 1u;            // nsym*bits < top_bits + bits <= 22
    const int down = nsym * bits - top_bits;
    // a thread's run and its overhang (nsym - 1 more bytes) as two 16-byte vectors when the
    // chunk starts on a 16-byte boundary (chunks are multiples of kKeyHistRun = 16)
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(text) + cb) & 15u) == 0 && nsym - 1 <= 16;
    for (uint64_t i0 = cb + (uint64_t)tid * kKeyHistRun; i0 < ce; i0 += (uint64_t)kBlock * kKeyHistRun) {
        uint32_t wnd = 0;
        if (vec_ok && i0 + 2 * kKeyHistRun <= n) {
            const uint4 va = *reinterpret_cast<const uint4*>(text + i0);
            const uint4 vb = *reinterpret_cast<const uint4*>(text + i0 + kKeyHistRun);
            const uint32_t wds[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
            const int lim = kKeyHistRun + nsym - 1;
#pragma unroll
            for (int j = 0; j < 2 * kKeyHistRun; j++) {
                if (j < lim) {
                    const uint32_t c = (uint32_t)lut[(wds[j >> 2] >> (8 * (j & 3))) & 255u];
                    wnd = ((wnd << bits) | c) & wmask;
                    if (j >= nsym - 1 && i0 + (uint64_t)(j - (nsym - 1)) < ce) atomicAdd(&h[wnd >> down], 1u);
                }
            }
            continue;
        }
        for (int j = 0; j < kKeyHistRun + nsym - 1; j++) {
            const uint64_t p = i0 + (uint64_t)j;
            const uint32_t c = p < n ? (uint32_t)lut[text[p]] : 0u;
            wnd = ((wnd << bits) | c) & wmask;
            if (j >= nsym - 1 && p - (uint64_t)(nsym - 1) < ce) atomicAdd(&h[wnd >> down], 1u);
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < nbins; i += kBlock)
        if (h[i]) atomicAdd(&bins[i], (unsigned long long)h[i]);
}

// Emit (key - first key of the range, suffix) for the suffixes whose top key bits fall in [bin_lo, bin_hi)
// (32-bit keys: as E64 elements in kout, vout unused).  Order and equality are those of the keys, and a slice's keys
// fit bits_for(range width - 1) bits: what the hybrid initial sort of a slice goes by (radix_sort_e64, elem_bits).
// phase 0 counts per workgroup, phase 1 writes at the scanned offsets (stream
// compaction; order = text order).
// (A one-pass variant that reserves output per tile with an atomic cursor was measured
// 2x slower: the returning device-scope atomic sits on every tile's critical path.)
constexpr int kFilterMaxSpw = 32;                    // symbols per packed word: floor(32 / bits) <= 32
constexpr int kFilterStageSpw = 16;                  // tiles of up to 256 x 16 positions are compacted in LDS
// A thread takes one packed word = spw consecutive positions: their keys are windows of the
// 2-3 words it loads (coalesced), no index arithmetic per position.  phase 0 counts the kept
// positions per chunk, phase 1 (after the scan of the counts) emits them in position order.
// BITS != 0: symbol width known at compile time (2 = DNA), every shift becomes a constant.
template <class KeyT, int BITS>
__global__ void __launch_bounds__(kBlock)
k_range_filter(PackedText src, int key_bits_used, int top_bits, uint32_t bin_lo, uint32_t bin_hi,
               uint64_t chunk_words, int phase, uint32_t* __restrict__ block_counts, uint64_t capacity,
               KeyT* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t* __restrict__ digit_partial)
{
    // digit_partial (32-bit keys, phase 1): the 8-bit digit counts of the emitted keys for the
    // sort that follows, [(pass * 256 + digit) * gridDim.x + blockIdx.x] -- the keys are in
    // registers here, the sort would read all its elements once more to count them
    constexpr int kDigitPasses = 4;
    __shared__ uint32_t dig[sizeof(KeyT) == 4 ? kDigitPasses * 256 : 1];
    const bool count_digits = sizeof(KeyT) == 4 && phase == 1 && digit_partial != nullptr;
    if (count_digits) {
        for (unsigned i = threadIdx.x; i < kDigitPasses * 256; i += kBlock) dig[i] = 0;
        __syncthreads();
    }
    const int digit_passes = (key_bits_used + 7) / 8;
    auto count_key = [&](uint32_t key) {
#pragma unroll
        for (int p = 0; p < kDigitPasses; p++) {
            if (p < digit_passes) {
                const int nb = key_bits_used - 8 * p < 8 ? key_bits_used - 8 * p : 8;
                atomicAdd(&dig[p * 256 + ((key >> (8 * p)) & ((1u << nb) - 1u))], 1u);
            }
        }
    };
    __shared__ uint32_t part[2][kWavesPerBlock];
    __shared__ uint64_t stage_k[kBlock * kFilterStageSpw];                       // 32 KiB
    __shared__ uint32_t stage_v[sizeof(KeyT) == 8 ? kBlock * kFilterStageSpw : 1];
    const unsigned tid = threadIdx.x;
    const int shift = key_bits_used - top_bits;
    const unsigned bits = BITS ? (unsigned)BITS : (unsigned)src.bits;
    const unsigned spw = BITS ? 32u / (unsigned)(BITS ? BITS : 1) : (unsigned)src.spw;
    const unsigned kbits = BITS ? spw * bits : (unsigned)src.kbits;
    const uint64_t n_words = (src.n + spw - 1) / spw;
    const uint64_t mask = (1ull << kbits) - 1ull;
    const KeyT key_lo = (KeyT)((uint64_t)bin_lo << shift);
    const KeyT key_span = (KeyT)((((uint64_t)(bin_hi - bin_lo)) << shift) - 1ull);   // (2^64 wraps to all ones: right)
    uint64_t wbegin = (uint64_t)blockIdx.x * chunk_words;          // chunk_words is a multiple of kBlock
    uint64_t wend = wbegin + chunk_words;
    if (wend > n_words) wend = n_words;
    uint64_t running = (phase == 1) ? (uint64_t)block_counts[blockIdx.x] : 0ull;
    unsigned par = 0;
    uint32_t mine = 0;
    for (uint64_t base = wbegin; base < wend; base += kBlock) {
        const uint64_t q = base + tid;
        const bool live = q < wend;
        const uint64_t p0 = q * spw;
        uint64_t x01 = 0, x12 = 0;                                 // the key windows: words (q, q+1) and (q+1, q+2)
        if (live) {
            const uint64_t w0 = src.words[q], w1 = src.words[q + 1];
            x01 = (w0 << kbits) | w1;
            if (sizeof(KeyT) == 8) x12 = (w1 << kbits) | (uint64_t)src.words[q + 2];
        }
        auto key_at = [&](unsigned j) -> KeyT {                    // key of position p0 + j, j < spw
            const unsigned sh = (spw - j) * bits;
            if (sizeof(KeyT) == 4 && kbits == 32u)                 // whole-word keys (DNA): one 32-bit funnel shift
                return (KeyT)(j ? (uint32_t)(x01 >> (sh & 31u)) : (uint32_t)(x01 >> 32));
            const uint64_t a = (x01 >> sh) & mask;
            if (sizeof(KeyT) == 4) return (KeyT)a;
            return (KeyT)((a << kbits) | ((x12 >> sh) & mask));
        };
        uint32_t keep = 0;
        if (live) {
            // bin in [bin_lo, bin_hi)  <=>  key - (bin_lo << shift) <= ((bin_hi - bin_lo) << shift) - 1
#pragma unroll
            for (unsigned j = 0; j < (unsigned)kFilterMaxSpw; j++) {
                if (j < spw) keep |= ((KeyT)(key_at(j) - key_lo) <= key_span ? 1u : 0u) << j;
            }
            const uint64_t left = src.n - p0;                      // positions past the end of the text: dropped
            if (left < spw) keep &= (1u << (unsigned)left) - 1u;
        }
        const uint32_t cnt = (uint32_t)__popc(keep);
        if (phase == 0) {                              // counting needs no order: per-thread sums, one reduction at the end
            mine += cnt;
            continue;
        }
        // exclusive prefix of the per-thread keep counts, one barrier (parity buffers)
        const uint32_t incl = wave_scan_add(cnt);
        if (lane_id() == 63) part[par][wave_id()] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (unsigned k = 0; k < (unsigned)kWavesPerBlock; k++) {
            const uint32_t qq = part[par][k];
            if (k < wave_id()) before += qq;
            total += qq;
        }
        par ^= 1u;
        if (spw <= (unsigned)kFilterStageSpw && total * 6u >= kBlock * spw) {
            // compact the tile in LDS, then write it out as one contiguous run (a thread's own
            // elements are up to spw * 8 bytes apart from its neighbour's: direct stores would
            // touch one line per lane)
            uint32_t at = before + incl - cnt;
            uint32_t k = keep;
            while (k) {
                const unsigned j = (unsigned)__ffs((int)k) - 1u;
                k &= k - 1u;
                const KeyT key = (KeyT)(key_at(j) - key_lo);
                if (count_digits) count_key((uint32_t)key);
                if (sizeof(KeyT) == 4) {                // E64 element: (key << 32) | suffix
                    stage_k[at] = ((uint64_t)key << 32) | (uint64_t)(uint32_t)(p0 + j);
                } else {
                    stage_k[at] = (uint64_t)key;
                    stage_v[at] = (uint32_t)(p0 + j);
                }
                at++;
            }
            __syncthreads();
            for (uint32_t i = tid; i < total; i += kBlock) {
                const uint64_t dst = running + i;
                if (dst < capacity) {
                    if (sizeof(KeyT) == 4) {
                        reinterpret_cast<uint64_t*>(kout)[dst] = stage_k[i];
                    } else {
                        kout[dst] = (KeyT)stage_k[i];
                        vout[dst] = stage_v[i];
                    }
                }
            }
            __syncthreads();
        } else {                                       // sparse tile (a narrow range of a long text): direct stores
            uint64_t dst = running + before + incl - cnt;
            uint32_t k = keep;
            while (k) {
                const unsigned j = (unsigned)__ffs((int)k) - 1u;
                k &= k - 1u;
                if (dst < capacity) {
                    const KeyT key = (KeyT)(key_at(j) - key_lo);
                    if (count_digits) count_key((uint32_t)key);
                    if (sizeof(KeyT) == 4) {
                        reinterpret_cast<uint64_t*>(kout)[dst] = ((uint64_t)key << 32) | (uint64_t)(uint32_t)(p0 + j);
                    } else {
                        kout[dst] = key;
                        vout[dst] = (uint32_t)(p0 + j);
                    }
                }
                dst++;
            }
        }
        running += total;
    }
    if (count_digits) {
        __syncthreads();
        for (int p = 0; p < digit_passes; p++)
            digit_partial[((uint64_t)p * 256 + tid) * gridDim.x + blockIdx.x] = dig[p * 256 + tid];
    }
    if (phase == 0) {
        for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
        if (lane_id() == 0) part[0][wave_id()] = mine;
        __syncthreads();
        if (tid == 0) {
            uint32_t c = 0;
            for (int w = 0; w < kWavesPerBlock; w++) c += part[0][w];
            block_counts[blockIdx.x] = c;
        }
    }
}

// exclusive scan of <= kMaxGrid per-workgroup counts (single workgroup); total -> out_total
__global__ void __launch_bounds__(kBlock)
k_scan_block_counts(uint32_t* __restrict__ counts, unsigned nb, uint32_t* __restrict__ out_total)
{
    __shared__ uint32_t part[kWavesPerBlock];
    uint32_t carry = 0;
    for (unsigned base = 0; base < nb; base += kBlock) {
        unsigned i = base + threadIdx.x;
        uint32_t v = (i < nb) ? counts[i] : 0u, total;
        uint32_t ex = block_scan_add_excl(v, part, total);
        if (i < nb) counts[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *out_total = carry;
}

// ---------------------------------------------------------------------------------
// 4a. composite keys for a refinement round
// ---------------------------------------------------------------------------------
// rank round: key2 = rank of suffix i+h (+h), or n-1-i when i+h runs off the text.
__global__ void __launch_bounds__(kBlock)
k_compose_rank_keys(const uint32_t* __restrict__ suf, const uint32_t* __restrict__ gid, uint64_t m,
                    const uint32_t* __restrict__ isa, uint64_t n, uint64_t h, int key2_bits,
                    uint64_t* __restrict__ keys)
{
    // 4 elements per thread and step: the dependent gathers (suffix -> rank) of all four
    // are in flight together instead of one memory round trip after the other
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q0 < m; q0 += U * stride) {
        uint64_t i[U];
        uint32_t g[U], rk[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t q = q0 + u * stride;
            i[u] = q < m ? suf[q] : 0;
            g[u] = q < m ? gid[q] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) rk[u] = (q0 + u * stride < m && i[u] + h < n) ? isa[i[u] + h] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t q = q0 + u * stride;
            if (q < m) {
                const uint64_t key2 = (i[u] + h < n) ? (uint64_t)rk[u] + h : (n - 1 - i[u]);
                keys[q] = ((uint64_t)g[u] << key2_bits) | key2;
            }
        }
    }
}

// text round: key2 = (1 << flag_shift) | the spw symbols at offset h, or n-1-i
// (< 2^flag_shift) when the suffix is consumed.
__global__ void __launch_bounds__(kBlock)
k_compose_text_keys(const uint32_t* __restrict__ suf, const uint32_t* __restrict__ gid, uint64_t m,
                    PackedText src, uint64_t h, int flag_shift, int key2_bits,
                    uint64_t* __restrict__ keys)
{
    constexpr int U = 4;                                   // see k_compose_rank_keys
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q0 < m; q0 += U * stride) {
        uint64_t i[U];
        uint32_t g[U], tk[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t q = q0 + u * stride;
            i[u] = q < m ? suf[q] : 0;
            g[u] = q < m ? gid[q] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) tk[u] = (q0 + u * stride < m && i[u] + h < src.n) ? packed_key32(src, i[u] + h) : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t q = q0 + u * stride;
            if (q < m) {
                const uint64_t key2 = (i[u] + h < src.n) ? ((1ull << flag_shift) | (uint64_t)tk[u]) : (src.n - 1 - i[u]);
                keys[q] = ((uint64_t)g[u] << key2_bits) | key2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// 3./4b. bucket boundaries, ranks, singleton removal  (reduce -> scan -> apply)
// ---------------------------------------------------------------------------------
constexpr int kGroupItems = 8;                       // consecutive elements per thread (reduce kernel; one flag word)
constexpr int kGroupTile = kBlock * kGroupItems;     // 2048 elements per workgroup step of the reduce kernel
constexpr int kApplySub = 4;                         // the apply kernel takes 4 such groups per thread
constexpr int kApplyItems = kGroupItems * kApplySub; // 32 consecutive elements per thread
constexpr int kApplyTile = kBlock * kApplyItems;     // 8192 elements per workgroup step; chunks are multiples of it
constexpr uint64_t kSparseApplyDivisor = 8;          // sparse form when at most 1/8 of the elements are kept

// A thread's 8 consecutive keys plus both neighbours, fetched with 16-byte loads (K is a
// workspace array, 256-B aligned, and tile bases are multiples of 2048).  Loading is split
// from flag computation so the NEXT tile's keys can be in flight while this one is scanned.
template <class KeyT>
struct GroupKeys {
    KeyT k[kGroupItems + 2];
};
template <class KeyT>
__device__ __forceinline__ void group_load(const KeyT* __restrict__ K, uint64_t i0, uint64_t m, GroupKeys<KeyT>& g)
{
    constexpr int kVec = 16 / sizeof(KeyT);
    struct alignas(16) Vec { KeyT v[kVec]; };
    if (i0 + kGroupItems <= m) {
#pragma unroll
        for (int q = 0; q < kGroupItems / kVec; q++) {
            Vec v = *reinterpret_cast<const Vec*>(K + i0 + q * kVec);
#pragma unroll
            for (int j = 0; j < kVec; j++) g.k[1 + q * kVec + j] = v.v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kGroupItems; j++) g.k[j + 1] = (i0 + j < m) ? K[i0 + j] : KeyT(0);
    }
    g.k[0] = (i0 > 0 && i0 <= m) ? K[i0 - 1] : KeyT(0);
    g.k[kGroupItems + 1] = (i0 + kGroupItems < m) ? K[i0 + kGroupItems] : KeyT(0);
}
// bit j of head: item j is the first of its bucket; bit j of single: bucket of size one
template <class KeyT>
__device__ __forceinline__ void group_flags(const GroupKeys<KeyT>& g, uint64_t i0, uint64_t m, unsigned& head,
                                            unsigned& single)
{
    head = 0;
    single = 0;
#pragma unroll
    for (int j = 0; j < kGroupItems; j++) {
        const uint64_t i = i0 + j;
        const bool h = (i < m) && ((i == 0) || (g.k[j] != g.k[j + 1]));
        const bool nh = (i + 1 >= m) || (g.k[j + 2] != g.k[j + 1]);
        head |= (h ? 1u : 0u) << j;
        single |= ((h && nh) ? 1u : 0u) << j;
    }
}
__device__ __forceinline__ unsigned valid_mask(uint64_t i0, uint64_t m)
{
    if (i0 + kGroupItems <= m) return (1u << kGroupItems) - 1u;
    return i0 >= m ? 0u : ((1u << (unsigned)(m - i0)) - 1u);
}

#ifndef SFX_REDUCE_DEPTH
#define SFX_REDUCE_DEPTH 2
#endif
// per-workgroup partials: last bucket-head index (+1) in the chunk, #kept, #kept bucket heads
// Fused LCP (sfx_build_sa_lcp_u32_dev): the sorted keys of the initial sort are in registers here, and
// for two neighbours with DIFFERENT keys the common prefix is the number of equal leading symbols of the
// keys -- no text access (97.7 % of the pairs of 100 MB of DNA).  Neighbours with equal keys get
// kLcpPending and are compared on the text once the suffix array is final (k_lcp_pending).
// depth of a bucket from its compressed key (ht_depth); ent == nullptr: fixed-width keys, uniform depth
constexpr unsigned kHtTableWords = 256 + 64;                 // device layout: code table (256), minimum depth + spare (64), fast table
struct HtDepth {
    const uint32_t* ent;        // [256] code table ... [kHtTableWords ..] the 4096-entry fast table (bytes)
    int sigma;
    int count_bits;             // > 0 (context codes, k_ht_keys_ctx): the key's low bits ARE the number of symbols it holds
};
struct LcpFuse {
    uint32_t* lcp;          // nullptr = off
    int pad_bits;           // unused high bits of a key
    uint32_t inv_bits;      // ceil(65536 / bits): x / bits for x < 64 (checked on the host)
    uint32_t pending;       // kLcpBoundFlag | symbols of the key: what neighbours with equal keys are known to share
    const uint32_t* ht;     // compressed keys (round 4): the device tables (HtDepth::ent); the symbols two keys share are the
                            // code words that end inside their common leading bits (ht_common_n), equal keys share all the
                            // symbols the key holds
};

// HT: the keys are compressed (order-preserving prefix code, k_ht_keys) and the LCP is wanted
template <class KeyT, bool HT = false>
__global__ void __launch_bounds__(kBlock)
k_groups_reduce(const KeyT* __restrict__ K, uint64_t m, uint64_t chunk,
                uint32_t* __restrict__ part_head, uint32_t* __restrict__ part_keep,
                uint32_t* __restrict__ part_ghead, uint16_t* __restrict__ flags_out, LcpFuse fuse)
{
    __shared__ uint32_t red[3][kWavesPerBlock];
    __shared__ uint32_t s_t12[HT ? (1 << kHtFastBits) / 2 : 1];
    const unsigned tid = threadIdx.x;
    if (HT) {
        for (unsigned i = threadIdx.x; i < (1u << kHtFastBits) / 2u; i += kBlock) s_t12[i] = fuse.ht[kHtTableWords + i];
        __syncthreads();
    }
    uint64_t begin = (uint64_t)blockIdx.x * chunk;               // chunk: a multiple of kApplyTile elements
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint32_t last_head = 0, keep = 0, ghead = 0;
    // kReduceDepth tiles of keys in flight per thread (the kernel is a pure stream: what bounds
    // it is bytes in flight per CU, not arithmetic)
    constexpr int kReduceDepth = SFX_REDUCE_DEPTH;
    GroupKeys<KeyT> nxt[kReduceDepth];
    uint64_t i0 = begin + (uint64_t)tid * kGroupItems;
#pragma unroll
    for (int u = 0; u < kReduceDepth; u++)
        if (i0 + (uint64_t)u * kGroupTile < end) group_load(K, i0 + (uint64_t)u * kGroupTile, m, nxt[u]);
    for (; i0 < end; i0 += (uint64_t)kReduceDepth * kGroupTile) {
#pragma unroll
        for (int u = 0; u < kReduceDepth; u++) {
            const uint64_t iu = i0 + (uint64_t)u * kGroupTile;
            if (iu >= end) break;
            const GroupKeys<KeyT> cur = nxt[u];
            if (iu + (uint64_t)kReduceDepth * kGroupTile < end) group_load(K, iu + (uint64_t)kReduceDepth * kGroupTile, m, nxt[u]);
            unsigned head, single;
            group_flags(cur, iu, m, head, single);
            // the apply kernel reads these 2 bits per element instead of the keys again
            flags_out[iu / kGroupItems] = (uint16_t)(head | (single << 8));
            if (fuse.lcp) {
                uint32_t l[kGroupItems];
                if (HT) {
                    uint64_t k64[kGroupItems];
                    unsigned common[kGroupItems];
                    // (equal keys are not decoded here: the pair stays pending with the bound every key guarantees --
                    // kHtKeyBits / kHtMaxLen whole code words -- and either a deep round overwrites it with the exact value or
                    // the pending pass starts its comparison one 8-byte step earlier; decoding them too cost 10 ms per 10^9)
                    bool same[kGroupItems];
#pragma unroll
                    for (int j = 0; j < kGroupItems; j++) {
                        const uint64_t x = (uint64_t)(cur.k[j] ^ cur.k[j + 1]);
                        k64[j] = (uint64_t)cur.k[j + 1];
                        same[j] = x == 0;
                        common[j] = x ? (unsigned)__clzll((long long)x) : 0u;
                    }
                    ht_common_n<kGroupItems>(k64, common, reinterpret_cast<const uint16_t*>(s_t12), l);
#pragma unroll
                    for (int j = 0; j < kGroupItems; j++)
                        l[j] = (iu + j == 0) ? 0u : (same[j] ? (kLcpBoundFlag | (uint32_t)(kHtKeyBits / kHtMaxLen)) : l[j]);
                } else {
#pragma unroll
                for (int j = 0; j < kGroupItems; j++) {
                    const uint64_t x = (uint64_t)(cur.k[j] ^ cur.k[j + 1]);
                    const unsigned lz = (unsigned)__clzll((long long)x) - (unsigned)(64 - 8 * (int)sizeof(KeyT)) - (unsigned)fuse.pad_bits;
                    l[j] = (iu + j == 0) ? 0u : (x ? (lz * fuse.inv_bits) >> 16 : fuse.pending);
                }
                }
                if (iu + kGroupItems <= m) {
                    *reinterpret_cast<uint4*>(fuse.lcp + iu) = uint4{l[0], l[1], l[2], l[3]};
                    *reinterpret_cast<uint4*>(fuse.lcp + iu + 4) = uint4{l[4], l[5], l[6], l[7]};
                } else {
#pragma unroll
                    for (int j = 0; j < kGroupItems; j++)
                        if (iu + j < m) fuse.lcp[iu + j] = l[j];
                }
            }
            const unsigned valid = valid_mask(iu, m);
            if (head) last_head = (uint32_t)iu + (32u - (unsigned)__clz((int)head));   // index+1 of the highest head bit
            keep += (uint32_t)__popc(valid & ~single);
            ghead += (uint32_t)__popc(head & ~single);
        }
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
        for (int w = 0; w < kWavesPerBlock; w++) { a = dmax(a, red[0][w]); b += red[1][w]; c += red[2][w]; }
        part_head[blockIdx.x] = a;
        part_keep[blockIdx.x] = b;
        part_ghead[blockIdx.x] = c;
    }
}

// single workgroup: turn the partials into carries (exclusive max / sums); totals[0..1]
__global__ void __launch_bounds__(kBlock)
k_groups_scan(uint32_t* __restrict__ part_head, uint32_t* __restrict__ part_keep,
              uint32_t* __restrict__ part_ghead, unsigned nb, uint32_t* __restrict__ totals,
              uint32_t* __restrict__ part_pairs = nullptr)
{
    // part_pairs (rank rounds): per-chunk counts of the members whose rank changes -> where each chunk's pairs start; totals[2]
    __shared__ uint32_t part[kWavesPerBlock];
    uint32_t c_head = 0, c_keep = 0, c_ghead = 0, c_pairs = 0;
    for (unsigned base = 0; base < nb; base += kBlock) {
        unsigned i = base + threadIdx.x;
        bool valid = i < nb;
        uint32_t vh = valid ? part_head[i] : 0u, vk = valid ? part_keep[i] : 0u,
                 vg = valid ? part_ghead[i] : 0u, th, tk, tg;
        uint32_t eh = block_scan_max_excl(vh, part, th);
        uint32_t ek = block_scan_add_excl(vk, part, tk);
        uint32_t eg = block_scan_add_excl(vg, part, tg);
        if (valid) {
            part_head[i] = dmax(c_head, eh);
            part_keep[i] = c_keep + ek;
            part_ghead[i] = c_ghead + eg;
        }
        if (part_pairs) {
            uint32_t tp;
            const uint32_t ep = block_scan_add_excl(valid ? part_pairs[i] : 0u, part, tp);
            if (valid) part_pairs[i] = c_pairs + ep;
            c_pairs += tp;
        }
        c_head = dmax(c_head, th);
        c_keep += tk;
        c_ghead += tg;
    }
    if (threadIdx.x == 0) { totals[0] = c_keep; totals[1] = c_ghead; totals[2] = c_pairs; }
}

// K,V: sorted keys / suffixes of the m active elements; S: their SA slots in
// ascending order (nullptr = identity).  Writes SA[slot] = suffix (skipped when
// sa_in_place: V IS the SA and slots are the identity -- the last radix pass of the
// initial sort already put every suffix in its slot); if isa:
// ISA[suffix] = slot of its bucket head; and compacts the elements of
// non-singleton buckets into (S_next, V_next, G_next = bucket id = position of the
// bucket's head in the compacted list: unique and increasing along the list; and, if
// R_next, R_next = slot of the bucket head).
// SUB = flag words (groups of 8 elements) per thread.  SUB = 4 is the sparse form: SA in place,
// no rank array, few kept elements -- only the set bits of the keep mask are visited.  SUB = 1
// is the dense form (every element writes its SA slot and / or rank): 8 elements per thread
// keep 4x as many gathers in flight.
// HT / PAIRS: the instantiation carries the end-mask table of compressed keys / the digit counts of the rank pairs in LDS
// (a launch without them keeps 32 KB of staging and nothing else: five workgroups per CU instead of three)
template <class KeyT, int SUB, bool HT = true, bool PAIRS = true>
__global__ void __launch_bounds__(kBlock) SFX_WAVES_PER_EU(SUB == 1 ? 4 : 1, 8)
k_groups_apply(const KeyT* __restrict__ K, const uint32_t* __restrict__ V,
               const uint32_t* __restrict__ S, uint64_t m, uint64_t chunk,
               const uint32_t* __restrict__ part_head, const uint32_t* __restrict__ part_keep,
               const uint32_t* __restrict__ part_ghead, uint32_t* __restrict__ sa,
               uint32_t* __restrict__ isa, uint32_t* __restrict__ S_next,
               uint32_t* __restrict__ V_next, uint32_t* __restrict__ G_next,
               uint32_t* __restrict__ R_next, int sa_in_place, uint64_t* __restrict__ rank_pairs,
               const uint16_t* __restrict__ flags_in, uint32_t* __restrict__ pair_hist, int pair_lo, int pair_nb,
               const uint16_t* __restrict__ Hd, uint16_t* __restrict__ Hd_next, uint32_t hd_floor, HtDepth ht,
               uint32_t* __restrict__ min_depth, const uint8_t* __restrict__ F8 = nullptr,
               const uint32_t* __restrict__ part_pairs = nullptr)
{
    // F8 + part_pairs (rank rounds): a member flagged kRankKept sits in the class that starts where its old bucket
    // started -- same head slot, same rank: no pair, no write.  The pairs of a chunk then start at part_pairs[chunk].
    // ht.ent (initial bucket pass over compressed keys): the depth of a bucket is what its key holds, ht_depth(K);
    // min_depth: smallest depth given to a kept element (the rank rounds' h, should the text rounds give way)
    __shared__ uint32_t s_t12[HT ? (1 << kHtFastBits) / 2 : 1];
    __shared__ uint32_t s_min;
    if (HT && ht.ent)
        for (unsigned i = threadIdx.x; i < (1u << kHtFastBits) / 2u; i += kBlock) s_t12[i] = ht.ent[kHtTableWords + i];
    if (ht.ent || min_depth) {
        if (threadIdx.x == 0) s_min = 0xFFFFFFFFu;
        __syncthreads();
    }
    uint32_t my_min = 0xFFFFFFFFu;
    // Hd_next (deep text rounds): the kept elements carry the depth of their bucket, at least hd_floor
    __shared__ uint32_t part_m[2][kWavesPerBlock], part_a[2][kWavesPerBlock], part_p[2][kWavesPerBlock];
    // pair_hist (with rank_pairs): digit counts of the passes that partition the pairs by suffix index (bits [pair_lo,
    // pair_nb), 8 per pass, at most 3) -- counted here, where the pairs are made, instead of by a pass over them
    constexpr int kPairPasses = 3;
    constexpr int kPhCopies = 2;                                          // (two waves share a copy: 6 KB, so that four workgroups fit a CU)
    __shared__ uint32_t ph[(SUB == 1 && PAIRS) ? kPhCopies : 1][kPairPasses][kRadixDev];
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int pair_passes = pair_hist ? (pair_nb - pair_lo + 7) / 8 : 0;
    // dense form: what a tile keeps (and its (suffix, rank) pairs) leaves through LDS.  A thread's elements are 8
    // consecutive ones, so direct stores put 4 bytes into each of 16 lines per instruction and touch every line eight
    // times -- and a line write costs the CU the same whether it is partial or whole (radix pass).
    __shared__ uint64_t stg_a[SUB == 1 ? kBlock * kGroupItems : 1];     // (slot, suffix) of the kept; then the pairs
    __shared__ uint32_t stg_g[SUB == 1 ? kBlock * kGroupItems : 1];     // bucket id of the kept
    __shared__ uint16_t stg_d[SUB == 1 ? kBlock * kGroupItems : 1];     // ... and the depth of their bucket
    if (SUB == 1 && PAIRS && pair_hist) {
        for (unsigned i = tid; i < (unsigned)(kPhCopies * kPairPasses * kRadixDev); i += kBlock) (&ph[0][0][0])[i] = 0u;
        __syncthreads();
    }
    uint64_t begin = (uint64_t)blockIdx.x * chunk;
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint32_t c_head = part_head[blockIdx.x];     // index+1 of the last head before the chunk
    uint32_t c_keep = part_keep[blockIdx.x];
    uint64_t c_pairs = part_pairs ? (uint64_t)part_pairs[blockIdx.x] : begin;    // (every element a pair: chunk i's start at i * chunk)
    (void)part_ghead;
    unsigned par = 0;
    // head / single bits of the thread's elements = SUB flag words of k_groups_reduce, one load
    // (25 MB per 10^8 elements instead of reading the keys a second time); the next tile's
    // words are in flight while this tile is scanned
    constexpr int kItems = kGroupItems * SUB;
    constexpr int kTile = kBlock * kItems;
    static_assert(SUB == 1 || SUB == 4, "one 2-byte or one 8-byte flag load per thread");
    auto load_flags = [&](uint64_t i0) -> uint64_t {
        if (i0 >= end) return 0ull;
        if (SUB == 4) return *reinterpret_cast<const uint64_t*>(flags_in + i0 / kGroupItems);
        return (uint64_t)flags_in[i0 / kGroupItems];
    };
    uint64_t nf = load_flags(begin + (uint64_t)tid * kItems);
    for (uint64_t tile = begin; tile < end; tile += kTile) {
        const uint64_t i0 = tile + (uint64_t)tid * kItems;
        const uint64_t f = nf;
        if (tile + kTile < end) nf = load_flags(i0 + kTile);
        // 32-bit masks over the thread's elements (groups past `end` were never written: masked off)
        uint32_t head = 0, single = 0, valid = 0;
#pragma unroll
        for (int b = 0; b < SUB; b++) {
            const uint64_t ib = i0 + (uint64_t)b * kGroupItems;
            if (ib < end) {
                const uint32_t fw = (uint32_t)(f >> (16 * b)) & 0xFFFFu;
                head |= (fw & 0xFFu) << (8 * b);
                single |= (fw >> 8) << (8 * b);
                valid |= valid_mask(ib, m) << (8 * b);
            }
        }
        const uint32_t keepm = valid & ~single;
        const uint32_t hmax = head ? (uint32_t)i0 + (32u - (unsigned)__clz((int)head)) : 0u;
        const uint32_t cnt = (uint32_t)__popc(keepm);
        // (dense form, rank rounds) the elements whose rank changes: the valid ones without kRankKept
        uint32_t pairm = SUB == 1 ? (valid & 0xFFu) : 0u;
        if (SUB == 1 && F8 && pairm) {
            const uint64_t f8 = *reinterpret_cast<const uint64_t*>(F8 + i0);       // (i0 is a multiple of 8, F8 padded)
            pairm &= ~(uint32_t)((((f8 >> 3) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
        }
        const uint32_t pcnt = (SUB == 1 && rank_pairs) ? (uint32_t)__popc(pairm) : 0u;
        // one barrier for the scans: exclusive max of hmax, exclusive sums of cnt and pcnt
        uint32_t im = wave_scan_max(hmax), ia = wave_scan_add(cnt), ip = 0;
        if (SUB == 1 && rank_pairs) ip = wave_scan_add(pcnt);
        uint32_t pm = __shfl_up(im, 1u);
        if (lane == 0) pm = 0;
        if (lane == 63) { part_m[par][w] = im; part_a[par][w] = ia; part_p[par][w] = ip; }
        __syncthreads();
        uint32_t bm = 0, ba = 0, bp = 0, tot_m = 0, tot_a = 0, tot_p = 0;
#pragma unroll
        for (unsigned k = 0; k < (unsigned)kWavesPerBlock; k++) {
            const uint32_t qm = part_m[par][k], qa = part_a[par][k], qp = part_p[par][k];
            if (k < w) { bm = dmax(bm, qm); ba += qa; bp += qp; }
            tot_m = dmax(tot_m, qm);
            tot_a += qa;
            tot_p += qp;
        }
        par ^= 1u;
        const uint32_t ec = ba + ia - cnt;
        uint32_t run_head = dmax(c_head, dmax(bm, pm));          // index+1 of the last head before item 0
        const uint32_t run_keep = c_keep + ec;
        if (SUB > 1) {                                   // (the host picks SUB = 4 only with sa_in_place && !isa)
            // only the kept elements have anything to write (a few per cent of a first round):
            // walk the set bits of the keep mask; every quantity is a bit trick on the two masks
            uint32_t k = keepm;
            while (k) {
                const int j = __ffs((int)k) - 1;
                k &= k - 1u;
                const uint64_t i = i0 + (unsigned)j;
                const uint32_t hb = head & ((2u << j) - 1u);     // heads at or before item j
                const uint32_t my_head = hb ? (uint32_t)i0 + 31u - (unsigned)__clz((int)hb) : run_head - 1u;
                const uint32_t pos = run_keep + (uint32_t)__popc(keepm & ((1u << j) - 1u));
                const uint32_t slot = S ? S[i] : (uint32_t)i;
                const uint32_t suffix = V[i];
                if (R_next) R_next[pos] = S ? S[my_head] : my_head;
                S_next[pos] = slot;
                V_next[pos] = suffix;
                G_next[pos] = pos - ((uint32_t)i - my_head);
                if (Hd_next) {
                    uint32_t d = Hd ? dmax<uint32_t>(Hd[i], hd_floor) : hd_floor;
                    if (HT && ht.ent) {
                        unsigned used = 0, cnt = 0;
                        const uint64_t k64 = (uint64_t)K[i];
                        if (ht.count_bits) cnt = (unsigned)(k64 & ((1ull << ht.count_bits) - 1ull));
                        else while (ht_depth_step(k64, used, cnt, reinterpret_cast<const uint16_t*>(s_t12))) {}
                        d = cnt < kHtMaxSym ? cnt : kHtMaxSym;
                    }
                    my_min = dmin(my_min, d);
                    Hd_next[pos] = (uint16_t)d;
                }
            }
        } else {
            const uint64_t ib = i0;
            const unsigned v8 = valid & 0xFFu, h8 = head & 0xFFu, k8 = keepm & 0xFFu;
            uint32_t slot[kGroupItems], suffix[kGroupItems], head_slot[kGroupItems];
            KeyT kk[kGroupItems];                                // (compressed keys: the depths are read off them -- fetched with the rest)
#pragma unroll
            for (int j = 0; j < kGroupItems; j++) kk[j] = KeyT(0);
            if (HT && ht.ent && k8) {
                if (v8 == 0xFFu) {
                    struct alignas(16) KV { KeyT v[16 / sizeof(KeyT)]; };
#pragma unroll
                    for (int q = 0; q < (int)(kGroupItems * sizeof(KeyT) / 16); q++) {
                        const KV x = *reinterpret_cast<const KV*>(K + ib + q * (16 / sizeof(KeyT)));
#pragma unroll
                        for (int u = 0; u < (int)(16 / sizeof(KeyT)); u++) kk[q * (16 / sizeof(KeyT)) + u] = x.v[u];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kGroupItems; j++) kk[j] = ((v8 >> j) & 1u) ? K[ib + j] : KeyT(0);
                }
            }
            uint32_t dep[kGroupItems];
            if (HT && ht.ent && k8) {                            // the depths of the kept elements' buckets, off their keys
                uint64_t k64[kGroupItems];
#pragma unroll
                for (int j = 0; j < kGroupItems; j++) k64[j] = (uint64_t)kk[j];
                // (a bucket's members share the key: one decode per run of equal keys)
                unsigned want = 0;
#pragma unroll
                for (int j = 0; j < kGroupItems; j++)
                    if (((k8 >> j) & 1u) && (j == 0 || !((k8 >> (j - 1)) & 1u) || k64[j] != k64[j - 1])) want |= 1u << j;
                // (one key at a time: a thread wants 1.7 of its 8 on English-like text, and the kernel is VALU-bound -- 92 % of the
                // SIMD cycles on the 10^9-element pass of config 3 -- so decoding all eight side by side, wanted or not, until the
                // slowest of a wave's 512 is done cost more than walking the set bits)
#pragma unroll
                for (int j = 0; j < kGroupItems; j++) dep[j] = 0;
                unsigned todo = want;
                while (todo) {
                    const unsigned j = (unsigned)__ffs((int)todo) - 1u;
                    todo &= todo - 1u;
                    uint64_t kx = k64[0];
#pragma unroll
                    for (int q = 1; q < kGroupItems; q++) kx = j == (unsigned)q ? k64[q] : kx;
                    const uint64_t one[1] = {kx};
                    const unsigned all[1] = {(unsigned)kHtKeyBits};
                    uint32_t d1[1];
                    if (ht.count_bits) d1[0] = (uint32_t)(kx & ((1ull << ht.count_bits) - 1ull));
                    else ht_common_n<1>(one, all, reinterpret_cast<const uint16_t*>(s_t12), d1);
#pragma unroll
                    for (int q = 0; q < kGroupItems; q++) dep[q] = j == (unsigned)q ? d1[0] : dep[q];
                }
#pragma unroll
                for (int j = 1; j < kGroupItems; j++)
                    if (((k8 >> j) & 1u) && !((want >> j) & 1u)) dep[j] = dep[j - 1];
            } else {
#pragma unroll
                for (int j = 0; j < kGroupItems; j++)
                    dep[j] = (Hd && ((k8 >> j) & 1u)) ? dmax<uint32_t>(Hd[ib + j], hd_floor) : hd_floor;
            }
            unsigned local_keep = ec;                            // the thread's first kept element, counted from the tile's
            if (k8 || ((isa || sa_in_place != 1) && v8)) {       // (all-singleton groups have nothing to write in place)
                // loads first (all in flight together, two 16-byte loads per array where the group is whole), stores after
                const bool whole = v8 == 0xFFu && ((reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(S)) & 15u) == 0;
                if (whole && S) {
                    const uint4 a = *reinterpret_cast<const uint4*>(S + ib), c = *reinterpret_cast<const uint4*>(S + ib + 4);
                    slot[0] = a.x; slot[1] = a.y; slot[2] = a.z; slot[3] = a.w; slot[4] = c.x; slot[5] = c.y; slot[6] = c.z; slot[7] = c.w;
                } else {
#pragma unroll
                    for (int j = 0; j < kGroupItems; j++) slot[j] = (((v8 >> j) & 1u) && S) ? S[ib + j] : (uint32_t)(ib + j);
                }
                if (whole && (sa_in_place != 1 || isa || k8 == 0xFFu)) {
                    const uint4 a = *reinterpret_cast<const uint4*>(V + ib), c = *reinterpret_cast<const uint4*>(V + ib + 4);
                    suffix[0] = a.x; suffix[1] = a.y; suffix[2] = a.z; suffix[3] = a.w; suffix[4] = c.x; suffix[5] = c.y; suffix[6] = c.z; suffix[7] = c.w;
                } else {
#pragma unroll
                    for (int j = 0; j < kGroupItems; j++)
                        suffix[j] = (((v8 >> j) & 1u) && (sa_in_place != 1 || ((k8 >> j) & 1u) || isa)) ? V[ib + j] : 0u;
                }
                uint32_t rh = run_head;
#pragma unroll
                for (int j = 0; j < kGroupItems; j++) {
                    const bool v = (v8 >> j) & 1u, keep = (k8 >> j) & 1u;
                    if ((h8 >> j) & 1u) rh = (uint32_t)(ib + j) + 1u;
                    const uint32_t my_head = rh - 1u;
                    head_slot[j] = (v && S && (isa || (keep && R_next))) ? S[my_head] : my_head;
                }
            }
            if (rank_pairs) {                                    // the tile's pairs, in stream order
                unsigned at = bp + ip - pcnt;
#pragma unroll
                for (int j = 0; j < kGroupItems; j++) {
                    if ((pairm >> j) & 1u) {
                        stg_a[at++] = ((uint64_t)suffix[j] << 32) | (uint64_t)head_slot[j];
                        if (PAIRS && pair_hist) {
#pragma unroll
                            for (int p = 0; p < kPairPasses; p++) {
                                const int sh = pair_lo + 8 * p, nbits = pair_nb - sh < 8 ? pair_nb - sh : 8;
                                if (p < pair_passes) atomicAdd(&ph[w % kPhCopies][p][(suffix[j] >> sh) & ((1u << nbits) - 1u)], 1u);
                            }
                        }
                    }
                }
                __syncthreads();
                for (unsigned k = tid; k < tot_p; k += kBlock) rank_pairs[c_pairs + k] = stg_a[k];
                __syncthreads();
            }
            {
                uint32_t rh = run_head;
#pragma unroll
                for (int j = 0; j < kGroupItems; j++) {
                    if ((h8 >> j) & 1u) rh = (uint32_t)(ib + j) + 1u;
                    if ((v8 >> j) & 1u) {
                        const bool keep = (k8 >> j) & 1u;
                        // sa_in_place: 0 = every element goes to its slot; 1 = V is the SA; 2 = only the elements that
                        // resolve now (the members of unresolved buckets would be rewritten every round)
                        if (sa_in_place == 0 || (sa_in_place == 2 && !keep)) sa[slot[j]] = suffix[j];
                        if (isa && !rank_pairs && ((pairm >> j) & 1u)) isa[suffix[j]] = head_slot[j];
                        if (keep) {
                            const uint32_t back = (uint32_t)(ib + j) - (rh - 1u);   // distance to the bucket head (all kept in between)
                            const uint32_t gpos = c_keep + local_keep;
                            if (R_next) R_next[gpos] = head_slot[j];
                            stg_a[local_keep] = ((uint64_t)slot[j] << 32) | (uint64_t)suffix[j];
                            stg_g[local_keep] = gpos - back;
                            stg_d[local_keep] = (uint16_t)dep[j];
                            my_min = dmin(my_min, dep[j]);
                            local_keep++;
                        }
                    }
                }
            }
            __syncthreads();
            for (unsigned k = tid; k < tot_a; k += kBlock) {
                const uint64_t e = stg_a[k];
                S_next[c_keep + k] = (uint32_t)(e >> 32);
                V_next[c_keep + k] = (uint32_t)e;
                G_next[c_keep + k] = stg_g[k];                   // bucket id = position of its head in the new list
                if (Hd_next) Hd_next[c_keep + k] = stg_d[k];
            }
            __syncthreads();
        }
        c_head = dmax(c_head, tot_m);
        c_keep += tot_a;
        c_pairs += tot_p;
    }
    if (min_depth) {
        for (int d = 32; d >= 1; d >>= 1) my_min = dmin(my_min, (uint32_t)__shfl_xor(my_min, d));
        if (lane == 0) atomicMin(&s_min, my_min);
        __syncthreads();
        if (tid == 0 && s_min != 0xFFFFFFFFu) atomicMin(min_depth, s_min);
    }
    if (SUB == 1 && PAIRS && pair_hist) {
        __syncthreads();
        for (int p = 0; p < pair_passes; p++) {
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < kPhCopies; k++) c += ph[k][p][tid];
            pair_hist[((uint64_t)p * kRadixDev + tid) * gridDim.x + blockIdx.x] = c;
        }
    }
}

// ---------------------------------------------------------------------------------
// 4c. direct ordering of small buckets
// ---------------------------------------------------------------------------------
// After the initial sort of low-entropy-free text (DNA) almost every unresolved bucket
// holds 2-3 suffixes.  Sending those through another composite-key radix sort costs seven
// passes; comparing the handful of suffixes directly on the packed text costs a few word
// loads.  One thread per active element: find the bucket's extent in the active list
// (<= kSmallCap members, else the bucket is left to the radix path), compare against every
// other member from offset h on (all members share their first h symbols), and take
// position = #smaller members (+ #undecided members before it).  A comparison is
// undecided when kSmallDepthWords packed words are equal; such members stay unresolved,
// in place, and go on to the next radix round together.
constexpr int kSmallCap = 32;
constexpr int kSmallDepthWords = 16;

// -1: suffix a < suffix b, +1: a > b, 0: equal for kSmallDepthWords words beyond offset h.
// "Shorter sorts first" (the reference's virtual sentinel, :422-425): when one suffix ends
// inside the window only the symbols both still have are compared, then the shorter wins.
__device__ __forceinline__ int direct_compare(const PackedText& t, uint64_t a, uint64_t b, uint64_t h)
{
    for (int wd = 0; wd < kSmallDepthWords; wd++) {
        const int64_t la = (int64_t)t.n - (int64_t)(a + h), lb = (int64_t)t.n - (int64_t)(b + h);
        const int64_t lim = la < lb ? la : lb;
        if (lim <= 0) return la < lb ? -1 : 1;
        uint32_t wa = packed_key32(t, a + h), wb = packed_key32(t, b + h);
        if (lim < (int64_t)t.spw) {
            const unsigned sh = (unsigned)(t.spw - (int)lim) * (unsigned)t.bits;
            wa >>= sh;
            wb >>= sh;
            if (wa != wb) return wa < wb ? -1 : 1;
            return la < lb ? -1 : 1;
        }
        if (wa != wb) return wa < wb ? -1 : 1;
        h += (uint64_t)t.spw;
    }
    return 0;
}

// V/S/G: the active list (suffix, SA slot, bucket id per position; buckets contiguous,
// id = position of the bucket's head).  V2/G2: suffixes re-ordered inside every small
// bucket and the ids of what is left of it -- members that tie with each other form a new
// bucket (id = position of its first member) which must not share an id, or a rank, with
// the rest of the old bucket: a suffix resolved BETWEEN two tie classes would otherwise
// outrank members that are in fact larger.  flag[p] = 1 where position p is still
// unresolved.  sa (and the rank array, if in use) are updated for every member of a small
// bucket.
__global__ void __launch_bounds__(kBlock)
k_small_groups(const uint32_t* __restrict__ V, const uint32_t* __restrict__ S, const uint32_t* __restrict__ G,
               uint64_t m, PackedText t, uint64_t h, uint32_t* __restrict__ sa, uint32_t* __restrict__ isa,
               uint32_t* __restrict__ V2, uint32_t* __restrict__ G2, uint32_t* __restrict__ flag, uint32_t* __restrict__ lcp)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < m; q += stride) {
        const uint32_t g = G[q];
        // fused LCP: whatever order the members end up in, neighbours inside the bucket share >= h symbols; the
        // exact value is found on the text at the end (ties that go on to later rounds are overwritten there)
        if (lcp && q != (uint64_t)g) lcp[S[q]] = kLcpBoundFlag | (uint32_t)h;
        const uint64_t lo = g;                                   // bucket id = position of its head
        const uint32_t my = V[q];
        // buckets of exactly two (the common case): the head orders the pair alone -- one
        // comparison and half the loads; the second member has nothing to do
        if (q == lo + 1 && (q + 1 >= m || G[q + 1] != g)) continue;
        if (q == lo && q + 1 < m && (q + 2 >= m || G[q + 2] != g)) {
            const uint32_t other = V[q + 1];
            const int c = direct_compare(t, (uint64_t)my, (uint64_t)other, h);
            const uint32_t first = c > 0 ? other : my, second = c > 0 ? my : other;
            const uint32_t s0 = S[lo], s1 = S[lo + 1];
            V2[lo] = first;
            V2[lo + 1] = second;
            G2[lo] = (uint32_t)lo;
            G2[lo + 1] = (uint32_t)(c == 0 ? lo : lo + 1);
            flag[lo] = flag[lo + 1] = (c == 0) ? 1u : 0u;
            sa[s0] = first;
            sa[s1] = second;
            if (isa) { isa[first] = s0; isa[second] = (c == 0) ? s0 : s1; }
            continue;
        }
        uint64_t hi = q;
        while (hi + 1 < m && hi + 1 - lo < (uint64_t)kSmallCap && G[hi + 1] == g) hi++;
        if (q - lo >= (uint64_t)kSmallCap || (hi + 1 < m && G[hi + 1] == g)) {   // large bucket: untouched
            V2[q] = my;
            G2[q] = g;
            flag[q] = 1u;
            continue;
        }
        uint32_t smaller = 0, ties = 0, ties_before = 0;
        for (uint64_t f = lo; f <= hi; f++) {
            if (f == q) continue;
            const int c = direct_compare(t, (uint64_t)my, (uint64_t)V[f], h);
            if (c > 0) smaller++;
            else if (c == 0) { ties++; if (f < q) ties_before++; }
        }
        const uint64_t cls = lo + smaller;                       // first position of my tie class
        const uint64_t pos = cls + ties_before;
        V2[pos] = my;
        G2[pos] = (uint32_t)cls;
        flag[pos] = ties ? 1u : 0u;
        sa[S[pos]] = my;                                         // keeps sa a permutation even where unresolved
        if (isa) isa[my] = S[cls];                               // final slot, or the tie class's head slot
    }
}

// ---- the tie mask of the hybrid initial sort (TieRecords, sfx_host.hpp; round 6) --------------------------------------------
// The LDS sort of a sub-bucket knows which of its elements share their whole key with a neighbour; k_groups_reduce /
// k_groups_apply found the same out by reading the sorted keys and suffixes again (12 bytes per suffix for the 2.3 % of them
// that stay tied on uniform DNA).  One bit per slot of the array: tmask (the element shares its key with a neighbour).  Equal
// keys are neighbours, so a STRETCH of consecutive tied slots is one run of equal keys or several adjacent ones, and the array
// holds the suffixes.
//
// k_tie_direct orders every stretch of up to kTieRunMax slots on the text from the suffixes' first symbol on (so it need not
// know where one run of a stretch ends and the next begins): a wave reads 64 x 2 mask words, every lane lists the stretches that
// start in its 64 slots, the wave's list is worked off 64 stretches at a time -- the suffixes of one stretch in the registers
// of one lane, insertion sort with direct_compare64 (as k_small_groups orders the small buckets of an active list, two key words
// per step) -- and written back in order.  A stretch it cannot finish -- longer, or two members equal for 8 more key pairs --
// stays as it is and is marked in lmask.  lines: counter lines [0] = tied slots, [1] = stretches, [2] = members of unfinished
// stretches.  Uniform DNA leaves none: the build is done.  Otherwise k_tie_heads marks the first slot of every run of the
// unfinished stretches (a key compare with the slot before) and they become the first active list (k_tie_list).
constexpr uint32_t kTieRunMax = 8;
constexpr unsigned kTieSlots = 1024;                              // counter lines of k_tie_direct (4 words each: in deep_slots)
constexpr int kTieBatch = 2;                                      // stretches a lane lists per round of its wave
__device__ __forceinline__ bool tie_bit(const uint32_t* __restrict__ mask, uint64_t r) { return (mask[r >> 5] >> (r & 31u)) & 1u; }
// bits [r0, r0 + len) of a mask that other lanes mark too
__device__ __forceinline__ void tie_mark(uint32_t* __restrict__ mask, uint64_t r0, uint32_t len)
{
    uint64_t r = r0;
    const uint64_t end = r0 + len;
    while (r < end) {
        const uint32_t bit = (uint32_t)(r & 31u);
        const uint64_t room = 32u - bit, take = end - r < room ? end - r : room;
        atomicOr(&mask[r >> 5], (take == 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << bit);
        r += take;
    }
}

// -1: suffix a < suffix b, +1: a > b, 0: equal for `steps` pairs of packed words beyond offset h (direct_compare, two words a step)
__device__ __forceinline__ int direct_compare64(const PackedText& t, uint64_t a, uint64_t b, uint64_t h, int steps)
{
    const int64_t sym = 2 * (int64_t)t.spw;
    for (int s = 0; s < steps; s++) {
        const int64_t la = (int64_t)t.n - (int64_t)(a + h), lb = (int64_t)t.n - (int64_t)(b + h);
        const int64_t lim = la < lb ? la : lb;
        if (lim <= 0) return la < lb ? -1 : 1;
        uint64_t wa = packed_key64(t, a + h), wb = packed_key64(t, b + h);
        if (lim < sym) {
            const unsigned sh = (unsigned)(sym - lim) * (unsigned)t.bits;
            wa >>= sh;
            wb >>= sh;
            if (wa != wb) return wa < wb ? -1 : 1;
            return la < lb ? -1 : 1;
        }
        if (wa != wb) return wa < wb ? -1 : 1;
        h += (uint64_t)sym;
    }
    return 0;
}

__global__ void __launch_bounds__(kBlock)
k_tie_direct(const uint32_t* __restrict__ tmask, uint64_t m, PackedText t, uint32_t* __restrict__ sa, uint32_t* __restrict__ lines,
             uint32_t run_max, uint32_t* __restrict__ lmask)
{
    __shared__ uint32_t s_ent[kWavesPerBlock][kWave * kTieBatch];
    const unsigned lane = lane_id(), w = wave_id();
    // The kernel is a chain of dependent misses (mask word -> the stretch's slots -> the text -> the slots again) and nothing
    // else: what counts is how few links a wave works off one after the other.  A lane owns 64 slots (0.7 stretches on uniform
    // DNA): most waves list their stretches in one round and order them in one pass.
    const uint64_t npairs = ((m + 31) / 32 + 1) / 2;              // (the mask is padded with zero words beyond that)
    const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
    const uint2* const tmask2 = reinterpret_cast<const uint2*>(tmask);
    uint32_t n_tied = 0, n_runs = 0, n_left = 0;                  // (per lane; summed over the wave at the end)
    for (uint64_t qbase = ((uint64_t)blockIdx.x * kWavesPerBlock + w) * kWave; qbase < npairs; qbase += nwaves * kWave) {
        const uint64_t q = qbase + lane;
        uint2 T = {0u, 0u};
        if (q < npairs) T = tmask2[q];
        // the slot before this lane's 64 and the 32 behind them: the neighbouring lanes' words (the wave's ends: one more load)
        uint32_t prev_top = (uint32_t)__shfl_up(T.y, 1) >> 31, next_w = (uint32_t)__shfl_down(T.x, 1);
        if (lane == 0) prev_top = qbase ? tmask[qbase * 2 - 1] >> 31 : 0u;
        if (lane == kWave - 1) next_w = qbase + kWave <= npairs ? tmask[(qbase + kWave) * 2] : 0u;
        const uint64_t lo = (uint64_t)T.x | ((uint64_t)T.y << 32);
        n_tied += (uint32_t)__popcll(lo);
        // bits [i, i + 64) of the lane's 96 (valid for the 33 bits a stretch can need)
        auto window = [&](unsigned i) -> uint64_t { return i ? (lo >> i) | ((uint64_t)next_w << (64u - i)) : lo; };
        // stretches that start here: a tied slot behind an untied one
        uint64_t slo = lo & ~((lo << 1) | (uint64_t)prev_top);
        while (__ballot(slo != 0ull) != 0ull) {
            uint32_t ent[kTieBatch];
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < kTieBatch; k++) {
                ent[k] = 0u;
                if (slo != 0ull) {
                    const unsigned i = (unsigned)__ffsll((unsigned long long)slo) - 1u;
                    slo &= slo - 1ull;
                    const uint64_t win = window(i);
                    uint32_t len = (~win) ? (uint32_t)__ffsll((unsigned long long)~win) - 1u : 64u;
                    const uint64_t r0 = q * 64 + i;
                    n_runs++;
                    if (len > run_max) {
                        uint64_t r = r0 + (len < 33u ? len : 33u);                // (the window holds 33 bits for sure)
                        if (len >= 33u) { len = 33u; while (tie_bit(tmask, r)) { len++; r++; } }
                        n_left += len;
                        tie_mark(lmask, r0, len);
                    } else {
                        ent[mine++] = (uint32_t)r0 | (len << 28);              // (r0 < m <= 2^28: the hybrid route's limit)
                    }
                }
            }
            const uint32_t incl = wave_scan_add(mine);
            const uint32_t total = (uint32_t)__shfl(incl, kWave - 1);
#pragma unroll
            for (int k = 0; k < kTieBatch; k++)
                if ((uint32_t)k < mine) s_ent[w][incl - mine + (uint32_t)k] = ent[k];
            wave_sync();
            for (uint32_t e = lane; e < total; e += kWave) {
                const uint32_t en = s_ent[w][e], len = en >> 28;
                uint32_t* const slot = sa + (en & 0x0FFFFFFFu);
                uint32_t suf[kTieRunMax], slot_was[kTieRunMax];
#pragma unroll
                for (uint32_t k = 0; k < kTieRunMax; k++) suf[k] = slot_was[k] = k < len ? slot[k] : 0u;
                // insertion sort on the text; a pair that stays equal leaves the stretch as it was
                bool undecided = false;
#pragma unroll
                for (uint32_t k = 1; k < kTieRunMax; k++) {
                    if (k < len && !undecided) {
                        const uint32_t x = suf[k];
                        uint32_t pos = k;
#pragma unroll
                        for (uint32_t j = kTieRunMax - 1; j >= 1; j--) {
                            if (j <= k && pos == j && !undecided) {
                                const int cmp = direct_compare64(t, (uint64_t)x, (uint64_t)suf[j - 1], 0, kSmallDepthWords / 2);
                                if (cmp == 0) undecided = true;
                                else if (cmp < 0) { suf[j] = suf[j - 1]; pos = j - 1; }
                            }
                        }
                        if (!undecided) {
#pragma unroll
                            for (uint32_t j = 0; j < kTieRunMax; j++)
                                if (j == pos) suf[j] = x;
                        }
                    }
                }
                if (undecided) { n_left += len; tie_mark(lmask, (uint64_t)(en & 0x0FFFFFFFu), len); continue; }
                // (only what moved: a store into the array is a partial block at the memory side, and half of the pairs lie in
                // order already)
#pragma unroll
                for (uint32_t k = 0; k < kTieRunMax; k++)
                    if (k < len && slot_was[k] != suf[k]) slot[k] = suf[k];
            }
            wave_sync();                                                        // (the list is read to the end)
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        n_tied += (uint32_t)__shfl_xor(n_tied, d);
        n_runs += (uint32_t)__shfl_xor(n_runs, d);
        n_left += (uint32_t)__shfl_xor(n_left, d);
    }
    // (one counter line per wave class, never one address for all waves: tens of thousands of atomics on one word are a queue at
    // one L2 channel -- 0.65 of this kernel's 0.78 ms when it was written that way; k_tie_totals sums the lines)
    if (lane == 0) {
        uint32_t* const line = lines + (size_t)((blockIdx.x * (unsigned)kWavesPerBlock + w) % kTieSlots) * 4u;
        if (n_tied) atomicAdd(&line[0], n_tied);
        if (n_runs) atomicAdd(&line[1], n_runs);
        if (n_left) atomicAdd(&line[2], n_left);
    }
}
__global__ void __launch_bounds__(kBlock)
k_tie_totals(const uint32_t* __restrict__ lines, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t part[3][kWavesPerBlock];
    uint32_t a = 0, b = 0, c = 0;
    for (unsigned i = threadIdx.x; i < kTieSlots; i += kBlock) { a += lines[i * 4u]; b += lines[i * 4u + 1u]; c += lines[i * 4u + 2u]; }
    for (int d = 32; d >= 1; d >>= 1) { a += (uint32_t)__shfl_xor(a, d); b += (uint32_t)__shfl_xor(b, d); c += (uint32_t)__shfl_xor(c, d); }
    if (lane_id() == 0) { part[0][wave_id()] = a; part[1][wave_id()] = b; part[2][wave_id()] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t x = 0;
        for (int w = 0; w < kWavesPerBlock; w++) x += part[threadIdx.x][w];
        totals[threadIdx.x] = x;
    }
}
// k_tie_heads (the leftover path): hmask bit r = slot r is tied and the first of its run -- the slot before is untied or holds
// another key (the key the sort used: the suffix's first kbits, zero-padded past the end).  One lane per mask word; lines[0] +=
// runs.
__global__ void __launch_bounds__(kBlock)
k_tie_heads(const uint32_t* __restrict__ tmask, uint64_t m, PackedText t, const uint32_t* __restrict__ sa, uint32_t* __restrict__ hmask,
            uint32_t* __restrict__ lines)
{
    const uint64_t nwords = (m + 31) / 32, stride = (uint64_t)gridDim.x * kBlock;
    uint32_t n_runs = 0;
    for (uint64_t wi = (uint64_t)blockIdx.x * kBlock + threadIdx.x; wi < nwords; wi += stride) {
        const uint32_t tw = tmask[wi];
        uint32_t hw = 0;
        if (tw != 0u) {
            const uint32_t prev_top = wi ? tmask[wi - 1] >> 31 : 0u;
            uint32_t starts = tw & ~((tw << 1) | prev_top), inner = tw & ~starts;     // (a tied slot behind a tied one: compare the keys)
            hw = starts;
            while (inner != 0u) {
                const uint32_t bit = (uint32_t)__ffs((int)inner) - 1u;
                inner &= inner - 1u;
                const uint64_t r = wi * 32 + bit;
                if (packed_key32(t, (uint64_t)sa[r]) != packed_key32(t, (uint64_t)sa[r - 1])) hw |= 1u << bit;
            }
            n_runs += (uint32_t)__popc(hw);
        }
        hmask[wi] = hw;
    }
    for (int d = 32; d >= 1; d >>= 1) n_runs += (uint32_t)__shfl_xor(n_runs, d);
    if (lane_id() == 0 && n_runs)
        atomicAdd(&lines[(size_t)((blockIdx.x * (unsigned)kWavesPerBlock + wave_id()) % kTieSlots) * 4u], n_runs);
}
// k_tie_list: the tied slots in ascending order are the first active list.  Two phases over chunks of mask words, as
// k_flag_compact: phase 0 counts the tied slots of every chunk (block_counts, then k_scan_block_counts); phase 1 writes, for the
// tied slot r at list position L: suffix = the array's entry, slot = r, bucket id = list position of the head of its run (every
// slot of a run is tied: the distance to the head is the same in the list as in the array).
__global__ void __launch_bounds__(kBlock)
k_tie_list(const uint32_t* __restrict__ tmask, const uint32_t* __restrict__ hmask, uint64_t nwords, uint64_t chunk, int phase,
           uint32_t* __restrict__ block_counts, const uint32_t* __restrict__ sa, uint32_t* __restrict__ S, uint32_t* __restrict__ V,
           uint32_t* __restrict__ G)
{
    __shared__ uint32_t part[kWavesPerBlock];
    const unsigned tid = threadIdx.x;
    const uint64_t wbegin = (uint64_t)blockIdx.x * chunk;
    uint64_t wend = wbegin + chunk;
    if (wend > nwords) wend = nwords;
    uint64_t running = (phase == 1) ? (uint64_t)block_counts[blockIdx.x] : 0ull;
    for (uint64_t base = wbegin; base < wend; base += kBlock) {
        const uint64_t wi = base + tid;
        const uint32_t tw = wi < wend ? tmask[wi] : 0u;
        uint32_t total;
        const uint32_t ex = block_scan_add_excl<uint32_t>((uint32_t)__popc(tw), part, total);
        if (phase == 1 && tw != 0u) {
            uint32_t L = (uint32_t)(running + ex), bits = tw;
            while (bits != 0u) {
                const uint32_t bit = (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                const uint64_t r = wi * 32 + bit;
                uint64_t hr = r;
                while (!tie_bit(hmask, hr)) hr--;                               // (a run starts at its head: there is a bit at or below)
                V[L] = sa[r];
                S[L] = (uint32_t)r;
                G[L] = L - (uint32_t)(r - hr);
                L++;
            }
        }
        running += total;
    }
    if (phase == 0 && tid == 0) block_counts[blockIdx.x] = (uint32_t)running;
}

// stream compaction of the positions with flag != 0 (order kept): two-phase, per-workgroup
// counts scanned by k_scan_block_counts
__global__ void __launch_bounds__(kBlock)
k_flag_compact(const uint32_t* __restrict__ flag, const uint32_t* __restrict__ S, const uint32_t* __restrict__ V2,
               const uint32_t* __restrict__ G, uint64_t m, uint64_t chunk, int phase,
               uint32_t* __restrict__ block_counts, uint32_t* __restrict__ S_next, uint32_t* __restrict__ V_next,
               uint32_t* __restrict__ G_next, const uint16_t* __restrict__ Hd, uint16_t* __restrict__ Hd_next)
{
    __shared__ uint32_t part[kWavesPerBlock];
    const unsigned tid = threadIdx.x;
    uint64_t begin = (uint64_t)blockIdx.x * chunk;
    uint64_t end = begin + chunk;
    if (end > m) end = m;
    uint64_t running = (phase == 1) ? (uint64_t)block_counts[blockIdx.x] : 0ull;
    for (uint64_t base = begin; base < end; base += kBlock) {
        const uint64_t p = base + tid;
        const bool keep = p < end && flag[p] != 0u;
        uint32_t total;
        const uint32_t ex = block_scan_add_excl<uint32_t>(keep ? 1u : 0u, part, total);
        if (phase == 1 && keep) {
            // a class that stays unresolved stays whole and contiguous: the distance to its head is
            // kept, so the id can again be the list position of the head
            S_next[running + ex] = S[p];
            V_next[running + ex] = V2[p];
            G_next[running + ex] = (uint32_t)(running + ex) - ((uint32_t)p - G[p]);
            if (Hd_next) Hd_next[running + ex] = Hd[p];
        }
        running += total;
    }
    if (phase == 0 && tid == 0) block_counts[blockIdx.x] = (uint32_t)running;
}

// (Re)build the rank array when refinement switches from text symbols to ranks: rank of a
// suffix = its slot, except the members of still-unresolved buckets, which share their head's
// slot.  H[r] = r for every slot, then H[slot] = head slot for the active list (slots ascend
// along the list: near-sequential writes); then ISA[SA[r]] = H[r] -- ONE n-element scatter, sent
// through the partitioned scatter for large texts.
__global__ void __launch_bounds__(kBlock)
k_iota(uint32_t* __restrict__ out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride) out[r] = (uint32_t)r;
}
__global__ void __launch_bounds__(kBlock)
k_scatter_by_slot(const uint32_t* __restrict__ suf, const uint32_t* __restrict__ slot, uint64_t m, uint32_t* __restrict__ sa)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < m; q += stride) sa[slot[q]] = suf[q];
}
__global__ void __launch_bounds__(kBlock)
k_head_slots(const uint32_t* __restrict__ slot, const uint32_t* __restrict__ gid, uint64_t m, uint32_t* __restrict__ H)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < m; q += stride) H[slot[q]] = slot[gid[q]];
}
__global__ void __launch_bounds__(kBlock)
k_isa_from_sa(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ H, uint64_t n, uint32_t* __restrict__ isa)
{
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t s0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s0 < n; s0 += U * stride) {
        uint32_t v[U], r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            v[u] = (s0 + u * stride < n) ? sa[s0 + u * stride] : 0u;
            r[u] = (s0 + u * stride < n) ? H[s0 + u * stride] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (s0 + u * stride < n) isa[v[u]] = r[u];
    }
}
// the same as (suffix << 32 | rank) pairs in slot order, for scatter_pairs_u32
__global__ void __launch_bounds__(kBlock)
k_rank_pairs(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ H, uint64_t n, uint64_t* __restrict__ pairs)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += stride)
        pairs[r] = ((uint64_t)sa[r] << 32) | (uint64_t)H[r];
}

// ---------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------
struct Alphabet {
    unsigned sigma;
    int bits;           // bits per symbol
    int spw;            // symbols per packed word = floor(32/bits)
    int kbits;          // bits * spw (<= 32)
    double h0;          // order-0 entropy in bits per symbol when the bins are the byte COUNTS of the text, else < 0
};

// n: length of the text the bins describe (bins that sum to it are counts, not presence flags)
static Alphabet make_alphabet(const unsigned long long* bins, uint64_t n = 0)
{
    Alphabet a;
    a.sigma = 0;
    a.h0 = -1.0;
    unsigned long long total = 0;
    for (int c = 0; c < 256; c++) total += bins[c];
    if (n > 256 && total == n) {
        a.h0 = 0.0;
        for (int c = 0; c < 256; c++)
            if (bins[c]) { const double q = (double)bins[c] / (double)n; a.h0 -= q * log2(q); }
    }
    for (int c = 0; c < 256; c++) if (bins[c]) a.sigma++;
    a.bits = bits_for(a.sigma > 1 ? a.sigma - 1 : 1);
    a.spw = 32 / a.bits;
    a.kbits = a.bits * a.spw;
    return a;
}

// 32-bit keys (spw symbols) when they are expected to separate at least half of the
// suffixes of a random text over this alphabet (spw*floor(log2 sigma) >= log2 n + 1):
// 4 cheap passes plus a refinement round on the rest move fewer bytes than 8 passes
// over 12-byte elements.  Else 64-bit keys (2*spw symbols).  Depends only on
// (alphabet, n): identical on every rank.
static void choose_key(const Alphabet& a, uint64_t n, int* key_bits, int* cpk)
{
    int spw = a.spw;
    int l2 = bits_for(a.sigma) - 1;                 // floor(log2 sigma), sigma >= 1
    if (l2 < 1) l2 = 1;
    // what a symbol tells apart: log2 sigma on evenly used symbols; the order-0 entropy when the counts are known and the
    // alphabet is large enough to be used unevenly (natural-language text: 4.2 bits of 7 -- four symbols of a 32-bit
    // key leave nearly every suffix of a megabyte tied: 8 MB of English-like text 3.1 ms with 32-bit keys, 1.9 ms with
    // 64-bit compressed keys; 4 MB 2.25 / 1.49; 1 MB 1.18 / 0.99).  Small alphabets (DNA) keep the first rule.
    double per = (double)l2;
    if (a.h0 > 0.0 && a.sigma > 16 && a.h0 < per) per = a.h0;
    // SFX_FORCE_KEY64=1 is a test hook (the 64-bit-key path on inputs small enough for the emulator)
    static const bool force64 = [] { const char* e = dev_env("SFX_FORCE_KEY64"); return e && atoi(e) != 0; }();
    if (!force64 && (double)spw * per >= (double)(bits_for(n) + 1)) { *key_bits = 32; *cpk = spw; }
    else { *key_bits = 64; *cpk = 2 * spw; }
}

static uint64_t packed_words(uint64_t n, const Alphabet* a)
{
    uint64_t spw = a ? (uint64_t)a->spw : 4;        // worst case: 4 symbols per word
    return (n + spw - 1) / spw + 3;
}

// ---- order-preserving code of the dense symbols (k_ht_keys, sfx_radix.hip) -------------------------
// Optimal alphabetic binary tree over the symbol counts (dynamic programme over symbol ranges with Knuth's bounds on the
// roots, O(sigma^2) on the host); counts are floored so that no code is longer than kHtMaxLen bits.  Returns false when the code would
// not pay: the alphabet fills its fixed width (random bytes), or fewer than two symbols.
struct HtHost {
    uint32_t ent[256];
    uint16_t t12[1 << kHtFastBits];
    int sigma;
    double avg_len;
    // context codes (round 6): ctx != 0 -> the keys are k_ht_keys_ctx's; cls / ent1 are what the device tables at kHtCtxOff hold
    int ctx;
    uint8_t cls[256];
    uint32_t ent1[kHtCtxClasses][256];
    double avg_ctx;                 // mean length of a stream code word (weighted by the pair counts)
};
// optimal alphabetic (order-preserving) prefix code of ns >= 2 symbols with weights w, code words of at most kHtMaxLen bits:
// ent[i] = code left-aligned | length.  (The DP of ht_build, without the table.)
static bool ht_codes(const double* wd, int ns, uint32_t* ent, double* avg_out);
static bool ht_build(const unsigned long long* counts256, int fixed_bits, HtHost* out)
{
    unsigned long long w[256];
    int ns = 0;
    unsigned long long total = 0;
    out->ctx = 0;
    for (int c = 0; c < 256; c++)
        if (counts256[c]) { w[ns++] = counts256[c]; total += counts256[c]; }
    if (ns < 2 || total == 0) return false;
    {
        double wd[256], avg = 0;
        for (int i = 0; i < ns; i++) wd[i] = (double)w[i];
        if (!ht_codes(wd, ns, out->ent, &avg)) return false;
        for (int i = ns; i < 256; i++) out->ent[i] = 0xFFFFFFFFu;
        out->sigma = ns;
        out->avg_len = avg;
        // fast table: where the codes end inside a 12-bit window, decoded greedily
        for (unsigned wv = 0; wv < (1u << kHtFastBits); wv++) {
            unsigned used = 0, ends = 0;
            for (;;) {
                const uint32_t win = used < (unsigned)kHtFastBits ? (wv << (32 - kHtFastBits)) << used : 0u;
                int sym = 0;                                   // the last symbol whose code word is <= the window (code words ascend)
                for (int lo_s = 0, hi_s = ns - 1; lo_s <= hi_s;) {
                    const int mid = (lo_s + hi_s) / 2;
                    if ((out->ent[mid] & ~31u) <= win) { sym = mid; lo_s = mid + 1; } else { hi_s = mid - 1; }
                }
                const unsigned l = out->ent[sym] & 31u;
                if (used + l > (unsigned)kHtFastBits) break;
                used += l;
                ends |= 1u << (used - 1u);
            }
            out->t12[wv] = (uint16_t)ends;
        }
        return avg + 0.75 < (double)fixed_bits;
    }
}

static bool ht_codes(const double* wd, int ns, uint32_t* ent, double* avg_out)
{
    if (ns < 2 || ns > 256) return false;
    double total = 0;
    for (int i = 0; i < ns; i++) total += wd[i];
    if (!(total > 0)) return false;
    std::vector<double> cost_v((size_t)ns * ns);
    std::vector<unsigned short> root_v((size_t)ns * ns);
    double* const C = cost_v.data();
    unsigned short* const R = root_v.data();
    auto at = [ns](int i, int j) { return (size_t)i * ns + j; };
    int len[256];
    // the weight floor total / 2^shift caps the depth of the rare symbols: the largest shift (the least distortion) whose tree has
    // no code word above kHtMaxLen -- feasibility is monotone in the shift, so a bisection over [4, 16] (4 trees instead of up to 13)
    int lo_s = 4, hi_s = 16, good = -1;
    uint32_t best_ent[256];
    double best_avg = 0;
    while (lo_s <= hi_s) {
        const int shift = (lo_s + hi_s) / 2;
        double ww[256], pre[257];
        const double floor_w = total / (double)(1ull << shift);
        pre[0] = 0;
        for (int i = 0; i < ns; i++) { ww[i] = wd[i] > floor_w ? wd[i] : floor_w; pre[i + 1] = pre[i] + ww[i]; }
        for (int i = 0; i < ns; i++) { C[at(i, i)] = 0; R[at(i, i)] = (unsigned short)i; }
        for (int L = 2; L <= ns; L++) {
            for (int i = 0; i + L <= ns; i++) {
                const int j = i + L - 1;
                double best = 1e300;
                int bk = i;
                int klo = R[at(i, j - 1)], khi = R[at(i + 1, j)];
                if (klo < i) klo = i;
                if (khi > j - 1) khi = j - 1;
                if (khi < klo) khi = klo;
                for (int k = klo; k <= khi; k++) {
                    const double v = C[at(i, k)] + C[at(k + 1, j)];
                    if (v < best) { best = v; bk = k; }
                }
                C[at(i, j)] = best + (pre[j + 1] - pre[i]);
                R[at(i, j)] = (unsigned short)bk;
            }
        }
        struct Item { int i, j, d; uint32_t code; } stack[512];
        int sp = 0;
        stack[sp++] = Item{0, ns - 1, 0, 0u};
        bool ok = true;
        while (sp) {
            const Item it = stack[--sp];
            if (it.i == it.j) {
                len[it.i] = it.d ? it.d : 1;
                if (len[it.i] > kHtMaxLen) { ok = false; break; }
                ent[it.i] = (it.d ? (it.code << (32 - it.d)) : 0u) | (uint32_t)len[it.i];
                continue;
            }
            if (it.d >= kHtMaxLen) { ok = false; break; }
            const int k = R[at(it.i, it.j)];
            stack[sp++] = Item{k + 1, it.j, it.d + 1, (it.code << 1) | 1u};
            stack[sp++] = Item{it.i, k, it.d + 1, it.code << 1};
        }
        if (!ok) { hi_s = shift - 1; continue; }
        double avg = 0;
        for (int i = 0; i < ns; i++) avg += wd[i] / total * len[i];
        good = shift;
        best_avg = avg;
        for (int i = 0; i < ns; i++) best_ent[i] = ent[i];
        lo_s = shift + 1;
    }
    if (good < 0) return false;
    for (int i = 0; i < ns; i++) ent[i] = best_ent[i];
    *avg_out = best_avg;
    return true;
}

// Context codes for a text whose pair counts are big[prev * sigma + cur] (dense symbols; `base` = its order-0 code, whose ent
// gives the first symbol of every key).  The predecessors are put into at most kHtCtxClasses classes -- the high nibble of their
// byte value to begin with (UTF-8 lead and continuation bytes, the ASCII ranges), then a few rounds of moving every predecessor
// to the class whose successor distribution it fits best (the gain in sum of n log n / N) -- and every class gets the optimal
// alphabetic code of ITS successor counts (all sigma symbols, a weight floor for those that never follow the class).  Taken
// when the symbols a key holds on average -- 1 + (60 - order-0 length) / stream length -- beat the order-0 key's 64 / length by
// 10 %: the count costs 4 key bits, the class tables a bigram pass over the text.
static bool ht_ctx_build(const std::vector<unsigned long long>& big, int sigma, const unsigned char* dense_byte, HtHost* out)
{
    out->ctx = 0;
    if (sigma < 2 || sigma > kHtCtxSigmaMax || out->sigma != sigma) return false;
    const int NC = kHtCtxClasses;
    std::vector<double> W((size_t)NC * sigma, 0.0), tot(NC, 0.0);
    int cls[256];
    {   // initial classes: the high nibbles that occur, in order
        int map[16], used = 0;
        for (int k = 0; k < 16; k++) map[k] = -1;
        for (int p = 0; p < sigma; p++) {
            const int nib = dense_byte[p] >> 4;
            if (map[nib] < 0) map[nib] = used++;
            cls[p] = map[nib];
        }
    }
    // cost of a class = N log N - sum n log n (N times its entropy).  A move of predecessor p changes the terms of ITS successors
    // only -- the rows of the pair counts are sparse (a UTF-8 lead byte is followed by 64 values at most) -- so a candidate move
    // costs nnz(p) logarithms, single precision: the gains need no more (the whole refinement: < 1 ms on the host).
    auto xlogx = [](double x) { return x > 0 ? x * (double)log2f((float)x) : 0.0; };
    std::vector<int> nz_start(sigma + 1, 0), nz_q;
    std::vector<double> nz_v, rowsum(sigma, 0.0);
    for (int p = 0; p < sigma; p++) {
        nz_start[p] = (int)nz_q.size();
        for (int q = 0; q < sigma; q++) {
            const double v = (double)big[(size_t)p * sigma + q];
            if (v > 0) { nz_q.push_back(q); nz_v.push_back(v); rowsum[p] += v; }
        }
    }
    nz_start[sigma] = (int)nz_q.size();
    for (int p = 0; p < sigma; p++)
        for (int k = nz_start[p]; k < nz_start[p + 1]; k++) { W[(size_t)cls[p] * sigma + nz_q[k]] += nz_v[k]; tot[cls[p]] += nz_v[k]; }
    // bits the code of class c grows by when p joins it (sign = +1), or shrinks by when p leaves it (sign = -1, negated)
    auto delta = [&](int c, int p, double sign) {
        double d = xlogx(tot[c] + sign * rowsum[p]) - xlogx(tot[c]);
        for (int k = nz_start[p]; k < nz_start[p + 1]; k++) {
            const double w = W[(size_t)c * sigma + nz_q[k]];
            d -= xlogx(w + sign * nz_v[k]) - xlogx(w);
        }
        return d;
    };
    for (int round = 0; round < 3; round++) {
        bool moved = false;
        for (int p = 0; p < sigma; p++) {
            if (!(rowsum[p] > 0)) continue;
            const int a = cls[p];
            const double leave = -delta(a, p, -1.0);                   // bits class a's code loses when p leaves
            double best_gain = 0;
            int best = a;
            for (int c = 0; c < NC; c++) {
                if (c == a) continue;
                const double gain = leave - delta(c, p, 1.0);          // bits saved by moving p from a to c
                if (gain > best_gain + 1e-6 * rowsum[p]) { best_gain = gain; best = c; }
            }
            if (best != a) {
                for (int k = nz_start[p]; k < nz_start[p + 1]; k++) {
                    W[(size_t)a * sigma + nz_q[k]] -= nz_v[k];
                    W[(size_t)best * sigma + nz_q[k]] += nz_v[k];
                }
                tot[a] -= rowsum[p];
                tot[best] += rowsum[p];
                cls[p] = best;
                moved = true;
            }
        }
        if (!moved) break;
    }
    // (the classes' code trees are independent: one host thread each -- 16 x 4 trees of sigma^2 cells are 3 ms in a row)
    double bits = 0, pairs = 0;
    double avg_c[kHtCtxClasses];
    bool ok_c[kHtCtxClasses];
    std::vector<std::thread> workers;
    for (int c = 0; c < NC; c++) {
        for (int q = 0; q < 256; q++) out->ent1[c][q] = out->ent[q < sigma ? q : 0];
        ok_c[c] = true;
        avg_c[c] = 0;
        if (!(tot[c] > 0)) continue;                                   // (an empty class: the order-0 code, never looked up)
        auto job = [&W, &ok_c, &avg_c, out, sigma, c] { ok_c[c] = ht_codes(&W[(size_t)c * sigma], sigma, out->ent1[c], &avg_c[c]); };
        try { workers.emplace_back(job); } catch (...) { job(); }       // (no thread to be had: in line)
    }
    for (auto& w : workers) w.join();
    for (int c = 0; c < NC; c++) {
        if (!ok_c[c]) return false;
        bits += avg_c[c] * tot[c];
        pairs += tot[c];
    }
    if (!(pairs > 0)) return false;
    for (int p = 0; p < 256; p++) out->cls[p] = (uint8_t)(p < sigma ? cls[p] : 0);
    out->avg_ctx = bits / pairs;
    const double sym_ctx = 1.0 + (64.0 - kHtCtxCountBits - out->avg_len) / out->avg_ctx;
    const double sym_0 = (double)kHtKeyBits / out->avg_len;
    static const int force = [] { const char* e = dev_env("SFX_HT_CTX"); return e ? atoi(e) : 1; }();   // 0: never, 2: always (development)
    if (force == 0) return false;
    out->ctx = (force == 2 || sym_ctx >= 1.10 * sym_0) ? 1 : 0;
    return out->ctx != 0;
}

struct SaBuffers {
    uint64_t kv_cap;                                    // (K0, VA) and (K1, VB) are each "kv_cap keys, then kv_cap values": radix_sort_kv64's kv12_cap
    uint64_t* K0; uint64_t* K1;                         // key ping-pong (8 B per element)
    uint32_t* VA; uint32_t* VB;                         // suffix ping-pong
    uint32_t* S0; uint32_t* S1;                         // slot lists
    uint32_t* G;                                        // bucket ids of the active list (G/G1 ping-pong)
    uint32_t* G1;
    uint16_t* F;                                        // head / single bits, 16 per 8 elements
    uint8_t* F8;                                        // one flag byte per element (tile rounds)
    uint16_t* Hd0; uint16_t* Hd1;                       // deep text rounds: depth of every list member's bucket, paired with S0 / S1
    unsigned long long* counters;                       // 4
    unsigned long long* deep_slots;                     // kDeepSlotWords
    uint32_t* block_counts;                             // kMaxGrid
    uint32_t* R;                                        // scratch (positions of the large-bucket members of a tile round)
    uint32_t* isa;
    uint32_t* packed;                                   // PackedText words
    uint32_t* hist;                                     // radix_scratch_words(cap)
    uint32_t* part_head; uint32_t* part_keep; uint32_t* part_ghead;   // kMaxGrid each
    uint32_t* totals;
    unsigned long long* bins;                           // 256
    uint8_t* lut;                                       // 256
    uint32_t* ht;                                       // order-preserving code of the dense symbols: [256] table, [256] minimum depth
};

static inline uint32_t* isa_scratch_h(SaBuffers& b) { return b.R; }       // n + 1024 u32, free between rounds
static inline uint16_t* hd_of(SaBuffers& b, const uint32_t* S) { return S == b.S0 ? b.Hd0 : b.Hd1; }

struct SizerArena : ArenaSizer {
    template <class T> T* take(uint64_t c) { ArenaSizer::take<T>(c); return nullptr; }
};

template <class A>
static void carve_sa(A& ar, uint64_t n, uint64_t cap, uint64_t isa_len, SaBuffers* b)
{
    // (K0, VA) and (K1, VB) back to back: each pair doubles as one array of 12-byte (key, suffix) elements for the middle
    // passes of the 64-bit-key sorts (kv12_region, sfx_radix.hip)
    uint64_t* K0 = ar.template take<uint64_t>(cap);
    uint32_t* VA = ar.template take<uint32_t>(cap);
    uint64_t* K1 = ar.template take<uint64_t>(cap);
    uint32_t* VB = ar.template take<uint32_t>(cap);
    uint32_t* S0 = ar.template take<uint32_t>(cap);
    uint32_t* S1 = ar.template take<uint32_t>(cap);
    uint32_t* G = ar.template take<uint32_t>(cap);
    uint32_t* G1 = ar.template take<uint32_t>(cap);
    uint16_t* F = ar.template take<uint16_t>(cap / kGroupItems + kBlock * kApplySub);
    uint32_t* bc = ar.template take<uint32_t>(kMaxGrid);
    uint8_t* F8 = ar.template take<uint8_t>(cap + 64);
    unsigned long long* counters = ar.template take<unsigned long long>(4);
    unsigned long long* deep_slots = ar.template take<unsigned long long>(kDeepSlotWords);
    uint32_t* R = ar.template take<uint32_t>(cap + 1024);
    uint32_t* isa = ar.template take<uint32_t>(isa_len);
    // the bucket depths of the text rounds live in the rank array, which is idle until the build switches to rank
    // rounds (and the depths are dead from then on); a slice of the partitioned build has no rank array
    uint16_t* Hd0 = isa_len >= cap ? reinterpret_cast<uint16_t*>(isa) : ar.template take<uint16_t>(cap);
    uint16_t* Hd1 = isa_len >= cap ? (Hd0 ? Hd0 + cap : nullptr) : ar.template take<uint16_t>(cap);
    uint32_t* packed = ar.template take<uint32_t>(packed_words(n, nullptr));
    uint32_t* hist = ar.template take<uint32_t>(radix_scratch_words(cap));
    uint32_t* ph = ar.template take<uint32_t>(kMaxGrid);
    uint32_t* pk = ar.template take<uint32_t>(kMaxGrid);
    uint32_t* pg = ar.template take<uint32_t>(kMaxGrid);
    uint32_t* totals = ar.template take<uint32_t>(64);
    unsigned long long* bins = ar.template take<unsigned long long>(256);
    uint8_t* lut = ar.template take<uint8_t>(256);
    static_assert(kHtCtxOff == kHtTableWords + (1u << kHtFastBits) / 2, "the context tables lie behind the fast table");
    uint32_t* ht = ar.template take<uint32_t>(kHtCtxOff + kHtCtxWords);
    if (b) {
        b->ht = ht;
        b->kv_cap = cap;
        b->K0 = K0; b->K1 = K1; b->VA = VA; b->VB = VB; b->S0 = S0; b->S1 = S1; b->G = G; b->G1 = G1; b->F = F; b->F8 = F8;
        b->Hd0 = Hd0; b->Hd1 = Hd1;
        b->counters = counters; b->deep_slots = deep_slots; b->block_counts = bc; b->R = R;
        b->isa = isa; b->packed = packed; b->hist = hist; b->part_head = ph; b->part_keep = pk;
        b->part_ghead = pg; b->totals = totals; b->bins = bins; b->lut = lut;
    }
}

uint64_t sa_workspace_bytes(uint64_t n)
{
    SizerArena s;
    uint64_t cap = n < 2 ? 2 : n;
    carve_sa(s, cap, cap, cap, (SaBuffers*)nullptr);
    return s.used + 256;
}
uint64_t sa_range_workspace_bytes(uint64_t n, uint64_t max_count)
{
    SizerArena s;
    carve_sa(s, n < 2 ? 2 : n, max_count < 2 ? 2 : max_count, 0, (SaBuffers*)nullptr);
    s.take<uint32_t>(kMaxGrid);
    return s.used + 256;
}

// bucket statistics of the sorted active list: reduce -> scan -> {kept, kept buckets} on the host
template <class KeyT>
static int round_totals(const KeyT* K, uint64_t m, SaBuffers& b, hipStream_t st, uint64_t* kept,
                        uint64_t* kept_groups, LcpFuse fuse = LcpFuse{nullptr, 0, 0, 0, nullptr})
{
    Chunking ch = make_chunking(m, kApplyTile);
    if (sizeof(KeyT) == 8 && fuse.lcp && fuse.ht)
        SFX_LAUNCH("groups_reduce", (double)m * (sizeof(KeyT) + 4), (k_groups_reduce<KeyT, sizeof(KeyT) == 8>), ch.blocks, kBlock,
                   st, K, m, ch.tiles_per_block * kApplyTile, b.part_head, b.part_keep, b.part_ghead, b.F, fuse);
    else
    SFX_LAUNCH("groups_reduce", (double)m * (sizeof(KeyT) + (fuse.lcp ? 4 : 0)), (k_groups_reduce<KeyT>), ch.blocks, kBlock,
               st, K, m, ch.tiles_per_block * kApplyTile, b.part_head, b.part_keep, b.part_ghead, b.F, fuse);
    SFX_LAUNCH("groups_scan", 0.0, k_groups_scan, 1, kBlock, st, b.part_head, b.part_keep,
               b.part_ghead, ch.blocks, b.totals);
    uint32_t host_totals[2] = {0, 0};
    SFX_TRY(read_back(host_totals, b.totals, sizeof(host_totals), st));
    *kept = host_totals[0];
    *kept_groups = host_totals[1];
    return SFX_OK;
}
// write SA (+ ranks) and compact the unresolved buckets for the next round
template <class KeyT>
static int round_apply(const KeyT* K, const uint32_t* V, const uint32_t* S, uint64_t m, SaBuffers& b,
                       uint32_t* sa, uint32_t* isa, uint32_t* S_next, uint32_t* V_next,
                       uint32_t* R_next, hipStream_t st, int sa_mode, uint64_t n, sfx_build_stats& stats,
                       uint64_t kept, const uint16_t* Hd = nullptr, uint16_t* Hd_next = nullptr, uint32_t hd_floor = 0,
                       HtDepth ht = HtDepth{nullptr, 0, 0}, uint32_t* min_depth = nullptr, const uint8_t* rank_flags = nullptr,
                       const uint32_t* part_pairs = nullptr, uint64_t npairs = 0)
{
    // rank_flags / part_pairs / npairs (rank rounds after an LDS or segmented sort): only the members whose rank changes
    // are written -- npairs of the m (k_flags_reduce counted them)
    const uint64_t pair_count = rank_flags ? npairs : m;
    const bool sa_in_place = sa_mode == 1;
    // the sorted keys K sit in one of K0/K1 (for 32-bit keys: in its first half); the other
    // one is free for the (suffix, rank) pairs, and K's own buffer is free once this kernel is done
    uint64_t* pairs = nullptr;
    uint64_t* pairs_tmp = nullptr;
    if (isa && n >= partitioned_scatter_min()) {
        const bool k_in_0 = (const void*)K >= (const void*)b.K0 && (const void*)K < (const void*)(b.K0 + m);
        pairs = k_in_0 ? b.K1 : b.K0;
        pairs_tmp = k_in_0 ? b.K0 : b.K1;
    }
    Chunking ch = make_chunking(m, kApplyTile);
    // the digit counts of the passes that partition the pairs are taken where the pairs are made
    uint32_t* pair_hist = nullptr;
    int pair_lo = 0, pair_nb = 0;
    unsigned pair_blocks = 0;
    if (pairs) {
        const unsigned most = scatter_pairs_presort_hist(pair_count, n, &pair_lo, &pair_nb);
        if (most && ch.blocks <= most && (pair_nb - pair_lo + 7) / 8 <= 3) { pair_hist = b.hist; pair_blocks = ch.blocks; }
    }
    const char* name = sizeof(KeyT) == 4 ? "groups_apply_u32" : "groups_apply_u64";
    // algorithmic bytes: every element's flags (2 bits), key and suffix are read (+ its slot from the second round on);
    // a resolved suffix is written to the SA (unless the array in V is the SA); a kept one leaves as suffix + slot + bucket
    // id + 16-bit depth = 14 bytes; a rank round adds one 8-byte (suffix, rank) pair per member whose rank changes
    const double algo = (double)m * (0.25 + sizeof(KeyT) + 4 + (S ? 4 : 0)) + (sa_in_place ? 0.0 : 4.0 * (double)(m - dmin<uint64_t>(kept, m))) +
                        14.0 * (double)kept + (isa ? 8.0 * (double)pair_count : 0.0);
    uint32_t* sa_arg = sa_in_place ? (uint32_t*)nullptr /* V is the SA */ : sa;
    if (sa_in_place && !isa && kept * kSparseApplyDivisor <= m) {
#define SFX_APPLY_SPARSE(HTV)                                                                                                       \
        SFX_LAUNCH(name, algo, (k_groups_apply<KeyT, kApplySub, HTV, false>), ch.blocks, kBlock, st, K, V, S, m,                    \
                   ch.tiles_per_block * kApplyTile, b.part_head, b.part_keep, b.part_ghead, sa_arg, isa, S_next, V_next,            \
                   b.G, R_next, 1, pairs, (const uint16_t*)b.F, (uint32_t*)nullptr, 0, 0, Hd, Hd_next, hd_floor, ht, min_depth)
        if (sizeof(KeyT) == 8 && ht.ent) SFX_APPLY_SPARSE((sizeof(KeyT) == 8));
        else if (ht.ent) return SFX_ERR_INTERNAL;
        else SFX_APPLY_SPARSE(false);
#undef SFX_APPLY_SPARSE
    }
    else {
#define SFX_APPLY(HTV, PV)                                                                                                          \
        SFX_LAUNCH(name, algo, (k_groups_apply<KeyT, 1, HTV, PV>), ch.blocks, kBlock, st, K, V, S, m,                               \
                   ch.tiles_per_block * kApplyTile, b.part_head, b.part_keep, b.part_ghead, sa_arg, isa, S_next, V_next,            \
                   b.G, R_next, sa_mode, pairs, (const uint16_t*)b.F, pair_hist, pair_lo, pair_nb, Hd, Hd_next, hd_floor, ht, min_depth, \
                   rank_flags, part_pairs)
        if (ht.ent && pair_hist) return SFX_ERR_INTERNAL;      // (compressed keys belong to the initial pass, pairs to rank rounds)
        if (sizeof(KeyT) == 8 && ht.ent) SFX_APPLY((sizeof(KeyT) == 8), false);
        else if (sizeof(KeyT) == 8 && pair_hist) SFX_APPLY(false, (sizeof(KeyT) == 8));
        else if (ht.ent || pair_hist) return SFX_ERR_INTERNAL;
        else SFX_APPLY(false, false);
#undef SFX_APPLY
    }
    if (pairs && pair_count) SFX_TRY(scatter_pairs_u32(pairs, pairs_tmp, pair_count, n, isa, b.hist, st, &stats, pair_blocks));
    return SFX_OK;
}

int byte_histogram_dev(const uint8_t* d_text, uint64_t begin, uint64_t end, uint64_t* d_bins,
                       hipStream_t st)
{
    if (!d_bins || (end > begin && !d_text)) return SFX_ERR_ARG;
    SFX_HIP(hipMemsetAsync(d_bins, 0, 256 * sizeof(uint64_t), st));
    if (end <= begin) return SFX_OK;
    uint64_t cnt = end - begin;
    unsigned grid = (unsigned)dmin<uint64_t>((cnt + kBlock * 16 - 1) / (kBlock * 16), kMaxGrid);
    SFX_LAUNCH("byte_hist", (double)cnt, k_byte_hist, grid, kBlock, st, d_text, begin, end,
               (unsigned long long*)d_bins);
    return SFX_OK;
}

// alphabet (host) + LUT and packed text (device) from device-resident global byte counts
// `n_words_out` > 0: write exactly that many words (zeros past the text) instead of the
// default (n + spw - 1) / spw + 3; `packed_in`: the text is already packed (partitioned
// build after an all-gather of packed shards) -- only the alphabet is derived.
static int prepare_text(const uint8_t* d_text, uint64_t n, const unsigned long long* d_bins,
                        uint8_t* d_lut, uint32_t* d_packed, hipStream_t st, Alphabet* alpha,
                        PackedText* pt, uint64_t n_words_out = 0, const uint32_t* packed_in = nullptr)
{
    unsigned long long host_bins[256];
    if (!packed_in) SFX_LAUNCH("make_lut", 0.0, k_make_lut, 1, kBlock, st, d_bins, d_lut);
    SFX_TRY(read_back(host_bins, d_bins, sizeof(host_bins), st));
    *alpha = make_alphabet(host_bins, n);
    if (packed_in) {
        pt->words = packed_in;
        pt->n = n;
        pt->bits = alpha->bits;
        pt->spw = alpha->spw;
        pt->kbits = alpha->kbits;
        pt->inv_spw = 1.0 / alpha->spw;
        return SFX_OK;
    }
    uint64_t nw = n_words_out ? n_words_out : packed_words(n, alpha);
    const double pack_bytes = (double)n * (1.0 + alpha->bits / 8.0);
    const bool pow2 = (alpha->bits * alpha->spw == 32 || (alpha->bits == 7 && alpha->spw == 4)) &&   // bits in {1, 2, 4, 8}, or 7
                      (reinterpret_cast<uintptr_t>(d_text) & 15u) == 0;
    if (pow2) {
        const unsigned grid = (unsigned)dmin<uint64_t>((nw + kBlock - 1) / kBlock, 4 * kMaxGrid);
        if (alpha->bits == 7) {
            SFX_LAUNCH("pack_text", pack_bytes, (k_pack_text_pow2<4, 7>), grid, kBlock, st, d_text, n, d_lut, nw, d_packed);
        } else if (alpha->spw == 4) {
            SFX_LAUNCH("pack_text", pack_bytes, (k_pack_text_pow2<4>), grid, kBlock, st, d_text, n, d_lut, nw, d_packed);
        } else if (alpha->spw == 8) {
            SFX_LAUNCH("pack_text", pack_bytes, (k_pack_text_pow2<8>), grid, kBlock, st, d_text, n, d_lut, nw, d_packed);
        } else if (alpha->spw == 16) {
            SFX_LAUNCH("pack_text", pack_bytes, (k_pack_text_pow2<16>), grid, kBlock, st, d_text, n, d_lut, nw, d_packed);
        } else {
            SFX_LAUNCH("pack_text", pack_bytes, (k_pack_text_pow2<32>), grid, kBlock, st, d_text, n, d_lut, nw, d_packed);
        }
    } else {
        Chunking ch = make_chunking(nw, kPackWords);
        SFX_LAUNCH("pack_text", pack_bytes, k_pack_text, ch.blocks, kBlock, st, d_text, n, d_lut, alpha->bits,
                   alpha->spw, ch.tiles_per_block, nw, d_packed);
    }
    pt->words = d_packed;
    pt->n = n;
    pt->bits = alpha->bits;
    pt->spw = alpha->spw;
    pt->kbits = alpha->kbits;
    pt->inv_spw = 1.0 / alpha->spw;
    return SFX_OK;
}

// which byte values occur in the text (host copy of the 256 presence flags); d_small4k: 4 KiB of device scratch
int byte_presence_host(const uint8_t* d_text, uint64_t n, void* d_small4k, unsigned long long* host_bins256,
                       hipStream_t st)
{
    unsigned long long* bins = reinterpret_cast<unsigned long long*>(d_small4k);
    SFX_HIP(hipMemsetAsync(bins, 0, 256 * sizeof(unsigned long long), st));
    if (n) {
        const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock * 64 - 1) / (kBlock * 64), kMaxGrid);
        SFX_LAUNCH("byte_presence", (double)n, k_byte_presence, grid, kBlock, st, d_text, n, bins);
    }
    return read_back(host_bins256, bins, 256 * sizeof(unsigned long long), st);
}

// For consumers outside the SA build (the direct LCP pass): pack the text if its alphabet needs
// at most max_bits per symbol.  small = 4 KiB of device scratch, d_packed = n / 8 + 8 words.
int pack_small_alphabet(const uint8_t* d_text, uint64_t n, int max_bits, void* small, uint32_t* d_packed,
                        hipStream_t st, PackedText* pt, bool* packed)
{
    *packed = false;
    unsigned long long* bins = reinterpret_cast<unsigned long long*>(small);
    uint8_t* lut = reinterpret_cast<uint8_t*>(small) + 256 * sizeof(unsigned long long);
    SFX_HIP(hipMemsetAsync(bins, 0, 256 * sizeof(unsigned long long), st));
    const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock * 64 - 1) / (kBlock * 64), kMaxGrid);
    SFX_LAUNCH("byte_presence", (double)n, k_byte_presence, grid, kBlock, st, d_text, n, bins);
    unsigned long long host_bins[256];
    SFX_TRY(read_back(host_bins, bins, sizeof(host_bins), st));
    const Alphabet alpha = make_alphabet(host_bins);
    if (alpha.bits > max_bits || packed_words(n, &alpha) > n / 8 + 8) return SFX_OK;
    Alphabet a2;
    SFX_TRY(prepare_text(d_text, n, bins, lut, d_packed, st, &a2, pt));
    *packed = true;
    return SFX_OK;
}


// Direct ordering of the small buckets of the active list (S_cur, *V_cur, b.G; m elements
// sharing their first h symbols inside each bucket), followed by compaction of what is
// still unresolved.  On return the active list is (*S_cur, *V_cur, b.G) with *m elements.
static int small_groups_pass(const PackedText& pt, uint64_t h, SaBuffers& b, uint32_t* sa, uint32_t* isa,
                             uint32_t** S_cur, uint32_t** V_cur, uint64_t* m, hipStream_t st,
                             sfx_build_stats& stats, uint32_t* lcp = nullptr, bool carry_hd = false)
{
    const uint64_t cnt = *m;
    uint32_t* V_other = (*V_cur == b.VA) ? b.VB : b.VA;
    uint32_t* S_next = (*S_cur == b.S0) ? b.S1 : b.S0;
    uint32_t* flag = (uint32_t*)b.K0;                       // free between rounds
    const uint16_t* Hd = carry_hd ? hd_of(b, *S_cur) : nullptr;
    uint16_t* Hd_next = carry_hd ? hd_of(b, S_next) : nullptr;
    unsigned grid = (unsigned)dmin<uint64_t>((cnt + kBlock - 1) / kBlock, kMaxGrid);
    SFX_LAUNCH("small_groups", (double)cnt * 32, k_small_groups, grid, kBlock, st, *V_cur, *S_cur, b.G, cnt, pt, h,
               sa, isa, V_other, b.G1, flag, lcp);
    Chunking ch = make_chunking(cnt, 1024);
    const uint64_t chunk = ch.tiles_per_block * 1024;
    SFX_LAUNCH("flag_count", (double)cnt * 4, k_flag_compact, ch.blocks, kBlock, st, flag, *S_cur, V_other, b.G1, cnt,
               chunk, 0, b.block_counts, S_next, *V_cur, b.G, Hd, Hd_next);
    SFX_LAUNCH("flag_scan", 0.0, k_scan_block_counts, 1, kBlock, st, b.block_counts, ch.blocks, b.totals);
    uint32_t left = 0;
    SFX_TRY(read_back(&left, b.totals, sizeof(left), st));
    if (left > 0) {
        SFX_LAUNCH("flag_compact", (double)cnt * 4 + (double)left * 24, k_flag_compact, ch.blocks, kBlock, st, flag,
                   *S_cur, V_other, b.G1, cnt, chunk, 1, b.block_counts, S_next, *V_cur, b.G, Hd, Hd_next);
        *S_cur = S_next;                                    // V stays in *V_cur (compacted from V_other), ids in b.G
    }
    stats.small_bucket_resolved += cnt - left;
    *m = left;
    return SFX_OK;
}
__global__ void __launch_bounds__(kBlock)
k_fill_u16(uint16_t* __restrict__ p, uint64_t n, uint16_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[i] = v;
}
// worth it when the average unresolved bucket is small
static bool small_groups_pay(uint64_t m, uint64_t groups) { return m > 0 && groups * 4 >= m; }

// rank array from the suffix array as far as it is known (see k_iota): b.R holds H, K0/K1 the pairs
static int build_ranks(SaBuffers& b, const uint32_t* sa, uint64_t n, const uint32_t* V_act, const uint32_t* S_act,
                       uint64_t m_act, uint32_t* isa, hipStream_t st, sfx_build_stats& stats)
{
    const unsigned g1 = (unsigned)dmin<uint64_t>((n + kBlock - 1) / kBlock, kMaxGrid);
    const unsigned g2 = (unsigned)dmin<uint64_t>((m_act + kBlock - 1) / kBlock, kMaxGrid);
    uint32_t* H = isa_scratch_h(b);
    // the rounds only write resolved suffixes to the array: the members of unresolved buckets go in now, in
    // list order (any order inside a bucket will do: all members of a bucket get the same rank)
    if (m_act) SFX_LAUNCH("rank_active_slots", (double)m_act * 12, k_scatter_by_slot, g2, kBlock, st, V_act, S_act, m_act,
                          const_cast<uint32_t*>(sa));
    SFX_LAUNCH("rank_iota", (double)n * 4, k_iota, g1, kBlock, st, H, n);
    if (m_act) SFX_LAUNCH("rank_head_slots", (double)m_act * 16, k_head_slots, g2, kBlock, st, S_act, b.G, m_act, H);
    if (n >= partitioned_scatter_min()) {
        SFX_LAUNCH("rank_pairs", (double)n * 16, k_rank_pairs, g1, kBlock, st, sa, H, n, b.K0);
        SFX_TRY(scatter_pairs_u32(b.K0, b.K1, n, n, isa, b.hist, st, &stats));
    } else {
        SFX_LAUNCH("isa_from_sa", (double)n * 12, k_isa_from_sa, g1, kBlock, st, sa, H, n, isa);
    }
    return SFX_OK;
}

// Device-wide composite-key rounds (round 1 of this engine): kept for rank rounds whose key2 = rank + h
// does not fit 32 bits (n > 2^31 with deep repeats); the tile rounds below are the normal path.
//   rank round: key2 = rank of the suffix h symbols on (needs ISA), h doubles
//   text round: key2 = the next spw symbols (needs only the packed text), h += spw
// The partitioned build (isa == nullptr) only has text rounds.  A full build runs
// `text_rounds` text rounds first (0 or 1) and rank rounds after.
static int refine_composite(const PackedText& pt, int cpk, SaBuffers& b, uint32_t* sa, uint32_t* isa,
                            int text_rounds, uint32_t* S_cur, uint32_t* V_cur, uint64_t m, uint64_t id_bound,
                            hipStream_t st, sfx_build_stats& stats, uint64_t h0)
{
    // bucket ids are positions in the active list as it was when they were assigned
    // (< id_bound); the direct pass and its compaction shrink the list but keep the ids
    const uint64_t n = pt.n;
    uint64_t h = h0;
    (void)cpk;
    const int spw = pt.spw;
    const int flag_shift = dmax(pt.kbits, bits_for(n));
    int rounds = 0;
    while (m > 0) {
        // rank rounds: <= ~log2(n); text rounds are bounded by the longest repeat
        if (++rounds > (isa ? 80 : 1 << 20)) return SFX_ERR_INTERNAL;
        const bool text_round = !isa || text_rounds > 0;
        int key2_bits = text_round ? flag_shift + 1 : bits_for(n - 1 + h);
        int gid_bits = bits_for(id_bound > 0 ? id_bound - 1 : 0);
        if (key2_bits + gid_bits > 64) return SFX_ERR_INTERNAL;
        unsigned grid = (unsigned)dmin<uint64_t>((m + kBlock - 1) / kBlock, kMaxGrid);
        if (text_round) {
            SFX_LAUNCH("compose_text_keys", (double)m * 24, k_compose_text_keys, grid, kBlock, st, V_cur,
                       b.G, m, pt, h, flag_shift, key2_bits, b.K0);
        } else {
            SFX_LAUNCH("compose_rank_keys", (double)m * 20, k_compose_rank_keys, grid, kBlock, st,
                       V_cur, b.G, m, isa, n, h, key2_bits, b.K0);
        }
        uint32_t* V_other = (V_cur == b.VA) ? b.VB : b.VA;
        int in1 = 0;
        SFX_TRY(radix_sort_kv64(b.K0, V_cur, b.K1, V_other, m, 0, key2_bits + gid_bits, b.hist, st, &in1,
                                &stats, nullptr, nullptr, V_cur == b.VA ? b.kv_cap : 0));
        const uint64_t* Kr = in1 ? b.K1 : b.K0;
        uint32_t* Vr = in1 ? V_other : V_cur;
        uint32_t* V_next = in1 ? V_cur : V_other;
        uint32_t* S_next = (S_cur == b.S0) ? b.S1 : b.S0;
        uint64_t kept = 0, kept_groups = 0;
        SFX_TRY(round_totals<uint64_t>(Kr, m, b, st, &kept, &kept_groups));
        const bool full_text_round = isa && text_round;
        SFX_TRY(round_apply<uint64_t>(Kr, Vr, S_cur, m, b, sa, (isa && !text_round) ? isa : nullptr,
                                      S_next, V_next, full_text_round ? b.R : nullptr, st, 0, n, stats, kept));
        h = text_round ? h + (uint64_t)spw : h * 2;
        if (full_text_round && --text_rounds == 0 && kept > 0)
            SFX_TRY(build_ranks(b, sa, n, V_next, S_next, kept, isa, st, stats));
        S_cur = S_next;
        V_cur = V_next;
        m = kept;
        id_bound = kept;
        stats.rounds++;
        if (small_groups_pay(m, kept_groups)) {
            // the rank array is in use from here on iff the next round is a rank round
            uint32_t* live_isa = (isa && text_rounds <= 0) ? isa : nullptr;
            SFX_TRY(small_groups_pass(pt, h, b, sa, live_isa, &S_cur, &V_cur, &m, st, stats));
        }
    }
    return SFX_OK;
}


// Refinement rounds shared by the full and the partitioned build: every round sorts each bucket of
// the active list by key2 inside LDS (sfx_tile.hip), large buckets through the device-wide sort.
//   text round: key2 = the next wsym symbols (needs only the packed text), h += wsym
//   rank round: key2 = rank of the suffix h symbols on (needs ISA), h doubles
// Rounds start on text symbols -- no rank array, so its n-element scatter is only paid if the
// text rounds stall: when a round resolves less than a quarter of what it was given, and the
// rounds spent stalling have cost about what building the rank array costs, the build switches
// to ranks (prefix doubling).  The partitioned build (isa == nullptr) only has text rounds and
// reports SFX_ERR_NEEDS_RANKS when they do not converge.
constexpr int kMaxTextOnlyRounds = 4096;
static int refine(const PackedText& pt, int cpk, SaBuffers& b, uint32_t* sa, uint32_t* isa, uint32_t* S_cur,
                  uint32_t* V_cur, uint64_t m, hipStream_t st, sfx_build_stats& stats, uint32_t* lcp = nullptr,
                  bool start_with_ranks = false)
{
    const uint64_t n = pt.n;
    uint64_t h = (uint64_t)cpk;
    // symbols a split by the large-bucket path adds to a bucket's depth (32-bit key2: flag + up to 31 bits of symbols).
    // (64-bit keys for that path were measured in round 2 and did not pay: a level is bound by sorting work, bits x members.)
    const int wsym = text_round_symbols(pt);
    bool rank_mode = false;
    uint64_t stalled = 0;
    int rounds = 0;
    bool any_deep = false;
    if (m > 0) SFX_HIP(hipMemsetAsync(b.counters, 0, 4 * sizeof(unsigned long long), st));
    if (start_with_ranks && isa && m > 0) {
        // (development route, SFX_START_RANKS=1: see sort_and_refine)
        SFX_TRY(build_ranks(b, sa, n, V_cur, S_cur, m, isa, st, stats));
        rank_mode = true;
    }
    while (m > 0) {
        if (++rounds > kMaxTextOnlyRounds + 80) return SFX_ERR_INTERNAL;
        // SFX_FORCE_COMPOSITE=1 is a test hook: take the fallback at once (it is otherwise only reachable beyond 2^31 bytes)
        static const bool force_composite = [] { const char* e = dev_env("SFX_FORCE_COMPOSITE"); return e && atoi(e) != 0; }();
        if (rank_mode && (n - 1 + h > 0xFFFFFFFFull || force_composite))   // key2 = rank + h would not fit 32 bits
            return refine_composite(pt, cpk, b, sa, isa, 0, S_cur, V_cur, m, m, st, stats, h);
        uint32_t* V_next = (V_cur == b.VA) ? b.VB : b.VA;
        uint32_t* S_next = (S_cur == b.S0) ? b.S1 : b.S0;
        TileRound tr;
        tr.emit = make_lcp_emit(lcp, S_cur, pt, h, rank_mode);
        tr.G = b.G; tr.V = V_cur; tr.F8 = b.F8; tr.F = b.F;
        const bool deep_round = !rank_mode;
        tr.Hd = deep_round ? hd_of(b, S_cur) : nullptr;
        tr.wsym = (uint32_t)wsym;
        tr.h = (uint32_t)(h > 0xFFFFFFFFull ? 0xFFFFFFFFull : h);
        tr.part_head = b.part_head; tr.part_keep = b.part_keep; tr.part_ghead = b.part_ghead;
        tr.block_counts = b.block_counts; tr.totals = b.totals; tr.counters = b.counters; tr.deep_slots = b.deep_slots;
        tr.part_pairs = rank_mode ? b.block_counts : nullptr;   // (idle during a round)
        // scratch of the segmented sort: element ping-pong in K0 / K1; tile table and segment list in the
        // slot list of the next round (free until round_apply), per-tile digit counts in G1, per-segment
        // digit offsets in R, status words in the radix scratch
        tr.EA = b.K0; tr.EB = b.K1; tr.V_other = V_next;
        tr.seg.tiles = S_next;
        tr.seg.segs = S_next + (m / 4) * 3;                    // (8 B per segment, <= m / 1025 of them; tiles: 32 B each)
        tr.seg.tilehist = b.G1;
        tr.seg.segexcl = b.R;
        tr.seg.status = b.hist + 64;
        tr.seg.status_words = radix_scratch_words(m) - 64;
        tr.seg.counters = b.hist;
        if (rank_mode) SFX_TRY(tile_round_rank(isa, n, h, tr, m, st, &stats));
        else { SFX_TRY(deep_round_text(pt, tr, m, st, &stats)); any_deep = true; }
        Chunking ch = make_chunking(m, kApplyTile);
        SFX_LAUNCH("groups_scan", 0.0, k_groups_scan, 1, kBlock, st, b.part_head, b.part_keep, b.part_ghead, ch.blocks,
                   b.totals, tr.part_pairs);
        uint32_t host_totals[3] = {0, 0, 0};
        SFX_TRY(read_back(host_totals, b.totals, sizeof(host_totals), st));
        const uint64_t kept = host_totals[0], kept_groups = host_totals[1], rank_changes = host_totals[2];
        // (SA slots are written when a suffix resolves; the members of still-unresolved buckets only if ranks
        // have to be built from the array: build_ranks below)
        // (deep rounds: the smallest depth a kept bucket leaves with -- what the rank rounds may assume of every bucket,
        // should the build switch now; usually a few symbols more than h + wsym, which only the large buckets are held to)
        uint32_t* min_depth = deep_round ? b.ht + 256 : nullptr;
        if (min_depth) SFX_HIP(hipMemsetAsync(min_depth, 0xFF, sizeof(uint32_t), st));
        SFX_TRY(round_apply<uint64_t>(b.K0, V_cur, S_cur, m, b, sa, rank_mode ? isa : nullptr, S_next, V_next, nullptr,
                                      st, 2, n, stats, kept, tr.Hd, deep_round ? hd_of(b, S_next) : nullptr, 0u,
                                      HtDepth{nullptr, 0, 0}, min_depth, rank_mode ? b.F8 : nullptr, tr.part_pairs, rank_changes));
        // SFX_TRACE=1 (development): one line per round
        static const bool trace = [] { const char* e = dev_env("SFX_TRACE"); return e && atoi(e) != 0; }();
        if (trace)
            fprintf(stderr, "round %d %s h=%llu m=%llu lds=%llu large=%llu kept=%llu kept_groups=%llu gathers=%llu rank_changes=%llu\n", rounds,
                    rank_mode ? "rank" : "text", (unsigned long long)h, (unsigned long long)m,
                    (unsigned long long)stats.tile_sorted, (unsigned long long)stats.large_sorted, (unsigned long long)kept,
                    (unsigned long long)kept_groups, (unsigned long long)stats.deep_gathers, (unsigned long long)rank_changes);
        h = rank_mode ? h * 2 : h + (uint64_t)wsym;
        stats.rounds++;
        if (rank_mode) stats.rank_rounds++; else stats.text_rounds++;
        if (!rank_mode && kept > 0) {
            // a text round is worth another one while it keeps resolving; a stalled one costs about
            // (kept + launch overheads) against ~n for the rank array
            // SFX_SWITCH=text|rank is a development hook: never / always switch after the first round
            static const int force = [] { const char* e = dev_env("SFX_SWITCH"); return !e ? 0 : (e[0] == 't' ? 1 : (e[0] == 'r' ? 2 : 0)); }();
            // (two thirds kept: measured on 1 GB of mixed-script UTF-8, whose first round keeps 75 % -- a second text
            // round costs more than the rank array it postpones)
            if (kept * 3 > m * 2) stalled += kept + (4u << 20);
            // (the depths of the deep rounds are 16-bit; a text that deep is a repeat anyway)
            const bool too_deep = h + 2 * (uint64_t)wsym > 60000;
            // SFX_TEXT_ROUNDS_MIN=<k> (development): at least k text rounds before the switch
            static const uint32_t min_text = [] { const char* e = dev_env("SFX_TEXT_ROUNDS_MIN"); return e ? (uint32_t)atoi(e) : 0u; }();
            // (too_deep overrides SFX_SWITCH=text: the depths are 16-bit whatever the hook asks for -- ADVICE round 5)
            if (isa && (force != 1 || too_deep) && (stalled * 2 > n || force == 2 || too_deep) && (too_deep || stats.text_rounds >= min_text)) {
                // switching to ranks: slot = rank for resolved suffixes, head slot for the rest
                uint32_t md = 0;
                SFX_TRY(read_back(&md, b.ht + 256, sizeof(md), st));
                // (every kept bucket went at least one level down: h is a bound of its own; a 16-bit depth that wrapped
                // is a smaller, still valid, bound of its bucket and must not pull h down)
                if (md != 0xFFFFFFFFu && (uint64_t)md > h) h = md;
                SFX_TRY(build_ranks(b, sa, n, V_next, S_next, kept, isa, st, stats));
                rank_mode = true;
            } else if (!isa && (stalled * 2 > n || stats.text_rounds > (uint32_t)kMaxTextOnlyRounds || too_deep)) {
                // a slice of the partitioned build cannot switch (the ranks of other slices' suffixes are
                // not here): the caller falls back to a whole-array build (suffix_amd/dist.py)
                return SFX_ERR_NEEDS_RANKS;
            }
        }
        S_cur = S_next;
        V_cur = V_next;
        m = kept;
        if (small_groups_pay(m, kept_groups))
            SFX_TRY(small_groups_pass(pt, h, b, sa, rank_mode ? isa : nullptr, &S_cur, &V_cur, &m, st, stats, lcp,
                                      !rank_mode));
    }
    if (any_deep) {
        unsigned long long g = 0;
        SFX_TRY(read_back(&g, b.counters + 3, sizeof(g), st));
        stats.deep_gathers = g;
    }
    return SFX_OK;
}

// initial sort + first bucket pass + refinement.  from_text: the `count` = n elements
// are (key of suffix i, i), fed to the first radix pass straight from the packed text;
// otherwise `count` explicit (key, suffix) pairs already sit in (K0 as KeyT, VA).
template <class KeyT>
static int sort_and_refine(const PackedText& pt, int cpk, uint64_t count, bool from_text, SaBuffers& b,
                           uint32_t* sa, uint32_t* isa, hipStream_t st, sfx_build_stats& stats,
                           unsigned hist_blocks = 0, uint32_t* lcp_fuse = nullptr, const HtHost* ht = nullptr, int elem_bits = 0)
{
    // ht (64-bit keys of a full build): the keys are the suffixes' symbols in an order-preserving prefix code (k_ht_keys);
    // a bucket's depth is what its key holds (between 64 / longest code and kHtMaxSym symbols)
    const KeyT* Kr;
    const uint32_t* Vr;
    uint32_t* V_next;
    bool in_place = false;
    int in1 = 0;
    TieRecords ties = {false, nullptr, nullptr, nullptr, lcp_fuse != nullptr};
    if (sizeof(KeyT) == 4) {
        // E64 elements; the last pass drops every suffix straight into its SA slot and
        // leaves the sorted 32-bit keys in the element buffer it did not read
        uint32_t* k32 = nullptr;
        // (nobody needs the sorted keys to find the buckets: the hybrid route leaves one tie bit per slot instead -- TieRecords; the
        // fused LCP reads the common prefix of neighbours off them once and has them written as well)
        SFX_TRY(radix_sort_e64(b.K0, b.K1, count, 32, 32 + pt.bits * cpk, b.hist, st, &in1, &stats,
                               from_text ? &pt : nullptr, sa, &k32, hist_blocks, from_text ? 0 : elem_bits, &ties));
        Kr = (const KeyT*)k32;
        Vr = sa;
        V_next = b.VA;
        in_place = true;
    } else {
        // (the last pass drops every suffix straight into its SA slot, as the E64 sort does)
        if (ht && from_text && count == pt.n)
            SFX_TRY(radix_sort_ht64(b.K0, b.VA, b.K1, b.VB, count, b.hist, st, &in1, &stats, pt, b.ht, sa, b.kv_cap, ht->ctx ? ht->sigma : 0));
        else
            SFX_TRY(radix_sort_kv64(b.K0, b.VA, b.K1, b.VB, count, 0, pt.bits * cpk, b.hist, st, &in1, &stats,
                                    from_text ? &pt : nullptr, sa, b.kv_cap));
        Kr = (const KeyT*)(in1 ? b.K1 : b.K0);
        Vr = sa;
        V_next = b.VA;
        in_place = true;
    }
    uint64_t kept = 0, groups = 0;
    LcpFuse fuse = {nullptr, 0, 0, 0, nullptr};
    if (lcp_fuse) {                                     // (full builds only: slot r of the sorted keys is SA slot r)
        fuse.lcp = lcp_fuse;
        fuse.ht = ht ? b.ht : nullptr;
        fuse.pending = kLcpBoundFlag | (uint32_t)cpk;
        fuse.pad_bits = 8 * (int)sizeof(KeyT) - pt.bits * cpk;
        fuse.inv_bits = (65536u + (unsigned)pt.bits - 1u) / (unsigned)pt.bits;
        for (unsigned x = 0; x < 64; x++)
            if (((x * fuse.inv_bits) >> 16) != x / (unsigned)pt.bits) return SFX_ERR_INTERNAL;
    }
    if (ties.produced) {
        // every suffix sits in a slot of its run of equal keys; the runs are ordered on the text where they are (k_tie_direct) ...
        if (ht) return SFX_ERR_INTERNAL;
        if (lcp_fuse) {
            // the LCP of neighbours with different 32-bit keys from the keys themselves, `pending` (>= cpk symbols) where they are
            // equal -- k_groups_reduce's emission; its flags and partial counts go unused.  Whatever order k_tie_direct or the rounds
            // give the members of a stretch, the pending entries are finished on the final array (lcp_finish_pending_dev), and the pair
            // at a seam of two runs shares what the two keys share whichever members meet there.
            if (!Kr) return SFX_ERR_INTERNAL;
            Chunking ch = make_chunking(count, kApplyTile);
            SFX_LAUNCH("groups_reduce", (double)count * 8, (k_groups_reduce<KeyT>), ch.blocks, kBlock, st, Kr, count, ch.tiles_per_block * kApplyTile,
                       b.part_head, b.part_keep, b.part_ghead, b.F, fuse);
        }
        const uint64_t nwords = (count + 31) / 32;
        uint32_t* const lines = reinterpret_cast<uint32_t*>(b.deep_slots);     // (idle until the first deep round)
        static_assert(kTieSlots * 4 * sizeof(uint32_t) <= kDeepSlotWords * sizeof(unsigned long long), "the counter lines fit");
        SFX_HIP(hipMemsetAsync(lines, 0, kTieSlots * 4 * sizeof(uint32_t), st));
        {
            const uint64_t waves = (nwords / 2 + kWave - 1) / kWave + 1;          // (a wave: 64 lanes x 2 mask words)
            const unsigned grid = (unsigned)dmin<uint64_t>((waves + kWavesPerBlock - 1) / kWavesPerBlock, 4 * kMaxGrid);
            // SFX_TIE_RUN_MAX=<2 .. 8> (development): the longest stretch k_tie_direct takes
            static const uint32_t run_max = [] { const char* e = dev_env("SFX_TIE_RUN_MAX"); const int v = e ? atoi(e) : (int)kTieRunMax; return (uint32_t)(v >= 2 && v <= (int)kTieRunMax ? v : (int)kTieRunMax); }();
            SFX_LAUNCH("tie_direct", (double)count * 0.125, k_tie_direct, grid, kBlock, st, ties.tmask, count, pt, sa, lines, run_max, ties.lmask);
            SFX_LAUNCH("tie_totals", 0.0, k_tie_totals, 1, kBlock, st, (const uint32_t*)lines, b.totals);
        }
        uint32_t host_totals[3] = {0, 0, 0};
        SFX_TRY(read_back(host_totals, b.totals, sizeof(host_totals), st));
        kept = host_totals[0];
        if (kept > count || (uint64_t)host_totals[1] * 2 > kept || host_totals[2] > kept) return SFX_ERR_INTERNAL;
        stats.active_after_initial = kept;
        if (host_totals[2] == 0) {
            stats.small_bucket_resolved += kept;
            return SFX_OK;
        }
        // ... and what that leaves tied (repeats beyond its depth, long stretches: marked in lmask) goes on as the first active
        // list, as the runs of equal keys it is made of (k_tie_heads)
        stats.small_bucket_resolved += kept - host_totals[2];
        kept = host_totals[2];
        {
            SFX_HIP(hipMemsetAsync(lines, 0, kTieSlots * 4 * sizeof(uint32_t), st));
            const unsigned hgrid = (unsigned)dmin<uint64_t>((nwords + kBlock - 1) / kBlock, kMaxGrid);
            SFX_LAUNCH("tie_heads", (double)count * 0.25 + (double)kept * 8, k_tie_heads, hgrid, kBlock, st, (const uint32_t*)ties.lmask, count, pt,
                       (const uint32_t*)sa, ties.hmask, lines);
            SFX_LAUNCH("tie_totals", 0.0, k_tie_totals, 1, kBlock, st, (const uint32_t*)lines, b.totals);
            uint32_t runs = 0;
            SFX_TRY(read_back(&runs, b.totals, sizeof(runs), st));
            groups = runs;
            if (groups * 2 > kept) return SFX_ERR_INTERNAL;
            Chunking ch = make_chunking(nwords, kBlock);
            const uint64_t chunk = ch.tiles_per_block * kBlock;
            SFX_LAUNCH("tie_list_count", (double)count * 0.125, k_tie_list, ch.blocks, kBlock, st, (const uint32_t*)ties.lmask, (const uint32_t*)ties.hmask,
                       nwords, chunk, 0, b.block_counts, (const uint32_t*)sa, b.S0, V_next, b.G);
            SFX_LAUNCH("flag_scan", 0.0, k_scan_block_counts, 1, kBlock, st, b.block_counts, ch.blocks, b.totals);
            SFX_LAUNCH("tie_list", (double)count * 0.25 + (double)kept * 16, k_tie_list, ch.blocks, kBlock, st, (const uint32_t*)ties.lmask,
                       (const uint32_t*)ties.hmask, nwords, chunk, 1, b.block_counts, (const uint32_t*)sa, b.S0, V_next, b.G);
        }
        uint32_t* S_cur = b.S0;
        if (small_groups_pay(kept, groups))
            SFX_TRY(small_groups_pass(pt, (uint64_t)cpk, b, sa, nullptr, &S_cur, &V_next, &kept, st, stats, lcp_fuse, false));
        if (kept > 0) {
            const unsigned grid = (unsigned)dmin<uint64_t>((kept + kBlock * 4 - 1) / (kBlock * 4), kMaxGrid);
            SFX_LAUNCH("depth_fill", (double)kept * 2, k_fill_u16, grid, kBlock, st, hd_of(b, S_cur), kept, (uint16_t)cpk);
        }
        return refine(pt, cpk, b, sa, isa, S_cur, V_next, kept, st, stats, lcp_fuse, false);
    }
    SFX_TRY(round_totals<KeyT>(Kr, count, b, st, &kept, &groups, fuse));
    stats.active_after_initial = kept;
    // A text most of whose suffixes stay tied on 64 key bits (mixed-script UTF-8: 86 %, frequent words in buckets of 10^5 ..
    // 10^7) will need rank rounds, whose splits only give lower bounds -- the one-call entry point then runs the separate
    // LCP routine whatever was emitted (build_sa_lcp_u32_dev), so the rounds stop emitting: stats.reserved bit 0
    if (lcp_fuse && ht && kept * 4 > count * 3) {
        lcp_fuse = nullptr;
        stats.reserved |= 1u;
    }
    // no rank array yet: its n-element scatter is only paid if the text rounds stall (refine)
    // (every bucket of the first active list shares the cpk symbols of the initial key: b.Hd0, the depths of the deep rounds)
    uint64_t h0 = (uint64_t)cpk;                        // symbols every bucket of the first active list shares
    if (ht && kept > 0) {
        SFX_HIP(hipMemsetAsync(b.ht + 256, 0xFF, sizeof(uint32_t), st));
        SFX_TRY(round_apply<KeyT>(Kr, Vr, nullptr, count, b, sa, nullptr, b.S0, V_next, nullptr, st, in_place ? 1 : 0, pt.n, stats,
                                  kept, nullptr, b.Hd0, 0u, HtDepth{b.ht, ht->sigma, ht->ctx ? kHtCtxCountBits : 0}, b.ht + 256));
        uint32_t md = 0;
        SFX_TRY(read_back(&md, b.ht + 256, sizeof(md), st));
        if (md == 0xFFFFFFFFu || md == 0) return SFX_ERR_INTERNAL;
        h0 = md;
    } else {
        // (fixed-width keys: every bucket of the first list shares cpk symbols -- the depths are filled in below, for what
        // the direct pass leaves: on uniform DNA that is nothing, and the bucket pass saves a fourth scattered store)
        SFX_TRY(round_apply<KeyT>(Kr, Vr, nullptr, count, b, sa, nullptr, b.S0, V_next, nullptr, st, in_place ? 1 : 0, pt.n, stats,
                                  kept, nullptr, nullptr, (uint32_t)cpk));
    }
    uint32_t* S_cur = b.S0;
    if (small_groups_pay(kept, groups))
        SFX_TRY(small_groups_pass(pt, h0, b, sa, nullptr, &S_cur, &V_next, &kept, st, stats, lcp_fuse, ht != nullptr));
    if (!ht && kept > 0) {
        const unsigned grid = (unsigned)dmin<uint64_t>((kept + kBlock * 4 - 1) / (kBlock * 4), kMaxGrid);
        SFX_LAUNCH("depth_fill", (double)kept * 2, k_fill_u16, grid, kBlock, st, hd_of(b, S_cur), kept, (uint16_t)cpk);
    }
    // Rank rounds from the first round on, for a text nearly all of whose suffixes are still tied after the initial key, were
    // measured in round 5 and do NOT pay (profiles/r5_start_ranks_ab.jsonl, 10^9 bytes each): near-duplicate documents (97 %
    // tied) 358.2 -> 368.4 ms, mixed-script UTF-8 (86 %) 208.2 -> 213.4 ms.  The first text round is not wasted on them: it lifts
    // the minimum depth the rank rounds start from (6 -> 13 and 5 -> 10 symbols) for the price of a deep round, where a rank round
    // at h = 6 pays a rank gather per member and a rank update for the same step.  SFX_START_RANKS=1 (development) keeps the
    // route reachable for the tests.
    static const int start_ranks = [] { const char* e = dev_env("SFX_START_RANKS"); return e ? atoi(e) : 0; }();
    const bool to_ranks = isa && kept > 0 && start_ranks == 1;
    return refine(pt, (int)h0, b, sa, isa, S_cur, V_next, kept, st, stats, lcp_fuse, to_ranks);
}

// lcp_fuse != nullptr: also leave, in lcp_fuse[r], the LCP of every adjacent pair that the initial sort
// already told apart (kLcpPending elsewhere); *cpk_out = symbols of the initial key
static int build_sa_impl(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* ws, uint64_t ws_bytes,
                         hipStream_t st, uint32_t* lcp_fuse, int* cpk_out, bool* fused_out = nullptr)
{
    sfx_build_stats& stats = tls_build_stats();
    memset(&stats, 0, sizeof(stats));
    stats.n = n;
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;            // src/table.rs:380
    if (n == 0) return SFX_OK;                                  // :396
    if (!d_text || !d_sa) return SFX_ERR_ARG;
    if (n == 1) {                                               // :397-400
        SFX_HIP(hipMemsetAsync(d_sa, 0, sizeof(uint32_t), st));
        return SFX_OK;
    }
    if (!ws || ws_bytes < sa_workspace_bytes(n)) return SFX_ERR_WORKSPACE;
    if (n <= tiny_limit()) {
        // one workgroup, one launch (sfx_tiny.hip); a text it gives up on (long runs of equal keys: repeats) goes on below
        bool done = false;
        SFX_TRY(tiny_build_sa_dev(d_text, n, d_sa, ws, st, &done));
        if (done) {
            if (fused_out) *fused_out = false;            // (the one-call SA + LCP entry point runs its LCP routine on the array)
            return SFX_OK;
        }
    }

    Arena ar(ws, ws_bytes);
    SaBuffers b;
    carve_sa(ar, n, n, n, &b);
    if (ar.overflow) return SFX_ERR_WORKSPACE;

    SFX_HIP(hipMemsetAsync(b.bins, 0, 256 * sizeof(unsigned long long), st));
    {
        const unsigned grid = (unsigned)dmin<uint64_t>((n + kBlock * 64 - 1) / (kBlock * 64), kMaxGrid);
        SFX_LAUNCH("byte_presence", (double)n, k_byte_presence, grid, kBlock, st, d_text, n, b.bins);
    }
    Alphabet alpha;
    PackedText pt;
    SFX_TRY(prepare_text(d_text, n, b.bins, b.lut, b.packed, st, &alpha, &pt));
    int key_bits, cpk;
    choose_key(alpha, n, &key_bits, &cpk);
    stats.sigma = alpha.sigma;
    stats.bits_per_symbol = (uint32_t)alpha.bits;
    stats.key_bits = (uint32_t)key_bits;
    stats.symbols_per_key = (uint32_t)cpk;
    if (cpk_out) *cpk_out = cpk;
    // a large alphabet may be used unevenly: the counts decide (one more pass over the text, only where it can matter)
    unsigned long long counts[256];
    bool have_counts = false;
    if (key_bits == 32 && alpha.sigma > 16 && n >= (1ull << 16)) {
        SFX_TRY(byte_histogram_dev(d_text, 0, n, reinterpret_cast<uint64_t*>(b.bins), st));
        SFX_TRY(read_back(counts, b.bins, sizeof(counts), st));
        have_counts = true;
        Alphabet counted = make_alphabet(counts, n);
        choose_key(counted, n, &key_bits, &cpk);
        stats.key_bits = (uint32_t)key_bits;
        stats.symbols_per_key = (uint32_t)cpk;
        if (cpk_out) *cpk_out = cpk;
    }
    if (key_bits == 32) return sort_and_refine<uint32_t>(pt, cpk, n, true, b, d_sa, b.isa, st, stats, 0, lcp_fuse);
    // 64-bit keys: compressed when the symbol counts say it pays (natural-language text: 14 symbols per key instead of 8).
    // SFX_HT=0 (development): fixed-width keys.
    static const bool ht_on = [] { const char* e = dev_env("SFX_HT"); return !e || atoi(e) != 0; }();
    static const uint64_t ht_min = [] { const char* e = dev_env("SFX_HT_MIN"); return e ? (uint64_t)strtoull(e, nullptr, 10) : (1ull << 16); }();
    HtHost ht;
    bool use_ht = false;
    if (ht_on && n >= ht_min) {
        if (!have_counts) {
            SFX_TRY(byte_histogram_dev(d_text, 0, n, reinterpret_cast<uint64_t*>(b.bins), st));
            SFX_TRY(read_back(counts, b.bins, sizeof(counts), st));
        }
        use_ht = ht_build(counts, alpha.bits, &ht);
        if (use_ht) {
            stats.symbols_per_key = (uint32_t)((double)kHtKeyBits / ht.avg_len);   // (on average: the code words differ in length)
            // Context codes (round 6, k_ht_keys_ctx) where they buy 10 % more symbols per key -- large alphabets with structure
            // between neighbouring bytes (mixed-script UTF-8: 13 symbols instead of 10).  Not with the fused LCP: the symbols two
            // different keys share cannot be counted off a context code with the order-0 end-mask table.
            static const uint64_t ctx_min = [] { const char* e = dev_env("SFX_HT_CTX_MIN"); return e ? (uint64_t)strtoull(e, nullptr, 10) : (1ull << 26); }();
            // (the bigram pass is only paid where contexts can buy 10 %: the count's 4 bits cost 4 / length symbols, and the order-1
            // entropy of text whose order-0 code already averages under 5 bits -- English-like: 4.6 -- is never that far below it)
            static const int ctx_force = [] { const char* e = dev_env("SFX_HT_CTX"); return e ? atoi(e) : 1; }();
            static const bool trace = [] { const char* e = dev_env("SFX_TRACE"); return e && atoi(e) != 0; }();
            bool ctx_try = !lcp_fuse && ctx_force != 0 && (ht.avg_len >= 5.0 || ctx_force == 2) && (int)alpha.sigma <= kHtCtxSigmaMax && n >= ctx_min &&
                           kHtKeyBits == 64;
            if (ctx_try && ctx_force != 2) {
                // Longer keys only pay where the order-0 keys leave many suffixes tied (word-structured text: most of them), and a
                // text of the same bytes without repeats (independent code points) has the same symbol statistics: a PILOT decides --
                // the first 2^22 suffixes sorted by their order-0 keys (0.3 % of the build), the share that stays tied.  The share only
                // grows with the number of suffixes sorted, so the pilot errs towards the plain keys.
                static const int pilot_log = [] { const char* e = dev_env("SFX_HT_CTX_PILOT"); const int v = e ? atoi(e) : 22; return v >= 4 && v <= 28 ? v : 22; }();
                const uint64_t mp = dmin<uint64_t>(1ull << pilot_log, n / 4);
                SFX_HIP(hipMemcpyAsync(b.ht, ht.ent, sizeof(ht.ent), hipMemcpyHostToDevice, st));
                SFX_HIP(hipMemcpyAsync(b.ht + kHtTableWords, ht.t12, sizeof(ht.t12), hipMemcpyHostToDevice, st));
                int pin1 = 0;
                sfx_build_stats pstats = {};
                SFX_TRY(radix_sort_ht64(b.K0, b.VA, b.K1, b.VB, mp, b.hist, st, &pin1, &pstats, pt, b.ht, nullptr, b.kv_cap, 0));
                uint64_t ptied = 0, pgroups = 0;
                SFX_TRY(round_totals<uint64_t>(pin1 ? b.K1 : b.K0, mp, b, st, &ptied, &pgroups));
                ctx_try = ptied * 16 >= mp;
                if (trace)
                    fprintf(stderr, "[sfx] context codes: pilot of %llu suffixes, %llu tied in %llu runs -> %s\n", (unsigned long long)mp,
                            (unsigned long long)ptied, (unsigned long long)pgroups, ctx_try ? "bigram pass" : "order-0 keys");
            }
            if (ctx_try) {
                const int sg = (int)alpha.sigma;
                unsigned long long* d_big = reinterpret_cast<unsigned long long*>(b.K0);           // (idle until the keys are made)
                SFX_HIP(hipMemsetAsync(d_big, 0, (size_t)sg * sg * sizeof(unsigned long long), st));
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
                const unsigned grid = (unsigned)dmin<uint64_t>((n / 16 + 1023) / 1024 + 1, dmin((unsigned)cus, grid_cap()));
                SFX_LAUNCH("bigram_hist", (double)n, k_bigram_hist, grid, 1024, st, d_text, n, (const uint8_t*)b.lut, sg, d_big);
                std::vector<unsigned long long> big((size_t)sg * sg);
                SFX_HIP(hipMemcpyAsync(big.data(), d_big, big.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                SFX_HIP(hipStreamSynchronize(st));
                unsigned char dense_byte[256];
                int q = 0;
                for (int c = 0; c < 256; c++)
                    if (counts[c]) dense_byte[q++] = (unsigned char)c;
                const auto t_host0 = std::chrono::steady_clock::now();
                const bool took = q == sg && ht_ctx_build(big, sg, dense_byte, &ht);
                if (trace)
                    fprintf(stderr, "[sfx] context codes: %s, order-0 %.3f bits, stream %.3f bits, host %.3f ms\n", took ? "taken" : "not taken", ht.avg_len,
                            ht.avg_ctx, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count());
                if (took) {
                    stats.symbols_per_key = (uint32_t)(1.0 + (64.0 - kHtCtxCountBits - ht.avg_len) / ht.avg_ctx);
                    stats.reserved |= 2u;                                          // (bit 1: context codes)
                }
            }
            SFX_HIP(hipMemcpyAsync(b.ht, ht.ent, sizeof(ht.ent), hipMemcpyHostToDevice, st));
            SFX_HIP(hipMemcpyAsync(b.ht + kHtTableWords, ht.t12, sizeof(ht.t12), hipMemcpyHostToDevice, st));
            if (ht.ctx) {
                SFX_HIP(hipMemcpyAsync(b.ht + kHtCtxOff, ht.cls, sizeof(ht.cls), hipMemcpyHostToDevice, st));
                SFX_HIP(hipMemcpyAsync(b.ht + kHtCtxOff + 64, ht.ent1, sizeof(ht.ent1), hipMemcpyHostToDevice, st));
            }
        }
    }
    // (compressed keys, round 4: the symbols two different keys share are the code words that end inside their common
    // leading bits -- ht_common_n in k_groups_reduce; equal keys share the symbols the key holds, at most kHtMaxSym, which
    // is also how many suffixes at the end of the text have zero-padded keys: the tail fix of the pending pass redoes them)
    if (use_ht && cpk_out) *cpk_out = (int)kHtMaxSym;
    return sort_and_refine<uint64_t>(pt, cpk, n, true, b, d_sa, b.isa, st, stats, 0, lcp_fuse, use_ht ? &ht : nullptr);
}

int build_sa_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* ws, uint64_t ws_bytes,
                     hipStream_t st)
{
    return build_sa_impl(d_text, n, d_sa, ws, ws_bytes, st, nullptr, nullptr);
}

// SuffixTable::new + lcp_lens in one call (src/table.rs:78-85 + :130-138).  When the initial sort
// separates most suffixes (uniform DNA: 98 %) the LCP of those pairs falls out of the sorted keys and
// only the rest is compared on the text; otherwise the separate LCP routine runs on the finished array.
uint64_t sa_lcp_workspace_bytes(uint64_t n)
{
    return dmax(sa_workspace_bytes(n), lcp_workspace_bytes(n));
}
int build_sa_lcp_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, uint32_t* d_lcp, void* ws,
                         uint64_t ws_bytes, hipStream_t st)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!d_text || !d_sa || !d_lcp) return SFX_ERR_ARG;
    if (!ws || ws_bytes < sa_lcp_workspace_bytes(n)) return SFX_ERR_WORKSPACE;
    if (n == 1) {
        SFX_HIP(hipMemsetAsync(d_sa, 0, sizeof(uint32_t), st));
        SFX_HIP(hipMemsetAsync(d_lcp, 0, sizeof(uint32_t), st));
        return SFX_OK;
    }
    int cpk = 0;
    // the lower-bound encoding needs the top bit of an LCP value
    const bool fuse = n < 0x80000000ull;
    bool fused = fuse;
    SFX_TRY(build_sa_impl(d_text, n, d_sa, ws, ws_bytes, st, fuse ? d_lcp : nullptr, &cpk, &fused));
    const sfx_build_stats stats = tls_build_stats();
    bool done = false;
    // pairs split by rank rounds only carry a lower bound: when the text needed ranks for most of its
    // suffixes, finishing those pairs one by one is the separate routine's job (sampling, Phi / PLCP)
    if (fused && !(stats.reserved & 1u) && (stats.rank_rounds == 0 || stats.active_after_initial * 4 <= n))
        SFX_TRY(lcp_finish_pending_dev(d_text, n, d_sa, d_lcp, (uint64_t)cpk, ws, ws_bytes, st, &done));
    if (!done) SFX_TRY(build_lcp_u32_dev(d_text, n, d_sa, d_lcp, ws, ws_bytes, st));
    tls_build_stats() = stats;
    return SFX_OK;
}


// ---- partitioned build -----------------------------------------------------------
int key_histogram_dev(const uint8_t* d_text, uint64_t n, uint64_t begin, uint64_t end,
                      const uint64_t* d_byte_bins, int top_bits, uint64_t* d_bins, hipStream_t st)
{
    if (!d_text || !d_byte_bins || !d_bins || begin > end || end > n) return SFX_ERR_ARG;
    if (top_bits < 1 || top_bits > kMaxTopBits) return SFX_ERR_ARG;
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    SFX_HIP(hipMemsetAsync(d_bins, 0, sizeof(uint64_t) << top_bits, st));
    if (begin == end) return SFX_OK;
    unsigned long long host_bins[256];
    SFX_TRY(read_back(host_bins, d_byte_bins, sizeof(host_bins), st));
    const Alphabet alpha = make_alphabet(host_bins, n);
    int key_bits, cpk;
    choose_key(alpha, n, &key_bits, &cpk);                  // (key width from the WHOLE text's length and counts)
    if (alpha.bits * cpk < top_bits) return SFX_ERR_ARG;
    const int nsym = (top_bits + alpha.bits - 1) / alpha.bits;
    const uint64_t cnt = end - begin;
    Chunking ch = make_chunking(cnt, (uint64_t)kBlock * kKeyHistRun, 512);
    const uint64_t chunk = ch.tiles_per_block * kBlock * kKeyHistRun;
    SFX_LAUNCH("key_hist", (double)cnt, k_key_hist_raw, ch.blocks, kBlock, st, d_text, n, begin, end,
               (const unsigned long long*)d_byte_bins, alpha.bits, nsym, top_bits, chunk,
               (unsigned long long*)d_bins);
    return SFX_OK;
}

template <class KeyT>
static int range_build(const PackedText& pt, int cpk, int top_bits, uint32_t bin_lo, uint32_t bin_hi,
                       uint64_t capacity, uint32_t* d_sa_part, uint64_t* count_out, SaBuffers& b,
                       uint32_t* block_counts, hipStream_t st, sfx_build_stats& stats)
{
    const uint64_t n = pt.n;
    if (bin_lo == 0 && bin_hi == (1u << top_bits)) {
        // the whole key space (a world of one): nothing to filter -- the text-fed initial sort of the full build, text rounds only
        *count_out = n;
        if (n > capacity) return SFX_ERR_WORKSPACE;
        return sort_and_refine<KeyT>(pt, cpk, n, true, b, d_sa_part, nullptr, st, stats);
    }
    // (at most as many workgroups as the sort accepts digit counts from)
    Chunking ch = make_chunking((n + (uint64_t)pt.spw - 1) / (uint64_t)pt.spw, kBlock, 1024);  // in packed words
    const uint64_t chunk = ch.tiles_per_block * kBlock;
    KeyT* k0 = (KeyT*)b.K0;
    const bool dna = pt.bits == 2;                              // the compile-time-width instance
    if (dna)
        SFX_LAUNCH("range_count", (double)n * pt.bits / 8.0, (k_range_filter<KeyT, 2>), ch.blocks, kBlock, st, pt,
                   pt.bits * cpk, top_bits, bin_lo, bin_hi, chunk, 0, block_counts, capacity, k0, b.VA, (uint32_t*)nullptr);
    else
        SFX_LAUNCH("range_count", (double)n * pt.bits / 8.0, (k_range_filter<KeyT, 0>), ch.blocks, kBlock, st, pt,
                   pt.bits * cpk, top_bits, bin_lo, bin_hi, chunk, 0, block_counts, capacity, k0, b.VA, (uint32_t*)nullptr);
    SFX_LAUNCH("range_scan", 0.0, k_scan_block_counts, 1, kBlock, st, block_counts, ch.blocks, b.totals);
    uint32_t host_total = 0;
    SFX_TRY(read_back(&host_total, b.totals, sizeof(host_total), st));
    *count_out = host_total;
    if (host_total > capacity) return SFX_ERR_WORKSPACE;
    if (host_total == 0) return SFX_OK;
    // 32-bit keys: the emit pass also counts the digits for the one-sweep sort of its output
    unsigned hist_blocks = 0;
    uint32_t* digit_partial = nullptr;
    // (the keys of the slice, less its first key, fit elem_bits bits)
    const uint64_t width = (uint64_t)(bin_hi - bin_lo) << (pt.bits * cpk - top_bits);
    const int elem_bits = bits_for(width > 1 ? width - 1 : 1);
    // (a slice the hybrid route will take gets its digit totals from the sub-bucket histogram: four LDS atomics per kept
    // element saved; should the route give way after all, the sort counts for itself)
    if (sizeof(KeyT) == 4 && !radix_e64_hybrid_expected(host_total, elem_bits) &&
        ch.blocks <= radix_e64_presort_hist(host_total, 32, 32 + pt.bits * cpk)) {
        hist_blocks = ch.blocks;
        digit_partial = radix_partial(b.hist);
    }
    if (dna)
        SFX_LAUNCH("range_emit", (double)n * pt.bits / 8.0 + (double)host_total * (sizeof(KeyT) + 4),
                   (k_range_filter<KeyT, 2>), ch.blocks, kBlock, st, pt, pt.bits * cpk, top_bits, bin_lo, bin_hi,
                   chunk, 1, block_counts, capacity, k0, b.VA, digit_partial);
    else
        SFX_LAUNCH("range_emit", (double)n * pt.bits / 8.0 + (double)host_total * (sizeof(KeyT) + 4),
                   (k_range_filter<KeyT, 0>), ch.blocks, kBlock, st, pt, pt.bits * cpk, top_bits, bin_lo, bin_hi,
                   chunk, 1, block_counts, capacity, k0, b.VA, digit_partial);
    return sort_and_refine<KeyT>(pt, cpk, host_total, false, b, d_sa_part, nullptr, st, stats, hist_blocks, nullptr, nullptr,
                                 sizeof(KeyT) == 4 ? elem_bits : 0);
}

// packed-text plumbing of the partitioned build: a rank packs its own shard (global symbol
// codes), the packed shards are all-gathered (bits/8 of the raw volume over xGMI), and the
// range build runs on the packed text directly.
int pack_text_dev(const uint8_t* d_text, uint64_t count, const uint64_t* d_byte_bins, uint8_t* d_lut256,
                  uint32_t* d_words, uint64_t n_words, hipStream_t st)
{
    if (!d_byte_bins || !d_lut256 || !d_words || (count && !d_text)) return SFX_ERR_ARG;
    Alphabet alpha;
    PackedText pt;
    return prepare_text(d_text, count, (const unsigned long long*)d_byte_bins, d_lut256, d_words, st, &alpha, &pt,
                        n_words ? n_words : 1);
}

int build_sa_range_u32_dev(const uint8_t* d_text, uint64_t n, const uint64_t* d_byte_bins,
                           int top_bits, uint32_t bin_lo, uint32_t bin_hi, uint64_t capacity,
                           uint32_t* d_sa_part, uint64_t* count_out, void* ws, uint64_t ws_bytes,
                           hipStream_t st, const uint32_t* d_packed_in)
{
    sfx_build_stats& stats = tls_build_stats();
    memset(&stats, 0, sizeof(stats));
    stats.n = n;
    if (!count_out) return SFX_ERR_ARG;
    *count_out = 0;
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0 || bin_lo >= bin_hi) return SFX_OK;
    if ((!d_text && !d_packed_in) || !d_byte_bins || !d_sa_part || capacity == 0) return SFX_ERR_ARG;
    if (top_bits < 1 || top_bits > kMaxTopBits || bin_hi > (1u << top_bits)) return SFX_ERR_ARG;
    if (!ws || ws_bytes < sa_range_workspace_bytes(n, capacity)) return SFX_ERR_WORKSPACE;

    Arena ar(ws, ws_bytes);
    SaBuffers b;
    carve_sa(ar, n < 2 ? 2 : n, capacity < 2 ? 2 : capacity, 0, &b);
    uint32_t* block_counts = ar.take<uint32_t>(kMaxGrid);
    if (ar.overflow) return SFX_ERR_WORKSPACE;

    Alphabet alpha;
    PackedText pt;
    SFX_TRY(prepare_text(d_text, n, (const unsigned long long*)d_byte_bins, b.lut, b.packed, st, &alpha,
                         &pt, 0, d_packed_in));
    int key_bits, cpk;
    choose_key(alpha, n, &key_bits, &cpk);
    if (alpha.bits * cpk < top_bits) return SFX_ERR_ARG;
    stats.sigma = alpha.sigma;
    stats.bits_per_symbol = (uint32_t)alpha.bits;
    stats.key_bits = (uint32_t)key_bits;
    stats.symbols_per_key = (uint32_t)cpk;
    if (key_bits == 32)
        return range_build<uint32_t>(pt, cpk, top_bits, bin_lo, bin_hi, capacity, d_sa_part, count_out, b,
                                     block_counts, st, stats);
    return range_build<uint64_t>(pt, cpk, top_bits, bin_lo, bin_hi, capacity, d_sa_part, count_out, b,
                                 block_counts, st, stats);
}

}  // namespace sfx
