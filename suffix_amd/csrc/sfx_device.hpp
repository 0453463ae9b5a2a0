// sfx_device.hpp -- wave64 / workgroup primitives shared by every kernel.
//
// gfx950 (CDNA4): a wavefront is 64 lanes, __ballot() is a 64-bit mask, a
// 256-thread workgroup is 4 waves (one per SIMD).  All constants below are
// written for that machine; nothing here is warp-32 shaped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfx {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // threads per workgroup used by every kernel
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned wave_id() { return threadIdx.x >> 6; }

// Orders LDS traffic between the lanes of ONE wave (lanes execute in lock-step;
// this only stops the compiler from moving memory operations across the point).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class T> __host__ __device__ __forceinline__ T dmin(T a, T b) { return a < b ? a : b; }
template <class T> __host__ __device__ __forceinline__ T dmax(T a, T b) { return a > b ? a : b; }

// ---- wave-level inclusive scans (6 shuffle steps over 64 lanes) -------------
template <class T> __device__ __forceinline__ T wave_scan_add(T v)
{
    const unsigned lane = lane_id();
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}
template <class T> __device__ __forceinline__ T wave_scan_max(T v)
{
    const unsigned lane = lane_id();
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d);
        if (lane >= d) v = dmax(v, o);
    }
    return v;
}

// ---- workgroup-level scans over one value per thread -------------------------
// `part` is kWavesPerBlock entries of LDS.  Both return the EXCLUSIVE prefix of
// the calling thread and the workgroup aggregate; two barriers each, so `part`
// may be reused immediately afterwards.
template <class T>
__device__ __forceinline__ T block_scan_add_excl(T v, T* part, T& total)
{
    T incl = wave_scan_add(v);
    if (lane_id() == 63) part[wave_id()] = incl;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (unsigned w = 0; w < (unsigned)kWavesPerBlock; w++) {
        T p = part[w];
        if (w < wave_id()) base += p;
        tot += p;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}
// max-scan: returns the max over all EARLIER threads (identity 0) and the aggregate
template <class T>
__device__ __forceinline__ T block_scan_max_excl(T v, T* part, T& total)
{
    T incl = wave_scan_max(v);
    T prev = __shfl_up(incl, 1u);
    if (lane_id() == 0) prev = 0;
    if (lane_id() == 63) part[wave_id()] = incl;
    __syncthreads();
    T base = 0, tot = 0;
#pragma unroll
    for (unsigned w = 0; w < (unsigned)kWavesPerBlock; w++) {
        T p = part[w];
        if (w < wave_id()) base = dmax(base, p);
        tot = dmax(tot, p);
    }
    __syncthreads();
    total = tot;
    return dmax(base, prev);
}

// ---- radix ranking primitives (device-wide passes in sfx_radix.hip, LDS-local sort in sfx_tile.hip) ----
constexpr int kRadixBitsDev = 8;
constexpr int kRadixDev = 1 << kRadixBitsDev;
// exclusive prefix of one value per thread; ONE barrier (callers alternate `par`)
template <int NW>
__device__ __forceinline__ uint32_t block_scan_excl_1b(uint32_t v, uint32_t (*part)[NW], unsigned& par)
{
    const uint32_t incl = wave_scan_add(v);
    if (lane_id() == 63) part[par][wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (unsigned k = 0; k < (unsigned)NW; k++)
        if (k < wave_id()) base += part[par][k];
    par ^= 1u;
    return base + incl - v;
}

// ... and the sum of all of them
template <int NW>
__device__ __forceinline__ uint32_t block_scan_excl_1b_total(uint32_t v, uint32_t (*part)[NW], unsigned& par, uint32_t& total)
{
    const uint32_t incl = wave_scan_add(v);
    if (lane_id() == 63) part[par][wave_id()] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (unsigned k = 0; k < (unsigned)NW; k++) {
        const uint32_t q = part[par][k];
        if (k < wave_id()) base += q;
        all += q;
    }
    par ^= 1u;
    total = all;
    return base + incl - v;
}

__device__ __forceinline__ unsigned lanes_below(unsigned long long peers)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
}

// rank of this lane's key among the keys with the same digit that its wave has seen so
// far in this tile (earlier rounds, then lower lanes of this round)
template <bool RANK_ATOMIC>
__device__ __forceinline__ uint32_t rank_round(unsigned d, unsigned long long* flags_w, uint32_t* cnt_w,
                                               unsigned long long mybit)
{
    unsigned long long peers;
    if (RANK_ATOMIC) {
        atomicOr(&flags_w[d], mybit);
        wave_sync();
        peers = flags_w[d];
    } else {
        peers = ~0ull;
#pragma unroll
        for (int b = 0; b < kRadixBitsDev; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long vote = __ballot(bit);
            peers &= bit ? vote : ~vote;
        }
    }
    const uint32_t pre = cnt_w[d];
    wave_sync();
    const unsigned below = lanes_below(peers);
    if (below == 0) {
        if (RANK_ATOMIC) flags_w[d] = 0ull;
        cnt_w[d] = pre + (uint32_t)__popcll(peers);
    }
    wave_sync();
    return pre + below;
}

// rank_round<true> with 16-bit counts (tiles of < 65536 elements: k_radix_sweep, k_tiny_sa)
__device__ __forceinline__ uint32_t rank_round16(unsigned d, unsigned long long* flags_w, uint16_t* cnt_w, unsigned long long mybit)
{
    atomicOr(&flags_w[d], mybit);
    wave_sync();
    const unsigned long long peers = flags_w[d];
    const uint32_t pre = cnt_w[d];
    wave_sync();
    const unsigned below = lanes_below(peers);
    if (below == 0) {
        flags_w[d] = 0ull;
        cnt_w[d] = (uint16_t)(pre + (uint32_t)__popcll(peers));
    }
    wave_sync();
    return pre + below;
}

// ---- packed text -------------------------------------------------------------------
// The text is re-coded once per build into dense symbol codes of `bits` bits and
// packed big-endian, spw = floor(32/bits) symbols per 32-bit word (DNA: 16 symbols per
// word, 25 MB for 100 MB of text -- resident in L2 / Infinity Cache).  Word j holds
// positions [j*spw, (j+1)*spw) in its low kbits = bits*spw bits; positions past the
// end of the text read as 0 and the array carries 3 extra zero words, so a key can be
// fetched at any position < n with 2-3 aligned loads and a funnel shift.
// Occupancy request of a kernel (waves per SIMD: the compiler fits the registers to it).  The CPU emulator of the tests
// (tests/emu/hip/hip_runtime.h) defines it away.
// A per-lane value the compiler must not reason about (it stops loop-invariant address arithmetic from being hoisted out of
// a tile loop that has no registers to keep it in).  Nothing to the CPU emulator.
#ifdef SFX_EMULATED
#define SFX_OPAQUE_VGPR(x) ((void)0)
#else
#define SFX_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))
#endif
#ifndef SFX_WAVES_PER_EU
#define SFX_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif
// A value every lane of the wave holds alike (a ticket read from LDS), moved to a scalar register: addresses built on it take a
// scalar base and leave the vector registers to the data.  The value itself to the CPU emulator.
#ifdef SFX_EMULATED
#define SFX_WAVE_UNIFORM(x) (x)
#else
#define SFX_WAVE_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif

struct PackedText {
    const uint32_t* words;
    uint64_t n;
    int bits;
    int spw;            // symbols per word
    int kbits;          // bits * spw  (<= 32)
    double inv_spw;     // 1.0 / spw
};

// p / spw for any spw in 1..32, exact for p < 2^52: (p + 0.5) / spw is at least
// 0.5/32 away from every integer, far more than the rounding error of the product.
// (Round 4 tried a shift for power-of-two spw behind a uniform branch: the text-fed radix pass gained 2 % (0.449 -> 0.440 ms),
// but the branch puts every key's loads into a basic block of their own, the gathers of k_small_groups no longer overlap,
// and the direct pass lost 30 % at 8 virtual ranks (0.55 -> 0.72 ms) -- A/B against round 3's library.  Branch-free it stays.)
__device__ __forceinline__ uint64_t packed_word_index(const PackedText& t, uint64_t p)
{
    return (uint64_t)(((double)p + 0.5) * t.inv_spw);
}

__device__ __forceinline__ unsigned packed_word_offset(const PackedText& t, uint64_t p, uint64_t q)
{
    return (unsigned)(p - q * (uint64_t)t.spw);
}
// the spw symbols starting at position p, as a kbits-bit big-endian number
__device__ __forceinline__ uint32_t packed_key32(const PackedText& t, uint64_t p)
{
    const uint64_t q = packed_word_index(t, p);
    const unsigned off = packed_word_offset(t, p, q);
    // (the two words as ONE 8-byte load: a gather is 64 different lines to the wave, and every load instruction
    // looks each of them up in the L1 again)
    uint64_t pair;
    __builtin_memcpy(&pair, t.words + q, 8);
    const uint64_t both = ((pair & 0xFFFFFFFFull) << t.kbits) | (pair >> 32);
    const uint64_t mask = (1ull << t.kbits) - 1ull;
    return (uint32_t)((both >> (((unsigned)t.spw - off) * (unsigned)t.bits)) & mask);
}
// the 2*spw symbols starting at p, as a 2*kbits-bit number
__device__ __forceinline__ uint64_t packed_key64(const PackedText& t, uint64_t p)
{
    const uint64_t q = packed_word_index(t, p);
    const unsigned off = packed_word_offset(t, p, q);
    struct { uint32_t w[3]; } three;                                  // (one 12-byte load)
    __builtin_memcpy(&three, t.words + q, 12);
    const uint64_t w0 = three.w[0], w1 = three.w[1], w2 = three.w[2];
    const uint64_t mask = (1ull << t.kbits) - 1ull;
    const unsigned sh = ((unsigned)t.spw - off) * (unsigned)t.bits;
    const uint64_t a = (((w0 << t.kbits) | w1) >> sh) & mask;
    const uint64_t b = (((w1 << t.kbits) | w2) >> sh) & mask;
    return (a << t.kbits) | b;
}
// Keys of consecutive positions: one word-index computation, then a step per position
// (the next word is fetched only when the window crosses a word boundary).
template <class KeyT>
struct PackedWalk {
    uint64_t q = 0;
    unsigned off = 0;
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    __device__ __forceinline__ void init(const PackedText& t, uint64_t p)
    {
        q = packed_word_index(t, p);
        off = packed_word_offset(t, p, q);
        w0 = t.words[q];
        w1 = t.words[q + 1];
        w2 = sizeof(KeyT) == 8 ? t.words[q + 2] : 0u;
    }
    __device__ __forceinline__ KeyT key(const PackedText& t) const
    {
        const uint64_t mask = (1ull << t.kbits) - 1ull;
        const unsigned sh = ((unsigned)t.spw - off) * (unsigned)t.bits;
        const uint64_t a = (((w0 << t.kbits) | w1) >> sh) & mask;
        if (sizeof(KeyT) == 4) return (KeyT)a;
        const uint64_t b = (((w1 << t.kbits) | w2) >> sh) & mask;
        return (KeyT)((a << t.kbits) | b);
    }
    // only while the next position is still < n (the array carries 3 zero words past the end)
    __device__ __forceinline__ void step(const PackedText& t)
    {
        if (++off == (unsigned)t.spw) {
            off = 0;
            q++;
            w0 = w1;
            if (sizeof(KeyT) == 8) {
                w1 = w2;
                w2 = t.words[q + 2];
            } else {
                w1 = t.words[q + 1];
            }
        }
    }
};

template <class KeyT> __device__ __forceinline__ KeyT packed_key(const PackedText& t, uint64_t p);
template <> __device__ __forceinline__ uint32_t packed_key<uint32_t>(const PackedText& t, uint64_t p)
{
    return packed_key32(t, p);
}
template <> __device__ __forceinline__ uint64_t packed_key<uint64_t>(const PackedText& t, uint64_t p)
{
    return packed_key64(t, p);
}

// ---- order-preserving compressed keys ----------------------------------------------------------
// A fixed-width key spends ceil(log2 sigma) bits on every symbol; natural-language text has 4-5 bits of
// entropy per symbol.  An ALPHABETIC prefix code (codes ordered like the symbols, none a prefix of another:
// Hu-Tucker / Garsia-Wachs trees) keeps the order of the strings -- two suffixes compare like the
// concatenations of their symbols' codes -- so the 64 key bits of the initial sort hold ~14 symbols of English
// instead of 8 for one more radix pass.  The buckets then share a VARIABLE number of symbols (as many codes as
// fit the key completely, at most kHtMaxSym): that number is recovered from the key itself (ht_depth) by the
// bucket pass and kept per bucket (Hd, sfx_sa.hip).
// ent[s], s = dense symbol code: code left-aligned in bits 31..5, length (1..27) in bits 4..0; the smallest symbol's
// code is all zeros, so the zero padding past the end of the text reads as that symbol.
constexpr int kHtMaxLen = 12;                       // (= kHtFastBits: every code ends inside one fast-table window)
constexpr unsigned kHtMaxSym = 16;                  // symbols a key is made from at most
// t12[w], w = the next 12 key bits: bit k set = a symbol's code ENDS after k + 1 of them (greedy decode of w, made on
// the host: ht_build) -- two or three symbols per look-up, and where fewer than 12 key bits are left the ends beyond
// them are masked off.  No code is longer than 12 bits (ht_build floors the counts until that holds).
constexpr int kHtFastBits = 12;
// Bits of a compressed key that are sorted (and mean anything): the top kHtKeyBits of the 64-bit word, the rest is zero.
// (56 = seven radix passes instead of eight was measured on 1 GB texts: one 7 ms pass less, but 12.1 symbols of
// English-like text per key instead of 13.8 leave 649 M suffixes to the rounds instead of 540 M -- 116.9 against 111.4 ms;
// mixed-script UTF-8 229 against 223, near-duplicate documents 399 against 390.  All 64 bits it is.)
constexpr int kHtKeyBits = 64;
// Context codes (round 6, k_ht_keys_ctx): one code table per class of the preceding symbol.  Device layout behind the order-0
// tables: at word kHtCtxOff the class of every dense symbol (256 bytes), then kHtCtxClasses x 256 entries (code | length).
constexpr int kHtCtxClasses = 16;
constexpr int kHtCtxSigmaMax = 160;                  // alphabets the bigram counts and the class tables are sized for
constexpr unsigned kHtCtxOff = 256 + 64 + (1u << 12) / 2;     // = kHtTableWords + the fast table (sfx_sa.hip)
constexpr unsigned kHtCtxWords = 64 + kHtCtxClasses * 256;
constexpr int kHtCtxCountBits = 4;                   // low bits of a context key: the number of symbols it holds
static_assert(kHtKeyBits % 8 == 0 && kHtKeyBits >= 32 && kHtKeyBits <= 64, "whole radix digits");
// the kHtFastBits key bits from bit `used` on (zeros past the key's end), used < 64.  32-bit funnel shifts: the 64-bit
// shift pair (key << used) >> 52 is two quarter-rate instructions, and the decode loops run this once per two or three
// symbols of every key they look at (round 4: k_groups_reduce with compressed keys 11.2 -> see DESIGN.md)
__device__ __forceinline__ unsigned ht_window(uint64_t key, unsigned used)
{
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    const uint32_t a = used < 32u ? hi : lo, b = used < 32u ? lo : 0u;
    const unsigned sh = used & 31u;
    // (a:b) << sh, top 32 bits: __funnelshift_l(b, a, sh) = (a << sh) | (b >> (32 - sh)), sh in 0 .. 31
    return __funnelshift_l(b, a, sh) >> (32 - kHtFastBits);
}
// one step of the decode of `key` from bit `used` on; returns false when the key is exhausted
__device__ __forceinline__ bool ht_depth_step(uint64_t key, unsigned& used, unsigned& cnt, const uint16_t* t12)
{
    if (used >= (unsigned)kHtKeyBits || cnt >= kHtMaxSym) return false;
    unsigned ends = t12[ht_window(key, used)];
    const unsigned left = (unsigned)kHtKeyBits - used;
    if (left < (unsigned)kHtFastBits) ends &= (1u << left) - 1u;
    if (!ends) return false;                             // (no code ends inside what is left; with 12 bits left one always does)
    cnt += (unsigned)__popc(ends);
    used += 32u - (unsigned)__clz((int)ends);
    return true;
}
// Symbols two DIFFERENT compressed keys share: the code words that end inside their common leading bits (the decodes of the
// two keys agree up to there, so either key will do); with common == kHtKeyBits: the symbols the key holds (ht_depth).
// N keys per lane at once.  The body of the loop is predicated, not branched: a per-key `if (still going)` inside the
// per-lane `while` cost ~360 instructions per key (8 divergent branches per step, exec mask saved and restored around
// each) -- 9.3 ms per 10^9 keys in k_groups_reduce, 2900 wave instructions per 512 keys; one back edge per step: 4.4 ms.
template <int N>
__device__ __forceinline__ void ht_common_n(const uint64_t (&key)[N], const unsigned (&common)[N], const uint16_t* t12, uint32_t (&sym)[N])
{
    unsigned used[N], cnt[N];
#pragma unroll
    for (int j = 0; j < N; j++) { used[j] = 0; cnt[j] = 0; }
    bool more = true;
    while (more) {
        more = false;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const unsigned lim = common[j] < (unsigned)kHtKeyBits ? common[j] : (unsigned)kHtKeyBits;
            const bool live = used[j] < lim && cnt[j] < kHtMaxSym;
            unsigned ends = t12[ht_window(key[j], used[j] & 63u)];
            const unsigned left = live ? lim - used[j] : 0u;
            if (left < (unsigned)kHtFastBits) ends &= (1u << left) - 1u;
            const bool ok = live && ends != 0u;
            cnt[j] += ok ? (unsigned)__popc(ends) : 0u;
            used[j] += ok ? 32u - (unsigned)__clz((int)ends) : 0u;
            more |= ok;
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) sym[j] = cnt[j] < kHtMaxSym ? cnt[j] : kHtMaxSym;
}
// depths of N keys at once (bit j of `want`: key j is wanted): the symbols each key holds
template <int N>
__device__ __forceinline__ void ht_depth_n(const uint64_t (&key)[N], unsigned want, const uint16_t* t12, uint32_t (&depth)[N])
{
    unsigned common[N];
#pragma unroll
    for (int j = 0; j < N; j++) common[j] = ((want >> j) & 1u) ? (unsigned)kHtKeyBits : 0u;
    ht_common_n<N>(key, common, t12, depth);
}

// number of bits needed to represent values in [0, v]
__host__ __device__ inline int bits_for(uint64_t v)
{
    int b = 0;
    while (v) { b++; v >>= 1; }
    return b ? b : 1;
}

}  // namespace sfx
