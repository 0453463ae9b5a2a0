// sfx_tiny.hip -- SuffixTable::new (src/table.rs:78-85 -> sais_table :378-386) for texts of up to kTinyMax bytes: ONE
// launch of ONE workgroup.
//
// The general build is ~15 launches and three host round trips whatever the length: 170 us for the 10 KB fixture of BASELINE
// config 1 (tests/AP009048_10000.fasta; README.md:115 quotes 713 us for the reference's sais on it).  A text of a few
// thousand symbols fits the LDS of one CU whole, so here a single 1024-thread workgroup does everything:
//   alphabet    byte presence -> dense symbol codes (what Bins::find_sizes :686-704 gives the reference), `bits` =
//               ceil(log2 sigma) per symbol, the codes packed into one LDS bit stream (zeros past the end);
//   sort        LSD radix sort of the suffix indices (16 bits each, ping-pong in LDS) by the first K bits of their symbols,
//               8 bits per pass with the match-mask ranking of the device-wide passes (rank_round16): K = 2 log2 n + 4,
//               rounded up to whole bytes, at most 64 (8-bit symbols: 64) -- for the fixture 4 passes;
//   ties        suffixes whose first K bits agree (repeats; suffixes that end inside the key and look like a run of the
//               smallest symbol) form runs of equal keys: a run of up to kTinyGroupMax members is put in order by one
//               thread comparing the suffixes themselves on the LDS stream (proper prefix first: the order of naive_table
//               :367-376); a longer run, or a thread that has compared more than kTinyStepCap 64-bit windows, gives up:
//               status = 1 and the caller runs the general build (unary and periodic texts end there -- correct either way).
// No MFMA, no global traffic but the text in and the table out.
#include "sfx_host.hpp"

namespace sfx {

constexpr int kTinyNW = 16, kTinyThreads = kTinyNW * kWave;
constexpr uint32_t kTinyMax = 16384;
constexpr int kTinyKPT = (int)(kTinyMax / kTinyThreads);
constexpr uint32_t kTinyGroupMax = 32;
constexpr uint32_t kTinyMaxBytes8 = 4096;              // more than 16 different bytes: texts above this go to the general build at once
constexpr uint32_t kTinyStepCap = 1u << 13;

struct TinySmem {
    alignas(16) uint8_t raw[kTinyMax + 16];                     // the text as it came (one coalesced load per thread)
    uint32_t stream[kTinyMax / 4 + 8];              // symbol codes, `bits` each, as one bit stream: word k = bits [32 k, 32 k + 32), first bit on top; zero tail
    uint16_t idx[2][kTinyMax];                      // suffix indices, ping-pong
    unsigned long long flags[kTinyNW][kRadix];      // match masks of the ranking; afterwards: one head flag byte per rank
    uint16_t cnt[kTinyNW][kRadix];
    uint32_t present[8];
    uint8_t lut[256];
    uint32_t part[2][kTinyNW];
    uint32_t give_up;
};

// the 64 stream bits from symbol `i` on (zeros past the end: the stream carries a tail of zero words): three aligned words
// and two funnel shifts
__device__ __forceinline__ uint64_t tiny_window(const uint32_t* stream, uint32_t i, unsigned bits)
{
    const uint32_t o = i * bits;
    const uint32_t q = o >> 5;
    const unsigned sh = o & 31u;
    const uint32_t w0 = stream[q], w1 = stream[q + 1], w2 = stream[q + 2];
    return ((uint64_t)__funnelshift_l(w1, w0, sh) << 32) | (uint64_t)__funnelshift_l(w2, w1, sh);
}
// the 8 stream bits from bit `o` on
__device__ __forceinline__ unsigned tiny_digit(const uint32_t* stream, uint32_t o)
{
    const uint32_t q = o >> 5;
    return __funnelshift_l(stream[q + 1], stream[q], o & 31u) >> 24;
}
// suffix a < suffix b (a != b): bytewise order of the symbol codes, a proper prefix first.  `steps` counts 64-bit windows.
__device__ __forceinline__ bool tiny_less(const uint32_t* stream, uint32_t n, unsigned bits, uint32_t a, uint32_t b, uint32_t& steps)
{
    const uint32_t spw = 64u / bits;                        // symbols per window
    const uint32_t la = n - a, lb = n - b, lmin = la < lb ? la : lb;
    uint32_t d = 0;
    while (d < lmin) {
        const uint64_t wa = tiny_window(stream, a + d, bits), wb = tiny_window(stream, b + d, bits);
        steps++;
        if (wa != wb) {
            const uint32_t same = (uint32_t)__clzll((long long)(wa ^ wb)) / bits;       // equal leading symbols of the two windows
            if (d + same >= lmin) return la < lb;           // they part beyond the shorter suffix's end: it is a prefix
            return wa < wb;
        }
        d += spw;
    }
    return la < lb;
}

__global__ void __launch_bounds__(kTinyThreads, 1)
k_tiny_sa(const uint8_t* __restrict__ text, uint32_t n, uint32_t* __restrict__ sa, uint32_t* __restrict__ status)
{
    __shared__ TinySmem s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
#ifdef SFX_TINY_TIMERS
    unsigned long long t_prev = wall_clock64(); int t_slot = 2;
#define TT() do { __syncthreads(); if (tid == 0) { unsigned long long now_ = wall_clock64(); status[t_slot++] = (uint32_t)(now_ - t_prev); t_prev = now_; } } while (0)
#else
#define TT() do {} while (0)
#endif
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    if (tid < 8) s.present[tid] = 0u;
    if (tid == 0) s.give_up = 0u;
    if (owner) {
#pragma unroll
        for (int k = 0; k < kTinyNW; k++) { s.cnt[k][tid] = 0; s.flags[k][tid] = 0ull; }
    }
    __syncthreads();
    // ---- the text into LDS: 16 bytes per thread and load (the kernel touches global memory twice: here and for the table)
    {
        const bool aligned = (reinterpret_cast<uintptr_t>(text) & 15u) == 0;
        for (uint32_t i = tid * 16u; i < n; i += kTinyThreads * 16u) {
            if (aligned && i + 16u <= n) {
                *reinterpret_cast<uint4*>(s.raw + i) = *reinterpret_cast<const uint4*>(text + i);
            } else {
                for (uint32_t k = i; k < i + 16u && k < n; k++) s.raw[k] = text[k];
            }
        }
    }
    __syncthreads();
    TT();
    // ---- alphabet: which byte values occur.  Every thread ORs its bytes into a private 256-bit set, a wave reduces its
    // 64 sets with shuffles, 16 waves then OR 8 words each into LDS (an LDS atomic per byte on 8 addresses was the whole
    // kernel's first 17 us: 64 lanes on one word serialise)
    {
        uint32_t mset[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i = tid; i < n; i += kTinyThreads) {
            const unsigned b = s.raw[i];
            const uint32_t bit = 1u << (b & 31u);
#pragma unroll
            for (int k = 0; k < 8; k++) mset[k] |= (b >> 5) == (unsigned)k ? bit : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mset[k] |= __shfl_xor(mset[k], d);
            if (lane == 0 && mset[k]) atomicOr(&s.present[k], mset[k]);
        }
    }
    __syncthreads();
    TT();
    unsigned sigma = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sigma += (unsigned)__popc(s.present[k]);
    if (owner) {
        unsigned below = 0;
        for (unsigned k = 0; k < (tid >> 5); k++) below += (unsigned)__popc(s.present[k]);
        below += (unsigned)__popc(s.present[tid >> 5] & ((1u << (tid & 31u)) - 1u));
        s.lut[tid] = (uint8_t)below;
    }
    unsigned bits = 1;
    while ((1u << bits) < sigma) bits++;                                     // 1 .. 8
    // (a large alphabet in a text of more than kTinyMaxBytes8 symbols is natural-language text more often than not: words
    // that repeat dozens of times make runs of equal keys this kernel does not order -- give up before any work is done)
    if (sigma > 16u && n > kTinyMaxBytes8) {
        if (tid == 0) *status = 1u;
        return;
    }
    const uint32_t nwords = (n * bits + 31u) >> 5;
    __syncthreads();
    // ---- the bit stream (word k = the codes of the symbols that overlap bits [32 k, 32 k + 32)) and the identity permutation
    for (uint32_t k = tid; k < nwords + 8u; k += kTinyThreads) {
        uint32_t v = 0;
        if (k < nwords) {
            const uint32_t b0 = k * 32u;
            for (uint32_t j = b0 / bits; j * bits < b0 + 32u; j++) {
                const uint32_t c = j < n ? (uint32_t)s.lut[s.raw[j]] : 0u;
                const int up = (int)(b0 + 32u) - (int)(j * bits + bits);     // code's lowest bit, counted from the word's bit 0
                v |= up >= 0 ? c << up : c >> (-up);
            }
        }
        s.stream[k] = v;
    }
    for (uint32_t i = tid; i < n; i += kTinyThreads) s.idx[0][i] = (uint16_t)i;
    __syncthreads();
    TT();
    // ---- LSD radix sort by the first K bits of every suffix
    int lg = 0;
    while ((1u << lg) < n) lg++;
    int K = ((2 * lg + 4 + 7) / 8) * 8;                                      // (n^2 / 2^(K + 1) pairs tie by chance: < 1/16; a pass costs 8.5 us)
    if (K > 64 || sigma > 16u) K = 64;                                       // (large alphabets: the whole window)
    const int passes = K / 8;
    const unsigned kpt = (n + kTinyThreads - 1) / kTinyThreads;              // elements per thread, <= kTinyKPT
    for (int p = 0; p < passes; p++) {
        const uint16_t* src = s.idx[p & 1];
        uint16_t* dst = s.idx[(p + 1) & 1];
        uint32_t el[kTinyKPT], dg[kTinyKPT], pos[kTinyKPT];
        // (all elements and digits first -- independent LDS reads in flight together --, then the ranking, whose rounds are
        // dependent chains through the match masks)
#pragma unroll
        for (int r = 0; r < kTinyKPT; r++) {
            el[r] = 0xFFFFFFFFu;
            if ((unsigned)r < kpt) {                                         // (uniform)
                const uint32_t e = w * (kWave * kpt) + (unsigned)r * kWave + lane;
                if (e < n) el[r] = (uint32_t)src[e];
            }
        }
#pragma unroll
        for (int r = 0; r < kTinyKPT; r++)
            dg[r] = el[r] != 0xFFFFFFFFu ? tiny_digit(s.stream, el[r] * bits + (uint32_t)(K - 8 * (p + 1))) : 255u;   // padding sorts last
        // (ranking by 8 wave ballots per round with the counts read before the leader's add, so that no round waits for the
        // LDS, was measured: 9.8 us per pass against 8.5 with the match masks -- 60 instructions per round on one CU's
        // four SIMDs cost what the three LDS round trips of a mask round cost)
#pragma unroll
        for (int r = 0; r < kTinyKPT; r++)
            if ((unsigned)r < kpt) pos[r] = rank_round16(dg[r], s.flags[w], s.cnt[w], mybit);
        __syncthreads();
        {
            uint32_t c[kTinyNW], total = 0;
#pragma unroll
            for (int k = 0; k < kTinyNW; k++) {
                c[k] = owner ? s.cnt[k][tid] : 0u;
                total += c[k];
            }
            const uint32_t ex = block_scan_excl_1b<kTinyNW>(total, s.part, par);
            if (owner) {
                uint32_t run = ex;
#pragma unroll
                for (int k = 0; k < kTinyNW; k++) {
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kTinyKPT; r++) {
            if ((unsigned)r < kpt) {
                const uint32_t at = pos[r] + s.cnt[w][dg[r]];
                if (at < n && el[r] != 0xFFFFFFFFu) dst[at] = (uint16_t)el[r];
            }
        }
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int k = 0; k < kTinyNW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
    }
    TT();
    // ---- runs of equal keys
    uint16_t* fin = s.idx[passes & 1];
    uint8_t* head = reinterpret_cast<uint8_t*>(&s.flags[0][0]);              // (32 KiB: the masks are idle now)
    const int drop = 64 - K;
    for (uint32_t r = tid; r < n; r += kTinyThreads) {
        bool h = true;
        if (r) h = (tiny_window(s.stream, fin[r], bits) >> drop) != (tiny_window(s.stream, fin[r - 1], bits) >> drop);
        head[r] = h ? 1 : 0;
    }
    __syncthreads();
    TT();
    for (uint32_t r = tid; r < n; r += kTinyThreads) {
        if (!head[r] || r + 1 >= n || head[r + 1]) continue;                 // not the first of a run of two or more
        uint32_t g = 2;
        while (r + g < n && !head[r + g] && g <= kTinyGroupMax) g++;
        if (g > kTinyGroupMax) { s.give_up = 1u; continue; }
        uint32_t steps = 0;
        for (uint32_t i = 1; i < g && steps <= kTinyStepCap; i++) {          // insertion sort of fin[r .. r + g)
            const uint32_t x = fin[r + i];
            uint32_t j = i;
            while (j > 0 && steps <= kTinyStepCap && tiny_less(s.stream, n, bits, x, fin[r + j - 1], steps)) {
                fin[r + j] = fin[r + j - 1];
                j--;
            }
            fin[r + j] = (uint16_t)x;
        }
        if (steps > kTinyStepCap) s.give_up = 1u;
    }
    __syncthreads();
    TT();
    const bool bad = s.give_up != 0u;
    if (tid == 0) *status = bad ? 1u : 0u;
    if (bad) return;
    for (uint32_t r = tid; r < n; r += kTinyThreads) sa[r] = (uint32_t)fin[r];
}

uint64_t tiny_max_default() { return kTinyMax; }
// (SFX_TINY=0 in builds with the development hooks: the emulator's variant runs keep small texts on the general path)
static uint64_t g_tiny_limit = kTinyMax;
uint64_t tiny_limit()
{
    const char* e = dev_env("SFX_TINY");
    if (e && atoi(e) == 0) return 0;
    return __atomic_load_n(&g_tiny_limit, __ATOMIC_RELAXED);
}
void tiny_set_limit(uint64_t n) { __atomic_store_n(&g_tiny_limit, n > kTinyMax ? (uint64_t)kTinyMax : n, __ATOMIC_RELAXED); }

// *done = false: the text is not one for this path (too long, or a run of equal keys too long to order by comparison)
int tiny_build_sa_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* ws, hipStream_t st, bool* done)
{
    *done = false;
    if (n < 2 || n > kTinyMax || n > tiny_limit()) return SFX_OK;
    uint32_t* status = reinterpret_cast<uint32_t*>(ws);
    SFX_LAUNCH("tiny_sa", (double)n * 5.0, k_tiny_sa, 1, kTinyThreads, st, d_text, (uint32_t)n, d_sa, status);
    uint32_t host = 1;
    SFX_TRY(read_back(&host, status, sizeof(host), st));
    *done = host == 0;
    return SFX_OK;
}

}  // namespace sfx
