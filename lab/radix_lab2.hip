// lab/radix_lab2.hip -- DEVELOPMENT ONLY (not part of the product).  Round 4: what bounds the one-sweep pass, and
// candidate geometries for it, measured on 100 M random E64 / KV elements against the shipped kernel.
//   1. store probes: streaming writes from a limited number of workgroups (is the 8 B/clk a CU limit or a chip limit?),
//      partial lines, and runs of R elements at unaligned places (what a longer run is worth);
//   2. k_lean: the shipped tile engine without the chunked-schedule baggage (16-bit counts, no second key array) so that
//      two 512-thread workgroups fit one CU (<= 80 KB of LDS, <= 128 VGPRs);
//   3. k_pipe: one 1024-thread workgroup per CU, software-pipelined: the stores of tile i are interleaved with the
//      ranking of tile i+1 (match masks outside the staging buffer).
#include <stdio.h>
#include <vector>
#include <string>
#include "../suffix_amd/csrc/sfx_radix.hip"

namespace sfx {
bool profile_on() { return false; }
void profile_begin(const char*, hipStream_t, double) {}
void profile_end(hipStream_t) {}
void note_hip_error(hipError_t e, const char* what, const char*, int) { fprintf(stderr, "HIP error %d at %s\n", (int)e, what); }

// ---------------------------------------------------------------------------------------------------------------
// probes
__global__ void k_fill(uint64_t* a, uint64_t n, int mode)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        if (mode == 1) {                      // skewed digits (Zipf-like: square of a uniform)
            const uint64_t u = z >> 40;       // 24 bits
            z = (((u * u) >> 24) << 40) | (z & 0xFFFFFFFFFFull);
        }
        a[i] = (z << 32) | (uint32_t)i;
    }
}
__global__ void k_fill_kv(uint64_t* k, uint32_t* v, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        k[i] = z;
        v[i] = (uint32_t)i;
    }
}
// persistent streaming write: `gridDim.x` workgroups share n16 16-byte words
__global__ void __launch_bounds__(256) k_write16(uint4* __restrict__ out, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) out[i] = uint4{(unsigned)i, 1u, 2u, 3u};
}
// every 128-byte line gets `h` of its 16 eight-byte words (the first h), one line per 16 lanes
__global__ void __launch_bounds__(256) k_write_partial(uint64_t* __restrict__ out, uint64_t nlines, int h)
{
    const uint64_t stride = (uint64_t)gridDim.x * 16;
    const unsigned j = threadIdx.x & 15u;
    for (uint64_t l = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4); l < nlines; l += stride)
        if ((int)j < h) out[l * 16 + j] = l + j;
}
// runs of R consecutive 8-byte elements; run r goes to slot perm(r) * R + misalign.  Lanes write consecutive elements,
// as the radix pass's output loop does (a wave's store covers the end of one run and the start of the next).
__global__ void __launch_bounds__(1024) k_write_runs(uint64_t* __restrict__ out, uint64_t m, unsigned R, unsigned nruns_pow2_mask,
                                                     uint64_t nruns, unsigned misalign)
{
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < m; i += stride) {
        const uint64_t r = i / R;
        const unsigned e = (unsigned)(i - r * R);
        // a bijection on [0, nruns): multiply by an odd constant modulo a power of two, cycle-walk into range
        uint64_t p = r;
        do { p = (p * 0x9E3779B1ull + 12345ull) & nruns_pow2_mask; } while (p >= nruns);
        out[p * R + e + misalign] = i;
    }
}

__global__ void __launch_bounds__(1024) k_write_runs32(uint32_t* __restrict__ out, uint64_t m, unsigned R, unsigned nruns_pow2_mask,
                                                       uint64_t nruns, unsigned misalign)
{
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < m; i += stride) {
        const uint64_t r = i / R;
        const unsigned e = (unsigned)(i - r * R);
        uint64_t p = r;
        do { p = (p * 0x9E3779B1ull + 12345ull) & nruns_pow2_mask; } while (p >= nruns);
        out[p * R + e + misalign] = (uint32_t)i;
    }
}
// ---------------------------------------------------------------------------------------------------------------
// sources / sinks of the lab kernels (KV as two arrays, or as 12-byte records k.lo, k.hi, v)
struct LSrcE64 {
    static constexpr bool kHasVal = false;
    const uint64_t* in;
    __device__ __forceinline__ void load(uint64_t i, uint64_t& k, uint32_t&) const { k = in[i]; }
};
struct LDstE64 {
    uint64_t* out;
    __device__ __forceinline__ void store(uint32_t d, uint64_t k, uint32_t) const { out[d] = k; }
};
struct LSrcKV {
    static constexpr bool kHasVal = true;
    const uint64_t* k; const uint32_t* v;
    __device__ __forceinline__ void load(uint64_t i, uint64_t& key, uint32_t& val) const { key = k[i]; val = v[i]; }
};
struct LDstKV {
    uint64_t* k; uint32_t* v;
    __device__ __forceinline__ void store(uint32_t d, uint64_t key, uint32_t val) const { k[d] = key; v[d] = val; }
};
struct alignas(4) Rec12 { uint32_t lo, hi, v; };
struct LSrcRec {
    static constexpr bool kHasVal = true;
    const Rec12* in;
    __device__ __forceinline__ void load(uint64_t i, uint64_t& key, uint32_t& val) const
    {
        const Rec12 r = in[i];
        key = ((uint64_t)r.hi << 32) | r.lo; val = r.v;
    }
};
struct LDstRec {
    Rec12* out;
    __device__ __forceinline__ void store(uint32_t d, uint64_t key, uint32_t val) const { out[d] = Rec12{(uint32_t)key, (uint32_t)(key >> 32), val}; }
};
__global__ void k_kv_to_rec(const uint64_t* k, const uint32_t* v, Rec12* r, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) r[i] = Rec12{(uint32_t)k[i], (uint32_t)(k[i] >> 32), v[i]};
}
__global__ void k_cmp_rec(const uint64_t* k, const uint32_t* v, const Rec12* r, uint64_t n, unsigned long long* bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        b += (r[i].lo != (uint32_t)k[i]) || (r[i].hi != (uint32_t)(k[i] >> 32)) || (r[i].v != v[i]);
    if (b) atomicAdd(bad, b);
}
__global__ void k_cmp64(const uint64_t* a, const uint64_t* b, uint64_t n, unsigned long long* bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}
__global__ void k_cmp32(const uint32_t* a, const uint32_t* b, uint64_t n, unsigned long long* bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

// (rank_round16 -- the match-mask ranking with 16-bit counts -- started here and now lives in sfx_device.hpp)
template <int KPT, bool HAS_VAL, int NW>
struct LeanSmem {
    uint64_t stage[NW * kWave * KPT];                    // (match masks of the ranking alias its first NW * 2 KiB)
    uint32_t stage_v[HAS_VAL ? NW * kWave * KPT : 1];
    uint16_t cnt[NW][kRadix];
    uint32_t off[kRadix];
    uint32_t part[2][NW];
    uint32_t ticket;
};

// ---- k_lean: the one-sweep tile engine, lean ---------------------------------------------------------------------
// MODE bits (timing experiments): 1 no stores, 2 no look-back, 4 no ranking, 8 no loads, 16 no LDS reorder, 32 tile leaves as one run
template <class Src, class Dst, int KPT, int NW, int MINW, int MODE = 0, int KO = 0>
__global__ void __launch_bounds__(NW * kWave, MINW)
k_lean(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
       uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    static_assert(kWave * KPT >= kRadix, "masks alias the stage");
    __shared__ LeanSmem<KPT, HAS_VAL, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    __syncthreads();
    for (;;) {
        if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile_no = s.ticket;
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            key[r] = ~0ull;
            if (HAS_VAL) val[r] = 0u;
            if (MODE & 8) { key[r] = ((tile + idx) * 0x9E3779B1ull) << 24; if (HAS_VAL) val[r] = idx; }
            else if (idx < nvalid) src.load(tile + idx, key[r], val[HAS_VAL ? r : 0]);
        }
        if (MODE & 4) {
#pragma unroll
            for (int r = 0; r < KPT; r++) pos[r] = (r * kWave + lane) & 3u;
            if (lane < 4) { for (int k = 0; k < 64; k++) s.cnt[w][k * 4 + lane] = (uint16_t)(KPT * 16 / 64); }
        } else {
#pragma unroll
            for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
            wave_sync();
#pragma unroll
            for (int r = 0; r < KPT; r++) pos[r] = rank_round16((unsigned)(key[r] >> shift) & 255u, my_flags, s.cnt[w], mybit);
        }
        __syncthreads();
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
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
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                if (!(MODE & 2)) lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
        }
        __syncthreads();
        if (!(MODE & 16)) {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned p = pos[r] + s.cnt[w][(unsigned)(key[r] >> shift) & 255u];
                s.stage[p] = key[r];
                if (HAS_VAL) s.stage_v[p] = val[r];
            }
        }
        if (owner) {
            if (MODE & 2) s.off[tid] = my_head + (uint32_t)((uint64_t)tile_no * 44u) - tile_ex;
            else s.off[tid] = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0) - tile_ex;
            if (MODE & 32) s.off[tid] = (uint32_t)tile;                         // timing: the sorted tile leaves as one run
        }
        __syncthreads();
        constexpr int kOut = KO ? KO : ((KPT % 8 == 0) ? 8 : ((KPT % 6 == 0) ? 6 : ((KPT % 4 == 0) ? 4 : ((KPT % 3 == 0) ? 3 : KPT))));
        static_assert(KPT % kOut == 0, "whole output batches");
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
            if (!(MODE & 16)) {
#pragma unroll
                for (int r = r0; r < r0 + kOut; r++) {
                    key[r] = s.stage[r * kThreads + tid];
                    if (HAS_VAL) val[r] = s.stage_v[r * kThreads + tid];
                }
            }
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) pos[r] = s.off[(unsigned)(key[r] >> shift) & 255u] + (r * kThreads + tid);
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++)
                if ((unsigned)(r * kThreads) + tid < nvalid && (!(MODE & 1) || pos[r] == 0x7FFFFFF1u)) dst.store(pos[r], key[r], HAS_VAL ? val[r] : 0u);
        }
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
    }
}

// k_leant: k_lean with per-phase cycle counters of wave 0 (clock64)
// MODE bits: 1 no stores, 2 no look-back, 4 no ranking, 8 no loads, 16 no LDS reorder, 32 tile leaves as one run
template <class Src, class Dst, int KPT, int NW, int MINW, int MODE = 0>
__global__ void __launch_bounds__(NW * kWave, MINW)
k_leant(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
       uint32_t* __restrict__ ticket, unsigned long long* __restrict__ phase)
{
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
#define PH(i) do { if (tid == 0) { const unsigned long long now_ = clock64(); ph[i] += now_ - tprev; tprev = now_; } } while (0)
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    static_assert(kWave * KPT >= kRadix, "masks alias the stage");
    __shared__ LeanSmem<KPT, HAS_VAL, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    __syncthreads();
    for (;;) {
        if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        if (tid == 0 && tprev == 0) tprev = clock64();
        PH(0);
        const uint32_t tile_no = s.ticket;
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            key[r] = ~0ull;
            if (HAS_VAL) val[r] = 0u;
            if (MODE & 8) { key[r] = ((tile + idx) * 0x9E3779B1ull) << 24; if (HAS_VAL) val[r] = idx; }
            else if (idx < nvalid) src.load(tile + idx, key[r], val[HAS_VAL ? r : 0]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PH(1);
        if (MODE & 4) {
#pragma unroll
            for (int r = 0; r < KPT; r++) pos[r] = (r * kWave + lane) & 3u;
            if (lane < 4) { for (int k = 0; k < 64; k++) s.cnt[w][k * 4 + lane] = (uint16_t)(KPT * 16 / 64); }
        } else {
#pragma unroll
            for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
            wave_sync();
#pragma unroll
            for (int r = 0; r < KPT; r++) pos[r] = rank_round16((unsigned)(key[r] >> shift) & 255u, my_flags, s.cnt[w], mybit);
        }
        __syncthreads();
        PH(2);
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
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
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                if (!(MODE & 2)) lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
        }
        __syncthreads();
        PH(3);
        if (!(MODE & 16)) {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned p = pos[r] + s.cnt[w][(unsigned)(key[r] >> shift) & 255u];
                s.stage[p] = key[r];
                if (HAS_VAL) s.stage_v[p] = val[r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PH(4);
        if (owner) {
            if (MODE & 2) s.off[tid] = my_head + (uint32_t)((uint64_t)tile_no * 44u) - tile_ex;
            else s.off[tid] = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0) - tile_ex;
            if (MODE & 32) s.off[tid] = (uint32_t)tile;                         // timing: the sorted tile leaves as one run
        }
        __syncthreads();
        PH(5);
        constexpr int kOut = (KPT % 8 == 0) ? 8 : ((KPT % 6 == 0) ? 6 : ((KPT % 4 == 0) ? 4 : ((KPT % 3 == 0) ? 3 : KPT)));
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
            if (!(MODE & 16)) {
#pragma unroll
                for (int r = r0; r < r0 + kOut; r++) {
                    key[r] = s.stage[r * kThreads + tid];
                    if (HAS_VAL) val[r] = s.stage_v[r * kThreads + tid];
                }
            }
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) pos[r] = s.off[(unsigned)(key[r] >> shift) & 255u] + (r * kThreads + tid);
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++)
                if ((unsigned)(r * kThreads) + tid < nvalid && (!(MODE & 1) || pos[r] == 0x7FFFFFF1u)) dst.store(pos[r], key[r], HAS_VAL ? val[r] : 0u);
        }
        PH(6);
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
        PH(7);
    }
    if (tid == 0) for (int i = 0; i < 8; i++) atomicAdd(&phase[i], ph[i]);
#undef PH
}

// ---- k_lean2: k_lean with the per-tile latencies taken out of the critical path ------------------------------------
//   * the NEXT tile's ticket is requested while this tile is processed (a device-scope atomic on one address is a
//     1-3 us round trip with 256 workgroups asking), and handed over through LDS at the barrier that ends the tile:
//     one barrier per tile less;
//   * look-back reads LA predecessors per step (LA status words in flight per owner thread);
//   * PF: the next tile's keys are requested before this tile's output phase.
template <int LA> struct LookBackN { uint32_t sv[LA]; };
template <int LA>
__device__ __forceinline__ void lookbackN_begin(uint32_t* status, uint32_t tile_no, unsigned tid, uint32_t agg, LookBackN<LA>& lb, bool first)
{
    uint32_t* mine = status + (uint64_t)tile_no * kRadix + tid;
    __hip_atomic_store(mine, (first ? kStatusPrefix : kStatusAgg) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int u = 0; u < LA; u++) {
        const int64_t t = (int64_t)tile_no - 1 - u;
        lb.sv[u] = t >= 0 ? __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatusPrefix;
    }
}
template <int LA>
__device__ __forceinline__ uint32_t lookbackN_finish(uint32_t* status, uint32_t tile_no, unsigned tid, uint32_t agg, LookBackN<LA>& lb, bool first)
{
    if (first) return 0u;
    uint32_t* mine = status + (uint64_t)tile_no * kRadix + tid;
    uint32_t excl = 0;
    int64_t j = (int64_t)tile_no - 1;
    for (;;) {
#pragma unroll
        for (int u = 0; u < LA; u++) {
            const int64_t t = j - u;
            while ((lb.sv[u] >> 30) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                lb.sv[u] = __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            excl += lb.sv[u] & kStatusValue;
            if ((lb.sv[u] >> 30) == 2u) {
                __hip_atomic_store(mine, kStatusPrefix | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return excl;
            }
        }
        j -= LA;
#pragma unroll
        for (int u = 0; u < LA; u++) {
            const int64_t t = j - u;
            lb.sv[u] = t >= 0 ? __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatusPrefix;
        }
    }
}
template <int KPT, bool HAS_VAL, int NW>
struct Lean2Smem {
    uint64_t stage[NW * kWave * KPT];
    uint32_t stage_v[HAS_VAL ? NW * kWave * KPT : 1];
    uint16_t cnt[NW][kRadix];
    uint32_t off[kRadix];
    uint32_t part[2][NW];
    uint32_t ticket[2];
};
template <class Src, class Dst, int KPT, int NW, int MINW, int LA, bool PF, int MODE = 0>
__global__ void __launch_bounds__(NW * kWave, MINW)
k_lean2(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
        uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    static_assert(kWave * KPT >= kRadix, "masks alias the stage");
    __shared__ Lean2Smem<KPT, HAS_VAL, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    if (tid == 0) s.ticket[0] = atomicAdd(ticket, 1u);
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    __syncthreads();
    // the digit sits inside one 32-bit half of the key (shift % 8 == 0 in every caller of the lab)
    const bool hi_half = shift >= 32;
    const unsigned dsh = (unsigned)shift & 31u;
    auto digit = [&](uint64_t k) -> unsigned { return (((hi_half ? (uint32_t)(k >> 32) : (uint32_t)k) >> dsh) & 255u); };
    unsigned cur = 0;
    uint64_t nkey[PF ? KPT : 1];
    uint32_t nval[(PF && HAS_VAL) ? KPT : 1];
    auto load_into = [&](uint64_t t, unsigned nv, uint64_t (&K)[PF ? KPT : 1], uint32_t (&V)[(PF && HAS_VAL) ? KPT : 1]) {
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            K[r] = ~0ull;
            if (HAS_VAL) V[r] = 0u;
            if (idx < nv) src.load(t + idx, K[r], V[HAS_VAL ? r : 0]);
        }
    };
    if (PF) {
        const uint64_t t0 = (uint64_t)s.ticket[0] * kTile;
        if (t0 < m) load_into(t0, (unsigned)dmin<uint64_t>(kTile, m - t0), nkey, nval);
    }
    for (;;) {
        const uint32_t tile_no = s.ticket[cur];
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        uint32_t next_ticket = 0;
        if (tid == 0) next_ticket = atomicAdd(ticket, 1u);            // consumed at the end of this tile
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos[KPT];
        if (PF) {
#pragma unroll
            for (int r = 0; r < KPT; r++) { key[r] = nkey[r]; if (HAS_VAL) val[r] = nval[r]; }
        } else {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
                key[r] = ~0ull;
                if (HAS_VAL) val[r] = 0u;
                if (idx < nvalid) src.load(tile + idx, key[r], val[HAS_VAL ? r : 0]);
            }
        }
#pragma unroll
        for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] = rank_round16(digit(key[r]), my_flags, s.cnt[w], mybit);
        __syncthreads();
        LookBackN<LA> lb;
        uint32_t real_count = 0, tile_ex = 0;
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
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                if (!(MODE & 2)) lookbackN_begin<LA>(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned p = pos[r] + s.cnt[w][digit(key[r])];
            s.stage[p] = key[r];
            if (HAS_VAL) s.stage_v[p] = val[r];
        }
        if (tid == 0) s.ticket[cur ^ 1u] = next_ticket;
        if (owner) {
            if (MODE & 2) s.off[tid] = my_head + (uint32_t)((uint64_t)tile_no * 44u) - tile_ex;
            else s.off[tid] = my_head + lookbackN_finish<LA>(status, tile_no, tid, real_count, lb, tile_no == 0) - tile_ex;
        }
        __syncthreads();
        if (PF) {
            const uint64_t t1 = (uint64_t)s.ticket[cur ^ 1u] * kTile;
            if (t1 < m) load_into(t1, (unsigned)dmin<uint64_t>(kTile, m - t1), nkey, nval);
        }
        if (owner) {                                       // (the per-wave bases were consumed by the staging above)
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        constexpr int kOut = (KPT % 8 == 0) ? 8 : ((KPT % 6 == 0) ? 6 : ((KPT % 4 == 0) ? 4 : ((KPT % 3 == 0) ? 3 : KPT)));
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) {
                key[r] = s.stage[r * kThreads + tid];
                if (HAS_VAL) val[r] = s.stage_v[r * kThreads + tid];
            }
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++) pos[r] = s.off[digit(key[r])] + (r * kThreads + tid);
#pragma unroll
            for (int r = r0; r < r0 + kOut; r++)
                if ((unsigned)(r * kThreads) + tid < nvalid && (!(MODE & 1) || pos[r] == 0x7FFFFFF1u)) dst.store(pos[r], key[r], HAS_VAL ? val[r] : 0u);
        }
        cur ^= 1u;
        __syncthreads();
    }
}

// ---- k_pipe: software-pipelined one-sweep ---------------------------------------------------------------------------
// One 1024-thread workgroup per CU.  While the stores of tile i leave (the CU's store path is what a pass waits for:
// ~20 cycles per line written), the same waves rank tile i+1, whose keys were requested before the staging of tile i.
// The match masks and the counts of the tile being ranked live outside the staging buffer.
template <int KPT, bool HAS_VAL, int NW>
struct PipeSmem {
    uint64_t stage[NW * kWave * KPT];
    uint32_t stage_v[HAS_VAL ? NW * kWave * KPT : 1];
    unsigned long long flags[NW][kRadix];
    uint16_t cnt[NW][kRadix];
    uint32_t off[kRadix];
    uint32_t part[2][NW];
    uint32_t ticket[2];
};
template <class Src, class Dst, int KPT, int NW>
__global__ void __launch_bounds__(NW * kWave, 1)
k_pipe(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
       uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    __shared__ PipeSmem<KPT, HAS_VAL, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) { s.cnt[k][tid] = 0; s.flags[k][tid] = 0ull; }
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    if (tid == 0) s.ticket[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    uint32_t tile_no = s.ticket[0];
    uint64_t tile = (uint64_t)tile_no * kTile;
    if (tile >= m) return;
    unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);

    uint64_t key[KPT];
    uint32_t val[HAS_VAL ? KPT : 1];
    uint32_t pos[KPT];
    auto load_tile = [&](uint64_t t, unsigned nv) {
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            key[r] = ~0ull;
            if (HAS_VAL) val[r] = 0u;
            if (idx < nv) src.load(t + idx, key[r], val[HAS_VAL ? r : 0]);
        }
    };
    // prologue: the first tile is loaded and ranked without anything to hide behind
    load_tile(tile, nvalid);
#pragma unroll
    for (int r = 0; r < KPT; r++) pos[r] = rank_round16((unsigned)(key[r] >> shift) & 255u, s.flags[w], s.cnt[w], mybit);
    unsigned slot = 1;
    for (;;) {
        // here: key/val/pos = tile `tile_no`, ranked; s.cnt = its per-wave digit counts
        __syncthreads();
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
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
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
            if (tid == 0) s.ticket[slot] = atomicAdd(ticket, 1u);            // the tile after this one
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned p = pos[r] + s.cnt[w][(unsigned)(key[r] >> shift) & 255u];
            s.stage[p] = key[r];
            if (HAS_VAL) s.stage_v[p] = val[r];
        }
        const uint32_t next_no = s.ticket[slot];
        slot ^= 1u;
        const uint64_t next_tile = (uint64_t)next_no * kTile;
        const bool more = next_tile < m;
        const unsigned next_valid = more ? (unsigned)dmin<uint64_t>(kTile, m - next_tile) : 0u;
        // the next tile's keys are requested BEFORE this tile's stores (loads and stores share the queue)
        if (more) load_tile(next_tile, next_valid);
        if (owner) s.off[tid] = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0) - tile_ex;
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
        // stores of this tile interleaved with the ranking of the next
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned p = r * kThreads + tid;
            const uint64_t ok = s.stage[p];
            const uint32_t ov = HAS_VAL ? s.stage_v[p] : 0u;
            const uint32_t dest = s.off[(unsigned)(ok >> shift) & 255u] + p;
            if (p < nvalid) dst.store(dest, ok, ov);
            if (more) pos[r] = rank_round16((unsigned)(key[r] >> shift) & 255u, s.flags[w], s.cnt[w], mybit);
        }
        if (!more) break;
        tile_no = next_no;
        tile = next_tile;
        nvalid = next_valid;
    }
}

// ---- k_big: block-aligned output ------------------------------------------------------------------------------------
// Store probes (this file): a store request that covers a whole, aligned 64-byte block costs ~14 ps chip-wide, one
// that covers part of a block ~46 ps -- whichever workgroup completes the block later.  The shipped output loop maps
// lane -> staging slot, so a bucket's run is cut wherever a 64-lane instruction ends and at both of its own ends:
// ~3 partial requests per 44-element run.  Here the lanes walk a VIRTUAL layout in which every bucket's run starts at
// its global address modulo BLK elements (BLK * element size = a multiple of 64 bytes): instruction boundaries fall on
// block boundaries, only a run's two ends are partial.  And the tile is ranked as a whole but staged in SR rounds of
// T / SR elements, so a run is SR times as long as the LDS alone would allow.
template <int KPT, bool HAS_VAL, int NW, int SR, int BLK>
struct BigSmem {
    static constexpr int kTile = NW * kWave * KPT;
    static constexpr int kHalf = kTile / SR;
    static constexpr int kBlocks = (kTile + kRadix * (2 * BLK - 2)) / BLK + 2;
    uint64_t stage[kHalf];
    uint32_t stage_v[HAS_VAL ? kHalf : 1];
    uint16_t cnt[NW][kRadix];
    uint4 binfo[kRadix];                  // x: delta (q - p), y: first valid q, z: one past the last valid q, w: global index of q = 0
    uint32_t vend[kRadix];
    uint8_t blk[kBlocks];
    uint32_t part[2][NW];
    uint32_t qcut[SR + 1];
    uint32_t ticket;
};
template <class Src, class Dst, int KPT, int NW, int SR, int BLK>
__global__ void __launch_bounds__(NW * kWave, 1)
k_big(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
      uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    constexpr int kHalf = kTile / SR;
    static_assert(KPT % SR == 0, "whole staging rounds");
    static_assert(kHalf * 8 >= NW * kRadix * 8, "masks alias the stage");
    static_assert(kTile + kRadix * 2 * BLK < 65536, "16-bit tile positions");
    __shared__ BigSmem<KPT, HAS_VAL, NW, SR, BLK> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    __syncthreads();
    for (;;) {
        if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile_no = s.ticket;
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            key[r] = ~0ull;
            if (HAS_VAL) val[r] = 0u;
            if (idx < nvalid) src.load(tile + idx, key[r], val[HAS_VAL ? r : 0]);
        }
#pragma unroll
        for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] = rank_round16((unsigned)(key[r] >> shift) & 255u, my_flags, s.cnt[w], mybit);
        __syncthreads();
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
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
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
        }
        __syncthreads();
        // tile-local positions of this thread's elements (the staging rounds pick theirs)
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] += s.cnt[w][(unsigned)(key[r] >> shift) & 255u];
        // the virtual layout: bucket d's run starts at its global address modulo BLK
        uint32_t ghead = 0, h = 0, vsize = 0;
        if (owner) {
            ghead = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0);
            h = ghead & (uint32_t)(BLK - 1);
            vsize = real_count ? ((h + real_count + BLK - 1) & ~(uint32_t)(BLK - 1)) : 0u;
        }
        const uint32_t vstart = block_scan_excl_1b<NW>(vsize, s.part, par);
        if (owner) {
            s.binfo[tid] = uint4{vstart + h - tile_ex, vstart + h, vstart + h + real_count, ghead - h - vstart};
            s.vend[tid] = vstart + vsize;
#pragma unroll
            for (int q = 0; q <= SR; q++) {
                const uint32_t cut = (uint32_t)q * kHalf;
                // (the tile's padding sits at the end of bucket 255: the last cut is the end of the layout)
                if (q == SR) { if (tid == 255u) s.qcut[SR] = vstart + vsize; }
                else if (cut >= tile_ex && (cut < tile_ex + real_count || (tid == 255u))) s.qcut[q] = dmin(cut + (vstart + h - tile_ex), vstart + vsize);
            }
        }
        __syncthreads();
        {
            const uint32_t nblocks = s.qcut[SR] / BLK;
            for (uint32_t b = tid; b < nblocks; b += kThreads) {
                const uint32_t q = b * BLK;
                unsigned lo = 0, hi = 255;                       // smallest d with vend[d] > q
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (s.vend[mid] > q) hi = mid; else lo = mid + 1;
                }
                s.blk[b] = (uint8_t)lo;
            }
        }
#pragma unroll
        for (int sr = 0; sr < SR; sr++) {
            const uint32_t plo = (uint32_t)sr * kHalf, phi = plo + kHalf;
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                if (pos[r] >= plo && pos[r] < phi) {
                    s.stage[pos[r] - plo] = key[r];
                    if (HAS_VAL) s.stage_v[pos[r] - plo] = val[r];
                }
            }
            __syncthreads();
            const uint32_t qlo = s.qcut[sr] & ~(uint32_t)(BLK - 1);
            const uint32_t qhi = s.qcut[sr + 1];
            for (uint32_t q = qlo + tid; q < qhi; q += kThreads) {
                const unsigned d = s.blk[q / BLK];
                const uint4 bi = s.binfo[d];
                const uint32_t p = q - bi.x;
                if (q >= bi.y && q < bi.z && p >= plo && p < phi) dst.store(bi.w + q, s.stage[p - plo], HAS_VAL ? s.stage_v[p - plo] : 0u);
            }
            __syncthreads();
        }
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
    }
}

// ---- k_pipe2: the look-back wait hidden behind the ranking of the next tile -----------------------------------------------
// Phase timers (k_leant) say where a tile's 14 us go: ranking 25 %, waiting for the look-back 26 %, waiting for the loads 16 %,
// scan 12 % -- one after the other, because a 1024-thread workgroup owns its CU.  Here the workgroup keeps two tiles in
// flight: tile i sits ranked in registers / staged in LDS while tile i+1 is loaded and ranked, and only then is the look-back of
// tile i finished (it had the whole ranking to resolve) and the tile written out.  A tile's aggregate is published as soon as it
// is ranked, a whole iteration before its own look-back begins, so successors find prefixes instead of aggregates.
template <int KPT, bool HAS_VAL, int NW>
struct Pipe2Smem {
    uint64_t stage[NW * kWave * KPT];
    uint32_t stage_v[HAS_VAL ? NW * kWave * KPT : 1];
    unsigned long long flags[NW][kRadix];
    uint16_t cnt[2][NW][kRadix];
    uint32_t off[kRadix];
    uint32_t part[2][NW];
    uint32_t ticket[2];
};
template <class Src, class Dst, int KPT, int NW, int LA, int MODE = 0>
__global__ void __launch_bounds__(NW * kWave, 1)
k_pipe2(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
        uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    __shared__ Pipe2Smem<KPT, HAS_VAL, NW> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) { s.cnt[0][k][tid] = 0; s.cnt[1][k][tid] = 0; s.flags[k][tid] = 0ull; }
    }
    if (tid == 0) s.ticket[0] = atomicAdd(ticket, 1u);
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    __syncthreads();
    const bool hi_half = shift >= 32;
    const unsigned dsh = (unsigned)shift & 31u;
    auto digit = [&](uint64_t k) -> unsigned { return (((hi_half ? (uint32_t)(k >> 32) : (uint32_t)k) >> dsh) & 255u); };

    uint32_t tile_no = s.ticket[0];
    uint64_t tile = (uint64_t)tile_no * kTile;
    if (tile >= m) return;
    unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
    uint32_t next_ticket = 0;
    if (tid == 0) next_ticket = atomicAdd(ticket, 1u);

    uint64_t key[KPT], nkey[KPT];
    uint32_t val[HAS_VAL ? KPT : 1], nval[HAS_VAL ? KPT : 1];
    uint32_t pos[KPT];
    // prologue: the first tile is loaded and ranked with nothing to hide behind
#pragma unroll
    for (int r = 0; r < KPT; r++) {
        const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
        key[r] = ~0ull;
        if (HAS_VAL) val[r] = 0u;
        if (idx < nvalid) src.load(tile + idx, key[r], val[HAS_VAL ? r : 0]);
    }
#pragma unroll
    for (int r = 0; r < KPT; r++) pos[r] = rank_round16(digit(key[r]), s.flags[w], s.cnt[0][w], mybit);
    __syncthreads();
    uint32_t c[NW], tile_count = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        c[k] = owner ? s.cnt[0][k][tid] : 0u;
        tile_count += c[k];
    }
    uint32_t real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
    if (owner && !(MODE & 2))
        __hip_atomic_store(status + (uint64_t)tile_no * kRadix + tid, (tile_no == 0 ? kStatusPrefix : kStatusAgg) | real_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned cur = 0;                     // cnt[cur] holds the counts of the tile in key/pos; ticket[cur] its number
    for (;;) {
        // A: tile-local bucket starts, per-wave bases; the look-back's first loads
        const uint32_t ex = block_scan_excl_1b<NW>(tile_count, s.part, par);
        LookBackN<LA> lb;
        if (owner) {
            uint32_t run = ex;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                s.cnt[cur][k][tid] = (uint16_t)run;
                run += c[k];
                s.cnt[cur ^ 1u][k][tid] = 0;                     // (consumed one iteration ago)
            }
            if (!(MODE & 2)) {
#pragma unroll
                for (int u = 0; u < LA; u++) {
                    const int64_t t = (int64_t)tile_no - 1 - u;
                    lb.sv[u] = t >= 0 ? __hip_atomic_load(status + (uint64_t)t * kRadix + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatusPrefix;
                }
            }
        }
        if (tid == 0) s.ticket[cur ^ 1u] = next_ticket;
        __syncthreads();
        const uint32_t next_no = s.ticket[cur ^ 1u];
        const uint64_t next_tile = (uint64_t)next_no * kTile;
        const bool more = next_tile < m;
        const unsigned next_valid = more ? (unsigned)dmin<uint64_t>(kTile, m - next_tile) : 0u;
        if (more) {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
                nkey[r] = ~0ull;
                if (HAS_VAL) nval[r] = 0u;
                if (idx < next_valid) src.load(next_tile + idx, nkey[r], nval[HAS_VAL ? r : 0]);
            }
            if (tid == 0) next_ticket = atomicAdd(ticket, 1u);
        }
        // B: staging of the current tile
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned p = pos[r] + s.cnt[cur][w][digit(key[r])];
            s.stage[p] = key[r];
            if (HAS_VAL) s.stage_v[p] = val[r];
        }
        // C: the next tile is ranked while the look-back of the current one resolves
        uint32_t ncount = 0, nreal = 0;
        if (more) {
#pragma unroll
            for (int r = 0; r < KPT; r++) pos[r] = rank_round16(digit(nkey[r]), s.flags[w], s.cnt[cur ^ 1u][w], mybit);
        }
        __syncthreads();
        if (more) {
#pragma unroll
            for (int k = 0; k < NW; k++) {
                c[k] = owner ? s.cnt[cur ^ 1u][k][tid] : 0u;
                ncount += c[k];
            }
            nreal = ncount - ((tid == 255u) ? (uint32_t)(kTile - next_valid) : 0u);
            if (owner && !(MODE & 2))
                __hip_atomic_store(status + (uint64_t)next_no * kRadix + tid, kStatusAgg | nreal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // D: the current tile's bucket heads
        if (owner) {
            if (MODE & 2) s.off[tid] = my_head + (uint32_t)((uint64_t)tile_no * 44u) - ex;
            else s.off[tid] = my_head + lookbackN_finish<LA>(status, tile_no, tid, real_count, lb, tile_no == 0) - ex;
        }
        __syncthreads();
        // E: out
        constexpr int kOut = (KPT % 8 == 0) ? 8 : ((KPT % 6 == 0) ? 6 : ((KPT % 4 == 0) ? 4 : ((KPT % 3 == 0) ? 3 : KPT)));
#pragma unroll
        for (int r0 = 0; r0 < KPT; r0 += kOut) {
            uint64_t ok[kOut];
            uint32_t ov[HAS_VAL ? kOut : 1];
            uint32_t od[kOut];
#pragma unroll
            for (int r = 0; r < kOut; r++) {
                ok[r] = s.stage[(r0 + r) * kThreads + tid];
                if (HAS_VAL) ov[r] = s.stage_v[(r0 + r) * kThreads + tid];
            }
#pragma unroll
            for (int r = 0; r < kOut; r++) od[r] = s.off[digit(ok[r])] + ((r0 + r) * kThreads + tid);
#pragma unroll
            for (int r = 0; r < kOut; r++)
                if ((unsigned)((r0 + r) * kThreads) + tid < nvalid && (!(MODE & 1) || od[r] == 0x7FFFFFF1u)) dst.store(od[r], ok[r], HAS_VAL ? ov[r] : 0u);
        }
        if (!more) break;
        cur ^= 1u;
        tile_no = next_no;
        nvalid = next_valid;
        tile_count = ncount;
        real_count = nreal;
#pragma unroll
        for (int r = 0; r < KPT; r++) { key[r] = nkey[r]; if (HAS_VAL) val[r] = nval[r]; }
    }
}

// ---- k_big2: k_big with the staging before the look-back result is needed, windows of W slots whose bucket comes from a
// table with one entry per window, output unrolled, tile positions packed two per register ------------------------------------
template <int KPT, bool HAS_VAL, int NW, int SR, int ALIGN, int W>
struct Big2Smem {
    static constexpr int kTile = NW * kWave * KPT;
    static constexpr int kHalf = kTile / SR;
    static constexpr int kWins = (kTile + kRadix * (ALIGN + W - 2)) / W + 2;
    uint64_t stage[kHalf];
    uint32_t stage_v[HAS_VAL ? kHalf : 1];
    uint16_t cnt[NW][kRadix];
    uint4 binfo[kRadix];                  // x: delta (q - p), y: first valid q, z: one past the last valid q, w: global index of q = 0
    uint32_t vend[kRadix];
    uint8_t win[kWins];
    uint32_t part[2][NW];
    uint32_t qcut[SR + 1];
    uint32_t ticket;
};
template <class Src, class Dst, int KPT, int NW, int SR, int ALIGN, int W, int MODE = 0>
__global__ void __launch_bounds__(NW * kWave, 1)
k_big2(Src src, Dst dst, uint64_t m, int shift, const uint32_t* __restrict__ digit_total, uint32_t* __restrict__ status,
       uint32_t* __restrict__ ticket)
{
    constexpr bool HAS_VAL = Src::kHasVal;
    constexpr int kThreads = NW * kWave;
    constexpr int kTile = kThreads * KPT;
    constexpr int kHalf = kTile / SR;
    static_assert(KPT % SR == 0 && KPT % 2 == 0 || SR == 1, "whole staging rounds");
    static_assert(kHalf * 8 >= NW * kRadix * 8, "masks alias the stage");
    static_assert(kTile + kRadix * (ALIGN + W) < 65536, "16-bit tile positions");
    static_assert(W % ALIGN == 0, "windows are whole blocks");
    __shared__ Big2Smem<KPT, HAS_VAL, NW, SR, ALIGN, W> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    const bool owner = tid < (unsigned)kRadix;
    unsigned par = 0;
    unsigned long long* const my_flags = reinterpret_cast<unsigned long long*>(s.stage) + w * kRadix;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
    }
    const uint32_t my_head = block_scan_excl_1b<NW>(owner ? digit_total[tid] : 0u, s.part, par);
    const bool hi_half = shift >= 32;
    const unsigned dsh = (unsigned)shift & 31u;
    auto digit = [&](uint64_t k) -> unsigned { return (((hi_half ? (uint32_t)(k >> 32) : (uint32_t)k) >> dsh) & 255u); };
    __syncthreads();
    for (;;) {
        if (tid == 0) s.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile_no = s.ticket;
        const uint64_t tile = (uint64_t)tile_no * kTile;
        if (tile >= m) break;
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, m - tile);
        uint64_t key[KPT];
        uint32_t val[HAS_VAL ? KPT : 1];
        uint16_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            key[r] = ~0ull;
            if (HAS_VAL) val[r] = 0u;
            if (idx < nvalid) src.load(tile + idx, key[r], val[HAS_VAL ? r : 0]);
        }
#pragma unroll
        for (int k = 0; k < kRadix / kWave; k++) my_flags[k * kWave + lane] = 0ull;
        wave_sync();
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] = (uint16_t)rank_round16(digit(key[r]), my_flags, s.cnt[w], mybit);
        __syncthreads();
        LookBack lb;
        uint32_t real_count = 0, tile_ex = 0;
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
                    s.cnt[k][tid] = (uint16_t)run;
                    run += c[k];
                }
                real_count = tile_count - ((tid == 255u) ? (uint32_t)(kTile - nvalid) : 0u);
                tile_ex = ex;
                lookback_begin(status, tile_no, tid, real_count, lb, tile_no == 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KPT; r++) pos[r] = (uint16_t)(pos[r] + s.cnt[w][digit(key[r])]);
        // first staging round (the look-back resolves meanwhile)
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            if (SR == 1 || pos[r] < kHalf) {
                s.stage[pos[r]] = key[r];
                if (HAS_VAL) s.stage_v[pos[r]] = val[r];
            }
        }
        uint32_t ghead = 0, h = 0, vsize = 0;
        if (owner) {
            ghead = my_head + lookback_finish(status, tile_no, tid, real_count, lb, tile_no == 0);
            h = ghead & (uint32_t)(ALIGN - 1);
            vsize = real_count ? ((h + real_count + W - 1) / W) * W : 0u;
        }
        const uint32_t vstart = block_scan_excl_1b<NW>(vsize, s.part, par);
        if (owner) {
            s.binfo[tid] = uint4{vstart + h - tile_ex, vstart + h, vstart + h + real_count, ghead - h - vstart};
            s.vend[tid] = vstart + vsize;
#pragma unroll
            for (int q = 0; q <= SR; q++) {
                const uint32_t cut = (uint32_t)q * kHalf;
                if (q == SR) { if (tid == 255u) s.qcut[SR] = vstart + vsize; }
                else if (cut >= tile_ex && (cut < tile_ex + real_count || (tid == 255u))) s.qcut[q] = dmin(cut + (vstart + h - tile_ex), vstart + vsize);
            }
        }
        __syncthreads();
        {
            const uint32_t nwins = s.qcut[SR] / W;
            for (uint32_t b = tid; b < nwins; b += kThreads) {
                const uint32_t q = b * W;
                unsigned lo = 0, hi = 255;                       // smallest d with vend[d] > q
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (s.vend[mid] > q) hi = mid; else lo = mid + 1;
                }
                s.win[b] = (uint8_t)lo;
            }
        }
        __syncthreads();
#pragma unroll
        for (int sr = 0; sr < SR; sr++) {
            const uint32_t plo = (uint32_t)sr * kHalf, phi = plo + kHalf;
            if (sr > 0) {
#pragma unroll
                for (int r = 0; r < KPT; r++) {
                    if (pos[r] >= plo && pos[r] < phi) {
                        s.stage[pos[r] - plo] = key[r];
                        if (HAS_VAL) s.stage_v[pos[r] - plo] = val[r];
                    }
                }
                __syncthreads();
            }
            const uint32_t qlo = s.qcut[sr] & ~(uint32_t)(W - 1);
            const uint32_t qhi = s.qcut[sr + 1];
            constexpr int U = 4;
            for (uint32_t q0 = qlo + tid; q0 < qhi; q0 += U * kThreads) {
                unsigned d[U];
                uint4 bi[U];
                uint64_t ok[U];
                uint32_t ov[HAS_VAL ? U : 1];
                bool go[U];
#pragma unroll
                for (int u = 0; u < U; u++) { const uint32_t q = q0 + u * kThreads; d[u] = q < qhi ? s.win[q / W] : 0u; }
#pragma unroll
                for (int u = 0; u < U; u++) bi[u] = s.binfo[d[u]];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t q = q0 + u * kThreads;
                    const uint32_t p = q - bi[u].x;
                    go[u] = q < qhi && q >= bi[u].y && q < bi[u].z && p >= plo && p < phi;
                    ok[u] = go[u] ? s.stage[p - plo] : 0ull;
                    if (HAS_VAL) ov[u] = go[u] ? s.stage_v[p - plo] : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (go[u] && !(MODE & 1)) dst.store(bi[u].w + (q0 + u * kThreads), ok[u], HAS_VAL ? ov[u] : 0u);
            }
            __syncthreads();
        }
        if (owner) {
#pragma unroll
            for (int k = 0; k < NW; k++) s.cnt[k][tid] = 0;
        }
        __syncthreads();
    }
}
}  // namespace sfx

using namespace sfx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class F> static float time_ms(F&& f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms / reps;
}

static unsigned g_grid_cap = 0;
static uint32_t* g_totals; static uint32_t* g_status; static uint32_t* g_ticket; static unsigned long long* g_bad;

template <class Src, class Dst, int KPT, int NW, int MINW, int MODE = 0, int KO = 0>
static void run_lean(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, g_grid_cap ? g_grid_cap : kMaxGrid);
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_lean<Src, Dst, KPT, NW, MINW, MODE, KO>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket);
    };
    const float t = time_ms(f);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
}
template <class Src, class Dst, int KPT, int NW>
static void run_pipe(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes, unsigned grid_cap = kMaxGrid)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, grid_cap);
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_pipe<Src, Dst, KPT, NW>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket);
    };
    const float t = time_ms(f);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
}

template <class Src, class Dst, int KPT, int NW, int SR, int BLK>
static void run_big(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, kMaxGrid);
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_big<Src, Dst, KPT, NW, SR, BLK>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket);
    };
    const float t = time_ms(f);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
}

template <class Src, class Dst, int KPT, int NW, int MINW, int LA, bool PF, int MODE = 0>
static void run_lean2(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, kMaxGrid);
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_lean2<Src, Dst, KPT, NW, MINW, LA, PF, MODE>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket);
    };
    const float t = time_ms(f);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
}

static unsigned long long* g_phase;
template <class Src, class Dst, int KPT, int NW, int MINW, int MODE = 0>
static void run_leant(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, g_grid_cap ? g_grid_cap : kMaxGrid);
    hipMemset(g_phase, 0, 64);
    int launches = 0;
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_leant<Src, Dst, KPT, NW, MINW, MODE>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket, g_phase);
        launches++;
    };
    const float t = time_ms(f);
    unsigned long long ph[8];
    hipMemcpy(ph, g_phase, 64, hipMemcpyDeviceToHost);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
    const char* names[8] = {"ticket+barrier", "loads", "rank+barrier", "scan+lb_begin+barrier", "staging", "lb_finish+barrier", "output issue", "zero+barrier"};
    double tot = 0; for (int i = 0; i < 8; i++) tot += (double)ph[i];
    for (int i = 0; i < 8; i++) printf("      %-24s %8.0f cycles/tile  %5.1f %%\n", names[i], (double)ph[i] / (double)(tiles * launches), 100.0 * ph[i] / tot);
}

template <class Src, class Dst, int KPT, int NW, int LA, int MODE = 0>
static void run_pipe2(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, g_grid_cap ? g_grid_cap : kMaxGrid);
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_pipe2<Src, Dst, KPT, NW, LA, MODE>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket);
    };
    const float t = time_ms(f);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
}

template <class Src, class Dst, int KPT, int NW, int SR, int ALIGN, int W, int MODE = 0>
static void run_big2(const char* label, Src src, Dst dst, uint64_t m, int shift, double bytes)
{
    constexpr int kTile = NW * 64 * KPT;
    const uint64_t tiles = (m + kTile - 1) / kTile;
    const unsigned grid = (unsigned)dmin<uint64_t>(tiles, g_grid_cap ? g_grid_cap : kMaxGrid);
    auto f = [&] {
        hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
        hipMemsetAsync(g_ticket, 0, 4, 0);
        hipLaunchKernelGGL((k_big2<Src, Dst, KPT, NW, SR, ALIGN, W, MODE>), dim3(grid), dim3(NW * 64), 0, 0, src, dst, m, shift,
                           (const uint32_t*)g_totals, g_status, g_ticket);
    };
    const float t = time_ms(f);
    printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", label, kTile, grid, t, bytes / t * 1e-9);
}
static unsigned long long bad_count()
{
    unsigned long long b = 0;
    hipMemcpy(&b, g_bad, 8, hipMemcpyDeviceToHost);
    hipMemset(g_bad, 0, 8);
    return b;
}

int main(int argc, char** argv)
{
    const uint64_t m = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
    const bool probes = argc > 2 ? atoi(argv[2]) != 0 : true;
    uint64_t *in, *out, *ref; uint32_t *vin, *vout, *vref; uint32_t* scratch;
    CK(hipMalloc(&in, m * 8)); CK(hipMalloc(&out, m * 12 + 65536)); CK(hipMalloc(&ref, m * 8));
    CK(hipMalloc(&vin, m * 4)); CK(hipMalloc(&vout, m * 4)); CK(hipMalloc(&vref, m * 4));
    CK(hipMalloc(&scratch, radix_scratch_words(m) * 4 + (m / 2048 + 64) * 1024));
    CK(hipMalloc(&g_bad, 8)); CK(hipMemset(g_bad, 0, 8));
    CK(hipMalloc(&g_phase, 64));
    RadixScratch scr(scratch, m);
    g_totals = scr.totals; g_status = scr.status; g_ticket = scr.tickets;

    if (probes) {
        printf("== store probes (%.0f MB)\n", m * 8e-6);
        for (unsigned g : {64u, 128u, 256u, 512u, 1024u, 2048u, 8192u}) {
            const float t = time_ms([&] { hipLaunchKernelGGL(k_write16, dim3(g), dim3(256), 0, 0, (uint4*)out, m / 2); });
            printf("write16  grid %5u x 256 threads: %.3f ms  %.2f TB/s\n", g, t, m * 8.0 / t * 1e-9);
        }
        for (int h : {16, 12, 8, 4, 1}) {
            const float t = time_ms([&] { hipLaunchKernelGGL(k_write_partial, dim3(4096), dim3(256), 0, 0, out, m / 16, h); });
            printf("partial lines: %2d of 16 words per line: %.3f ms  (%.1f G lines/s)\n", h, t, (m / 16) / t * 1e-6);
        }
        for (unsigned R : {16u, 24u, 32u, 36u, 44u, 64u, 72u, 88u, 128u, 176u, 256u, 512u, 2048u}) {
            const uint64_t nruns = m / R;
            uint64_t p2 = 1; while (p2 < nruns) p2 <<= 1;
            for (unsigned mis : {0u, 5u}) {
                const float t = time_ms([&] { hipLaunchKernelGGL(k_write_runs, dim3(1024), dim3(1024), 0, 0, out, nruns * R, R, (unsigned)(p2 - 1), nruns, mis); });
                printf("runs of %4u elements (%5u B), misalign %u: %.3f ms  %.2f TB/s\n", R, R * 8, mis, t, nruns * R * 8.0 / t * 1e-9);
            }
        }
    }

    for (int dist = 0; dist < 2; dist++) {
        printf("== E64 pass, %llu elements, %s digits\n", (unsigned long long)m, dist ? "skewed" : "uniform");
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, in, m, dist);
        CK(hipDeviceSynchronize());
        const int shift = 40;
        // digit totals of this pass
        {
            Chunking ch = make_chunking(m, kBlock * 8, kHistAllGrid);
            const uint64_t chunk = ch.tiles_per_block * kBlock * 8;
            hipLaunchKernelGGL((k_radix_hist_all<SrcE64>), dim3(ch.blocks), dim3(kBlock), 0, 0, SrcE64{in}, m, shift, shift + 8, 1, chunk, scr.partial);
            hipLaunchKernelGGL(k_radix_scan, dim3(kRadix), dim3(kBlock), 0, 0, scr.partial, ch.blocks, scr.totals);
            CK(hipDeviceSynchronize());
        }
        // the shipped kernel: reference output + time
        {
            constexpr int kTile = 16 * 64 * 11;
            const uint64_t tiles = (m + kTile - 1) / kTile;
            auto f = [&] {
                hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
                hipMemsetAsync(g_ticket, 0, 4, 0);
                hipLaunchKernelGGL((k_radix_pass<SrcE64, DstE64, 11, true, true, 16>), dim3((unsigned)dmin<uint64_t>(tiles, kMaxGrid)), dim3(1024), 0, 0,
                                   SrcE64{in}, DstE64{ref}, m, shift, 255u, (uint64_t)0, (const uint32_t*)nullptr, (const uint32_t*)g_totals, g_status, g_ticket, SegArgs{nullptr, nullptr, nullptr, 0, 0});
            };
            const float t = time_ms(f);
            printf("%-44s tile %5d grid %4u  %.3f ms  %.2f TB/s\n", "shipped k_radix_pass<E64,11,16>", kTile, (unsigned)dmin<uint64_t>(tiles, kMaxGrid), t, m * 16.0 / t * 1e-9);
        }
#define CHECK64() do { hipLaunchKernelGGL(k_cmp64, dim3(1024), dim3(256), 0, 0, (const uint64_t*)out, (const uint64_t*)ref, m, g_bad); printf("    mismatches vs shipped: %llu\n", bad_count()); } while (0)
        run_lean<LSrcE64, LDstE64, 11, 16, 1>("lean 16w x 11 (1 wg/cu)", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 12, 16, 1, 0, 4>("lean E64 16w x 12 out-batch 4", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 14, 16, 1, 0, 2>("lean E64 16w x 14 out-batch 2", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 15, 16, 1, 0, 1>("lean E64 16w x 15 out-batch 1", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 15, 16, 1, 0, 3>("lean E64 16w x 15 out-batch 3", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 15, 16, 1, 0, 5>("lean E64 16w x 15 out-batch 5", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 16, 16, 1, 0, 2>("lean E64 16w x 16 out-batch 2", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 16, 16, 1, 0, 4>("lean E64 16w x 16 out-batch 4", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 17, 16, 1, 0, 1>("lean E64 16w x 17 out-batch 1", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 18, 16, 1, 0, 2>("lean E64 16w x 18 out-batch 2", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        run_lean<LSrcE64, LDstE64, 18, 16, 1, 0, 3>("lean E64 16w x 18 out-batch 3", LSrcE64{in}, LDstE64{out}, m, shift, m * 16.0); CHECK64();
        g_grid_cap = 0;
        g_grid_cap = 0;
    }

    printf("== KV pass (u64 key + u32 value), %llu elements\n", (unsigned long long)m);
    {
        hipLaunchKernelGGL(k_fill_kv, dim3(2048), dim3(256), 0, 0, in, vin, m);
        CK(hipDeviceSynchronize());
        const int shift = 24;
        Chunking ch = make_chunking(m, kBlock * 8, kHistAllGrid);
        const uint64_t chunk = ch.tiles_per_block * kBlock * 8;
        hipLaunchKernelGGL((k_radix_hist_all<SrcE64>), dim3(ch.blocks), dim3(kBlock), 0, 0, SrcE64{in}, m, shift, shift + 8, 1, chunk, scr.partial);
        hipLaunchKernelGGL(k_radix_scan, dim3(kRadix), dim3(kBlock), 0, 0, scr.partial, ch.blocks, scr.totals);
        CK(hipDeviceSynchronize());
        {
            constexpr int kTile = 16 * 64 * 9;
            const uint64_t tiles = (m + kTile - 1) / kTile;
            auto f = [&] {
                hipMemsetAsync(g_status, 0, tiles * 256 * 4, 0);
                hipMemsetAsync(g_ticket, 0, 4, 0);
                hipLaunchKernelGGL((k_radix_pass<SrcKV, DstKV, 9, true, true, 16>), dim3((unsigned)dmin<uint64_t>(tiles, kMaxGrid)), dim3(1024), 0, 0,
                                   SrcKV{in, vin}, DstKV{ref, vref}, m, shift, 255u, (uint64_t)0, (const uint32_t*)nullptr, (const uint32_t*)g_totals, g_status, g_ticket, SegArgs{nullptr, nullptr, nullptr, 0, 0});
            };
            const float t = time_ms(f);
            printf("%-44s tile %5d            %.3f ms  %.2f TB/s\n", "shipped k_radix_pass<KV,9,16>", kTile, t, m * 24.0 / t * 1e-9);
        }
#define CHECKKV() do { hipLaunchKernelGGL(k_cmp64, dim3(1024), dim3(256), 0, 0, (const uint64_t*)out, (const uint64_t*)ref, m, g_bad); \
                       hipLaunchKernelGGL(k_cmp32, dim3(1024), dim3(256), 0, 0, (const uint32_t*)vout, (const uint32_t*)vref, m, g_bad); printf("    mismatches vs shipped: %llu\n", bad_count()); } while (0)
        run_lean<LSrcKV, LDstKV, 9, 16, 1>("lean KV 16w x 9 (1 wg/cu)", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        run_lean<LSrcKV, LDstKV, 10, 16, 1, 0, 1>("lean KV 16w x 10 out-batch 1", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        run_lean<LSrcKV, LDstKV, 10, 16, 1, 0, 2>("lean KV 16w x 10 out-batch 2", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        run_lean<LSrcKV, LDstKV, 11, 16, 1, 0, 1>("lean KV 16w x 11 out-batch 1", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        run_lean<LSrcKV, LDstKV, 12, 16, 1, 0, 1>("lean KV 16w x 12 out-batch 1", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        run_lean<LSrcKV, LDstKV, 12, 16, 1, 0, 2>("lean KV 16w x 12 out-batch 2", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        run_lean<LSrcKV, LDstKV, 12, 16, 1, 0, 3>("lean KV 16w x 12 out-batch 3", LSrcKV{in, vin}, LDstKV{out, vout}, m, shift, m * 24.0); CHECKKV();
        return 0;
        // 12-byte records in and out (one run per bucket instead of two)
        Rec12* rin = (Rec12*)ref;                       // (ref / vref are no longer needed as arrays: rebuild the check from them first)
        Rec12* rout = (Rec12*)out;
        Rec12* rtmp;
        CK(hipMalloc(&rtmp, m * 12));
        hipLaunchKernelGGL(k_kv_to_rec, dim3(2048), dim3(256), 0, 0, in, vin, rtmp, m);
        CK(hipDeviceSynchronize());
#define CHECKREC() do { hipLaunchKernelGGL(k_cmp_rec, dim3(1024), dim3(256), 0, 0, (const uint64_t*)ref, (const uint32_t*)vref, (const Rec12*)rout, m, g_bad); printf("    mismatches vs shipped: %llu\n", bad_count()); } while (0)
        (void)rin;
        run_lean<LSrcRec, LDstRec, 9, 16, 1>("lean REC12 16w x 9 (1 wg/cu)", LSrcRec{rtmp}, LDstRec{rout}, m, shift, m * 24.0); CHECKREC();
        run_lean<LSrcRec, LDstRec, 12, 8, 4>("lean REC12 8w x 12 (2 wg/cu)", LSrcRec{rtmp}, LDstRec{rout}, m, shift, m * 24.0); CHECKREC();
        run_pipe<LSrcRec, LDstRec, 9, 16>("pipe REC12 16w x 9", LSrcRec{rtmp}, LDstRec{rout}, m, shift, m * 24.0); CHECKREC();
        run_lean<LSrcKV, LDstRec, 9, 16, 1>("lean KV->REC12 16w x 9", LSrcKV{in, vin}, LDstRec{rout}, m, shift, m * 24.0); CHECKREC();
        run_pipe<LSrcKV, LDstRec, 9, 16>("pipe KV->REC12 16w x 9", LSrcKV{in, vin}, LDstRec{rout}, m, shift, m * 24.0); CHECKREC();
    }
    return 0;
}
