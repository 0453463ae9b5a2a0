// lab/radix_lab.hip -- DEVELOPMENT ONLY (not part of the product): ablation of the chunked
// radix pass on 100 M random E64 elements.  Built by lab/Makefile, run on the GPU box.
#include <stdio.h>
#include <vector>
#include "../suffix_amd/csrc/sfx_radix.hip"

namespace sfx {
bool profile_on() { return false; }
void profile_begin(const char*, hipStream_t, double) {}
void profile_end(hipStream_t) {}
void note_hip_error(hipError_t e, const char* what, const char*, int) { fprintf(stderr, "HIP error %d at %s\n", (int)e, what); }

// ablation kernel: same skeleton as k_radix_pass (chunked), with phases switchable
//   MODE bit0: skip ranking (digits taken from the slot number, 16 per bucket and tile)
//   MODE bit1: skip global stores     MODE bit2: skip global loads
//   MODE bit3: skip the LDS reorder (store straight from registers at rank-derived places)
template <int KPT, int MODE, bool PREFETCH>
__global__ void __launch_bounds__(kBlock)
k_lab_pass(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t m, int shift, unsigned mask,
           uint64_t chunk, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ digit_total)
{
    constexpr int kTile = kBlock * KPT;
    __shared__ RadixSmem<KPT, false> s;
    const unsigned tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const unsigned long long mybit = 1ull << lane;
    unsigned par = 0;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock; k++) { s.flags[k][tid] = 0ull; s.cnt[k][tid] = 0u; }
    uint32_t my_head = block_scan_excl_1b(digit_total[tid], s.part, par) + hist[(uint64_t)tid * gridDim.x + blockIdx.x];
    if (MODE & 16) my_head &= ~15u;          // timing experiment: every run line-aligned (wrong places)
    uint64_t next = (uint64_t)blockIdx.x * chunk;
    const uint64_t limit = dmin<uint64_t>(m, next + chunk);
    __syncthreads();
    uint64_t nkey[KPT];
    auto load_tile = [&](uint64_t tile) {
#pragma unroll
        for (int r = 0; r < KPT; r++) {
            const unsigned idx = w * (kWave * KPT) + r * kWave + lane;
            if (MODE & 4) nkey[r] = (tile + idx) * 0x9E3779B97F4A7C15ull;
            else nkey[r] = (tile + idx < limit) ? in[tile + idx] : ~0ull;
        }
    };
    if (PREFETCH && next < limit) load_tile(next);
    for (uint64_t tile = next; tile < limit; tile += kTile) {
        const unsigned nvalid = (unsigned)dmin<uint64_t>(kTile, limit - tile);
        if (!PREFETCH) load_tile(tile);
        uint64_t key[KPT];
        uint32_t pos[KPT];
#pragma unroll
        for (int r = 0; r < KPT; r++) key[r] = nkey[r];
        if (PREFETCH && tile + kTile < limit) load_tile(tile + kTile);
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < KPT; r++) pos[r] = w * (kWave * KPT) + r * kWave + lane;
            s.off[tid] = my_head - tid * KPT;
            my_head += KPT;
            __syncthreads();
        } else {
#pragma unroll
            for (int r = 0; r < KPT; r++)
                pos[r] = rank_round<true>(digit_of(key[r], shift, mask), s.flags[w], s.cnt[w], mybit);
            __syncthreads();
            const uint32_t c0 = s.cnt[0][tid], c1 = s.cnt[1][tid], c2 = s.cnt[2][tid], c3 = s.cnt[3][tid];
            const uint32_t tile_count = c0 + c1 + c2 + c3;
            const uint32_t ex = block_scan_excl_1b(tile_count, s.part, par);
            s.cnt[0][tid] = ex; s.cnt[1][tid] = ex + c0; s.cnt[2][tid] = ex + c0 + c1; s.cnt[3][tid] = ex + c0 + c1 + c2;
            s.off[tid] = my_head - ex;
            my_head += tile_count;
            __syncthreads();
        }
        if (MODE & 8) {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned d = digit_of(key[r], shift, mask);
                const unsigned p = (MODE & 1) ? pos[r] : pos[r] + s.cnt[w][d];
                const uint32_t dest = s.off[(MODE & 1) ? (p / KPT) : d] + p;
                if (!(MODE & 2)) out[dest] = key[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned p = (MODE & 1) ? pos[r] : pos[r] + s.cnt[w][digit_of(key[r], shift, mask)];
                s.stage[p] = key[r];
            }
            __syncthreads();
            if (MODE & 32) {
                // two adjacent slots per lane: one 16-byte store when both belong to the same bucket
#pragma unroll
                for (int r = 0; r < KPT / 2; r++) {
                    const unsigned p = 2 * (r * kBlock + tid);
                    key[2 * r] = s.stage[p];
                    key[2 * r + 1] = s.stage[p + 1];
                }
#pragma unroll
                for (int r = 0; r < KPT / 2; r++) {
                    const unsigned p = 2 * (r * kBlock + tid);
                    const unsigned d0 = (MODE & 1) ? (p / KPT) : digit_of(key[2 * r], shift, mask);
                    const unsigned d1 = (MODE & 1) ? ((p + 1) / KPT) : digit_of(key[2 * r + 1], shift, mask);
                    const uint32_t a0 = s.off[d0] + p, a1 = s.off[d1] + p + 1;
                    if ((MODE & 64) || d0 == d1) {
                        struct alignas(16) P2 { uint64_t a, b; };
                        *reinterpret_cast<P2*>(out + ((MODE & 64) ? (a0 & ~1u) : a0)) = P2{key[2 * r], key[2 * r + 1]};
                    } else {
                        out[a0] = key[2 * r];
                        out[a1] = key[2 * r + 1];
                    }
                }
            } else {
#pragma unroll
            for (int r = 0; r < KPT; r++) key[r] = s.stage[r * kBlock + tid];
#pragma unroll
            for (int r = 0; r < KPT; r++) {
                const unsigned p = r * kBlock + tid;
                pos[r] = s.off[(MODE & 1) ? (p / KPT) : digit_of(key[r], shift, mask)] + p;
            }
#pragma unroll
            for (int r = 0; r < KPT; r++)
                if ((unsigned)(r * kBlock) + tid < nvalid && !(MODE & 2)) out[pos[r]] = key[r];
            }
        }
        if ((MODE & 2) && key[0] == 0x1234567ull && pos[0] == 77u) out[0] = key[1];   // keep values alive
#pragma unroll
        for (int k = 0; k < kWavesPerBlock; k++) s.cnt[k][tid] = 0u;
        __syncthreads();
    }
}

__global__ void k_fill(uint64_t* a, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        a[i] = (z << 32) | (uint32_t)i;
    }
}
// read-only streaming: U independent 16-byte loads per thread in flight
template <int U>
__global__ void __launch_bounds__(kBlock) k_read(const uint4* __restrict__ in, uint64_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; j++) v[j] = in[i + j * stride];
#pragma unroll
        for (int j = 0; j < U; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int U>
__global__ void __launch_bounds__(kBlock) k_write(uint4* __restrict__ out, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i < n16; i += stride) out[i] = uint4{(unsigned)i, 1u, 2u, 3u};
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
    return ms / reps;
}

int main()
{
    const uint64_t m = 100000000;
    uint64_t *in, *out; uint32_t* scratch;
    CK(hipMalloc(&in, m * 8)); CK(hipMalloc(&out, m * 8 + 4096));
    CK(hipMalloc(&scratch, radix_scratch_words(m) * 4));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, in, m);
    CK(hipDeviceSynchronize());
    RadixScratch scr(scratch, m);
    const int shift = 40; const unsigned mask = 255;
    printf("== streaming references (800 MB)\n");
    for (unsigned g : {1024u, 2048u, 4096u, 16384u}) {
        float t1 = time_ms([&] { hipLaunchKernelGGL((k_read<1>), dim3(g), dim3(kBlock), 0, 0, (const uint4*)in, m / 2, scratch); });
        float t4 = time_ms([&] { hipLaunchKernelGGL((k_read<4>), dim3(g), dim3(kBlock), 0, 0, (const uint4*)in, m / 2, scratch); });
        float t8 = time_ms([&] { hipLaunchKernelGGL((k_read<8>), dim3(g), dim3(kBlock), 0, 0, (const uint4*)in, m / 2, scratch); });
        float tw = time_ms([&] { hipLaunchKernelGGL((k_write<1>), dim3(g), dim3(kBlock), 0, 0, (uint4*)out, m / 2); });
        printf("grid %5u: read U1 %.3f ms (%.0f GB/s)  U4 %.3f (%.0f)  U8 %.3f (%.0f)  write %.3f (%.0f)\n", g, t1, 0.8 / t1 * 1e3,
               t4, 0.8 / t4 * 1e3, t8, 0.8 / t8 * 1e3, tw, 0.8 / tw * 1e3);
    }
#define RUN(KPT, MODE, PF, label)                                                                              \
    {                                                                                                          \
        Chunking ch = make_chunking(m, kBlock * KPT);                                                          \
        const uint64_t chunk = ch.tiles_per_block * kBlock * KPT;                                              \
        hipLaunchKernelGGL((k_radix_hist_chunk<SrcE64>), dim3(ch.blocks), dim3(kBlock), 0, 0, SrcE64{in}, m, shift, mask, chunk, scr.partial); \
        hipLaunchKernelGGL(k_radix_scan, dim3(kRadix), dim3(kBlock), 0, 0, scr.partial, ch.blocks, scr.totals); \
        float t = time_ms([&] { hipLaunchKernelGGL((k_lab_pass<KPT, MODE, PF>), dim3(ch.blocks), dim3(kBlock), 0, 0, in, out, m, shift, mask, chunk, (const uint32_t*)scr.partial, (const uint32_t*)scr.totals); }); \
        printf("KPT %2d mode %2d pf %d  %-34s %.3f ms\n", KPT, MODE, (int)PF, label, t);                         \
    }
    printf("== ablation, chunked E64 pass, 100 M elements\n");
    RUN(16, 0, true, "full");
    RUN(16, 0, false, "full, no prefetch");
    RUN(16, 1, true, "no ranking");
    RUN(16, 2, true, "no stores");
    RUN(16, 3, true, "no ranking, no stores");
    RUN(16, 4, true, "no loads");
    RUN(16, 5, true, "no loads, no ranking");
    RUN(16, 6, true, "no loads, no stores");
    RUN(16, 8, true, "no LDS reorder (scattered 8B)");
    RUN(16, 16, true, "full, line-aligned heads");
    RUN(16, 17, true, "no ranking, line-aligned heads");
    RUN(16, 32, true, "full, 16-byte pair stores");
    RUN(16, 33, true, "no ranking, 16-byte pair stores");
    RUN(16, 48, true, "full, aligned + pair stores");
    RUN(16, 49, true, "no ranking, aligned + pair stores");
    RUN(16, 48, false, "full, aligned + pair, no prefetch");
    RUN(16, 113, true, "no ranking, aligned, dwordx4 stores");
    RUN(16, 112, true, "full, forced-aligned dwordx4 stores");
    RUN(16, 97, true, "no ranking, unaligned heads, dwordx4");
    RUN(8, 113, true, "no ranking, aligned, dwordx4 stores");
    RUN(8, 17, true, "no ranking, line-aligned heads");
    RUN(8, 48, true, "full, aligned + pair stores");
    RUN(8, 0, true, "full");
    RUN(8, 1, true, "no ranking");
    RUN(8, 2, true, "no stores");
    RUN(8, 4, true, "no loads");
    {
        Chunking ch = make_chunking(m, kBlock * 16);
        const uint64_t chunk = ch.tiles_per_block * kBlock * 16;
        float th = time_ms([&] { hipLaunchKernelGGL((k_radix_hist_chunk<SrcE64>), dim3(ch.blocks), dim3(kBlock), 0, 0, SrcE64{in}, m, shift, mask, chunk, scr.partial); });
        printf("hist_chunk %.3f ms\n", th);
    }
    return 0;
}
