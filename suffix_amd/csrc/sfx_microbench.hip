// sfx_microbench.hip -- memory-system micro-benchmarks shipped with the engine
// (the scatter/gather roofline "must be measured", not assumed).
//
// Every kernel of the suffix sorter is a scan / scatter / gather over HBM, so what
// bounds it is not the 8 TB/s streaming peak but what the chip sustains for the
// access shapes the engine actually issues:
//   SFX_MB_COPY        streaming 16-byte copy                (radix pass, read side)
//   SFX_MB_SCATTER4    random 4-byte writes into a 4n array  (ISA[suffix] = rank,
//                      the reference's head_insert/tail_insert, src/table.rs:723-736)
//   SFX_MB_GATHER1     random 1-byte reads from an n array   (T[s-1] in induce, :429)
//   SFX_MB_GATHER4     random 4-byte reads from a 4n array   (ISA[suffix + h])
//   SFX_MB_RUNSCATTER  contiguous runs of `param` bytes written to pseudo-random,
//                      8-byte-aligned (param2 = 0) or run-aligned (param2 = 1)
//                      places: the write side of a radix pass, where one tile sends
//                      one run to each of 256 buckets
// Reported rate = algorithmic bytes (each logical element once) / HIP-event time.
#include "sfx_host.hpp"

namespace sfx {

// odd multiplier -> bijection on [0, 2^k)
__device__ __forceinline__ uint64_t mb_perm(uint64_t i, uint64_t mask)
{
    return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask;
}

__global__ void __launch_bounds__(kBlock)
k_mb_copy(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}

__global__ void __launch_bounds__(kBlock)
k_mb_scatter4(uint32_t* __restrict__ out, uint64_t n, uint64_t mask)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        out[mb_perm(i, mask)] = (uint32_t)i;
}

template <class T>
__global__ void __launch_bounds__(kBlock)
k_mb_gather(const T* __restrict__ in, uint64_t n, uint64_t mask, uint32_t* __restrict__ sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        acc += (uint32_t)in[mb_perm(i, mask)];
    if (acc == 0x12345678u) sink[0] = acc;          // keeps the loads alive
}

// element e (8 bytes) belongs to run e / L; run r lands at perm(r) * L (+ 1 element when
// not run-aligned, so runs straddle 64-byte sectors the way bucket heads do)
__global__ void __launch_bounds__(kBlock)
k_mb_runscatter(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t n8, unsigned log2_L,
                uint64_t run_mask, unsigned misalign)
{
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t L = 1ull << log2_L;
    for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < n8; e += stride) {
        const uint64_t r = e >> log2_L;
        out[(mb_perm(r, run_mask) << log2_L) + (e & (L - 1)) + misalign] = in[e];
    }
}

static int floor_log2(uint64_t v)
{
    int k = 0;
    while ((2ull << k) <= v) k++;
    return k;
}

int microbench(int kind, uint64_t bytes, int param, int param2, int reps, double* gbps_out)
{
    if (!gbps_out || bytes < (1u << 20) || reps < 1) return SFX_ERR_ARG;
    *gbps_out = 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SFX_ERR_NO_DEVICE;
    hipStream_t st = nullptr;
    void* a = nullptr;
    void* b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SFX_OK;
    double algo = 0.0;
    const unsigned grid = kMaxGrid;
    do {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = SFX_ERR_HIP; break; }
        // power-of-two element counts so that mb_perm is a bijection
        if (kind == SFX_MB_COPY) {
            const uint64_t n16 = bytes / 16;
            if (hipMalloc(&a, n16 * 16) != hipSuccess || hipMalloc(&b, n16 * 16) != hipSuccess) { rc = SFX_ERR_HIP; break; }
            (void)hipMemsetAsync(a, 1, n16 * 16, st);
            algo = 2.0 * n16 * 16;
            for (int r = -1; r < reps; r++) {
                if (r == 0) (void)hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_mb_copy, dim3(grid), dim3(kBlock), 0, st, (const uint4*)a, (uint4*)b, n16);
            }
        } else if (kind == SFX_MB_SCATTER4) {
            const uint64_t n = 1ull << floor_log2(bytes / 4);
            if (hipMalloc(&a, n * 4) != hipSuccess) { rc = SFX_ERR_HIP; break; }
            algo = 4.0 * n;
            for (int r = -1; r < reps; r++) {
                if (r == 0) (void)hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_mb_scatter4, dim3(grid), dim3(kBlock), 0, st, (uint32_t*)a, n, n - 1);
            }
        } else if (kind == SFX_MB_GATHER1 || kind == SFX_MB_GATHER4) {
            const uint64_t esz = kind == SFX_MB_GATHER1 ? 1 : 4;
            const uint64_t n = 1ull << floor_log2(bytes / esz);
            if (hipMalloc(&a, n * esz) != hipSuccess || hipMalloc(&b, 256) != hipSuccess) { rc = SFX_ERR_HIP; break; }
            (void)hipMemsetAsync(a, 1, n * esz, st);
            algo = (double)esz * n;
            for (int r = -1; r < reps; r++) {
                if (r == 0) (void)hipEventRecord(e0, st);
                if (esz == 1)
                    hipLaunchKernelGGL((k_mb_gather<uint8_t>), dim3(grid), dim3(kBlock), 0, st, (const uint8_t*)a, n, n - 1, (uint32_t*)b);
                else
                    hipLaunchKernelGGL((k_mb_gather<uint32_t>), dim3(grid), dim3(kBlock), 0, st, (const uint32_t*)a, n, n - 1, (uint32_t*)b);
            }
        } else if (kind == SFX_MB_RUNSCATTER) {
            if (param < 8 || (param & (param - 1))) { rc = SFX_ERR_ARG; break; }      // run bytes: power of two >= 8
            const unsigned log2_L = (unsigned)floor_log2((uint64_t)param / 8);
            const uint64_t n8 = 1ull << floor_log2(bytes / 8);
            const uint64_t runs = n8 >> log2_L;
            if (runs < 2) { rc = SFX_ERR_ARG; break; }
            if (hipMalloc(&a, n8 * 8) != hipSuccess || hipMalloc(&b, n8 * 8 + 64) != hipSuccess) { rc = SFX_ERR_HIP; break; }
            (void)hipMemsetAsync(a, 1, n8 * 8, st);
            algo = 16.0 * n8;                                                          // read + write
            for (int r = -1; r < reps; r++) {
                if (r == 0) (void)hipEventRecord(e0, st);
                hipLaunchKernelGGL(k_mb_runscatter, dim3(grid), dim3(kBlock), 0, st, (const uint64_t*)a, (uint64_t*)b, n8,
                                   log2_L, runs - 1, param2 ? 0u : 1u);
            }
        } else {
            rc = SFX_ERR_ARG;
            break;
        }
        (void)hipEventRecord(e1, st);
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = SFX_ERR_HIP; break; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) { rc = SFX_ERR_HIP; break; }
        *gbps_out = algo * reps / (ms * 1e-3) / 1e9;
    } while (0);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

}  // namespace sfx

extern "C" int sfx_microbench(int kind, uint64_t bytes, int param, int param2, int reps, double* gbps_out)
{
    return sfx::microbench(kind, bytes, param, param2, reps, gbps_out);
}
