// lab/store_probe.hip -- DEVELOPMENT ONLY.  How should a tile's runs leave the CU?  Same bytes, same places, different lane -> address maps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t perm(uint64_t r, uint64_t mask, uint64_t n) { uint64_t p = r; do { p = (p * 0x9E3779B1ull + 12345ull) & mask; } while (p >= n); return p; }
__device__ __forceinline__ uint32_t hash32(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; return (uint32_t)x; }
// run r: R elements at perm(r) * PITCH + (hash % (PITCH - R + 1))
template <int R, int PITCH>
__device__ __forceinline__ uint64_t run_base(uint64_t r, uint64_t mask, uint64_t nruns) { return perm(r, mask, nruns) * PITCH + hash32(r) % (PITCH - R + 1); }
// (a) consecutive lanes = consecutive elements of consecutive runs (the shipped output loop)
template <int R, int PITCH>
__global__ void __launch_bounds__(1024) k_a(uint64_t* out, uint64_t nruns, uint64_t mask)
{
    const uint64_t m = nruns * R, stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < m; i += stride) {
        const uint64_t r = i / R; const unsigned e = (unsigned)(i - r * R);
        out[run_base<R, PITCH>(r, mask, nruns) + e] = i;
    }
}
// (b) every run gets VS virtual slots that start at its base rounded down to ALIGN elements: a 16-lane group writes inside one aligned line
template <int R, int PITCH, int VS, int ALIGN>
__global__ void __launch_bounds__(1024) k_b(uint64_t* out, uint64_t nruns, uint64_t mask)
{
    const uint64_t m = nruns * VS, stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < m; i += stride) {
        const uint64_t r = i / VS; const unsigned v = (unsigned)(i - r * VS);
        const uint64_t b = run_base<R, PITCH>(r, mask, nruns);
        const unsigned h = (unsigned)(b & (ALIGN - 1));
        if (v >= h && v < h + R) out[(b - h) + v] = i;
    }
}
// (c) like (a), but the runs themselves are placed at multiples of ALIGN elements and padded to whole blocks (what write combining would give)
template <int R, int PITCH>
__global__ void __launch_bounds__(1024) k_c(uint64_t* out, uint64_t nruns, uint64_t mask)
{
    const uint64_t m = nruns * R, stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < m; i += stride) {
        const uint64_t r = i / R; const unsigned e = (unsigned)(i - r * R);
        out[perm(r, mask, nruns) * PITCH + e] = i;
    }
}
template <class F> static float time_ms(F&& f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
template <int R, int PITCH> static int go(uint64_t* out, uint64_t total)
{
    const uint64_t nruns = total / PITCH; uint64_t p2 = 1; while (p2 < nruns) p2 <<= 1;
    const double gb = nruns * R * 8e-9;
    float ta = time_ms([&] { hipLaunchKernelGGL((k_a<R, PITCH>), dim3(1024), dim3(1024), 0, 0, out, nruns, p2 - 1); });
    float tb8 = time_ms([&] { hipLaunchKernelGGL((k_b<R, PITCH, ((R + 7 + 15) / 16) * 16, 8>), dim3(1024), dim3(1024), 0, 0, out, nruns, p2 - 1); });
    float tb16 = time_ms([&] { hipLaunchKernelGGL((k_b<R, PITCH, ((R + 15 + 15) / 16) * 16, 16>), dim3(1024), dim3(1024), 0, 0, out, nruns, p2 - 1); });
    float tb64 = time_ms([&] { hipLaunchKernelGGL((k_b<R, PITCH, ((R + 7 + 63) / 64) * 64, 8>), dim3(1024), dim3(1024), 0, 0, out, nruns, p2 - 1); });
    float tc = time_ms([&] { hipLaunchKernelGGL((k_c<R, PITCH>), dim3(1024), dim3(1024), 0, 0, out, nruns, p2 - 1); });
    printf("runs of %3d (pitch %3d, %.2f GB): (a) lanes=elements %.3f ms %.2f TB/s | (b) aligned windows: align 8 -> %.3f, align 16 -> %.3f, align 8 in 64-slot frames -> %.3f | (c) runs at line starts %.3f\n",
           R, PITCH, gb, ta, gb / ta, tb8, tb16, tb64, tc);
    return 0;
}
int main()
{
    const uint64_t total = 140000000ull;    // elements of the output array
    uint64_t* out; CK(hipMalloc(&out, total * 8 + 4096));
    go<44, 64>(out, total); go<36, 48>(out, total); go<88, 112>(out, total); go<64, 80>(out, total); go<16, 32>(out, total); go<176, 208>(out, total);
    return 0;
}
