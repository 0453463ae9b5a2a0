// lab/partition_lab.hip -- DEVELOPMENT ONLY.  Round 4: the non-stable partition pass of the hybrid route (k_partition) on 100 M
// random E64 elements, whole and with phases switched off (-DSFX_PART_ABL=<bits>: 1 no LDS atomics, 2 no global atomics,
// 4 no stores, 8 no loads), next to the one-sweep pass on the same elements.
#include <stdio.h>
#include <vector>
#ifndef LAB_NW
#define LAB_NW 16
#endif
#ifndef LAB_KPT
#define LAB_KPT 16
#endif
#include "../suffix_amd/csrc/sfx_radix.hip"
namespace sfx {
bool profile_on() { return false; }
void profile_begin(const char*, hipStream_t, double) {}
void profile_end(hipStream_t) {}
void note_hip_error(hipError_t e, const char* what, const char*, int) { fprintf(stderr, "HIP error %d at %s\n", (int)e, what); }
__global__ void k_fill(uint64_t* a, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
        a[i] = (z << 32) | (uint32_t)i;
    }
}
__global__ void k_hist8(const uint64_t* a, uint64_t n, int shift, uint32_t* h)
{
    __shared__ uint32_t l[256];
    l[threadIdx.x] = 0; __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) atomicAdd(&l[(a[i] >> shift) & 255u], 1u);
    __syncthreads();
    atomicAdd(&h[threadIdx.x], l[threadIdx.x]);
}
}  // namespace sfx
using namespace sfx;
int main(int argc, char** argv)
{
    const uint64_t m = 100000000ull;
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    uint64_t *a, *b; uint32_t *h, *cur, *bst;
    hipMalloc(&a, m * 8); hipMalloc(&b, m * 8); hipMalloc(&h, 256 * 4); hipMalloc(&cur, 256 * kCursorPad * 4); hipMalloc(&bst, 65537 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, a, m);
    hipMemset(h, 0, 1024);
    hipLaunchKernelGGL(k_hist8, dim3(2048), dim3(256), 0, 0, (const uint64_t*)a, m, 56, h);
    std::vector<uint32_t> hh(256), start(256), pad(256 * kCursorPad, 0);
    hipMemcpy(hh.data(), h, 1024, hipMemcpyDeviceToHost);
    uint32_t run = 0;
    for (int d = 0; d < 256; d++) { start[d] = run; run += hh[d]; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        for (int d = 0; d < 256; d++) pad[d * kCursorPad] = start[d];
        hipMemcpy(cur, pad.data(), pad.size() * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_partition<SrcE64, LAB_KPT, LAB_NW, false>), dim3(grid), dim3(LAB_NW * 64), 0, 0, SrcE64{a}, b, m, 56, cur, (const uint32_t*)bst);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("k_partition (top 8 bits, abl %d, grid %d): %.3f ms\n", (int)SFX_PART_ABL, grid, ms);
    }
    if (hipGetLastError() != hipSuccess) printf("error\n");
    return 0;
}
