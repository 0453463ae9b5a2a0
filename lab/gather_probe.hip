// lab/gather_probe.hip -- DEVELOPMENT ONLY.  One random 128-byte line per "member": does it matter how the wave asks for it?
//   (a) one lane per member, one 16-byte load (64 different lines per wave instruction) -- what k_deep_wave's key gather does
//   (b) 8 lanes per member, 16 bytes each = the whole line, 8 members per instruction, 8 instructions for 64 members
//   (c) like (a) but 8 bytes / 4 bytes per lane
//   (d) 2 / 4 lanes per member (32 / 64 bytes of the line)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// members m = 0 .. M-1, member m's line = mix(m) % nlines
template <int LPM, class V>     // lanes per member, vector type per lane
__global__ void __launch_bounds__(256) k_gather(const V* __restrict__ base, uint64_t nlines, uint64_t members, uint32_t* sink, int waves_per_eu_dummy)
{
    constexpr int kPerLine = 128 / sizeof(V);
    const uint64_t stride = (uint64_t)gridDim.x * 256 / LPM;
    const unsigned sub = threadIdx.x % LPM;
    uint32_t acc = 0;
    for (uint64_t m = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / LPM; m < members; m += stride * 4) {
        V v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t mm = m + u * stride;
            const uint64_t line = mix(mm) % nlines;
            v[u] = mm < members ? base[line * kPerLine + (sub % kPerLine)] : V{};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t* w = reinterpret_cast<const uint32_t*>(&v[u]); for (unsigned k = 0; k < sizeof(V) / 4; k++) acc ^= w[k]; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// (e) round 6: the same 4-byte gather with a non-temporal load / with system-scope bits -- does the L2 still fill a whole 128-byte line
// per miss?  (FETCH_SIZE per member from `rocprofv3 --pmc FETCH_SIZE` over this binary; kernel names tell the variants apart)
template <int MODE>
__global__ void __launch_bounds__(256) k_gather4_mode(const uint32_t* __restrict__ base, uint64_t nlines, uint64_t members, uint32_t* sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (uint64_t m = (uint64_t)blockIdx.x * 256 + threadIdx.x; m < members; m += stride * 4) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t mm = m + u * stride;
            const uint32_t* a = base + (mix(mm) % nlines) * 32;
            if (mm >= members) { v[u] = 0; continue; }
            if (MODE == 0) v[u] = *a;
            else if (MODE == 1) v[u] = __builtin_nontemporal_load(a);
            else if (MODE == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v[u]) : "v"(a) : "memory");
            else asm volatile("global_load_dword %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v[u]) : "v"(a) : "memory");
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= v[u];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <class F> static float time_ms(F&& f, int reps = 3)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
    const uint64_t bytes = 1ull << 30;          // 1 GiB of text-like data: far beyond the Infinity Cache
    const uint64_t nlines = bytes / 128, members = 1ull << 27;
    void* buf; uint32_t* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, bytes));
    const unsigned grid = 256 * 8;
#define RUN(LPM, V, label) { float t = time_ms([&] { hipLaunchKernelGGL((k_gather<LPM, V>), dim3(grid * (LPM >= 8 ? 4 : 1)), dim3(256), 0, 0, (const V*)buf, nlines, members, sink, 0); }); \
        printf("%-44s %.3f ms  %.1f G members/s  (%.2f TB/s of lines)\n", label, t, members / t * 1e-6, members * 128.0 / t * 1e-9); }
    RUN(1, uint4, "(a) 1 lane/member, 16 B");
    RUN(1, uint2, "(c) 1 lane/member, 8 B");
    RUN(1, uint32_t, "(c) 1 lane/member, 4 B");
    RUN(2, uint4, "(d) 2 lanes/member, 32 B");
    RUN(4, uint4, "(d) 4 lanes/member, 64 B");
    RUN(8, uint4, "(b) 8 lanes/member, whole line");
    RUN(16, uint2, "(b) 16 lanes/member x 8 B, whole line");
#define RUNM(MODE, label) { float t = time_ms([&] { hipLaunchKernelGGL((k_gather4_mode<MODE>), dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, nlines, members, sink); }); \
        printf("%-44s %.3f ms  %.1f G members/s\n", label, t, members / t * 1e-6); }
    RUNM(0, "(e) 4 B, plain load");
    RUNM(1, "(e) 4 B, non-temporal load");
    RUNM(2, "(e) 4 B, sc0 sc1 (waits per load)");
    RUNM(3, "(e) 4 B, sc0 sc1 nt (waits per load)");
    // the same plain gather from memory allocated uncached / fine-grained
    for (int kind = 0; kind < 2; kind++) {
        void* ub = nullptr;
        if (hipExtMallocWithFlags(&ub, bytes, kind == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained) != hipSuccess) { printf("allocation kind %d refused\n", kind); continue; }
        CK(hipMemset(ub, 1, bytes));
        float t = time_ms([&] { hipLaunchKernelGGL((k_gather4_mode<0>), dim3(grid), dim3(256), 0, 0, (const uint32_t*)ub, nlines, members, sink); });
        printf("%-44s %.3f ms  %.1f G members/s\n", kind == 0 ? "(f) 4 B, plain load, hipDeviceMallocUncached" : "(f) 4 B, plain load, hipDeviceMallocFinegrained", t, members / t * 1e-6);
        hipFree(ub);
    }
    return 0;
}
