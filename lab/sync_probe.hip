// lab: what does a small device -> host read-back in the middle of a build cost, and what would a mapped host word that the
// device writes and the host polls cost instead?  (round 4: the headline build waits on four read-backs, ~19 us of idle GPU each)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_work(unsigned* out, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v; }
__global__ void k_post(const unsigned* src, volatile unsigned* host, unsigned seq)
{
    if (threadIdx.x == 0) {
        host[0] = src[0];
        __threadfence_system();
        host[16] = seq;                        // (its own 64-byte line)
    }
}
int main()
{
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned* d; CK(hipMalloc(&d, 256));
    unsigned* pinned; CK(hipHostMalloc(&pinned, 4096, hipHostMallocDefault));
    unsigned* mapped; CK(hipHostMalloc(&mapped, 4096, hipHostMallocMapped));
    unsigned* mapped_dev; CK(hipHostGetDevicePointer((void**)&mapped_dev, mapped, 0));
    std::memset(mapped, 0, 4096);
    const int iters = 2000;
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int mode = 0; mode < 4; mode++) {
        CK(hipStreamSynchronize(st));
        auto t0 = now();
        unsigned sum = 0;
        for (int i = 1; i <= iters; i++) {
            hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, st, d, (unsigned)i);
            if (mode == 0) {                   // as read_back does: async copy into pinned memory + stream synchronize
                CK(hipMemcpyAsync(pinned, d, 16, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                sum += pinned[0];
            } else if (mode == 1) {            // synchronous copy into pageable memory
                unsigned h[4];
                CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
                sum += h[0];
            } else if (mode == 2) {            // a kernel posts the words into mapped host memory, the host polls the sequence word
                hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, d, mapped_dev, (unsigned)i);
                while (((volatile unsigned*)mapped)[16] != (unsigned)i) {}
                sum += ((volatile unsigned*)mapped)[0];
            } else {                           // the same kernel, then stream synchronize (no polling)
                hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, d, mapped_dev, (unsigned)i);
                CK(hipStreamSynchronize(st));
                sum += mapped[0];
            }
        }
        CK(hipStreamSynchronize(st));
        const double us = std::chrono::duration<double, std::micro>(now() - t0).count() / iters;
        const char* names[4] = {"memcpyAsync -> pinned + streamSynchronize", "hipMemcpy -> pageable", "post kernel -> mapped host word, host polls",
                                "post kernel -> mapped host word, streamSynchronize"};
        printf("%-52s %7.2f us per (kernel + read-back)   [check %u]\n", names[mode], us, sum);
    }
    return 0;
}
