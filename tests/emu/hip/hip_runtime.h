// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-process emulator of the subset of the HIP programming model the
// kernels under suffix_amd/csrc/ use, so that their *logic* (indexing, scans,
// ranking, barrier placement) can be exercised by `pytest -m "not gpu"` in a
// container without a GPU.  The product never builds against this header: the
// shipped library is compiled by hipcc for gfx950 from the very same sources
// (no #ifdefs in them), and suffix_amd/ refuses to run without it.
//
// Model: one workgroup at a time; each work-item is a ucontext fiber;
// __syncthreads() and the wave-level collectives (__ballot/__shfl*/wave
// barrier) are rendezvous points between fibers.  Between rendezvous points a
// fiber runs alone, so a *missing* barrier or wave sync shows up as a wrong
// result here even where 64-wide lock-step would hide it on hardware.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define SFX_WAVES_PER_EU(lo, hi)                 /* occupancy requests mean nothing to the emulator */
#define SFX_EMULATED 1                            /* (read_back: a copy, no polled host word) */

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
typedef struct emu_stream_s* hipStream_t;
typedef struct emu_event_s* hipEvent_t;

hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }   /* a 4-CU device: grids of the persistent kernels */
hipError_t hipStreamCreate(hipStream_t* st);
constexpr unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { return hipStreamCreate(st); }
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipEventCreate(hipEvent_t* e);
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
template <class T> hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc((void**)p, bytes); }
constexpr unsigned hipHostMallocDefault = 0;
constexpr unsigned hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000;
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned = 0) { return hipMalloc(p, bytes); }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }

// ---- launch ----------------------------------------------------------------
void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu_launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

// ---- rendezvous ------------------------------------------------------------
void emu_syncthreads();
void emu_wave_sync();
// deposit 8 bytes for this lane, wait for the whole wave, return the wave's
// 64-entry table (valid until this lane's next collective) and the active mask
const uint64_t* emu_wave_exchange(uint64_t mine, uint64_t* active_mask);
#define __syncthreads() emu_syncthreads()
#define __builtin_amdgcn_wave_barrier() emu_wave_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)

static inline unsigned long long __ballot(int pred)
{
    uint64_t act;
    const uint64_t* t = emu_wave_exchange(pred ? 1 : 0, &act);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) if (((act >> l) & 1) && t[l]) m |= 1ull << l;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred)
{
    uint64_t act;
    const uint64_t* t = emu_wave_exchange(pred ? 1 : 0, &act);
    for (int l = 0; l < 64; l++) if (((act >> l) & 1) && !t[l]) return 0;
    return 1;
}
template <class T> static inline T emu_shfl_from(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shfl of <= 8 bytes only");
    uint64_t bits = 0, act;
    memcpy(&bits, &v, sizeof(T));
    const uint64_t* t = emu_wave_exchange(bits, &act);
    if (src < 0 || src > 63 || !((act >> src) & 1)) return v;
    T out;
    memcpy(&out, &t[src], sizeof(T));
    return out;
}
template <class T> static inline T __shfl(T v, int src_lane, int width = 64)
{
    (void)width;
    return emu_shfl_from(v, src_lane & 63);
}
template <class T> static inline T __shfl_up(T v, unsigned delta, int width = 64)
{
    (void)width;
    int lane = (int)(threadIdx.x & 63);
    return emu_shfl_from(v, lane - (int)delta);   // lanes < delta keep their own value
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64)
{
    (void)width;
    int lane = (int)(threadIdx.x & 63);
    int src = lane + (int)delta;
    return emu_shfl_from(v, src > 63 ? -1 : src);
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64)
{
    (void)width;
    int lane = (int)(threadIdx.x & 63);
    return emu_shfl_from(v, lane ^ mask);
}

// ---- bit ops / atomics -------------------------------------------------------
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned __funnelshift_l(unsigned lo, unsigned hi, unsigned shift) { shift &= 31u; return shift ? (hi << shift) | (lo >> (32u - shift)) : hi; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> static inline T atomicSub(T* p, T v) { T o = *p; *p = (T)(o - v); return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
static inline unsigned atomicAdd(unsigned* p, int v) { unsigned o = *p; *p = o + (unsigned)v; return o; }

// agent-scope atomics: one workgroup runs at a time, so plain accesses are exact
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T, class U> static inline void __hip_atomic_store(T* p, U v, int, int) { *p = (T)v; }
// v_mbcnt: bits of `mask` below this lane (+ add)
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add)
{
    unsigned lane = threadIdx.x & 63u;
    unsigned lt = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
    return add + (unsigned)__builtin_popcount(mask & lt);
}
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add)
{
    unsigned lane = threadIdx.x & 63u;
    unsigned lt = lane <= 32 ? 0u : ((1u << (lane - 32)) - 1u);
    return add + (unsigned)__builtin_popcount(mask & lt);
}

template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
