// tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h).
//
// Fiber scheduler: a workgroup's work-items are ucontext fibers run round-robin
// on the calling OS thread; blocks run one after another.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <chrono>
#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {

constexpr size_t kStackBytes = 128 * 1024;
constexpr int kMaxThreads = 1024;

struct WaveState {
    uint64_t table[2][64];
    uint64_t active[2];
    unsigned arrived = 0, gen = 0, live = 0;
};

// Minimal x86-64 SysV context switch (callee-saved registers + stack pointer);
// ucontext's swapcontext makes a sigprocmask syscall per switch, ~30x slower.
extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_ctx_switch, .-emu_ctx_switch
)");

struct BlockState {
    void* main_sp = nullptr;
    std::vector<void*> fibers;              // saved stack pointers
    std::vector<char> done;
    std::vector<unsigned> op_parity;        // per fiber: parity of its next wave collective
    std::vector<WaveState> waves;
    char* stacks = nullptr;
    unsigned nthreads = 0, live = 0, cur = 0;
    unsigned bar_arrived = 0, bar_gen = 0;
    const std::function<void()>* body = nullptr;
};

BlockState g;

void yield_to_scheduler() { emu_ctx_switch(&g.fibers[g.cur], g.main_sp); }

void release_wave_if_complete(WaveState& w)
{
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
}

void fiber_entry()
{
    (*g.body)();
    unsigned t = g.cur;
    g.done[t] = 1;
    g.live--;
    WaveState& w = g.waves[t >> 6];
    w.live--;
    release_wave_if_complete(w);
    if (g.live > 0 && g.bar_arrived == g.live) { g.bar_arrived = 0; g.bar_gen++; }
    void* dead;
    emu_ctx_switch(&dead, g.main_sp);           // never resumed
    abort();
}

void wave_rendezvous(WaveState& w)
{
    unsigned my_gen = w.gen;
    w.arrived++;
    if (w.arrived == w.live) { w.arrived = 0; w.gen++; return; }
    while (w.gen == my_gen) yield_to_scheduler();
}

}  // namespace

void emu_syncthreads()
{
    unsigned my_gen = g.bar_gen;
    g.bar_arrived++;
    if (g.bar_arrived == g.live) { g.bar_arrived = 0; g.bar_gen++; return; }
    while (g.bar_gen == my_gen) yield_to_scheduler();
}

void emu_wave_sync() { wave_rendezvous(g.waves[g.cur >> 6]); }

const uint64_t* emu_wave_exchange(uint64_t mine, uint64_t* active_mask)
{
    unsigned t = g.cur, lane = t & 63;
    WaveState& w = g.waves[t >> 6];
    unsigned par = g.op_parity[t];
    g.op_parity[t] = par ^ 1;
    if (w.arrived == 0) w.active[par] = 0;          // first arriver of this collective
    w.table[par][lane] = mine;
    w.active[par] |= 1ull << lane;
    wave_rendezvous(w);
    *active_mask = w.active[par];
    return w.table[par];
}

void emu_launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
    unsigned T = block.x * block.y * block.z;
    if (T == 0 || T > kMaxThreads || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1) {
        fprintf(stderr, "emu_launch: unsupported geometry\n");
        abort();
    }
    if (!g.stacks) g.stacks = (char*)malloc(kStackBytes * kMaxThreads);
    g.fibers.resize(T);
    g.done.assign(T, 0);
    g.op_parity.assign(T, 0);
    g.nthreads = T;
    g.body = &body;
    blockDim = block;
    gridDim = grid;
    for (unsigned b = 0; b < grid.x; b++) {
        blockIdx = dim3(b, 0, 0);
        g.live = T;
        g.bar_arrived = 0;
        g.bar_gen = 0;
        g.waves.assign((T + 63) / 64, WaveState());
        for (unsigned t = 0; t < T; t++) {
            g.done[t] = 0;
            g.op_parity[t] = 0;
            g.waves[t >> 6].live++;
            // fresh stack: [6 zeroed callee-saved regs][&fiber_entry][fake return address]
            uintptr_t top = ((uintptr_t)(g.stacks + (size_t)(t + 1) * kStackBytes)) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;
            *--sp = (void*)fiber_entry;
            for (int k = 0; k < 6; k++) *--sp = nullptr;
            g.fibers[t] = (void*)sp;
        }
        unsigned long spins = 0;
        while (g.live > 0) {
            for (unsigned t = 0; t < T; t++) {
                if (g.done[t]) continue;
                g.cur = t;
                threadIdx = dim3(t, 0, 0);
                emu_ctx_switch(&g.main_sp, g.fibers[t]);
            }
            if (++spins > 100000000ul) {
                fprintf(stderr, "emu_launch: deadlock (divergent barrier?)\n");
                abort();
            }
        }
    }
}

// ---- host API --------------------------------------------------------------
struct emu_event_s { std::chrono::steady_clock::time_point t; };

hipError_t hipMalloc(void** p, size_t bytes)
{
    *p = malloc(bytes ? bytes : 1);
    if (*p) memset(*p, 0xCD, bytes);               // poison: catch reads of uninitialised memory
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* st) { *st = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b)
{
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
