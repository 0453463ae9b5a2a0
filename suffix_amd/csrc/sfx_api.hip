// sfx_api.hip -- the extern "C" boundary (include/suffix_hip.h): argument
// checking, host<->HBM staging for the host-pointer entry points, the
// device-resident index handle, error text and the event profiler.
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "sfx_host.hpp"

namespace sfx {

// ---- error capture ---------------------------------------------------------------
static thread_local char tls_hip_error[512] = "";

void note_hip_error(hipError_t e, const char* what, const char* file, int line)
{
    snprintf(tls_hip_error, sizeof(tls_hip_error), "%s (%d) at %s:%d in `%s`", hipGetErrorString(e),
             (int)e, file, line, what);
}

sfx_build_stats& tls_build_stats()
{
    static thread_local sfx_build_stats s;
    return s;
}

// ---- profiler ---------------------------------------------------------------------
struct ProfRecord {
    const char* name;
    double bytes;
    hipEvent_t a, b;
};
static bool g_profile = false;
static std::mutex g_prof_mu;
static std::vector<ProfRecord> g_prof_open;      // recorded, not yet folded
struct ProfStat { std::string name; uint64_t launches; double ms, bytes; };
static std::vector<ProfStat> g_prof_stats;

bool profile_on() { return g_profile; }

void profile_begin(const char* name, hipStream_t st, double algo_bytes)
{
    ProfRecord r;
    r.name = name;
    r.bytes = algo_bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_open.push_back(r);
}
void profile_end(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_open.empty()) (void)hipEventRecord(g_prof_open.back().b, st);
}
static void profile_fold()
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRecord& r : g_prof_open) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.b);
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        ProfStat* s = nullptr;
        for (ProfStat& t : g_prof_stats) if (t.name == r.name) { s = &t; break; }
        if (!s) { g_prof_stats.push_back(ProfStat{r.name, 0, 0.0, 0.0}); s = &g_prof_stats.back(); }
        s->launches++;
        s->ms += ms;
        s->bytes += r.bytes;
    }
    g_prof_open.clear();
}

// ---- device buffers of the host-pointer entry points ----------------------------------
// A small mutex-guarded pool ("workspace from a mutex-guarded pool"): a buffer
// released by one call is handed to the next call on the same device that asks for at most
// that size and at least half of it, so a loop of SuffixTable::new over similar texts does
// not pay hipMalloc/hipFree of ~50 n bytes every time.  At most kPoolMaxBuffers buffers
// are kept; sfx_release_cached_buffers() returns them to the driver.
struct PooledBuf { void* p; uint64_t bytes; int device; };
static std::mutex g_pool_mu;
static std::vector<PooledBuf> g_pool;
constexpr size_t kPoolMaxBuffers = 8;

static void* pool_take(uint64_t bytes, int device, uint64_t* got_bytes)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); i++) {
        const PooledBuf& b = g_pool[i];
        if (b.device == device && b.bytes >= bytes && b.bytes / 2 <= bytes &&
            (best == g_pool.size() || b.bytes < g_pool[best].bytes))
            best = i;
    }
    if (best == g_pool.size()) return nullptr;
    void* p = g_pool[best].p;
    *got_bytes = g_pool[best].bytes;
    g_pool.erase(g_pool.begin() + (long)best);
    return p;
}
static void pool_give(void* p, uint64_t bytes, int device)
{
    void* evict = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool.push_back(PooledBuf{p, bytes, device});
        if (g_pool.size() > kPoolMaxBuffers) {                 // drop the oldest
            evict = g_pool.front().p;
            g_pool.erase(g_pool.begin());
        }
    }
    if (evict) (void)hipFree(evict);
}

struct DevBuf {
    void* p = nullptr;
    uint64_t bytes = 0;
    int device = 0;
    ~DevBuf() { release(); }
    int alloc(uint64_t want)
    {
        if (want == 0) want = 1;
        (void)hipGetDevice(&device);
        p = pool_take(want, device, &bytes);
        if (p) return SFX_OK;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            // the pool may be what is in the way: give everything back and retry once
            sfx_release_cached_buffers();
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) { note_hip_error(e, "hipMalloc", __FILE__, __LINE__); p = nullptr; return SFX_ERR_HIP; }
        bytes = want;
        return SFX_OK;
    }
    void release()
    {
        if (p) pool_give(p, bytes, device);
        p = nullptr;
    }
};

// Stream of the host-pointer entry points: one non-blocking stream per calling thread, created on first
// use, so that SuffixTable::new from several threads runs concurrently on the device instead of
// serialising on the NULL stream ("per-call stream").  nullptr if creation fails.
// A stream belongs to the device that was current when it was made: one per (thread, device), looked up by the
// device current NOW (suffix_amd/device.py switches devices per call).
constexpr int kMaxDevices = 16;
// -1: the ordinal does not fit the per-device tables (or cannot be told): such a call runs on the NULL stream and without
// cached scratch -- never on a stream, event or buffer that was made on another device
static int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (d >= 0 && d < kMaxDevices) ? d : -1;
}
static hipStream_t call_stream()
{
    thread_local hipStream_t st[kMaxDevices] = {};
    thread_local bool tried[kMaxDevices] = {};
    const int d = current_device();
    if (d < 0) return nullptr;
    if (!tried[d]) {
        tried[d] = true;
        if (hipStreamCreateWithFlags(&st[d], hipStreamNonBlocking) != hipSuccess) {
            st[d] = nullptr;
            (void)hipGetLastError();
        }
    }
    return st[d];
}
// pooled device buffers must not go back to the pool while work that uses them may still be queued
struct StreamDrain {
    hipStream_t st;
    ~StreamDrain() { (void)hipStreamSynchronize(st); }
};

static int check_device()
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return SFX_ERR_NO_DEVICE;
    return SFX_OK;
}

}  // namespace sfx

using namespace sfx;

struct sfx_index {
    uint8_t* d_text = nullptr;
    uint32_t* d_sa = nullptr;
    uint64_t n = 0;
    bool owns_arrays = true;        // false: created over the caller's device arrays (sfx_index_create_dev)
    // bucket directory (sfx_query.hip): first k symbols of a query -> its stretch of the suffix array
    uint32_t* d_dir = nullptr;
    uint16_t* d_lut = nullptr;      // 256 entries: byte -> symbol code + 1, 0 = byte absent from the text
    int bits = 0, k = 0, dbits = 0;
    uint64_t entries = 0;
    // prefix-key B+tree (sfx_query.hip): 8.6 n bytes; when it cannot be allocated the directory alone serves
    uint64_t* d_tree = nullptr;
    uint64_t tree_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int tree_levels = 0;
};

namespace sfx {
// alphabet -> dense codes, directory shape, device build; SFX_ERR_ARG if the table holds an entry >= n
static int index_build_directory(sfx_index* ix, hipStream_t st)
{
    if (ix->n == 0) return SFX_OK;
    void* small = nullptr;
    SFX_HIP(hipMalloc(&small, 4096));
    unsigned long long bins[256];
    int rc = byte_presence_host(ix->d_text, ix->n, small, bins, st);
    (void)hipFree(small);
    if (rc != SFX_OK) return rc;
    uint16_t lut[256];
    unsigned sigma = 0;
    for (int c = 0; c < 256; c++) lut[c] = bins[c] ? (uint16_t)(++sigma) : (uint16_t)0;
    ix->bits = bits_for(sigma > 1 ? sigma - 1 : 1);
    SFX_TRY(dir_shape(ix->n, ix->bits, &ix->k, &ix->dbits, &ix->entries));
    SFX_HIP(hipMalloc((void**)&ix->d_dir, ix->entries * sizeof(uint32_t)));
    SFX_HIP(hipMalloc((void**)&ix->d_lut, 256 * sizeof(uint16_t)));
    uint32_t* scratch = nullptr;
    SFX_HIP(hipMalloc((void**)&scratch, (dir_scratch_words(ix->entries) + 2) * sizeof(uint32_t)));
    uint64_t bad = 0;
    rc = dir_build_dev(ix->d_text, ix->n, ix->d_sa, lut, ix->bits, ix->k, ix->dbits, ix->entries, ix->d_lut, ix->d_dir, scratch, st, &bad);
    (void)hipFree(scratch);
    if (rc != SFX_OK) return rc;
    if (bad) return SFX_ERR_ARG;
    // SFX_INDEX_TREE=0 (development): directory only
    static const bool want_tree = [] { const char* e = dev_env("SFX_INDEX_TREE"); return !e || atoi(e) != 0; }();
    if (want_tree && hipMalloc((void**)&ix->d_tree, key_tree_words(ix->n) * sizeof(uint64_t)) == hipSuccess) {
        rc = key_tree_build_dev(ix->d_text, ix->n, ix->d_sa, ix->d_tree, ix->tree_off, &ix->tree_levels, st);
        if (rc != SFX_OK) return rc;
        // the index is handed to callers who will query it on OTHER streams: the tree must be complete, not queued
        SFX_HIP(hipStreamSynchronize(st));
    } else {
        ix->d_tree = nullptr;
        (void)hipGetLastError();
    }
    return SFX_OK;
}
}  // namespace sfx

extern "C" {

const char* sfx_strerror(int status)
{
    switch (status) {
    case SFX_OK: return "ok";
    case SFX_ERR_ARG: return "invalid argument";
    case SFX_ERR_TOO_LARGE: return "text longer than u32::MAX bytes";
    case SFX_ERR_NO_DEVICE: return "no HIP device available";
    case SFX_ERR_HIP: return "HIP runtime error (see sfx_last_hip_error)";
    case SFX_ERR_WORKSPACE: return "device workspace too small";
    case SFX_ERR_INTERNAL: return "internal invariant violated";
    case SFX_ERR_NEEDS_RANKS: return "slice needs rank refinement: build the whole suffix array";
    default: return "unknown status";
    }
}

int sfx_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

const char* sfx_last_hip_error(void) { return tls_hip_error; }

// ---- suffix array --------------------------------------------------------------------
uint64_t sfx_sa_workspace_bytes(uint64_t n) { return sa_workspace_bytes(n); }

int sfx_build_sa_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, void* d_workspace,
                         uint64_t workspace_bytes, void* stream)
{
    return build_sa_u32_dev(d_text, n, d_sa, d_workspace, workspace_bytes, (hipStream_t)stream);
}

int sfx_build_sa_u32(const uint8_t* text, uint64_t n, uint32_t* sa_out)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!text || !sa_out) return SFX_ERR_ARG;
    SFX_TRY(check_device());
    DevBuf dt, ds, dw;
    uint64_t wsb = sa_workspace_bytes(n);
    SFX_TRY(dt.alloc(n));
    SFX_TRY(ds.alloc(n * sizeof(uint32_t)));
    SFX_TRY(dw.alloc(wsb));
    hipStream_t st = call_stream();
    StreamDrain drain{st};            // (declared after the buffers: runs before they return to the pool)
    SFX_HIP(hipMemcpyAsync(dt.p, text, n, hipMemcpyHostToDevice, st));
    SFX_TRY(build_sa_u32_dev((const uint8_t*)dt.p, n, (uint32_t*)ds.p, dw.p, wsb, st));
    SFX_HIP(hipMemcpyAsync(sa_out, ds.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SFX_HIP(hipStreamSynchronize(st));
    return SFX_OK;
}

// u64 index array (BASELINE config 4).  Positions fit u32 (n <= u32::MAX, :380), so the u32
// engine runs and the result is widened on the device before the copy back.
int sfx_build_sa_u64(const uint8_t* text, uint64_t n, uint64_t* sa_out)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!text || !sa_out) return SFX_ERR_ARG;
    SFX_TRY(check_device());
    DevBuf dt, ds, dw, d64;
    uint64_t wsb = sa_workspace_bytes(n);
    SFX_TRY(dt.alloc(n));
    SFX_TRY(ds.alloc(n * sizeof(uint32_t)));
    SFX_TRY(dw.alloc(wsb));
    hipStream_t st = call_stream();
    StreamDrain drain{st};            // (declared after the buffers: runs before they return to the pool)
    SFX_HIP(hipMemcpyAsync(dt.p, text, n, hipMemcpyHostToDevice, st));
    SFX_TRY(build_sa_u32_dev((const uint8_t*)dt.p, n, (uint32_t*)ds.p, dw.p, wsb, st));
    SFX_HIP(hipStreamSynchronize(st));
    dw.release();                                      // the workspace is larger than the u64 array
    SFX_TRY(d64.alloc(n * sizeof(uint64_t)));
    SFX_TRY(widen_u32_to_u64_dev((const uint32_t*)ds.p, n, (uint64_t*)d64.p, st));
    SFX_HIP(hipMemcpyAsync(sa_out, d64.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    SFX_HIP(hipStreamSynchronize(st));
    return SFX_OK;
}

// ---- LCP ---------------------------------------------------------------------------------
uint64_t sfx_lcp_workspace_bytes(uint64_t n) { return lcp_workspace_bytes(n); }

int sfx_build_lcp_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, uint32_t* d_lcp,
                          void* d_workspace, uint64_t workspace_bytes, void* stream)
{
    return build_lcp_u32_dev(d_text, n, d_sa, d_lcp, d_workspace, workspace_bytes, (hipStream_t)stream);
}

int sfx_build_lcp_u32(const uint8_t* text, uint64_t n, const uint32_t* sa, uint32_t* lcp_out)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!text || !sa || !lcp_out) return SFX_ERR_ARG;
    SFX_TRY(check_device());
    DevBuf dt, ds, dl, dw;
    uint64_t wsb = lcp_workspace_bytes(n);
    SFX_TRY(dt.alloc(n));
    SFX_TRY(ds.alloc(n * sizeof(uint32_t)));
    SFX_TRY(dl.alloc(n * sizeof(uint32_t)));
    SFX_TRY(dw.alloc(wsb));
    hipStream_t st = call_stream();
    StreamDrain drain{st};            // (declared after the buffers: runs before they return to the pool)
    SFX_HIP(hipMemcpyAsync(dt.p, text, n, hipMemcpyHostToDevice, st));
    SFX_HIP(hipMemcpyAsync(ds.p, sa, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    SFX_TRY(build_lcp_u32_dev((const uint8_t*)dt.p, n, (const uint32_t*)ds.p, (uint32_t*)dl.p, dw.p,
                              wsb, st));
    SFX_HIP(hipMemcpyAsync(lcp_out, dl.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SFX_HIP(hipStreamSynchronize(st));
    return SFX_OK;
}

// ---- SA + LCP in one call --------------------------------------------------------------------
uint64_t sfx_sa_lcp_workspace_bytes(uint64_t n) { return sa_lcp_workspace_bytes(n); }

int sfx_build_sa_lcp_u32_dev(const uint8_t* d_text, uint64_t n, uint32_t* d_sa, uint32_t* d_lcp, void* d_workspace,
                             uint64_t workspace_bytes, void* stream)
{
    return build_sa_lcp_u32_dev(d_text, n, d_sa, d_lcp, d_workspace, workspace_bytes, (hipStream_t)stream);
}

int sfx_build_sa_lcp_u32(const uint8_t* text, uint64_t n, uint32_t* sa_out, uint32_t* lcp_out)
{
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n == 0) return SFX_OK;
    if (!text || !sa_out || !lcp_out) return SFX_ERR_ARG;
    SFX_TRY(check_device());
    DevBuf dt, ds, dl, dw;
    uint64_t wsb = sa_lcp_workspace_bytes(n);
    SFX_TRY(dt.alloc(n));
    SFX_TRY(ds.alloc(n * sizeof(uint32_t)));
    SFX_TRY(dl.alloc(n * sizeof(uint32_t)));
    SFX_TRY(dw.alloc(wsb));
    hipStream_t st = call_stream();
    StreamDrain drain{st};            // (declared after the buffers: runs before they return to the pool)
    SFX_HIP(hipMemcpyAsync(dt.p, text, n, hipMemcpyHostToDevice, st));
    SFX_TRY(build_sa_lcp_u32_dev((const uint8_t*)dt.p, n, (uint32_t*)ds.p, (uint32_t*)dl.p, dw.p, wsb, st));
    SFX_HIP(hipMemcpyAsync(sa_out, ds.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SFX_HIP(hipMemcpyAsync(lcp_out, dl.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SFX_HIP(hipStreamSynchronize(st));
    return SFX_OK;
}

// ---- index + queries ---------------------------------------------------------------------
int sfx_index_create(const uint8_t* text, uint64_t n, const uint32_t* sa, sfx_index** out)
{
    if (!out) return SFX_ERR_ARG;
    *out = nullptr;
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n && !text) return SFX_ERR_ARG;
    SFX_TRY(check_device());
    sfx_index* ix = new sfx_index();
    ix->n = n;
    int rc = SFX_OK;
    if (n) {
        DevBuf dw;
        hipStream_t st = call_stream();
        StreamDrain drain{st};        // (declared after the buffer: runs before it returns to the pool)
        auto hip_ok = [&](hipError_t e, const char* what) {
            if (e == hipSuccess) return true;
            note_hip_error(e, what, __FILE__, __LINE__);
            rc = SFX_ERR_HIP;
            return false;
        };
        do {
            if (!hip_ok(hipMalloc((void**)&ix->d_text, n), "hipMalloc(text)") ||
                !hip_ok(hipMalloc((void**)&ix->d_sa, n * sizeof(uint32_t)), "hipMalloc(sa)")) break;
            if (!hip_ok(hipMemcpyAsync(ix->d_text, text, n, hipMemcpyHostToDevice, st), "H2D text")) break;
            if (sa) {
                if (!hip_ok(hipMemcpyAsync(ix->d_sa, sa, n * sizeof(uint32_t), hipMemcpyHostToDevice, st), "H2D sa")) break;
            } else {
                uint64_t wsb = sa_workspace_bytes(n);
                rc = dw.alloc(wsb);
                if (rc != SFX_OK) break;
                rc = build_sa_u32_dev(ix->d_text, n, ix->d_sa, dw.p, wsb, st);
                if (rc != SFX_OK) break;
            }
            // the directory build also checks every table entry against n (an unchecked from_parts table
            // must not make the kernels read out of bounds: SFX_ERR_ARG instead of the reference's panic)
            rc = index_build_directory(ix, st);
            if (rc != SFX_OK) break;
            if (!hip_ok(hipStreamSynchronize(st), "sync")) break;
        } while (0);
        if (rc != SFX_OK) (void)hipStreamSynchronize(st);          // nothing may still use the buffers we release
    }
    if (rc != SFX_OK) { sfx_index_destroy(ix); return rc; }
    *out = ix;
    return SFX_OK;
}

// The same over arrays that already live in HBM (not copied: the caller keeps d_text / d_sa alive and
// unchanged for the life of the index); builds only the bucket directory.
int sfx_index_create_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa, void* stream, sfx_index** out)
{
    if (!out) return SFX_ERR_ARG;
    *out = nullptr;
    if (n > 0xFFFFFFFFull) return SFX_ERR_TOO_LARGE;
    if (n && (!d_text || !d_sa)) return SFX_ERR_ARG;
    SFX_TRY(check_device());
    sfx_index* ix = new sfx_index();
    ix->n = n;
    ix->owns_arrays = false;
    ix->d_text = const_cast<uint8_t*>(d_text);
    ix->d_sa = const_cast<uint32_t*>(d_sa);
    int rc = index_build_directory(ix, (hipStream_t)stream);
    if (rc != SFX_OK) { (void)hipStreamSynchronize((hipStream_t)stream); sfx_index_destroy(ix); return rc; }
    *out = ix;
    return SFX_OK;
}

int sfx_index_query_dev(const sfx_index* ix, const uint8_t* d_qbytes, const uint64_t* d_qoff, uint64_t nq,
                        uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any, void* stream)
{
    if (!ix) return SFX_ERR_ARG;
    if (ix->n == 0 || !ix->d_dir)
        return query_batch_dev(ix->d_text, ix->n, ix->d_sa, ix->n, d_qbytes, d_qoff, nq, d_start, d_end, d_found, d_any,
                               (hipStream_t)stream);
    if (ix->d_tree) {
        // SFX_QUERY_ORDER=1 (development): answer large batches in the order of their first 8 bytes, so that
        // neighbouring lanes share tree nodes and probes.  Measured on config 5's 10^6 queries: the search kernel
        // 1.39 -> 1.23 ms, the 8-pass sort of the (key, query) pairs 0.24 ms -- not worth it, off by default
        static const bool want_order = [] { const char* e = dev_env("SFX_QUERY_ORDER"); return e && atoi(e) != 0; }();
        // scratch of phase 2 (the list of queries that go on; the ordering), kept across calls per (thread, device): no
        // allocation and no host synchronisation on the hot path.  The previous batch may still be using it on another
        // stream (whose handle may be gone by now): it left an EVENT behind, and this batch's stream waits on that -- on
        // the device.  Without scratch the batch is answered in one phase.
        struct QueryScratch { void* p = nullptr; uint64_t bytes = 0; hipEvent_t done = nullptr; bool used = false; };
        thread_local QueryScratch scs[kMaxDevices];
        const int dev = current_device();
        QueryScratch none;
        QueryScratch& sc = dev >= 0 ? scs[dev] : none;
        void* os = nullptr;
        if (dev >= 0 && nq >= query_two_phase_min()) {
            const uint64_t need = query_scratch_bytes(nq, want_order);
            if (!sc.done && hipEventCreateWithFlags(&sc.done, hipEventDisableTiming) != hipSuccess) { sc.done = nullptr; (void)hipGetLastError(); }
            if (sc.done) {
                if (sc.bytes < need) {
                    if (sc.p) { if (sc.used) (void)hipEventSynchronize(sc.done); (void)hipFree(sc.p); sc.p = nullptr; sc.bytes = 0; sc.used = false; }
                    if (hipMalloc(&sc.p, need) == hipSuccess) sc.bytes = need; else { sc.p = nullptr; (void)hipGetLastError(); }
                }
                if (sc.p) {
                    if (sc.used) (void)hipStreamWaitEvent((hipStream_t)stream, sc.done, 0);
                    os = sc.p;
                }
            }
        }
        const int qrc = query_batch_tree_dev(ix->d_text, ix->n, ix->d_sa, ix->d_tree, ix->tree_off, ix->tree_levels, d_qbytes, d_qoff, nq,
                                             d_start, d_end, d_found, d_any, (hipStream_t)stream, os, want_order, ix->d_dir, ix->d_lut,
                                             ix->bits, ix->k, ix->dbits);
        if (os) { (void)hipEventRecord(sc.done, (hipStream_t)stream); sc.used = true; }
        return qrc;
    }
    return query_batch_dir_dev(ix->d_text, ix->n, ix->d_sa, ix->d_dir, ix->d_lut, ix->bits, ix->k, ix->dbits, d_qbytes, d_qoff, nq,
                               d_start, d_end, d_found, d_any, (hipStream_t)stream);
}

void sfx_index_destroy(sfx_index* ix)
{
    if (!ix) return;
    if (ix->owns_arrays) {
        if (ix->d_text) (void)hipFree(ix->d_text);
        if (ix->d_sa) (void)hipFree(ix->d_sa);
    }
    if (ix->d_dir) (void)hipFree(ix->d_dir);
    if (ix->d_tree) (void)hipFree(ix->d_tree);
    if (ix->d_lut) (void)hipFree(ix->d_lut);
    delete ix;
}

uint64_t sfx_index_len(const sfx_index* ix) { return ix ? ix->n : 0; }

int sfx_index_table(const sfx_index* ix, uint32_t* sa_out)
{
    if (!ix || (ix->n && !sa_out)) return SFX_ERR_ARG;
    if (ix->n == 0) return SFX_OK;
    SFX_HIP(hipMemcpy(sa_out, ix->d_sa, ix->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SFX_OK;
}

int sfx_query_batch_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa,
                        const uint8_t* d_qbytes, const uint64_t* d_qoff, uint64_t nq,
                        uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any,
                        void* stream)
{
    return query_batch_dev(d_text, n, d_sa, n, d_qbytes, d_qoff, nq, d_start, d_end, d_found, d_any,
                           (hipStream_t)stream);
}
int sfx_query_batch_range_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa_part, uint64_t count,
                              const uint8_t* d_qbytes, const uint64_t* d_qoff, uint64_t nq,
                              uint32_t* d_start, uint32_t* d_end, uint8_t* d_found, uint32_t* d_any,
                              void* stream)
{
    return query_batch_dev(d_text, n, d_sa_part, count, d_qbytes, d_qoff, nq, d_start, d_end, d_found, d_any,
                           (hipStream_t)stream);
}
int sfx_build_lcp_range_u32_dev(const uint8_t* d_text, uint64_t n, const uint32_t* d_sa_part, uint64_t count,
                                uint32_t prev_suffix, uint32_t* d_lcp_part, void* stream)
{
    return build_lcp_range_u32_dev(d_text, n, d_sa_part, count, prev_suffix, d_lcp_part, (hipStream_t)stream);
}
int sfx_widen_u32_to_u64_dev(const uint32_t* d_in, uint64_t count, uint64_t* d_out, void* stream)
{
    return widen_u32_to_u64_dev(d_in, count, d_out, (hipStream_t)stream);
}

static int query_host(const sfx_index* ix, const uint8_t* qbytes, const uint64_t* qoff, uint64_t nq,
                      uint32_t* start_out, uint32_t* end_out, uint8_t* found_out, uint32_t* any_out)
{
    if (!ix || (nq && !qoff)) return SFX_ERR_ARG;
    if (nq == 0) return SFX_OK;
    uint64_t qtotal = qoff[nq];
    if (qtotal && !qbytes) return SFX_ERR_ARG;
    for (uint64_t k = 0; k < nq; k++) if (qoff[k + 1] < qoff[k]) return SFX_ERR_ARG;
    DevBuf dq, doff, ds, de, df, da;
    SFX_TRY(dq.alloc(qtotal));
    SFX_TRY(doff.alloc((nq + 1) * sizeof(uint64_t)));
    if (start_out) SFX_TRY(ds.alloc(nq * 4));
    if (end_out) SFX_TRY(de.alloc(nq * 4));
    if (found_out) SFX_TRY(df.alloc(nq));
    if (any_out) SFX_TRY(da.alloc(nq * 4));
    hipStream_t st = call_stream();
    StreamDrain drain{st};            // (declared after the buffers: runs before they return to the pool)
    if (qtotal) SFX_HIP(hipMemcpyAsync(dq.p, qbytes, qtotal, hipMemcpyHostToDevice, st));
    SFX_HIP(hipMemcpyAsync(doff.p, qoff, (nq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    SFX_TRY(sfx_index_query_dev(ix, (const uint8_t*)dq.p, (const uint64_t*)doff.p, nq, (uint32_t*)ds.p, (uint32_t*)de.p,
                                (uint8_t*)df.p, (uint32_t*)da.p, st));
    if (start_out) SFX_HIP(hipMemcpyAsync(start_out, ds.p, nq * 4, hipMemcpyDeviceToHost, st));
    if (end_out) SFX_HIP(hipMemcpyAsync(end_out, de.p, nq * 4, hipMemcpyDeviceToHost, st));
    if (found_out) SFX_HIP(hipMemcpyAsync(found_out, df.p, nq, hipMemcpyDeviceToHost, st));
    if (any_out) SFX_HIP(hipMemcpyAsync(any_out, da.p, nq * 4, hipMemcpyDeviceToHost, st));
    SFX_HIP(hipStreamSynchronize(st));
    return SFX_OK;
}

int sfx_positions_batch(const sfx_index* ix, const uint8_t* qbytes, const uint64_t* qoff,
                        uint64_t nq, uint32_t* start_out, uint32_t* end_out)
{
    return query_host(ix, qbytes, qoff, nq, start_out, end_out, nullptr, nullptr);
}

int sfx_contains_batch(const sfx_index* ix, const uint8_t* qbytes, const uint64_t* qoff,
                       uint64_t nq, uint8_t* found_out, uint32_t* any_out)
{
    return query_host(ix, qbytes, qoff, nq, nullptr, nullptr, found_out, any_out);
}

// ---- suffix-tree topology, generalized suffix array -------------------------------------------
uint64_t sfx_lcp_intervals_workspace_bytes(uint64_t n) { return lcp_intervals_workspace_bytes(n); }
int sfx_lcp_intervals_dev(const uint32_t* d_lcp, uint64_t n, uint32_t* d_lb, uint32_t* d_rb, uint32_t* d_node,
                          uint32_t* d_parent, uint32_t* d_leaf_parent, void* d_workspace, uint64_t workspace_bytes,
                          void* stream)
{
    return lcp_intervals_dev(d_lcp, n, d_lb, d_rb, d_node, d_parent, d_leaf_parent, d_workspace, workspace_bytes,
                             (hipStream_t)stream);
}
int sfx_doc_lookup_dev(const uint32_t* d_positions, uint64_t count, const uint64_t* d_doc_starts, uint64_t ndocs,
                       uint32_t* d_doc, uint32_t* d_offset, void* stream)
{
    return doc_lookup_dev(d_positions, count, d_doc_starts, ndocs, d_doc, d_offset, (hipStream_t)stream);
}

// ---- partitioned build ---------------------------------------------------------------------
int sfx_byte_histogram_dev(const uint8_t* d_text, uint64_t shard_begin, uint64_t shard_end,
                           uint64_t* d_bins256, void* stream)
{
    return byte_histogram_dev(d_text, shard_begin, shard_end, d_bins256, (hipStream_t)stream);
}
int sfx_key_histogram_dev(const uint8_t* d_text, uint64_t n, uint64_t shard_begin,
                          uint64_t shard_end, const uint64_t* d_global_byte_bins256, int top_bits,
                          uint64_t* d_bins, void* stream)
{
    return key_histogram_dev(d_text, n, shard_begin, shard_end, d_global_byte_bins256, top_bits,
                             d_bins, (hipStream_t)stream);
}
uint64_t sfx_sa_range_workspace_bytes(uint64_t n, uint64_t capacity) { return sa_range_workspace_bytes(n, capacity); }
int sfx_build_sa_range_u32_dev(const uint8_t* d_text, uint64_t n,
                               const uint64_t* d_global_byte_bins256, int top_bits, uint32_t bin_lo,
                               uint32_t bin_hi, uint64_t capacity, uint32_t* d_sa_part,
                               uint64_t* count_out, void* d_workspace, uint64_t workspace_bytes,
                               void* stream)
{
    return build_sa_range_u32_dev(d_text, n, d_global_byte_bins256, top_bits, bin_lo, bin_hi, capacity,
                                  d_sa_part, count_out, d_workspace, workspace_bytes,
                                  (hipStream_t)stream);
}

int sfx_pack_text_dev(const uint8_t* d_text, uint64_t count, const uint64_t* d_global_byte_bins256,
                      uint8_t* d_scratch256, uint32_t* d_words, uint64_t n_words, void* stream)
{
    return pack_text_dev(d_text, count, d_global_byte_bins256, d_scratch256, d_words, n_words, (hipStream_t)stream);
}
int sfx_build_sa_range_packed_u32_dev(const uint32_t* d_packed, uint64_t n,
                                      const uint64_t* d_global_byte_bins256, int top_bits, uint32_t bin_lo,
                                      uint32_t bin_hi, uint64_t capacity, uint32_t* d_sa_part,
                                      uint64_t* count_out, void* d_workspace, uint64_t workspace_bytes,
                                      void* stream)
{
    if (!d_packed) return SFX_ERR_ARG;
    return build_sa_range_u32_dev(nullptr, n, d_global_byte_bins256, top_bits, bin_lo, bin_hi, capacity, d_sa_part,
                                  count_out, d_workspace, workspace_bytes, (hipStream_t)stream, d_packed);
}

void sfx_release_cached_buffers(void)
{
    std::vector<PooledBuf> take;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        take.swap(g_pool);
    }
    for (PooledBuf& b : take) (void)hipFree(b.p);
}

// ---- profiling --------------------------------------------------------------------------------
void sfx_profile_enable(int on) { g_profile = on != 0; }
void sfx_profile_reset(void)
{
    profile_fold();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_stats.clear();
}
int sfx_profile_report(sfx_kernel_stat* out, int cap)
{
    profile_fold();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = (int)g_prof_stats.size();
    for (int i = 0; i < n && i < cap && out; i++) {
        memset(&out[i], 0, sizeof(out[i]));
        snprintf(out[i].name, sizeof(out[i].name), "%s", g_prof_stats[i].name.c_str());
        out[i].launches = g_prof_stats[i].launches;
        out[i].total_ms = g_prof_stats[i].ms;
        out[i].algo_bytes = g_prof_stats[i].bytes;
    }
    return n;
}
void sfx_last_build_stats(sfx_build_stats* out)
{
    if (out) *out = tls_build_stats();
}
int sfx_set_option(int option, uint64_t value)
{
    if (option == SFX_OPT_TINY_MAX && value <= tiny_max_default()) { tiny_set_limit(value); return SFX_OK; }
    return SFX_ERR_ARG;
}
uint64_t sfx_get_option(int option)
{
    return option == SFX_OPT_TINY_MAX ? tiny_limit() : 0;
}
uint64_t sfx_build_stats_read(void* out, uint64_t out_bytes)
{
    if (out && out_bytes) {
        const sfx_build_stats& s = tls_build_stats();
        memcpy(out, &s, (size_t)(out_bytes < sizeof(s) ? out_bytes : sizeof(s)));
    }
    return (uint64_t)sizeof(sfx_build_stats);
}

}  // extern "C"
