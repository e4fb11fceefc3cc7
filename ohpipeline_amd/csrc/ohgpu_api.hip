// ohgpu_api.hip -- the C ABI of include/ohgpu.h: validation, descriptor upload, launches.
// No exception crosses this boundary; every failure is a negative code plus ohgpu_last_error().
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "ohgpu_internal.h"
#include "src_mfma_common.h"

namespace ohgpu {

static thread_local char g_err[512] = "";
static int g_plan_threads = 0;
int plan_thread_cap() { return g_plan_threads; }

// ---- the planning pool: up to 15 helper threads, started on first use, one job at a time (callers queue on a mutex) ----
// The pool is a heap object that is never destroyed: its helpers are detached and sleep in `wake` between jobs, and a mutex or a
// condition variable must not be destroyed under a sleeper (glibc's pthread_cond_destroy waits for them) -- least of all in a forked
// child's exit(), whose copy of the object still counts the parent's sleepers but has none of its threads.  A child gets a pool of
// its own (pthread_atfork): the parent's is left where it lies.
namespace {
struct PlanPool {
    // A job is n_ranges independent ranges; the caller and whichever helpers are awake CLAIM them one at a time (an atomic counter), so
    // a helper that wakes late -- sixteen sleepers behind one mutex do not all start at once -- costs the job nothing but its share:
    // with a fixed range per thread the pass lasted as long as its last waker (0.63-0.99 ms for the same 0.5 ms of work, round 5).
    std::mutex one_job;                       // a job owns the pool from start to finish
    std::mutex m;
    std::condition_variable wake, done;
    unsigned n_helpers = 0;
    void (*job)(void*, unsigned) = nullptr;
    void* arg = nullptr;
    unsigned n_ranges = 0;
    bool job_open = false;                    // (under m) the job and its argument are alive: a helper may join
    unsigned active = 0;                      // (under m) helpers that have joined and not yet left
    uint64_t generation = 0;
    std::atomic<unsigned> next{0}, finished{0};
    void helper()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return generation != seen; });
            seen = generation;
            if (!job_open) continue;          // (woke after the job it was woken for had finished)
            void (*j)(void*, unsigned) = job;
            void* a = arg;
            const unsigned nr = n_ranges;
            active++;
            lk.unlock();
            for (unsigned t; (t = next.fetch_add(1, std::memory_order_relaxed)) < nr;) { j(a, t); finished.fetch_add(1, std::memory_order_release); }
            lk.lock();
            if (--active == 0) done.notify_one();
        }
    }
    void run(unsigned nr, unsigned threads, void (*j)(void*, unsigned), void* a)
    {
        std::lock_guard<std::mutex> hold(one_job);
        {
            std::unique_lock<std::mutex> lk(m);
            while (n_helpers + 1 < threads) { ++n_helpers; std::thread([this] { helper(); }).detach(); }
            job = j; arg = a; n_ranges = nr;
            next.store(0, std::memory_order_relaxed); finished.store(0, std::memory_order_relaxed);
            job_open = true;
            generation++;
        }
        // (as many sleepers as the job may use: a helper beyond `threads` started by an earlier, wider job stays asleep)
        if (threads - 1 >= n_helpers) wake.notify_all();
        else for (unsigned k = 1; k < threads; k++) wake.notify_one();
        for (unsigned t; (t = next.fetch_add(1, std::memory_order_relaxed)) < nr;) { j(a, t); finished.fetch_add(1, std::memory_order_release); }
        std::unique_lock<std::mutex> lk(m);
        job_open = false;                     // (no range is left to claim: whoever has not joined yet need not)
        done.wait(lk, [&] { return active == 0; });
        // (every claimed range was run by the caller or by a helper that has left: finished == nr)
    }
};
std::atomic<PlanPool*> g_pool{nullptr};
std::once_flag g_pool_once;
PlanPool& pool()
{
    std::call_once(g_pool_once, [] {
        g_pool.store(new PlanPool());
        // (a fork taken while a job runs leaves the child a locked copy: it is abandoned with the rest)
        (void)pthread_atfork(nullptr, nullptr, [] { g_pool.store(new PlanPool()); });
    });
    return *g_pool.load();
}
}  // namespace

void run_on_pool(unsigned n_ranges, unsigned threads, void (*job)(void* arg, unsigned t), void* arg)
{
    if (threads > n_ranges) threads = n_ranges;
    if (threads > 16) threads = 16;
    if (threads <= 1) { for (unsigned t = 0; t < n_ranges; t++) job(arg, t); return; }
    pool().run(n_ranges, threads, job, arg);
}

// The CPUs this process may keep busy: the affinity mask's, capped by the container's CPU quota (cgroup v2 cpu.max; v1
// cpu.cfs_quota_us / cpu.cfs_period_us) -- a box of 256 logical CPUs whose container is granted sixteen runs sixteen planner threads'
// worth of work however many are started.  Callers that share the grant (one rank per GPU on one host) divide it themselves:
// ohgpu_set_plan_threads.
unsigned usable_cpus()
{
    static const unsigned cached = [] {
        unsigned n = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
        double quota = 0.0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[64] = "";
            long long period = 0;
            if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
            fclose(f);
        } else {
            long long q = -1, per = 0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (q > 0 && per > 0) quota = (double)q / (double)per;
        }
        if (quota >= 1.0 && (unsigned)(quota + 0.5) < n) n = (unsigned)(quota + 0.5);
        return n ? n : 1u;
    }();
    return cached;
}

int set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static bool valid_bits(uint32_t bits) { return bits == 8 || bits == 16 || bits == 24 || bits == 32; }
static bool valid_endian(uint32_t e) { return e == OHGPU_ENDIAN_LITTLE || e == OHGPU_ENDIAN_BIG; }

static hipStream_t pick_stream(const ohgpu_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

}  // namespace ohgpu

using namespace ohgpu;

extern "C" {

int ohgpu_abi_version(void) { return OHGPU_ABI_VERSION; }

const char* ohgpu_last_error(void) { return g_err; }

int ohgpu_device_count(void)
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) return 0;
    if (e != hipSuccess) return set_error(OHGPU_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int ohgpu_init(int device, ohgpu_ctx** out)
{
    if (!out) return set_error(OHGPU_ERR_INVALID, "ohgpu_init: null out pointer");
    *out = nullptr;
    const int n = ohgpu_device_count();
    if (n < 0) return n;
    if (n == 0) return set_error(OHGPU_ERR_NO_DEVICE, "ohgpu_init: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return set_error(OHGPU_ERR_INVALID, "ohgpu_init: device %d out of range [0,%d)", device, n);
    OHGPU_HIP_TRY(hipSetDevice(device));
    ohgpu_ctx* ctx = new (std::nothrow) ohgpu_ctx();
    if (!ctx) return set_error(OHGPU_ERR_NOMEM, "ohgpu_init: out of host memory");
    ctx->device = device;
    ctx->variant = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) {
        snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
        ctx->num_cus = prop.multiProcessorCount;
    } else {
        snprintf(ctx->name, sizeof(ctx->name), "unknown");
        ctx->num_cus = 256;
    }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return set_error(OHGPU_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
    uint16_t table[512];
    build_ramp_table(table);
    e = hipMalloc((void**)&ctx->d_ramp_table, sizeof(table));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_ramp_table, table, sizeof(table), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return set_error(OHGPU_ERR_DEVICE, "ramp table upload: %s", hipGetErrorString(e));
    }
    // (the planner's own device pass -- the ramp planes' kernel -- is loaded here, not inside the first batch's creation)
    (void)load_ramp_plane_kernel();
    *out = ctx;
    return OHGPU_OK;
}

int ohgpu_shutdown(ohgpu_ctx* ctx)
{
    if (!ctx) return OHGPU_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(ctx->d_ramp_table);
    for (auto& list : ctx->cache.idle) for (void* p : list) (void)hipFree(p);
    if (ctx->stage.d_src) (void)hipFree(ctx->stage.d_src);
    if (ctx->stage.d_dst) (void)hipFree(ctx->stage.d_dst);
    if (ctx->stage.h_bounce) (void)hipHostFree(ctx->stage.h_bounce);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return OHGPU_OK;
}

int ohgpu_device_name(ohgpu_ctx* ctx, char* buf, size_t buf_bytes)
{
    if (!ctx || !buf || buf_bytes == 0) return set_error(OHGPU_ERR_INVALID, "ohgpu_device_name: bad argument");
    snprintf(buf, buf_bytes, "%s", ctx->name);
    return OHGPU_OK;
}

int ohgpu_device_pci_bus_id(ohgpu_ctx* ctx, char* buf, size_t buf_bytes)
{
    if (!ctx || !buf || buf_bytes < 16) return set_error(OHGPU_ERR_INVALID, "ohgpu_device_pci_bus_id: bad argument");
    OHGPU_HIP_TRY(hipDeviceGetPCIBusId(buf, (int)buf_bytes, ctx->device));
    for (char* c = buf; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');     // (sysfs spells it in lower case)
    return OHGPU_OK;
}

int ohgpu_set_kernel_variant(ohgpu_ctx* ctx, int variant)
{
    if (!ctx || variant < 0 || variant > 5) return set_error(OHGPU_ERR_INVALID, "ohgpu_set_kernel_variant: bad argument");
    ctx->variant = variant;
    return OHGPU_OK;
}

/* ---------------------------------------------------------------- plumbing */
#define CTX_GUARD(name)                                                                    \
    if (!ctx) return set_error(OHGPU_ERR_INVALID, name ": null context");                 \
    OHGPU_HIP_TRY(hipSetDevice(ctx->device))

int ohgpu_malloc(ohgpu_ctx* ctx, size_t bytes, void** dptr)
{
    CTX_GUARD("ohgpu_malloc");
    if (!dptr) return set_error(OHGPU_ERR_INVALID, "ohgpu_malloc: null out pointer");
    *dptr = nullptr;
    const hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) return set_error(OHGPU_ERR_NOMEM, "hipMalloc(%zu): out of device memory", bytes);
    OHGPU_HIP_TRY(e);
    return OHGPU_OK;
}

int ohgpu_free(ohgpu_ctx* ctx, void* dptr)
{
    CTX_GUARD("ohgpu_free");
    if (dptr) OHGPU_HIP_TRY(hipFree(dptr));
    return OHGPU_OK;
}

int ohgpu_malloc_host(ohgpu_ctx* ctx, size_t bytes, void** hptr)
{
    CTX_GUARD("ohgpu_malloc_host");
    if (!hptr) return set_error(OHGPU_ERR_INVALID, "ohgpu_malloc_host: null out pointer");
    OHGPU_HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return OHGPU_OK;
}

int ohgpu_free_host(ohgpu_ctx* ctx, void* hptr)
{
    CTX_GUARD("ohgpu_free_host");
    if (hptr) OHGPU_HIP_TRY(hipHostFree(hptr));
    return OHGPU_OK;
}

int ohgpu_memcpy_h2d(ohgpu_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream)
{
    CTX_GUARD("ohgpu_memcpy_h2d");
    if (bytes) OHGPU_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, pick_stream(ctx, stream)));
    return OHGPU_OK;
}

int ohgpu_memcpy_d2h(ohgpu_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
    CTX_GUARD("ohgpu_memcpy_d2h");
    if (bytes) OHGPU_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, pick_stream(ctx, stream)));
    return OHGPU_OK;
}

int ohgpu_memset(ohgpu_ctx* ctx, void* dptr, int value, size_t bytes, void* stream)
{
    CTX_GUARD("ohgpu_memset");
    if (bytes) OHGPU_HIP_TRY(hipMemsetAsync(dptr, value, bytes, pick_stream(ctx, stream)));
    return OHGPU_OK;
}

int ohgpu_stream_create(ohgpu_ctx* ctx, void** stream)
{
    CTX_GUARD("ohgpu_stream_create");
    if (!stream) return set_error(OHGPU_ERR_INVALID, "ohgpu_stream_create: null out pointer");
    hipStream_t s;
    OHGPU_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return OHGPU_OK;
}

int ohgpu_stream_destroy(ohgpu_ctx* ctx, void* stream)
{
    CTX_GUARD("ohgpu_stream_destroy");
    if (stream) OHGPU_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return OHGPU_OK;
}

int ohgpu_stream_sync(ohgpu_ctx* ctx, void* stream)
{
    CTX_GUARD("ohgpu_stream_sync");
    OHGPU_HIP_TRY(hipStreamSynchronize(pick_stream(ctx, stream)));
    return OHGPU_OK;
}

int ohgpu_event_create(ohgpu_ctx* ctx, void** event)
{
    CTX_GUARD("ohgpu_event_create");
    if (!event) return set_error(OHGPU_ERR_INVALID, "ohgpu_event_create: null out pointer");
    hipEvent_t ev;
    OHGPU_HIP_TRY(hipEventCreate(&ev));
    *event = (void*)ev;
    return OHGPU_OK;
}

int ohgpu_event_destroy(ohgpu_ctx* ctx, void* event)
{
    CTX_GUARD("ohgpu_event_destroy");
    if (event) OHGPU_HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return OHGPU_OK;
}

int ohgpu_event_record(ohgpu_ctx* ctx, void* event, void* stream)
{
    CTX_GUARD("ohgpu_event_record");
    OHGPU_HIP_TRY(hipEventRecord((hipEvent_t)event, pick_stream(ctx, stream)));
    return OHGPU_OK;
}

int ohgpu_stream_wait_event(ohgpu_ctx* ctx, void* stream, void* event)
{
    CTX_GUARD("ohgpu_stream_wait_event");
    if (!event) return set_error(OHGPU_ERR_INVALID, "ohgpu_stream_wait_event: null event");
    OHGPU_HIP_TRY(hipStreamWaitEvent(pick_stream(ctx, stream), (hipEvent_t)event, 0));
    return OHGPU_OK;
}

int ohgpu_event_elapsed_ms(ohgpu_ctx* ctx, void* start, void* stop, float* ms)
{
    CTX_GUARD("ohgpu_event_elapsed_ms");
    if (!ms) return set_error(OHGPU_ERR_INVALID, "ohgpu_event_elapsed_ms: null out pointer");
    OHGPU_HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    OHGPU_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return OHGPU_OK;
}

int ohgpu_ramp_table(uint16_t out[512])
{
    if (!out) return set_error(OHGPU_ERR_INVALID, "ohgpu_ramp_table: null out pointer");
    build_ramp_table(out);
    return OHGPU_OK;
}

/* ---------------------------------------------------------------- pcm batches */
static int validate_msg(const ohgpu_msg_desc& d, size_t i, uint64_t src_arena, uint64_t dst_arena)
{
    if (d.channels < 1 || d.channels > OHGPU_MAX_CHANNELS)
        return set_error(OHGPU_ERR_INVALID, "desc %zu: channels %u outside 1..8", i, d.channels);
    if (!valid_bits(d.src_bits) || !valid_bits(d.dst_bits))
        return set_error(OHGPU_ERR_INVALID, "desc %zu: bit depth %u -> %u (must be 8/16/24/32)", i, d.src_bits, d.dst_bits);
    if (!valid_endian(d.src_endian) || !valid_endian(d.dst_endian))
        return set_error(OHGPU_ERR_INVALID, "desc %zu: endian %u -> %u", i, d.src_endian, d.dst_endian);
    if (d.flags & ~(OHGPU_FLAG_RAMP | OHGPU_FLAG_SILENCE | OHGPU_FLAG_ZERO_LSB32))
        return set_error(OHGPU_ERR_INVALID, "desc %zu: unknown flag bits 0x%x", i, d.flags);
    if (d.ramp_start > OHGPU_RAMP_MAX || d.ramp_end > OHGPU_RAMP_MAX)
        return set_error(OHGPU_ERR_INVALID, "desc %zu: ramp [%u..%u] beyond Ramp::kMax", i, d.ramp_start, d.ramp_end);
    if ((d.flags & OHGPU_FLAG_RAMP) && d.n_frames > 131071u)     // i*iTotalRamp is TInt arithmetic (Msg.cpp:835)
        return set_error(OHGPU_ERR_INVALID, "desc %zu: ramped message of %u frames overflows the reference's TInt ramp product", i, d.n_frames);
    if (d.attenuation != OHGPU_UNITY_ATTENUATION && d.src_bits != 16)   // ASSERT(iBitDepth == 16), Msg.cpp:2741
        return set_error(OHGPU_ERR_UNSUPPORTED, "desc %zu: attenuation %u on %u-bit audio (16-bit only)", i, d.attenuation, d.src_bits);
    const uint64_t src_bytes = (uint64_t)d.n_frames * d.channels * (d.src_bits / 8);
    const uint64_t dst_bytes = (uint64_t)d.n_frames * d.channels * (d.dst_bits / 8);
    if (!(d.flags & OHGPU_FLAG_SILENCE) && (d.src_offset > src_arena || src_bytes > src_arena - d.src_offset))
        return set_error(OHGPU_ERR_BOUNDS, "desc %zu: reads [%llu, +%llu) beyond the %llu-byte source arena", i,
                         (unsigned long long)d.src_offset, (unsigned long long)src_bytes, (unsigned long long)src_arena);
    if (d.dst_offset > dst_arena || dst_bytes > dst_arena - d.dst_offset)
        return set_error(OHGPU_ERR_BOUNDS, "desc %zu: writes [%llu, +%llu) beyond the %llu-byte destination arena", i,
                         (unsigned long long)d.dst_offset, (unsigned long long)dst_bytes, (unsigned long long)dst_arena);
    return OHGPU_OK;
}

}  // extern "C"

namespace ohgpu {

hipError_t ctx_dev_alloc(ohgpu_ctx* ctx, void** p, size_t bytes)
{
    *p = nullptr;
    int c = 0;
    while (c < DevCache::kClasses && ((size_t)256 << c) < bytes) c++;
    DevCache& k = ctx->cache;
    std::lock_guard<std::mutex> hold(k.m);
    // (an idle block of the request's class, or of one of the two above it: a caller whose batches straddle a class boundary from one
    // period to the next -- a plan of 60 KB, then one of 70 -- is served by what the larger of them left behind)
    for (int q = c; q < DevCache::kClasses && q <= c + 2; q++) {
        if (k.idle[q].empty()) continue;
        *p = k.idle[q].back();
        k.idle[q].pop_back();
        return hipSuccess;
    }
    const hipError_t e = hipMalloc(p, c < DevCache::kClasses ? ((size_t)256 << c) : bytes);
    if (e != hipSuccess) { *p = nullptr; return e; }
    k.device_allocs++;
    k.cls[*p] = c < DevCache::kClasses ? c : -1;
    return hipSuccess;
}

void ctx_dev_free(ohgpu_ctx* ctx, void* p)
{
    if (!p) return;
    DevCache& k = ctx->cache;
    std::lock_guard<std::mutex> hold(k.m);
    const auto it = k.cls.find(p);
    if (it != k.cls.end() && it->second >= 0 && k.idle[it->second].size() < 64) {      // (at most 64 idle blocks per class)
        k.idle[it->second].push_back(p);
        return;
    }
    if (it != k.cls.end()) k.cls.erase(it);
    (void)hipFree(p);
}

// one of HostStage's buffers, at least `bytes` long: kept while it is, replaced by one half as large again when it is not
static int stage_reserve(ohgpu_ctx* ctx, void** p, size_t* cap, size_t bytes, bool pinned_host)
{
    if (*cap >= bytes && *p) return OHGPU_OK;
    if (*p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)(pinned_host ? hipHostFree(*p) : hipFree(*p));
        *p = nullptr; *cap = 0;
    }
    size_t want = bytes + bytes / 2;
    if (want < (64u << 10)) want = 64u << 10;
    want = (want + 4095) & ~(size_t)4095;
    const hipError_t e = pinned_host ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want);
    if (e != hipSuccess) {
        *p = nullptr;
        return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "host-buffer staging (%zu bytes): %s", want, hipGetErrorString(e));
    }
    *cap = want;
    if (!pinned_host) { std::lock_guard<std::mutex> hold(ctx->cache.m); ctx->cache.device_allocs++; }
    return OHGPU_OK;
}

int host_roundtrip(ohgpu_ctx* ctx, const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes,
                   std::vector<std::pair<uint64_t, uint64_t>>& ranges, const std::function<int(const void*, void*)>& run)
{
    HostStage& st = ctx->stage;
    st.calls++;
    int err = stage_reserve(ctx, &st.d_src, &st.src_cap, src_bytes ? src_bytes : 1, false);
    if (err == OHGPU_OK) err = stage_reserve(ctx, &st.d_dst, &st.dst_cap, dst_bytes ? dst_bytes : 1, false);
    if (err != OHGPU_OK) return err;
    hipStream_t s = ctx->stream;
    if (src_bytes) {
        OHGPU_HIP_TRY(hipMemcpyAsync(st.d_src, src_host, src_bytes, hipMemcpyHostToDevice, s));
        st.h2d_bytes += src_bytes;
    }
    // the covered runs of the destination, merged where they touch or overlap
    std::sort(ranges.begin(), ranges.end());
    std::vector<std::pair<uint64_t, uint64_t>> runs;                  // [lo, hi)
    for (const auto& r : ranges) {
        if (r.second == 0) continue;
        if (!runs.empty() && r.first <= runs.back().second) runs.back().second = std::max(runs.back().second, r.first + r.second);
        else runs.emplace_back(r.first, r.first + r.second);
    }
    err = run(st.d_src, st.d_dst);
    if (err != OHGPU_OK) { (void)hipStreamSynchronize(s); return err; }
    if (runs.empty()) { OHGPU_HIP_TRY(hipStreamSynchronize(s)); return OHGPU_OK; }
    const uint64_t lo = runs.front().first, hi = runs.back().second;
    if (runs.size() == 1) {                                           // the outputs tile [lo, hi): one copy, straight home
        OHGPU_HIP_TRY(hipMemcpyAsync((uint8_t*)dst_host + lo, (const uint8_t*)st.d_dst + lo, hi - lo, hipMemcpyDeviceToHost, s));
        st.d2h_bytes += hi - lo;
        OHGPU_HIP_TRY(hipStreamSynchronize(s));
        return OHGPU_OK;
    }
    // holes between the outputs: the span comes back to the bounce buffer in one copy, the covered runs go home from there
    err = stage_reserve(ctx, &st.h_bounce, &st.bounce_cap, hi - lo, true);
    if (err != OHGPU_OK) { (void)hipStreamSynchronize(s); return err; }
    OHGPU_HIP_TRY(hipMemcpyAsync(st.h_bounce, (const uint8_t*)st.d_dst + lo, hi - lo, hipMemcpyDeviceToHost, s));
    st.d2h_bytes += hi - lo;
    OHGPU_HIP_TRY(hipStreamSynchronize(s));
    for (const auto& r : runs) memcpy((uint8_t*)dst_host + r.first, (const uint8_t*)st.h_bounce + (r.first - lo), r.second - r.first);
    return OHGPU_OK;
}

}  // namespace ohgpu

extern "C" {

int ohgpu_device_allocations(ohgpu_ctx* ctx, uint64_t* count)
{
    CTX_GUARD("ohgpu_device_allocations");
    if (!count) return set_error(OHGPU_ERR_INVALID, "ohgpu_device_allocations: null result");
    std::lock_guard<std::mutex> hold(ctx->cache.m);
    *count = ctx->cache.device_allocs;
    return OHGPU_OK;
}

static int upload_batch(ohgpu_ctx* ctx, ohgpu_batch* b, const void* host_descs, size_t bytes)
{
    if (bytes == 0) return OHGPU_OK;
    hipError_t e = ctx_dev_alloc(ctx, &b->d_descs, bytes);
    if (e == hipErrorOutOfMemory) return set_error(OHGPU_ERR_NOMEM, "descriptor upload: out of device memory");
    OHGPU_HIP_TRY(e);
    e = hipMemcpy(b->d_descs, host_descs, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ctx_dev_free(ctx, b->d_descs);
        b->d_descs = nullptr;
        return set_error(OHGPU_ERR_DEVICE, "descriptor upload: %s", hipGetErrorString(e));
    }
    return OHGPU_OK;
}

int ohgpu_pcm_batch_create(ohgpu_ctx* ctx, const ohgpu_msg_desc* descs, size_t n,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** out)
{
    return pcm_batch_create_prefixed(ctx, descs, n, src_arena_bytes, dst_arena_bytes, nullptr, nullptr, 0, out);
}

}  // extern "C"

int ohgpu::pcm_batch_create_prefixed(ohgpu_ctx* ctx, const ohgpu_msg_desc* descs, size_t n, uint64_t src_arena_bytes, uint64_t dst_arena_bytes,
                                     const MsgPrefix* prefixes, const uint8_t* blob, size_t blob_bytes, ohgpu_batch** out)
{
    CTX_GUARD("ohgpu_pcm_batch_create");
    if (!out || (n && !descs)) return set_error(OHGPU_ERR_INVALID, "ohgpu_pcm_batch_create: null argument");
    *out = nullptr;
    if (n > 0xffffffffull) return set_error(OHGPU_ERR_INVALID, "ohgpu_pcm_batch_create: too many descriptors");
    ohgpu_batch* b = new (std::nothrow) ohgpu_batch();
    if (!b) return set_error(OHGPU_ERR_NOMEM, "ohgpu_pcm_batch_create: out of host memory");
    b->kind = kBatchPcm;
    b->n = n;
    b->src_arena_bytes = src_arena_bytes;
    b->dst_arena_bytes = dst_arena_bytes;
    b->uniform = true;
    for (size_t i = 0; i < n; i++) {
        const int err = validate_msg(descs[i], i, src_arena_bytes, dst_arena_bytes);
        if (err != OHGPU_OK) { delete b; return err; }
        const ohgpu_msg_desc& d = descs[i];
        b->in_frames += d.n_frames;
        b->out_frames += d.n_frames;
        if (!(d.flags & OHGPU_FLAG_SILENCE)) b->src_bytes_touched += (uint64_t)d.n_frames * d.channels * (d.src_bits / 8);
        b->dst_bytes_written += (uint64_t)d.n_frames * d.channels * (d.dst_bits / 8);
        if (d.n_frames > b->max_frames) b->max_frames = d.n_frames;
        if (i == 0) {
            b->channels = d.channels; b->src_bits = d.src_bits; b->src_endian = d.src_endian;
            b->dst_bits = d.dst_bits; b->dst_endian = d.dst_endian;
        } else if (d.channels != b->channels || d.src_bits != b->src_bits || d.src_endian != b->src_endian ||
                   d.dst_bits != b->dst_bits || d.dst_endian != b->dst_endian) {
            b->uniform = false;
        }
    }
    int err = upload_batch(ctx, b, descs, n * sizeof(ohgpu_msg_desc));
    if (err == OHGPU_OK) err = plan_pcm_line(ctx, b, descs, n, prefixes, blob, blob_bytes);
    if (err != OHGPU_OK) { if (b->d_descs) ctx_dev_free(ctx, b->d_descs); delete b; return err; }
    *out = b;
    return OHGPU_OK;
}

extern "C" {

int ohgpu_pcm_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream)
{
    CTX_GUARD("ohgpu_pcm_batch_run");
    if (!batch || batch->kind != kBatchPcm) return set_error(OHGPU_ERR_INVALID, "ohgpu_pcm_batch_run: not a pcm batch");
    if (batch->n == 0) return OHGPU_OK;
    if (!dst_base || (!src_base && batch->src_bytes_touched)) return set_error(OHGPU_ERR_INVALID, "ohgpu_pcm_batch_run: null arena pointer");
    if (ctx->variant != 1 && batch->line.enabled)
        OHGPU_HIP_TRY(launch_pcm_line(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, pick_stream(ctx, stream)));
    else
        OHGPU_HIP_TRY(launch_pcm_v1(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, pick_stream(ctx, stream)));
    return OHGPU_OK;
}

// Batches with per-launch device state (see ohgpu_batch::last_done): refuse a launch on another stream while the previous one
// has not finished; remember this one.
static int claim_single_launch(const ohgpu_batch* b, hipStream_t s, const char* who)
{
    if (batch_busy_on_another_stream(b, s))
        return set_error(OHGPU_ERR_INVALID, "%s: the batch is still running on another stream (its unit counters / workspace serve one "
                         "launch at a time: wait for it, use the same stream, or create a second batch)", who);
    if (b->last_done == nullptr && hipEventCreate(&b->last_done) != hipSuccess) {       // (it rides on a dispatch as its stop event: src_batch_run)
        b->last_done = nullptr;
        return set_error(OHGPU_ERR_DEVICE, "%s: hipEventCreate failed", who);
    }
    b->last_stream = s;
    return OHGPU_OK;
}
static void launched(const ohgpu_batch* b, hipStream_t s) { b->last_untracked = false; if (b->last_done) (void)hipEventRecord(b->last_done, s); }

int ohgpu_batch_destroy(ohgpu_ctx* ctx, ohgpu_batch* batch)
{
    CTX_GUARD("ohgpu_batch_destroy");
    if (!batch) return OHGPU_OK;
    for (ohgpu_batch* part : batch->parts) ohgpu_batch_destroy(ctx, part);
    // (its blocks go back to the context's cache, for the next batch to write into: nothing of this one may still be running --
    // what hipFree used to see to by itself)
    if (batch->d_descs || batch->kind == kBatchPcm || batch->kind == kBatchFlywheel || batch->kind == kBatchFmt) (void)hipDeviceSynchronize();
    if (batch->kind == kBatchSrc) free_src_fast(ctx, batch);          // (waits for the batch's last launch: before its event goes)
    if (batch->last_done) hipEventDestroy(batch->last_done);
    if (batch->d_descs) ctx_dev_free(ctx, batch->d_descs);
    if (batch->kind == kBatchPcm) free_pcm_line(ctx, batch);
    if (batch->kind == kBatchFlywheel) free_flywheel(ctx, batch);
    if (batch->kind == kBatchFmt) { free_fmt_line(ctx, batch); free_pcm_line(ctx, batch); }
    if (batch->kind == kBatchOhm) free_ohm(ctx, batch);
    delete batch;
    return OHGPU_OK;
}

int ohgpu_batch_info(const ohgpu_batch* b, uint64_t* n_msgs, uint64_t* in_frames, uint64_t* out_frames,
                     uint64_t* src_bytes_touched, uint64_t* dst_bytes_written)
{
    if (!b) return set_error(OHGPU_ERR_INVALID, "ohgpu_batch_info: null batch");
    if (n_msgs) *n_msgs = b->n;
    if (in_frames) *in_frames = b->in_frames;
    if (out_frames) *out_frames = b->out_frames;
    if (src_bytes_touched) *src_bytes_touched = b->src_bytes_touched;
    if (dst_bytes_written) *dst_bytes_written = b->dst_bytes_written;
    return OHGPU_OK;
}

int ohgpu_pcm_process_host(ohgpu_ctx* ctx, const ohgpu_msg_desc* descs, size_t n,
                           const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes)
{
    CTX_GUARD("ohgpu_pcm_process_host");
    ohgpu_batch* b = nullptr;
    int err = ohgpu_pcm_batch_create(ctx, descs, n, src_bytes, dst_bytes, &b);
    if (err != OHGPU_OK) return err;
    std::vector<std::pair<uint64_t, uint64_t>> out(n);
    for (size_t i = 0; i < n; i++) out[i] = {descs[i].dst_offset, (uint64_t)descs[i].n_frames * descs[i].channels * (descs[i].dst_bits / 8)};
    err = host_roundtrip(ctx, src_host, src_bytes, dst_host, dst_bytes, out,
                         [&](const void* d_src, void* d_dst) { return ohgpu_pcm_batch_run(ctx, b, d_src, d_dst, nullptr); });
    ohgpu_batch_destroy(ctx, b);
    return err;
}

/* ---------------------------------------------------------------- layout-changing processors (a11, a13, a14) */
// end = off + a * b + c in 64 bits; false when any step wraps (a descriptor that wraps must fail validation as out of
// bounds: a wrapped end can look smaller than the arena)
static bool span_end(uint64_t off, uint64_t a, uint64_t b, uint64_t c, uint64_t* end)
{
    uint64_t p = 0, q = 0;
    if (__builtin_mul_overflow(a, b, &p) || __builtin_add_overflow(off, p, &q) || __builtin_add_overflow(q, c, end)) return false;
    return true;
}

int ohgpu_fmt_batch_create(ohgpu_ctx* ctx, const ohgpu_fmt_desc* descs, size_t n,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** out)
{
    CTX_GUARD("ohgpu_fmt_batch_create");
    if (!out || (n && !descs)) return set_error(OHGPU_ERR_INVALID, "ohgpu_fmt_batch_create: null argument");
    *out = nullptr;
    if (n > 0xffffffffull) return set_error(OHGPU_ERR_INVALID, "ohgpu_fmt_batch_create: too many descriptors");
    ohgpu_batch* b = new (std::nothrow) ohgpu_batch();
    if (!b) return set_error(OHGPU_ERR_NOMEM, "ohgpu_fmt_batch_create: out of host memory");
    b->kind = kBatchFmt;
    b->n = n;
    b->src_arena_bytes = src_arena_bytes;
    b->dst_arena_bytes = dst_arena_bytes;
    for (size_t i = 0; i < n; i++) {
        const ohgpu_fmt_desc& d = descs[i];
        int err = OHGPU_OK;
        const uint64_t ch = d.channels, nf = d.n_frames, sb = d.src_bits / 8;
        uint64_t src_lo = d.src_offset, src_hi = 0, dst_lo = d.dst_offset, dst_hi = 0;
        if (ch < 1 || ch > 10) err = set_error(OHGPU_ERR_INVALID, "fmt desc %zu: channels %u outside 1..10", i, d.channels);
        else if (d.kind == OHGPU_FMT_UNPACK_PLANAR || d.kind == OHGPU_FMT_SENDER_PACK) {
            if (!valid_bits(d.src_bits)) err = set_error(OHGPU_ERR_INVALID, "fmt desc %zu: source depth %u", i, d.src_bits);   // ASSERTS(), StarvationRamper.cpp:178-180
            else {
                if (!span_end(d.src_offset, nf, ch * sb, 0, &src_hi)) src_hi = UINT64_MAX;
                if (d.kind == OHGPU_FMT_UNPACK_PLANAR) {
                    if (ch > 1 && d.dst_plane_stride < nf * 4) err = set_error(OHGPU_ERR_INVALID, "fmt desc %zu: planes overlap (stride %llu < %llu)", i, (unsigned long long)d.dst_plane_stride, (unsigned long long)(nf * 4));
                    if (!span_end(d.dst_offset, ch - 1, d.dst_plane_stride, nf * 4, &dst_hi)) dst_hi = UINT64_MAX;
                } else {
                    if (!span_end(d.dst_offset, nf, (ch < 2 ? ch : 2) * (sb < 3 ? sb : 3), 0, &dst_hi)) dst_hi = UINT64_MAX;
                }
            }
        } else if (d.kind == OHGPU_FMT_FLAC_PACK) {
            if (!(d.dst_bits == 8 || d.dst_bits == 16 || d.dst_bits == 24))       // THROW(CodecStreamFeatureUnsupported), Flac.cpp:404-407
                err = set_error(OHGPU_ERR_UNSUPPORTED, "fmt desc %zu: FLAC bit depth %u (8/16/24 only)", i, d.dst_bits);
            else if (d.src_bits != 32) err = set_error(OHGPU_ERR_INVALID, "fmt desc %zu: FLAC planes are TInt32 (src_bits must be 32)", i);
            else if (d.src_offset % 4 != 0 || d.src_plane_stride % 4 != 0) err = set_error(OHGPU_ERR_INVALID, "fmt desc %zu: TInt32 planes must be 4-byte aligned", i);
            else {
                if (!span_end(d.src_offset, ch - 1, d.src_plane_stride, nf * 4, &src_hi)) src_hi = UINT64_MAX;
                if (!span_end(d.dst_offset, nf, ch * (d.dst_bits / 8), 0, &dst_hi)) dst_hi = UINT64_MAX;
            }
        } else {
            err = set_error(OHGPU_ERR_INVALID, "fmt desc %zu: unknown kind %u", i, d.kind);
        }
        if (err == OHGPU_OK && nf > 0 && (src_lo > src_arena_bytes || src_hi > src_arena_bytes || src_hi < src_lo))
            err = set_error(OHGPU_ERR_BOUNDS, "fmt desc %zu: reads up to %llu beyond the %llu-byte source arena", i, (unsigned long long)src_hi, (unsigned long long)src_arena_bytes);
        if (err == OHGPU_OK && nf > 0 && (dst_lo > dst_arena_bytes || dst_hi > dst_arena_bytes || dst_hi < dst_lo))
            err = set_error(OHGPU_ERR_BOUNDS, "fmt desc %zu: writes up to %llu beyond the %llu-byte destination arena", i, (unsigned long long)dst_hi, (unsigned long long)dst_arena_bytes);
        if (err != OHGPU_OK) { delete b; return err; }
        b->in_frames += nf;
        b->out_frames += nf;
        b->src_bytes_touched += nf ? src_hi - src_lo : 0;
        b->dst_bytes_written += nf ? dst_hi - dst_lo : 0;
    }
    int err = upload_batch(ctx, b, descs, n * sizeof(ohgpu_fmt_desc));
    if (err == OHGPU_OK) err = plan_fmt_line(ctx, b, descs, n);
    if (err != OHGPU_OK) { if (b->d_descs) ctx_dev_free(ctx, b->d_descs); delete b; return err; }
    *out = b;
    return OHGPU_OK;
}

int ohgpu_fmt_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream)
{
    CTX_GUARD("ohgpu_fmt_batch_run");
    if (!batch || batch->kind != kBatchFmt) return set_error(OHGPU_ERR_INVALID, "ohgpu_fmt_batch_run: not a fmt batch");
    if (batch->n == 0) return OHGPU_OK;
    if (!src_base || !dst_base) return set_error(OHGPU_ERR_INVALID, "ohgpu_fmt_batch_run: null arena pointer");
    if (ctx->variant != 1 && batch->line.enabled)                       // stereo Songcast packs planned onto the PCM line kernel
        OHGPU_HIP_TRY(launch_pcm_line(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, pick_stream(ctx, stream)));
    else if (ctx->variant != 1 && batch->fmtline.n_wide)                // Songcast packs of wider streams
        OHGPU_HIP_TRY(launch_ohm_wide(ctx, batch->fmtline.d_wide, batch->fmtline.n_wide, (const uint8_t*)src_base, (uint8_t*)dst_base, nullptr,
                                      pick_stream(ctx, stream)));
    else if (ctx->variant != 1 && batch->fmtline.enabled)
        OHGPU_HIP_TRY(launch_fmt_line(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, pick_stream(ctx, stream)));
    else
        OHGPU_HIP_TRY(launch_fmt_v1(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, pick_stream(ctx, stream)));
    return OHGPU_OK;
}

/* ---------------------------------------------------------------- FlywheelRamper (N1) */
int ohgpu_flywheel_batch_create(ohgpu_ctx* ctx, const ohgpu_flywheel_desc* descs, size_t n,
                                uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** out)
{
    CTX_GUARD("ohgpu_flywheel_batch_create");
    if (!out || (n && !descs)) return set_error(OHGPU_ERR_INVALID, "ohgpu_flywheel_batch_create: null argument");
    *out = nullptr;
    if (n > 0x0fffffffull) return set_error(OHGPU_ERR_INVALID, "ohgpu_flywheel_batch_create: too many descriptors");
    ohgpu_batch* b = new (std::nothrow) ohgpu_batch();
    if (!b) return set_error(OHGPU_ERR_NOMEM, "ohgpu_flywheel_batch_create: out of host memory");
    b->kind = kBatchFlywheel;
    b->n = n;
    b->src_arena_bytes = src_arena_bytes;
    b->dst_arena_bytes = dst_arena_bytes;
    for (size_t i = 0; i < n; i++) {
        const ohgpu_flywheel_desc& d = descs[i];
        int err = OHGPU_OK;
        const uint32_t dec = (d.sample_rate == 192000 || d.sample_rate == 176400) ? 4 : ((d.sample_rate == 88200 || d.sample_rate == 96000) ? 2 : 1);
        const uint64_t plane = d.channel_bytes, need = (uint64_t)d.in_samples * 4, out_bytes = (uint64_t)d.out_frames * d.channels * 4;
        if (d.channels < 1 || d.channels > 10) err = set_error(OHGPU_ERR_INVALID, "flywheel desc %zu: channels %u outside 1..10", i, d.channels);
        else if (d.sample_rate > 384000) err = set_error(OHGPU_ERR_INVALID, "flywheel desc %zu: sample rate %u above 384000", i, d.sample_rate);   // ASSERT, FlywheelRamper.cpp:178
        else if (need > plane) err = set_error(OHGPU_ERR_INVALID, "flywheel desc %zu: %llu-byte planes hold fewer than %u samples", i, (unsigned long long)plane, d.in_samples);   // ASSERT, :180
        else if (d.in_samples / dec < 4) err = set_error(OHGPU_ERR_INVALID, "flywheel desc %zu: %u training samples after decimation by %u (need 4)", i, d.in_samples, dec);
        else if (d.in_samples > 65536) err = set_error(OHGPU_ERR_INVALID, "flywheel desc %zu: %u training samples (limit 65536)", i, d.in_samples);
        else if (d.block_frames == 0 && d.out_frames != 0) err = set_error(OHGPU_ERR_INVALID, "flywheel desc %zu: block_frames is 0", i);
        else if (d.src_offset > src_arena_bytes || plane > src_arena_bytes || plane * d.channels > src_arena_bytes - d.src_offset)   // (plane <= arena: the product cannot wrap)
            err = set_error(OHGPU_ERR_BOUNDS, "flywheel desc %zu: training audio beyond the %llu-byte source arena", i, (unsigned long long)src_arena_bytes);
        else if (d.dst_offset > dst_arena_bytes || out_bytes > dst_arena_bytes - d.dst_offset)
            err = set_error(OHGPU_ERR_BOUNDS, "flywheel desc %zu: writes up to %llu beyond the %llu-byte destination arena", i,
                            (unsigned long long)(d.dst_offset + out_bytes), (unsigned long long)dst_arena_bytes);
        if (err != OHGPU_OK) { delete b; return err; }
        b->in_frames += d.in_samples;
        b->out_frames += d.out_frames;
        b->src_bytes_touched += need * d.channels;
        b->dst_bytes_written += out_bytes;
    }
    int err = upload_batch(ctx, b, descs, n * sizeof(ohgpu_flywheel_desc));
    if (err == OHGPU_OK) err = plan_flywheel(ctx, b, descs, n);
    if (err != OHGPU_OK) { if (b->d_descs) ctx_dev_free(ctx, b->d_descs); delete b; return err; }
    *out = b;
    return OHGPU_OK;
}

int ohgpu_flywheel_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream)
{
    CTX_GUARD("ohgpu_flywheel_batch_run");
    if (!batch || batch->kind != kBatchFlywheel) return set_error(OHGPU_ERR_INVALID, "ohgpu_flywheel_batch_run: not a flywheel batch");
    if (batch->n == 0) return OHGPU_OK;
    if (!src_base || !dst_base) return set_error(OHGPU_ERR_INVALID, "ohgpu_flywheel_batch_run: null arena pointer");
    hipStream_t s = pick_stream(ctx, stream);
    const int claim = claim_single_launch(batch, s, "ohgpu_flywheel_batch_run");       // (Burg's workspace is the batch's)
    if (claim != OHGPU_OK) return claim;
    OHGPU_HIP_TRY(launch_flywheel(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, s));
    launched(batch, s);
    return OHGPU_OK;
}

int ohgpu_flywheel_process_host(ohgpu_ctx* ctx, const ohgpu_flywheel_desc* descs, size_t n,
                                const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes)
{
    CTX_GUARD("ohgpu_flywheel_process_host");
    ohgpu_batch* b = nullptr;
    int err = ohgpu_flywheel_batch_create(ctx, descs, n, src_bytes, dst_bytes, &b);
    if (err != OHGPU_OK) return err;
    std::vector<std::pair<uint64_t, uint64_t>> out(n);
    for (size_t i = 0; i < n; i++) out[i] = {descs[i].dst_offset, (uint64_t)descs[i].out_frames * descs[i].channels * 4u};
    err = host_roundtrip(ctx, src_host, src_bytes, dst_host, dst_bytes, out,
                         [&](const void* d_src, void* d_dst) { return ohgpu_flywheel_batch_run(ctx, b, d_src, d_dst, nullptr); });
    ohgpu_batch_destroy(ctx, b);
    return err;
}

/* ---------------------------------------------------------------- sample-rate converter */
int ohgpu_src_design(uint32_t rate_in, uint32_t rate_out, uint32_t taps_per_phase, double beta, double f_pass_hz,
                     int32_t* coef_q28, size_t coef_capacity, uint32_t* L, uint32_t* M)
{
    if (!L || !M) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_design: null L/M");
    if (!coef_q28) return design_src(rate_in, rate_out, taps_per_phase, beta, f_pass_hz, nullptr, L, M);
    std::vector<int32_t> coef;
    const int err = design_src(rate_in, rate_out, taps_per_phase, beta, f_pass_hz, &coef, L, M);
    if (err != OHGPU_OK) return err;
    if (coef.size() > coef_capacity)
        return set_error(OHGPU_ERR_INVALID, "ohgpu_src_design: capacity %zu < L*T = %zu", coef_capacity, coef.size());
    memcpy(coef_q28, coef.data(), coef.size() * sizeof(int32_t));
    return OHGPU_OK;
}

uint64_t ohgpu_src_out_frames(uint32_t L, uint32_t M, uint64_t in_frames)
{
    if (in_frames == 0 || M == 0) return 0;
    return (in_frames * L + M - 1) / M;
}

int ohgpu_src_mfma_tables(uint32_t L, uint32_t M, uint32_t T, const int32_t* coef_q28, uint32_t max_blocks_per_row,
                          uint8_t* coef_digits, size_t coef_digits_capacity, void* steps_out, size_t steps_capacity,
                          size_t* coef_digits_bytes, size_t* steps_bytes, uint32_t* block_outputs)
{
    if (!coef_q28 || L == 0 || M == 0 || (uint64_t)L * T > (1u << 22) || max_blocks_per_row == 0 || max_blocks_per_row > 64)
        return set_error(OHGPU_ERR_INVALID, "ohgpu_src_mfma_tables: bad argument");
    std::vector<uint8_t> adig;
    std::vector<MfStep> steps;
    const uint32_t L_blk = T == 32 ? src_block_outputs(L, 6) : 0;
    if (L_blk == 0 || !build_mfma_tables(L, M, T, coef_q28, L_blk, max_blocks_per_row, &adig, &steps))
        return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_mfma_tables: L=%u M=%u T=%u does not fit the 16-output tiling", L, M, T);
    if (coef_digits_bytes) *coef_digits_bytes = adig.size();
    if (steps_bytes) *steps_bytes = steps.size() * sizeof(MfStep);
    if (block_outputs) *block_outputs = L_blk;
    if (coef_digits) {
        if (coef_digits_capacity < adig.size()) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_mfma_tables: coef_digits too small");
        memcpy(coef_digits, adig.data(), adig.size());
    }
    if (steps_out) {
        if (steps_capacity < steps.size() * sizeof(MfStep)) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_mfma_tables: steps too small");
        memcpy(steps_out, steps.data(), steps.size() * sizeof(MfStep));
    }
    return OHGPU_OK;
}

int ohgpu_src_mfma_halfband_tables(const int32_t* coef_q28, uint8_t* image, int64_t* bias, uint32_t* block_outputs)
{
    if (!coef_q28 || !image) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_mfma_halfband_tables: null argument");
    const uint32_t L_blk = src_block_outputs(1, 6);
    std::vector<MfStep> steps;
    std::vector<uint8_t> amat;
    if (L_blk == 0 || !build_mfma_halfband(coef_q28, L_blk, &steps, &amat) || steps.empty())
        return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_mfma_halfband_tables: not a half-band decimator of 64 taps");
    memcpy(image, amat.data(), kMfStepImage);
    // (the steps carry the bias in pieces, the same for every output: bits 0..15 and, signed, bits 16..)
    if (bias) *bias = (int64_t)steps[0].b0[0] + ((int64_t)(int32_t)steps[0].b1[0]) * 65536 + ((int64_t)(int32_t)steps[0].b2[0]) * 4294967296ll;
    if (block_outputs) *block_outputs = L_blk;
    return OHGPU_OK;
}

}  // extern "C"

namespace {
// What the planner and the dispatch look at in a filter, from its coefficients alone (no device): the exactness bound's figure, the
// half-band structure, and whether the matrix-pipe kernels' tables exist for it (and which).  ohgpu_src_create and
// ohgpu_src_plan_digest both come through here, so that the digest's plan IS the plan.  Returns false with the error set.
struct SrcTables { std::vector<uint8_t> amat; std::vector<MfStep> steps; bool made = false; };
bool src_describe(uint32_t L, uint32_t M, uint32_t T, const int32_t* coef_q28, ohgpu_src* s, SrcTables* tables, const char* who)
{
    int64_t max_sum_abs = 0;
    for (uint32_t p = 0; p < L; p++) {
        int64_t sabs = 0;
        for (uint32_t k = 0; k < T; k++) {
            const int32_t q = coef_q28[(size_t)p * T + k];
            sabs += q < 0 ? -(int64_t)q : (int64_t)q;
        }
        if (sabs > max_sum_abs) max_sum_abs = sabs;
        if (sabs >= ((int64_t)1 << 30)) {
            set_error(OHGPU_ERR_INVALID, "%s: phase %u has sum|c| = %lld >= 2^30 (exact fp64 accumulation bound)", who, p, (long long)sabs);
            return false;
        }
    }
    s->L = L; s->M = M; s->T = T;
    s->max_sum_abs = max_sum_abs;
    // a half-band 2:1 decimator (what ohgpu_src_design makes for 96 -> 48 kHz): of its odd taps only the centre one is not zero
    s->halfband = L == 1 && M == 2 && T == 64 && coef_q28[T - 1] == 0;
    for (uint32_t k = 1; k < T && s->halfband; k += 2)
        if (k != T / 2 - 1 && coef_q28[k] != 0) s->halfband = false;
    // the matrix-pipe kernels' digit tables, for the block length the planner gives 24-bit stereo output (rows of up to 8 blocks)
    const uint32_t mf_L_blk = (T == 32 || s->halfband) ? src_block_outputs(L, 6) : 0;
    std::vector<uint8_t> adig;
    s->mf_halfband = false;
    s->mf_L_blk = 0; s->mf_kb_cap = 0;
    if (mf_L_blk != 0 && s->halfband) {
        tables->made = build_mfma_halfband(coef_q28, mf_L_blk, &tables->steps, &tables->amat);
        s->mf_halfband = tables->made;
    } else if (mf_L_blk != 0 && build_mfma_tables(L, M, T, coef_q28, mf_L_blk, 8, &adig, &tables->steps)) {
        build_mfma_images(adig, tables->steps, L, &tables->amat);
        tables->made = true;
    }
    if (tables->made) {
        s->mf_L_blk = mf_L_blk;
        s->mf_kb_cap = s->mf_halfband ? 1 : 8;
    }
    return true;
}
}  // namespace

extern "C" {

int ohgpu_src_create(ohgpu_ctx* ctx, uint32_t L, uint32_t M, uint32_t T, const int32_t* coef_q28, ohgpu_src** out)
{
    CTX_GUARD("ohgpu_src_create");
    if (!out || !coef_q28) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_create: null argument");
    *out = nullptr;
    // (M < 2^15: a descriptor's out_frame0 may be 2^48, and out_frame0 * M is computed in 64 bits)
    if (L == 0 || M == 0 || M >= (1u << 15) || T == 0 || (uint64_t)L * T > (1u << 22))
        return set_error(OHGPU_ERR_INVALID, "ohgpu_src_create: bad geometry L=%u M=%u T=%u", L, M, T);
    const size_t n = (size_t)L * T;
    ohgpu_src* s = new (std::nothrow) ohgpu_src();
    if (!s) return set_error(OHGPU_ERR_NOMEM, "ohgpu_src_create: out of host memory");
    SrcTables tables;
    if (!src_describe(L, M, T, coef_q28, s, &tables, "ohgpu_src_create")) { delete s; return OHGPU_ERR_INVALID; }
    std::vector<double> cd(n);
    for (size_t i = 0; i < n; i++) cd[i] = (double)coef_q28[i];
    hipError_t e = hipMalloc((void**)&s->d_coef, n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_coef_q28, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(s->d_coef, cd.data(), n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_coef_q28, coef_q28, n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && tables.made) {
        e = hipMalloc((void**)&s->d_mf_amat, tables.amat.size());
        if (e == hipSuccess) e = hipMalloc((void**)&s->d_mf_steps, tables.steps.size() * sizeof(MfStep));
        if (e == hipSuccess) e = hipMemcpy(s->d_mf_amat, tables.amat.data(), tables.amat.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->d_mf_steps, tables.steps.data(), tables.steps.size() * sizeof(MfStep), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (s->d_coef) hipFree(s->d_coef);
        if (s->d_coef_q28) hipFree(s->d_coef_q28);
        if (s->d_mf_amat) hipFree(s->d_mf_amat);
        if (s->d_mf_steps) hipFree(s->d_mf_steps);
        delete s;
        return set_error(OHGPU_ERR_DEVICE, "ohgpu_src_create: %s", hipGetErrorString(e));
    }
    *out = s;
    return OHGPU_OK;
}

int ohgpu_src_destroy(ohgpu_ctx* ctx, ohgpu_src* src)
{
    CTX_GUARD("ohgpu_src_destroy");
    if (!src) return OHGPU_OK;
    hipFree(src->d_coef);
    hipFree(src->d_coef_q28);
    if (src->d_mf_amat) hipFree(src->d_mf_amat);
    if (src->d_mf_steps) hipFree(src->d_mf_steps);
    delete src;
    return OHGPU_OK;
}

}  // extern "C"

namespace ohgpu {

// messages [lo, hi) of a resampled batch: validation (ohgpu.h: ohgpu_src_msg_desc), the batch's totals, whether they come in the
// planner's order (a message against its predecessor: src_msg_before) -- and, where `dev` is given, the generic kernel's form of each
void src_check_range(const ohgpu_src* src, const ohgpu_src_msg_desc* descs, size_t lo_i, size_t hi_i, uint64_t src_arena_bytes,
                     uint64_t dst_arena_bytes, DevSrcDesc* dev, SrcRangeResult* out)
{
    SrcRangeResult& r = *out;
    const uint64_t L = src->L, M = src->M, T = src->T;
    const FastDiv64 by_L(L);
    const ohgpu_src_msg_desc& d0 = descs[0];
    for (size_t i = lo_i; i < hi_i; i++) {
        const ohgpu_src_msg_desc& d = descs[i];
        int err = OHGPU_OK;
        if (d.channels < 1 || d.channels > OHGPU_MAX_CHANNELS) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: channels %u outside 1..8", i, d.channels);
        else if (!valid_bits(d.src_bits) || !valid_bits(d.dst_bits)) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: bit depth %u -> %u", i, d.src_bits, d.dst_bits);
        else if (!valid_endian(d.src_endian) || !valid_endian(d.dst_endian)) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: endian %u -> %u", i, d.src_endian, d.dst_endian);
        else if (d.flags & ~(OHGPU_FLAG_RAMP | OHGPU_FLAG_ZERO_LSB32 | OHGPU_FLAG_SRC_PLANAR32)) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: flag bits 0x%x not valid for a resampled message", i, d.flags);
        else if (!(d.flags & OHGPU_FLAG_SRC_PLANAR32) && d.src_plane_stride != 0) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: src_plane_stride without OHGPU_FLAG_SRC_PLANAR32", i);
        else if ((d.flags & OHGPU_FLAG_SRC_PLANAR32) && (d.src_bits == 32 || (d.src_offset & 3) || (d.src_plane_stride & 3) || (d.src_plane_stride >> 34)))
            err = set_error(OHGPU_ERR_INVALID, "src desc %zu: planar source needs 8/16/24-bit samples, 4-byte aligned planes less than 16 GiB apart", i);
        else if (d.ramp_start > OHGPU_RAMP_MAX || d.ramp_end > OHGPU_RAMP_MAX) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: ramp beyond Ramp::kMax", i);
        else if ((d.flags & OHGPU_FLAG_RAMP) && d.n_frames > 131071u) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: ramped message of %u frames", i, d.n_frames);
        else if (d.attenuation != OHGPU_UNITY_ATTENUATION) err = set_error(OHGPU_ERR_UNSUPPORTED, "src desc %zu: attenuation %u (resampled audio is 24-bit; Msg.cpp:2741 allows 16-bit only)", i, d.attenuation);
        else if (d.out_frame0 > (1ull << 48) || d.src_frame0 > (1ull << 48) || d.src_frames > (1ull << 40)) err = set_error(OHGPU_ERR_INVALID, "src desc %zu: frame index out of range", i);
        if (err != OHGPU_OK) { r.fail(err); return; }
        const uint64_t fb_src = (uint64_t)d.channels * (d.src_bits / 8);
        const uint64_t fb_dst = (uint64_t)d.channels * (d.dst_bits / 8);
        const bool planar = (d.flags & OHGPU_FLAG_SRC_PLANAR32) != 0;
        // (planar: the window is one run of src_frames * 4 bytes per plane; the last plane's run ends furthest out -- span_end
        // is overflow-safe, see the fmt batches)
        const uint64_t src_bytes = planar ? d.src_frames * 4 : d.src_frames * fb_src;
        const uint64_t dst_bytes = (uint64_t)d.n_frames * fb_dst;
        uint64_t planes_end = 0;
        if (planar && (!span_end(d.src_offset, d.src_plane_stride, d.channels - 1u, src_bytes, &planes_end) || planes_end > src_arena_bytes ||
                       (d.channels > 1 && d.src_plane_stride < src_bytes))) {
            r.fail(set_error(OHGPU_ERR_BOUNDS, "src desc %zu: %u planes of %llu bytes, %llu apart from %llu, beyond the %llu-byte source arena (or overlapping)", i,
                             d.channels, (unsigned long long)src_bytes, (unsigned long long)d.src_plane_stride, (unsigned long long)d.src_offset,
                             (unsigned long long)src_arena_bytes));
            return;
        }
        if (d.src_offset > src_arena_bytes || src_bytes > src_arena_bytes - d.src_offset) {
            r.fail(set_error(OHGPU_ERR_BOUNDS, "src desc %zu: input window [%llu, +%llu) beyond the %llu-byte source arena", i,
                             (unsigned long long)d.src_offset, (unsigned long long)src_bytes, (unsigned long long)src_arena_bytes));
            return;
        }
        if (d.dst_offset > dst_arena_bytes || dst_bytes > dst_arena_bytes - d.dst_offset) {
            r.fail(set_error(OHGPU_ERR_BOUNDS, "src desc %zu: writes [%llu, +%llu) beyond the %llu-byte destination arena", i,
                             (unsigned long long)d.dst_offset, (unsigned long long)dst_bytes, (unsigned long long)dst_arena_bytes));
            return;
        }
        if (d.n_frames > 0) {
            const uint64_t t_first = d.out_frame0 * M, t_last = (d.out_frame0 + d.n_frames - 1) * M;
            const int64_t n0_first = (int64_t)by_L.div(t_first), n0_last = (int64_t)by_L.div(t_last);
            const int64_t n_lo = n0_first - (int64_t)(T - 1);
            if (n_lo >= 0 ? (uint64_t)n_lo < d.src_frame0 : d.src_frame0 != 0) {
                r.fail(set_error(OHGPU_ERR_BOUNDS, "src desc %zu: filter history starts at input frame %lld but the buffer starts at %llu", i,
                                 (long long)(n_lo < 0 ? 0 : n_lo), (unsigned long long)d.src_frame0));
                return;
            }
            if ((uint64_t)n0_last >= d.src_frame0 + d.src_frames) {
                r.fail(set_error(OHGPU_ERR_BOUNDS, "src desc %zu: needs input frame %lld but the buffer ends at %llu", i,
                                 (long long)n0_last, (unsigned long long)(d.src_frame0 + d.src_frames)));
                return;
            }
            const int64_t lo = n_lo < 0 ? 0 : n_lo;
            r.in_frames += (uint64_t)(n0_last - n0_first + 1);   // new input frames this message advances over
            r.src_bytes_touched += (uint64_t)(n0_last - lo + 1) * (planar ? 4ull * d.channels : fb_src);
        }
        if (dev) dev[i] = src_convert_desc(d, L, M);
        r.out_frames += d.n_frames;
        r.dst_bytes_written += dst_bytes;
        if (d.n_frames > r.max_frames) r.max_frames = d.n_frames;
        if (d.channels != d0.channels || d.src_bits != d0.src_bits || d.src_endian != d0.src_endian || d.dst_bits != d0.dst_bits ||
            d.dst_endian != d0.dst_endian || planar != ((d0.flags & OHGPU_FLAG_SRC_PLANAR32) != 0)) r.uniform = false;
        // (the planner's order, message against predecessor -- the range's first against the last of the range before it: a caller
        // that lists its streams one after the other, each in time order, spares the planner its own pass and the sort)
        if (i > 0 && r.ordered && src_msg_before(descs[i], descs[i - 1], planar ? 4u : (uint32_t)fb_src, (uint32_t)fb_dst)) r.ordered = false;
    }
}

// A resampled batch's messages checked and -- if they share a layout -- planned (b->fast), by the shorter of two routes; the batch's
// totals, `uniform` and layout fields are set.  `dev`: where to put the generic kernel's form of every message (null: nowhere).
// `digest`: the plan hashed instead of uploaded (ohgpu_src_plan_digest: ctx has no device behind it).
int src_check_and_plan(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_src_msg_desc* descs, size_t n, DevSrcDesc* dev, PlanDigest* digest)
{
    const ohgpu_src* src = b->src;
    auto layout_of_first = [&] {
        const ohgpu_src_msg_desc& d0 = descs[0];
        b->channels = d0.channels; b->src_bits = d0.src_bits; b->src_endian = d0.src_endian;
        b->dst_bits = d0.dst_bits; b->dst_endian = d0.dst_endian;
        b->src_planar = (d0.flags & OHGPU_FLAG_SRC_PLANAR32) != 0;
    };
#ifdef OHGPU_PLAN_TIMING
    const auto tp0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
#endif
    // A large batch is 64 bytes a message to read -- 32 MB for the headline's half a million -- and both the checks and the planner's
    // cut into segments are bound by exactly that.  So the planner is let loose on the messages FIRST, on the usual caller's terms
    // (one layout, streams one after the other in time order), and checks each message itself the first time it looks at it; a batch
    // that is not what it assumed -- several layouts, another order, a layout no block kernel has -- goes the two-pass way below.
    if (!dev && n >= 4096) {
        SrcRangeResult first;
        src_check_range(src, descs, 0, 1, b->src_arena_bytes, b->dst_arena_bytes, nullptr, &first);      // (its layout is the batch's: the planner's geometry needs it sane)
        if (first.err != OHGPU_OK) return set_error(first.err, "%s", first.msg);
        layout_of_first();
        PlanFusedCheck fused;
        fused.src = src;
        const int err = plan_src_fast(ctx, b, descs, n, true, digest, &fused);
#ifdef OHGPU_PLAN_TIMING
        fprintf(stderr, "[plan timing] check fused with the plan: %.2f ms (checked %d, retry %d)\n", since(tp0), (int)fused.checked, (int)fused.retry);
#endif
        if (err != OHGPU_OK) return err;
        if (fused.checked && fused.total.err != OHGPU_OK) return set_error(fused.total.err, "%s", fused.total.msg);
        if (fused.checked && !fused.retry) {
            b->in_frames = fused.total.in_frames; b->out_frames = fused.total.out_frames;
            b->src_bytes_touched = fused.total.src_bytes_touched; b->dst_bytes_written = fused.total.dst_bytes_written;
            b->max_frames = fused.total.max_frames;
            return OHGPU_OK;                                 // (checked and uniform; a plan, or none: no whole block anywhere -- the generic kernel's batch)
        }
    }
    // every message checked on its own: in ranges, on as many threads as the batch is worth (the first error in message order is the
    // one reported); then the plan
    bool ordered = true;
    {
        const unsigned n_thr = plan_threads(n, 16384);
        std::vector<SrcRangeResult> res(n_thr);
        parallel_ranges(n, n_thr, [&](unsigned t, size_t lo, size_t hi) { src_check_range(src, descs, lo, hi, b->src_arena_bytes, b->dst_arena_bytes, dev, &res[t]); });
        b->in_frames = b->out_frames = b->src_bytes_touched = b->dst_bytes_written = 0;
        b->max_frames = 0;
        b->uniform = true;
        for (const SrcRangeResult& r : res) {
            if (r.err != OHGPU_OK) return set_error(r.err, "%s", r.msg);
            b->in_frames += r.in_frames; b->out_frames += r.out_frames;
            b->src_bytes_touched += r.src_bytes_touched; b->dst_bytes_written += r.dst_bytes_written;
            if (r.max_frames > b->max_frames) b->max_frames = r.max_frames;
            b->uniform = b->uniform && r.uniform;
            ordered = ordered && r.ordered;
        }
        if (n > 0) layout_of_first();
    }
#ifdef OHGPU_PLAN_TIMING
    const auto tp1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[plan timing] validate%s %.2f ms\n", dev ? "+convert" : "", since(tp0));
#endif
    int err = OHGPU_OK;
    if (b->uniform && n > 0) err = plan_src_fast(ctx, b, descs, n, ordered, digest);
#ifdef OHGPU_PLAN_TIMING
    fprintf(stderr, "[plan timing] plan_src_fast %.2f ms\n", since(tp1));
#endif
    return err;
}

}  // namespace ohgpu

extern "C" {

int ohgpu_src_batch_create(ohgpu_ctx* ctx, const ohgpu_src* src, const ohgpu_src_msg_desc* descs, size_t n,
                           uint64_t src_arena_bytes, uint64_t dst_arena_bytes, ohgpu_batch** out)
{
    CTX_GUARD("ohgpu_src_batch_create");
    if (!out || !src || (n && !descs)) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_create: null argument");
    *out = nullptr;
    if (n > 0xffffffffull) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_create: too many descriptors");
    ohgpu_batch* b = new (std::nothrow) ohgpu_batch();
    if (!b) return set_error(OHGPU_ERR_NOMEM, "ohgpu_src_batch_create: out of host memory");
    b->kind = kBatchSrc;
    b->n = n;
    b->src = src;
    b->src_arena_bytes = src_arena_bytes;
    b->dst_arena_bytes = dst_arena_bytes;
    b->uniform = true;
    // The generic kernel's per-message form (56 bytes a message: 28 MB written for the headline's half a million, half of a checking
    // pass's time) is made only where that kernel will run the whole batch: a batch created while variant 1 is in force, or one no
    // block kernel takes (below).  A batch planned for the block kernels keeps nothing per message.
    const bool keep_generic = ctx->variant == 1;
    auto convert_all = [&]() -> bool {
        b->host_descs.reset((DevSrcDesc*)host_alloc_huge((n ? n : 1) * sizeof(DevSrcDesc)));     // (not zeroed here: the ranges' threads touch their own pages)
        return b->host_descs != nullptr;
    };
    if (keep_generic && !convert_all()) { delete b; return set_error(OHGPU_ERR_NOMEM, "ohgpu_src_batch_create: out of host memory"); }
    int err = src_check_and_plan(ctx, b, descs, n, b->host_descs.get(), nullptr);
    if (err != OHGPU_OK) { delete b; return err; }
    if (err == OHGPU_OK && !b->uniform) {
        // Mixed layouts (channel counts, depths, byte orders, planar or packed sources): the block kernels are instantiated per
        // layout, so the batch becomes one uniform batch per layout, messages in their given order.  (More than 32 layouts: the
        // generic kernel takes the whole batch, as it did for every mixed batch before.)
        auto key = [](const ohgpu_src_msg_desc& d) -> uint64_t {
            return (uint64_t)d.channels | ((uint64_t)d.src_bits << 8) | ((uint64_t)d.dst_bits << 16) | ((uint64_t)d.src_endian << 24) |
                   ((uint64_t)d.dst_endian << 32) | ((uint64_t)((d.flags & OHGPU_FLAG_SRC_PLANAR32) ? 1 : 0) << 40);
        };
        std::vector<uint64_t> keys;
        std::vector<std::vector<ohgpu_src_msg_desc>> groups;
        for (size_t i = 0; i < n && keys.size() <= 32; i++) {
            const uint64_t k = key(descs[i]);
            size_t g = 0;
            while (g < keys.size() && keys[g] != k) g++;
            if (g == keys.size()) { keys.push_back(k); groups.emplace_back(); }
            groups[g].push_back(descs[i]);
        }
        if (keys.size() <= 32) {
            for (size_t g = 0; g < groups.size() && err == OHGPU_OK; g++) {
                ohgpu_batch* part = nullptr;
                err = ohgpu_src_batch_create(ctx, src, groups[g].data(), groups[g].size(), src_arena_bytes, dst_arena_bytes, &part);
                if (err == OHGPU_OK) b->parts.push_back(part);
            }
        }
    }
    if (err == OHGPU_OK && !b->host_descs && !b->fast.enabled && b->parts.empty() && n > 0) {
        // no block kernel takes this batch (a layout none is instantiated for, more than 32 layouts, nothing block-aligned): the generic
        // kernel will run it whole, from its own form of the messages -- made now, in a second pass over descriptors known to be good
        if (!convert_all()) err = set_error(OHGPU_ERR_NOMEM, "ohgpu_src_batch_create: out of host memory");
        else {
            DevSrcDesc* const dev = b->host_descs.get();
            const uint64_t L = src->L, M = src->M;
            parallel_ranges(n, plan_threads(n, 16384), [&](unsigned, size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) dev[i] = src_convert_desc(descs[i], L, M); });
        }
    }
    if (err != OHGPU_OK) {
        for (ohgpu_batch* part : b->parts) ohgpu_batch_destroy(ctx, part);
        if (b->d_descs) ctx_dev_free(ctx, b->d_descs);
        delete b;
        return err;
    }
    *out = b;
    return OHGPU_OK;
}

int ohgpu_src_batch_plan(const ohgpu_batch* b, uint64_t* block_kernel_out_frames, uint64_t* generic_pieces)
{
    if (!b || b->kind != kBatchSrc) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_plan: not a src batch");
    if (!b->parts.empty()) {
        uint64_t fast = 0, pieces = 0;
        for (const ohgpu_batch* part : b->parts) {
            uint64_t f = 0, p = 0;
            ohgpu_src_batch_plan(part, &f, &p);
            fast += f; pieces += p;
        }
        if (block_kernel_out_frames) *block_kernel_out_frames = fast;
        if (generic_pieces) *generic_pieces = pieces;
        return OHGPU_OK;
    }
    if (block_kernel_out_frames) *block_kernel_out_frames = b->fast.enabled ? b->fast.fast_out_frames : 0;
    if (generic_pieces) *generic_pieces = b->fast.enabled ? b->fast.n_rem : b->n;
    return OHGPU_OK;
}

int ohgpu_set_plan_threads(int threads)
{
    if (threads < 0 || threads > 256) return set_error(OHGPU_ERR_INVALID, "ohgpu_set_plan_threads: %d", threads);
    g_plan_threads = threads;
    return OHGPU_OK;
}

int ohgpu_src_plan_digest(uint32_t L, uint32_t M, uint32_t taps_per_phase, const ohgpu_src_msg_desc* descs, size_t n,
                          uint64_t src_arena_bytes, uint64_t dst_arena_bytes, int kernel_variant,
                          const int32_t* coef_q28, int num_cus,
                          uint64_t* digest, uint64_t* units, uint64_t* generic_pieces, int* kernel)
{
    if (!descs || n == 0 || L == 0 || M == 0 || taps_per_phase == 0 || (uint64_t)L * taps_per_phase > (1u << 22))
        return set_error(OHGPU_ERR_INVALID, "ohgpu_src_plan_digest: bad argument");
    // a filter and a context as far as the planner looks at them: no device behind either.  With the coefficients the filter is
    // described exactly as ohgpu_src_create describes it (src_describe: the half-band form, the tables, the gain); without them it
    // is "a polyphase filter of sane gain whose tables exist if its geometry allows".
    ohgpu_src flt{};
    SrcTables tables;
    if (coef_q28) {
        if (!src_describe(L, M, taps_per_phase, coef_q28, &flt, &tables, "ohgpu_src_plan_digest")) return OHGPU_ERR_INVALID;
    } else {
        flt.L = L; flt.M = M; flt.T = taps_per_phase;
        flt.max_sum_abs = (int64_t)1 << 28;
        flt.halfband = false;
        flt.mf_L_blk = taps_per_phase == 32 ? src_block_outputs(L, 6) : 0;
        flt.mf_kb_cap = 8;
        tables.made = flt.mf_L_blk != 0;
    }
    flt.d_mf_amat = tables.made ? (uint8_t*)&flt : nullptr;         // (only its being there is looked at)
    ohgpu_ctx ctx{};
    ctx.variant = kernel_variant;
    ctx.num_cus = num_cus > 0 ? num_cus : 256;
    ohgpu_batch b;
    b.kind = kBatchSrc; b.n = n; b.src = &flt; b.src_arena_bytes = src_arena_bytes; b.dst_arena_bytes = dst_arena_bytes; b.uniform = true;
    PlanDigest pd{};
    const int err = src_check_and_plan(&ctx, &b, descs, n, nullptr, &pd);
    if (err != OHGPU_OK) return err;
    if (digest) *digest = pd.hash;
    if (units) *units = pd.units;
    if (generic_pieces) *generic_pieces = pd.pieces;
    if (kernel) *kernel = pd.kernel;
    return OHGPU_OK;
}

int ohgpu_src_batch_units(const ohgpu_batch* b, uint64_t* units, uint64_t* long_units)
{
    if (!b || b->kind != kBatchSrc) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_units: not a src batch");
    uint64_t u = 0, l = 0;
    if (!b->parts.empty()) {
        for (const ohgpu_batch* part : b->parts) {
            uint64_t pu = 0, pl = 0;
            ohgpu_src_batch_units(part, &pu, &pl);
            u += pu; l += pl;
        }
    } else if (b->fast.enabled) {
        u = b->fast.lean ? b->fast.n_lean : b->fast.n_work;
        l = b->fast.lean ? b->fast.n_long : 0;
    }
    if (units) *units = u;
    if (long_units) *long_units = l;
    return OHGPU_OK;
}

// Which kernel runs a (uniform) resampled batch's whole blocks: ONE decision, taken from the plan (what it serves: made under the
// variant in force at creation) and the variant in force NOW, and used by the launch and by the name a benchmark prints alike.
enum SrcKernel { kSrcGeneric, kSrcWg, kSrcMfma, kSrcLean, kSrcBlock };
static SrcKernel src_kernel_choice(const ohgpu_ctx* ctx, const ohgpu_batch* b, bool arena_aligned = true)
{
    const int v = ctx->variant;
    // (the block kernels' staging moves aligned 16-byte pieces of the arena; a plan for the workgroup kernel alone has nothing for
    // a variant that asks for another)
    if (v == 1 || !b->fast.enabled || !arena_aligned || (b->fast.wg_only && v != 0)) return kSrcGeneric;
    if (b->fast.mfma_wg && v == 0) return kSrcWg;                                         // the taps on the matrix pipe (round 4), a unit per workgroup
#ifdef OHGPU_LEGACY_KERNELS
    if (b->fast.mfma && (v == 0 || v == 3 || v == 5)) return kSrcMfma;                   // ... a unit per wave (variant 5, and where the workgroup kernel's block geometry does not hold)
    if (b->fast.lean && (v != 2 || b->fast.lean_only || !b->fast.d_work)) return kSrcLean;   // round 2's, under every variant but 2 -- and under 2 where round 1's has no layout or no tables
    if (b->fast.d_work) return kSrcBlock;                                                  // round 1's (variant 2; a filter beyond the lean kernel's rounding bound under any)
#else
    if (b->fast.lean) return kSrcLean;                                                     // round 2's, under every other variant (rounds 1's and 4's unit-per-wave kernels as variants: legacy builds)
    if (b->fast.d_work) return kSrcBlock;                                                  // round 1's: the fallback for a filter beyond the lean kernel's rounding bound
#endif
    return kSrcGeneric;
}
static const char* src_kernel_name(SrcKernel k)
{
    switch (k) {
    case kSrcWg: return "src_mfma_wg_kernel";
    case kSrcMfma: return "src_mfma_kernel";
    case kSrcLean: return "src_lean_kernel";
    case kSrcBlock: return "src_block_kernel";
    default: return "src_kernel_v1";
    }
}
static const char* src_kernel_of(const ohgpu_ctx* ctx, const ohgpu_batch* b) { return src_kernel_name(src_kernel_choice(ctx, b)); }

int ohgpu_src_batch_kernel_name(ohgpu_ctx* ctx, const ohgpu_batch* batch, char* out, size_t cap)
{
    CTX_GUARD("ohgpu_src_batch_kernel_name");
    if (!batch || batch->kind != kBatchSrc || !out || cap == 0) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_kernel_name: bad argument");
    std::string name;
    if (ctx->variant != 1 && !batch->parts.empty()) {
        for (const ohgpu_batch* part : batch->parts) {
            const char* k = src_kernel_of(ctx, part);
            if (name.find(k) == std::string::npos) name += (name.empty() ? "" : ",") + std::string(k);
        }
    } else {
        name = src_kernel_of(ctx, batch);
    }
    snprintf(out, cap, "%s", name.c_str());
    return OHGPU_OK;
}

int ohgpu_src_batch_occupancy(ohgpu_ctx* ctx, const ohgpu_batch* batch, int* workgroups_per_cu, int* designed_for, uint32_t* lds_bytes)
{
    CTX_GUARD("ohgpu_src_batch_occupancy");
    if (!batch || batch->kind != kBatchSrc || !workgroups_per_cu) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_occupancy: bad argument");
    const ohgpu_batch* one = batch->parts.empty() ? batch : batch->parts.front();
    if (src_kernel_choice(ctx, one) != kSrcWg) return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_batch_occupancy: the batch does not run on the workgroup kernel (%s)", src_kernel_of(ctx, one));
    WgOccupancy q;
    OHGPU_HIP_TRY(launch_src_mfma_wg(ctx, one, nullptr, nullptr, nullptr, &q));
    *workgroups_per_cu = q.groups_per_cu;
    if (designed_for) *designed_for = q.designed_for;
    if (lds_bytes) *lds_bytes = q.lds_bytes;
    return OHGPU_OK;
}

// ohgpu_measure_shader_clock: every wave runs a chain of dependent integer multiply-adds (about 0.2 ms at 2.4 GHz); wave 0 of every
// workgroup reports the shader cycles and the 100 MHz reference ticks its chain took.
__global__ __launch_bounds__(256) void clock_probe_kernel(uint64_t* __restrict__ out, uint32_t iters)
{
    uint32_t x = threadIdx.x + 1u;
    const uint64_t c0 = __builtin_readcyclecounter();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    asm volatile("" : "+v"(x));
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = (r1 - r0) + (x == 0x12345u ? 1u : 0u);      // (x is used)
    }
}

int ohgpu_measure_shader_clock(ohgpu_ctx* ctx, void* stream, double* mhz)
{
    CTX_GUARD("ohgpu_measure_shader_clock");
    if (!mhz) return set_error(OHGPU_ERR_INVALID, "ohgpu_measure_shader_clock: null result");
    hipStream_t s = pick_stream(ctx, stream);
    const uint32_t groups = ctx->num_cus > 0 ? (uint32_t)ctx->num_cus : 256u;
    uint64_t* d = nullptr;
    OHGPU_HIP_TRY(hipMalloc((void**)&d, groups * 2 * sizeof(uint64_t)));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(groups), dim3(256), 0, s, d, 60000u);
    std::vector<uint64_t> h(groups * 2);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(OHGPU_ERR_DEVICE, "ohgpu_measure_shader_clock: %s", hipGetErrorString(e));
    double cyc = 0.0, ref = 0.0;
    for (uint32_t i = 0; i < groups; i++) { cyc += (double)h[2 * i]; ref += (double)h[2 * i + 1]; }
    if (ref <= 0.0) return set_error(OHGPU_ERR_DEVICE, "ohgpu_measure_shader_clock: the reference counter did not advance");
    *mhz = cyc / ref * 100.0;
    return OHGPU_OK;
}

static int src_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream, hipEvent_t ev_start, hipEvent_t ev_stop);

int ohgpu_src_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream)
{
    return src_batch_run(ctx, batch, src_base, dst_base, stream, nullptr, nullptr);
}

int ohgpu_src_batch_run_timed(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream, void* start_event, void* stop_event)
{
    if (!start_event || !stop_event) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_run_timed: null event");
    return src_batch_run(ctx, batch, src_base, dst_base, stream, (hipEvent_t)start_event, (hipEvent_t)stop_event);
}

// (ev_start / ev_stop: both or neither.  A batch that is ONE launch of the workgroup matrix kernel carries them on its dispatch; any other
// -- several layouts, block-unaligned pieces on the generic kernel behind the block kernel, another kernel -- gets them recorded in
// front of its first launch and behind its last)
static int src_batch_run(ohgpu_ctx* ctx, const ohgpu_batch* batch, const void* src_base, void* dst_base, void* stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    CTX_GUARD("ohgpu_src_batch_run");
    if (!batch || batch->kind != kBatchSrc) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_run: not a src batch");
    hipStream_t s = pick_stream(ctx, stream);
    if (batch->n == 0) {
        if (ev_start) { OHGPU_HIP_TRY(hipEventRecord(ev_start, s)); OHGPU_HIP_TRY(hipEventRecord(ev_stop, s)); }
        return OHGPU_OK;
    }
    if (!src_base || !dst_base) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_run: null arena pointer");
    if (ctx->variant != 1 && !batch->parts.empty()) {               // one uniform batch per layout
        // "nothing is launched" on refusal holds for the whole batch: every part is asked first whether it is free (a part still
        // running on another stream refuses), and only then does the first one launch
        for (const ohgpu_batch* part : batch->parts)
            if (batch_busy_on_another_stream(part, s))
                return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_run: a part of the batch is still running on another stream (its unit counters "
                                 "serve one launch at a time: wait for it, use the same stream, or create a second batch); nothing was launched");
        if (ev_start) OHGPU_HIP_TRY(hipEventRecord(ev_start, s));
        for (const ohgpu_batch* part : batch->parts) {
            const int err = ohgpu_src_batch_run(ctx, part, src_base, dst_base, s);
            if (err != OHGPU_OK) return err;                        // (a device error: the destination may be partly written, as for any failed launch)
        }
        if (ev_stop) OHGPU_HIP_TRY(hipEventRecord(ev_stop, s));
        return OHGPU_OK;
    }
    const SrcKernel which = src_kernel_choice(ctx, batch, ((uintptr_t)src_base & 15u) == 0);
    // (a batch that is ONE launch of the workgroup matrix kernel: its dispatch carries an event -- the caller's two, or the batch's
    // "last launch done" -- instead of a marker packet behind it: back-to-back launches were 10 us apart with the marker)
    const bool one_launch = which == kSrcWg && batch->fast.n_rem == 0;
    const bool on_dispatch = ev_start && one_launch;
    if (ev_start && !on_dispatch) OHGPU_HIP_TRY(hipEventRecord(ev_start, s));
    if (which != kSrcGeneric) {
        const int claim = claim_single_launch(batch, s, "ohgpu_src_batch_run");        // (the block kernels' unit counters are the batch's)
        if (claim != OHGPU_OK) return claim;
        if (batch->fast.planes_ready) OHGPU_HIP_TRY(hipStreamWaitEvent(s, batch->fast.planes_ready, 0));     // (the ramp planes are filled on the context's stream)
        // whole phase-aligned blocks on the chosen block kernel, block-unaligned heads/tails on the generic one
        switch (which) {
        case kSrcWg: {
            WgOccupancy x;
            x.query = false; x.start = on_dispatch ? ev_start : nullptr; x.stop = on_dispatch ? ev_stop : batch->last_done;
            OHGPU_HIP_TRY(launch_src_mfma_wg(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, s, one_launch ? &x : nullptr));
            break;
        }
#ifdef OHGPU_LEGACY_KERNELS
        case kSrcMfma: OHGPU_HIP_TRY(launch_src_mfma(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, s)); break;
#endif
        case kSrcBlock: OHGPU_HIP_TRY(launch_src_block(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, s)); break;
        default: OHGPU_HIP_TRY(launch_src_lean(ctx, batch, (const uint8_t*)src_base, (uint8_t*)dst_base, s)); break;
        }
        OHGPU_HIP_TRY(launch_src_v1(ctx, batch->fast.d_rem, batch->fast.n_rem, batch->src, (const uint8_t*)src_base, (uint8_t*)dst_base, s));
        if (!one_launch) launched(batch, s);
        else batch->last_untracked = on_dispatch;            // (else: last_done rode on the dispatch)
    } else {
        if (!batch->host_descs)
            return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_batch_run: this batch was planned for the block kernels and keeps no per-message descriptors for the "
                             "generic kernel, which %s asks for: create it while ohgpu_set_kernel_variant(1) is in force%s",
                             ctx->variant == 1 ? "kernel variant 1" : (batch->fast.wg_only ? "this kernel variant (the plan is the workgroup matrix kernel's alone)" : "a source arena that is not 16-byte aligned"),
                             ctx->variant == 1 ? "" : ", or run it under the variant / with the alignment it was planned for");
        {   // (the whole batch on the generic kernel: its per-message descriptors go to the device the first time this happens)
            std::lock_guard<std::mutex> hold(batch->lazy);
            if (!batch->d_descs && batch->n) {
                const int err = upload_batch(ctx, const_cast<ohgpu_batch*>(batch), batch->host_descs.get(), batch->n * sizeof(DevSrcDesc));
                if (err != OHGPU_OK) return err;
            }
        }
        OHGPU_HIP_TRY(launch_src_v1(ctx, batch->d_descs, batch->n, batch->src, (const uint8_t*)src_base, (uint8_t*)dst_base, s));
    }
    if (ev_stop && !on_dispatch) OHGPU_HIP_TRY(hipEventRecord(ev_stop, s));
    return OHGPU_OK;
}

int ohgpu_src_batch_block(const ohgpu_batch* b, uint32_t* block_outputs, uint32_t* block_inputs)
{
    if (!b || b->kind != kBatchSrc) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_block: not a src batch");
    const ohgpu_batch* p = b->parts.empty() ? b : b->parts[0];
    if (!p->fast.enabled) return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_batch_block: the batch has no block-kernel plan");
    for (const ohgpu_batch* q : b->parts)
        if (!q->fast.enabled || q->fast.params.L_blk != p->fast.params.L_blk) return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_batch_block: the batch's layouts are cut into blocks of different lengths");
    if (block_outputs) *block_outputs = p->fast.params.L_blk;
    if (block_inputs) *block_inputs = p->fast.params.M_blk;
    return OHGPU_OK;
}

int ohgpu_src_batch_advance(ohgpu_ctx* ctx, ohgpu_batch* b, uint64_t blocks)
{
    CTX_GUARD("ohgpu_src_batch_advance");
    if (!b || b->kind != kBatchSrc) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_advance: not a src batch");
    std::vector<ohgpu_batch*> all(b->parts.begin(), b->parts.end());
    if (all.empty()) all.push_back(b);
    for (const ohgpu_batch* p : all) {
        if (!p->fast.enabled)
            return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_batch_advance: the batch (or one of its layouts) has no block-kernel plan: its generic-kernel descriptors hold the positions themselves");
        if (p->fast.stream_start)
            return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_advance: a message of the batch starts its stream (its filter window reaches in front of input frame 0, "
                             "read as zeros): the same window a period later holds real history the batch's source windows do not declare");
    }
    // Nothing of the plan names an absolute position: a unit is where its rows lie in the two arenas, a ramp job where its frames lie in
    // their message, a generic-kernel piece its window relative to the buffer -- and a whole number of blocks later every message has
    // the phase it had.  The plan IS the next period's plan.
    for (ohgpu_batch* p : all) p->fast.advanced_blocks += blocks;
    if (all[0] != b) b->fast.advanced_blocks += blocks;
    return OHGPU_OK;
}

int ohgpu_src_batch_set_ramps(ohgpu_ctx* ctx, ohgpu_batch* b, const uint16_t* ramp_start, const uint16_t* ramp_end, size_t n)
{
    CTX_GUARD("ohgpu_src_batch_set_ramps");
    if (!b || b->kind != kBatchSrc || !ramp_start || !ramp_end) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_set_ramps: bad argument");
    if (n != b->n) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_set_ramps: %zu endpoints for a batch of %zu messages", n, b->n);
    if (!b->parts.empty()) return set_error(OHGPU_ERR_UNSUPPORTED, "ohgpu_src_batch_set_ramps: a batch of several layouts (create one batch per layout to re-ramp it)");
    SrcFastPlan& f = b->fast;
    // (only the messages that carry a ramp are looked at: the flags are the plan's)
    for (uint32_t m : f.job_msg) if (ramp_start[m] > OHGPU_RAMP_MAX || ramp_end[m] > OHGPU_RAMP_MAX) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_set_ramps: message %u: ramp beyond Ramp::kMax", m);
    for (uint32_t m : f.rem_msg) if (ramp_start[m] > OHGPU_RAMP_MAX || ramp_end[m] > OHGPU_RAMP_MAX) return set_error(OHGPU_ERR_INVALID, "ohgpu_src_batch_set_ramps: message %u: ramp beyond Ramp::kMax", m);
    OHGPU_HIP_TRY(batch_wait_last_launch(b));                                    // (the batch's last launch reads what is rewritten here)
    hipStream_t s0 = ctx->stream;
    if (f.enabled) {
        for (size_t k = 0; k < f.host_jobs.size(); k++) { f.host_jobs[k].ramp_start = ramp_start[f.job_msg[k]]; f.host_jobs[k].ramp_end = ramp_end[f.job_msg[k]]; }
        for (size_t k = 0; k < f.host_rem.size(); k++) { f.host_rem[k].ramp_start = ramp_start[f.rem_msg[k]]; f.host_rem[k].ramp_end = ramp_end[f.rem_msg[k]]; }
        if (!f.host_jobs.empty()) {
            OHGPU_HIP_TRY(hipMemcpyAsync(f.d_ramp_jobs, f.host_jobs.data(), f.host_jobs.size() * sizeof(RampJob), hipMemcpyHostToDevice, s0));
            OHGPU_HIP_TRY(hipMemsetAsync(f.d_planes, 0xff, (f.plane_entries ? f.plane_entries : 8) * sizeof(uint16_t), s0));
            OHGPU_HIP_TRY(launch_ramp_planes(ctx, f.d_ramp_jobs, (uint32_t)f.host_jobs.size(), f.d_planes, s0));
        }
        if (!f.host_rem.empty()) OHGPU_HIP_TRY(hipMemcpyAsync(f.d_rem, f.host_rem.data(), f.host_rem.size() * sizeof(DevSrcDesc), hipMemcpyHostToDevice, s0));
        if (f.planes_ready) OHGPU_HIP_TRY(hipEventRecord(f.planes_ready, s0));       // (a run on any stream waits for this: the new planes)
        OHGPU_HIP_TRY(hipStreamSynchronize(s0));                                    // (the host copies above are the caller's to change again)
    }
    if (b->host_descs) {                                                           // the generic kernel's form of every message (a batch created under variant 1)
        DevSrcDesc* const dev = b->host_descs.get();
        for (size_t i = 0; i < n; i++) { dev[i].ramp_start = ramp_start[i]; dev[i].ramp_end = ramp_end[i]; }
        std::lock_guard<std::mutex> hold(b->lazy);
        if (b->d_descs) OHGPU_HIP_TRY(hipMemcpy(b->d_descs, dev, n * sizeof(DevSrcDesc), hipMemcpyHostToDevice));
    }
    return OHGPU_OK;
}

int ohgpu_src_process_host(ohgpu_ctx* ctx, const ohgpu_src* src, const ohgpu_src_msg_desc* descs, size_t n,
                           const void* src_host, uint64_t src_bytes, void* dst_host, uint64_t dst_bytes)
{
    CTX_GUARD("ohgpu_src_process_host");
    ohgpu_batch* b = nullptr;
    int err = ohgpu_src_batch_create(ctx, src, descs, n, src_bytes, dst_bytes, &b);
    if (err != OHGPU_OK) return err;
    ctx->stage.src_calls++;
    std::vector<std::pair<uint64_t, uint64_t>> out(n);
    for (size_t i = 0; i < n; i++) out[i] = {descs[i].dst_offset, (uint64_t)descs[i].n_frames * descs[i].channels * (descs[i].dst_bits / 8)};
    err = host_roundtrip(ctx, src_host, src_bytes, dst_host, dst_bytes, out,
                         [&](const void* d_src, void* d_dst) { return ohgpu_src_batch_run(ctx, b, d_src, d_dst, nullptr); });
    ohgpu_batch_destroy(ctx, b);
    return err;
}

int ohgpu_host_transfer_stats(ohgpu_ctx* ctx, uint64_t* calls, uint64_t* src_calls, uint64_t* h2d_bytes, uint64_t* d2h_bytes)
{
    if (!ctx) return set_error(OHGPU_ERR_INVALID, "ohgpu_host_transfer_stats: null context");
    if (calls) *calls = ctx->stage.calls;
    if (src_calls) *src_calls = ctx->stage.src_calls;
    if (h2d_bytes) *h2d_bytes = ctx->stage.h2d_bytes;
    if (d2h_bytes) *d2h_bytes = ctx->stage.d2h_bytes;
    return OHGPU_OK;
}

}  // extern "C"
