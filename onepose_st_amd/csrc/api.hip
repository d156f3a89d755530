// C-ABI plumbing: version, error reporting, device queries.  See include/onepose_hip.h.
#include "tile.h"
#include "onepose_hip.h"
#include <atomic>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

int ophip_fail(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: HIP error %d (%s)", where, (int)e, hipGetErrorString(e));
    return (int)e;
}

int ophip_bad_arg(const char* where, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: invalid argument: %s", where, what);
    return -1;
}

// ---- dynamic-LDS limit of a kernel, once per (kernel, device): a process may drive several GPUs ------------------------
#include <map>
#include <mutex>
#include <utility>
int ophip_lds_attr(const void* fn, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return ophip_fail(e, what);
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[{fn, dev}];
    if (bytes <= have) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return ophip_fail(e, what);
    have = bytes;
    return 0;
}

// ---- per-kernel event timing ---------------------------------------------------------------------
namespace {
constexpr int kMaxEvents = 8192;
char g_sel[64] = "";
hipEvent_t* g_start = nullptr;
hipEvent_t* g_stop = nullptr;
int g_used = 0;
int g_dev = -1;              // device the events were created on
int g_every = 1;             // sample: bracket every g_every-th launch of the selected kernel
int g_seen = 0;
}

bool ophip_timed_events(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    if (g_sel[0] == 0 || strcmp(name, g_sel) != 0 || g_used >= kMaxEvents) return false;
    if (g_seen++ % g_every != 0) return false;
    const int slot = g_used++;
    *start = g_start[slot];
    *stop = g_stop[slot];
    return true;
}

extern "C" int ophip_timing_select(const char* kernel_name) {
    if (!kernel_name || strlen(kernel_name) >= sizeof(g_sel)) return ophip_bad_arg(__func__, "kernel name");
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (g_start && kernel_name[0] && dev != g_dev) {        // events belong to a device: re-create them when the caller switched
        for (int i = 0; i < kMaxEvents; ++i) { (void)hipEventDestroy(g_start[i]); (void)hipEventDestroy(g_stop[i]); }
        delete[] g_start; delete[] g_stop;
        g_start = g_stop = nullptr;
    }
    if (!g_start && kernel_name[0]) {
        g_dev = dev;
        g_start = new hipEvent_t[kMaxEvents];
        g_stop = new hipEvent_t[kMaxEvents];
        for (int i = 0; i < kMaxEvents; ++i) {
            hipError_t e = hipEventCreate(&g_start[i]);
            if (e == hipSuccess) e = hipEventCreate(&g_stop[i]);
            if (e != hipSuccess) return ophip_fail(e, __func__);
        }
    }
    strcpy(g_sel, kernel_name);
    g_used = 0;
    g_seen = 0;
    return 0;
}

extern "C" int ophip_timing_every(int n) {
    if (n < 1) return ophip_bad_arg(__func__, "n >= 1");
    g_every = n;
    return 0;
}

extern "C" int ophip_timing_read(int* launches, double* total_ms) {
    double tot = 0.0;
    for (int i = 0; i < g_used; ++i) {
        hipError_t e = hipEventSynchronize(g_stop[i]);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_start[i], g_stop[i]);
        if (e != hipSuccess) return ophip_fail(e, __func__);
        tot += ms;
    }
    if (launches) *launches = g_used;
    if (total_ms) *total_ms = tot;
    g_used = 0;
    return 0;
}

namespace { unsigned long long* g_stamps = nullptr; }
extern "C" unsigned long long* ophip_stamp_buffer(void) { return g_stamps; }
extern "C" int ophip_debug_stamps(void* device_buffer) { g_stamps = reinterpret_cast<unsigned long long*>(device_buffer); return 0; }

// ---- tracing hook: roctx ranges (tile.h OPHIP_LAUNCH, frame.hip) ------------------------------------------------------------------------
namespace {
int (*g_roctx_push)(const char*) = nullptr;
int (*g_roctx_pop)() = nullptr;
int g_roctx_state = -1;              // -1: not decided (OPHIP_ROCTX is read at the first range), 0: off, 1: on
std::atomic<long long> g_roctx_ranges{0};

int roctx_load() {
    if (g_roctx_push) return 0;
    void* lib = nullptr;
    // rocprofv3 (--marker-trace) intercepts the rocprofiler-sdk library's ranges; libroctx64 is the older roctracer one
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return ophip_bad_arg("ophip_roctx_enable", "no roctx library found (libroctx64.so / librocprofiler-sdk-roctx.so)");
    g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    g_roctx_pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    if (!g_roctx_push || !g_roctx_pop) { g_roctx_push = nullptr; return ophip_bad_arg("ophip_roctx_enable", "roctxRangePushA / roctxRangePop not found"); }
    return 0;
}
}  // namespace

extern "C" int ophip_roctx_enable(int on) {
    if (on) { if (int rc = roctx_load()) { g_roctx_state = 0; return rc; } }
    g_roctx_state = on ? 1 : 0;
    return 0;
}
// ranges opened since the library was loaded (tests: the hook really fires on the default path)
extern "C" long long ophip_roctx_ranges(void) { return g_roctx_ranges.load(); }

void ophip_range_push(const char* name) {
    if (g_roctx_state < 0) {
        const char* e = getenv("OPHIP_ROCTX");
        g_roctx_state = (e && e[0] && e[0] != '0' && roctx_load() == 0) ? 1 : 0;
    }
    if (g_roctx_state == 1) { g_roctx_push(name); g_roctx_ranges.fetch_add(1, std::memory_order_relaxed); }
}
void ophip_range_pop() {
    if (g_roctx_state == 1) g_roctx_pop();
}

extern "C" int ophip_abi_version(void) { return OPHIP_ABI_VERSION; }

#include "build/src_hash.h"
// 16 hex digits of the sha256 over the library's sources (csrc/Makefile): which build produced a measurement
extern "C" const char* ophip_build_stamp(void) { return OPHIP_SRC_HASH; }

extern "C" const char* ophip_last_error(void) { return g_err; }

extern "C" int ophip_device_info(int* cu_count, int* lds_per_block, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return ophip_fail(e, __func__);
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return ophip_fail(e, __func__);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_per_block) *lds_per_block = (int)p.sharedMemPerBlock;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return 0;
}
