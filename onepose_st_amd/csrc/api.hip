// C-ABI plumbing: version, error reporting, device queries.  See include/onepose_hip.h.
#include "tile.h"
#include "onepose_hip.h"
#include <stdio.h>
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
