// Error plumbing of the C ABI: one message buffer per calling thread, no exceptions.
#include "equss_common.h"
#include <mutex>
#include <vector>

namespace equss {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- dynamic-LDS opt-in, once per (kernel, device) ---------------------------------------------
namespace {
struct OptIn { const void* kernel; int device; int bytes; };
std::mutex g_optin_mutex;
std::vector<OptIn> g_optin;
}  // namespace

bool lds_opt_in(const void* kernel, int bytes) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_optin_mutex);
    for (OptIn& o : g_optin)
        if (o.kernel == kernel && o.device == device) {
            if (o.bytes >= bytes) return true;
            if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
            o.bytes = bytes;
            return true;
        }
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    g_optin.push_back({kernel, device, bytes});
    return true;
}

// ---- profiling pool (host objects only; no device memory) ------------------------------------
namespace {
struct Pair { hipEvent_t a, b; int cls; };
Pair* g_pairs = nullptr;
int g_cap = 0, g_used = 0;
bool g_on = false;
}  // namespace

bool prof_on() { return g_on && g_used < g_cap; }
void prof_begin(int cls, hipStream_t s) {
    g_pairs[g_used].cls = cls;
    (void)hipEventRecord(g_pairs[g_used].a, s);
}
void prof_end(hipStream_t s) {
    (void)hipEventRecord(g_pairs[g_used].b, s);
    ++g_used;
}

}  // namespace equss

extern "C" int equss_profile_enable(int max_pairs) {
    using namespace equss;
    if (g_pairs) {
        for (int i = 0; i < g_cap; ++i) { (void)hipEventDestroy(g_pairs[i].a); (void)hipEventDestroy(g_pairs[i].b); }
        delete[] g_pairs;
        g_pairs = nullptr;
    }
    g_cap = g_used = 0;
    g_on = false;
    if (max_pairs <= 0) return EQUSS_OK;
    g_pairs = new Pair[max_pairs];
    for (int i = 0; i < max_pairs; ++i) {
        if (hipEventCreate(&g_pairs[i].a) != hipSuccess || hipEventCreate(&g_pairs[i].b) != hipSuccess)
            return fail(EQUSS_E_UNSUPPORTED, "profile_enable: cannot create event pair %d", i);
        g_cap = i + 1;
    }
    g_on = true;
    return EQUSS_OK;
}

extern "C" int equss_profile_read(double* ms, int64_t* launches, int reset) {
    using namespace equss;
    if (!ms || !launches) return fail(EQUSS_E_NULL, "profile_read: null pointer argument");
    for (int i = 0; i < g_used; ++i) {
        hipError_t e = hipEventSynchronize(g_pairs[i].b);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, g_pairs[i].a, g_pairs[i].b);
        if (e != hipSuccess) { fail(0, "profile_read: %s", hipGetErrorString(e)); return -(int)e; }
        ms[g_pairs[i].cls] += (double)t;
        launches[g_pairs[i].cls] += 1;
    }
    if (reset) g_used = 0;
    return EQUSS_OK;
}

extern "C" int equss_abi_version(void) { return EQUSS_ABI_VERSION; }
extern "C" const char* equss_last_error(void) { return equss::error_buffer(); }
