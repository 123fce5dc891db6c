// Error reporting, device selection and HIP-event profiling shared by the C ABI.
#include "opal_internal.hpp"

#include <cstring>
#include <mutex>
#include <vector>

namespace opal {

static thread_local std::string g_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    // the runtime also keeps the error as the thread's "last error": taken here, or the next launch check
    // (hipGetLastError after a <<<>>>) of a caller that carried on -- with another transport, say -- would report it
    (void)hipGetLastError();
    return OPAL_ERR_HIP;
}

// ---- per-kernel timing with HIP events on the launching stream.  Events are
// recorded around each launch of a tracked kernel while enabled and resolved
// lazily in opal_prof_get (which synchronises).
struct ProfSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
    hipEvent_t open = nullptr;
    uint64_t seen = 0;  // launches observed while enabled (timed or not)
    uint64_t launches = 0;
    double total_ms = 0.0;
};
static ProfSlot g_prof[KERNEL_COUNT];
// launches of the push kernels by the FORM that ran (prof_form), while the timers are on: what bench.py prices
// a launch's algorithmic bytes by
static uint64_t g_prof_forms[PUSH_FORMS];
static int g_prof_stride = 0;  // 0 = off; n = time every n-th launch of each tracked kernel
static std::mutex g_prof_mu;

static hipEvent_t get_event(ProfSlot &s)
{
    if (!s.pool.empty()) {
        hipEvent_t e = s.pool.back();
        s.pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static void resolve(ProfSlot &s)
{
    for (auto &pr : s.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            s.total_ms += ms;
            s.launches++;
        }
        s.pool.push_back(pr.first);
        s.pool.push_back(pr.second);
    }
    s.pending.clear();
}

void prof_begin(int kernel, hipStream_t st)
{
    if (!g_prof_stride) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot &s = g_prof[kernel];
    if (s.seen++ % (uint64_t)g_prof_stride != 0) return;
    s.open = get_event(s);
    if (s.open) hipEventRecord(s.open, st);
}

void prof_form(unsigned form)
{
    if (!g_prof_stride) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_forms[form & (PUSH_FORMS - 1)]++;
}

void prof_end(int kernel, hipStream_t st)
{
    if (!g_prof_stride) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfSlot &s = g_prof[kernel];
    if (!s.open) return;
    hipEvent_t e = get_event(s);
    if (e) {
        hipEventRecord(e, st);
        s.pending.emplace_back(s.open, e);
        if (s.pending.size() >= 4096) resolve(s);
    }
    s.open = nullptr;
}

}  // namespace opal

using namespace opal;

extern "C" {

const char *opal_last_error(void) { return g_error.c_str(); }
const char *opal_version(void) { return "opal_amd 0.1 (gfx950)"; }

int opal_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    if (e != hipSuccess || n == 0) {
        set_error("no HIP device visible (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return OPAL_ERR_HIP;
    }
    return OPAL_OK;
}

int opal_set_device(int device)
{
    OPAL_HIP(hipSetDevice(device));
    return OPAL_OK;
}

int opal_device_identity(int device, char *pci_bus_id, size_t len, char *name, size_t name_len)
{
    if (device < 0) OPAL_HIP(hipGetDevice(&device));
    if (pci_bus_id && len) OPAL_HIP(hipDeviceGetPCIBusId(pci_bus_id, (int)len, device));
    if (name && name_len) {
        hipDeviceProp_t prop;
        OPAL_HIP(hipGetDeviceProperties(&prop, device));
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return OPAL_OK;
}

int opal_stream_create(void **out)
{
    if (!out) {
        set_error("opal_stream_create: null output");
        return OPAL_ERR_ARG;
    }
    hipStream_t st;
    OPAL_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out = st;
    return OPAL_OK;
}

int opal_stream_destroy(void *stream)
{
    if (stream) OPAL_HIP(hipStreamDestroy((hipStream_t)stream));
    return OPAL_OK;
}

int opal_prof_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_stride = on < 0 ? 0 : on;
    return OPAL_OK;
}

int opal_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &s : g_prof) {
        resolve(s);
        s.seen = 0;
        s.launches = 0;
        s.total_ms = 0.0;
    }
    for (auto &f : g_prof_forms) f = 0;
    return OPAL_OK;
}

int opal_prof_forms(uint64_t *launches_by_form)
{
    if (!launches_by_form) {
        set_error("opal_prof_forms: null output");
        return OPAL_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int i = 0; i < PUSH_FORMS; i++) launches_by_form[i] = g_prof_forms[i];
    return OPAL_OK;
}

int opal_prof_seen(int kernel, uint64_t *launches_seen)
{
    if (kernel < 0 || kernel >= KERNEL_COUNT || !launches_seen) {
        set_error("opal_prof_seen: bad arguments");
        return OPAL_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *launches_seen = g_prof[kernel].seen;
    return OPAL_OK;
}

int opal_prof_get(int kernel, uint64_t *launches, double *total_ms)
{
    if (kernel < 0 || kernel >= KERNEL_COUNT) {
        set_error("opal_prof_get: bad kernel id %d", kernel);
        return OPAL_ERR_ARG;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    resolve(g_prof[kernel]);
    if (launches) *launches = g_prof[kernel].launches;
    if (total_ms) *total_ms = g_prof[kernel].total_ms;
    return OPAL_OK;
}

}  // extern "C"
