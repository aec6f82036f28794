// Error reporting, build identification and the optional per-kernel-family HIP-event timer.
#include "common.h"
#include "prof.h"
#include <string.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

int bigc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* bigc_last_error(void) { return g_err; }
extern "C" const char* bigc_build_info(void) { return "bigc_hip gfx950 split-f16x3/f32-mfma " __DATE__ " " __TIME__; }

// ---- profiling ------------------------------------------------------------------------------
namespace {
struct Pending { hipEvent_t a, b; int fam; double flops, bytes; };
struct FamStat { double ms = 0, flops = 0, bytes = 0; long long launches = 0; };
std::mutex g_mu;
bool g_on = false;
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_free;
FamStat g_stat[FAM_COUNT];
const char* g_names[FAM_COUNT] = {"linear_f32", "mha_core", "role_attention", "role_values", "layernorm", "add_rows",
                                  "frontend", "head_gather", "construct_triplets", "grounding", "linear_dw", "backward_misc",
                                  "attention_bwd", "loss", "linear_aux", "pack"};

hipEvent_t take_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

static thread_local double g_hint_flops = 0.0, g_hint_bytes = 0.0;
void bigc_prof_hint(double flops, double bytes) { g_hint_flops = flops; g_hint_bytes = bytes; }

ProfScope::ProfScope(void* stream, int family, double flops, double bytes) : s((hipStream_t)stream), slot(-1) {
    if (flops == 0.0 && bytes == 0.0) { flops = g_hint_flops; bytes = g_hint_bytes; }
    g_hint_flops = g_hint_bytes = 0.0;
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Pending p{take_event(), take_event(), family, flops, bytes};
    (void)hipEventRecord(p.a, s);
    slot = (int)g_pending.size();
    g_pending.push_back(p);
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_pending[slot].b, s);
}

extern "C" void bigc_prof_enable(int on) { std::lock_guard<std::mutex> lk(g_mu); g_on = on != 0; }

extern "C" void bigc_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) { g_free.push_back(p.a); g_free.push_back(p.b); }
    g_pending.clear();
    for (auto& f : g_stat) f = FamStat();
}

// call after the stream has been synchronised: folds the pending event pairs into the per-family totals
extern "C" int bigc_prof_collect(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& p : g_pending) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
        if (e != hipSuccess) return bigc_fail(BIGC_ERR_HIP, "prof_collect: %s", hipGetErrorString(e));
        FamStat& f = g_stat[p.fam];
        f.ms += ms; f.flops += p.flops; f.bytes += p.bytes; f.launches += 1;
        g_free.push_back(p.a); g_free.push_back(p.b);
    }
    g_pending.clear();
    return BIGC_OK;
}

extern "C" int bigc_prof_get(int family, double* ms, long long* launches, double* flops, double* bytes) {
    if (family < 0 || family >= FAM_COUNT) return bigc_fail(BIGC_ERR_ARG, "prof_get: family %d", family);
    std::lock_guard<std::mutex> lk(g_mu);
    if (ms) *ms = g_stat[family].ms;
    if (launches) *launches = g_stat[family].launches;
    if (flops) *flops = g_stat[family].flops;
    if (bytes) *bytes = g_stat[family].bytes;
    return BIGC_OK;
}

extern "C" int bigc_prof_family_count(void) { return FAM_COUNT; }
extern "C" const char* bigc_prof_family_name(int family) { return (family >= 0 && family < FAM_COUNT) ? g_names[family] : ""; }
