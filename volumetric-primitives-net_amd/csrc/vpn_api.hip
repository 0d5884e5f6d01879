// vpn_api.hip — ABI version and error strings of libvpn_hip.so
#include "vpn_common.h"

extern "C" int vpn_abi_version(void) { return VPN_ABI_VERSION; }

extern "C" const char* vpn_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case VPN_E_BADARG: return "vpn: null pointer or non-positive size";
        case VPN_E_TOOBIG: return "vpn: size above a documented limit";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "vpn: unknown error code";
}

// ---- per-kernel timing for bench.py (off by default; not for use under graph capture) ----
#include <string.h>
#include <string>
#include <vector>
#include <map>
namespace vpn {
struct ProfRec { const char* name; hipEvent_t a, b; };
static int g_prof_on = 0;
static std::vector<ProfRec> g_recs;
void prof_begin(const char* name, hipStream_t s) {
    if (!g_prof_on) return;
    ProfRec r; r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
}
void prof_end(hipStream_t s) {
    if (!g_prof_on || g_recs.empty()) return;
    (void)hipEventRecord(g_recs.back().b, s);
}
}  // namespace vpn

extern "C" int vpn_profile_enable(int on) {
    for (auto& r : vpn::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    vpn::g_recs.clear();
    vpn::g_prof_on = on ? 1 : 0;
    return 0;
}

// Synchronises, aggregates by kernel name; writes "name\n" strings into names (NUL terminated), mean
// milliseconds and call counts; returns the number of distinct kernels (or a negative code).
extern "C" int vpn_profile_read(char* names, int names_len, float* mean_ms, int* calls, int max_entries) {
    if (!names || !mean_ms || !calls || names_len <= 0 || max_entries <= 0) return VPN_E_BADARG;
    std::map<std::string, std::pair<double, int>> agg;
    std::vector<std::string> order;
    for (auto& r : vpn::g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        auto it = agg.find(r.name);
        if (it == agg.end()) { agg[r.name] = {ms, 1}; order.push_back(r.name); }
        else { it->second.first += ms; it->second.second += 1; }
    }
    int n = 0, pos = 0;
    for (auto& k : order) {
        if (n >= max_entries || pos + (int)k.size() + 2 > names_len) break;
        memcpy(names + pos, k.c_str(), k.size()); pos += (int)k.size(); names[pos++] = '\n';
        mean_ms[n] = (float)(agg[k].first / agg[k].second); calls[n] = agg[k].second; ++n;
    }
    names[pos] = 0;
    return n;
}
