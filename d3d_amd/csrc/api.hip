// api.hip -- library-wide entry points of the C ABI (include/d3d_hip.h) + the opt-in
// per-kernel HIP-event profiler used by bench.py's roofline leg.
#include "common.hpp"
#include <string.h>
#include <stdio.h>
#include <vector>

int g_d3d_last_hip_error = 0;
int g_d3d_prof_on = 0;

namespace {
struct ProfRec { const char *name; hipEvent_t a, b; };
std::vector<ProfRec> g_recs;
}

void d3d_prof_pre(const char *name, hipStream_t st)
{
    ProfRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
}

void d3d_prof_post(const char *name, hipStream_t st)
{
    if (g_recs.empty() || g_recs.back().name != name) return;
    (void)hipEventRecord(g_recs.back().b, st);
}

extern "C" int d3d_abi_version(void) { return 4; }
extern "C" int d3d_last_hip_error(void) { return g_d3d_last_hip_error; }
extern "C" const char *d3d_status_string(int status)
{
    switch (status) {
    case D3D_OK: return "ok";
    case D3D_ERR_BAD_ARG: return "bad argument";
    case D3D_ERR_UNSUPPORTED: return "unsupported option";
    case D3D_ERR_WORKSPACE: return "workspace too small";
    case D3D_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
    }
}

// When enabled every kernel the library launches is bracketed by a pair of HIP events on the
// launch stream.  d3d_profile_report() synchronises, then writes "name,calls,total_ms\n" lines.
extern "C" int d3d_profile_enable(int on)
{
    g_d3d_prof_on = on ? 1 : 0;
    return D3D_OK;
}

extern "C" int d3d_profile_report(char *buf, size_t buf_bytes)
{
    struct Agg { const char *name; long calls; double ms; };
    std::vector<Agg> aggs;
    for (auto &r : g_recs) {
        float ms = 0.f;
        if (r.a && r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            bool found = false;
            for (auto &a : aggs)
                if (!strcmp(a.name, r.name)) { a.calls++; a.ms += ms; found = true; break; }
            if (!found) aggs.push_back(Agg{r.name, 1, ms});
        }
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
    size_t off = 0;
    if (buf && buf_bytes) buf[0] = 0;
    for (auto &a : aggs) {
        int k = snprintf(buf + off, off < buf_bytes ? buf_bytes - off : 0, "%s,%ld,%.6f\n", a.name, a.calls, a.ms);
        if (k < 0 || off + (size_t)k >= buf_bytes) return D3D_ERR_WORKSPACE;
        off += (size_t)k;
    }
    return D3D_OK;
}
