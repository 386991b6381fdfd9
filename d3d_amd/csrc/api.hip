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

extern "C" int d3d_abi_version(void) { return 13; }   // 13: dtypes D3D_F64_M32 (d3d_iou2d_forward / _backward: fp64 arithmetic, fp32 [n,m] matrix) and D3D_F32_WIDE (d3d_iou2d_forward, d3d_nms2d: fp32 in memory, fp64 arithmetic), D3D_NMS_KEEP_MASK, d3d_score_match_batched; 12: D3DSparseFilterCall, d3d_voxelize_dense_last_plan, D3D_VOXEL_INDEX_V1 retired; 11: point_off of d3d_owner_merge, D3D_VOXEL_INDEX_V1; 10: d3d_voxelize_3d_dense_resident, row_state of d3d_owner_dense; 9: d3d_nms2d_status, flags of d3d_owner_merge / _dense, points of d3d_owner_pack, status word behind d3d_owner_mark_first's bitmap; 8: coord_offset of d3d_voxelize_3d_sparse_filter, D3D_VOXEL_PARTITION_3PASS
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

// ---------------------------------------------------------------------------------------------------------------
// Stream-bandwidth probes for bench.py's roofline legs ("fraction of the MEASURED copy bandwidth on the same box",
// SURVEY 8d): the access patterns of the HBM-bound kernels with the work stripped off.  mode 0: nontemporal 16-byte
// stores over `bytes` (the pattern of k_fill_c4 / the IoU zero fill); mode 1: copy of bytes/2 -> bytes/2
// (16-byte loads + nontemporal stores); mode 2: read-only sweep (the sum lands in the first word, so the loads stay); mode 3:
// hipMemsetAsync; mode 4: nontemporal stores, every wavefront through a 32 KiB stretch of its own (the pattern of k_emit);
// mode 5: nontemporal stores in the launch shape of k_iou_pre's fill (short-lived workgroups, 4 KiB chunks dealt round-robin).
namespace {
typedef float bvec4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_store(bvec4 *p, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bvec4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, &p[i]);
}
// the store pattern of the fused output kernels (k_emit, k_owner_dense): every wavefront streams through a stretch of its
// own (32 KiB here), 1 KiB per store instruction -- "chunked", against the single moving window of the grid-stride form
__global__ __launch_bounds__(256) void k_probe_store_chunked(bvec4 *p, size_t n)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const size_t per = 2048;                                   // 16-byte vectors per stretch
    const bvec4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t s0 = wave * per; s0 < n; s0 += (size_t)gridDim.x * (blockDim.x >> 6) * per)
        for (size_t k = lane; k < per && s0 + k < n; k += 64) __builtin_nontemporal_store(v, &p[s0 + k]);
}
__global__ __launch_bounds__(256) void k_probe_store_dealt(bvec4 *p, size_t nvec)
{
    const size_t nchunk = (nvec + 255) / 256, G = gridDim.x, L = blockIdx.x;
    const bvec4 v = {0.f, 0.f, 0.f, 0.f};
    for (size_t c = L; c < nchunk; c += G) {
        const size_t i = c * 256 + threadIdx.x;
        if (i < nvec) __builtin_nontemporal_store(v, &p[i]);
    }
}
__global__ __launch_bounds__(256) void k_probe_copy(const bvec4 *__restrict__ src, bvec4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
__global__ __launch_bounds__(256) void k_probe_read(const bvec4 *__restrict__ src, size_t n, float *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bvec4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += __builtin_nontemporal_load(&src[i]);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.f;      // never true for the zero-filled probe buffer
}
}  // namespace

extern "C" int d3d_stream_probe(int mode, void *buf, size_t bytes, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (!buf || bytes < 64 || (reinterpret_cast<uintptr_t>(buf) & 15)) return D3D_ERR_BAD_ARG;
    // launch shapes: the best of tools/fill_bench.hip on MI355X (grid-stride over 64 k workgroups: 6.5 TB/s of nt stores on
    // 3 GB; 8 k workgroups reach 5.2)
    const unsigned blocks = 65536;
    if (mode == 0) {
        D3D_LAUNCH("k_probe_store", k_probe_store, dim3(blocks), dim3(256), 0, st, (bvec4 *)buf, bytes / 16);
    } else if (mode == 1) {
        const size_t half = bytes / 32 * 16;
        D3D_LAUNCH("k_probe_copy", k_probe_copy, dim3(blocks), dim3(256), 0, st, (const bvec4 *)buf,
                   (bvec4 *)((char *)buf + half), half / 16);
    } else if (mode == 2) {
        D3D_LAUNCH("k_probe_read", k_probe_read, dim3(blocks), dim3(256), 0, st, (const bvec4 *)buf, bytes / 16, (float *)buf);
    } else if (mode == 3) {
        D3D_HIP_CHECK(hipMemsetAsync(buf, 0, bytes, st));          // the runtime's own fill kernel, for reference
    } else if (mode == 5) {
        // the launch shape of k_iou_pre's fill (box.hip): one short-lived workgroup per 64 x 1024 tile of an n x n fp64 matrix,
        // each writing ~64 of the 4 KiB chunks, dealt L, L + G, ... -- the store part of that kernel with everything else removed
        const size_t nvec = bytes / 16, nchunk = (nvec + 255) / 256;
        size_t g = (nchunk + 63) / 64;
        if (g < 1) g = 1;
        D3D_LAUNCH("k_probe_store_dealt", k_probe_store_dealt, dim3((unsigned)(g < 2147483647u ? g : 2147483647u)), dim3(256), 0, st,
                   (bvec4 *)buf, nvec);
    } else if (mode == 6) {
        // an EMPTY launch: what the event pair around a launch measures by itself -- the record of the first event waits for the
        // launch before it, the kernel is dispatched behind it: d3d_profile_* durations exceed rocprofv3's dispatch timestamps by
        // about this much (bench.py: roofline.event_overhead_us)
        D3D_LAUNCH("k_probe_empty", k_probe_store, dim3(1), dim3(64), 0, st, (bvec4 *)buf, (size_t)0);
    } else if (mode == 4) {
        const size_t nvec = bytes / 16, stretches = (nvec + 2047) / 2048;
        const unsigned b = (unsigned)(stretches / 4 < 65536 ? (stretches + 3) / 4 : 65536);
        D3D_LAUNCH("k_probe_store_chunked", k_probe_store_chunked, dim3(b ? b : 1), dim3(256), 0, st, (bvec4 *)buf, nvec);
    } else return D3D_ERR_BAD_ARG;
    return D3D_OK;
}
