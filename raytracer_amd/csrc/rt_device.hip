// C ABI implementation: contexts, scene upload and kernel launches (see include/rt_amd.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // (types and signatures only: the library is bound at run time, see rccl_load)
#include <dlfcn.h>
#include <type_traits>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <stdexcept>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/rt_amd.h"
#include "host_photon.h"
#include "host_scene.h"
#include "rt_flat.h"
#include "rt_gather_tile.hip.h"
#include "rt_gather_stream.hip.h"
#include "rt_kernels.hip.h"
#include "rt_math.h"
#include "rt_emit.hip.h"
#include "rt_kd.hip.h"
#include "rt_lbvh.hip.h"
#include "rt_render.hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// No exception crosses the C boundary: an entry point's body runs inside this.
template <class F>
int guarded(F&& body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID, "%s", e.what());
    } catch (...) {
        return fail(RT_ERR_INVALID, "unknown exception");
    }
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int reserve(size_t n)
    {
        if (n <= bytes) return RT_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n ? n : 16);
        if (e != hipSuccess) return fail(RT_ERR_HIP, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        bytes = n;
        return RT_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

template <class T>
int upload(DevBuf& b, const std::vector<T>& v, hipStream_t s)
{
    int rc = b.reserve(std::max<size_t>(v.size() * sizeof(T), 64));
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return RT_OK;
}

template <class T>
int upload_n(DevBuf& b, const T* v, size_t n, hipStream_t s)
{
    int rc = b.reserve(std::max<size_t>(n * sizeof(T), 64));
    if (rc) return rc;
    if (n) HIP_TRY(hipMemcpyAsync(b.p, v, n * sizeof(T), hipMemcpyHostToDevice, s));
    return RT_OK;
}

}  // namespace

// ---- working memory of the entry point that allocates nothing (rt_knn_estimate_dev_noalloc) -----------------------
// One fixed array per GPU the library runs on, held until the process ends: [counter block + overflow latch]
// [request queue: RT_NOALLOC_MAX_REQUESTS records] [list of the requests the group kernel hands on] [packed photon
// lists of NA_PACK_WAVES waves] [window parts the group kernel's waves hand to each other].
constexpr size_t NA_COUNTERS = rtk::LATCH_OFFSET + 256;
constexpr size_t NA_QUEUE = (size_t)RT_NOALLOC_MAX_REQUESTS * sizeof(rtg::GatherRec);
constexpr size_t NA_REDO = (size_t)RT_NOALLOC_MAX_REQUESTS * sizeof(uint32_t);
constexpr int NA_PACK_WAVES = 512;      // waves of the group kernel this entry point launches: each needs a packed photon list
constexpr int NA_PACK_CAP = 4096;       // photons per packed list here (the entry points that allocate use rtg::S_PACK_CAP)
constexpr size_t NA_PACK = (size_t)NA_PACK_WAVES * NA_PACK_CAP * sizeof(float4);
constexpr unsigned NA_CONT_ENTRIES = RT_NOALLOC_MAX_REQUESTS / 64;      // window parts handed from wave to wave, per list: one per window (beyond that, rounds stay with their wave)
constexpr size_t NA_CONT = rtg::CONT_SHARDS * (4 + 2 * (size_t)NA_CONT_ENTRIES) * sizeof(unsigned long long);
static_assert(RT_NOALLOC_MAX_REQUESTS % 64 == 0 && NA_QUEUE % 256 == 0, "the queue is whole windows and the list behind it stays aligned");
static_assert(NA_REDO % 256 == 0, "the packed lists behind the hand-over list stay aligned");
static_assert(NA_PACK % 256 == 0, "the list of handed-on parts behind the packed lists stays aligned");
static __device__ __align__(256) unsigned char scratch[NA_COUNTERS + NA_QUEUE + NA_REDO + NA_PACK + NA_CONT];
// `scratch` is a host-side stand-in here, not a device address: host code uses the address hipGetSymbolAddress
// gave for it on the device in question (every GPU holds its own copy of the array at its own address).
constexpr int NA_MAX_DEVICES = 64;
static unsigned char* g_scratch_at[NA_MAX_DEVICES];
// Every call on a GPU uses that GPU's one array, whichever context it comes through.  The calls are therefore put in
// order: each records g_scratch_done[device] behind its kernels and the next one's stream waits for it (events and
// stream waits allocate nothing); g_scratch_mu covers the enqueue, so two host threads cannot interleave theirs.
static hipEvent_t g_scratch_done[NA_MAX_DEVICES];
static bool g_scratch_used[NA_MAX_DEVICES];
static std::mutex g_scratch_mu;

struct rt_scene {
    rth::Scene s;
    std::vector<rth::PhotonRec> photons[2];   // 0 surface, 1 volume
};

// The order in which the camera-ray kernel takes a frame's tiles, chosen by measurement per view (see tile_order_of): which
// of the two orders a launch of this frame used when it was timed, how long it took, and which one has won.
constexpr int TILE_TUNE_ENTRIES = 32, TILE_TUNE_TRIALS = 4, TILE_ORDERS = 3;
struct TileTune {
    uint64_t key = 0;                    // of scene, camera, frame and query
    int decided = 0;                     // 0: still measuring; else the order (1: plain, 2: two strata, 3: two tiles per wave)
    int candidates = 2;                  // orders measured for this view
    int seen = 0;                        // launches of the view so far (the first is not yet a trial: a camera that moves never comes back)
    int started = 0;                     // timed launches begun: launch i runs between ev[2 i] and ev[2 i + 1] in order 1 + i % candidates
    hipEvent_t ev[2 * TILE_ORDERS * TILE_TUNE_TRIALS] = {};
    uint64_t used = 0;                   // for replacement: the least recently used entry goes
};

struct rt_context {
    int device = 0;
    int num_cus = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;         // the per-request gather that runs alongside the group gather
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // order between the two streams
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> chunk_ev;   // begin / end of the trace + shade launches of every row chunk of a timed frame
    // flattened scene in HBM, both instantiations
    DevBuf nodes64, prims64, planes64, nodes32, prims32, planes32, objs, texdesc, texels;
    DevBuf obj_live;          // one byte per object: its hits spawn rays (special > 0)
    rtf::SceneView<double> v64{};
    rtf::SceneView<float> v32{};
    int depth = 0;
    int n_nodes = 0;
    int n_prims = 0;          // primitive records in the tree's leaves
    int n_objects = 0;
    bool has_scene = false;
    bool has_glossy = false;  // some material takes the 16-sample loops of main.cpp:177-257
    bool has_bounce = false;  // some material reflects, refracts or scatters below the surface (special != 0): hits can queue rays
    bool has_sss = false;     // some material scatters below the surface (special < 0)
    DevBuf counters;          // rtk::TraceCounters / rtr::RenderCounters
    DevBuf ctr_log;           // the counter blocks of a frame's row chunks, fetched once at its end
    // photon maps in HBM (0 surface, 1 volume)
    DevBuf ph_pos[2], ph_aux[2], ph_axis[2], ph_order[2], ph_box[2];
    DevBuf ph_src[2];         // photon records (9 doubles each) of a map built on the device, in list order
    int64_t ph_src_n[2] = {0, 0};
    DevBuf ph_raw[2], kd_work;   // emission output before ordering; working set of the device build
    rtg::PhotonMapView pm[2] = {};
    // render queues
    DevBuf hits_a, hits_b, rays, gq, acc, redo;
    DevBuf pack;              // packed photon lists of the group gather's waves
    DevBuf cont;              // window parts the group gather's waves hand to each other (rtg::TileArgs::cont)
    // staging for the host-buffer convenience entry points
    DevBuf st_org, st_dir, st_dist, st_ids, st_aux;
    TileTune tile_tune[TILE_TUNE_ENTRIES];
    uint64_t tile_tune_clock = 0;
    uint64_t scene_generation = 0;       // counts the uploads: a new scene is a new view for tile_tune
};

namespace {

int use_device(rt_context* ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    return RT_OK;
}

// deepest path: `depth` inner nodes, one deferred sibling each
int stack_cap(const rt_context* ctx) { return std::max(1, ctx->depth); }

template <class R>
const rtf::SceneView<R>& view(const rt_context* ctx);
template <> const rtf::SceneView<double>& view<double>(const rt_context* ctx) { return ctx->v64; }
template <> const rtf::SceneView<float>& view<float>(const rt_context* ctx) { return ctx->v32; }

// The context's overflow latch (its stream must be idle): read into v and cleared when set.
int take_overflow_latch(rt_context* ctx, unsigned long long v[2])
{
    char* latch = (char*)ctx->counters.p + rtk::LATCH_OFFSET;
    HIP_TRY(hipMemcpy(v, latch, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (v[0] | v[1]) HIP_TRY(hipMemset(latch, 0, 2 * sizeof(unsigned long long)));
    return RT_OK;
}

// An overflow found in a call's own counters: reported now, so the latch is cleared with it.
int overflow_error(rt_context* ctx, const char* what)
{
    unsigned long long v[2];
    int rc = take_overflow_latch(ctx, v);
    return rc ? rc : fail(RT_ERR_LIMIT, "%s", what);
}

int read_counters(rt_context* ctx, rtk::TraceCounters* out)
{
    HIP_TRY(hipMemcpyAsync(out, ctx->counters.p, sizeof(*out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// A launch with more than 64 KiB of dynamic LDS has to be allowed per kernel and device first.  The call is made once per
// (device, kernel, size reached): between the launches of a frame it would leave the GPU waiting for the host.
int allow_dynamic_lds(const rt_context* ctx, const void* kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return RT_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> allowed;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[{ctx->device, kernel}];
    if (have >= bytes) return RT_OK;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return RT_OK;
}

// Launch geometry of the traversal kernels.  LDS per block = node image (if staged) +
// per-wave stacks; the block size is chosen to keep the most waves resident per CU.
struct TraceGeom { int threads, blocks; size_t lds; int mode; };   // mode: rtk::TRAV_*

// RT_GATHER_FRONT_CAP=<n> (32 <= n <= FRONT) shrinks the gather's frontier so that tests reach its serial walk
int gather_front_cap()
{
    const char* e = getenv("RT_GATHER_FRONT_CAP");
    int v = e ? atoi(e) : rtg::FRONT;
    return std::max(32, std::min(v, rtg::FRONT));
}

template <class R>
int trace_geometry(const rt_context* ctx, int n_nodes, TraceGeom* g, int vgpr_waves = 12)
{
    // Blocks of 4 waves (one 8x8 tile each) unless a deep tree's stacks then leave fewer waves resident
    // than smaller blocks would (LDS is the limit: depth x 64 lanes x bytes per entry, per wave).
    // Deep trees switch to 4-byte stack entries with a box re-test at pop (TRAV_RETEST) when that
    // keeps more waves resident; small trees copy their node array into each block's LDS
    // (TRAV_NODES) when the copy is cheap next to the block's work and leaves room for 3 blocks per CU.
    const int cap = stack_cap(ctx);
    const size_t lds_max = 160 * 1024;
    // vgpr_waves: waves per CU the register budget of the kernel family admits (16 for the render path's kernels,
    // which are held to 128 registers; 20 for the primary-ray kernels at about 100 and 75)
    int best_w = 0, best_resident = 0, best_mode = rtk::TRAV_PLAIN;
    const char* force = getenv("RT_TRAV_MODE");   // tuning aid: "plain" / "retest"
    for (int mode : {rtk::TRAV_PLAIN, rtk::TRAV_RETEST}) {
        if (force && ((force[0] == 'p') != (mode == rtk::TRAV_PLAIN))) continue;
        const size_t per_wave = rtk::stack_lds_bytes(cap, mode == rtk::TRAV_RETEST ? 0 : sizeof(float), 1);
        for (int w : {4, 2, 1}) {
            size_t lds = w * per_wave;
            if (lds > lds_max) continue;
            int resident = std::min<int>((int)(lds_max / lds) * w, vgpr_waves);
            if (resident > best_resident) { best_resident = resident; best_w = w; best_mode = mode; }
        }
    }
    if (const char* fw = getenv("RT_BLOCK_WAVES")) best_w = std::max(1, std::min(12, atoi(fw)));   // tuning aid
    if (!best_w) return fail(RT_ERR_LIMIT, "BVH depth %d: the traversal stack of one wave does not fit 160 KiB of LDS", ctx->depth);
    size_t lds = best_w * rtk::stack_lds_bytes(cap, best_mode == rtk::TRAV_RETEST ? 0 : sizeof(float), 1);
    const size_t node_bytes = rtk::node_lds_bytes<R>((size_t)n_nodes);
    if (best_mode == rtk::TRAV_PLAIN && best_w >= 4 && n_nodes > 0 && node_bytes <= 32 * 1024 && 2 * (lds + node_bytes) <= lds_max && !getenv("RT_NO_NODE_LDS")) {
        best_mode = rtk::TRAV_NODES;
        lds += node_bytes;
    }
    if (getenv("RT_TRACE_GEOM"))
        fprintf(stderr, "trace geometry: %d waves per block, %zu B of LDS (%zu B node image), mode %d, depth %d, up to %d waves per CU by registers\n", best_w, lds,
                best_mode == rtk::TRAV_NODES ? node_bytes : (size_t)0, best_mode, ctx->depth, vgpr_waves);
    *g = TraceGeom{best_w * 64, 0, lds, best_mode};
    return RT_OK;
}

// BVH::Shoot for a batch of rays: the same choice of stack form and node staging as the pixel kernels make.
template <class R, bool TRI_ONLY>
int launch_batch_t(rt_context* ctx, int64_t n, const double* d_org, const double* d_dir, double* d_dist, int32_t* d_ids)
{
    const int cap = stack_cap(ctx);
    TraceGeom g;
    int rc = trace_geometry<R>(ctx, ctx->n_nodes, &g, 16);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, sizeof(rtk::TraceCounters), ctx->stream));
    const int blocks = (int)std::max<int64_t>(1, (n + g.threads - 1) / g.threads);   // one chunk of 64 rays per wave
#define RT_LAUNCH_BATCH(MODE)                                                                                                                       \
    if ((rc = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtk::trace_batch_kernel<R, TRI_ONLY, MODE>), g.lds))) return rc;                      \
    hipLaunchKernelGGL((rtk::trace_batch_kernel<R, TRI_ONLY, MODE>), dim3(blocks), dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), (long long)n, d_org, \
                       d_dir, d_dist, d_ids, (rtk::TraceCounters*)ctx->counters.p, cap, ctx->n_nodes)
    if (g.mode == rtk::TRAV_NODES) { RT_LAUNCH_BATCH(rtk::TRAV_NODES); }
    else if (g.mode == rtk::TRAV_RETEST) { RT_LAUNCH_BATCH(rtk::TRAV_RETEST); }
    else { RT_LAUNCH_BATCH(rtk::TRAV_PLAIN); }
#undef RT_LAUNCH_BATCH
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

template <class R>
int launch_batch(rt_context* ctx, int64_t n, const double* d_org, const double* d_dir, double* d_dist, int32_t* d_ids)
{
    return view<R>(ctx).tri_only ? launch_batch_t<R, true>(ctx, n, d_org, d_dir, d_dist, d_ids) : launch_batch_t<R, false>(ctx, n, d_org, d_dir, d_dist, d_ids);
}

// How long the camera-ray kernel takes depends on which tiles are under way together (rtk::tile_of_wave), by 5-25 % between
// the plain sequence and the one taken from the frame's two halves in turn, and which of the two is the faster depends on
// the view.  So it is measured: once a view -- same scene, camera, frame and query -- comes a second time, its next
// 2 x TILE_TUNE_TRIALS launches alternate between the two orders, each inside its own pair of events; the launches after them take the plain order until
// the last of those has finished (nobody waits: a caller that enqueues ahead of the device just stays with the plain order
// a little longer), and from then on the order with the shortest launch is the view's.  RT_TILE_STRATA=<n> fixes the order
// instead.  The frames are the same either way.
uint64_t tile_view_key(const rt_context* ctx, const rtk::PrimaryArgs& a, int precision)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    mix(&ctx->scene_generation, sizeof ctx->scene_generation);
    mix(a.cam_pos, sizeof a.cam_pos); mix(a.cam_dir, sizeof a.cam_dir); mix(a.cam_right, sizeof a.cam_right); mix(a.cam_down, sizeof a.cam_down);
    mix(a.light, sizeof a.light);
    const int v[] = {a.width, a.height, a.row_begin, a.row_end, a.shadow, a.band_rows, a.n_parts, a.part, precision};
    mix(v, sizeof v);
    return h ? h : 1;
}

// -> the order this launch is to use (1: plain, 2: two strata, 3: two tiles per wave); *ev0 / *ev1: the events it is to run
// between (null: untimed).  `candidates`: the orders worth measuring for this kind of launch (the first `candidates` of them).
int tile_order_of(rt_context* ctx, uint64_t key, int candidates, hipEvent_t* ev0, hipEvent_t* ev1)
{
    *ev0 = *ev1 = nullptr;
    // (a stream under graph capture is not asked about its events, nor given timing events to record)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) return 1;
    TileTune* t = nullptr;
    TileTune* oldest = &ctx->tile_tune[0];
    for (auto& e : ctx->tile_tune) {
        if (e.key == key) { t = &e; break; }
        if (e.used < oldest->used) oldest = &e;
    }
    if (!t) {
        t = oldest;
        // (its events are about to be recorded anew: the launches they belong to have to be past them)
        if (!t->decided && t->started > 0 && hipEventSynchronize(t->ev[2 * t->started - 1]) != hipSuccess) return 1;
        t->key = key; t->decided = 0; t->started = 0; t->seen = 0;
        t->candidates = std::max(2, std::min(candidates, TILE_ORDERS));
    }
    t->used = ++ctx->tile_tune_clock;
    if (t->decided) return t->decided;
    if (t->seen++ == 0) return 1;      // (a view seen once is not worth measuring: frames of a moving camera all take the plain order)
    const int n_timed = t->candidates * TILE_TUNE_TRIALS;
    if (t->started < n_timed) {
        hipEvent_t& a = t->ev[2 * t->started];
        hipEvent_t& b = t->ev[2 * t->started + 1];
        if ((!a && hipEventCreate(&a) != hipSuccess) || (!b && hipEventCreate(&b) != hipSuccess)) return 1;
        *ev0 = a; *ev1 = b;
        return 1 + (t->started++ % t->candidates);
    }
    if (hipEventQuery(t->ev[2 * n_timed - 1]) != hipSuccess) return 1;      // (the timed launches are still under way)
    float best[TILE_ORDERS] = {3.0e38f, 3.0e38f, 3.0e38f};
    for (int i = 0; i < n_timed; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t->ev[2 * i], t->ev[2 * i + 1]) == hipSuccess) best[i % t->candidates] = std::min(best[i % t->candidates], ms);
    }
    t->decided = 1;
    for (int o = 1; o < t->candidates; o++)
        if (best[o] < best[t->decided - 1]) t->decided = o + 1;
    return t->decided;
}

// (`two_tiles`: the kernel form in which a wave takes two tiles -- single precision, node image in LDS; plain otherwise)
template <class R, bool COUNT, bool TRI_ONLY>
int launch_primary_t(rt_context* ctx, const rtk::PrimaryArgs& a, int32_t* d_ids, double* d_t, int32_t* d_shadow, bool want_counts, bool two_tiles)
{
    int cap = stack_cap(ctx);
    TraceGeom g;
    int rc = trace_geometry<R>(ctx, ctx->n_nodes, &g, 20);
    if (rc) return rc;
    // the per-call counters are cleared only for a caller that reads them (an overflow is latched on the device either way)
    if (want_counts) HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, sizeof(rtk::TraceCounters), ctx->stream));
    int wpb = g.threads / 64;
    int blocks = (int)rtk::tile_grid_blocks((a.width + 7) / 8, (a.row_end - a.row_begin + 7) / 8, wpb, g.mode == rtk::TRAV_NODES, 1, a.strata);
#define RT_LAUNCH_PRIMARY(MODE)                                                                                                              \
    if ((rc = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtk::trace_primary_kernel<R, COUNT, TRI_ONLY, MODE>), g.lds))) return rc;     \
    hipLaunchKernelGGL((rtk::trace_primary_kernel<R, COUNT, TRI_ONLY, MODE>), dim3(blocks), dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), a, \
                       d_ids, d_t, d_shadow, (rtk::TraceCounters*)ctx->counters.p, cap, ctx->n_nodes)
    if constexpr (std::is_same<R, float>::value && !COUNT) {
        if (g.mode == rtk::TRAV_NODES && two_tiles) {
            blocks = ((blocks + rtk::N_XCD - 1) / rtk::N_XCD + 1) / 2 * rtk::N_XCD;      // (two places of an XCD's sequence per block)
            if ((rc = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtk::trace_primary_kernel<R, COUNT, TRI_ONLY, rtk::TRAV_NODES, 2>), g.lds))) return rc;
            hipLaunchKernelGGL((rtk::trace_primary_kernel<R, COUNT, TRI_ONLY, rtk::TRAV_NODES, 2>), dim3(blocks), dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), a,
                               d_ids, d_t, d_shadow, (rtk::TraceCounters*)ctx->counters.p, cap, ctx->n_nodes);
            HIP_TRY(hipGetLastError());
            return RT_OK;
        }
    }
    if (g.mode == rtk::TRAV_NODES) { RT_LAUNCH_PRIMARY(rtk::TRAV_NODES); }
    else if (g.mode == rtk::TRAV_RETEST) { RT_LAUNCH_PRIMARY(rtk::TRAV_RETEST); }
    else { RT_LAUNCH_PRIMARY(rtk::TRAV_PLAIN); }
#undef RT_LAUNCH_PRIMARY
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

template <class R, bool COUNT>
int launch_primary(rt_context* ctx, const rtk::PrimaryArgs& a, int32_t* d_ids, double* d_t, int32_t* d_shadow, bool want_counts, bool two_tiles)
{
    return view<R>(ctx).tri_only ? launch_primary_t<R, COUNT, true>(ctx, a, d_ids, d_t, d_shadow, want_counts, two_tiles)
                                 : launch_primary_t<R, COUNT, false>(ctx, a, d_ids, d_t, d_shadow, want_counts, two_tiles);
}

int check_params(const rt_render_params* p)
{
    if (!p) return fail(RT_ERR_INVALID, "params is NULL");
    if (p->struct_size != sizeof(rt_render_params)) return fail(RT_ERR_INVALID, "rt_render_params.struct_size %u != %zu", p->struct_size, sizeof(rt_render_params));
    if (p->width <= 0 || p->height <= 0) return fail(RT_ERR_INVALID, "image size %d x %d", p->width, p->height);
    if (p->precision != RT_F64 && p->precision != RT_F32) return fail(RT_ERR_INVALID, "precision %d", p->precision);
    if (p->band_rows < 0 || (p->band_rows > 0 && (p->n_parts < 1 || p->part < 0 || p->part >= p->n_parts || p->height % (p->band_rows * p->n_parts) != 0)))
        return fail(RT_ERR_INVALID, "row interleave band_rows %d n_parts %d part %d does not tile height %d", p->band_rows, p->n_parts, p->part, p->height);
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void)
{
    return guarded([&]() -> int {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
    });
}

int rt_context_create(int device, rt_context** out)
{
    return guarded([&]() -> int {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(RT_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(RT_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    rt_context* ctx = new rt_context();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(RT_ERR_NO_DEVICE, "cannot create a stream on device %d", device);
    }
    // where this device keeps its copy of the fixed working array (once per device, with that device current)
    std::lock_guard<std::mutex> scratch_lock(g_scratch_mu);
    if (device < NA_MAX_DEVICES && !g_scratch_at[device]) {
        void* at = nullptr;
        hipError_t se = hipGetSymbolAddress(&at, HIP_SYMBOL(scratch));
        if (se != hipSuccess || !at) {
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(se));
        }
        // (the hand-over list is all "empty" between calls: set once here, kept so by the kernels that read it; the
        // memset is waited for, since the contexts' streams do not wait for the null stream)
        if (hipMemset((unsigned char*)at + NA_COUNTERS + NA_QUEUE, 0xFF, NA_REDO) != hipSuccess ||
            hipMemset((unsigned char*)at + NA_COUNTERS + NA_QUEUE + NA_REDO + NA_PACK, 0, NA_CONT) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipEventCreateWithFlags(&g_scratch_done[device], hipEventDisableTiming) != hipSuccess) {
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return fail(RT_ERR_HIP, "setting up the working array on device %d failed", device);
        }
        g_scratch_at[device] = (unsigned char*)at;
    }
    for (auto& ev : ctx->ev)
        if (hipEventCreate(&ev) != hipSuccess) { delete ctx; return fail(RT_ERR_HIP, "hipEventCreate failed"); }
    if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess) {
        rt_context_destroy(ctx);
        return fail(RT_ERR_HIP, "cannot create the second stream of device %d", device);
    }
    // per-call counter block of every kernel family + the overflow latch behind it; never reallocated
    static_assert(sizeof(rtr::RenderCounters) <= rtk::LATCH_OFFSET && sizeof(rte::EmitCounters) <= rtk::LATCH_OFFSET && sizeof(rtk::TraceCounters) <= rtk::LATCH_OFFSET,
                  "the counter blocks end before the overflow latch");
    int rc = ctx->counters.reserve(rtk::LATCH_OFFSET + 64);
    if (rc) { delete ctx; return rc; }
    if (hipMemset(ctx->counters.p, 0, rtk::LATCH_OFFSET + 64) != hipSuccess) { rt_context_destroy(ctx); return fail(RT_ERR_HIP, "hipMemset failed"); }
    *out = ctx;
    return RT_OK;
    });
}

void rt_context_destroy(rt_context* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);
    for (DevBuf* b : {&ctx->nodes64, &ctx->prims64, &ctx->planes64, &ctx->nodes32, &ctx->prims32, &ctx->planes32, &ctx->objs, &ctx->obj_live, &ctx->texdesc,
                      &ctx->texels, &ctx->counters, &ctx->ctr_log, &ctx->st_org, &ctx->st_dir, &ctx->st_dist, &ctx->st_ids, &ctx->st_aux, &ctx->hits_a,
                      &ctx->hits_b, &ctx->rays, &ctx->gq, &ctx->acc, &ctx->redo, &ctx->pack, &ctx->cont, &ctx->ph_pos[0], &ctx->ph_pos[1], &ctx->ph_aux[0], &ctx->ph_aux[1], &ctx->ph_axis[0],
                      &ctx->ph_axis[1], &ctx->ph_order[0], &ctx->ph_order[1], &ctx->ph_box[0], &ctx->ph_box[1], &ctx->ph_src[0], &ctx->ph_src[1],
                      &ctx->ph_raw[0], &ctx->ph_raw[1], &ctx->kd_work})
        b->release();
    for (auto& ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->chunk_ev) (void)hipEventDestroy(ev);
    for (auto& t : ctx->tile_tune)
        for (auto& ev : t.ev)
            if (ev) (void)hipEventDestroy(ev);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void* rt_context_stream(rt_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt_context_synchronize(rt_context* ctx)
{
    return guarded([&]() -> int {
    if (!ctx) return fail(RT_ERR_INVALID, "ctx is NULL");
    int rc = use_device(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long v[2];
    if ((rc = take_overflow_latch(ctx, v))) return rc;
    if (v[0]) return fail(RT_ERR_LIMIT, "traversal stack overflow in an earlier call on this context");
    if (v[1]) return fail(RT_ERR_LIMIT, "bounce/gather/photon queue overflow in an earlier call on this context");
    return RT_OK;
    });
}

void rt_default_params(rt_render_params* p)
{
    memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(*p);
    p->width = 512; p->height = 512;    // main.cpp:606-607
    p->aadepth = 1;                     // main.cpp:610
    p->tracing_depth = 4;               // main.cpp:28
    p->photon_num = 50000;              // main.cpp:21
    p->photon_use = 1000;               // main.cpp:22
    p->head_quirk = 0;
    p->use_photon_map = 1;
    p->precision = RT_F64;
    p->shadow_rays = 0;
    p->accuracy = 0.000001;             // main.cpp:614
    p->photon_dist = 0.28;              // Kdtree.cpp:6
    p->focal_len = 2.6;                 // main.cpp:36
    p->aperture = 0.2;                  // main.cpp:37
    p->dof_samples = 1;                 // main.cpp:35
    p->rng_seed = 12345u;
    p->light_pos[0] = 0.1; p->light_pos[1] = 0.2; p->light_pos[2] = 0;   // main.cpp:654
}

void rt_camera_look_at(const double pos[3], const double look_at[3], rt_camera* out)
{
    // main.cpp:626-635
    rtm::D3 Y = {0, 1, 0};
    rtm::D3 diff = {pos[0] - look_at[0], pos[1] - look_at[1], pos[2] - look_at[2]};
    rtm::D3 dir = rtm::vnorm(rtm::vneg(diff));
    rtm::D3 right = rtm::vnorm(rtm::vcross(Y, dir));
    rtm::D3 down = rtm::vcross(right, dir);
    for (int i = 0; i < 3; i++) out->pos[i] = pos[i];
    out->dir[0] = dir.x; out->dir[1] = dir.y; out->dir[2] = dir.z;
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->down[0] = down.x; out->down[1] = down.y; out->down[2] = down.z;
}

// ---------------------------------------------------------------- host scene
int rt_scene_create(rt_scene** out)
{
    return guarded([&]() -> int {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = new rt_scene();
    return RT_OK;
    });
}
void rt_scene_destroy(rt_scene* s) { delete s; }
int rt_scene_size(const rt_scene* s) { return s ? (int)s->s.objects.size() : 0; }

int rt_scene_add_triangle(rt_scene* s, const double A[3], const double C[3], const double B[3], const double rgbs[4], float ior)
{
    return guarded([&]() -> int {
    if (!s || !A || !B || !C || !rgbs) return fail(RT_ERR_INVALID, "null argument");
    return s->s.add_triangle(A, C, B, rgbs, ior);
    });
}
int rt_scene_set_normals(rt_scene* s, int obj, const double na[3], const double nc[3], const double nb[3])
{
    return guarded([&]() -> int {
    if (!s || !na || !nb || !nc) return fail(RT_ERR_INVALID, "null argument");
    return s->s.set_normals(obj, na, nc, nb) ? fail(RT_ERR_INVALID, "object %d is not a triangle", obj) : RT_OK;
    });
}
int rt_scene_set_texture(rt_scene* s, int obj, const double ta[2], const double tc[2], const double tb[2], int use_texture, int map)
{
    return guarded([&]() -> int {
    if (!s || !ta || !tb || !tc) return fail(RT_ERR_INVALID, "null argument");
    return s->s.set_texture(obj, ta, tc, tb, use_texture, map) ? fail(RT_ERR_INVALID, "object %d is not a triangle or map %d unknown", obj, map)
                                                                 : RT_OK;
    });
}
int rt_scene_add_sphere(rt_scene* s, const double center[3], double radius, const double rgbs[4], float ior)
{
    return guarded([&]() -> int {
    if (!s || !center || !rgbs) return fail(RT_ERR_INVALID, "null argument");
    return s->s.add_sphere(center, radius, rgbs, ior);
    });
}
int rt_scene_add_plane(rt_scene* s, const double normal[3], double distance, const double rgbs[4])
{
    return guarded([&]() -> int {
    if (!s || !normal || !rgbs) return fail(RT_ERR_INVALID, "null argument");
    return s->s.add_plane(normal, distance, rgbs);
    });
}
int rt_scene_add_cornell_box(rt_scene* s, const double corner1[3], const double corner2[3])
{
    return guarded([&]() -> int {
    if (!s || !corner1 || !corner2) return fail(RT_ERR_INVALID, "null argument");
    s->s.add_cornell_box(corner1, corner2);
    return RT_OK;
    });
}
int rt_scene_add_cube(rt_scene* s, const double corner1[3], const double corner2[3], const double rgbs[4])
{
    return guarded([&]() -> int {
    if (!s || !corner1 || !corner2 || !rgbs) return fail(RT_ERR_INVALID, "null argument");
    s->s.add_cube(corner1, corner2, rgbs);
    return RT_OK;
    });
}
int rt_scene_load_obj(rt_scene* s, const char* path, const double rgbs[4], float ior, float xoff, float yoff, float zoff, int use_texture,
                      int map)
{
    return guarded([&]() -> int {
    if (!s || !path || !rgbs) return fail(RT_ERR_INVALID, "null argument");
    int rc = s->s.load_obj(path, rgbs, ior, xoff, yoff, zoff, use_texture, map);
    if (rc == -1) return fail(RT_ERR_IO, "cannot open %s", path);
    if (rc) return fail(RT_ERR_IO, "bad vertex index in %s", path);
    return RT_OK;
    });
}
int rt_scene_add_texture_bmp(rt_scene* s, const char* path)
{
    return guarded([&]() -> int {
    if (!s || !path) return fail(RT_ERR_INVALID, "null argument");
    int rc = s->s.add_texture_bmp(path);
    return rc < 0 ? fail(RT_ERR_IO, "cannot read texture %s", path) : rc;
    });
}
int rt_scene_add_texture_rgb(rt_scene* s, int x_res, int y_res, const double* rgb)
{
    return guarded([&]() -> int {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    int rc = s->s.add_texture_rgb(x_res, y_res, rgb);
    return rc < 0 ? fail(RT_ERR_INVALID, "bad texture") : rc;
    });
}
int rt_scene_build_bvh(rt_scene* s)
{
    return guarded([&]() -> int {
    if (!s) return fail(RT_ERR_INVALID, "null argument");
    s->s.build_bvh();
    return RT_OK;
    });
}
int rt_scene_bvh_dump(const rt_scene* s, int32_t* rec, int64_t* n_rec, double* boxes, int64_t* n_boxes)
{
    return guarded([&]() -> int {
    if (!s || !n_rec || !n_boxes) return fail(RT_ERR_INVALID, "null argument");
    if (!s->s.bvh.built) return fail(RT_ERR_STATE, "rt_scene_build_bvh has not been called");
    std::vector<int32_t> r;
    std::vector<double> b;
    s->s.bvh_dump(r, b);
    if (rec && boxes) {
        if (*n_rec < (int64_t)r.size() || *n_boxes < (int64_t)b.size() / 6) return fail(RT_ERR_INVALID, "buffers too small");
        memcpy(rec, r.data(), r.size() * sizeof(int32_t));
        memcpy(boxes, b.data(), b.size() * sizeof(double));
    }
    *n_rec = (int64_t)r.size();
    *n_boxes = (int64_t)b.size() / 6;
    return RT_OK;
    });
}
int rt_scene_bvh_depth(const rt_scene* s) { return (s && s->s.bvh.built) ? s->s.bvh.depth : -1; }

int rt_scene_get_triangle(const rt_scene* s, int obj, double points[9], double normals[9], double uvs[6], int* flags)
{
    return guarded([&]() -> int {
    if (!s || obj < 0 || obj >= (int)s->s.objects.size()) return fail(RT_ERR_INVALID, "object %d out of range", obj);
    const rth::Object& o = s->s.objects[obj];
    if (o.sh.kind != rtf::PRIM_TRI) return fail(RT_ERR_INVALID, "object %d is not a triangle", obj);
    for (int i = 0; i < 3; i++) {
        if (points) { points[i] = o.sh.A[i]; points[3 + i] = o.sh.B[i]; points[6 + i] = o.sh.C[i]; }
        if (normals) { normals[i] = o.sh.NA[i]; normals[3 + i] = o.sh.NB[i]; normals[6 + i] = o.sh.NC[i]; }
    }
    if (uvs) { uvs[0] = o.sh.tA[0]; uvs[1] = o.sh.tA[1]; uvs[2] = o.sh.tB[0]; uvs[3] = o.sh.tB[1]; uvs[4] = o.sh.tC[0]; uvs[5] = o.sh.tC[1]; }
    if (flags) *flags = (o.sh.has_normal ? 1 : 0) | (o.has_uv ? 2 : 0);
    return RT_OK;
    });
}

int rt_scene_emit_photons(rt_scene* s, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce, double accuracy)
{
    return guarded([&]() -> int {
    if (!s || !light_pos || n_emit < 0) return fail(RT_ERR_INVALID, "bad argument");
    if (!s->s.bvh.built) return fail(RT_ERR_STATE, "rt_scene_build_bvh has not been called");
    rth::EmitParams ep;
    ep.bounce = bounce;
    ep.accuracy = accuracy;
    rth::emit_photons(s->s, n_emit, light_pos, seed, ep, s->photons[0], s->photons[1]);
    return RT_OK;
    });
}
}  // extern "C"
namespace {

// The emission kernel over batches of 2^20 photons.  After each batch, `sink(which, records, count)` is
// called for every non-empty queue shard (device pointer; the stream is idle at that point).
template <class Sink>
int emit_batches(rt_context* ctx, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce, double accuracy, rt_stats* stats, Sink sink)
{
    int rc;
    const int cap = stack_cap(ctx);
    TraceGeom g;
    if ((rc = trace_geometry<double>(ctx, ctx->n_nodes, &g, 12))) return rc;
    const bool tri_only = ctx->v64.tri_only != 0;
    const int64_t batch = 1 << 20;
    // a photon stores at most once per bounce level, and scatters at most 4 volume photons (a subsurface hit ends the walk)
    const unsigned long long cap_s = ((unsigned long long)batch * (bounce + 1) + rtq::SHARDS - 1) / rtq::SHARDS + 16384;
    const unsigned long long cap_v = ((unsigned long long)batch * 4 + rtq::SHARDS - 1) / rtq::SHARDS + 16384;
    if ((rc = ctx->hits_a.reserve(cap_s * rtq::SHARDS * sizeof(rte::PhotonOut)))) return rc;
    if ((rc = ctx->hits_b.reserve(cap_v * rtq::SHARDS * sizeof(rte::PhotonOut)))) return rc;
    // the photons between two bounces: two lists that take turns (the ray and gather queues of the pixel path are idle here)
    if (bounce < 0 || bounce >= rte::MAX_EMIT_LEVELS) return fail(RT_ERR_LIMIT, "bounce limit %d outside [0,%d)", bounce, rte::MAX_EMIT_LEVELS);
    if ((rc = ctx->rays.reserve((size_t)batch * sizeof(rte::PhotonState)))) return rc;
    if ((rc = ctx->gq.reserve((size_t)batch * sizeof(rte::PhotonState)))) return rc;
    rte::ShadeTables sd{(const rtf::ObjShade*)ctx->objs.p, (const rtf::TexDesc*)ctx->texdesc.p, (const double*)ctx->texels.p};
    rte::EmitCounters* ctr = (rte::EmitCounters*)ctx->counters.p;
    uint64_t rays = 0;
    float ms_total = 0;
    for (int64_t first = 0; first < n_emit; first += batch) {
        rte::EmitArgs a;
        for (int i = 0; i < 3; i++) a.light[i] = light_pos[i];
        a.n_emit = std::min(batch, n_emit - first);
        a.first = first;
        a.seed = seed;
        a.bounce = bounce;
        a.accuracy = accuracy;
        HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(rte::EmitCounters), ctx->stream));
        HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        // the photons of the launch in the order of their starting directions (which photons share a wave at the first bounce)
        const uint32_t* order = nullptr;
        if (a.n_emit >= 4096 && !getenv("RT_EMIT_UNORDERED")) {
            const int ne = (int)a.n_emit;
            const size_t a4 = ((size_t)ne * 4 + 255) & ~(size_t)255;
            size_t sort_bytes = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ne, 0,
                                                       rte::EMIT_ORDER_KEY_BITS, ctx->stream));
            if ((rc = ctx->kd_work.reserve(4 * a4 + sort_bytes + 256))) return rc;
            char* w = (char*)ctx->kd_work.p;
            uint32_t *k0 = (uint32_t*)w, *k1 = (uint32_t*)(w + a4), *v0 = (uint32_t*)(w + 2 * a4), *v1 = (uint32_t*)(w + 3 * a4);
            hipLaunchKernelGGL(rte::emit_order_keys_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, a, k0, v0);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w + 4 * a4, sort_bytes, k0, k1, v0, v1, ne, 0, rte::EMIT_ORDER_KEY_BITS, ctx->stream));
            order = v1;
        }
        // RT_EMIT_RESORT=1: the survivors of every bounce ordered by direction and starting point before the next (comparison runs)
        const bool resort = getenv("RT_EMIT_RESORT") != nullptr && a.n_emit >= 4096;
        uint32_t *sk0 = nullptr, *sk1 = nullptr, *sv0 = nullptr, *sv1 = nullptr;
        void* s_tmp = nullptr;
        size_t s_bytes = 0;
        if (resort) {
            const int ne = (int)a.n_emit;
            const size_t a4 = ((size_t)ne * 4 + 255) & ~(size_t)255;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, s_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, ne, 0,
                                                       rte::SURVIVOR_KEY_BITS, ctx->stream));
            if ((rc = ctx->st_aux.reserve(4 * a4 + s_bytes + 256))) return rc;
            char* w = (char*)ctx->st_aux.p;
            sk0 = (uint32_t*)w; sk1 = (uint32_t*)(w + a4); sv0 = (uint32_t*)(w + 2 * a4); sv1 = (uint32_t*)(w + 3 * a4);
            s_tmp = w + 4 * a4;
        }
        const uint32_t* from_order = nullptr;
        // one launch per bounce; a later bounce's grid is sized for all photons and the lanes beyond the survivors leave at once
        rte::PhotonState* lists[2] = {(rte::PhotonState*)ctx->rays.p, (rte::PhotonState*)ctx->gq.p};
        for (int b = 0; b <= bounce; b++) {
            const bool key_next = resort && b < bounce;
            if (key_next) HIP_TRY(hipMemsetAsync(sk0, 0xFF, (size_t)a.n_emit * 4, ctx->stream));
#define RT_LAUNCH_EMIT(TRI, MODE)                                                                                                                       \
    if ((rc = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rte::emit_kernel<double, TRI, MODE>), g.lds))) return rc;                               \
    hipLaunchKernelGGL((rte::emit_kernel<double, TRI, MODE>), dim3((unsigned)((a.n_emit + g.threads - 1) / g.threads)), dim3(g.threads), g.lds, ctx->stream, \
                       ctx->v64, sd, a, b, (const rte::PhotonState*)lists[b & 1], lists[(b + 1) & 1], (rte::PhotonOut*)ctx->hits_a.p, cap_s,               \
                       (rte::PhotonOut*)ctx->hits_b.p, cap_v, ctr, cap, ctx->n_nodes, order, from_order, key_next ? sk0 : nullptr, key_next ? sv0 : nullptr)
            if (tri_only) {
                if (g.mode == rtk::TRAV_NODES) { RT_LAUNCH_EMIT(true, rtk::TRAV_NODES); }
                else if (g.mode == rtk::TRAV_RETEST) { RT_LAUNCH_EMIT(true, rtk::TRAV_RETEST); }
                else { RT_LAUNCH_EMIT(true, rtk::TRAV_PLAIN); }
            } else {
                if (g.mode == rtk::TRAV_NODES) { RT_LAUNCH_EMIT(false, rtk::TRAV_NODES); }
                else if (g.mode == rtk::TRAV_RETEST) { RT_LAUNCH_EMIT(false, rtk::TRAV_RETEST); }
                else { RT_LAUNCH_EMIT(false, rtk::TRAV_PLAIN); }
            }
#undef RT_LAUNCH_EMIT
            HIP_TRY(hipGetLastError());
            if (key_next) {
                HIP_TRY(hipcub::DeviceRadixSort::SortPairs(s_tmp, s_bytes, sk0, sk1, sv0, sv1, (int)a.n_emit, 0, rte::SURVIVOR_KEY_BITS, ctx->stream));
                from_order = sv1;
            }
        }
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        rte::EmitCounters hc;
        HIP_TRY(hipMemcpyAsync(&hc, ctr, sizeof hc, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (hc.overflow_stack) return overflow_error(ctx, "traversal stack overflow");
        if (hc.overflow_queue) return overflow_error(ctx, "photon queue overflow");
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        ms_total += ms;
        rays += hc.rays;
        for (int which = 0; which < 2; which++) {
            const rtq::Counts& c = which ? hc.volume : hc.surface;
            const unsigned long long shard_cap = which ? cap_v : cap_s;
            const rte::PhotonOut* dev = (const rte::PhotonOut*)(which ? ctx->hits_b.p : ctx->hits_a.p);
            for (int sh = 0; sh < rtq::SHARDS; sh++) {
                size_t n = (size_t)std::min<unsigned long long>(c.n[sh][0], shard_cap);
                if (n && (rc = sink(which, dev + (size_t)sh * shard_cap, n))) return rc;
            }
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->secondary_rays = rays;
        stats->kernel_ms = stats->trace_kernel_ms = ms_total;
    }
    return RT_OK;
}

// keeps the first `used` bytes across a growth of the buffer
int grow_keep(DevBuf& b, size_t need, size_t used, hipStream_t stream)
{
    if (need <= b.bytes) return RT_OK;
    DevBuf bigger;
    int rc = bigger.reserve(std::max(need, b.bytes * 2));
    if (rc) return rc;
    if (used) {
        HIP_TRY(hipMemcpyAsync(bigger.p, b.p, used, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    b.release();
    b = bigger;
    return RT_OK;
}

// Photon map `which` from n records on the device (9 doubles every `stride` doubles; must not live in
// ctx->kd_work).  Kdtree.cpp:11-52 restated level by level, see rt_kd.hip.h.
int build_photon_map_dev(rt_context* ctx, int which, long long n, const double* d_rec, int stride)
{
    int rc;
    if (n >= (1ll << 31)) return fail(RT_ERR_LIMIT, "photon count %lld exceeds 2^31", n);
    rtg::PhotonMapView& pm = ctx->pm[which];
    if ((rc = ctx->ph_pos[which].reserve(std::max<size_t>(64, (size_t)n * 16)))) return rc;
    if ((rc = ctx->ph_aux[which].reserve(std::max<size_t>(64, (size_t)n * 24)))) return rc;
    if ((rc = ctx->ph_order[which].reserve(std::max<size_t>(64, (size_t)n * 4)))) return rc;
    if ((rc = ctx->ph_box[which].reserve((size_t)(n / 32 + 1) * 24))) return rc;
    pm.pos = (const float4*)ctx->ph_pos[which].p;
    pm.aux = (const float*)ctx->ph_aux[which].p;
    pm.order = (const int*)ctx->ph_order[which].p;
    pm.leaf_box = (const float*)ctx->ph_box[which].p;
    pm.n = n;
    if (n == 0) return RT_OK;
    const int ni = (int)n;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t at = off; off += align(bytes); return at; };
    // (bounds of the ranges of every level behind one another: level l starts at 3 (2^l - 1))
    const int levels = rtkd::levels_for(n);
    const size_t n_bounds = ((size_t)3 << levels) + 3;
    const size_t o_k0 = carve(n * 8), o_k1 = carve(n * 8), o_v0 = carve(n * 4), o_v1 = carve(n * 4), o_ax = carve(n), o_mn = carve(n_bounds * 4),
                 o_mx = carve(n_bounds * 4), o_xyz = carve(n * 16);
    size_t sort_bytes = 0;
    int key_bits = 33;
    while (key_bits < 64 && (n >> (key_bits - 32)) != 0) key_bits++;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                               (unsigned*)nullptr, ni, 0, key_bits, ctx->stream));
    const size_t o_tmp = carve(sort_bytes + 256);
    if ((rc = ctx->kd_work.reserve(off))) return rc;
    char* w = (char*)ctx->kd_work.p;
    auto* k0 = (unsigned long long*)(w + o_k0); auto* k1 = (unsigned long long*)(w + o_k1);
    auto* v0 = (unsigned*)(w + o_v0); auto* v1 = (unsigned*)(w + o_v1);
    auto* axis_at = (unsigned char*)(w + o_ax); auto* mins = (unsigned*)(w + o_mn); auto* maxs = (unsigned*)(w + o_mx);
    const int T = 256;
    const dim3 grid((unsigned)((n + T - 1) / T));
    rtkd::Source src{d_rec, stride, (const float4*)(w + o_xyz)};
    hipLaunchKernelGGL(rtkd::compact_kernel, grid, dim3(T), 0, ctx->stream, d_rec, stride, ni, (float4*)(w + o_xyz));
    hipLaunchKernelGGL(rtkd::iota_kernel, grid, dim3(T), 0, ctx->stream, ni, v0);
    HIP_TRY(hipMemsetAsync(axis_at, 0, (size_t)n, ctx->stream));
    HIP_TRY(hipMemsetAsync(mins, 0xFF, (o_mx - o_mn) + n_bounds * 4, ctx->stream));      // both arrays, every level (they lie behind one another)
    for (int level = 0; level < levels; level++) {
        const size_t at = (size_t)3 * (((size_t)1 << level) - 1);
        const long long per_block = (long long)T * rtkd::run_of_level(level);
        hipLaunchKernelGGL(rtkd::bounds_kernel, dim3((unsigned)((n + per_block - 1) / per_block)), dim3(T), 0, ctx->stream, src, ni, level,
                           (const unsigned*)v0, mins + at, maxs + at);
        hipLaunchKernelGGL(rtkd::key_kernel, grid, dim3(T), 0, ctx->stream, src, ni, level, (const unsigned*)v0, (const unsigned*)(mins + at),
                           (const unsigned*)(maxs + at), k0, axis_at);
        // (the key's high part is the segment's ordinal among the level's: level + 2 bits)
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w + o_tmp, sort_bytes, k0, k1, v0, v1, ni, 0, std::min(key_bits, rtkd::key_bits_of_level(level)), ctx->stream));
        std::swap(v0, v1);
    }
    hipLaunchKernelGGL(rtkd::emit_kernel, grid, dim3(T), 0, ctx->stream, src, ni, (const unsigned*)v0, (const unsigned char*)axis_at,
                       (float4*)ctx->ph_pos[which].p, (float*)ctx->ph_aux[which].p, (int*)ctx->ph_order[which].p);
    hipLaunchKernelGGL(rtkd::leaf_box_kernel, grid, dim3(T), 0, ctx->stream, ni, levels, (const float4*)ctx->ph_pos[which].p, (float*)ctx->ph_box[which].p);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // namespace
extern "C" {

int rt_scene_emit_photons_gpu(rt_context* ctx, rt_scene* s, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce, double accuracy,
                              rt_stats* stats)
{
    return guarded([&]() -> int {
    if (!ctx || !s || !light_pos || n_emit < 0 || bounce < 0 || bounce > 7) return fail(RT_ERR_INVALID, "bad argument");
    if (!ctx->has_scene) return fail(RT_ERR_STATE, "no scene uploaded");
    if ((int64_t)s->s.objects.size() != ctx->n_objects) return fail(RT_ERR_STATE, "the scene is not the one uploaded to this context");
    int rc;
    if ((rc = use_device(ctx))) return rc;
    std::vector<rte::PhotonOut> host[2];
    rc = emit_batches(ctx, n_emit, light_pos, seed, bounce, accuracy, stats, [&](int which, const rte::PhotonOut* dev, size_t n) {
        size_t at = host[which].size();
        host[which].resize(at + n);
        HIP_TRY(hipMemcpyAsync(host[which].data() + at, dev, n * sizeof(rte::PhotonOut), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));     // the vector may move at the next call
        return (int)RT_OK;
    });
    if (rc) return rc;
    for (int which = 0; which < 2; which++) {
        std::vector<uint32_t> order(host[which].size());
        for (size_t i = 0; i < order.size(); i++) order[i] = (uint32_t)i;
        const rte::PhotonOut* h = host[which].data();
        std::sort(order.begin(), order.end(), [h](uint32_t x, uint32_t y) { return h[x].key < h[y].key; });
        auto& out = s->photons[which];
        out.reserve(out.size() + order.size());
        for (uint32_t i : order) {
            rth::PhotonRec r;
            memcpy(&r, &h[i], sizeof r);
            out.push_back(r);
        }
    }
    return RT_OK;
    });
}

int rt_emit_photon_maps_dev(rt_context* ctx, int64_t n_emit, const double light_pos[3], uint32_t seed, int bounce, double accuracy, int64_t counts[2],
                            rt_stats* stats)
{
    return guarded([&]() -> int {
    if (!ctx || !light_pos || n_emit < 0 || bounce < 0 || bounce > 7) return fail(RT_ERR_INVALID, "bad argument");
    if (!ctx->has_scene) return fail(RT_ERR_STATE, "no scene uploaded");
    int rc;
    if ((rc = use_device(ctx))) return rc;
    static_assert(sizeof(rte::PhotonOut) == 80, "PhotonOut is 9 doubles and a key");
    size_t got[2] = {0, 0};
    rt_stats est;
    rc = emit_batches(ctx, n_emit, light_pos, seed, bounce, accuracy, &est, [&](int which, const rte::PhotonOut* dev, size_t n) {
        int r = grow_keep(ctx->ph_raw[which], (got[which] + n) * sizeof(rte::PhotonOut), got[which] * sizeof(rte::PhotonOut), ctx->stream);
        if (r) return r;
        HIP_TRY(hipMemcpyAsync((rte::PhotonOut*)ctx->ph_raw[which].p + got[which], dev, n * sizeof(rte::PhotonOut), hipMemcpyDeviceToDevice, ctx->stream));
        got[which] += n;
        return (int)RT_OK;
    });
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
    for (int which = 0; which < 2; which++) {
        const long long n = (long long)got[which];
        if (n >= (1ll << 31)) return fail(RT_ERR_LIMIT, "photon count %lld exceeds 2^31", n);
        if ((rc = ctx->ph_src[which].reserve(std::max<size_t>(64, (size_t)n * 72)))) return rc;
        ctx->ph_src_n[which] = n;
        if (n > 0) {
            // launch-independent order: by (photon, bounce, slot)
            const int ni = (int)n;
            size_t sort_bytes = 0;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                                       (unsigned*)nullptr, ni, 0, 64, ctx->stream));
            const size_t a8 = ((size_t)n * 8 + 255) & ~(size_t)255, a4 = ((size_t)n * 4 + 255) & ~(size_t)255;
            if ((rc = ctx->kd_work.reserve(2 * a8 + 2 * a4 + sort_bytes + 256))) return rc;
            char* w = (char*)ctx->kd_work.p;
            auto* k0 = (unsigned long long*)w; auto* k1 = (unsigned long long*)(w + a8);
            auto* v0 = (unsigned*)(w + 2 * a8); auto* v1 = (unsigned*)(w + 2 * a8 + a4);
            const int T = 256;
            const double* raw = (const double*)ctx->ph_raw[which].p;
            hipLaunchKernelGGL(rtkd::gather_keys_kernel, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, ctx->stream, ni, raw, 10, 9, k0, v0);
            // (keys are photon number x 64 + bounce x 8 + slot: the bits of the launch's photon numbers and six more)
            int order_bits = 7;
            while (order_bits < 64 && ((unsigned long long)n_emit >> (order_bits - 6)) != 0) order_bits++;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w + 2 * a8 + 2 * a4, sort_bytes, k0, k1, v0, v1, ni, 0, order_bits, ctx->stream));
            hipLaunchKernelGGL(rtkd::permute_kernel, dim3((unsigned)((n * 9 + T - 1) / T)), dim3(T), 0, ctx->stream, ni, (const unsigned*)v1, raw, 10,
                               (double*)ctx->ph_src[which].p);
            HIP_TRY(hipGetLastError());
        }
        if ((rc = build_photon_map_dev(ctx, which, n, (const double*)ctx->ph_src[which].p, 9))) return rc;
        if (counts) counts[which] = n;
    }
    HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (stats) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
        *stats = est;
        stats->kernel_ms = est.trace_kernel_ms + ms;     // emission + ordering + both tree builds
    }
    return RT_OK;
    });
}

int rt_photons_upload_dev_build(rt_context* ctx, int which, int64_t n, const double* photons)
{
    return guarded([&]() -> int {
    if (!ctx || (which != 0 && which != 1) || n < 0 || (n > 0 && !photons)) return fail(RT_ERR_INVALID, "bad argument");
    if (n >= (1ll << 31)) return fail(RT_ERR_LIMIT, "photon count %lld exceeds 2^31", (long long)n);
    int rc = use_device(ctx);
    if (rc) return rc;
    if ((rc = ctx->ph_src[which].reserve(std::max<size_t>(64, (size_t)n * 72)))) return rc;
    ctx->ph_src_n[which] = n;
    if (n) HIP_TRY(hipMemcpyAsync(ctx->ph_src[which].p, photons, (size_t)n * 72, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = build_photon_map_dev(ctx, which, n, (const double*)ctx->ph_src[which].p, 9))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));     // the caller's buffer is free again
    return RT_OK;
    });
}

int64_t rt_photon_map_size(const rt_context* ctx, int which) { return (ctx && (which == 0 || which == 1)) ? (int64_t)ctx->pm[which].n : -1; }

int rt_photon_map_get(rt_context* ctx, int which, double* photons)
{
    return guarded([&]() -> int {
    if (!ctx || (which != 0 && which != 1) || !photons) return fail(RT_ERR_INVALID, "bad argument");
    if (ctx->ph_src_n[which] != ctx->pm[which].n) return fail(RT_ERR_STATE, "photon map %d was not built on the device", which);
    int rc = use_device(ctx);
    if (rc) return rc;
    if (ctx->ph_src_n[which])
        HIP_TRY(hipMemcpyAsync(photons, ctx->ph_src[which].p, (size_t)ctx->ph_src_n[which] * 72, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
    });
}

int64_t rt_scene_photon_count(const rt_scene* s, int which) { return (s && (which == 0 || which == 1)) ? (int64_t)s->photons[which].size() : -1; }
int rt_scene_photons_get(const rt_scene* s, int which, double* photons)
{
    return guarded([&]() -> int {
    if (!s || !photons || (which != 0 && which != 1)) return fail(RT_ERR_INVALID, "bad argument");
    static_assert(sizeof(rth::PhotonRec) == 9 * sizeof(double), "PhotonRec is 9 doubles");
    memcpy(photons, s->photons[which].data(), s->photons[which].size() * sizeof(rth::PhotonRec));
    return RT_OK;
    });
}
void rt_scene_photons_clear(rt_scene* s)
{
    if (!s) return;
    s->photons[0].clear();
    s->photons[1].clear();
}

// ---------------------------------------------------------------- device scene
}  // extern "C"
namespace {
// per-object shading records and the texel pool (synchronises: the caller's arrays may die at return)
int upload_shading(rt_context* ctx, const rtf::ObjShade* sh, size_t n_objects, const rtf::TexDesc* td, size_t n_textures, const double* texels, size_t n_texel_doubles)
{
    int rc;
    std::vector<unsigned char> live(n_objects);
    ctx->has_glossy = ctx->has_bounce = ctx->has_sss = false;
    for (size_t i = 0; i < n_objects; i++) {
        if (sh[i].color[3] != 0.0) ctx->has_bounce = true;
        if (sh[i].color[3] < 0.0) ctx->has_sss = true;
        if (rtm::glossy_special(sh[i].color[3])) ctx->has_glossy = true;
        // (a textured triangle takes its colour from the map, whose `special` is 0: see rtm::obj_color)
        const bool textured = sh[i].kind == rtf::PRIM_TRI && sh[i].has_texture && sh[i].map >= 0;
        live[i] = sh[i].color[3] > 0 && !textured ? 1 : 0;
    }
    if ((rc = upload_n(ctx->objs, sh, n_objects, ctx->stream))) return rc;
    if ((rc = upload(ctx->obj_live, live, ctx->stream))) return rc;
    if ((rc = upload_n(ctx->texdesc, td, n_textures, ctx->stream))) return rc;
    if ((rc = upload_n(ctx->texels, texels, n_texel_doubles, ctx->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int upload_shading(rt_context* ctx, const rth::Scene& s)
{
    std::vector<rtf::ObjShade> sh(s.objects.size());
    for (size_t i = 0; i < sh.size(); i++) sh[i] = s.objects[i].sh;
    std::vector<rtf::TexDesc> td;
    std::vector<double> pool;
    for (const auto& t : s.textures) {
        td.push_back({t.x_res, t.y_res, (int64_t)pool.size()});
        pool.insert(pool.end(), t.rgb.begin(), t.rgb.end());
    }
    return upload_shading(ctx, sh.data(), sh.size(), td.data(), td.size(), pool.data(), pool.size());
}

// The host tree and its primitive records, in both precisions, to HBM; the context's views follow.
struct HostTree {
    const rtf::Node<double>* nodes; size_t n_nodes;
    const rtf::Prim<double>* prims; size_t n_prims;
    const rtf::Prim<double>* planes; size_t n_planes;
    double root_lo[3], root_hi[3];
    uint32_t root_ref;
    int depth;
};

int upload_tree(rt_context* ctx, const HostTree& t, size_t n_objects)
{
    int rc;
    auto n32 = rth::to_f32(t.nodes, t.n_nodes);
    auto p32 = rth::to_f32(t.prims, t.n_prims);
    auto pl32 = rth::to_f32(t.planes, t.n_planes);
    if ((rc = upload_n(ctx->nodes64, t.nodes, t.n_nodes, ctx->stream))) return rc;
    if ((rc = upload_n(ctx->prims64, t.prims, t.n_prims, ctx->stream))) return rc;
    if ((rc = upload_n(ctx->planes64, t.planes, t.n_planes, ctx->stream))) return rc;
    if ((rc = upload(ctx->nodes32, n32, ctx->stream))) return rc;
    if ((rc = upload(ctx->prims32, p32, ctx->stream))) return rc;
    if ((rc = upload(ctx->planes32, pl32, ctx->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // the float copies above die at return
    ctx->v64.nodes = (const rtf::Node<double>*)ctx->nodes64.p;
    ctx->v64.prims = (const rtf::Prim<double>*)ctx->prims64.p;
    ctx->v64.planes = (const rtf::Prim<double>*)ctx->planes64.p;
    ctx->v32.nodes = (const rtf::Node<float>*)ctx->nodes32.p;
    ctx->v32.prims = (const rtf::Prim<float>*)ctx->prims32.p;
    ctx->v32.planes = (const rtf::Prim<float>*)ctx->planes32.p;
    for (int i = 0; i < 3; i++) {
        ctx->v64.root_lo[i] = t.root_lo[i]; ctx->v64.root_hi[i] = t.root_hi[i];
        ctx->v32.root_lo[i] = (float)t.root_lo[i]; ctx->v32.root_hi[i] = (float)t.root_hi[i];
    }
    ctx->v64.root_ref = ctx->v32.root_ref = t.root_ref;
    ctx->v64.n_planes = ctx->v32.n_planes = (int32_t)t.n_planes;
    bool tri_only = true;
    for (size_t i = 0; i < t.n_prims; i++) tri_only = tri_only && t.prims[i].kind == rtf::PRIM_TRI;
    ctx->v64.tri_only = ctx->v32.tri_only = tri_only ? 1 : 0;
    ctx->depth = t.depth;
    ctx->n_nodes = (int)t.n_nodes;
    ctx->n_prims = (int)t.n_prims;
    ctx->n_objects = (int)n_objects;
    ctx->has_scene = true;
    ctx->scene_generation++;
    return RT_OK;
}
}  // namespace
extern "C" {

int rt_scene_upload(rt_context* ctx, const rt_scene* scene)
{
    return guarded([&]() -> int {
    if (!ctx || !scene) return fail(RT_ERR_INVALID, "null argument");
    const rth::Scene& s = scene->s;
    if (!s.bvh.built) return fail(RT_ERR_STATE, "rt_scene_build_bvh has not been called");
    int rc = use_device(ctx);
    if (rc) return rc;
    const rth::FlatBVH& b = s.bvh;
    if ((rc = upload_shading(ctx, s))) return rc;
    HostTree t{b.nodes.data(), b.nodes.size(), b.prims.data(), b.prims.size(), b.planes.data(), b.planes.size(), {}, {}, b.root_ref, b.depth};
    for (int i = 0; i < 3; i++) { t.root_lo[i] = b.root_box.lo[i]; t.root_hi[i] = b.root_box.hi[i]; }
    return upload_tree(ctx, t, s.objects.size());
    });
}

int rt_scene_export_flat(const rt_scene* scene, void* buf, size_t* bytes)
{
    return guarded([&]() -> int {
    if (!scene || !bytes) return fail(RT_ERR_INVALID, "null argument");
    if (!scene->s.bvh.built) return fail(RT_ERR_STATE, "rt_scene_build_bvh has not been called");
    const size_t need = rth::flat_image_bytes(scene->s);
    if (buf) {
        if (*bytes < need) return fail(RT_ERR_INVALID, "buffer of %zu bytes, the image takes %zu", *bytes, need);
        if ((uintptr_t)buf % 64 != 0) return fail(RT_ERR_INVALID, "the buffer is not 64-byte aligned");
        rth::flat_image_write(scene->s, buf);
    }
    *bytes = need;
    return RT_OK;
    });
}

int rt_scene_import_flat(rt_scene* scene, const void* buf, size_t bytes)
{
    return guarded([&]() -> int {
    if (!scene || !buf) return fail(RT_ERR_INVALID, "null argument");
    rth::FlatImageView v;
    std::string why;
    if (rth::flat_image_open(buf, bytes, &v, &why)) return fail(RT_ERR_INVALID, "%s", why.c_str());
    rth::flat_image_read(v, scene->s);
    scene->photons[0].clear();
    scene->photons[1].clear();
    return RT_OK;
    });
}

int rt_scene_upload_flat(rt_context* ctx, const void* buf, size_t bytes)
{
    return guarded([&]() -> int {
    if (!ctx || !buf) return fail(RT_ERR_INVALID, "null argument");
    rth::FlatImageView v;
    std::string why;
    if (rth::flat_image_open(buf, bytes, &v, &why)) return fail(RT_ERR_INVALID, "%s", why.c_str());
    int rc = use_device(ctx);
    if (rc) return rc;
    const rth::FlatImageHeader& h = *v.h;
    if ((rc = upload_shading(ctx, v.objects, h.n_objects, v.texdesc, h.n_textures, v.texels, h.n_texel_doubles))) return rc;
    HostTree t{v.nodes, (size_t)h.n_nodes, v.prims, (size_t)h.n_prims, v.planes, (size_t)h.n_planes, {}, {}, h.root_ref, v.depth};
    for (int i = 0; i < 3; i++) { t.root_lo[i] = h.root_lo[i]; t.root_hi[i] = h.root_hi[i]; }
    return upload_tree(ctx, t, (size_t)h.n_objects);
    });
}

int rt_scene_upload_lbvh(rt_context* ctx, const rt_scene* scene, double* build_ms)
{
    return guarded([&]() -> int {
    if (!ctx || !scene) return fail(RT_ERR_INVALID, "null argument");
    const rth::Scene& s = scene->s;
    int rc = use_device(ctx);
    if (rc) return rc;
    std::vector<rtf::Prim<double>> prims, planes;
    bool tri_only = true;
    for (size_t i = 0; i < s.objects.size(); i++) {
        const rtf::ObjShade& o = s.objects[i].sh;
        if (o.kind == rtf::PRIM_PLANE) { planes.push_back(rth::make_prim(o, (uint32_t)i)); continue; }
        prims.push_back(rth::make_prim(o, (uint32_t)i));
        tri_only = tri_only && o.kind == rtf::PRIM_TRI;
    }
    const long long n = (long long)prims.size();
    if (n >= (1ll << rtf::LEAF_COUNT_SHIFT)) return fail(RT_ERR_LIMIT, "%lld primitives exceed the leaf reference range", n);
    auto pl32 = rth::to_f32(planes);
    if ((rc = upload(ctx->planes64, planes, ctx->stream))) return rc;
    if ((rc = upload(ctx->planes32, pl32, ctx->stream))) return rc;
    if ((rc = upload_shading(ctx, s))) return rc;

    // working set of the build, carved out of the context's scratch buffers
    const long long ni = n > 1 ? n - 1 : 0;
    DevBuf &in = ctx->hits_a, &work = ctx->hits_b, &tmp = ctx->rays;
    if ((rc = upload(in, prims, ctx->stream))) return rc;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t at = off; off += align(bytes); return at; };
    const size_t o_pbox = carve(n * sizeof(rtl::Box6)), o_lbox = carve(n * sizeof(rtl::Box6)), o_nbox = carve(ni * sizeof(rtl::Box6));
    const size_t o_key = carve(n * 8), o_key2 = carve(n * 8), o_val = carve(n * 4), o_val2 = carve(n * 4);
    const size_t o_left = carve(ni * 4), o_right = carve(ni * 4), o_first = carve(ni * 4), o_last = carve(ni * 4), o_pin = carve(ni * 4), o_plf = carve(n * 4);
    const size_t o_arr = carve(ni * 4), o_keep = carve(ni * 4), o_new = carve(ni * 4), o_cb = carve(64), o_depth = carve(64);
    if ((rc = work.reserve(off + 256))) return rc;
    char* w = (char*)work.p;
    auto* pbox = (rtl::Box6*)(w + o_pbox); auto* lbox = (rtl::Box6*)(w + o_lbox); auto* nbox = (rtl::Box6*)(w + o_nbox);
    auto* key = (unsigned long long*)(w + o_key); auto* key2 = (unsigned long long*)(w + o_key2);
    auto* val = (uint32_t*)(w + o_val); auto* val2 = (uint32_t*)(w + o_val2);
    auto* left = (uint32_t*)(w + o_left); auto* right = (uint32_t*)(w + o_right); auto* first = (uint32_t*)(w + o_first); auto* last = (uint32_t*)(w + o_last);
    auto* pin = (uint32_t*)(w + o_pin); auto* plf = (uint32_t*)(w + o_plf); auto* arrived = (unsigned*)(w + o_arr);
    auto* keep = (uint32_t*)(w + o_keep); auto* newidx = (uint32_t*)(w + o_new);
    auto* cb = (unsigned long long*)(w + o_cb); int* d_depth = (int*)(w + o_depth);

    if ((rc = ctx->prims64.reserve(std::max<size_t>(64, n * sizeof(rtf::Prim<double>))))) return rc;
    if ((rc = ctx->prims32.reserve(std::max<size_t>(64, n * sizeof(rtf::Prim<float>))))) return rc;
    uint32_t root_ref = rtf::NO_NODE;
    rtl::Box6 root_box{};
    int depth = 0;
    size_t n_nodes = 0;
    float ms = 0;
    if (n > 0) {
        const int T = 256;
        auto grid = [&](long long m) { return dim3((unsigned)((m + T - 1) / T)); };
        size_t sort_bytes = 0, scan_bytes = 0;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, key, key2, val, val2, (int)n, 0, 63, ctx->stream));
        if (ni) HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep, newidx, (int)ni, ctx->stream));
        if ((rc = tmp.reserve(std::max(sort_bytes, scan_bytes) + 256))) return rc;
        unsigned long long cb0[8] = {~0ull, ~0ull, ~0ull, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(cb, cb0, sizeof cb0, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemsetAsync(d_depth, 0, 4, ctx->stream));
        HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
        hipLaunchKernelGGL(rtl::bounds_kernel, grid(n), dim3(T), 0, ctx->stream, (const rtf::Prim<double>*)in.p, n, pbox, cb);
        hipLaunchKernelGGL(rtl::morton_kernel, grid(n), dim3(T), 0, ctx->stream, pbox, n, cb, key, val);
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, sort_bytes, key, key2, val, val2, (int)n, 0, 63, ctx->stream));
        if (ni) {
            HIP_TRY(hipMemsetAsync(arrived, 0, ni * 4, ctx->stream));
            hipLaunchKernelGGL(rtl::radix_tree_kernel, grid(ni), dim3(T), 0, ctx->stream, rtl::Radix{key2, n}, left, right, first, last, pin, plf);
        }
        hipLaunchKernelGGL(rtl::refit_kernel, grid(n), dim3(T), 0, ctx->stream, n, val2, pbox, lbox, nbox, left, right, pin, plf, arrived);
        if (ni) {
            hipLaunchKernelGGL(rtl::keep_kernel, grid(ni), dim3(T), 0, ctx->stream, ni, first, last, keep);
            HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, scan_bytes, keep, newidx, (int)ni, ctx->stream));
        }
        hipLaunchKernelGGL(rtl::emit_prims_kernel, grid(n), dim3(T), 0, ctx->stream, n, val2, (const rtf::Prim<double>*)in.p, (rtf::Prim<double>*)ctx->prims64.p,
                           (rtf::Prim<float>*)ctx->prims32.p);
        HIP_TRY(hipGetLastError());
        uint32_t last_keep = 0, last_new = 0;
        if (ni) {
            HIP_TRY(hipMemcpyAsync(&last_keep, keep + ni - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipMemcpyAsync(&last_new, newidx + ni - 1, 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(hipMemcpyAsync(&root_box, ni ? nbox : lbox, sizeof root_box, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        n_nodes = (size_t)last_keep + last_new;
        if ((rc = ctx->nodes64.reserve(std::max<size_t>(64, n_nodes * sizeof(rtf::Node<double>))))) return rc;
        if ((rc = ctx->nodes32.reserve(std::max<size_t>(64, n_nodes * sizeof(rtf::Node<float>))))) return rc;
        if (n_nodes) {
            rtl::EmitArgs ea{left, right, first, last, keep, newidx, pin, lbox, nbox};
            hipLaunchKernelGGL(rtl::emit_nodes_kernel, grid(ni), dim3(T), 0, ctx->stream, ni, ea, (rtf::Node<double>*)ctx->nodes64.p,
                               (rtf::Node<float>*)ctx->nodes32.p, d_depth);
            HIP_TRY(hipGetLastError());
            root_ref = 0;      // the radix tree's root is inner node 0, first in the renumbering
        } else {
            root_ref = rtf::make_leaf(0, (uint32_t)n);
        }
        HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
        HIP_TRY(hipMemcpyAsync(&depth, d_depth, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    } else {
        if ((rc = ctx->nodes64.reserve(64))) return rc;
        if ((rc = ctx->nodes32.reserve(64))) return rc;
    }
    if (build_ms) *build_ms = ms;

    ctx->v64.nodes = (const rtf::Node<double>*)ctx->nodes64.p;
    ctx->v64.prims = (const rtf::Prim<double>*)ctx->prims64.p;
    ctx->v64.planes = (const rtf::Prim<double>*)ctx->planes64.p;
    ctx->v32.nodes = (const rtf::Node<float>*)ctx->nodes32.p;
    ctx->v32.prims = (const rtf::Prim<float>*)ctx->prims32.p;
    ctx->v32.planes = (const rtf::Prim<float>*)ctx->planes32.p;
    for (int i = 0; i < 3; i++) {
        ctx->v64.root_lo[i] = root_box.lo[i]; ctx->v64.root_hi[i] = root_box.hi[i];
        ctx->v32.root_lo[i] = (float)root_box.lo[i]; ctx->v32.root_hi[i] = (float)root_box.hi[i];
    }
    ctx->v64.root_ref = ctx->v32.root_ref = root_ref;
    ctx->v64.n_planes = ctx->v32.n_planes = (int32_t)planes.size();
    ctx->v64.tri_only = ctx->v32.tri_only = tri_only ? 1 : 0;
    ctx->depth = depth;
    ctx->n_nodes = (int)n_nodes;
    ctx->n_prims = (int)n;
    ctx->n_objects = (int)s.objects.size();
    ctx->has_scene = true;
    ctx->scene_generation++;
    return RT_OK;
    });
}

int rt_trace_closest_dev(rt_context* ctx, int precision, int64_t n, const double* d_org, const double* d_dir, double* d_dist, int32_t* d_ids)
{
    return guarded([&]() -> int {
    if (!ctx || !d_org || !d_dir || !d_dist || !d_ids) return fail(RT_ERR_INVALID, "null argument");
    if (!ctx->has_scene) return fail(RT_ERR_STATE, "no scene uploaded");
    if (n <= 0) return RT_OK;
    int rc = use_device(ctx);
    if (rc) return rc;
    if (precision == RT_F64) return launch_batch<double>(ctx, n, d_org, d_dir, d_dist, d_ids);
    if (precision == RT_F32) return launch_batch<float>(ctx, n, d_org, d_dir, d_dist, d_ids);
    return fail(RT_ERR_INVALID, "precision %d", precision);
    });
}

int rt_trace_closest(rt_context* ctx, int precision, int64_t n, const double* org, const double* dir, double* dist, int32_t* ids)
{
    return guarded([&]() -> int {
    if (!ctx || !org || !dir || !dist || !ids) return fail(RT_ERR_INVALID, "null argument");
    if (!ctx->has_scene) return fail(RT_ERR_STATE, "no scene uploaded");
    if (n <= 0) return RT_OK;
    int rc = use_device(ctx);
    if (rc) return rc;
    // AoS (n x 3) -> SoA
    std::vector<double> so((size_t)3 * n), sd((size_t)3 * n);
    for (int64_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) { so[(size_t)c * n + i] = org[3 * i + c]; sd[(size_t)c * n + i] = dir[3 * i + c]; }
    if ((rc = ctx->st_org.reserve(so.size() * 8))) return rc;
    if ((rc = ctx->st_dir.reserve(sd.size() * 8))) return rc;
    if ((rc = ctx->st_dist.reserve((size_t)n * 8))) return rc;
    if ((rc = ctx->st_ids.reserve((size_t)n * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->st_org.p, so.data(), so.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->st_dir.p, sd.data(), sd.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->st_dist.p, dist, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = rt_trace_closest_dev(ctx, precision, n, (const double*)ctx->st_org.p, (const double*)ctx->st_dir.p, (double*)ctx->st_dist.p,
                              (int32_t*)ctx->st_ids.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dist, ctx->st_dist.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ids, ctx->st_ids.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    rtk::TraceCounters c;
    if ((rc = read_counters(ctx, &c))) return rc;
    if (c.overflow) return overflow_error(ctx, "traversal stack overflow");
    return RT_OK;
    });
}

int rt_trace_primary_dev(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end, int32_t* d_ids,
                         double* d_t, int32_t* d_shadow_ids, rt_stats* stats)
{
    return guarded([&]() -> int {
    if (!ctx || !cam || !d_ids) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (!ctx->has_scene) return fail(RT_ERR_STATE, "no scene uploaded");
    int part_rows = p->band_rows > 0 ? p->height / p->n_parts : p->height;
    if (row_begin < 0 || row_end > part_rows || row_begin >= row_end) return fail(RT_ERR_INVALID, "rows [%d,%d) outside [0,%d)", row_begin, row_end, part_rows);
    if (p->aadepth != 1) return fail(RT_ERR_INVALID, "rt_trace_primary handles aadepth 1 only");
    if ((rc = use_device(ctx))) return rc;
    rtk::PrimaryArgs a;
    for (int i = 0; i < 3; i++) {
        a.cam_pos[i] = cam->pos[i]; a.cam_dir[i] = cam->dir[i]; a.cam_right[i] = cam->right[i]; a.cam_down[i] = cam->down[i];
        a.light[i] = p->light_pos[i];
    }
    a.width = p->width; a.height = p->height; a.row_begin = row_begin; a.row_end = row_end;
    a.shadow = p->shadow_rays == 2 ? 2 : (p->shadow_rays ? 1 : 0);
    a.band_rows = p->band_rows; a.n_parts = p->n_parts; a.part = p->part;
    // the order the frame's tiles are taken in: measured per view (tile_order_of) unless a switch fixes it
    int order = 1;
    a.strata = 1;
    hipEvent_t tune_ev0 = nullptr, tune_ev1 = nullptr;
    if (const char* e = getenv("RT_TILE_STRATA")) a.strata = std::max(1, std::min(16, atoi(e)));      // tuning aids (see tile_of_wave)
    else if (const char* e = getenv("RT_TILE_ORDER")) order = std::max(1, std::min(TILE_ORDERS, atoi(e)));
    else if (!p->count_work) order = tile_order_of(ctx, tile_view_key(ctx, a, p->precision), p->precision == RT_F32 ? 3 : 2, &tune_ev0, &tune_ev1);
    if (order == 2) a.strata = 2;
    const bool two_tiles = order == 3;
    if (stats) HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));      // (timing events only for a caller that reads them)
    if (tune_ev0) HIP_TRY(hipEventRecord(tune_ev0, ctx->stream));
    if (p->count_work)
        rc = p->precision == RT_F64 ? launch_primary<double, true>(ctx, a, d_ids, d_t, d_shadow_ids, stats != nullptr, false) : launch_primary<float, true>(ctx, a, d_ids, d_t, d_shadow_ids, stats != nullptr, false);
    else
        rc = p->precision == RT_F64 ? launch_primary<double, false>(ctx, a, d_ids, d_t, d_shadow_ids, stats != nullptr, false) : launch_primary<float, false>(ctx, a, d_ids, d_t, d_shadow_ids, stats != nullptr, two_tiles);
    if (rc) return rc;
    if (tune_ev1) HIP_TRY(hipEventRecord(tune_ev1, ctx->stream));
    if (stats) HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (stats) {
        rtk::TraceCounters c;
        if ((rc = read_counters(ctx, &c))) return rc;
        if (c.overflow) return overflow_error(ctx, "traversal stack overflow");
        memset(stats, 0, sizeof(*stats));
        stats->primary_rays = (uint64_t)p->width * (uint64_t)(row_end - row_begin);
        stats->shadow_rays = c.shadow;
        stats->inner_visits = c.inner;
        stats->prim_tests = c.prims;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        stats->kernel_ms = ms;
        stats->trace_kernel_ms = ms;
        stats->tile_order = getenv("RT_TILE_STRATA") ? (uint64_t)a.strata : (uint64_t)order;
    }
    return RT_OK;
    });
}

int rt_trace_primary(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end, int32_t* ids, double* t,
                     int32_t* shadow_ids, rt_stats* stats)
{
    return guarded([&]() -> int {
    if (!ctx || !ids) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(p);
    if (rc) return rc;
    int part_rows = p->band_rows > 0 ? p->height / p->n_parts : p->height;
    if (row_begin < 0 || row_end > part_rows || row_begin >= row_end) return fail(RT_ERR_INVALID, "rows [%d,%d) outside [0,%d)", row_begin, row_end, part_rows);
    if ((rc = use_device(ctx))) return rc;
    size_t n = (size_t)p->width * (row_end - row_begin);
    if ((rc = ctx->st_ids.reserve(n * 4))) return rc;
    if ((rc = ctx->st_dist.reserve(n * 8))) return rc;
    if ((rc = ctx->st_aux.reserve(n * 4))) return rc;
    rt_stats local;
    rc = rt_trace_primary_dev(ctx, cam, p, row_begin, row_end, (int32_t*)ctx->st_ids.p, t ? (double*)ctx->st_dist.p : nullptr,
                              (shadow_ids && p->shadow_rays) ? (int32_t*)ctx->st_aux.p : nullptr, &local);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ids, ctx->st_ids.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (t) HIP_TRY(hipMemcpyAsync(t, ctx->st_dist.p, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (shadow_ids && p->shadow_rays) HIP_TRY(hipMemcpyAsync(shadow_ids, ctx->st_aux.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (stats) *stats = local;
    return RT_OK;
    });
}

// ---------------------------------------------------------------- photon map
int rt_photons_upload(rt_context* ctx, int which, int64_t n, const double* photons)
{
    return guarded([&]() -> int {
    if (!ctx || (which != 0 && which != 1) || n < 0 || (n > 0 && !photons)) return fail(RT_ERR_INVALID, "bad argument");
    if (n >= (1ll << 31)) return fail(RT_ERR_LIMIT, "photon count %lld exceeds 2^31", (long long)n);
    int rc = use_device(ctx);
    if (rc) return rc;
    rth::FlatPhotonMap fm;
    rth::build_photon_map(photons, n, fm);
    if ((rc = upload(ctx->ph_pos[which], fm.pos, ctx->stream))) return rc;
    if ((rc = upload(ctx->ph_aux[which], fm.aux, ctx->stream))) return rc;
    if ((rc = upload(ctx->ph_order[which], fm.order, ctx->stream))) return rc;
    if ((rc = upload(ctx->ph_box[which], fm.leaf_box, ctx->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->pm[which].pos = (const float4*)ctx->ph_pos[which].p;
    ctx->pm[which].aux = (const float*)ctx->ph_aux[which].p;
    ctx->pm[which].order = (const int*)ctx->ph_order[which].p;
    ctx->pm[which].leaf_box = (const float*)ctx->ph_box[which].p;
    ctx->pm[which].n = n;
    ctx->ph_src_n[which] = -1;
    return RT_OK;
    });
}

}  // extern "C"

namespace {
int knn_launch(rt_context* ctx, int which, int k, double max_dist, int photon_num, int64_t n, const double* pos, const double* nrm, const double* col,
               int32_t* counts, double* r2, double* rgb, int32_t* idx = nullptr)
{
    if (k < 1 || k > rtg::MAX_K) return fail(RT_ERR_LIMIT, "k = %d outside [1, %d]", k, rtg::MAX_K);
    int rc = use_device(ctx);
    if (rc) return rc;
    size_t b3 = (size_t)n * 3 * sizeof(double);
    if ((rc = ctx->st_org.reserve(b3))) return rc;
    if ((rc = ctx->st_dir.reserve(b3 * 3))) return rc;          // normals | colours | rgb out
    if ((rc = ctx->st_dist.reserve((size_t)n * 8))) return rc;
    if ((rc = ctx->st_ids.reserve((size_t)n * 4))) return rc;
    if (idx && (rc = ctx->st_aux.reserve((size_t)n * k * 4))) return rc;
    double* d_pos = (double*)ctx->st_org.p;
    double* d_nrm = (double*)ctx->st_dir.p;
    double* d_col = d_nrm + (size_t)n * 3;
    double* d_rgb = d_col + (size_t)n * 3;
    HIP_TRY(hipMemcpyAsync(d_pos, pos, b3, hipMemcpyHostToDevice, ctx->stream));
    if (nrm) HIP_TRY(hipMemcpyAsync(d_nrm, nrm, b3, hipMemcpyHostToDevice, ctx->stream));
    if (col) HIP_TRY(hipMemcpyAsync(d_col, col, b3, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, 64, ctx->stream));
    rtg::GatherParams gp;
    gp.front_cap = gather_front_cap();
    gp.k = k;
    gp.r2 = (float)(max_dist * max_dist);
    gp.norm = (double)(photon_num / 16) * 3.1415926;            // PHOTONMUM/16*PI, integer division first (main.cpp:293)
    int blocks = (int)std::min<int64_t>(n, (int64_t)ctx->num_cus * 8);
    // RT_GATHER_PROF=1: per-phase cycle totals of the gather, printed to stderr (development aid)
    const bool prof = getenv("RT_GATHER_PROF") != nullptr;
    unsigned long long* d_prof = prof ? (unsigned long long*)ctx->counters.p + 8 : nullptr;
    if (prof) HIP_TRY(hipMemsetAsync(d_prof, 0, rtg::P_SLOTS * 8, ctx->stream));
    hipLaunchKernelGGL(rtg::knn_batch_kernel, dim3(blocks), dim3(rtg::WAVE), 0, ctx->stream, ctx->pm[which], (long long)n, d_pos,
                       nrm ? d_nrm : nullptr, col ? d_col : nullptr, gp, (int*)ctx->st_ids.p, (double*)ctx->st_dist.p, rgb ? d_rgb : nullptr,
                       idx ? (int*)ctx->st_aux.p : nullptr, (unsigned long long*)ctx->counters.p, d_prof);
    HIP_TRY(hipGetLastError());
    if (prof) {
        unsigned long long v[rtg::P_SLOTS];
        HIP_TRY(hipMemcpyAsync(v, d_prof, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        double nq = (double)std::max<unsigned long long>(1, v[rtg::P_QUERIES]);
        fprintf(stderr, "gather cycles/query (k=%d): descent %.0f, first scan %.0f, select %.0f (%.2f calls), rounds %.0f (%.1f), leaves %.0f (%.1f), estimate %.0f\n",
                k, v[rtg::P_DESCENT] / nq, v[rtg::P_SCAN0] / nq, v[rtg::P_SELECT] / nq, v[rtg::P_N_SELECT] / nq, v[rtg::P_ROUNDS] / nq,
                v[rtg::P_N_ROUNDS] / nq, v[rtg::P_LEAVES] / nq, v[rtg::P_N_LEAVES] / nq, v[rtg::P_ESTIMATE] / nq);
    }
    if (counts) HIP_TRY(hipMemcpyAsync(counts, ctx->st_ids.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (r2) HIP_TRY(hipMemcpyAsync(r2, ctx->st_dist.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (rgb) HIP_TRY(hipMemcpyAsync(rgb, d_rgb, b3, hipMemcpyDeviceToHost, ctx->stream));
    if (idx) HIP_TRY(hipMemcpyAsync(idx, ctx->st_aux.p, (size_t)n * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_knn_radius(rt_context* ctx, int which, int k, double max_dist, int64_t n, const double* centers, int32_t* counts, double* r2)
{
    return guarded([&]() -> int {
    if (!ctx || !centers || (which != 0 && which != 1) || n < 0) return fail(RT_ERR_INVALID, "bad argument");
    if (n == 0) return RT_OK;
    return knn_launch(ctx, which, k, max_dist, 16, n, centers, nullptr, nullptr, counts, r2, nullptr);
    });
}

int rt_knn_query(rt_context* ctx, int which, int k, double max_dist, int64_t n, const double* centers, int32_t* idx, int32_t* counts)
{
    return guarded([&]() -> int {
    if (!ctx || !centers || !idx || (which != 0 && which != 1) || n < 0) return fail(RT_ERR_INVALID, "bad argument");
    if (n == 0) return RT_OK;
    return knn_launch(ctx, which, k, max_dist, 16, n, centers, nullptr, nullptr, counts, nullptr, nullptr, idx);
    });
}

int rt_knn_estimate(rt_context* ctx, const rt_render_params* p, int64_t n, const double* pos, const double* normal, const double* color,
                    double* rgb, int32_t* counts)
{
    return guarded([&]() -> int {
    if (!ctx || !pos || !normal || !color || !rgb || n < 0) return fail(RT_ERR_INVALID, "bad argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (n == 0) return RT_OK;
    return knn_launch(ctx, 0, p->photon_use, p->photon_dist, p->photon_num, n, pos, normal, color, counts, nullptr, rgb);
    });
}

// ---------------------------------------------------------------- full pixel path
}  // extern "C"

namespace {
int launch_gather(rt_context* ctx, const rtg::GatherRec* recs, const rtq::Counts* cnt, unsigned long long shard_cap, unsigned long long* next_window,
                  unsigned long long* n_redo, unsigned long long* next_gather, unsigned long long* windows_done, const rtg::GatherParams& gp, double* d_acc,
                  uint32_t* d_redo, size_t redo_entries, bool may_profile = true, float4* d_pack = nullptr, int pack_waves = 0, int pack_cap = 0,
                  unsigned long long* d_cont = nullptr, unsigned cont_cap = 0);
int reserve_redo(rt_context* ctx, size_t entries);
}
extern "C" {

int rt_knn_estimate_grouped(rt_context* ctx, const rt_render_params* p, int64_t n, const double* pos, const double* normal, const double* color,
                            double* rgb)
{
    return guarded([&]() -> int {
    if (!ctx || !pos || !normal || !color || !rgb || n < 0) return fail(RT_ERR_INVALID, "bad argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (p->photon_use < 1 || p->photon_use > rtg::MAX_K) return fail(RT_ERR_LIMIT, "photon_use %d outside [1,%d]", p->photon_use, rtg::MAX_K);
    if (n == 0) return RT_OK;
    if (n >= (1ll << 31)) return fail(RT_ERR_LIMIT, "%lld requests", (long long)n);
    if ((rc = use_device(ctx))) return rc;
    // the requests as the shading kernel would queue them: windows of 64 in one shard
    const size_t slots = (size_t)((n + 63) / 64) * 64;
    std::vector<rtg::GatherRec> recs(slots);
    memset(recs.data(), 0, slots * sizeof(rtg::GatherRec));
    for (int64_t i = 0; i < n; i++) {
        rtg::GatherRec& g = recs[(size_t)i];
        for (int c = 0; c < 3; c++) { g.pos[c] = pos[3 * i + c]; g.normal[c] = normal[3 * i + c]; g.color[c] = color[3 * i + c]; }
        g.weight = 1.0;
        g.pixel = (uint32_t)i;
        if (i % 64 == 0) g.flags = (uint32_t)std::min<int64_t>(64, n - i) << rtg::GATHER_COUNT_SHIFT;
    }
    if ((rc = upload(ctx->gq, recs, ctx->stream))) return rc;
    if ((rc = reserve_redo(ctx, slots))) return rc;
    if ((rc = ctx->acc.reserve((size_t)n * 3 * sizeof(double)))) return rc;
    rtr::RenderCounters* ctr = (rtr::RenderCounters*)ctx->counters.p;
    HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(rtr::RenderCounters), ctx->stream));
    const unsigned long long n_slots = slots;
    HIP_TRY(hipMemcpyAsync(&ctr->gather.n[0][0], &n_slots, 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->acc.p, 0, (size_t)n * 3 * sizeof(double), ctx->stream));
    rtg::GatherParams gp;
    gp.front_cap = gather_front_cap();
    gp.k = p->photon_use;
    gp.r2 = (float)(p->photon_dist * p->photon_dist);
    gp.norm = (double)(p->photon_num / 16) * 3.1415926;
    gp.photon_num = (double)p->photon_num;
    gp.seed = 12345u;
    if ((rc = launch_gather(ctx, (const rtg::GatherRec*)ctx->gq.p, &ctr->gather, slots, &ctr->next_window, &ctr->n_redo, &ctr->next_gather, ctr->tile_progress, gp,
                            (double*)ctx->acc.p, (uint32_t*)ctx->redo.p, ctx->redo.bytes / sizeof(uint32_t))))
        return rc;
    HIP_TRY(hipMemcpyAsync(rgb, ctx->acc.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));    // also: `recs` and `n_slots` die at return
    return RT_OK;
    });
}
}  // extern "C"

namespace {
// Requests from device arrays (n x 3 doubles each) into windows of 64 queue records, as a shading wave would queue
// them; the output doubles as the accumulator and starts at zero; one thread notes the queue's fill.
__global__ void pack_requests_kernel(long long n, long long slots, const double* __restrict__ pos, const double* __restrict__ nrm,
                                     const double* __restrict__ col, rtg::GatherRec* __restrict__ recs, rtq::Counts* cnt, double* __restrict__ rgb)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    rtg::GatherRec g;
    for (int c = 0; c < 3; c++) { g.pos[c] = 0; g.normal[c] = 0; g.color[c] = 0; g.march[c] = 0; }
    g.weight = 0; g.pixel = 0; g.flags = 0;
    if (i < n) {
        for (int c = 0; c < 3; c++) { g.pos[c] = pos[3 * i + c]; g.normal[c] = nrm[3 * i + c]; g.color[c] = col[3 * i + c]; rgb[3 * i + c] = 0.0; }
        g.weight = 1.0;
        g.pixel = (uint32_t)i;
        if (i % 64 == 0) g.flags = (uint32_t)(n - i < 64 ? n - i : 64) << rtg::GATHER_COUNT_SHIFT;
    }
    recs[i] = g;
    if (i == 0) cnt->n[0][0] = (unsigned long long)slots;
}
}  // namespace

extern "C" {
int rt_knn_estimate_dev_noalloc(rt_context* ctx, const rt_render_params* p, int64_t n, const double* d_pos, const double* d_normal,
                                const double* d_color, double* d_rgb)
{
    return guarded([&]() -> int {
    if (!ctx || !d_pos || !d_normal || !d_color || !d_rgb || n < 0) return fail(RT_ERR_INVALID, "bad argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (p->photon_use < 1 || p->photon_use > rtg::MAX_K) return fail(RT_ERR_LIMIT, "photon_use %d outside [1,%d]", p->photon_use, rtg::MAX_K);
    if (n > RT_NOALLOC_MAX_REQUESTS)
        return fail(RT_ERR_LIMIT, "%lld requests exceed the %d this entry point has working memory for (split the batch, or use rt_knn_estimate_grouped)",
                    (long long)n, RT_NOALLOC_MAX_REQUESTS);
    if (n == 0) return RT_OK;
    if (ctx->device >= NA_MAX_DEVICES || !g_scratch_at[ctx->device]) return fail(RT_ERR_STATE, "no working array on device %d", ctx->device);
    if ((rc = use_device(ctx))) return rc;
    unsigned char* const base = g_scratch_at[ctx->device];          // the device address, never the symbol's name
    // one call at a time on this GPU's array: behind the kernels of the call before, whichever context made it
    std::lock_guard<std::mutex> scratch_lock(g_scratch_mu);
    if (g_scratch_used[ctx->device]) HIP_TRY(hipStreamWaitEvent(ctx->stream, g_scratch_done[ctx->device], 0));
    rtr::RenderCounters* ctr = (rtr::RenderCounters*)base;
    rtg::GatherRec* recs = (rtg::GatherRec*)(base + NA_COUNTERS);
    uint32_t* redo = (uint32_t*)(base + NA_COUNTERS + NA_QUEUE);
    const long long slots = (n + 63) / 64 * 64;
    // the array keeps the last call's contents: counters (queue fill, draw counters, hand-over count) start at zero
    HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(rtr::RenderCounters), ctx->stream));
    const int T = 256;
    hipLaunchKernelGGL(pack_requests_kernel, dim3((unsigned)((slots + T - 1) / T)), dim3(T), 0, ctx->stream, (long long)n, slots, d_pos, d_normal, d_color, recs,
                       &ctr->gather, d_rgb);
    HIP_TRY(hipGetLastError());
    rtg::GatherParams gp;
    gp.front_cap = gather_front_cap();
    gp.k = p->photon_use;
    gp.r2 = (float)(p->photon_dist * p->photon_dist);
    gp.norm = (double)(p->photon_num / 16) * 3.1415926;
    gp.photon_num = (double)p->photon_num;
    gp.seed = 12345u;
    rc = launch_gather(ctx, recs, &ctr->gather, (unsigned long long)slots, &ctr->next_window, &ctr->n_redo, &ctr->next_gather, ctr->tile_progress, gp, d_rgb, redo,
                       NA_REDO / sizeof(uint32_t), false, (float4*)(base + NA_COUNTERS + NA_QUEUE + NA_REDO), NA_PACK_WAVES, NA_PACK_CAP,
                       (unsigned long long*)(base + NA_COUNTERS + NA_QUEUE + NA_REDO + NA_PACK), NA_CONT_ENTRIES);
    // (recorded even when a launch failed part-way: whatever did get enqueued uses the array)
    if (hipEventRecord(g_scratch_done[ctx->device], ctx->stream) == hipSuccess) g_scratch_used[ctx->device] = true;
    return rc;
    });
}
}  // extern "C"

namespace {

// The gather over a queue of request windows: the group kernel, then the per-request kernel for what it
// left.  RT_GATHER_MODE=query hands every request to the per-request kernel (comparison runs).
constexpr unsigned CONT_ENTRIES = 1u << 15;      // window parts per list handed between the group kernel's waves at most (8 lists, 4 MB; a full list keeps rounds with their wave)
int launch_gather(rt_context* ctx, const rtg::GatherRec* recs, const rtq::Counts* cnt, unsigned long long shard_cap, unsigned long long* next_window,
                  unsigned long long* n_redo, unsigned long long* next_gather, unsigned long long* windows_done, const rtg::GatherParams& gp, double* d_acc,
                  uint32_t* d_redo, size_t redo_entries, bool may_profile, float4* d_pack, int pack_waves, int pack_cap, unsigned long long* d_cont, unsigned cont_cap)
{
    const char* mode = getenv("RT_GATHER_MODE");
    // RT_GATHER_FORM=tile: the earlier group kernel, which streams the listed leaves themselves (comparison runs)
    const char* form = getenv("RT_GATHER_FORM");
    const bool packed = !(form && form[0] == 't');
    // the group kernel over packed lists: twelve waves per CU, each with a list of its own (the caller's, or the context's)
    int tile_waves = ctx->num_cus * (packed ? 12 : 8);
    if (const char* e = getenv("RT_GATHER_WAVES")) tile_waves = ctx->num_cus * std::max(1, std::min(12, atoi(e)));      // tuning aid: waves per CU
    if (packed) {
        if (!d_pack) {
            pack_cap = rtg::S_PACK_CAP;
            if (const char* e = getenv("RT_GATHER_PACK_CAP")) pack_cap = std::max(1024, std::min(rtg::S_PACK_CAP, atoi(e)));      // tuning aid
            int rc_p = ctx->pack.reserve((size_t)tile_waves * rtg::S_PACK_CAP * sizeof(float4));
            if (rc_p) return rc_p;
            d_pack = (float4*)ctx->pack.p;
            // the list of window parts handed from wave to wave: zero when fresh, left zero by the kernel that used it
            const size_t cont_bytes = rtg::CONT_SHARDS * (4 + 2 * (size_t)CONT_ENTRIES) * sizeof(unsigned long long);
            if (ctx->cont.bytes < cont_bytes) {
                if ((rc_p = ctx->cont.reserve(cont_bytes))) return rc_p;
                HIP_TRY(hipMemsetAsync(ctx->cont.p, 0, cont_bytes, ctx->stream));
            }
            d_cont = (unsigned long long*)ctx->cont.p;
            cont_cap = CONT_ENTRIES;
        } else {
            tile_waves = std::min(tile_waves, pack_waves);
        }
        if (getenv("RT_GATHER_NO_PARTS")) d_cont = nullptr;      // (the switch: comparison runs -- every round of a window stays with the wave that drew it)
        // (the lists' counts start at zero; their entries are zero already)
        if (d_cont) HIP_TRY(hipMemset2DAsync(d_cont, (4 + 2 * (size_t)cont_cap) * sizeof(unsigned long long), 0, 4 * sizeof(unsigned long long), rtg::CONT_SHARDS, ctx->stream));
    }
    // RT_GATHER_PROF=1: per-phase cycle totals of the group kernel, printed to stderr (development aid)
    const bool prof = may_profile && getenv("RT_GATHER_PROF") != nullptr;      // (the cycle accounting allocates its table)
    int rc;
    // RT_GATHER_SPAN=1: when each wave of the packed form began and ended, condensed to stderr (development aid)
    const bool span = may_profile && packed && !prof && getenv("RT_GATHER_SPAN") != nullptr;
    if ((prof || span) && (rc = ctx->st_aux.reserve(std::max<size_t>(rtg::TP_SLOTS * 8, (size_t)tile_waves * 24)))) return rc;
    if (span) HIP_TRY(hipMemsetAsync(ctx->st_aux.p, 0, (size_t)tile_waves * 24, ctx->stream));
    if (prof) HIP_TRY(hipMemsetAsync(ctx->st_aux.p, 0, rtg::TP_SLOTS * 8, ctx->stream));
    if (prof) HIP_TRY(hipMemsetAsync((unsigned long long*)ctx->st_aux.p + rtg::TP_W_FIRST_MIN, 0xFF, 8, ctx->stream));
    rtg::GatherParams gpt = gp;
    // The relative range around the centre's k-th distance searched for a neighbour's: it has to cover the scatter of a
    // k-th distance (about 1 / (2 sqrt(k)) of it) and the change of the photon density across a ball; what it misses
    // is answered per request.  Measured on the water frame: k = 1000 fastest at 0.15 (0.3: +11 %, 0.1: +3 %), k = 100 at 0.25 - 0.3.
    gpt.opt_range = std::min(0.5f, 0.1f + 1.5f / sqrtf((float)gp.k));
    if (const char* e = getenv("RT_GATHER_OPT_RANGE")) gpt.opt_range = (float)atof(e);      // tuning aids
    if (const char* e = getenv("RT_GATHER_MIN_LANES")) gpt.min_lanes = atoi(e);
    if (const char* e = getenv("RT_GATHER_ACT_REACH")) gpt.act_reach = (float)atof(e);
    if (const char* e = getenv("RT_GATHER_ROUNDS")) gpt.max_rounds = atoi(e);
    int part_tail = 2;
    if (const char* e = getenv("RT_GATHER_PART_TAIL")) part_tail = std::max(0, atoi(e));

    // Subsurface materials: the group kernel (packed form) answers their surface estimates and their volume terms; a request it
    // cannot finish is left in its record with the surface estimate for the per-request kernel (the one alongside reads
    // such records past its caches: fresh_record).
    const bool sss_scene = ctx->has_sss && !getenv("RT_SSS_PER_REQUEST");      // (the switch: comparison runs)
    const int sss_mode = !sss_scene ? 0 : (getenv("RT_SSS_VOLUME_PER_REQUEST") ? 1 : 2);      // (the switch: the volume terms per request, as the earlier form has them)
    rtg::TileArgs ta{recs, cnt, shard_cap, next_window, d_redo, n_redo, windows_done, ((mode && mode[0] == 'q') || ctx->pm[0].n == 0) ? 1 : 0,      // (an empty surface map has no tree to walk)
                     sss_mode, prof ? (unsigned long long*)ctx->st_aux.p : nullptr, span ? (unsigned long long*)ctx->st_aux.p : nullptr,
                     packed ? d_cont : nullptr, cont_cap, part_tail, getenv("RT_GATHER_QUEUE_ORDER") ? 0 : 1, pack_cap};
    HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));      // (the queue and the counters are ready here)
    if (packed && prof) hipLaunchKernelGGL(rtg::gather_stream_prof_kernel, dim3(tile_waves), dim3(rtg::WAVE), 0, ctx->stream, ctx->pm[0], ctx->pm[1], ta, gpt, d_acc, d_pack);
    else if (packed && sss_scene) hipLaunchKernelGGL(rtg::gather_stream_sss_kernel, dim3(tile_waves), dim3(rtg::WAVE), 0, ctx->stream, ctx->pm[0], ctx->pm[1], ta, gpt, d_acc, d_pack);
    else if (packed) hipLaunchKernelGGL(rtg::gather_stream_kernel, dim3(tile_waves), dim3(rtg::WAVE), 0, ctx->stream, ctx->pm[0], ta, gpt, d_acc, d_pack);
    else hipLaunchKernelGGL(rtg::gather_tile_kernel, dim3(tile_waves), dim3(rtg::WAVE), 0, ctx->stream, ctx->pm[0], ta, gpt, d_acc);
    HIP_TRY(hipGetLastError());
    // The per-request kernel alongside, on the second stream.  Its waves stay only once every wave of the group kernel
    // has begun, so they never hold a place the group kernel waits for; six blocks per CU leave room for one of the
    // group kernel's in any case.
    // (the switch: comparison runs.  Not in subsurface scenes with the earlier group form, which has no fence before its hand-overs,
    // nor with the volume terms left per request, the form that comparison is meant to show: every such request is then handed
    // over, and 96 000 of them taken one at a time beside the group kernel took 13.9 ms where the launch after it takes 11.5)
    const bool alongside = (!sss_scene || (packed && sss_mode == 2)) && !getenv("RT_GATHER_NO_ALONGSIDE");
    if (alongside) {
        HIP_TRY(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        hipLaunchKernelGGL(rtg::gather_alongside_kernel, dim3(ctx->num_cus * 6), dim3(rtg::WAVE), 0, ctx->side, ctx->pm[0], ctx->pm[1], recs, d_redo,
                           (uint32_t)std::min<size_t>(redo_entries, 0xFFFFFFFFu), cnt, shard_cap, (const unsigned long long*)windows_done, (unsigned)tile_waves,
                           next_gather, gp, d_acc, sss_scene ? 1 : 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->side));
    }
    if (span) {
        std::vector<unsigned long long> v((size_t)tile_waves * 3);
        HIP_TRY(hipMemcpyAsync(v.data(), ctx->st_aux.p, v.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < tile_waves; i++) if (v[3 * i + 1]) { t0 = std::min(t0, v[3 * i]); t1 = std::max(t1, v[3 * i + 1]); }
        std::vector<double> ends, begins;
        double busy = 0;
        unsigned long long idle_waves = 0, spins = 0, parts = 0;
        for (int i = 0; i < tile_waves; i++) {
            if (!v[3 * i + 1]) continue;
            if (!(v[3 * i + 2] & 0xFFFFFFFFull)) idle_waves++;
            spins += v[3 * i + 2] >> 32; parts += v[3 * i + 2] & 0xFFFFFFFFull;
            ends.push_back((double)(v[3 * i + 1] - t0) / 100.0); begins.push_back((double)(v[3 * i] - t0) / 100.0);
            busy += (double)(v[3 * i + 1] - v[3 * i]) / 100.0;
        }
        std::sort(ends.begin(), ends.end()); std::sort(begins.begin(), begins.end());
        const size_t n = ends.size();
        if (n) fprintf(stderr, "packed group gather, waves in time (us from the first wave's start): %zu waves (%llu took no window), span %.0f, mean wave %.0f; began: median %.0f, 90 %% %.0f, last %.0f; "
                               "ended: 10 %% %.0f, 25 %% %.0f, median %.0f, 75 %% %.0f, 90 %% %.0f, 99 %% %.0f, last %.0f; %llu windows and parts taken, %llu polls for a part's entry\n",
                       n, idle_waves, (double)(t1 - t0) / 100.0, busy / n, begins[n / 2], begins[n * 9 / 10], begins[n - 1], ends[n / 10], ends[n / 4], ends[n / 2], ends[n * 3 / 4], ends[n * 9 / 10],
                       ends[n * 99 / 100], ends[n - 1], parts, spins);
    }
    if (prof) {
        unsigned long long v[rtg::TP_SLOTS];
        HIP_TRY(hipMemcpyAsync(v, ctx->st_aux.p, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        const double nr = (double)std::max<unsigned long long>(1, v[rtg::TP_ROUNDS]);
        if (packed)
            fprintf(stderr, "packed group gather (k=%d): %llu windows, %llu rounds, %.1f requests per round, %.2f attempts at the centre's radius, %.0f photons packed for it, "
                            "%.0f streamed by the passes (%.1f steps); cycles per round: walk + pack + centre %.0f, filter %.0f, hist %.0f, crit %.0f, rank %.0f, sum %.0f\n",
                    gp.k, v[rtg::TP_WINDOWS], v[rtg::TP_ROUNDS], v[rtg::TP_LANES] / nr, v[rtg::TP_N_HIST] / nr, v[rtg::TP_N_CRIT] / nr, v[rtg::TP_N_SUM] / nr, v[rtg::TP_STEPS] / nr,
                    v[rtg::TP_KTH] / nr, v[rtg::TP_WALK] / nr, v[rtg::TP_HIST] / nr, v[rtg::TP_CRIT] / nr, v[rtg::TP_RANK] / nr, v[rtg::TP_SUM] / nr);
        if (packed)
            fprintf(stderr, "  the centre's radius, cycles per round: guess %.0f, walks %.0f, counts %.0f, sort %.0f, packing %.0f, radix passes %.0f; attempts per round that ended in: "
                            "no guess %.3f, lists too small %.3f, fewer than k %.3f, far more than k %.3f, packed list too small %.3f\n",
                    v[rtg::TP_C_GUESS] / nr, v[rtg::TP_C_WALK] / nr, v[rtg::TP_C_COUNT] / nr, v[rtg::TP_C_SORT] / nr, v[rtg::TP_C_PACK] / nr, v[rtg::TP_C_KTH] / nr,
                    v[rtg::TP_A_NOGUESS] / nr, v[rtg::TP_A_LISTS] / nr, v[rtg::TP_A_FEW] / nr, v[rtg::TP_A_MANY] / nr, v[rtg::TP_A_PACK] / nr);
        if (packed)
            fprintf(stderr, "  windows by rounds taken (<= 1, 2, 3, 4 and more): %llu / %llu / %llu / %llu, mean cycles %.0f / %.0f / %.0f / %.0f, longest window %llu (%llu rounds, %llu attempts at a radius, %llu steps); kernel span first wave's start to last wave's end %llu cycles\n",
                    v[rtg::TP_W_ROUNDS1], v[rtg::TP_W_ROUNDS2], v[rtg::TP_W_ROUNDS3], v[rtg::TP_W_ROUNDS4],
                    v[rtg::TP_W_CYC1] / (double)std::max<unsigned long long>(1, v[rtg::TP_W_ROUNDS1]), v[rtg::TP_W_CYC2] / (double)std::max<unsigned long long>(1, v[rtg::TP_W_ROUNDS2]),
                    v[rtg::TP_W_CYC3] / (double)std::max<unsigned long long>(1, v[rtg::TP_W_ROUNDS3]), v[rtg::TP_W_CYC4] / (double)std::max<unsigned long long>(1, v[rtg::TP_W_ROUNDS4]),
                    v[rtg::TP_W_MAX] >> 24, (v[rtg::TP_W_MAX] >> 16) & 255, (v[rtg::TP_W_MAX] >> 8) & 255, (v[rtg::TP_W_MAX] & 255) * 4, v[rtg::TP_W_LAST_MAX] - v[rtg::TP_W_FIRST_MIN]);
        fprintf(stderr, "group gather (k=%d): %llu windows, %llu rounds, %.1f requests and %.1f steps of 64 photons per round (hist %.1f, crit %.1f, sum %.1f steps); "
                        "cycles per round: kth %.0f, walk %.0f, hist %.0f, crit %.0f, rank %.0f, sum %.0f; requests handed on: %llu other materials, %llu in small rounds, "
                        "%llu box too large, %llu range too low, %llu range too high or crowded bin, %llu left after the last round, %llu ties to split; busy cycles per wave: mean %.0f, longest %llu\n",
                gp.k, v[rtg::TP_WINDOWS], v[rtg::TP_ROUNDS], v[rtg::TP_LANES] / nr, v[rtg::TP_STEPS] / nr, v[rtg::TP_N_HIST] / nr, v[rtg::TP_N_CRIT] / nr, v[rtg::TP_N_SUM] / nr,
                v[rtg::TP_KTH] / nr, v[rtg::TP_WALK] / nr, v[rtg::TP_HIST] / nr, v[rtg::TP_CRIT] / nr, v[rtg::TP_RANK] / nr, v[rtg::TP_SUM] / nr, v[rtg::TP_R_FLAGS],
                v[rtg::TP_R_FEW], v[rtg::TP_R_WALK], v[rtg::TP_R_SHORT], v[rtg::TP_R_CROWD], v[rtg::TP_R_LEFT], v[rtg::TP_R_TIE], v[rtg::TP_BUSY] / (double)tile_waves, v[rtg::TP_BUSY_MAX]);
    }
    hipLaunchKernelGGL(rtg::gather_kernel, dim3(ctx->num_cus * 8), dim3(rtg::WAVE), 0, ctx->stream, ctx->pm[0], ctx->pm[1], recs, d_redo,
                       (const unsigned long long*)n_redo, next_gather, gp, d_acc);
    HIP_TRY(hipGetLastError());
    if (alongside) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));      // what follows on the stream follows both
    return RT_OK;
}

// The hand-over list for `entries` requests.  Fresh memory is set to "empty"; the kernels that read entries reset them.
int reserve_redo(rt_context* ctx, size_t entries)
{
    const size_t before = ctx->redo.bytes;
    int rc = ctx->redo.reserve(std::max<size_t>(entries, 64) * sizeof(uint32_t));
    if (rc) return rc;
    if (ctx->redo.bytes != before) HIP_TRY(hipMemsetAsync(ctx->redo.p, 0xFF, ctx->redo.bytes, ctx->stream));
    return RT_OK;
}

// the three traversal kernels of the render path for one (R, TRI_ONLY, MODE) choice
template <class R, bool TRI_ONLY, int MODE, bool COUNT = false>
int launch_render_levels(rt_context* ctx, const rtr::RenderArgs& a, const TraceGeom& g, int levels, unsigned long long shard_cap, unsigned long long gq_cap,
                         rtr::RenderCounters* ctr, double* d_acc)
{
    const int rows = a.row_end - a.row_begin;
    const int cap = stack_cap(ctx);
    rtr::ShadeData sd{(const rtf::ObjShade*)ctx->objs.p, (const rtf::TexDesc*)ctx->texdesc.p, (const double*)ctx->texels.p};
    const int wpb = g.threads / 64;
    const dim3 grid((unsigned)rtk::tile_grid_blocks((a.col_end - a.col_begin + 7) / 8, (rows + 7) / 8, wpb, MODE == rtk::TRAV_NODES, a.aadepth));
    int rc_lds;
    if (COUNT) {
        if ((rc_lds = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtr::render_primary_kernel<R, TRI_ONLY, MODE, false, COUNT>), g.lds))) return rc_lds;
        hipLaunchKernelGGL((rtr::render_primary_kernel<R, TRI_ONLY, MODE, false, COUNT>), grid, dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), a,
                           (rtr::HitRec*)ctx->hits_a.p, shard_cap, ctr, cap, ctx->n_nodes, d_acc);
    } else if (a.dof_samples > 1) {
        if ((rc_lds = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtr::render_primary_kernel<R, TRI_ONLY, MODE, true>), g.lds))) return rc_lds;
        hipLaunchKernelGGL((rtr::render_primary_kernel<R, TRI_ONLY, MODE, true>), grid, dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), a,
                           (rtr::HitRec*)ctx->hits_a.p, shard_cap, ctr, cap, ctx->n_nodes, d_acc);
    } else {
        if ((rc_lds = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtr::render_primary_kernel<R, TRI_ONLY, MODE, false>), g.lds))) return rc_lds;
        hipLaunchKernelGGL((rtr::render_primary_kernel<R, TRI_ONLY, MODE, false>), grid, dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), a,
                           (rtr::HitRec*)ctx->hits_a.p, shard_cap, ctr, cap, ctx->n_nodes, d_acc);
    }
    HIP_TRY(hipGetLastError());
    // persistent grids for the queue consumers (their work counts live in device memory)
    // (the queue tracer is held to 128 registers: four waves per SIMD, sixteen per CU)
    const int trace_blocks_cu = std::max(1, std::min<int>((int)(160 * 1024 / std::max<size_t>(g.lds, 1)), 16 / wpb));
    const int trace_blocks = ctx->num_cus * trace_blocks_cu;
    const int shade_blocks = ctx->num_cus * 4;
    // Small scenes: the queue tracer keeps nodes AND primitive records in LDS, one block per CU with as many waves as the
    // stacks leave room for (a window's 64 rays meet 64 different triangles; LDS returns them four times as fast as L1).
    int resident_waves = 0;
    size_t resident_lds = 0;
    if (MODE == rtk::TRAV_NODES && !getenv("RT_NO_SCENE_LDS")) {
        const size_t image = rtk::node_lds_bytes<R>((size_t)ctx->n_nodes) + rtk::prim_lds_bytes<R>((size_t)ctx->n_prims);
        const size_t per_wave = rtk::stack_lds_bytes(cap, sizeof(float), 1);
        if (image + 8 * per_wave <= 160 * 1024) {
            // (at most 15: a block of 16 waves measured a third slower -- the float glass frame 1.39 ms against 1.02)
            resident_waves = (int)std::min<size_t>(15, (160 * 1024 - image) / per_wave);
            if (const char* e = getenv("RT_SCENE_WAVES")) resident_waves = std::max(1, std::min(resident_waves, atoi(e)));      // tuning aid
            resident_lds = image + resident_waves * per_wave;
        }
    }
    // Scenes of a few thousand triangles: the node image alone beside the stacks (up to ~100 KB of it), primitives through L1 --
    // a third of the records a diverging wave fetches are nodes.
    int nodes_waves = 0;
    size_t nodes_lds = 0;
    if (MODE == rtk::TRAV_PLAIN && ctx->n_nodes > 0 && !getenv("RT_NO_SCENE_LDS")) {
        const size_t image = rtk::node_lds_bytes<R>((size_t)ctx->n_nodes);
        const size_t per_wave = rtk::stack_lds_bytes(cap, sizeof(float), 1);
        if (image + 12 * per_wave <= 160 * 1024) {
            nodes_waves = (int)std::min<size_t>(15, (160 * 1024 - image) / per_wave);
            if (const char* e = getenv("RT_SCENE_WAVES")) nodes_waves = std::max(1, std::min(nodes_waves, atoi(e)));      // tuning aid
            nodes_lds = image + nodes_waves * per_wave;
        }
    }
    rtr::HitRec* qin = (rtr::HitRec*)ctx->hits_a.p;
    rtr::HitRec* qout = (rtr::HitRec*)ctx->hits_b.p;
    for (int level = 1; level <= levels; level++) {
        const int last = level == levels ? 1 : 0;
        if (ctx->has_glossy)
            hipLaunchKernelGGL(rtr::shade_kernel<true>, dim3(shade_blocks), dim3(rtk::BLOCK), 0, ctx->stream, sd, a, level, last, qin, (rtr::RayRec*)ctx->rays.p,
                               (rtg::GatherRec*)ctx->gq.p, shard_cap, gq_cap, ctr);
        else
            hipLaunchKernelGGL(rtr::shade_kernel<false>, dim3(shade_blocks), dim3(rtk::BLOCK), 0, ctx->stream, sd, a, level, last, qin, (rtr::RayRec*)ctx->rays.p,
                               (rtg::GatherRec*)ctx->gq.p, shard_cap, gq_cap, ctr);
        HIP_TRY(hipGetLastError());
        if (!last) {
            if (MODE == rtk::TRAV_NODES && resident_waves > 0) {
                if (int rc_lds = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtr::trace_queue_kernel<R, TRI_ONLY, rtk::TRAV_SCENE, COUNT>), resident_lds)) return rc_lds;
                hipLaunchKernelGGL((rtr::trace_queue_kernel<R, TRI_ONLY, rtk::TRAV_SCENE, COUNT>), dim3(ctx->num_cus), dim3(resident_waves * 64), resident_lds, ctx->stream,
                                   view<R>(ctx), a, level, (const rtr::RayRec*)ctx->rays.p, qout, shard_cap, ctr, cap, ctx->n_nodes, ctx->n_prims);
            } else if (MODE == rtk::TRAV_PLAIN && nodes_waves > 0) {
                if (int rc_lds = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtr::trace_queue_kernel<R, TRI_ONLY, rtk::TRAV_NODES, COUNT>), nodes_lds)) return rc_lds;
                hipLaunchKernelGGL((rtr::trace_queue_kernel<R, TRI_ONLY, rtk::TRAV_NODES, COUNT>), dim3(ctx->num_cus), dim3(nodes_waves * 64), nodes_lds, ctx->stream,
                                   view<R>(ctx), a, level, (const rtr::RayRec*)ctx->rays.p, qout, shard_cap, ctr, cap, ctx->n_nodes, ctx->n_prims);
            } else {
                if ((rc_lds = allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&rtr::trace_queue_kernel<R, TRI_ONLY, MODE, COUNT>), g.lds))) return rc_lds;
                hipLaunchKernelGGL((rtr::trace_queue_kernel<R, TRI_ONLY, MODE, COUNT>), dim3(trace_blocks), dim3(g.threads), g.lds, ctx->stream, view<R>(ctx), a, level,
                                   (const rtr::RayRec*)ctx->rays.p, qout, shard_cap, ctr, cap, ctx->n_nodes, ctx->n_prims);
            }
            HIP_TRY(hipGetLastError());
        }
        std::swap(qin, qout);
    }
    return RT_OK;
}

// Capacities of the queues of the render path, per shard (32 shards each; multiples of 64: queues are filled in
// windows of 64 slots).  hits / rays hold one recursion level at a time; the gather queue takes requests from
// every level.
struct QueueCaps {
    unsigned long long shard_cap, gq_cap;
    bool retry;      // the capacities are a budget, not a bound: a block that overflows them is rendered again in parts
};

// For blocks of up to `samples` camera rays.  Where every hit spawns at most two rays (reflection + refraction) the
// capacities are the worst case -- 2^(level-1) per camera ray at a level, 2^levels - 1 hits altogether -- up to four
// levels; a glossy hit spawns 16 + 16 (main.cpp:177-257), which no capacity bounds: three such levels in a row get
// room (32^2 per camera ray), and what outgrows that is split (see rt_render_rows_dev).
QueueCaps queue_caps(const rt_context* ctx, const rt_render_params* p, unsigned long long samples)
{
    const int levels = ctx->has_bounce ? std::min(p->tracing_depth, rtr::MAX_LEVELS) : 1;
    const bool glossy = ctx->has_glossy && levels > 1;
    const unsigned long long cap_hits = samples << (glossy ? 10 : std::max(0, std::min(levels - 1, 3)));
    const unsigned long long cap_gq = glossy ? cap_hits + cap_hits / 8 : samples * ((1ull << std::min(levels, 4)) - 1);
    QueueCaps q;
    // (shards fill nearly evenly -- producers pick them round-robin -- hence the slack)
    q.shard_cap = (((cap_hits + rtq::SHARDS - 1) / rtq::SHARDS + 16384) + 63) & ~63ull;
    q.gq_cap = (((cap_gq + rtq::SHARDS - 1) / rtq::SHARDS + 16384) + 63) & ~63ull;
    q.retry = glossy || levels > 4;
    // RT_SHARD_CAP=<slots>: smaller queues, to exercise the overflow handling (development aid)
    if (const char* e = getenv("RT_SHARD_CAP")) q.shard_cap = q.gq_cap = std::max<unsigned long long>(64, (strtoull(e, nullptr, 10) + 63) & ~63ull);
    return q;
}

template <class R>
int render_block(rt_context* ctx, const rtr::RenderArgs& a, const rt_render_params* p, const QueueCaps& caps, double* d_acc, rtr::RenderCounters* ctr_copy,
                 hipEvent_t ev_begin, hipEvent_t ev_end)
{
    // (a scene of diffuse surfaces only queues no rays: one level, and the deeper launches are not made)
    const int levels = ctx->has_bounce ? std::min(p->tracing_depth, rtr::MAX_LEVELS) : 1;
    const unsigned long long shard_cap = caps.shard_cap, gq_cap = caps.gq_cap;
    const unsigned long long cap_total = shard_cap * rtq::SHARDS, gq_total = gq_cap * rtq::SHARDS;
    int rc;
    if ((rc = ctx->hits_a.reserve(cap_total * sizeof(rtr::HitRec)))) return rc;
    if ((rc = ctx->hits_b.reserve(cap_total * sizeof(rtr::HitRec)))) return rc;
    if ((rc = ctx->rays.reserve(cap_total * sizeof(rtr::RayRec)))) return rc;
    if ((rc = ctx->gq.reserve(gq_total * sizeof(rtg::GatherRec)))) return rc;
    if ((rc = reserve_redo(ctx, gq_total))) return rc;
    TraceGeom g;
    if ((rc = trace_geometry<R>(ctx, ctx->n_nodes, &g, 16))) return rc;
    rtr::RenderCounters* ctr = (rtr::RenderCounters*)ctx->counters.p;
    HIP_TRY(hipMemsetAsync(ctr, 0, sizeof(rtr::RenderCounters), ctx->stream));
    if (ev_begin) HIP_TRY(hipEventRecord(ev_begin, ctx->stream));
    const bool tri_only = view<R>(ctx).tri_only;
    // (the instrumented kernels exist for the double instantiation without depth of field, which is what measurements run)
    const bool count = p->count_work && sizeof(R) == sizeof(double) && a.dof_samples <= 1;
#define RT_LEVELS(TRI, MODE)                                                                                                  \
    (count ? launch_render_levels<double, TRI, MODE, true>(ctx, a, g, levels, shard_cap, gq_cap, ctr, d_acc)                   \
           : launch_render_levels<R, TRI, MODE, false>(ctx, a, g, levels, shard_cap, gq_cap, ctr, d_acc))
    if (tri_only) {
        if (g.mode == rtk::TRAV_NODES) rc = RT_LEVELS(true, rtk::TRAV_NODES);
        else if (g.mode == rtk::TRAV_RETEST) rc = RT_LEVELS(true, rtk::TRAV_RETEST);
        else rc = RT_LEVELS(true, rtk::TRAV_PLAIN);
    } else {
        if (g.mode == rtk::TRAV_NODES) rc = RT_LEVELS(false, rtk::TRAV_NODES);
        else if (g.mode == rtk::TRAV_RETEST) rc = RT_LEVELS(false, rtk::TRAV_RETEST);
        else rc = RT_LEVELS(false, rtk::TRAV_PLAIN);
    }
#undef RT_LEVELS
    if (rc) return rc;
    if (ev_end) HIP_TRY(hipEventRecord(ev_end, ctx->stream));
    if (p->use_photon_map && (ctx->pm[0].n > 0 || ctx->pm[1].n > 0)) {
        if (caps.retry && ctr_copy) {
            // Queues that are a budget: a part whose rays outgrew them is rendered again in pieces, so its gather would be
            // work thrown away, over a request queue whose fill counters exceed its capacity.  The caller waits for every
            // part anyway; wait here, and leave the gather out when the part overflowed.
            HIP_TRY(hipMemcpyAsync(ctr_copy, ctr, sizeof(*ctr_copy), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctr_copy->overflow_queue || ctr_copy->overflow_stack) return RT_OK;
        }
        rtg::GatherParams gp;
        gp.front_cap = gather_front_cap();
        gp.k = p->photon_use;
        gp.r2 = (float)(p->photon_dist * p->photon_dist);
        gp.norm = (double)(p->photon_num / 16) * 3.1415926;
        gp.photon_num = (double)p->photon_num;
        gp.seed = p->rng_seed;
        if ((rc = launch_gather(ctx, (const rtg::GatherRec*)ctx->gq.p, &ctr->gather, gq_cap, &ctr->next_window, &ctr->n_redo, &ctr->next_gather, ctr->tile_progress, gp,
                                d_acc, (uint32_t*)ctx->redo.p, ctx->redo.bytes / sizeof(uint32_t))))
            return rc;
    }
    // (a copy of the counters for the caller: into device memory where the caller does not wait for this block -- a copy
    // to the host would hold up the launches of the next one)
    if (ctr_copy) HIP_TRY(hipMemcpyAsync(ctr_copy, ctr, sizeof(*ctr_copy), hipMemcpyDefault, ctx->stream));
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_render_rows_dev(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end, double* d_rgb,
                       uint8_t* d_bgr8, rt_stats* stats)
{
    return guarded([&]() -> int {
    if (!ctx || !cam) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(p);
    if (rc) return rc;
    if (!ctx->has_scene) return fail(RT_ERR_STATE, "no scene uploaded");
    int part_rows = p->band_rows > 0 ? p->height / p->n_parts : p->height;
    if (row_begin < 0 || row_end > part_rows || row_begin >= row_end) return fail(RT_ERR_INVALID, "rows [%d,%d) outside [0,%d)", row_begin, row_end, part_rows);
    if (p->aadepth < 1 || p->aadepth > 16) return fail(RT_ERR_INVALID, "aadepth %d", p->aadepth);
    if (p->dof_samples < 1 || p->dof_samples > 4096) return fail(RT_ERR_INVALID, "dof_samples %d outside [1,4096]", p->dof_samples);
    if (p->tracing_depth < 1 || p->tracing_depth > rtr::MAX_LEVELS) return fail(RT_ERR_INVALID, "tracing_depth %d outside [1,%d]", p->tracing_depth, rtr::MAX_LEVELS);
    if (p->use_photon_map && (p->photon_use < 1 || p->photon_use > rtg::MAX_K)) return fail(RT_ERR_LIMIT, "photon_use %d outside [1,%d]", p->photon_use, rtg::MAX_K);
    if ((rc = use_device(ctx))) return rc;

    rtr::RenderArgs a;
    for (int i = 0; i < 3; i++) { a.cam_pos[i] = cam->pos[i]; a.cam_dir[i] = cam->dir[i]; a.cam_right[i] = cam->right[i]; a.cam_down[i] = cam->down[i]; }
    a.width = p->width; a.height = p->height;
    a.band_rows = p->band_rows; a.n_parts = p->n_parts; a.part = p->part;
    a.aadepth = p->aadepth;
    a.accuracy = p->accuracy;
    a.head_quirk = p->head_quirk; a.use_photon_map = p->use_photon_map;
    a.obj_live = p->use_photon_map ? nullptr : (const unsigned char*)ctx->obj_live.p;
    a.seed = p->rng_seed;
    a.dof_samples = p->dof_samples; a.focal_len = p->focal_len; a.aperture = p->aperture;

    // row chunks bound the queue memory: at most ~4M distinct camera rays per chunk (16M when no surface can queue a ray)
    long long max_samples = ctx->has_glossy ? 32ll << 10 : (ctx->has_bounce ? 4ll << 20 : 16ll << 20);
    if (ctx->has_bounce && !ctx->has_glossy) {
        // where memory is plentiful a 3840 x 2160 frame is one chunk (its queues take about 30 GB): every chunk pays the
        // ramp-up and the tail of each of its launches
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + ctx->hits_a.bytes + ctx->hits_b.bytes + ctx->rays.bytes + ctx->gq.bytes > (96ull << 30))
            max_samples = 17ll << 19;
    }
    const long long per_pixel = p->dof_samples > 1 ? (long long)p->aadepth * p->aadepth * p->dof_samples : p->aadepth;
    int chunk_rows = (int)std::max<long long>(8, (max_samples / ((long long)p->width * per_pixel)) / 8 * 8);
    if ((long long)p->width * chunk_rows >= (1ll << 25)) return fail(RT_ERR_LIMIT, "width %d: a block of 8 rows exceeds 2^25 pixels", p->width);
    const size_t n_chunks = (size_t)((row_end - row_begin + chunk_rows - 1) / chunk_rows);
    const QueueCaps caps = queue_caps(ctx, p, (unsigned long long)p->width * std::min(chunk_rows, row_end - row_begin) * per_pixel);
    std::vector<rtr::RenderCounters> hc(caps.retry ? 0 : n_chunks);
    if (stats && !caps.retry && (rc = ctx->ctr_log.reserve(n_chunks * sizeof(rtr::RenderCounters)))) return rc;
    rtr::RenderCounters* const d_log = (rtr::RenderCounters*)ctx->ctr_log.p;
    if (stats || caps.retry)
        while (ctx->chunk_ev.size() < 2 * (caps.retry ? 1 : n_chunks)) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            ctx->chunk_ev.push_back(e);
        }
    rt_stats total;
    memset(&total, 0, sizeof total);
    auto add_counters = [&](const rtr::RenderCounters& c) {
        total.secondary_rays += c.secondary_rays;
        for (int sh = 0; sh < rtq::SHARDS; sh++) total.gather_queries += c.gather.n[sh][1];
        total.gather_single += c.n_redo;
        total.inner_visits += c.work_inner;
        total.prim_tests += c.work_prims;
    };
    HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    size_t ci = 0;
    a.col_begin = 0; a.col_end = p->width;
    for (int r0 = row_begin; r0 < row_end; r0 += chunk_rows, ci++) {
        int r1 = std::min(row_end, r0 + chunk_rows);
        size_t npx = (size_t)p->width * (r1 - r0);
        size_t off = (size_t)p->width * (r0 - row_begin);
        double* acc;
        if (d_rgb) {
            acc = d_rgb + off * 3;
        } else {
            if ((rc = ctx->acc.reserve(npx * 3 * sizeof(double)))) return rc;
            acc = (double*)ctx->acc.p;
        }
        if (!caps.retry) {
            a.row_begin = r0; a.row_end = r1;
            hipEvent_t eb = stats ? ctx->chunk_ev[2 * ci] : nullptr, ee = stats ? ctx->chunk_ev[2 * ci + 1] : nullptr;
            rc = p->precision == RT_F64 ? render_block<double>(ctx, a, p, caps, acc, stats ? d_log + ci : nullptr, eb, ee)
                                        : render_block<float>(ctx, a, p, caps, acc, stats ? d_log + ci : nullptr, eb, ee);
            if (rc) return rc;
        } else {
            // The queues are a budget here.  A part whose rays outgrow them is split -- rows first, then columns, down
            // to one tile of 8 x 8 pixels -- and rendered again: its first kernel starts the pixels' sums afresh, so
            // nothing of the attempt that overflowed stays.  Each part is waited for, to learn how it went.
            struct Part { int r0, r1, c0, c1; };
            std::vector<Part> todo{{r0, r1, 0, p->width}};
            while (!todo.empty()) {
                const Part q = todo.back();
                todo.pop_back();
                a.row_begin = q.r0; a.row_end = q.r1; a.col_begin = q.c0; a.col_end = q.c1;
                double* const acc_part = acc + (size_t)p->width * (q.r0 - r0) * 3;
                rtr::RenderCounters c;
                rc = p->precision == RT_F64 ? render_block<double>(ctx, a, p, caps, acc_part, &c, ctx->chunk_ev[0], ctx->chunk_ev[1])
                                            : render_block<float>(ctx, a, p, caps, acc_part, &c, ctx->chunk_ev[0], ctx->chunk_ev[1]);
                if (rc) return rc;
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                if (c.overflow_stack) return overflow_error(ctx, "traversal stack overflow");
                if (c.overflow_queue) {
                    unsigned long long latch[2];
                    if ((rc = take_overflow_latch(ctx, latch))) return rc;       // (handled here: not an error of the frame)
                    const int rows = q.r1 - q.r0, cols = q.c1 - q.c0;
                    if (rows > 8) {
                        const int mid = q.r0 + (rows / 2 + 7) / 8 * 8;
                        todo.push_back({mid, q.r1, q.c0, q.c1});
                        todo.push_back({q.r0, mid, q.c0, q.c1});
                    } else if (cols > 8) {
                        const int mid = q.c0 + (cols / 2 + 7) / 8 * 8;
                        todo.push_back({q.r0, q.r1, mid, q.c1});
                        todo.push_back({q.r0, q.r1, q.c0, mid});
                    } else {
                        return fail(RT_ERR_LIMIT, "the rays of one tile of 8 x 8 pixels overflow the queues (%llu slots a level): lower tracing_depth",
                                    caps.shard_cap * rtq::SHARDS);
                    }
                    continue;
                }
                add_counters(c);
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, ctx->chunk_ev[0], ctx->chunk_ev[1]));
                total.trace_kernel_ms += ms;
            }
            a.col_begin = 0; a.col_end = p->width;
        }
        int threads = 256;
        hipLaunchKernelGGL(rtr::resolve_kernel, dim3((unsigned)((npx + threads - 1) / threads)), dim3(threads), 0, ctx->stream, (long long)npx, p->aadepth,
                           acc, d_rgb ? acc : nullptr, d_bgr8 ? d_bgr8 + off * 3 : nullptr);
        HIP_TRY(hipGetLastError());
        total.primary_rays += (uint64_t)npx * (uint64_t)per_pixel;
    }
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (stats) {
        if (!hc.empty()) HIP_TRY(hipMemcpyAsync(hc.data(), d_log, hc.size() * sizeof(rtr::RenderCounters), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < hc.size(); i++) {
            if (hc[i].overflow_stack) return overflow_error(ctx, "traversal stack overflow");
            if (hc[i].overflow_queue) return overflow_error(ctx, "bounce/gather queue overflow");
            add_counters(hc[i]);
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        total.kernel_ms = ms;
        for (size_t i = 0; i < hc.size(); i++) {                     // trace + shade launches, summed over the row chunks
            HIP_TRY(hipEventElapsedTime(&ms, ctx->chunk_ev[2 * i], ctx->chunk_ev[2 * i + 1]));
            total.trace_kernel_ms += ms;
        }
        *stats = total;
    }
    return RT_OK;
    });
}

int rt_render_rows(rt_context* ctx, const rt_camera* cam, const rt_render_params* p, int row_begin, int row_end, double* rgb, uint8_t* bgr8,
                   rt_stats* stats)
{
    return guarded([&]() -> int {
    if (!ctx || !cam || (!rgb && !bgr8)) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(p);
    if (rc) return rc;
    int part_rows = p->band_rows > 0 ? p->height / p->n_parts : p->height;
    if (row_begin < 0 || row_end > part_rows || row_begin >= row_end) return fail(RT_ERR_INVALID, "rows [%d,%d) outside [0,%d)", row_begin, row_end, part_rows);
    if ((rc = use_device(ctx))) return rc;
    size_t npx = (size_t)p->width * (row_end - row_begin);
    if ((rc = ctx->st_org.reserve(npx * 3 * sizeof(double)))) return rc;
    if ((rc = ctx->st_aux.reserve(npx * 3))) return rc;
    rt_stats local;
    rc = rt_render_rows_dev(ctx, cam, p, row_begin, row_end, (double*)ctx->st_org.p, (uint8_t*)ctx->st_aux.p, &local);
    if (rc) return rc;
    if (rgb) HIP_TRY(hipMemcpyAsync(rgb, ctx->st_org.p, npx * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (bgr8) HIP_TRY(hipMemcpyAsync(bgr8, ctx->st_aux.p, npx * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (stats) *stats = local;
    return RT_OK;
    });
}

// ---------------------------------------------------------------- frame assembly over the GPUs of one process
int rt_gather_rows(rt_context* const* ctxs, const void* const* d_parts, int n_parts, int band_rows, int height, size_t row_bytes, void* d_frame)
{
    return guarded([&]() -> int {
    if (!ctxs || !d_parts || !d_frame || n_parts < 1 || band_rows < 1 || height < 1 || row_bytes == 0) return fail(RT_ERR_INVALID, "bad argument");
    if (height % (band_rows * n_parts) != 0)
        return fail(RT_ERR_INVALID, "row interleave band_rows %d n_parts %d does not tile height %d", band_rows, n_parts, height);
    for (int r = 0; r < n_parts; r++)
        if (!ctxs[r] || !d_parts[r]) return fail(RT_ERR_INVALID, "part %d: null context or buffer", r);
    const int dst_device = ctxs[0]->device;
    const size_t band_bytes = (size_t)band_rows * row_bytes;
    const size_t n_bands = (size_t)height / ((size_t)band_rows * n_parts);        // per part
    for (int r = 0; r < n_parts; r++) {
        rt_context* c = ctxs[r];
        int rc = use_device(c);
        if (rc) return rc;
        bool direct = true;
        if (c->device != dst_device) {
            // a direct xGMI copy where the source device can reach the frame's memory
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, c->device, dst_device));
            if (can) {
                hipError_t e = hipDeviceEnablePeerAccess(dst_device, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                else if (e != hipSuccess) return fail(RT_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d) failed: %s", c->device, dst_device, hipGetErrorString(e));
            }
            direct = can != 0;
        }
        // band b of part r is band b * n_parts + r of the frame: one strided copy per part, in the order of the
        // stream that traced the part
        if (direct) {
            HIP_TRY(hipMemcpy2DAsync((char*)d_frame + (size_t)r * band_bytes, band_bytes * n_parts, d_parts[r], band_bytes, band_bytes, n_bands, hipMemcpyDeviceToDevice,
                                     c->stream));
        } else {
            // no peer access between the two devices: through pinned host memory (down on the part's stream, which waits for
            // its kernels; up on the frame's device)
            void* host = nullptr;
            HIP_TRY(hipHostMalloc(&host, band_bytes * n_bands, hipHostMallocDefault));
            hipError_t e = hipMemcpyAsync(host, d_parts[r], band_bytes * n_bands, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipSetDevice(dst_device);
            if (e == hipSuccess) e = hipMemcpy2D((char*)d_frame + (size_t)r * band_bytes, band_bytes * n_parts, host, band_bytes, band_bytes, n_bands, hipMemcpyHostToDevice);
            (void)hipHostFree(host);
            if (e != hipSuccess) return fail(RT_ERR_HIP, "staged copy of part %d (device %d -> %d) failed: %s", r, c->device, dst_device, hipGetErrorString(e));
        }
    }
    for (int r = 0; r < n_parts; r++) {
        int rc = use_device(ctxs[r]);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctxs[r]->stream));
    }
    return use_device(ctxs[0]);
    });
}

// ---------------------------------------------------------------- one process per GPU: the frame over RCCL
}  // extern "C"

namespace {
// RCCL's entry points, bound when first needed: the library itself does not link against RCCL, so processes that never
// gather across ranks load nothing, and a Python host keeps the one copy its torch brought along.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.ok) return RT_OK;
    if (!g_rccl.lib) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!g_rccl.lib) return fail(RT_ERR_STATE, "RCCL is not available: %s", dlerror());
#define RT_RCCL_SYM(field, sym)                                                                                   \
    if (!(g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, #sym)))) return fail(RT_ERR_STATE, "RCCL lacks " #sym)
    RT_RCCL_SYM(get_unique_id, ncclGetUniqueId);
    RT_RCCL_SYM(comm_init_rank, ncclCommInitRank);
    RT_RCCL_SYM(comm_destroy, ncclCommDestroy);
    RT_RCCL_SYM(group_start, ncclGroupStart);
    RT_RCCL_SYM(group_end, ncclGroupEnd);
    RT_RCCL_SYM(send, ncclSend);
    RT_RCCL_SYM(recv, ncclRecv);
    RT_RCCL_SYM(error_string, ncclGetErrorString);
#undef RT_RCCL_SYM
    g_rccl.ok = true;
    return RT_OK;
}
#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, g_rccl.error_string(r_));       \
    } while (0)
}  // namespace

struct rt_comm {
    rt_context* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    DevBuf stage;      // rank 0: the other ranks' parts as they arrive, compact
};
static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "the id travels as RT_COMM_ID_BYTES bytes");

extern "C" {

int rt_comm_unique_id(unsigned char id[RT_COMM_ID_BYTES])
{
    return guarded([&]() -> int {
    if (!id) return fail(RT_ERR_INVALID, "null argument");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(g_rccl.get_unique_id(&u));
    memcpy(id, &u, RT_COMM_ID_BYTES);
    return RT_OK;
    });
}

int rt_comm_create(rt_context* ctx, const unsigned char id[RT_COMM_ID_BYTES], int rank, int n_ranks, rt_comm** out)
{
    return guarded([&]() -> int {
    if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_ERR_INVALID, "bad argument");
    int rc = rccl_load();
    if (rc) return rc;
    if ((rc = use_device(ctx))) return rc;
    ncclUniqueId u;
    memcpy(&u, id, RT_COMM_ID_BYTES);
    rt_comm* c = new rt_comm;
    c->ctx = ctx; c->rank = rank; c->n_ranks = n_ranks;
    ncclResult_t r = g_rccl.comm_init_rank(&c->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(RT_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, n_ranks, g_rccl.error_string(r));
    }
    *out = c;
    return RT_OK;
    });
}

void rt_comm_destroy(rt_comm* comm)
{
    if (!comm) return;
    if (comm->ctx) (void)hipSetDevice(comm->ctx->device);
    if (comm->comm && g_rccl.ok) (void)g_rccl.comm_destroy(comm->comm);
    comm->stage.release();
    delete comm;
}

int rt_gather_frame(rt_comm* comm, const void* d_part, int band_rows, int height, size_t row_bytes, void* d_frame)
{
    return guarded([&]() -> int {
    if (!comm || !comm->ctx || !d_part || band_rows < 1 || height < 1 || row_bytes == 0) return fail(RT_ERR_INVALID, "bad argument");
    const int n = comm->n_ranks;
    if (height % (band_rows * n) != 0) return fail(RT_ERR_INVALID, "row interleave band_rows %d n_parts %d does not tile height %d", band_rows, n, height);
    if (comm->rank == 0 && !d_frame) return fail(RT_ERR_INVALID, "rank 0 needs the frame");
    rt_context* ctx = comm->ctx;
    int rc = use_device(ctx);
    if (rc) return rc;
    const size_t band_bytes = (size_t)band_rows * row_bytes;
    const size_t n_bands = (size_t)height / ((size_t)band_rows * n);      // per part
    const size_t part_bytes = band_bytes * n_bands;
    if (comm->rank == 0 && n > 1 && (rc = comm->stage.reserve(part_bytes * (size_t)(n - 1)))) return rc;
    if (n > 1) {
        // one grouped exchange: every peer sends its part straight to rank 0 (point-to-point over its own link)
        RCCL_TRY(g_rccl.group_start());
        if (comm->rank == 0) {
            for (int r = 1; r < n; r++) RCCL_TRY(g_rccl.recv((char*)comm->stage.p + part_bytes * (size_t)(r - 1), part_bytes, ncclUint8, r, comm->comm, ctx->stream));
        } else {
            RCCL_TRY(g_rccl.send(d_part, part_bytes, ncclUint8, 0, comm->comm, ctx->stream));
        }
        RCCL_TRY(g_rccl.group_end());
    }
    if (comm->rank == 0) {
        // band b of part r is band b * n + r of the frame: one strided copy per part, behind the exchange on the stream
        for (int r = 0; r < n; r++) {
            const void* src = r == 0 ? d_part : (const void*)((const char*)comm->stage.p + part_bytes * (size_t)(r - 1));
            HIP_TRY(hipMemcpy2DAsync((char*)d_frame + (size_t)r * band_bytes, band_bytes * n, src, band_bytes, band_bytes, n_bands, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    return RT_OK;
    });
}

// ---------------------------------------------------------------- host I/O
int rt_save_bmp(const char* path, int w, int h, int dpi, const double* rgb)
{
    return guarded([&]() -> int {
    if (!path || !rgb || w <= 0 || h <= 0) return fail(RT_ERR_INVALID, "bad argument");
    std::vector<uint8_t> bgr((size_t)w * h * 3);
    rth::quantize_bgr8((int64_t)w * h, rgb, bgr.data());
    return rth::save_bmp_bgr8(path, w, h, dpi, bgr.data()) ? fail(RT_ERR_IO, "cannot write %s", path) : RT_OK;
    });
}
int rt_save_bmp_bgr8(const char* path, int w, int h, int dpi, const uint8_t* bgr)
{
    return guarded([&]() -> int {
    if (!path || !bgr || w <= 0 || h <= 0) return fail(RT_ERR_INVALID, "bad argument");
    return rth::save_bmp_bgr8(path, w, h, dpi, bgr) ? fail(RT_ERR_IO, "cannot write %s", path) : RT_OK;
    });
}

}  // extern "C"
