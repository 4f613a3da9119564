// hsr_api.hip — the C ABI of libhsr_rast.so (see include/hsr_rasterizer.h): argument validation,
// state-buffer carving and stage orchestration.  Host code only; kernels live in the other units.
//
// Stage order of one forward (reference Rasterizer::forward[_semantic], rasterizer_impl.cu:198-345,
// :460-610):   preprocess (+ per-block tile-count sums) -> scan of the block sums -> 4-byte D2H of
// num_rendered (the reference's blocking cudaMemcpy, rasterizer_impl.cu:285/:548; it sizes the
// binning buffer) -> key emission (finishes the scan in-block) -> radix sort -> tile ranges -> tile
// render.  One backward: zero the atomically-accumulated sums -> tile backward -> fused per-Gaussian
// backward.  The library holds no state between calls; the reference's per-call cudaMalloc/cudaFree
// scratch in backward_semantic (rasterizer_impl.cu:673-701) has no counterpart.
#include <stdarg.h>

#include <chrono>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/hsr_rasterizer.h"
#include "hsr_tile_common.h"

namespace {

thread_local char g_err[512] = "";

template <typename T>
inline void take(char*& p, T*& out, size_t count)
{
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255;
    out = reinterpret_cast<T*>(a);
    p = reinterpret_cast<char*>(out + count);
}

uint32_t higher_msb(uint32_t n)  // reference getHigherMsb, rasterizer_impl.cu:35-50
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// backward accumulation mode: 0 packed per-Gaussian rows (default), 1 per-instance rows (experiment: ablate build only),
// 2 legacy separate arrays.  Initialised from HSR_BWD_IMPL=legacy (rows: ablate build), changed with hsr_set_backward_mode().
int g_bwd_mode = -1;
int backward_mode()
{
    if (g_bwd_mode < 0) {
        const char* e = getenv("HSR_BWD_IMPL");
        g_bwd_mode = (e && !strcmp(e, "legacy")) ? 2 : 0;
#ifdef HSR_ABLATE
        if (e && !strcmp(e, "rows")) g_bwd_mode = 1;
#endif
    }
    return g_bwd_mode;
}
bool rows_mode_requested() { return backward_mode() == 1; }

// semantic -> alpha gradient: 0 as the reference computes it (none: backward.cu:834-845 reads a scratch nothing wrote), 1 exact
// (opt-in; HSR_SEMANTIC_ALPHA=exact or hsr_set_semantic_alpha_mode)
int g_sem_alpha_mode = -1;
int semantic_alpha_mode()
{
    if (g_sem_alpha_mode < 0) {
        const char* e = getenv("HSR_SEMANTIC_ALPHA");
        g_sem_alpha_mode = (e && !strcmp(e, "exact")) ? 1 : 0;
    }
    return g_sem_alpha_mode;
}

int acquire(hsr_buffer* b, size_t need, const char* what, char** out)
{
    if (!b) {
        hsr_set_error("%s buffer descriptor is NULL", what);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!(b->ptr && b->capacity >= need)) {
        if (!b->grow) {
            hsr_set_error("%s buffer too small: %zu bytes needed, %zu available, no grow callback", what, need, b->capacity);
            return HSR_ERR_BUFFER_TOO_SMALL;
        }
        char* p = b->grow(need, b->user);
        if (!p) {
            hsr_set_error("%s buffer grow callback failed for %zu bytes", what, need);
            return HSR_ERR_BUFFER_TOO_SMALL;
        }
        b->ptr = p;
        b->capacity = need;
    }
    *out = b->ptr;
    return HSR_OK;
}

// Host-mapped read-back slot of the calling thread for ONE device: [0] num_rendered, [1] sequence number of the call that
// wrote it, [2] prefilter-violation flag.  One slot per (thread, device): a thread that renders on two devices gets two,
// each allocated (portable + mapped) while its device is current; freed when the thread ends (thread-local destructors of
// the main thread run at the start of exit(), before the HIP runtime's own teardown).
struct PinnedCounter {
    uint32_t* host = nullptr;
    uint32_t* dev = nullptr;   // the device's view of it
    uint32_t seq = 0;
    hipEvent_t event = nullptr;
    double wait_ema_us = 0.0;   // how long the host recently waited for the count: sizes the sleep in front of the poll
    ~PinnedCounter()
    {
        if (event) (void)hipEventDestroy(event);
        if (host) (void)hipHostFree(host);
    }
};
constexpr int HSR_MAX_DEVICES = 64;
thread_local PinnedCounter g_counters_by_device[HSR_MAX_DEVICES];
thread_local PinnedCounter* g_pc = nullptr;      // the current call's slot (set by counter_buffer())
#define g_pinned (g_pc->host)
#define g_pinned_dev (g_pc->dev)
#define g_counter_seq (g_pc->seq)
#define g_counter_event (g_pc->event)

// Non-blocking forward (hsr_forward_arm_async): the count lands in a slot of a per-device ring of host-mapped words that is
// process-wide — the ticket may be resolved on another thread (autograd's backward thread) than the one that called the forward.
struct AsyncRing {
    uint32_t* host = nullptr;
    uint32_t* dev = nullptr;
    uint32_t seq = 0;
};
constexpr int HSR_ASYNC_SLOTS = 256;   // forwards in flight per device before a slot is reused
std::mutex g_async_mu;
AsyncRing g_async_ring[64];
thread_local hsr_ticket* g_armed_ticket = nullptr;

// ---- optional per-stage timing with HIP events (hsr_profile_*) ----
struct StageEvents {
    int stage;
    hipEvent_t start, stop;
};
bool g_prof_on = false;
unsigned g_prof_mask = ~0u;
std::mutex g_prof_mu;
std::vector<StageEvents> g_prof_pending;
std::vector<hipEvent_t> g_prof_free;
hsr_profile g_prof_acc;

hipEvent_t prof_get_event()
{
    if (!g_prof_free.empty()) {
        hipEvent_t e = g_prof_free.back();
        g_prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
struct StageTimer {
    int stage;
    hipStream_t stream;
    hipEvent_t start = nullptr, stop = nullptr;
    StageTimer(int st, hipStream_t s) : stage(st), stream(s)
    {
        if (!g_prof_on || !((g_prof_mask >> st) & 1u)) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        start = prof_get_event();
        stop = prof_get_event();
        if (start) (void)hipEventRecord(start, stream);
    }
    ~StageTimer()
    {
        if (!start || !stop) return;
        (void)hipEventRecord(stop, stream);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_pending.push_back({stage, start, stop});
    }
};

// num_rendered read-back in two steps: the copy is enqueued right behind the scan, the host waits only after it has
// enqueued the work that does not depend on the value (an event, not a stream sync, so that work keeps the GPU busy)
int counter_buffer()
{
    int d = 0;
    HSR_HIP_CHECK(hipGetDevice(&d));
    if (d < 0 || d >= HSR_MAX_DEVICES) {
        hsr_set_error("device index %d out of range", d);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    g_pc = &g_counters_by_device[d];
    if (!g_pinned) {
        HSR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&g_pinned), 64, hipHostMallocPortable | hipHostMallocMapped));
        g_pinned[0] = g_pinned[1] = g_pinned[2] = 0;
        HSR_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&g_pinned_dev), g_pinned, 0));
    }
    return HSR_OK;
}
int read_counter_begin(const uint32_t* dev, hipStream_t stream)
{
    int rc;
    if ((rc = counter_buffer()) != HSR_OK) return rc;
    if (!g_counter_event) HSR_HIP_CHECK(hipEventCreateWithFlags(&g_counter_event, hipEventDisableTiming));
    HSR_HIP_CHECK(hipMemcpyAsync(g_pinned, dev, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HSR_HIP_CHECK(hipMemcpyAsync(g_pinned + 2, dev + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));   // prefilter flag
    HSR_HIP_CHECK(hipEventRecord(g_counter_event, stream));
    return HSR_OK;
}
thread_local double g_host_wait_ms = 0.0;
thread_local int g_last_per_tile = 0;   // num_rendered / tiles of this thread's previous forward: picks the per-tile sort's launch shape
int read_counter_end(uint32_t* host_out)
{
    const auto t0 = std::chrono::steady_clock::now();
    HSR_HIP_CHECK(hipEventSynchronize(g_counter_event));
    g_host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *host_out = g_pinned[0];
    return HSR_OK;
}
// Direct-binning path: bin_hist_kernel's last workgroup stores {num_rendered, seq} into the host-mapped buffer itself (no copy
// kernel, no event): the host polls the sequence number.  A launch that never completes is caught by a stream query after 10 s.
int poll_counter(uint32_t seq, hipStream_t stream, uint32_t* host_out, volatile uint32_t* slot = nullptr)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (slot) {   // a ring slot of the non-blocking forward (may be resolved on another thread: no per-thread back-off state)
        unsigned spins = 0;
        uint32_t cur;
        while ((cur = __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE)) != seq) {
            // sequence numbers of a device only grow: a LATER one in this slot means HSR_ASYNC_SLOTS further non-blocking forwards were
            // started on the device before this one was resolved, and its count has been overwritten
            if (cur != 0 && (int32_t)(cur - seq) > 0) {
                hsr_set_error("non-blocking forward: this call's num_rendered was overwritten — more than %d non-blocking forwards were "
                              "started on the device before it was resolved (resolve or drop earlier ones first)", HSR_ASYNC_SLOTS - 1);
                return HSR_ERR_INVALID_ARGUMENT;
            }
            if ((++spins & 0xFFFu) == 0) {
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (el > 10.0) {
                    const hipError_t e = hipStreamSynchronize(stream);
                    if (e != hipSuccess || __atomic_load_n(&slot[1], __ATOMIC_ACQUIRE) != seq) {
                        hsr_set_error("num_rendered never arrived (%s)", hipGetErrorString(e));
                        return HSR_ERR_HIP;
                    }
                }
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        g_host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *host_out = __atomic_load_n(&slot[0], __ATOMIC_RELAXED);
        return HSR_OK;
    }
    // Back-off: the count arrives when the kernels queued in front of it (typically the previous step's backward) have run —
    // a few hundred microseconds in a training loop — and a `pause` loop would hold a core at 100 % for all of it.  Sleep
    // through the first half of the recently observed wait (an average over the last calls), then poll; short waits (< 0.15
    // ms: the count is already there or nearly) are polled from the start.
    if (g_pc->wait_ema_us > 150.0 && __atomic_load_n(&g_pinned[1], __ATOMIC_ACQUIRE) != seq) {
        timespec ts{0, (long)(g_pc->wait_ema_us * 0.5 * 1000.0)};
        if (ts.tv_nsec > 2000000L) ts.tv_nsec = 2000000L;
        nanosleep(&ts, nullptr);
    }
    unsigned spins = 0;
    while (__atomic_load_n(&g_pinned[1], __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xFFFu) == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > 10.0) {
                const hipError_t e = hipStreamSynchronize(stream);
                if (e != hipSuccess || __atomic_load_n(&g_pinned[1], __ATOMIC_ACQUIRE) != seq) {
                    hsr_set_error("num_rendered never arrived (%s)", hipGetErrorString(e));
                    return HSR_ERR_HIP;
                }
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    const double waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g_host_wait_ms += waited_ms;
    g_pc->wait_ema_us = 0.75 * g_pc->wait_ema_us + 0.25 * waited_ms * 1000.0;
    *host_out = __atomic_load_n(&g_pinned[0], __ATOMIC_RELAXED);
    return HSR_OK;
}

struct FwdIn {
    int P, D, M, K, semantic, W, H, prefiltered, debug;
    const float *background, *means3D, *shs, *colors_precomp, *semantics, *opacities, *scales, *rotations, *cov3D_precomp,
        *viewmatrix, *projmatrix, *cam_pos;
    float scale_modifier, tan_fovx, tan_fovy;
    float *out_color, *out_semantic, *out_depth, *out_median, *out_opacity, *out_mask;
    int* radii;
};

// a slot of the device's ring for one non-blocking forward
int ring_slot(uint32_t* seq_out, volatile uint32_t** host_out, uint32_t** dev_out, int* device_out)
{
    int d = 0;
    HSR_HIP_CHECK(hipGetDevice(&d));
    if (d < 0 || d >= 64) {
        hsr_set_error("device index %d out of range", d);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    std::lock_guard<std::mutex> lk(g_async_mu);
    AsyncRing& r = g_async_ring[d];
    if (!r.host) {
        HSR_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&r.host), HSR_ASYNC_SLOTS * 16, hipHostMallocPortable | hipHostMallocMapped));
        memset(r.host, 0, HSR_ASYNC_SLOTS * 16);
        HSR_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&r.dev), r.host, 0));
    }
    uint32_t s = ++r.seq;
    if (s == 0) s = ++r.seq;
    const uint32_t idx = s % HSR_ASYNC_SLOTS;
    *seq_out = s; *host_out = r.host + 4 * idx; *dev_out = r.dev + 4 * idx; *device_out = d;
    return HSR_OK;
}

int forward_impl(hsr_buffer* geometry, hsr_buffer* binning, hsr_buffer* image, const FwdIn& in, hipStream_t stream)
{
    const int P = in.P, W = in.W, H = in.H;
    hsr_ticket* ticket = g_armed_ticket;   // hsr_forward_arm_async: this call may return before num_rendered is known
    g_armed_ticket = nullptr;
    if (ticket) memset(ticket, 0, sizeof(*ticket));
    if (P < 0 || W <= 0 || H <= 0) {
        hsr_set_error("invalid sizes P=%d W=%d H=%d", P, W, H);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!in.out_color || !in.out_depth || !in.out_median || !in.out_opacity || (in.semantic && in.K > 0 && !in.out_semantic)) {
        hsr_set_error("an output image pointer is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (in.semantic && in.K < 0) {
        hsr_set_error("K must be >= 0");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    const size_t N = (size_t)W * H;
    if (P == 0) {
        // reference: P == 0 short-circuits with zero-filled outputs and rendered = 0 (rasterize_points.cu:294-295)
        HSR_HIP_CHECK(hipMemsetAsync(in.out_color, 0, sizeof(float) * 3 * N, stream));
        HSR_HIP_CHECK(hipMemsetAsync(in.out_depth, 0, sizeof(float) * N, stream));
        HSR_HIP_CHECK(hipMemsetAsync(in.out_median, 0, sizeof(float) * N, stream));
        HSR_HIP_CHECK(hipMemsetAsync(in.out_opacity, 0, sizeof(float) * N, stream));
        if (in.out_mask) HSR_HIP_CHECK(hipMemsetAsync(in.out_mask, 0, sizeof(float) * N, stream));
        if (in.semantic && in.K > 0) HSR_HIP_CHECK(hipMemsetAsync(in.out_semantic, 0, sizeof(float) * in.K * N, stream));
        return 0;
    }
    if (!in.means3D || !in.opacities || !in.viewmatrix || !in.projmatrix) {
        hsr_set_error("means3D, opacities, viewmatrix and projmatrix are required");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!in.colors_precomp && !(in.shs && in.cam_pos && in.M > 0)) {
        hsr_set_error("provide colors_precomp, or shs with M > 0 and cam_pos");  // rasterizer_impl.cu:246-249
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!in.cov3D_precomp && !(in.scales && in.rotations)) {
        hsr_set_error("provide cov3D_precomp, or scales and rotations");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (in.semantic && in.K > 0 && !in.semantics) {
        hsr_set_error("semantics_precomp is NULL but K=%d", in.K);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (in.shs && !in.colors_precomp && (in.D < 0 || in.D > 3 || in.M < (in.D + 1) * (in.D + 1))) {
        hsr_set_error("SH degree %d needs M >= %d coefficients (got %d)", in.D, (in.D + 1) * (in.D + 1), in.M);
        return HSR_ERR_INVALID_ARGUMENT;
    }

    int rc;
    char* gptr;
    char* iptr;
    if ((rc = acquire(geometry, hsr_required_geometry_bytes(P), "geometry", &gptr)) != HSR_OK) return rc;
    if ((rc = acquire(image, hsr_required_image_bytes(W, H), "image", &iptr)) != HSR_OK) return rc;
    GeomState g;
    ImgState im;
    hsr_carve_geom(gptr, P, &g);
    hsr_carve_img(iptr, W, H, &im);

    const int tiles_x = (W + HSR_TILE_X - 1) / HSR_TILE_X, tiles_y = (H + HSR_TILE_Y - 1) / HSR_TILE_Y;
    const int T = tiles_x * tiles_y;
    int* radii = in.radii ? in.radii : g.radii;

    PreprocessArgs pa;
    pa.P = P; pa.D = in.D; pa.M = in.M; pa.W = W; pa.H = H;
    pa.means3D = in.means3D; pa.scales = in.scales; pa.scale_modifier = in.scale_modifier; pa.rotations = in.rotations;
    pa.opacities = in.opacities; pa.shs = in.shs; pa.cov3D_precomp = in.cov3D_precomp; pa.colors_precomp = in.colors_precomp;
    pa.viewmatrix = in.viewmatrix; pa.projmatrix = in.projmatrix; pa.cam_pos = in.cam_pos;
    pa.tan_fovx = in.tan_fovx; pa.tan_fovy = in.tan_fovy;
    pa.focal_y = H / (2.0f * in.tan_fovy);  // rasterizer_impl.cu:226-227
    pa.focal_x = W / (2.0f * in.tan_fovx);
    pa.radii = radii; pa.prefiltered = in.prefiltered; pa.tiles_x = tiles_x; pa.tiles_y = tiles_y;

    {
        StageTimer tm(HSR_STAGE_FWD_PREPROCESS, stream);
        // counters[1]: set by preprocess_kernel when prefiltered is on and a point fails the frustum test
        if (in.prefiltered) HSR_HIP_CHECK(hipMemsetAsync(g.counters + 1, 0, sizeof(uint32_t), stream));
        hsr_launch_preprocess(pa, g, stream);
    }
    HSR_LAUNCH_CHECK(in.debug, stream);
    const int end_bit = 32 + (int)higher_msb((uint32_t)T);  // rasterizer_impl.cu:304-312
    // Default: count per tile, emit every instance straight into its tile's segment, then order each segment by
    // (depth, index) in LDS.  HSR_SORT_IMPL=radix or more than 8192 tiles take the emission-order + stable tile-bit
    // radix passes instead; both give the same sorted keys, values and ranges.  The count phase needs neither
    // num_rendered nor the binning buffer — it PRODUCES num_rendered (the scan over the per-block sums rides along in
    // bin_hist_kernel) — so it is enqueued before the host waits for anything (the reference idles the stream at this
    // point, rasterizer_impl.cu:285).
    static const bool force_radix = getenv("HSR_SORT_IMPL") && !strcmp(getenv("HSR_SORT_IMPL"), "radix");
    HsrBinPlan plan{0, 0};
    uint32_t* bin_scratch = reinterpret_cast<uint32_t*>(im.final_T);   // free until the render kernel writes it
    const bool binned = !force_radix && hsr_bin_plan(P, T, (size_t)W * H, &plan);
    static const bool no_speculation = hsr_ablate_env("HSR_NO_SPECULATION") != nullptr;
    const bool will_speculate = binned && !no_speculation && !in.debug && binning && binning->ptr && binning->capacity >= 4096;
    const bool go_async = ticket != nullptr && will_speculate;
    uint32_t seq = 0;
    volatile uint32_t* slot_host = nullptr;   // non-blocking forward: this call's slot of the device's ring
    uint32_t* slot_dev = nullptr;
    int slot_device = 0;
    if (binned) {
        if (go_async) {
            if ((rc = ring_slot(&seq, &slot_host, &slot_dev, &slot_device)) != HSR_OK) return rc;
        } else {
            if ((rc = counter_buffer()) != HSR_OK) return rc;
            seq = ++g_counter_seq;
            if (seq == 0) seq = ++g_counter_seq;
            slot_dev = g_pinned_dev;
        }
        StageTimer tm(HSR_STAGE_FWD_DUPLICATE, stream);
        hsr_launch_bin_count(plan, P, radii, tiles_x, tiles_y, g, bin_scratch, im.ranges, stream, slot_dev, seq);
    } else {
        StageTimer tm(HSR_STAGE_FWD_SCAN, stream);
        hsr_launch_scan_block_sums(P, g, stream);
    }
    HSR_LAUNCH_CHECK(in.debug, stream);

    if (!binned && (rc = read_counter_begin(g.counters, stream)) != HSR_OK) return rc;

    RenderFwdArgs ra;
    ra.W = W; ra.H = H; ra.K = in.semantic ? in.K : 0; ra.semantic = in.semantic;
    ra.ranges = im.ranges; ra.point_list = nullptr; ra.masks = nullptr; ra.means2D = g.means2D; ra.conic_opacity = g.conic_opacity;
    ra.depths = g.depths; ra.colors = in.colors_precomp ? in.colors_precomp : g.rgb; ra.semantics = in.semantics;
    ra.rec = g.rec;
    ra.final_T = im.final_T; ra.n_contrib = im.n_contrib; ra.median_pos = im.median_pos;
    ra.out_color = in.out_color; ra.out_semantic = in.out_semantic; ra.out_depth = in.out_depth;
    ra.out_median_depth = in.out_median; ra.out_opacity = in.out_opacity; ra.out_mask = in.out_mask;
    {
        static const int dbg = hsr_ablate_env("HSR_DEBUG_FLAGS") ? atoi(hsr_ablate_env("HSR_DEBUG_FLAGS")) : 0;
        ra.debug_flags = dbg & 16;   // 0 in the product build; ablate build: bit 4 = no sub-block culling
    }
    ra.bin = BinDevRef{nullptr, nullptr, 0};
    if (!in.semantic && !in.out_mask) {
        hsr_set_error("out_mask is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }

    // Speculative tail: when the caller's binning buffer already has room (diff_gaussian_rasterization sizes it from the
    // previous frame), the emit, per-tile sort and render kernels are enqueued BEFORE the host reads num_rendered back; they
    // derive their array bases from the device-side counter (BinDevRef) and do nothing if the buffer turns out too small.
    // The host then waits on an event that completed long ago instead of idling the stream while it wakes up and launches
    // the rest — the reference stalls here on every frame (rasterizer_impl.cu:285), and with a 0.65 ms render the host
    // side (~0.45 ms per fwd+bwd through Python) would otherwise be on the critical path.
    bool speculated = false;
    if (will_speculate) {
        const BinDevRef ref{static_cast<char*>(binning->ptr), reinterpret_cast<const uint32_t*>(g.counters), binning->capacity};
        BinState none{nullptr, nullptr, nullptr, nullptr, nullptr};
        {
            StageTimer tm(HSR_STAGE_FWD_DUPLICATE, stream);
            hsr_launch_bin_emit(plan, P, radii, tiles_x, tiles_y, g, bin_scratch, im.ranges, nullptr, stream, &ref);
        }
        {
            StageTimer tm(HSR_STAGE_FWD_SORT, stream);
            hsr_launch_tile_sort(none, T, P, im.ranges, stream, &ref, g_last_per_tile);
        }
        ra.bin = ref;
        {
            StageTimer tm(HSR_STAGE_FWD_RENDER, stream);
            hsr_launch_render_forward(ra, stream);
        }
        HSR_HIP_CHECK(hipGetLastError());
        speculated = true;
    }
    if (go_async) {
        // everything is enqueued; the count is read by hsr_forward_end (typically from the backward, when it has long arrived)
        ticket->seq = seq;
        ticket->device = slot_device;
        ticket->slot = slot_host;
        ticket->binning_base = static_cast<char*>(binning->ptr);
        ticket->binning_capacity = binning->capacity;
        ticket->prefiltered = in.prefiltered;
        return HSR_PENDING;
    }

    uint32_t R32 = 0;
    if ((rc = binned ? poll_counter(seq, stream, &R32) : read_counter_end(&R32)) != HSR_OK) return rc;
    if (R32 > 0x7fffffffu) {
        hsr_set_error("num_rendered %u overflows int", R32);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    const int R = (int)R32;
    if (in.prefiltered && __atomic_load_n(&g_pinned[2], __ATOMIC_RELAXED) != 0) {
        // the reference's device-side message and __trap() (auxiliary.h:156-160), as an error return instead of a dead queue
        hsr_set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (speculated) {
        BinState chk;
        const BinDevRef ref{static_cast<char*>(binning->ptr), nullptr, binning->capacity};
        g_last_per_tile = R / (T > 0 ? T : 1);
        if (hsr_bin_resolve(ref, R32, &chk)) return R;   // the kernels found the same layout: done
        // too small after all: the speculative kernels returned at once; the render kernel's final_T (= the count table)
        // was not touched either, but recount anyway to keep this rare path independent of that
        hsr_launch_bin_count(plan, P, radii, tiles_x, tiles_y, g, bin_scratch, im.ranges, stream, g_pinned_dev, seq);
        ra.bin = BinDevRef{nullptr, nullptr, 0};
    }

    char* bptr;
    if ((rc = acquire(binning, hsr_required_binning_bytes(R), "binning", &bptr)) != HSR_OK) return rc;
    BinState b;
    hsr_carve_bin(bptr, R, &b);

    if (binned) {   // also when R == 0: its first workgroup writes the tile ranges
        StageTimer tm(HSR_STAGE_FWD_DUPLICATE, stream);
        hsr_launch_bin_emit(plan, P, radii, tiles_x, tiles_y, g, bin_scratch, im.ranges, b.keys, stream);
    }
    if (binned) {
        HSR_LAUNCH_CHECK(in.debug, stream);
        StageTimer tm(HSR_STAGE_FWD_SORT, stream);
        if (R > 0) hsr_launch_tile_sort(b, T, P, im.ranges, stream, nullptr, R / (T > 0 ? T : 1));
    } else {
        // emit into the buffer pair from which the sort's ping-pong passes end in (keys, vals)
        const bool emit_sorted = hsr_sort_emit_into_sorted_buffers(end_bit);
        uint64_t* emit_k = emit_sorted ? b.keys : b.keys_unsorted;
        uint32_t* emit_v = emit_sorted ? b.vals : b.vals_unsorted;
        {
            StageTimer tm(HSR_STAGE_FWD_DUPLICATE, stream);
            BinState be = b;
            be.keys_unsorted = emit_k;
            be.vals_unsorted = emit_v;
            hsr_launch_duplicate(P, radii, tiles_x, tiles_y, g, be, im.ranges, stream);
        }
        HSR_LAUNCH_CHECK(in.debug, stream);
        // tile-bit radix passes -> tile ranges -> per-tile depth sort (ranges are a by-product of the sort)
        StageTimer tm(HSR_STAGE_FWD_SORT, stream);
        if ((rc = hsr_launch_sort_pairs(b, R, end_bit, T, im.ranges, stream)) != HSR_OK) return rc;
    }
    HSR_LAUNCH_CHECK(in.debug, stream);

    ra.point_list = b.vals;
    ra.masks = b.vals_unsorted;   // free after the sort: the staging phase leaves the sub-block masks there for the backward
    {
        StageTimer tm(HSR_STAGE_FWD_RENDER, stream);
        hsr_launch_render_forward(ra, stream);
    }
    HSR_LAUNCH_CHECK(in.debug, stream);
    return R;
}

struct BwdIn {
    int P, D, M, K, semantic, R, W, H, debug;
    const float *background, *means3D, *shs, *colors_precomp, *semantics, *scales, *rotations, *cov3D_precomp, *viewmatrix,
        *projmatrix, *campos;
    float scale_modifier, tan_fovx, tan_fovy;
    const int* radii;
    const char *geom, *binning, *img;
    const float *dL_dpix, *dL_dpix_sem, *dL_dpix_depth, *dL_dpix_median, *dL_dpix_opacity;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dsemantics, *dL_ddepth, *dL_dmean3D, *dL_dcov3D, *dL_dsh,
        *dL_dscale, *dL_drot;
    char* scratch;
    size_t scratch_bytes;
};

// the sums the tile kernel accumulates atomically start from zero (the reference relies on the
// caller's torch::zeros, rasterize_points.cu:378-388)
int zero_accumulators(const BwdIn& in, int K, hipStream_t stream)
{
    StageTimer tm(HSR_STAGE_BWD_ZERO, stream);
    const size_t P = (size_t)in.P;
    HSR_HIP_CHECK(hipMemsetAsync(in.dL_dmean2D, 0, sizeof(float) * 3 * P, stream));
    HSR_HIP_CHECK(hipMemsetAsync(in.dL_dconic, 0, sizeof(float) * 4 * P, stream));
    HSR_HIP_CHECK(hipMemsetAsync(in.dL_dopacity, 0, sizeof(float) * P, stream));
    HSR_HIP_CHECK(hipMemsetAsync(in.dL_dcolor, 0, sizeof(float) * 3 * P, stream));
    HSR_HIP_CHECK(hipMemsetAsync(in.dL_ddepth, 0, sizeof(float) * P, stream));
    if (K > 0) HSR_HIP_CHECK(hipMemsetAsync(in.dL_dsemantics, 0, sizeof(float) * (size_t)K * P, stream));
    // SH coefficients above the active degree (and those of culled Gaussians) receive no gradient
    if (!in.colors_precomp && in.shs && in.dL_dsh && in.M > 0)
        HSR_HIP_CHECK(hipMemsetAsync(in.dL_dsh, 0, sizeof(float) * 3 * (size_t)in.M * P, stream));
    return HSR_OK;
}

int backward_impl(const BwdIn& in, hipStream_t stream)
{
    int rc;
    const int P = in.P, W = in.W, H = in.H;
    if (P < 0 || W <= 0 || H <= 0 || in.R < 0) {
        hsr_set_error("invalid sizes P=%d W=%d H=%d R=%d", P, W, H, in.R);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return HSR_OK;
    if (!in.geom || !in.img || (in.R > 0 && !in.binning)) {
        hsr_set_error("state buffers from the forward call are required");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    // geometry-only call: no gradient wanted for colours, opacities and semantics (all three NULL) — a tracking iteration;
    // needs precomputed colours (with SH colours the view direction carries dL_dcolor into dL_dmean3D) and a scratch buffer
    const bool geo_request = !in.dL_dcolor && !in.dL_dopacity && !in.dL_dsemantics && in.colors_precomp != nullptr;
    if (!in.dL_dpix || !in.dL_dpix_depth || !in.dL_dpix_median || !in.dL_dpix_opacity ||
        (in.semantic && in.K > 0 && !geo_request && (!in.dL_dpix_sem || !in.dL_dsemantics))) {
        hsr_set_error("an upstream-gradient pointer is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    // dL_dconic, dL_ddepth (intermediates the reference keeps to itself, rasterize_points.cu:380-383) and dL_dcov3D may be
    // NULL = not wanted, as long as a scratch buffer carries the accumulation (checked below for the legacy mode)
    if (!in.dL_dmean2D || !in.dL_dmean3D || (!geo_request && (!in.dL_dopacity || !in.dL_dcolor))) {
        hsr_set_error("a gradient output pointer is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (in.scales && !in.rotations) {   // dL_dscale / dL_drot may be NULL: not wanted
        hsr_set_error("scales given without rotations");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!in.background) {
        hsr_set_error("background is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    GeomState g;
    ImgState im;
    BinState b;
    hsr_carve_geom(const_cast<char*>(in.geom), P, &g);
    hsr_carve_img(const_cast<char*>(in.img), W, H, &im);
    hsr_carve_bin(const_cast<char*>(in.binning), in.R, &b);
    const int* radii = in.radii ? in.radii : g.radii;
    const int K = in.semantic ? in.K : 0;

    // Accumulation modes (hsr_backward_scratch_bytes() sizes the scratch for the one in force):
    //   packed (default with scratch): fp32 atomics into ONE 64-byte-aligned row per Gaussian, unpacked by
    //       the per-Gaussian kernel — half the atomic requests of the reference's six separate arrays;
    //   rows (HSR_BWD_IMPL=rows, K <= 27): per-instance rows, no global atomics (experimental);
    //   legacy (no scratch): atomics straight into the six output arrays.
#ifdef HSR_ABLATE
    const bool want_rows = rows_mode_requested();
    const bool use_rows = want_rows && in.scratch && hsr_rows_supported(K) && in.R > 0 &&
                          in.scratch_bytes >= hsr_backward_scratch_bytes(P, K, in.R);
#else
    const bool use_rows = false;
#endif
    int glayout = 0;   // set below once the accumulation mode is known
    int gstride = hsr_grow_stride(K);
    const bool use_packed = !use_rows && backward_mode() != 2 && in.scratch &&
                            in.scratch_bytes >= (size_t)P * gstride * sizeof(float) + 256;
    const bool geo = geo_request && use_packed && (size_t)P * 16 < ((size_t)1 << 30);
    if (geo_request && !geo) {
        hsr_set_error("dL_dcolor / dL_dopacity / dL_dsemantics may only all be NULL (geometry-only gradients) in the packed accumulation mode");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!use_rows && !use_packed && (!in.dL_dconic || !in.dL_ddepth)) {
        hsr_set_error("dL_dconic and dL_ddepth may only be NULL when a scratch buffer carries the accumulation (packed / rows mode)");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    // opt-in exact semantic -> alpha term: extra passes over the packed rows (a geometry-only caller that passes no dL_dpix_sem has no
    // semantic loss: nothing to add)
    const bool sem_alpha = semantic_alpha_mode() == 1 && K > 0 && in.dL_dpix_sem != nullptr;
    if (sem_alpha) {
        if (!use_packed || (size_t)P * (size_t)(geo ? 16 : hsr_grow_stride(K)) >= ((size_t)1 << 30)) {
            hsr_set_error("the exact semantic -> alpha mode needs the packed accumulation mode (a scratch buffer of hsr_backward_scratch_bytes) and P * row stride < 2^30");
            return HSR_ERR_INVALID_ARGUMENT;
        }
        if (!in.semantics) {
            hsr_set_error("the exact semantic -> alpha mode reads semantics_precomp, which is NULL");
            return HSR_ERR_INVALID_ARGUMENT;
        }
    }
    float* grow = nullptr;
    int rows_kc = 0;
    float* rows = nullptr;
    uint32_t* inv = nullptr;
#ifdef HSR_ABLATE
    if (use_rows) {
        char* sp = in.scratch;
        take(sp, rows, (size_t)in.R * (size_t)hsr_rows_row_floats(K));
        take(sp, inv, (size_t)in.R);
        // SH coefficients above the active degree (and those of culled Gaussians) receive no gradient
        if (!in.colors_precomp && in.shs && in.dL_dsh && in.M > 0)
            HSR_HIP_CHECK(hipMemsetAsync(in.dL_dsh, 0, sizeof(float) * 3 * (size_t)in.M * (size_t)P, stream));
    } else
#endif
    if (use_packed) {
        if (geo) gstride = 16;   // one 64-byte line per Gaussian: columns 0..6
        else {
            glayout = hsr_backward_row_layout(K, true, P);   // compact rows where the tile kernel that will run writes them and they save a line
            gstride = hsr_grow_stride_l(glayout, K);
        }
        char* sp = in.scratch;
        take(sp, grow, (size_t)P * gstride);
        StageTimer tm(HSR_STAGE_BWD_ZERO, stream);
        // only the rows of visible Gaussians (radii > 0): nothing else is added into or read back (hsr_backward_pre.hip)
        hsr_launch_zero_visible_rows(P, radii, grow, gstride, stream);
        if (!in.colors_precomp && in.shs && in.dL_dsh && in.M > 0)
            HSR_HIP_CHECK(hipMemsetAsync(in.dL_dsh, 0, sizeof(float) * 3 * (size_t)in.M * (size_t)P, stream));
    } else {
        if ((rc = zero_accumulators(in, K, stream)) != HSR_OK) return rc;
    }

    if (in.R > 0) {
        RenderBwdArgs ra;
        ra.W = W; ra.H = H; ra.K = K; ra.semantic = in.semantic; ra.P = P;
        {
            static const int dbg = hsr_ablate_env("HSR_DEBUG_FLAGS") ? atoi(hsr_ablate_env("HSR_DEBUG_FLAGS")) : 0;
            ra.debug_flags = dbg;   // 0 in the product build
        }
        ra.bg = in.background; ra.ranges = im.ranges; ra.point_list = b.vals; ra.masks = b.vals_unsorted; ra.means2D = g.means2D;
        ra.conic_opacity = g.conic_opacity; ra.depths = g.depths; ra.colors = in.colors_precomp ? in.colors_precomp : g.rgb;
        ra.rec = g.rec;
        ra.final_T = im.final_T; ra.n_contrib = im.n_contrib; ra.median_pos = im.median_pos;
        ra.dL_dpix = in.dL_dpix; ra.dL_dpix_sem = in.dL_dpix_sem; ra.dL_dpix_depth = in.dL_dpix_depth;
        ra.dL_dpix_median = in.dL_dpix_median; ra.dL_dpix_opacity = in.dL_dpix_opacity;
        ra.dL_dmean2D = in.dL_dmean2D; ra.dL_dconic = in.dL_dconic; ra.dL_dopacity = in.dL_dopacity;
        ra.dL_dcolor = in.dL_dcolor; ra.dL_dsemantics = in.dL_dsemantics; ra.dL_ddepth = in.dL_ddepth;
        ra.rows = rows;
        ra.grow = grow;
        ra.grow_stride = gstride;
        ra.grow_layout = glayout;
        StageTimer tm(HSR_STAGE_BWD_RENDER, stream);
#ifdef HSR_ABLATE
        if (use_rows) {
            const int tiles_x = (W + HSR_TILE_X - 1) / HSR_TILE_X, tiles_y = (H + HSR_TILE_Y - 1) / HSR_TILE_Y;
            hsr_launch_inverse_map(in.R, tiles_x, tiles_y, b.keys, b.vals, g.means2D, radii, g.point_offsets, inv, stream);
            rows_kc = hsr_launch_render_backward_rows(ra, stream);
        } else
#endif
        if (geo) {
            static const char* e_impl = getenv("HSR_BWD_IMPL");
            static const bool old_sub = e_impl && !strcmp(e_impl, "sub");   // round 3's butterfly kernel (A/B timing, parity-tested)
            if (old_sub) hsr_launch_render_backward_geo(ra, stream);
            else hsr_launch_render_backward_qgeo(ra, stream);
        } else {
            hsr_launch_render_backward(ra, stream);
        }
        if (sem_alpha) {
            ra.semantics = in.semantics;
            hsr_launch_render_backward_qsema(ra, stream);
        }
    }
    HSR_LAUNCH_CHECK(in.debug, stream);

    PreBwdArgs pb;
    pb.P = P; pb.D = in.D; pb.M = in.M; pb.means3D = in.means3D; pb.radii = radii;
    pb.shs = in.colors_precomp ? nullptr : in.shs; pb.clamped = g.clamped; pb.scales = in.scales; pb.rotations = in.rotations;
    pb.scale_modifier = in.scale_modifier; pb.cov3Ds = in.cov3D_precomp ? in.cov3D_precomp : g.cov3D;
    pb.viewmatrix = in.viewmatrix; pb.projmatrix = in.projmatrix;
    pb.focal_y = H / (2.0f * in.tan_fovy); pb.focal_x = W / (2.0f * in.tan_fovx);
    pb.tan_fovx = in.tan_fovx; pb.tan_fovy = in.tan_fovy; pb.campos = in.campos;
    pb.dL_dmean2D = in.dL_dmean2D; pb.dL_dconic = in.dL_dconic; pb.dL_dmean3D = in.dL_dmean3D; pb.dL_dcolor = in.dL_dcolor;
    pb.dL_ddepth = in.dL_ddepth; pb.dL_dcov3D = in.dL_dcov3D; pb.dL_dsh = in.dL_dsh; pb.dL_dscale = in.dL_dscale;
    pb.dL_drot = in.dL_drot;
    pb.rows_kc = rows_kc; pb.K = K; pb.rows = rows; pb.inv = inv; pb.point_offsets = g.point_offsets;
    pb.out_mean2D = in.dL_dmean2D; pb.out_conic = in.dL_dconic; pb.out_opacity = in.dL_dopacity; pb.out_color = in.dL_dcolor;
    pb.out_semantics = in.dL_dsemantics; pb.out_depth = in.dL_ddepth;
    pb.grow = grow; pb.grow_stride = gstride; pb.grow_layout = glayout; pb.geo = geo ? 1 : 0;
    if (pb.shs && (!in.dL_dsh || !in.campos)) {
        hsr_set_error("shs given without dL_dsh / campos");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    {
        StageTimer tm(HSR_STAGE_BWD_PREPROCESS, stream);
        hsr_launch_preprocess_backward(pb, stream);
    }
    HSR_LAUNCH_CHECK(in.debug, stream);
    return HSR_OK;
}

}  // namespace

void hsr_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

size_t hsr_carve_geom(char* base, int P, GeomState* out)
{
    char* p = base;
    GeomState g;
    const size_t Pn = (size_t)(P > 0 ? P : 1);
    take(p, g.depths, Pn);
    take(p, g.means2D, Pn);
    take(p, g.conic_opacity, Pn);
    take(p, g.cov3D, Pn * 6);
    take(p, g.rgb, Pn * 3);
    take(p, g.clamped, Pn * 3);
    take(p, g.tiles_touched, Pn);
    take(p, g.point_offsets, Pn);
    take(p, g.radii, Pn);
    take(p, g.block_sums, (Pn + 255) / 256 + 1);
    take(p, g.counters, 8);
    take(p, g.rec, Pn * 4);
    if (out) *out = g;
    return (size_t)(p - base);
}
size_t hsr_carve_img(char* base, int W, int H, ImgState* out)
{
    char* p = base;
    ImgState s;
    const size_t N = (size_t)W * H;
    const size_t T = (size_t)((W + HSR_TILE_X - 1) / HSR_TILE_X) * ((H + HSR_TILE_Y - 1) / HSR_TILE_Y);
    take(p, s.ranges, T);
    take(p, s.final_T, N);
    take(p, s.n_contrib, N);
    take(p, s.median_pos, N);
    if (out) *out = s;
    return (size_t)(p - base);
}
size_t hsr_carve_bin(char* base, int R, BinState* out)
{
    // ONE layout rule for the binning buffer: hsr_bin_resolve (hsr_common.h), which the speculative kernels also evaluate on the
    // device from num_rendered.  It aligns each array from the real address, so the size query runs it from a 256-aligned origin.
    BinState b;
    char* origin = base ? base : reinterpret_cast<char*>(uintptr_t(256));
    const BinDevRef ref{origin, nullptr, ~size_t(0) >> 1};
    hsr_bin_resolve(ref, (uint32_t)(R > 0 ? R : 0), &b);
    const uint32_t Rn = (uint32_t)(R > 0 ? R : 0);
    const size_t bytes = (size_t)(reinterpret_cast<char*>(b.hist + hsr_sort_hist_entries_inline(Rn)) - origin);
    if (!base) {   // layout query: offsets from a NULL base, as hsr_get_state_layout reports them
        auto rebase = [&](auto*& q) { q = reinterpret_cast<std::remove_reference_t<decltype(q)>>(reinterpret_cast<char*>(q) - 256); };
        rebase(b.keys_unsorted); rebase(b.keys); rebase(b.vals_unsorted); rebase(b.vals); rebase(b.hist);
    }
    if (out) *out = b;
    return bytes;
}

extern "C" {

size_t hsr_required_geometry_bytes(int P) { return hsr_carve_geom(nullptr, P, nullptr) + 256; }
size_t hsr_required_image_bytes(int width, int height) { return hsr_carve_img(nullptr, width, height, nullptr) + 256; }
size_t hsr_required_binning_bytes(int num_rendered) { return hsr_carve_bin(nullptr, num_rendered, nullptr) + 256; }
size_t hsr_backward_scratch_bytes(int P, int K, int num_rendered)
{
    if (P <= 0 || K < 0 || backward_mode() == 2) return 0;
#ifdef HSR_ABLATE
    if (rows_mode_requested() && hsr_rows_supported(K) && num_rendered > 0)
        return (size_t)num_rendered * ((size_t)hsr_rows_row_floats(K) * 4 + 4) + 1024;
#endif
    return (size_t)P * hsr_grow_stride(K) * sizeof(float) + 512;  // packed per-Gaussian rows
}

const char* hsr_last_error(void) { return g_err; }

int hsr_get_backward_mode(void) { return backward_mode(); }

int hsr_set_backward_mode(int mode)
{
    if (mode < 0 || mode > 2) {
        hsr_set_error("backward mode must be 0 (packed), 1 (rows) or 2 (legacy)");
        return HSR_ERR_INVALID_ARGUMENT;
    }
#ifndef HSR_ABLATE
    if (mode == 1) {
        hsr_set_error("the per-instance rows mode is an experiment (measured slower): it exists in the ablate build only");
        return HSR_ERR_INVALID_ARGUMENT;
    }
#endif
    g_bwd_mode = mode;
    return HSR_OK;
}

int hsr_get_semantic_alpha_mode(void) { return semantic_alpha_mode(); }

int hsr_set_semantic_alpha_mode(int mode)
{
    if (mode != 0 && mode != 1) {
        hsr_set_error("semantic alpha mode must be 0 (as the reference: no semantic -> alpha gradient) or 1 (exact)");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    g_sem_alpha_mode = mode;
    return HSR_OK;
}

double hsr_profile_host_wait_ms(int reset)
{
    const double v = g_host_wait_ms;
    if (reset) g_host_wait_ms = 0.0;
    return v;
}

int hsr_profile_select(unsigned stage_mask)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_mask = stage_mask;
    return HSR_OK;
}

int hsr_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return HSR_OK;
}

int hsr_profile_read(hsr_profile* out, int reset)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& pe : g_prof_pending) {
        HSR_HIP_CHECK(hipEventSynchronize(pe.stop));
        float ms = 0.f;
        HSR_HIP_CHECK(hipEventElapsedTime(&ms, pe.start, pe.stop));
        g_prof_acc.ms[pe.stage] += ms;
        g_prof_acc.calls[pe.stage] += 1;
        g_prof_free.push_back(pe.start);
        g_prof_free.push_back(pe.stop);
    }
    g_prof_pending.clear();
    if (out) *out = g_prof_acc;
    if (reset) memset(&g_prof_acc, 0, sizeof(g_prof_acc));
    return HSR_OK;
}

const char* hsr_stage_name(int stage)
{
    static const char* names[HSR_STAGE_COUNT] = {"fwd_preprocess", "fwd_scan", "fwd_duplicate", "fwd_sort", "fwd_ranges",
                                                 "fwd_render", "bwd_zero", "bwd_render", "bwd_preprocess"};
    return (stage >= 0 && stage < HSR_STAGE_COUNT) ? names[stage] : "?";
}
const char* hsr_version(void) { return "hsr_rast 0.1 gfx950"; }

int hsr_get_state_layout(int P, int width, int height, int num_rendered, hsr_state_layout* out)
{
    if (!out) return HSR_ERR_INVALID_ARGUMENT;
    GeomState g;
    ImgState im;
    BinState b;
    hsr_carve_geom(nullptr, P, &g);
    hsr_carve_img(nullptr, width, height, &im);
    hsr_carve_bin(nullptr, num_rendered, &b);
#define OFF(p) ((size_t) reinterpret_cast<uintptr_t>(p))
    out->geom_depths = OFF(g.depths); out->geom_means2D = OFF(g.means2D); out->geom_conic_opacity = OFF(g.conic_opacity);
    out->geom_cov3D = OFF(g.cov3D); out->geom_rgb = OFF(g.rgb); out->geom_clamped = OFF(g.clamped);
    out->geom_tiles_touched = OFF(g.tiles_touched); out->geom_point_offsets = OFF(g.point_offsets); out->geom_radii = OFF(g.radii);
    out->bin_keys_unsorted = OFF(b.keys_unsorted); out->bin_keys = OFF(b.keys); out->bin_vals_unsorted = OFF(b.vals_unsorted);
    out->bin_vals = OFF(b.vals);
    out->img_ranges = OFF(im.ranges); out->img_final_T = OFF(im.final_T); out->img_n_contrib = OFF(im.n_contrib);
    out->img_median_pos = OFF(im.median_pos);
#undef OFF
    return HSR_OK;
}

int hsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
    if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) {
        hsr_set_error("hsr_mark_visible: invalid arguments");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    hsr_launch_mark_visible(P, means3D, viewmatrix, projmatrix, present, static_cast<hipStream_t>(stream));
    HSR_HIP_CHECK(hipGetLastError());
    return HSR_OK;
}

int hsr_forward_arm_async(hsr_ticket* ticket)
{
    if (!ticket) {
        hsr_set_error("hsr_forward_arm_async: ticket is NULL");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    memset(ticket, 0, sizeof(*ticket));
    g_armed_ticket = ticket;
    return HSR_OK;
}

int hsr_forward_end(hsr_ticket* ticket, int block, void* stream)
{
    if (!ticket || ticket->seq == 0 || !ticket->slot) {
        hsr_set_error("hsr_forward_end: the ticket does not belong to a forward call that ran ahead");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    if (!block) {
        const uint32_t cur = __atomic_load_n(&ticket->slot[1], __ATOMIC_ACQUIRE);
        if (cur != ticket->seq && !(cur != 0 && (int32_t)(cur - ticket->seq) > 0)) return HSR_PENDING;   // (overwritten: reported below)
    }
    uint32_t R32 = 0;
    int rc;
    if ((rc = poll_counter(ticket->seq, static_cast<hipStream_t>(stream), &R32, ticket->slot)) != HSR_OK) return rc;
    if (R32 > 0x7fffffffu) {
        hsr_set_error("num_rendered %u overflows int", R32);
        return HSR_ERR_INVALID_ARGUMENT;
    }
    ticket->rendered = (int32_t)R32;
    if (ticket->prefiltered && __atomic_load_n(&ticket->slot[2], __ATOMIC_RELAXED) != 0) {
        hsr_set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
        return HSR_ERR_INVALID_ARGUMENT;
    }
    BinState chk;
    const BinDevRef ref{ticket->binning_base, nullptr, ticket->binning_capacity};
    if (!hsr_bin_resolve(ref, R32, &chk)) {
        hsr_set_error("non-blocking forward: num_rendered = %u does not fit the binning buffer (%zu bytes, %zu needed): the output images of "
                      "that call were filled with NaN; run the forward again with a larger buffer", R32, ticket->binning_capacity,
                      hsr_required_binning_bytes((int)R32));
        return HSR_ERR_BUFFER_TOO_SMALL;
    }
    return (int)R32;
}

int hsr_forward(hsr_buffer* geometry, hsr_buffer* binning, hsr_buffer* image, int P, int D, int M, const float* background,
                int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_median_depth,
                float* out_opacity, float* out_mask, int* radii, int debug, void* stream)
{
    FwdIn in;
    in.P = P; in.D = D; in.M = M; in.K = 0; in.semantic = 0; in.W = width; in.H = height; in.prefiltered = prefiltered; in.debug = debug;
    in.background = background; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.semantics = nullptr;
    in.opacities = opacities; in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp;
    in.viewmatrix = viewmatrix; in.projmatrix = projmatrix; in.cam_pos = cam_pos;
    in.scale_modifier = scale_modifier; in.tan_fovx = tan_fovx; in.tan_fovy = tan_fovy;
    in.out_color = out_color; in.out_semantic = nullptr; in.out_depth = out_depth; in.out_median = out_median_depth;
    in.out_opacity = out_opacity; in.out_mask = out_mask; in.radii = radii;
    return forward_impl(geometry, binning, image, in, static_cast<hipStream_t>(stream));
}

int hsr_forward_semantic(hsr_buffer* geometry, hsr_buffer* binning, hsr_buffer* image, int P, int D, int M, int K,
                         const float* background, int width, int height, const float* means3D, const float* shs,
                         const float* colors_precomp, const float* semantics_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                         float tan_fovy, int prefiltered, float* out_color, float* out_semantic, float* out_depth,
                         float* out_median_depth, float* out_opacity, int* radii, int debug, void* stream)
{
    FwdIn in;
    in.P = P; in.D = D; in.M = M; in.K = K; in.semantic = 1; in.W = width; in.H = height; in.prefiltered = prefiltered; in.debug = debug;
    in.background = background; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.semantics = semantics_precomp;
    in.opacities = opacities; in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp;
    in.viewmatrix = viewmatrix; in.projmatrix = projmatrix; in.cam_pos = cam_pos;
    in.scale_modifier = scale_modifier; in.tan_fovx = tan_fovx; in.tan_fovy = tan_fovy;
    in.out_color = out_color; in.out_semantic = out_semantic; in.out_depth = out_depth; in.out_median = out_median_depth;
    in.out_opacity = out_opacity; in.out_mask = nullptr; in.radii = radii;
    return forward_impl(geometry, binning, image, in, static_cast<hipStream_t>(stream));
}

int hsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* campos, float tan_fovx, float tan_fovy, const int* radii, const char* geom_buffer,
                 const char* binning_buffer, const char* img_buffer, const float* dL_dpix, const float* dL_dpix_depth,
                 const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity, float* dL_dmean2D, float* dL_dconic,
                 float* dL_dopacity, float* dL_dcolor, float* dL_ddepth, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot, char* scratch, size_t scratch_bytes, int debug, void* stream)
{
    BwdIn in;
    in.scratch = scratch; in.scratch_bytes = scratch_bytes;
    in.P = P; in.D = D; in.M = M; in.K = 0; in.semantic = 0; in.R = R; in.W = width; in.H = height; in.debug = debug;
    in.background = background; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.semantics = nullptr;
    in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp; in.viewmatrix = viewmatrix;
    in.projmatrix = projmatrix; in.campos = campos; in.scale_modifier = scale_modifier; in.tan_fovx = tan_fovx; in.tan_fovy = tan_fovy;
    in.radii = radii; in.geom = geom_buffer; in.binning = binning_buffer; in.img = img_buffer;
    in.dL_dpix = dL_dpix; in.dL_dpix_sem = nullptr; in.dL_dpix_depth = dL_dpix_depth; in.dL_dpix_median = dL_dpix_median_depth;
    in.dL_dpix_opacity = dL_dpix_final_opacity;
    in.dL_dmean2D = dL_dmean2D; in.dL_dconic = dL_dconic; in.dL_dopacity = dL_dopacity; in.dL_dcolor = dL_dcolor;
    in.dL_dsemantics = nullptr; in.dL_ddepth = dL_ddepth; in.dL_dmean3D = dL_dmean3D; in.dL_dcov3D = dL_dcov3D; in.dL_dsh = dL_dsh;
    in.dL_dscale = dL_dscale; in.dL_drot = dL_drot;
    return backward_impl(in, static_cast<hipStream_t>(stream));
}

int hsr_backward_semantic(int P, int D, int M, int K, int R, const float* background, int width, int height,
                          const float* means3D, const float* shs, const float* colors_precomp, const float* semantics_precomp,
                          const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                          const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                          const int* radii, const char* geom_buffer, const char* binning_buffer, const char* img_buffer,
                          const float* dL_dpix, const float* dL_dpix_semantic, const float* dL_dpix_depth,
                          const float* dL_dpix_median_depth, const float* dL_dpix_final_opacity, float* dL_dmean2D,
                          float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dsemantics, float* dL_ddepth,
                          float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                          char* scratch, size_t scratch_bytes, int debug, void* stream)
{
    BwdIn in;
    in.scratch = scratch; in.scratch_bytes = scratch_bytes;
    in.P = P; in.D = D; in.M = M; in.K = K; in.semantic = 1; in.R = R; in.W = width; in.H = height; in.debug = debug;
    in.background = background; in.means3D = means3D; in.shs = shs; in.colors_precomp = colors_precomp; in.semantics = semantics_precomp;
    in.scales = scales; in.rotations = rotations; in.cov3D_precomp = cov3D_precomp; in.viewmatrix = viewmatrix;
    in.projmatrix = projmatrix; in.campos = campos; in.scale_modifier = scale_modifier; in.tan_fovx = tan_fovx; in.tan_fovy = tan_fovy;
    in.radii = radii; in.geom = geom_buffer; in.binning = binning_buffer; in.img = img_buffer;
    in.dL_dpix = dL_dpix; in.dL_dpix_sem = dL_dpix_semantic; in.dL_dpix_depth = dL_dpix_depth; in.dL_dpix_median = dL_dpix_median_depth;
    in.dL_dpix_opacity = dL_dpix_final_opacity;
    in.dL_dmean2D = dL_dmean2D; in.dL_dconic = dL_dconic; in.dL_dopacity = dL_dopacity; in.dL_dcolor = dL_dcolor;
    in.dL_dsemantics = dL_dsemantics; in.dL_ddepth = dL_ddepth; in.dL_dmean3D = dL_dmean3D; in.dL_dcov3D = dL_dcov3D; in.dL_dsh = dL_dsh;
    in.dL_dscale = dL_dscale; in.dL_drot = dL_drot;
    return backward_impl(in, static_cast<hipStream_t>(stream));
}

}  // extern "C"
