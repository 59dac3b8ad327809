// Host runtime of the render core: context, device buffers, stream-ordered kernel sequencing and the C ABI.
//
// A context plays the part of the reference's PathTracer + VulkanRenderer pair
// (src/vulkan/path_tracer/mod.rs:56-344, src/vulkan/renderer.rs:62-507) without a window: it owns the scene in
// device layout, the path-state and queue buffers, the fp32 accumulator and the HIP streams on which the
// kernels of a render call are queued back to back (no host synchronisation inside rt_render).
//
// Stream plan: a wavefront pass runs on a `lane` (a stream, a side stream for its shadow rays, and a private set of
// path-state buffers; as side stream it borrows the next lane's stream). By default there are three lanes, long passes use the
// first (or the first two, where paths are cheap) and short passes go round all of them; with RT_HIP_LANES=n consecutive passes always alternate between n lanes and overlap. Accumulation of every pass is
// queued on the context's main stream in pass order (it waits on the pass's lane through an event, and the lane's next
// pass waits until its radiance buffer has been consumed), so the floating-point summation order does not depend on how
// the lanes interleave.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_api.h"
#include "rt_internal.h"
#include "rt_kernels.h"

using namespace rtdev;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(bytes, o.bytes); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

#ifndef RT_DYNAMIC
#define RT_DYNAMIC 1
#endif
// Work distribution of the traversal kernels: block-strided chunks of neighbouring items, with (1) or without (0) a
// device-wide cursor that hands out the tail of each list dynamically.
#define RT_CURSOR(p) (RT_DYNAMIC ? (p) : nullptr)
// Grid sizes in 256-thread units per CU. Traversal runs RT_TRACE_WAVES_PER_EU waves per SIMD (the LDS each workgroup
// needs for stack heads and resident node records sets how many fit). Shading: three workgroups per CU are resident at the
// kernel's 134 VGPRs, but its grid is larger, eight per CU: vertices differ a lot in cost (a miss, a textured hit, a light
// sample) and a grid-stride loop over exactly the resident workgroups ends when the unluckiest one does, while a grid with
// workgroups to spare is rebalanced by the dispatcher (per 1080p sample of the atrium: 3/CU 0.366 ms, 4/CU 0.310, 8/CU 0.319,
// 16/CU 0.306, but 5 or 6/CU 0.37: grids that are a power of two per CU only; four resident waves per SIMD at 127 VGPRs change
// nothing). Over all configurations 8/CU is 2-8 % of the frame; one sample per call does not care.
// RT_HIP_TRACE_BLOCKS / RT_HIP_SHADE_BLOCKS override them for tuning runs (read once per context, at creation).
constexpr uint32_t kDefaultTraceBlocksPerCu = RT_TRACE_WAVES_PER_EU;
constexpr uint32_t kDefaultTraceBlocksShort = 4;
constexpr uint32_t kDefaultShadeBlocksPerCu = 8;
constexpr float kCheapSlotNs = 0.7f;   // long passes cheaper than this per path (open scenes, tiny scenes) alternate between lanes too:
                                       // measured 0.38-0.50 ns per path on the car, glass and Cornell scenes, 0.92-1.4 on the atrium and the interior
constexpr uint64_t kOverlapConnectSlots = 12ull << 20;   // passes of up to this many paths (5 frames of 1080p) trace their shadow rays on the side stream
constexpr uint32_t kOverlapFromBounce = 4;               // ... and in longer passes from this bounce on, where few paths are left (on half-size grids)
constexpr uint64_t kGraphPassSlots = kOverlapConnectSlots;   // largest pass that can be replayed from a captured HIP graph (RT_HIP_GRAPH=1)
// ... and the largest that is by default: replay halves the host's time to queue a pass (0.10 -> 0.05 ms), which a host
// that waits for every sample sees on small images (one sample per call: 256x256 Cornell box 0.363 -> 0.32 ms, 320x180
// atrium 1.33 -> 1.20, 640x360 1.83 -> 1.70, 1280x720 2.98 -> 2.87, 1920x1080 4.19 -> 4.16) and nobody else
constexpr uint64_t kGraphDefaultSlots = 1ull << 20;
constexpr uint64_t kSmallPassSlots = 24ull << 20;   // passes of up to this many paths (12 frames of 1080p) alternate between lanes

// The per-path kernel (k_path_tail) is offered the queue of every bounce from kDefaultTailFrom on, in passes of at most
// kDefaultTailPassSlots paths, and takes a queue over once it holds at most kDefaultTailMaxPaths paths. Measured, one sample
// per call queued back to back (tools/one_sample.py): a small pass cannot fill the machine in any bounce, every wavefront
// launch of it costs the 0.1-0.2 ms its longest ray takes, and the kernel replaces all that are left with one launch
// (atrium 320x180: 1.08 -> 0.80 ms per sample, car 640x360: 1.06 -> 0.91, interior 480x270: 1.98 -> 1.63, glass blob
// 640x360: 2.01 -> 1.60, Cornell box 256x256 and 512x512: within 1 %). In a 1080p sample the queues that short come late
// (118 thousand paths at bounce 5 of the atrium) after bounces that did fill the machine, the kernel is slower than the three
// bounces of wavefront launches it replaces (3.72 -> 3.83 ms), and merely offering it costs 4 % (the previous bounce's shadow
// rays have to finish first): hence the limit on the pass size. RT_HIP_TAIL_FROM / RT_HIP_TAIL_MAX / RT_HIP_TAIL_SLOTS
// override the three numbers (read once per context, at creation; RT_HIP_TAIL_MAX=0 switches the kernel off).
constexpr uint32_t kDefaultTailFrom = 2;
constexpr uint32_t kDefaultTailMaxPaths = 128u << 10;
constexpr uint32_t kDefaultTailPassSlots = 512u << 10;

constexpr uint32_t kMaxLanes = 8;
constexpr uint32_t kDefaultLanes = 3;   // concurrent passes of a context (RT_HIP_LANES overrides): with the main stream, four streams
constexpr uint32_t kFullLanes = 2;      // lanes sized for passes of any length; the others hold kSmallPassSlots paths

enum Stage { ST_RAYGEN = 0, ST_EXTEND, ST_SHADE, ST_CONNECT, ST_ACCUM, ST_COUNT };

struct Span { int stage; hipEvent_t a, b; };

// One in-flight wavefront pass: its stream, path state, queues and counters.
struct Lane {
    hipStream_t stream = nullptr;
    // path state of the current and the next bounce (ping-pong, queue order), closest hits (queue order), light samples
    // (their own queue order), per-slot radiance, counters
    DevBuf st_o[2], st_d[2], st_t[2], hit, rad, sq_o, sq_d, sq_c, counts;
    hipEvent_t pass_done = nullptr, consumed = nullptr;
    // connect(b) and extend(b+1) both need only shade(b): the shadow rays of a bounce run on `side` beside the next
    // bounce's closest-hit search, each persistent grid filling the compute units the other leaves as it runs down
    hipStream_t side = nullptr;   // the next lane's stream (or, with a single lane, a stream of its own: side_owned)
    bool side_owned = false;
    hipEvent_t shaded = nullptr, connected = nullptr;
    // device time of the last pass this lane ran with the machine to itself (harvested by the next rt_render call)
    hipEvent_t t_begin = nullptr, t_end = nullptr;
    bool timed = false;
    uint64_t timed_slots = 0;
    bool busy = false;   // a pass was queued whose radiance the main stream has (or will have) consumed at `consumed`
    // Launch sequences of short passes on this lane, captured once and replayed (see queue_pass)
    struct PassGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipGraphNode_t raygen = nullptr;
                       uint64_t epoch = 0; uint32_t frames = 0; bool side_ok = false; };
    PassGraph graphs[4];
    uint32_t next_graph = 0;
    uint64_t seen_epoch = 0; uint32_t seen_frames = 0; bool seen_side_ok = false;   // the last shape that was queued directly
    void drop_graphs() {
        for (PassGraph& g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
            g = PassGraph{};
        }
    }
    uint32_t shard_cap = 0;   // records per region of this lane's queues
    uint64_t slot_cap = 0;    // paths a pass on this lane may hold (lanes beyond the second serve short passes only and are smaller)

    PathDev state(int k) const { return PathDev{st_o[k].as<float4>(), st_d[k].as<float4>(), st_t[k].as<float4>()}; }
    ShadowDev sq() const { return ShadowDev{sq_o.as<float4>(), sq_d.as<float4>(), sq_c.as<float4>()}; }
    std::vector<DevBuf*> path_buffers() {
        return {&st_o[0], &st_o[1], &st_d[0], &st_d[1], &st_t[0], &st_t[1], &hit, &rad, &sq_o, &sq_d, &sq_c};
    }
    // work cursors: a set of kTailCursors counters (kCursorSetWords words) per traversal launch of a pass, and one set for
    // the stage-level entry points
    uint32_t* aux_cursor() const { return counts.as<uint32_t>(); }
    uint32_t* ext_cursor(uint32_t bounce) const { return counts.as<uint32_t>() + size_t(1 + bounce) * kCursorSetWords; }
    uint32_t* con_cursor(uint32_t bounce) const { return counts.as<uint32_t>() + size_t(1 + kMaxBounces + 1 + bounce) * kCursorSetWords; }
    // the per-path kernel's "workgroups finished" counter of a bounce (the cursor set of the bounce that never exists)
    uint32_t* tail_done(uint32_t bounce) const { static_assert(kCursorSetWords >= kMaxBounces, "one word per bounce"); return ext_cursor(kMaxBounces) + bounce; }
    // per bounce: kQueueShards append counters for the next bounce's paths, then kQueueShards for the light samples
    uint32_t* shard_counts(uint32_t bounce) const { return counts.as<uint32_t>() + kCountersAt + size_t(bounce) * 2 * kQueueShards * kCounterStride; }
    uint32_t* shadow_counts(uint32_t bounce) const { return shard_counts(bounce) + size_t(kQueueShards) * kCounterStride; }
    static constexpr size_t kCountersAt = size_t(1 + 2 * (kMaxBounces + 1)) * kCursorSetWords;   // behind the cursor sets
    static size_t counts_words(uint32_t bounces = kMaxBounces) { return kCountersAt + size_t(bounces) * 2 * kQueueShards * kCounterStride; }
};

}  // namespace

// Fixed staging area of rt_trace_closest_static: [work cursors, 2 KiB][origins 3 x f32][directions 3 x f32][hits 16 B].
// Host code never uses the array's name as an address: each device holds its own copy, whose address
// hipGetSymbolAddress reports once per device (static_scratch_for) and which is kept in g_static_scratch.
constexpr size_t kStaticCursorBytes = size_t(kCursorSetWords) * 4;
constexpr size_t kStaticBytes = kStaticCursorBytes + size_t(RT_STATIC_MAX_RAYS) * (12 + 12 + 16);
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless the environment says otherwise, and
// streams that share a queue run one after the other whatever their events say. A context therefore runs four streams by
// default (three lanes and the main stream; a lane borrows its neighbour's stream as its side stream while the neighbour is
// idle). A host that sets GPU_MAX_HW_QUEUES=16 before the HIP runtime initialises (which happens before any code of this
// library runs) can afford more lanes (RT_HIP_LANES): 8 lanes and 32 queues bring one 1080p sample per call to 2.8 ms.

constexpr int kMaxDevices = 64;
static std::mutex g_static_mutex;
static unsigned char* g_static_scratch[kMaxDevices] = {};

// Device address of `scratch` on the current device `device` (resolved on first use, under a lock).
static hipError_t static_scratch_for(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_scratch[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_static_scratch[device] = static_cast<unsigned char*>(p);
    }
    *out = g_static_scratch[device];
    return hipSuccess;
}

struct rt_ctx {
    int device = 0;
    unsigned char* static_scratch = nullptr;  // this device's copy of `scratch`
    hipStream_t stream = nullptr;
    std::string err;
    uint32_t W = 0, H = 0, npix = 0;
    uint32_t max_bounces = 24, flags = 0, fpp_cfg = 0, fpp = 1;
    uint32_t local_frame = 0;                 // samples issued by this context since the last reset
    uint32_t slice_first = 0, slice_stride = 1;
    uint32_t tile_first = 0, tile_stride = 1; // rt_set_tile_slice: which 8x8 tiles of the image this context renders (1 = all)
    rt_camera cam{};
    bool have_scene = false, have_camera = false;
    bool fast_requested = true;               // the caller did not ask for RT_EXACT_ORDER
    bool exact_arith = false;                 // shade stage in the exact arithmetic mode (RT_EXACT_ARITH, implied by RT_EXACT_ORDER)
    bool fast = true;                         // walk the compressed 4-wide tree (false: RT_EXACT_ORDER, or a scene the compressed tree cannot hold)
    bool camera_within_limit = true;          // the camera sits where the compressed tree's boxes are conservative (SceneDev::origin_limit)
    bool lanes_must_wait = true;              // something other than rt_render has used the main stream since the last pass was queued
    float ns_per_slot = 0.0f;                 // device time per path of the last pass that ran alone; 0 = not measured yet
    float cheap_slot_ns = kCheapSlotNs;       // long passes cheaper than this per path are cut in two and share the machine
    bool second_lane_for_short_passes = false; // lanes beyond the first take passes of at most kSmallPassSlots paths only
    bool overlap_connect = true;              // RT_HIP_OVERLAP_CONNECT=0 keeps every launch of a pass on one stream
    uint64_t overlap_slots = kOverlapConnectSlots;   // RT_HIP_OVERLAP_SLOTS: largest pass whose shadow rays all go to the side stream
    bool paths_ok = false;                    // path buffers match W x H (false after a resize that ran out of memory)
    int n_cu = 256;
    uint32_t trace_blocks_per_cu = kDefaultTraceBlocksPerCu, shade_blocks_per_cu = kDefaultShadeBlocksPerCu;
    uint32_t trace_blocks_late = kDefaultTraceBlocksShort, overlap_from = kOverlapFromBounce;   // tuning aids: RT_HIP_TRACE_BLOCKS_LATE, RT_HIP_OVERLAP_FROM
    uint32_t trace_blocks_short = kDefaultTraceBlocksShort;   // ... of passes whose launches run beside other launches (run_pass)
    uint32_t tail_from = kDefaultTailFrom, tail_max_paths = kDefaultTailMaxPaths, tail_pass_slots = kDefaultTailPassSlots;
    bool use_graphs = true;                   // passes of up to graph_slots paths are replayed from captured HIP graphs
    uint64_t graph_slots = kGraphDefaultSlots; // RT_HIP_GRAPH=0: none, =1: every pass whose shadow rays may use the side stream
    uint64_t graph_epoch = 1;                 // bumped by every entry point that may change what a pass's kernels are given

    // scene
    DevBuf nodes, wide, cw, isect, shade, mats, lights, light_tris, cdf, texels, tex_table, sky, sky_f32, srgb;
    SceneDev sd{};
    // host copies of the three node arrays, kept for rt_tune_resident_set
    std::vector<rtint::NodePair> h_nodes;
    std::vector<rtint::NodeQuad> h_wide;
    std::vector<rtint::NodeCw> h_cw;
    // paths: lane 0 doubles as the working set of the stage-level entry points, which run on the main stream
    Lane lanes[kMaxLanes];
    uint32_t n_lanes = 1, next_lane = 0;
    hipEvent_t call_mark = nullptr;
    DevBuf stats, seeds, probe;
    uint32_t shard_cap = 0;
    uint32_t slot_cap = 0;
    // output
    DevBuf sum, accum8, fb, scratch;

    // timing
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<Span> spans;
    hipEvent_t call_begin = nullptr, call_end = nullptr;
    rt_timings last{};
    bool timings_dirty = false;

    uint32_t* aux_cursor() const { return lanes[0].aux_cursor(); }
    TraceStats* tstats() const { return stats.as<TraceStats>(); }
    // the tile slice as the kernels take it, the local pixels per frame it makes (64 per owned tile, overhanging ones
    // included), and how many of those are pixels of the image
    TileSlice tiles() const { return TileSlice{tile_first, tile_stride, (W + 7u) / 8u, W, H}; }
    uint32_t tiles_owned() const {
        const uint32_t n_tiles = ((W + 7u) / 8u) * ((H + 7u) / 8u);
        return tile_first < n_tiles ? (n_tiles - tile_first + tile_stride - 1u) / tile_stride : 0u;
    }
    uint32_t npix_pass() const { return tile_stride == 1u ? npix : tiles_owned() * 64u; }
    uint32_t pixels_owned() const {
        if (tile_stride == 1u) return npix;
        const uint32_t tx = (W + 7u) / 8u;
        uint64_t n = 0;
        for (uint32_t k = 0, t = tile_first; k < tiles_owned(); ++k, t += tile_stride) {
            const uint32_t x0 = (t % tx) * 8u, y0 = (t / tx) * 8u;
            n += uint64_t(std::min(8u, W - x0)) * std::min(8u, H - y0);
        }
        return uint32_t(n);
    }
};

namespace rtint {
void set_ctx_error(rt_ctx* ctx, const std::string& msg) { ctx->err = msg; }
}
using rtint::fail;

#define HIPCHK(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return fail(ctx, e__ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_DEVICE,                       \
                        std::string(#call) + ": " + hipGetErrorString(e__));                                \
    } while (0)

namespace {

uint32_t grid_for(const rt_ctx* c, uint64_t n, uint32_t blocks_per_cu) {
    uint64_t need = (n + kBlock - 1) / kBlock;
    uint64_t cap = uint64_t(c->n_cu) * blocks_per_cu;
    return uint32_t(std::max<uint64_t>(1, std::min(need, cap)));
}

// Persistent traversal grid: `blocks_per_cu` counts 256-thread units, so the waves per CU stay the same whatever kTraceBlock is.
uint32_t trace_grid_for(const rt_ctx* c, uint64_t n, uint32_t blocks_per_cu) {
    uint64_t need = (n + kTraceBlock - 1) / kTraceBlock;
    uint64_t cap = std::max<uint64_t>(1, uint64_t(c->n_cu) * blocks_per_cu * kBlock / kTraceBlock);
    return uint32_t(std::max<uint64_t>(1, std::min(need, cap)));
}

// Every entry point that touches the device comes through here; all of them except rt_render may queue work on the main
// stream that the next pass has to see (clears, uploads, stage-level calls on lane 0's buffers).
int bind_device(rt_ctx* c) { HIPCHK(c, hipSetDevice(c->device)); c->lanes_must_wait = true; ++c->graph_epoch; return RT_OK; }
// For entry points that leave scene, camera, sizes and slices alone (rendering, waiting, reading results): captured pass
// graphs stay valid.
int bind_device_keep_graphs(rt_ctx* c) {
    const uint64_t epoch = c->graph_epoch;
    const int rc = bind_device(c);
    c->graph_epoch = epoch;
    return rc;
}

hipEvent_t take_event(rt_ctx* c) {
    if (c->events_used == c->event_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->event_pool.push_back(e);
    }
    return c->event_pool[c->events_used++];
}

struct StageTimer {  // brackets one launch with events when kernel timing is on
    rt_ctx* c; hipStream_t st; Span s{}; bool on;
    StageTimer(rt_ctx* ctx, int stage, hipStream_t stream = nullptr)
        : c(ctx), st(stream ? stream : ctx->stream), on((ctx->flags & RT_KERNEL_TIMING) != 0) {
        if (!on) return;
        s.stage = stage; s.a = take_event(c); s.b = take_event(c);
        if (!s.a || !s.b) { on = false; return; }
        (void)hipEventRecord(s.a, st);
    }
    ~StageTimer() {
        if (!on) return;
        (void)hipEventRecord(s.b, st);
        c->spans.push_back(s);
    }
};

void begin_call(rt_ctx* c) {
    c->events_used = 0;
    c->spans.clear();
    if (c->flags & RT_KERNEL_TIMING) {
        if (!c->call_begin) (void)hipEventCreate(&c->call_begin);
        if (!c->call_end) (void)hipEventCreate(&c->call_end);
        (void)hipEventRecord(c->call_begin, c->stream);
    }
}
void end_call(rt_ctx* c) {
    if (c->flags & RT_KERNEL_TIMING) { (void)hipEventRecord(c->call_end, c->stream); c->timings_dirty = true; }
}

int resolve_timings(rt_ctx* c) {
    if (!c->timings_dirty) return RT_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    rt_timings t{};
    (void)hipEventElapsedTime(&t.total_ms, c->call_begin, c->call_end);
    for (const Span& s : c->spans) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, s.a, s.b);
        switch (s.stage) {
            case ST_RAYGEN: t.raygen_ms += ms; t.raygen_launches++; break;
            case ST_EXTEND: t.extend_ms += ms; t.extend_launches++; break;
            case ST_SHADE: t.shade_ms += ms; t.shade_launches++; break;
            case ST_CONNECT: t.connect_ms += ms; t.connect_launches++; break;
            default: t.accumulate_ms += ms; t.accumulate_launches++; break;
        }
    }
    c->last = t;
    c->timings_dirty = false;
    return RT_OK;
}

int alloc_paths_for(rt_ctx* c, uint32_t fpp_limit);

// Path buffers for the configured number of frames per pass. The automatic size (a few GiB per lane, chosen for the
// 288 GB of an MI355X) is halved until the allocation succeeds, so the library also runs beside other users of the device.
int alloc_paths(rt_ctx* c) {
    const bool automatic = c->fpp_cfg == 0 && !(c->flags & RT_REF_RGBA8_ACCUM);
    for (uint32_t limit = 64;; limit /= 2) {
        int rc = alloc_paths_for(c, limit);
        if (rc != RT_ERR_OOM || !automatic || limit <= 1 || c->fpp <= 1) return rc;
        for (uint32_t i = 0; i < c->n_lanes; ++i) {          // give back what the failed attempt got
            Lane& l = c->lanes[i];
            for (DevBuf* b : l.path_buffers()) b->release();
        }
        (void)hipGetLastError();
        limit = std::min(limit, c->fpp);
    }
}

int alloc_paths_for(rt_ctx* c, uint32_t fpp_limit) {
    uint32_t fpp = c->fpp_cfg;
    if (c->flags & RT_REF_RGBA8_ACCUM) fpp = 1;
    if (fpp == 0) {
        const uint64_t target = 64ull << 20;  // ~64 M paths per lane keep every CU fed through the late, sparse bounces (~10 GiB of path state and queues each)
        fpp = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(fpp_limit, target / std::max<uint32_t>(1, c->npix))));
    }
    c->fpp = fpp;
    uint64_t slots = uint64_t(fpp) * c->npix;
    // the kernels address the 16-byte path and queue records with 32-bit byte offsets: every queue position (a region of
    // shard_cap per shard) has to stay below 2^28. (Work distribution adds whole chunk strides to 32-bit item indices,
    // which this keeps clear of wrap-around as well.)
    if ((slots + kBlock * kQueueShards) > (1ull << 28))
        return fail(c, RT_ERR_UNSUPPORTED, "resolution x frames_per_pass exceeds 2^28 path slots (268 M paths per pass)");
    c->slot_cap = uint32_t(slots);
    // append regions: shard k receives at most ceil(ceil(slots/256)/shards) block-iterations of 256 records
    auto shard_cap_for = [](uint64_t n) { return uint32_t(((n + kBlock - 1) / kBlock + kQueueShards - 1) / kQueueShards) * kBlock; };
    c->shard_cap = shard_cap_for(slots);
    for (uint32_t i = 0; i < c->n_lanes; ++i) {
        Lane& l = c->lanes[i];
        // the first two lanes take passes of any size; the others only ever see short ones (see rt_render)
        l.slot_cap = i < kFullLanes || !c->second_lane_for_short_passes ? slots : std::min<uint64_t>(slots, kSmallPassSlots);
        l.shard_cap = shard_cap_for(l.slot_cap);
        const size_t v4 = size_t(l.shard_cap) * kQueueShards * sizeof(float4);      // >= slot_cap records: holds bounce 0 as well
        for (DevBuf* b : l.path_buffers()) HIPCHK(c, b->ensure(v4));
        HIPCHK(c, l.counts.ensure(sizeof(uint32_t) * Lane::counts_words()));
        HIPCHK(c, hipMemsetAsync(l.counts.p, 0, l.counts.bytes, c->stream));
    }
    HIPCHK(c, c->seeds.ensure(size_t(c->npix) * 4));
    HIPCHK(c, c->sum.ensure(size_t(c->npix) * sizeof(float4)));
    HIPCHK(c, c->accum8.ensure(size_t(c->npix) * 4));
    HIPCHK(c, c->fb.ensure(size_t(c->npix) * 4));
    return RT_OK;
}

int clear_accum(rt_ctx* c) {
    HIPCHK(c, hipMemsetAsync(c->sum.p, 0, size_t(c->npix) * sizeof(float4), c->stream));
    HIPCHK(c, hipMemsetAsync(c->accum8.p, 0, size_t(c->npix) * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->fb.p, 0, size_t(c->npix) * 4, c->stream));
    c->local_frame = 0;
    return RT_OK;
}

// Camera rays start within the aperture around the camera location: are they where the compressed tree's boxes are
// conservative? (Re-evaluated when the camera or the scene changes.)
// The cost of a path depends on the scene, the camera and the resolution: forget the measurement when one of them changes.
void forget_pass_cost(rt_ctx* c) {
    c->ns_per_slot = 0.0f;
    for (Lane& l : c->lanes) l.timed = false;
}

void update_camera_limit(rt_ctx* c) {
    const float reach = std::fabs(c->cam.aperture_radius);
    bool ok = true;
    for (int k = 0; k < 3; ++k) ok = ok && (std::fabs(c->cam.location[k]) + reach <= c->sd.origin_limit);
    c->camera_within_limit = ok;
}

void refresh_camera(rt_ctx* c) {
    c->cam.resolution[0] = c->W;
    c->cam.resolution[1] = c->H;
    float ratio = float(c->W) / float(c->H);  // viewport_uv, camera_buffer_obj.rs:26-33
    if (ratio > 1.0f) { c->cam.view_port_uv[0] = ratio; c->cam.view_port_uv[1] = 1.0f; }
    else { c->cam.view_port_uv[0] = 1.0f; c->cam.view_port_uv[1] = 1.0f / ratio; }
    c->cam.frame = 0;
}

// One wavefront pass on lane `l`: `frames` samples of every pixel, bounce by bounce. Leaves radiance in l.rad.
int run_pass(rt_ctx* c, Lane& l, hipStream_t st, uint32_t frames, uint32_t frame_first, uint32_t frame_stride, uint32_t* seeds_out,
             bool side_stream_ok = true, bool beside_another_long_pass = false) {
    const uint32_t npix = c->npix_pass();
    const uint32_t n_slots = frames * npix;
    if (uint64_t(n_slots) > l.slot_cap) return fail(c, RT_ERR_DEVICE, "internal: a pass larger than its lane's path buffers");
    const bool count = (c->flags & RT_COUNT_TRAVERSAL) != 0;
    const PathDev state[2] = {l.state(0), l.state(1)};
    ShadowDev sq = l.sq();
    float4* hit = l.hit.as<float4>();
    float4* rad = l.rad.as<float4>();
    HIPCHK(c, hipMemsetAsync(l.counts.p, 0, Lane::counts_words(c->max_bounces) * 4, st));   // cursors and the append counters of the bounces in use
    if (n_slots == 0) return RT_OK;                // a tile slice that owns no tile of a small image: nothing to trace
    {
        StageTimer tm(c, ST_RAYGEN, st);
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(c, n_slots, 8)), dim3(kBlock), 0, st, c->cam, state[0], rad, npix, n_slots,
                           frame_first, frame_stride, seeds_out, c->tiles());
    }
    // A pass short enough for its shadow rays to go to the side stream, in a context whose launches can run side by side at
    // all (side stream or several lanes), gets traversal grids of ONE workgroup per compute unit: a full grid holds every
    // wave slot and all the LDS until its last ray is done, so another traversal launch can only follow it; two half grids
    // run beside each other. 1080p, samples per call 1 / 2 / 3 / 4 / 6: 3.92 -> 3.49, 2.91 -> 2.77, 2.63 -> 2.51, 2.44 -> 2.37,
    // 2.28 -> 2.24 ms per sample; one sample per call queued back to back 2.95 -> 2.65 (car 1.41 -> 1.25, interior 4.60 ->
    // 4.08, glass 1.81 -> 1.64). From 8 samples per call on the full grid wins (2.15 vs 2.69), and so it does where nothing
    // overlaps (one lane without side stream, 4 samples per call: 2.45 vs 2.95). profiles/r3_trace_grid_by_call_size.txt
    const bool side_by_side = !(c->flags & (RT_KERNEL_TIMING | RT_SINGLE_STREAM)) && (c->n_lanes > 1 || (c->overlap_connect && l.side != nullptr));
    // (A long pass that runs beside another long pass of the same call, on the other lane, gets half grids for the same reason:
    // 64 samples per call at 1080p, one lane / two lanes with full grids / with half grids: Cornell box 10250 / 10440 / 12120
    // Mrays/s, glass blob 7050 / 7190 / 7340, car 5820 / 5600 / 5830.)
    const bool half_grids = (side_by_side && uint64_t(n_slots) <= c->overlap_slots) || beside_another_long_pass;
    const uint32_t trace_grid = trace_grid_for(c, n_slots, half_grids ? c->trace_blocks_short : c->trace_blocks_per_cu);
    const uint32_t late_grid = trace_grid_for(c, n_slots, c->trace_blocks_late);   // launches of a long pass that overlap (see below)
    const uint32_t shade_grid = grid_for(c, n_slots, c->shade_blocks_per_cu);
    // Shadow rays on the side stream only for passes on the lane's own stream, and never when launches are timed one by one.
    // ... and only for short passes: a launch of a few frames is mostly run-down, which the other grid fills (one 1080p sample per
    // call: 4.4 instead of 5.1 ms with the host waiting, 3.4 instead of 3.8 ms queued back to back); from about eight frames
    // on the two grids only take each other's cache lines (2.43 vs 2.47 ms per sample at 8 frames, 2.23 vs 2.27 at 20).
    // In a long pass the later bounces are sparse whatever the scene: from the fifth on their shadow rays go to the side stream
    // as well, and the launches that then overlap (connect of bounce b, extend of bounce b + 1) run on half-size grids like
    // those of a short pass. 64 samples per call, Mrays/s: 24-bounce glass 5975, 5983 -> 6295, 6146; atrium 6700, 6611 ->
    // 6714, 6741; interior 4K 4211, 4209 -> 4223, 4242. With full grids the same overlap gained nothing from bounce 4 (6674)
    // and cost 1-2 % from bounce 3 (6579); profiles/r3_late_bounce_overlap.txt.
    // `side_stream_ok`: no other lane has a pass in flight. When one has, that pass fills this one's run-downs already, and the
    // side stream, which is the next lane's stream, is not free.
    const bool may_overlap = side_stream_ok && c->overlap_connect && st == l.stream && l.side != nullptr && !(c->flags & (RT_KERNEL_TIMING | RT_SINGLE_STREAM));
    const bool overlap_all = may_overlap && uint64_t(n_slots) <= c->overlap_slots;
    bool connect_pending = false;
    const bool lds_tables = c->sd.n_mat <= kLdsMats && c->sd.n_emissive <= kLdsLights;
    // the per-path kernel walks the compressed tree only, keeps no traversal counters and is not timed as a stage
    const bool offer_tail = c->fast && !count && c->tail_max_paths != 0 && n_slots <= c->tail_pass_slots && !(c->flags & RT_KERNEL_TIMING);
    for (uint32_t b = 0; b < c->max_bounces; ++b) {
        // this bounce's paths: every slot at bounce 0, afterwards whatever the previous shade appended to its regions
        const Regions rin = b == 0 ? Regions{nullptr, l.shard_cap, n_slots} : Regions{l.shard_counts(b - 1), l.shard_cap, 0u};
        const PathDev in = state[b & 1], out = state[(b + 1) & 1];
        if (offer_tail && b >= c->tail_from) {
            // its vertices add to the radiance slots the previous bounce's shadow rays still use
            if (connect_pending) { HIPCHK(c, hipStreamWaitEvent(st, l.connected, 0)); connect_pending = false; }
            const TailParams tp{b, c->max_bounces, c->flags, l.shard_counts(b - 1), l.shard_cap, c->tail_max_paths, rad, l.tail_done(b), c->tstats()};
            const uint32_t grid = (c->tail_max_paths + kTailBlock - 1) / kTailBlock;
            const bool any = c->sd.has_glass == 0;
            if (!c->exact_arith) rtint::launch_tail_fast(lds_tables, any, grid, st, c->sd, in, tp);
            else if (lds_tables) {
                if (any) hipLaunchKernelGGL((k_path_tail<true, true>), dim3(grid), dim3(kTailBlock), 0, st, c->sd, in, tp);
                else hipLaunchKernelGGL((k_path_tail<true, false>), dim3(grid), dim3(kTailBlock), 0, st, c->sd, in, tp);
            } else {
                if (any) hipLaunchKernelGGL((k_path_tail<false, true>), dim3(grid), dim3(kTailBlock), 0, st, c->sd, in, tp);
                else hipLaunchKernelGGL((k_path_tail<false, false>), dim3(grid), dim3(kTailBlock), 0, st, c->sd, in, tp);
            }
        }
        {
            StageTimer tm(c, ST_EXTEND, st);
            // camera rays may start anywhere; every later ray starts on a surface, inside the scene bounds
            const bool fast = c->fast && (b != 0 || c->camera_within_limit);
            auto kern = count ? (fast ? k_extend<true, true> : k_extend<true, false>) : (fast ? k_extend<false, true> : k_extend<false, false>);
            const bool beside_connect = !overlap_all && may_overlap && b > c->overlap_from && c->sd.n_light + c->sd.n_emissive != 0;   // the previous bounce's shadow rays are on the side stream
            hipLaunchKernelGGL(kern, dim3(beside_connect ? late_grid : trace_grid), dim3(kTraceBlock), 0, st, c->sd, in, hit, rin, RT_CURSOR(l.ext_cursor(b)), c->tstats());
        }
        // shade rewrites the light-sample queue and adds to the radiance slots the previous bounce's shadow rays still use
        if (connect_pending) { HIPCHK(c, hipStreamWaitEvent(st, l.connected, 0)); connect_pending = false; }
        {
            StageTimer tm(c, ST_SHADE, st);
            ShadeParams sp{b, c->max_bounces, c->flags, rin, hit, out, l.shard_counts(b), sq, l.shadow_counts(b), l.shard_cap, rad, nullptr};
            if (!c->exact_arith) rtint::launch_shade_fast(lds_tables, false, shade_grid, st, c->sd, in, sp);
            else if (lds_tables) hipLaunchKernelGGL((k_shade<true, false>), dim3(shade_grid), dim3(kBlock), 0, st, c->sd, in, sp);
            else hipLaunchKernelGGL((k_shade<false, false>), dim3(shade_grid), dim3(kBlock), 0, st, c->sd, in, sp);
        }
        if (c->sd.n_light + c->sd.n_emissive != 0) {
            hipStream_t cs = st;
            const bool overlap = overlap_all || (may_overlap && b >= c->overlap_from);
            if (overlap) {
                HIPCHK(c, hipEventRecord(l.shaded, st));
                HIPCHK(c, hipStreamWaitEvent(l.side, l.shaded, 0));
                cs = l.side;
            }
            {
                StageTimer tm(c, ST_CONNECT, cs);
                const bool any = c->sd.has_glass == 0;
                const Regions rsh{l.shadow_counts(b), l.shard_cap, 0u};
                auto kern = c->fast ? (count ? (any ? k_connect<true, true, true> : k_connect<true, false, true>) : (any ? k_connect<false, true, true> : k_connect<false, false, true>))
                                    : (count ? (any ? k_connect<true, true, false> : k_connect<true, false, false>) : (any ? k_connect<false, true, false> : k_connect<false, false, false>));
                hipLaunchKernelGGL(kern, dim3(overlap && !overlap_all ? late_grid : trace_grid), dim3(kTraceBlock), 0, cs, c->sd, sq, rad, rsh, RT_CURSOR(l.con_cursor(b)), c->tstats());
            }
            if (overlap) { HIPCHK(c, hipEventRecord(l.connected, l.side)); connect_pending = true; }
        }
    }
    if (connect_pending) HIPCHK(c, hipStreamWaitEvent(st, l.connected, 0));
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

// Folds the pass's queue lengths into the work counters; runs on the stream that consumed the pass.
int finish_pass(rt_ctx* c, Lane& l, hipStream_t st, uint32_t n_slots) {
    hipLaunchKernelGGL(k_finish_pass, dim3(1), dim3(64), 0, st, l.shard_counts(0), c->max_bounces, n_slots, c->tstats());
    HIPCHK(c, hipGetLastError());
    return RT_OK;
}

// Queues one pass on its lane's stream: directly, or, for short passes, as a HIP graph captured from the second such pass in a
// row and replayed afterwards. A one-sample pass is 26 kernel launches, two dozen event operations and a memset, 0.38 ms of host time
// and 4-26 % of the device time between kernels; replayed, it is one launch. Everything a pass's kernels are given is fixed
// while scene, camera, sizes and slices stay as they are (graph_epoch) except the first frame index, a kernel argument of the
// ray-generation node that is updated before every replay.
int queue_pass(rt_ctx* c, Lane& l, uint32_t frames, uint32_t frame_first, bool side_ok, bool beside_another_long_pass) {
    const uint32_t npix = c->npix_pass();
    const uint64_t n_slots = uint64_t(frames) * npix;
    if (!c->use_graphs || n_slots == 0 || n_slots > c->graph_slots || beside_another_long_pass || (c->flags & (RT_KERNEL_TIMING | RT_COUNT_TRAVERSAL | RT_SINGLE_STREAM)))
        return run_pass(c, l, l.stream, frames, frame_first, c->slice_stride, nullptr, side_ok, beside_another_long_pass);
    Lane::PassGraph* g = nullptr;
    for (Lane::PassGraph& k : l.graphs)
        if (k.exec && k.epoch == c->graph_epoch && k.frames == frames && k.side_ok == side_ok) g = &k;
    if (!g) {
        // capture a shape the second time it comes by: capture and instantiation cost about as much as four direct passes, which a
        // call size that is used once would never get back
        const bool seen = l.seen_epoch == c->graph_epoch && l.seen_frames == frames && l.seen_side_ok == side_ok;
        l.seen_epoch = c->graph_epoch; l.seen_frames = frames; l.seen_side_ok = side_ok;
        if (!seen) return run_pass(c, l, l.stream, frames, frame_first, c->slice_stride, nullptr, side_ok);
        g = &l.graphs[l.next_graph];
        l.next_graph = (l.next_graph + 1) % (sizeof(l.graphs) / sizeof(l.graphs[0]));
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        *g = Lane::PassGraph{};
        HIPCHK(c, hipStreamBeginCapture(l.stream, hipStreamCaptureModeThreadLocal));
        const int rc = run_pass(c, l, l.stream, frames, frame_first, c->slice_stride, nullptr, side_ok);
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(l.stream, &graph);
        if (rc != RT_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIPCHK(c, e);
        size_t n_nodes = 0;
        HIPCHK(c, hipGraphGetNodes(graph, nullptr, &n_nodes));
        std::vector<hipGraphNode_t> nodes(n_nodes);
        HIPCHK(c, hipGraphGetNodes(graph, nodes.data(), &n_nodes));
        hipGraphNode_t raygen = nullptr;
        for (hipGraphNode_t nd : nodes) {
            hipGraphNodeType ty;
            if (hipGraphNodeGetType(nd, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
            hipKernelNodeParams kp{};
            if (hipGraphKernelNodeGetParams(nd, &kp) == hipSuccess && kp.func == reinterpret_cast<void*>(k_raygen)) raygen = nd;
        }
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = raygen ? hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) : hipErrorInvalidValue;
        if (ie != hipSuccess) {
            // no graph for this pass: queue it the ordinary way and stop trying
            (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            c->use_graphs = false;
            return run_pass(c, l, l.stream, frames, frame_first, c->slice_stride, nullptr, side_ok);
        }
        g->graph = graph; g->exec = exec; g->raygen = raygen; g->epoch = c->graph_epoch; g->frames = frames; g->side_ok = side_ok;
    } else {
        // the same arguments run_pass gave the node, with this pass's first frame index
        rt_camera cam = c->cam;
        PathDev pd = l.state(0);
        float4* rad = l.rad.as<float4>();
        uint32_t a_npix = npix, a_slots = uint32_t(n_slots), a_first = frame_first, a_stride = c->slice_stride;
        uint32_t* seeds_out = nullptr;
        TileSlice tiles = c->tiles();
        void* args[] = {&cam, &pd, &rad, &a_npix, &a_slots, &a_first, &a_stride, &seeds_out, &tiles};
        hipKernelNodeParams kp{};
        kp.func = reinterpret_cast<void*>(k_raygen);
        kp.gridDim = dim3(grid_for(c, n_slots, 8));
        kp.blockDim = dim3(kBlock);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        HIPCHK(c, hipGraphExecKernelNodeSetParams(g->exec, g->raygen, &kp));
    }
    HIPCHK(c, hipGraphLaunch(g->exec, l.stream));
    return RT_OK;
}

int ready_to_render(rt_ctx* c) {
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (!c->have_camera) return fail(c, RT_ERR_NO_SCENE, "no camera set");
    if (!c->paths_ok) return fail(c, RT_ERR_OOM, "no path buffers: the last resize ran out of device memory");
    return RT_OK;
}

// n x 3 floats on the host -> n float4 on the device (w from `w`, or 0).
int upload_vec3(rt_ctx* c, DevBuf& buf, uint32_t n, const float* xyz3, const float* w) {
    std::vector<float> tmp(size_t(n) * 4);
    for (uint32_t i = 0; i < n; ++i) {
        tmp[4 * i] = xyz3[3 * i]; tmp[4 * i + 1] = xyz3[3 * i + 1]; tmp[4 * i + 2] = xyz3[3 * i + 2];
        tmp[4 * i + 3] = w ? w[i] : 0.0f;
    }
    HIPCHK(c, buf.ensure(tmp.size() * 4));
    HIPCHK(c, hipMemcpyAsync(buf.p, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

}  // namespace

// =====================================================================================================

extern "C" const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : rtint::thread_error(); }

extern "C" int rt_create(const rt_config* cfg, rt_ctx** out) {
    if (!cfg || !out) return fail(nullptr, RT_ERR_INVALID, "rt_create: null argument");
    if (cfg->width == 0 || cfg->height == 0) return fail(nullptr, RT_ERR_INVALID, "rt_create: zero resolution");
    uint32_t mb = cfg->max_bounces ? cfg->max_bounces : 24;
    if (mb > kMaxBounces) return fail(nullptr, RT_ERR_UNSUPPORTED, "rt_create: max_bounces above 64");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(nullptr, RT_ERR_DEVICE, std::string("rt_create: no HIP device available (") + hipGetErrorString(e) + ")");
    if (cfg->device < 0 || cfg->device >= n_dev) return fail(nullptr, RT_ERR_INVALID, "rt_create: device ordinal out of range");
    rt_ctx* c = new rt_ctx;
    c->device = cfg->device;
    c->W = cfg->width; c->H = cfg->height; c->npix = cfg->width * cfg->height;
    c->max_bounces = mb; c->flags = cfg->flags; c->fpp_cfg = cfg->frames_per_pass;
    c->fast_requested = (cfg->flags & RT_EXACT_ORDER) == 0;
    c->fast = c->fast_requested;
    c->exact_arith = (cfg->flags & (RT_EXACT_ARITH | RT_EXACT_ORDER)) != 0;
    auto bail = [&](int code) { std::string m = c->err; rt_destroy(c); fail(nullptr, code, m); return code; };
    if (uint64_t(cfg->width) * cfg->height > 0x7FFFFFFFull) { c->err = "resolution too large"; return bail(RT_ERR_UNSUPPORTED); }
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return bail(rc);
    if (hipError_t se = static_scratch_for(c->device, &c->static_scratch); se != hipSuccess) {
        c->err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(se); return bail(RT_ERR_DEVICE);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) c->n_cu = std::max(1, prop.multiProcessorCount);
    // Three lanes, all but the first for short passes only: a pass of more than kSmallPassSlots paths keeps the machine busy
    // on its own, and a second concurrent pass only evicts the first one's tree from L2 / Infinity Cache (4583 vs 4446
    // Mrays/s at 32-frame passes), but short passes (an interactive host renders one sample per call) spend most of
    // their time running persistent grids down, which the other lanes' passes fill. Three, because with the main stream
    // that makes four streams, one per hardware queue HIP gives a process by default: streams that share a queue run one
    // after the other whatever their events say, and which streams share is up to the runtime (with four lanes and a side
    // stream each, one 1080p sample per call took 3.16 or 3.77 ms depending on what the process had created before; with
    // four streams it takes 3.2 ms either way, and the host that waits for every sample gets 4.33 instead of 4.5 ms).
    // profiles/r3_lanes_and_hw_queues.txt. The third lane is sized for short passes only (kFullLanes). RT_HIP_LANES=1..8
    // fixes the number of lanes and alternates every pass (bit-identical results either way).
    c->n_lanes = (c->flags & (RT_SINGLE_STREAM | RT_KERNEL_TIMING | RT_REF_RGBA8_ACCUM)) ? 1 : kDefaultLanes;
    c->second_lane_for_short_passes = true;
    if (const char* env = std::getenv("RT_HIP_TRACE_BLOCKS")) c->trace_blocks_per_cu = c->trace_blocks_short = c->trace_blocks_late = uint32_t(std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_TRACE_BLOCKS_LATE")) c->trace_blocks_late = uint32_t(std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_OVERLAP_FROM")) c->overlap_from = uint32_t(std::max(0, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_TRACE_BLOCKS_SHORT")) c->trace_blocks_short = uint32_t(std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_SHADE_BLOCKS")) c->shade_blocks_per_cu = uint32_t(std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_OVERLAP_CONNECT")) c->overlap_connect = std::atoi(env) != 0;
    if (const char* env = std::getenv("RT_HIP_OVERLAP_SLOTS")) c->overlap_slots = uint64_t(std::max(0ll, std::atoll(env)));   // tuning aid
    if (const char* env = std::getenv("RT_HIP_TAIL_FROM")) c->tail_from = uint32_t(std::max(1, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_TAIL_MAX")) c->tail_max_paths = uint32_t(std::max(0, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_TAIL_SLOTS")) c->tail_pass_slots = uint32_t(std::max(0, std::atoi(env)));
    if (const char* env = std::getenv("RT_HIP_GRAPH")) { c->use_graphs = std::atoi(env) != 0; c->graph_slots = kGraphPassSlots; }
    if (const char* env = std::getenv("RT_HIP_CHEAP_NS")) c->cheap_slot_ns = float(std::atof(env));   // tuning aid: 0 = long passes never share the machine
    if (const char* env = std::getenv("RT_HIP_LANES")) {   // tuning aid: 1..4 concurrent passes
        int v = std::atoi(env);
        if (v >= 1 && v <= int(kMaxLanes) && !(c->flags & (RT_SINGLE_STREAM | RT_KERNEL_TIMING | RT_REF_RGBA8_ACCUM))) { c->n_lanes = uint32_t(v); c->second_lane_for_short_passes = false; }
    }
    // Measurement aid: RT_HIP_CUS=<n> confines the render streams to n compute units (a CU-masked stream) and sizes the
    // persistent grids for n. The driver deals mask bits round-robin over the XCDs, so n is rounded up to a multiple of 8
    // and every XCD keeps n/8 CUs (an XCD left without CUs would still be handed workgroups and never run them). L2,
    // Infinity Cache and HBM stay whole: the figure is an upper bound for a true one-eighth partition.
    uint32_t cu_mask[8] = {0};
    bool masked = false;
    if (const char* env = std::getenv("RT_HIP_CUS")) {
        int v = std::atoi(env);
        if (v >= 8 && v < c->n_cu && c->n_cu <= 256) {
            v = (v + 7) & ~7;
            for (int b = 0; b < v; ++b) cu_mask[b >> 5] |= 1u << (b & 31);
            c->n_cu = v;
            masked = true;
            c->use_graphs = false;      // a replayed graph's kernel nodes need not run under the capturing stream's CU mask
        }
    }
    // Every stream of a confined context carries the mask, the main stream (accumulate, clears, copies) included: a process
    // whose context launched on masked and on ordinary streams side by side did not return from the HIP runtime's exit
    // handler, one whose streams are all masked does.
    {
        // lanes + main = four streams by default, one per hardware queue HIP gives a process unless told otherwise; a lane has no
        // side stream of its own but borrows the next lane's stream, which is idle whenever a side stream is used at all
        auto make = [&](hipStream_t* st) { return masked ? hipExtStreamCreateWithCUMask(st, 8, cu_mask) : hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
        for (uint32_t i = 0; i < c->n_lanes; ++i)
            if (make(&c->lanes[i].stream) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(RT_ERR_DEVICE); }
        if (make(&c->stream) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(RT_ERR_DEVICE); }
        if (c->n_lanes == 1 && !(c->flags & (RT_SINGLE_STREAM | RT_KERNEL_TIMING))) {
            if (make(&c->lanes[0].side) != hipSuccess) { c->err = "hipStreamCreate failed"; return bail(RT_ERR_DEVICE); }
            c->lanes[0].side_owned = true;
        }
        for (uint32_t i = 0; c->n_lanes > 1 && i < c->n_lanes; ++i) c->lanes[i].side = c->lanes[(i + 1) % c->n_lanes].stream;
    }
    for (uint32_t i = 0; i < c->n_lanes; ++i) {
        Lane& l = c->lanes[i];
        hipError_t se = hipEventCreateWithFlags(&l.shaded, hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreateWithFlags(&l.connected, hipEventDisableTiming);
        if (se == hipSuccess) se = hipEventCreate(&l.t_begin);
        if (se == hipSuccess) se = hipEventCreate(&l.t_end);
        if (se != hipSuccess || hipEventCreateWithFlags(&l.pass_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l.consumed, hipEventDisableTiming) != hipSuccess) { c->err = "lane stream/event creation failed"; return bail(RT_ERR_DEVICE); }
    }
    if (hipEventCreateWithFlags(&c->call_mark, hipEventDisableTiming) != hipSuccess) { c->err = "hipEventCreate failed"; return bail(RT_ERR_DEVICE); }
    if ((rc = alloc_paths(c)) != RT_OK) return bail(rc);
    c->paths_ok = true;
    if (c->stats.ensure(sizeof(TraceStats)) != hipSuccess) { c->err = "counter allocation failed"; return bail(RT_ERR_OOM); }
    (void)hipMemsetAsync(c->stats.p, 0, c->stats.bytes, c->stream);
    // sRGB decode table, identical to what a CPU evaluation computes: exact curve in double, rounded once
    float lut[256];
    for (int i = 0; i < 256; ++i) {
        double v = i / 255.0;
        lut[i] = float(v <= 0.04045 ? v / 12.92 : std::pow((v + 0.055) / 1.055, 2.4));
    }
    if (c->srgb.ensure(sizeof(lut)) != hipSuccess || hipMemcpy(c->srgb.p, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess) {
        c->err = "sRGB table upload failed"; return bail(RT_ERR_DEVICE);
    }
    if ((rc = rt_upload_textures(c, nullptr, 0, nullptr)) != RT_OK) return bail(rc);
    if ((rc = clear_accum(c)) != RT_OK) return bail(rc);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "stream sync failed"; return bail(RT_ERR_DEVICE); }
    *out = c;
    return RT_OK;
}

extern "C" void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (Lane& l : c->lanes) {
        l.drop_graphs();
        if (l.pass_done) (void)hipEventDestroy(l.pass_done);
        if (l.consumed) (void)hipEventDestroy(l.consumed);
        if (l.t_begin) (void)hipEventDestroy(l.t_begin);
        if (l.t_end) (void)hipEventDestroy(l.t_end);
        if (l.shaded) (void)hipEventDestroy(l.shaded);
        if (l.connected) (void)hipEventDestroy(l.connected);
        if (l.side && l.side_owned) (void)hipStreamDestroy(l.side);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    if (c->call_mark) (void)hipEventDestroy(c->call_mark);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    if (c->call_begin) (void)hipEventDestroy(c->call_begin);
    if (c->call_end) (void)hipEventDestroy(c->call_end);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int rt_upload_scene(rt_ctx* c, const void* blob, const uint64_t sizes[6], uint32_t n_bvh, uint32_t n_mat,
                               uint32_t n_tri, uint32_t n_light, uint32_t n_emissive) {
    if (!c) return RT_ERR_INVALID;
    if (!blob || !sizes) return fail(c, RT_ERR_INVALID, "rt_upload_scene: null argument");
    uint64_t want[6];
    rt_scene_buffer_sizes(n_bvh, n_mat, n_tri, n_light, n_emissive, want);
    for (int s = 0; s < 6; ++s)
        if (sizes[s] < want[s]) return fail(c, RT_ERR_INVALID, "rt_upload_scene: section " + std::to_string(s) + " smaller than its element count requires");
    if (n_bvh == 0) return fail(c, RT_ERR_INVALID, "rt_upload_scene: a scene has at least the root node");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    const uint8_t* p = static_cast<const uint8_t*>(blob);
    const rt_bvh_node* nodes = reinterpret_cast<const rt_bvh_node*>(p); p += sizes[0];
    const rt_material* mats = reinterpret_cast<const rt_material*>(p); p += sizes[1];
    const rt_triangle* tris = reinterpret_cast<const rt_triangle*>(p); p += sizes[2];
    const rt_light* lights = reinterpret_cast<const rt_light*>(p); p += sizes[3];
    const rt_emissive_tri* emis = reinterpret_cast<const rt_emissive_tri*>(p); p += sizes[4];
    const float* cdf = reinterpret_cast<const float*>(p);

    rtint::GpuScene gs;
    std::string err;
    rc = rtint::convert_scene(nodes, n_bvh, tris, n_tri, mats, n_mat, gs, err);
    if (rc != RT_OK) return fail(c, rc, "rt_upload_scene: " + err);

    // device light table: {v0, area} {v1, Le.r} {v2, Le.g} {unit normal, Le.b} per emissive triangle
    std::vector<float> lt(size_t(n_emissive) * 16);
    for (uint32_t i = 0; i < n_emissive; ++i) {
        const rt_emissive_tri& e = emis[i];
        if (e.tri_index >= n_tri) return fail(c, RT_ERR_INVALID, "rt_upload_scene: emissive entry references a missing triangle");
        const rt_triangle& t = tris[e.tri_index];
        const rt_material& m = mats[t.material_idx];
        const rtint::TriIsect& q = gs.isect[e.tri_index];
        float* o = &lt[size_t(i) * 16];
        o[0] = t.v[0][0]; o[1] = t.v[0][1]; o[2] = t.v[0][2]; o[3] = e.area;
        o[4] = t.v[1][0]; o[5] = t.v[1][1]; o[6] = t.v[1][2]; o[7] = m.emission[0];
        o[8] = t.v[2][0]; o[9] = t.v[2][1]; o[10] = t.v[2][2]; o[11] = m.emission[1];
        o[12] = q.a[3];   o[13] = q.b[3];   o[14] = q.c[3];    o[15] = m.emission[2];
    }

    // Every table is staged in a fresh buffer and all of them are committed together, so an upload that fails half-way
    // (out of memory, a failed copy) leaves the previous scene in place and renderable.
    DevBuf nb_nodes, nb_wide, nb_cw, nb_isect, nb_shade, nb_mats, nb_lights, nb_lt, nb_cdf;
    auto put = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(bytes);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIPCHK(c, put(nb_nodes, gs.nodes.data(), gs.nodes.size() * sizeof(rtint::NodePair)));
    HIPCHK(c, put(nb_wide, gs.wide.data(), gs.wide.size() * sizeof(rtint::NodeQuad)));
    HIPCHK(c, put(nb_cw, gs.cw.data(), gs.cw.size() * sizeof(rtint::NodeCw)));
    HIPCHK(c, put(nb_isect, gs.isect.data(), gs.isect.size() * sizeof(rtint::TriIsect)));
    HIPCHK(c, put(nb_shade, gs.shade.data(), gs.shade.size() * sizeof(rtint::TriShade)));
    HIPCHK(c, put(nb_mats, mats, size_t(n_mat) * sizeof(rt_material)));
    HIPCHK(c, put(nb_lights, lights, size_t(n_light) * sizeof(rt_light)));
    HIPCHK(c, put(nb_lt, lt.data(), lt.size() * 4));
    HIPCHK(c, put(nb_cdf, cdf, size_t(n_emissive) * 4));
    // nothing queued earlier may still read the tables about to be replaced (every pass on a lane is followed by its
    // accumulation on the main stream, so the main stream is the last to finish)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nodes.swap(nb_nodes); c->wide.swap(nb_wide); c->cw.swap(nb_cw); c->isect.swap(nb_isect); c->shade.swap(nb_shade);
    c->mats.swap(nb_mats); c->lights.swap(nb_lights); c->light_tris.swap(nb_lt); c->cdf.swap(nb_cdf);
    c->h_nodes.swap(gs.nodes); c->h_wide.swap(gs.wide); c->h_cw.swap(gs.cw);

    SceneDev& sd = c->sd;
    sd.nodes = c->nodes.as<float4>(); sd.isect = c->isect.as<float4>(); sd.shade = c->shade.as<float4>();
    sd.mats = c->mats.as<float4>(); sd.lights = c->lights.as<float4>(); sd.light_tris = c->light_tris.as<float4>();
    sd.cdf = c->cdf.as<float>();
    sd.has_glass = 0;
    for (uint32_t i = 0; i < n_mat; ++i) sd.has_glass |= mats[i].spec_trans > 0.5f ? 1u : 0u;
    sd.wide = c->wide.as<float4>();
    sd.wide_root = c->h_wide.empty() ? gs.root_ref : gs.wide_root;     // a leaf (or empty) root has no interior records at all
    sd.n_hot_wide = gs.n_hot_wide;
    sd.cw = c->cw.as<float4>();
    sd.origin_limit = 30.0f * gs.scene_size;
    c->fast = c->fast_requested && gs.cw_complete;
    sd.root_ref = gs.root_ref; sd.n_hot = gs.n_hot; sd.n_tri = n_tri; sd.n_light = n_light; sd.n_emissive = n_emissive; sd.n_mat = n_mat;
    // The shader sizes the CDF from its 16-byte padded binding; an empty table is bound as one dummy word
    // (src/vulkan/path_tracer/descriptors.rs:85-93).
    uint32_t seen = n_emissive;
    if (c->flags & RT_REF_CDF_PAD) seen = uint32_t(want[5] / 4);
    sd.cdf_len_seen = std::max(1u, seen);
    c->have_scene = true;
    update_camera_limit(c);
    forget_pass_cost(c);
    return clear_accum(c);
}

extern "C" int rt_tune_resident_set(rt_ctx* c, rt_tune_report* rep) {
    if (!c) return RT_ERR_INVALID;
    if (rep) *rep = rt_tune_report{};
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    if ((rc = ready_to_render(c)) != RT_OK) return rc;
    const size_t nb = c->h_nodes.size(), nw = c->h_wide.size();
    if (nb + nw == 0) return RT_OK;                              // the root is a leaf: nothing is kept in LDS
    HIPCHK(c, hipStreamSynchronize(c->stream));                  // every pass is followed by its accumulation on the main stream

    // one sample per pixel through the counting kernels, with the visit counters switched on; the work counters are put back afterwards
    DevBuf visits;
    HIPCHK(c, visits.ensure((nb + nw) * sizeof(uint32_t)));
    HIPCHK(c, hipMemsetAsync(visits.p, 0, (nb + nw) * sizeof(uint32_t), c->stream));
    TraceStats saved, with;
    HIPCHK(c, hipMemcpy(&saved, c->stats.p, sizeof(saved), hipMemcpyDeviceToHost));
    with = saved;
    with.visits[0] = visits.as<uint32_t>();
    with.visits[1] = visits.as<uint32_t>() + nb;
    HIPCHK(c, hipMemcpy(c->stats.p, &with, sizeof(with), hipMemcpyHostToDevice));
    const uint32_t flags = c->flags;
    c->flags = (flags | RT_COUNT_TRAVERSAL) & ~uint32_t(RT_KERNEL_TIMING);
    rc = run_pass(c, c->lanes[0], c->stream, 1, c->slice_first, 1, nullptr);
    c->flags = flags;
    const hipError_t se = hipStreamSynchronize(c->stream);
    saved.visits[0] = saved.visits[1] = nullptr;
    const hipError_t re = hipMemcpy(c->stats.p, &saved, sizeof(saved), hipMemcpyHostToDevice);
    if (rc != RT_OK) return rc;
    HIPCHK(c, se);
    HIPCHK(c, re);
    std::vector<uint32_t> v(nb + nw);
    HIPCHK(c, hipMemcpy(v.data(), visits.p, v.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));

    auto share = [](const uint32_t* cnt, size_t n, const std::vector<uint32_t>* new_of_old, uint32_t resident, uint64_t& total) {
        uint64_t in = 0;
        total = 0;
        for (size_t i = 0; i < n; ++i) {
            total += cnt[i];
            if ((new_of_old ? (*new_of_old)[i] : uint32_t(i)) < resident) in += cnt[i];
        }
        return total ? float(double(in) / double(total)) : 0.0f;
    };
    std::vector<uint32_t> perm_b, perm_w;
    const uint32_t moved_b = rtint::order_by_visits(v.data(), nb, c->sd.n_hot, perm_b);
    const uint32_t moved_w = rtint::order_by_visits(v.data() + nb, nw, c->sd.n_hot_wide, perm_w);
    if (rep) {
        const uint32_t res[3] = {std::min<uint32_t>(c->sd.n_hot, kHotLds), std::min<uint32_t>(c->sd.n_hot_wide, kHotLds), std::min<uint32_t>(c->sd.n_hot_wide, kHotWide)};
        for (int k = 0; k < 3; ++k) {
            const uint32_t* cnt = k == 0 ? v.data() : v.data() + nb;
            const size_t n = k == 0 ? nb : nw;
            uint64_t total = 0;
            rep->resident_records[k] = res[k];
            rep->resident_share_before[k] = share(cnt, n, nullptr, res[k], total);
            rep->resident_share_after[k] = share(cnt, n, k == 0 ? &perm_b : &perm_w, res[k], total);
            rep->sampled_visits[k] = total;
            rep->records_moved[k] = k == 0 ? moved_b : moved_w;
        }
    }
    if (moved_b + moved_w == 0) return RT_OK;

    // re-ordered copies, staged and committed together like an upload
    std::vector<rtint::NodePair> nodes2 = c->h_nodes;
    std::vector<rtint::NodeQuad> wide2 = c->h_wide;
    std::vector<rtint::NodeCw> cw2 = c->h_cw;
    rtint::permute_records(nodes2, perm_b, 2);
    rtint::permute_records(wide2, perm_w, 4);
    rtint::permute_records(cw2, perm_w, 4);
    DevBuf nb_nodes, nb_wide, nb_cw;
    auto put = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(bytes);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIPCHK(c, put(nb_nodes, nodes2.data(), nodes2.size() * sizeof(rtint::NodePair)));
    HIPCHK(c, put(nb_wide, wide2.data(), wide2.size() * sizeof(rtint::NodeQuad)));
    HIPCHK(c, put(nb_cw, cw2.data(), cw2.size() * sizeof(rtint::NodeCw)));
    c->nodes.swap(nb_nodes); c->wide.swap(nb_wide); c->cw.swap(nb_cw);
    c->h_nodes.swap(nodes2); c->h_wide.swap(wide2); c->h_cw.swap(cw2);
    SceneDev& sd = c->sd;
    sd.nodes = c->nodes.as<float4>(); sd.wide = c->wide.as<float4>(); sd.cw = c->cw.as<float4>();
    auto renumber = [](uint32_t ref, const std::vector<uint32_t>& perm) {
        return (!(ref & rtint::kLeafBit) && ref != rtint::kRefNone && ref < perm.size()) ? perm[ref] : ref;
    };
    sd.root_ref = renumber(sd.root_ref, perm_b);
    if (nw != 0) sd.wide_root = renumber(sd.wide_root, perm_w);
    return RT_OK;
}

extern "C" int rt_upload_textures(rt_ctx* c, const rt_texture* textures, uint32_t n, const rt_texture* skybox) {
    if (!c) return RT_ERR_INVALID;
    if (n && !textures) return fail(c, RT_ERR_INVALID, "rt_upload_textures: null texture array");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    std::vector<uint32_t> table(size_t(std::max(1u, n)) * 4, 0u);
    size_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!textures[i].pixels || textures[i].width == 0 || textures[i].height == 0)
            return fail(c, RT_ERR_INVALID, "rt_upload_textures: empty texture " + std::to_string(i));
        if (total + size_t(textures[i].width) * textures[i].height > 0xFFFFFFFFull)
            return fail(c, RT_ERR_UNSUPPORTED, "rt_upload_textures: more than 2^32 texels in total");
        table[4 * i] = uint32_t(total); table[4 * i + 1] = textures[i].width; table[4 * i + 2] = textures[i].height;
        total += size_t(textures[i].width) * textures[i].height;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->texels.ensure(std::max<size_t>(16, total * 4)));
    for (uint32_t i = 0; i < n; ++i)
        HIPCHK(c, hipMemcpy(c->texels.as<uint32_t>() + table[4 * i], textures[i].pixels,
                            size_t(textures[i].width) * textures[i].height * 4, hipMemcpyHostToDevice));
    HIPCHK(c, c->tex_table.ensure(table.size() * 4));
    HIPCHK(c, hipMemcpy(c->tex_table.p, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    TexView& tv = c->sd.tex;
    tv.texels = c->texels.as<uint32_t>(); tv.table = c->tex_table.as<uint4>(); tv.n_tex = n; tv.srgb = c->srgb.as<float>();
    if (skybox) {
        if (!skybox->pixels || skybox->width == 0 || skybox->height == 0) return fail(c, RT_ERR_INVALID, "rt_upload_textures: empty skybox");
        size_t bytes = size_t(skybox->width) * skybox->height * 4 * 6;
        HIPCHK(c, c->sky.ensure(bytes));
        HIPCHK(c, hipMemcpy(c->sky.p, skybox->pixels, bytes, hipMemcpyHostToDevice));
        tv.sky_w = skybox->width; tv.sky_h = skybox->height;
    } else {
        const uint32_t white[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        HIPCHK(c, c->sky.ensure(sizeof(white)));
        HIPCHK(c, hipMemcpy(c->sky.p, white, sizeof(white), hipMemcpyHostToDevice));
        tv.sky_w = tv.sky_h = 1;
    }
    tv.sky = c->sky.as<uint32_t>();
    tv.sky_f32 = nullptr;
    c->sky_f32.release();
    if (c->sum.p) return clear_accum(c);
    return RT_OK;
}


extern "C" int rt_upload_skybox_f32(rt_ctx* c, uint32_t width, uint32_t height, const float* rgba) {
    if (!c) return RT_ERR_INVALID;
    if (!rgba || width == 0 || height == 0) return fail(c, RT_ERR_INVALID, "rt_upload_skybox_f32: empty skybox");
    if (uint64_t(width) * height * 6 > 0x7FFFFFFFull) return fail(c, RT_ERR_UNSUPPORTED, "rt_upload_skybox_f32: more than 2^31 texels");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (Lane& l : c->lanes) if (l.stream) HIPCHK(c, hipStreamSynchronize(l.stream));      // passes in flight still read the old cube map
    const size_t bytes = size_t(width) * height * 6 * sizeof(float4);
    DevBuf fresh;
    HIPCHK(c, fresh.ensure(bytes));
    HIPCHK(c, hipMemcpy(fresh.p, rgba, bytes, hipMemcpyHostToDevice));
    c->sky_f32.swap(fresh);
    TexView& tv = c->sd.tex;
    tv.sky_w = width; tv.sky_h = height;
    tv.sky_f32 = c->sky_f32.as<float4>();
    return clear_accum(c);
}

extern "C" int rt_set_camera(rt_ctx* c, const rt_camera* cam) {
    if (!c) return RT_ERR_INVALID;
    if (!cam) return fail(c, RT_ERR_INVALID, "rt_set_camera: null camera");
    if (!(cam->focal_length != 0.0f)) return fail(c, RT_ERR_INVALID, "rt_set_camera: focal_length is zero");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    c->cam = *cam;
    refresh_camera(c);
    c->have_camera = true;
    update_camera_limit(c);
    forget_pass_cost(c);
    return clear_accum(c);
}

extern "C" int rt_resize(rt_ctx* c, uint32_t w, uint32_t h) {
    if (!c) return RT_ERR_INVALID;
    if (w == 0 || h == 0 || uint64_t(w) * h > 0x7FFFFFFFull) return fail(c, RT_ERR_INVALID, "rt_resize: bad resolution");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    forget_pass_cost(c);
    const uint32_t old_w = c->W, old_h = c->H;
    c->W = w; c->H = h; c->npix = w * h;
    c->paths_ok = false;
    if ((rc = alloc_paths(c)) != RT_OK) {
        // go back to the previous resolution; if even that fails the context refuses to render until a resize succeeds
        const std::string why = c->err;
        c->W = old_w; c->H = old_h; c->npix = old_w * old_h;
        (void)hipGetLastError();
        if (alloc_paths(c) == RT_OK) {
            c->paths_ok = true;
            refresh_camera(c);
            (void)clear_accum(c);
            return fail(c, rc, why + " (resolution left at " + std::to_string(old_w) + "x" + std::to_string(old_h) + ")");
        }
        return fail(c, rc, why + " (path buffers lost: resize again before rendering)");
    }
    c->paths_ok = true;
    refresh_camera(c);
    return clear_accum(c);
}

extern "C" int rt_reset_accumulation(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    return clear_accum(c);
}

extern "C" int rt_set_frame_slice(rt_ctx* c, uint32_t first, uint32_t stride) {
    if (!c) return RT_ERR_INVALID;
    if (stride == 0) return fail(c, RT_ERR_INVALID, "rt_set_frame_slice: stride is zero");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    c->slice_first = first; c->slice_stride = stride;
    return clear_accum(c);
}

extern "C" int rt_set_tile_slice(rt_ctx* c, uint32_t first, uint32_t stride) {
    if (!c) return RT_ERR_INVALID;
    if (stride == 0) return fail(c, RT_ERR_INVALID, "rt_set_tile_slice: stride is zero");
    if (first >= stride) return fail(c, RT_ERR_INVALID, "rt_set_tile_slice: first must be below stride");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    const uint32_t old_first = c->tile_first, old_stride = c->tile_stride;
    c->tile_first = first; c->tile_stride = stride;
    if (c->npix_pass() > c->slot_cap) {          // an image of a few pixels, whose one overhanging tile outgrows the path buffers
        c->tile_first = old_first; c->tile_stride = old_stride;
        return fail(c, RT_ERR_UNSUPPORTED, "rt_set_tile_slice: the image is too small to be cut into tile slices");
    }
    forget_pass_cost(c);
    return clear_accum(c);
}

extern "C" int rt_render(rt_ctx* c, uint32_t n_samples) {
    if (!c) return RT_ERR_INVALID;
    int rc;
    if ((rc = ready_to_render(c)) != RT_OK) return rc;
    const bool must_wait = c->lanes_must_wait;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    begin_call(c);
    // Whatever other entry points queued on the main stream (clears, uploads, stage-level calls) precedes this call's passes.
    // Between two rt_render calls with nothing in between there is no such work, and waiting for the main stream would
    // only keep this call's first pass from overlapping the previous call's last one (what a host that queues one sample
    // per call relies on: 4.85 -> 3.7 ms per sample at 1080p).
    if (must_wait) {
        HIPCHK(c, hipEventRecord(c->call_mark, c->stream));
        for (uint32_t i = 0; i < c->n_lanes; ++i) HIPCHK(c, hipStreamWaitEvent(c->lanes[i].stream, c->call_mark, 0));
    }
    // Split the call into passes of at most fpp frames, as even as possible: long calls run the largest passes the path
    // buffers hold (fewer, longer launches: less time in the ramp-down of each persistent traversal grid), short passes
    // alternate between lanes and overlap. With a fixed lane count (RT_HIP_LANES) every call is split into at least one
    // pass per lane.
    // How the long passes of a call are scheduled depends on what a path costs here. Every traversal launch ends with a few
    // waves walking its longest rays at memory latency; where paths are cheap (open scenes whose paths leave early, tiny
    // scenes) that run-down is a larger share of the pass and the call's next pass, on the other lane, fills it (64 samples
    // per call at 1080p: glass +3.4 %, Cornell box +1.5 %, car -2.5 %), where they are expensive it only evicts the first
    // pass's tree from the caches (-6 % on the atrium). The device time of the last pass that ran alone decides.
    for (uint32_t i = 0; i < c->n_lanes; ++i) {
        Lane& l = c->lanes[i];
        if (!l.timed || hipEventQuery(l.t_end) != hipSuccess) continue;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, l.t_begin, l.t_end) == hipSuccess && l.timed_slots) c->ns_per_slot = ms * 1e6f / float(l.timed_slots);
        l.timed = false;
    }
    (void)hipGetLastError();                                     // a pass still running reports "not ready": not an error
    const bool cheap_paths = c->ns_per_slot > 0.0f && c->ns_per_slot < c->cheap_slot_ns;
    const bool always_alternate = !c->second_lane_for_short_passes || (cheap_paths && c->n_lanes > 1);
    // a long call is cut into one pass per lane that takes long passes (two, unless RT_HIP_LANES fixes the lanes), no more
    const uint32_t long_pass_lanes = c->second_lane_for_short_passes ? std::min(c->n_lanes, kFullLanes) : c->n_lanes;
    // (cheap paths: the passes a call has anyway go to alternating lanes, but a call that fits one pass stays one pass: two
    // 10-frame passes side by side measured 6-7 % slower than one 20-frame pass on the car and the glass scene)
    const uint32_t min_passes = !c->second_lane_for_short_passes ? std::min<uint32_t>(long_pass_lanes, n_samples) : 1u;
    if (c->npix_pass() > c->slot_cap) return fail(c, RT_ERR_UNSUPPORTED, "rt_render: the image is too small for the tile slice that is set");
    // a tile slice has fewer pixels per frame than the path buffers were sized for: more frames fit a pass
    uint32_t fpp = c->fpp;
    if (c->tile_stride > 1u && !(c->flags & RT_REF_RGBA8_ACCUM))
        fpp = std::max(1u, std::min(c->fpp_cfg ? c->fpp_cfg : 64u, c->slot_cap / std::max(1u, c->npix_pass())));
    uint32_t n_pass = std::max<uint32_t>((n_samples + fpp - 1) / fpp, min_passes);
    const bool single_pass_call = n_pass == 1;
    // cheap paths: the long passes of this call run side by side on the two full-size lanes. They get half-size traversal grids
    // (run_pass), and none of them borrows the other lane's stream for its shadow rays: the other pass is about to be queued there
    const bool long_passes_side_by_side = c->second_lane_for_short_passes && cheap_paths && c->n_lanes > 1 && n_pass > 1;
    uint32_t left = n_samples;
    const uint32_t npix_pass = c->npix_pass(), pixels_owned = c->pixels_owned();
    while (left) {
        uint32_t frames = (left + n_pass - 1) / n_pass;
        --n_pass;
        uint32_t frame_first = c->slice_first + c->local_frame * c->slice_stride;
        const bool short_pass = uint64_t(frames) * npix_pass <= kSmallPassSlots;
        const bool alternate = always_alternate || short_pass;
        // short passes go round all lanes, long ones (when they alternate at all) only the lanes sized for them
        const uint32_t pool = short_pass || !c->second_lane_for_short_passes ? c->n_lanes : std::min(c->n_lanes, kFullLanes);
        const uint32_t lane_index = alternate && c->next_lane < pool ? c->next_lane : 0u;
        Lane& l = c->lanes[lane_index];
        if (alternate) c->next_lane = (lane_index + 1) % pool;
        if (l.busy) HIPCHK(c, hipStreamWaitEvent(l.stream, l.consumed, 0));   // its previous radiance has been accumulated
        // time the pass when it has the machine to itself as far as can be known here: the only pass of its call, every
        // other lane idle when it is queued
        bool others_busy = false;
        for (uint32_t i = 0; !others_busy && i < c->n_lanes; ++i)
            if (&c->lanes[i] != &l && c->lanes[i].busy && hipEventQuery(c->lanes[i].pass_done) != hipSuccess) others_busy = true;
        (void)hipGetLastError();
        const bool time_it = c->second_lane_for_short_passes && single_pass_call && l.t_begin != nullptr && !others_busy;
        if (time_it) HIPCHK(c, hipEventRecord(l.t_begin, l.stream));
        const bool beside_long = long_passes_side_by_side && !short_pass;
        if ((rc = queue_pass(c, l, frames, frame_first, !others_busy && !beside_long, beside_long)) != RT_OK) return rc;
        if (time_it) { HIPCHK(c, hipEventRecord(l.t_end, l.stream)); l.timed = true; l.timed_slots = uint64_t(frames) * npix_pass; }
        HIPCHK(c, hipEventRecord(l.pass_done, l.stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, l.pass_done, 0));
        {
            StageTimer tm(c, ST_ACCUM);
            if (npix_pass != 0) {
                hipLaunchKernelGGL(k_accumulate, dim3(grid_for(c, npix_pass, 8)), dim3(kBlock), 0, c->stream, l.rad.as<float4>(), c->sum.as<float4>(),
                                   c->accum8.as<uint32_t>(), c->fb.as<uint32_t>(), npix_pass, frames, frame_first, c->slice_stride,
                                   (c->flags & RT_REF_RGBA8_ACCUM) ? 1u : 0u, c->tiles());
            }
        }
        if ((rc = finish_pass(c, l, c->stream, frames * pixels_owned)) != RT_OK) return rc;     // camera rays = pixels of the image
        HIPCHK(c, hipEventRecord(l.consumed, c->stream));
        l.busy = true;
        c->local_frame += frames;
        left -= frames;
    }
    end_call(c);
    c->lanes_must_wait = false;
    return RT_OK;
}

extern "C" int rt_sync(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    const bool must_wait = c->lanes_must_wait;
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    c->lanes_must_wait = must_wait;                              // waiting queues nothing
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_poll(rt_ctx* c, int* done) {
    if (!c) return RT_ERR_INVALID;
    if (!done) return fail(c, RT_ERR_INVALID, "rt_poll: null output");
    const bool must_wait = c->lanes_must_wait;
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    c->lanes_must_wait = must_wait;                              // asking queues nothing
    const hipError_t e = hipStreamQuery(c->stream);              // every pass is followed by its accumulation on the main stream
    if (e != hipSuccess && e != hipErrorNotReady) HIPCHK(c, e);
    (void)hipGetLastError();
    *done = e == hipSuccess ? 1 : 0;
    return RT_OK;
}

extern "C" uint32_t rt_sample_count(const rt_ctx* c) { return c ? c->local_frame : 0; }

extern "C" int rt_read_sum_f32(rt_ctx* c, float* rgba, size_t n_floats) {
    if (!c) return RT_ERR_INVALID;
    if (!rgba || n_floats < size_t(c->npix) * 4) return fail(c, RT_ERR_INVALID, "rt_read_sum_f32: buffer too small");
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    HIPCHK(c, hipMemcpyAsync(rgba, c->sum.p, size_t(c->npix) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_read_accum_f32(rt_ctx* c, float* rgba, size_t n_floats) {
    int rc = rt_read_sum_f32(c, rgba, n_floats);
    if (rc != RT_OK) return rc;
    if (c->flags & RT_REF_RGBA8_ACCUM) {  // the mean lives in the 8-bit image in legacy mode
        std::vector<uint32_t> a8(c->npix);
        HIPCHK(c, hipMemcpy(a8.data(), c->accum8.p, size_t(c->npix) * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < c->npix; ++i)
            for (int k = 0; k < 4; ++k) rgba[4 * size_t(i) + k] = float((a8[i] >> (8 * k)) & 255u) / 255.0f;
        return RT_OK;
    }
    for (uint32_t i = 0; i < c->npix; ++i) {
        float* p = rgba + 4 * size_t(i);
        float n = p[3];
        float inv = n > 0.0f ? 1.0f / n : 0.0f;
        p[0] *= inv; p[1] *= inv; p[2] *= inv; p[3] = n > 0.0f ? 1.0f : 0.0f;
    }
    return RT_OK;
}

extern "C" int rt_add_sum_f32(rt_ctx* c, const float* rgba_sum, size_t n_floats) {
    if (!c) return RT_ERR_INVALID;
    if (!rgba_sum || n_floats < size_t(c->npix) * 4) return fail(c, RT_ERR_INVALID, "rt_add_sum_f32: buffer too small");
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    HIPCHK(c, c->scratch.ensure(size_t(c->npix) * 16));
    HIPCHK(c, hipMemcpyAsync(c->scratch.p, rgba_sum, size_t(c->npix) * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_add_sum, dim3(grid_for(c, c->npix, 8)), dim3(kBlock), 0, c->stream, c->sum.as<float4>(), c->scratch.as<float4>(), c->npix);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_accum_device_buffer(rt_ctx* c, void** device_ptr, size_t* n_bytes) {
    if (!c) return RT_ERR_INVALID;
    if (!device_ptr || !n_bytes) return fail(c, RT_ERR_INVALID, "rt_accum_device_buffer: null argument");
    *device_ptr = c->sum.p;
    *n_bytes = size_t(c->npix) * 16;
    return RT_OK;
}

extern "C" int rt_read_framebuffer_rgba8(rt_ctx* c, uint8_t* rgba, size_t n_bytes) {
    if (!c) return RT_ERR_INVALID;
    if (!rgba || n_bytes < size_t(c->npix) * 4) return fail(c, RT_ERR_INVALID, "rt_read_framebuffer_rgba8: buffer too small");
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    if (!(c->flags & RT_REF_RGBA8_ACCUM)) {
        hipLaunchKernelGGL(k_tonemap, dim3(grid_for(c, c->npix, 8)), dim3(kBlock), 0, c->stream, c->sum.as<float4>(), c->fb.as<uint32_t>(), c->npix);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(rgba, c->fb.p, size_t(c->npix) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_last_timings(const rt_ctx* cc, rt_timings* out) {
    rt_ctx* c = const_cast<rt_ctx*>(cc);
    if (!c) return RT_ERR_INVALID;
    if (!out) return fail(c, RT_ERR_INVALID, "rt_last_timings: null output");
    if (!(c->flags & RT_KERNEL_TIMING)) return fail(c, RT_ERR_INVALID, "rt_last_timings: context created without RT_KERNEL_TIMING");
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    if ((rc = resolve_timings(c)) != RT_OK) return rc;
    *out = c->last;
    return RT_OK;
}

extern "C" int rt_get_counters(rt_ctx* c, rt_counters* out) {
    if (!c) return RT_ERR_INVALID;
    if (!out) return fail(c, RT_ERR_INVALID, "rt_get_counters: null output");
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    TraceStats s;
    HIPCHK(c, hipMemcpyAsync(&s, c->stats.p, sizeof(s), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->camera_rays = s.camera_rays; out->extend_rays = s.extend_rays; out->shadow_rays = s.shadow_rays;
    out->extend_node_pops = s.extend_node_pops; out->extend_tri_tests = s.extend_tri_tests; out->extend_leaf_pops = s.extend_leaf_pops;
    out->shadow_node_pops = s.shadow_node_pops; out->shadow_tri_tests = s.shadow_tri_tests; out->shadow_leaf_pops = s.shadow_leaf_pops;
    out->tail_paths = s.tail_paths;
    return RT_OK;
}

extern "C" int rt_reset_counters(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    int rc;
    if ((rc = bind_device_keep_graphs(c)) != RT_OK) return rc;
    HIPCHK(c, hipMemsetAsync(c->stats.p, 0, sizeof(TraceStats), c->stream));
    return RT_OK;
}

// ---- stage-level entry points ---------------------------------------------------------------------------

// True when every origin (n x 3 floats) lies where the compressed tree's boxes are conservative; other rays are traced
// on the exact path.
static bool origins_within_limit(const rt_ctx* c, uint32_t n, const float* origins) {
    const float lim = c->sd.origin_limit;
    for (size_t i = 0; i < size_t(n) * 3; ++i)
        if (!(std::fabs(origins[i]) <= lim)) return false;
    return true;
}

static int trace_common(rt_ctx* c, uint32_t n, const float* origins, const float* directions, const float* tmax, int shadow,
                        std::vector<float>& host_hits) {
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    host_hits.assign(size_t(n) * 4, 0.0f);
    if (n == 0) return RT_OK;
    DevBuf ro, rd, hits;
    if ((rc = upload_vec3(c, ro, n, origins, tmax)) != RT_OK) return rc;
    if ((rc = upload_vec3(c, rd, n, directions, nullptr)) != RT_OK) return rc;
    HIPCHK(c, hits.ensure(size_t(n) * 16));
    const bool count = (c->flags & RT_COUNT_TRAVERSAL) != 0;
    begin_call(c);
    HIPCHK(c, hipMemsetAsync(c->aux_cursor(), 0, kCursorSetWords * 4, c->stream));
    {
        StageTimer tm(c, shadow ? ST_CONNECT : ST_EXTEND);
        const bool fast = c->fast && origins_within_limit(c, n, origins);
        auto kern = count ? (fast ? k_trace_rays<true, true> : k_trace_rays<true, false>) : (fast ? k_trace_rays<false, true> : k_trace_rays<false, false>);
        hipLaunchKernelGGL(kern, dim3(trace_grid_for(c, n, c->trace_blocks_per_cu)), dim3(kTraceBlock), 0, c->stream, c->sd, ro.as<float4>(), rd.as<float4>(), n, shadow, hits.as<float4>(), RT_CURSOR(c->aux_cursor()), c->tstats());
    }
    end_call(c);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(host_hits.data(), hits.p, size_t(n) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_trace_closest(rt_ctx* c, uint32_t n, const float* origins, const float* directions, rt_hit* hits) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!origins || !directions || !hits)) return fail(c, RT_ERR_INVALID, "rt_trace_closest: null argument");
    std::vector<float> h;
    int rc = trace_common(c, n, origins, directions, nullptr, 0, h);
    if (rc != RT_OK) return rc;
    std::memcpy(hits, h.data(), size_t(n) * sizeof(rt_hit));
    return RT_OK;
}

extern "C" int rt_trace_closest_static(rt_ctx* c, uint32_t n, const float* origins, const float* directions, rt_hit* hits) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!origins || !directions || !hits)) return fail(c, RT_ERR_INVALID, "rt_trace_closest_static: null argument");
    if (n > RT_STATIC_MAX_RAYS)
        return fail(c, RT_ERR_UNSUPPORTED, "rt_trace_closest_static: " + std::to_string(n) + " rays exceed RT_STATIC_MAX_RAYS (" +
                                               std::to_string(RT_STATIC_MAX_RAYS) + "); split the call or use rt_trace_closest");
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    if (n == 0) return RT_OK;
    // no hipMalloc / hipFree and no allocating container from here on: everything lives in this device's `scratch`
    unsigned char* base = c->static_scratch;
    uint32_t* cursor = reinterpret_cast<uint32_t*>(base);
    float* d_o = reinterpret_cast<float*>(base + kStaticCursorBytes);
    float* d_d = reinterpret_cast<float*>(base + kStaticCursorBytes + size_t(RT_STATIC_MAX_RAYS) * 12);
    float4* d_h = reinterpret_cast<float4*>(base + kStaticCursorBytes + size_t(RT_STATIC_MAX_RAYS) * 24);
    HIPCHK(c, hipMemsetAsync(cursor, 0, kStaticCursorBytes, c->stream));      // the array keeps the previous call's contents
    HIPCHK(c, hipMemcpyAsync(d_o, origins, size_t(n) * 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_d, directions, size_t(n) * 12, hipMemcpyHostToDevice, c->stream));
    const uint32_t grid = trace_grid_for(c, n, c->trace_blocks_per_cu);
    const bool count = (c->flags & RT_COUNT_TRAVERSAL) != 0;
    const bool fast = c->fast && origins_within_limit(c, n, origins);
    auto kern = count ? (fast ? k_trace_packed<true, true> : k_trace_packed<true, false>) : (fast ? k_trace_packed<false, true> : k_trace_packed<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTraceBlock), 0, c->stream, c->sd, d_o, d_d, n, d_h, RT_CURSOR(cursor), c->tstats());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hits, d_h, size_t(n) * sizeof(rt_hit), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_trace_shadow(rt_ctx* c, uint32_t n, const float* origins, const float* directions, const float* tmax,
                               uint8_t* visible) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!origins || !directions || !tmax || !visible)) return fail(c, RT_ERR_INVALID, "rt_trace_shadow: null argument");
    std::vector<float> h;
    int rc = trace_common(c, n, origins, directions, tmax, 1, h);
    if (rc != RT_OK) return rc;
    for (uint32_t i = 0; i < n; ++i) visible[i] = h[4 * size_t(i)] != 0.0f ? 1 : 0;
    return RT_OK;
}

extern "C" int rt_bench_trace(rt_ctx* c, uint32_t n, const float* origins, const float* directions, int shadow, uint32_t repeat,
                              float* ms_per_launch) {
    if (!c) return RT_ERR_INVALID;
    if (!n || !origins || !directions || !ms_per_launch || !repeat) return fail(c, RT_ERR_INVALID, "rt_bench_trace: bad argument");
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    DevBuf ro, rd, hits;
    std::vector<float> far_t(n, INFINITY);
    if ((rc = upload_vec3(c, ro, n, origins, shadow ? far_t.data() : nullptr)) != RT_OK) return rc;
    if ((rc = upload_vec3(c, rd, n, directions, nullptr)) != RT_OK) return rc;
    HIPCHK(c, hits.ensure(size_t(n) * 16));
    hipEvent_t a, b;
    HIPCHK(c, hipEventCreate(&a));
    HIPCHK(c, hipEventCreate(&b));
    const bool fast = c->fast && origins_within_limit(c, n, origins);
    auto kern = fast ? k_trace_rays<false, true> : k_trace_rays<false, false>;
    auto launch = [&] {
        (void)hipMemsetAsync(c->aux_cursor(), 0, kCursorSetWords * 4, c->stream);
        hipLaunchKernelGGL(kern, dim3(trace_grid_for(c, n, c->trace_blocks_per_cu)), dim3(kTraceBlock), 0, c->stream, c->sd, ro.as<float4>(), rd.as<float4>(), n, shadow, hits.as<float4>(), RT_CURSOR(c->aux_cursor()), c->tstats());
    };
    launch();  // warm-up
    (void)hipEventRecord(a, c->stream);
    for (uint32_t r = 0; r < repeat; ++r) launch();
    (void)hipEventRecord(b, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIPCHK(c, e);
    HIPCHK(c, hipGetLastError());
    *ms_per_launch = ms / float(repeat);
    return RT_OK;
}

static int download_vec3(rt_ctx* c, const float4* dev, uint32_t n, float* out3) {
    std::vector<float> tmp(size_t(n) * 4);
    HIPCHK(c, hipMemcpyAsync(tmp.data(), dev, tmp.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; ++i) { out3[3 * size_t(i)] = tmp[4 * size_t(i)]; out3[3 * size_t(i) + 1] = tmp[4 * size_t(i) + 1]; out3[3 * size_t(i) + 2] = tmp[4 * size_t(i) + 2]; }
    return RT_OK;
}

extern "C" int rt_generate_rays(rt_ctx* c, uint32_t frame, float* origins, float* directions, uint32_t* seeds) {
    if (!c) return RT_ERR_INVALID;
    if (!origins || !directions) return fail(c, RT_ERR_INVALID, "rt_generate_rays: null argument");
    if (!c->have_camera) return fail(c, RT_ERR_NO_SCENE, "no camera set");
    if (!c->paths_ok) return fail(c, RT_ERR_OOM, "no path buffers: the last resize ran out of device memory");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    hipLaunchKernelGGL(k_raygen, dim3(grid_for(c, c->npix, 8)), dim3(kBlock), 0, c->stream, c->cam, c->lanes[0].state(0), c->lanes[0].rad.as<float4>(),
                       c->npix, c->npix, frame, 1u, c->seeds.as<uint32_t>(), TileSlice{0u, 1u, (c->W + 7u) / 8u, c->W, c->H});
    HIPCHK(c, hipGetLastError());
    if ((rc = download_vec3(c, c->lanes[0].st_o[0].as<float4>(), c->npix, origins)) != RT_OK) return rc;
    if ((rc = download_vec3(c, c->lanes[0].st_d[0].as<float4>(), c->npix, directions)) != RT_OK) return rc;
    if (seeds) HIPCHK(c, hipMemcpy(seeds, c->seeds.p, size_t(c->npix) * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_render_radiance(rt_ctx* c, uint32_t frame, float* rgb) {
    if (!c) return RT_ERR_INVALID;
    if (!rgb) return fail(c, RT_ERR_INVALID, "rt_render_radiance: null output");
    int rc;
    if ((rc = ready_to_render(c)) != RT_OK) return rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    begin_call(c);
    const uint32_t tile_first = c->tile_first, tile_stride = c->tile_stride;      // this probe renders the whole frame
    c->tile_first = 0; c->tile_stride = 1;
    rc = run_pass(c, c->lanes[0], c->stream, 1, frame, 1, nullptr);
    c->tile_first = tile_first; c->tile_stride = tile_stride;
    if (rc != RT_OK) return rc;
    if ((rc = finish_pass(c, c->lanes[0], c->stream, c->npix)) != RT_OK) return rc;
    end_call(c);
    return download_vec3(c, c->lanes[0].rad.as<float4>(), c->npix, rgb);
}

extern "C" int rt_shade_vertices(rt_ctx* c, uint32_t n, uint32_t bounce, const rt_shade_vertex* vin, rt_shade_result* vout) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!vin || !vout)) return fail(c, RT_ERR_INVALID, "rt_shade_vertices: null argument");
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    if (!c->paths_ok) return fail(c, RT_ERR_OOM, "no path buffers: the last resize ran out of device memory");
    if (bounce >= c->max_bounces) return fail(c, RT_ERR_INVALID, "rt_shade_vertices: bounce must be below max_bounces");
    if (n > c->slot_cap) return fail(c, RT_ERR_UNSUPPORTED, "rt_shade_vertices: more vertices than the context has path slots (width x height x frames per pass)");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    if (n == 0) return RT_OK;
    // The vertices go into lane 0's path-state buffers at queue positions 0 .. n-1 (slot i = vertex i, radiance zero, so
    // what comes back is what this vertex added). The main stream is ordered behind every pass queued so far.
    Lane& l = c->lanes[0];
    std::vector<float4> ro(n), rd(n), hit(n), thr(n);
    for (uint32_t i = 0; i < n; ++i) {
        const rt_shade_vertex& v = vin[i];
        uint32_t prim = v.hit.prim;
        if (prim != kNoPrim && prim >= c->sd.n_tri) return fail(c, RT_ERR_INVALID, "rt_shade_vertices: vertex " + std::to_string(i) + " names a triangle the scene does not have");
        ro[i] = make_float4(v.ray_o[0], v.ray_o[1], v.ray_o[2], v.prev_pdf);
        rd[i] = make_float4(v.ray_d[0], v.ray_d[1], v.ray_d[2], 0.0f);
        std::memcpy(&rd[i].w, &v.seed, 4);
        thr[i] = make_float4(v.throughput[0], v.throughput[1], v.throughput[2], 0.0f);
        std::memcpy(&thr[i].w, &i, 4);
        hit[i] = make_float4(v.hit.t, v.hit.u, v.hit.v, 0.0f);
        std::memcpy(&hit[i].w, &prim, 4);
    }
    const size_t bytes = size_t(n) * sizeof(float4);
    HIPCHK(c, c->probe.ensure(3 * bytes));
    HIPCHK(c, hipMemcpyAsync(l.st_o[0].p, ro.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(l.st_d[0].p, rd.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(l.st_t[0].p, thr.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(l.hit.p, hit.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(l.counts.p, 0, Lane::counts_words() * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->probe.p, 0, 3 * bytes, c->stream));
    for (DevBuf* b : {&l.rad, &l.st_o[1], &l.st_d[1], &l.st_t[1], &l.sq_o, &l.sq_d, &l.sq_c}) HIPCHK(c, hipMemsetAsync(b->p, 0, bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));     // the host vectors above are pageable: do not let them go before the copies ran
    ShadeParams sp{bounce, c->max_bounces, c->flags, Regions{nullptr, c->shard_cap, n}, l.hit.as<float4>(), l.state(1), l.shard_counts(bounce),
                   l.sq(), l.shadow_counts(bounce), c->shard_cap, l.rad.as<float4>(), c->probe.as<uint4>()};
    const uint32_t grid = grid_for(c, n, c->shade_blocks_per_cu);
    const bool lds_tables = c->sd.n_mat <= kLdsMats && c->sd.n_emissive <= kLdsLights;
    if (!c->exact_arith) rtint::launch_shade_fast(lds_tables, true, grid, c->stream, c->sd, l.state(0), sp);
    else if (lds_tables) hipLaunchKernelGGL((k_shade<true, true>), dim3(grid), dim3(kBlock), 0, c->stream, c->sd, l.state(0), sp);
    else hipLaunchKernelGGL((k_shade<false, true>), dim3(grid), dim3(kBlock), 0, c->stream, c->sd, l.state(0), sp);
    HIPCHK(c, hipGetLastError());
    std::vector<float4> rad(n), so(n), sdir(n), scon(n);
    std::vector<uint4> pr(3 * size_t(n));
    HIPCHK(c, hipMemcpyAsync(ro.data(), l.st_o[1].p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rd.data(), l.st_d[1].p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(thr.data(), l.st_t[1].p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(rad.data(), l.rad.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(so.data(), l.sq_o.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(sdir.data(), l.sq_d.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(scon.data(), l.sq_c.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(pr.data(), c->probe.p, 3 * bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; ++i) {
        rt_shade_result& o = vout[i];
        std::memset(&o, 0, sizeof(o));
        const uint4 p0 = pr[3 * size_t(i)], p1 = pr[3 * size_t(i) + 1], p2 = pr[3 * size_t(i) + 2];
        o.emitted[0] = rad[i].x; o.emitted[1] = rad[i].y; o.emitted[2] = rad[i].z;
        std::memcpy(&o.seed, &rd[i].w, 4);
        o.throughput[0] = thr[i].x; o.throughput[1] = thr[i].y; o.throughput[2] = thr[i].z; o.pdf = ro[i].w;
        o.alive = p0.x & 1u;
        o.has_shadow = (p0.x >> 1) & 1u;
        if (o.alive) {
            o.next_o[0] = ro[i].x; o.next_o[1] = ro[i].y; o.next_o[2] = ro[i].z;
            o.next_d[0] = rd[i].x; o.next_d[1] = rd[i].y; o.next_d[2] = rd[i].z;
        }
        if (o.has_shadow) {
            o.shadow_o[0] = so[i].x; o.shadow_o[1] = so[i].y; o.shadow_o[2] = so[i].z; o.shadow_tmax = so[i].w;
            o.shadow_d[0] = sdir[i].x; o.shadow_d[1] = sdir[i].y; o.shadow_d[2] = sdir[i].z;
            o.shadow_contrib[0] = scon[i].x; o.shadow_contrib[1] = scon[i].y; o.shadow_contrib[2] = scon[i].z;
        }
        o.light_index = int32_t(p0.y); o.lobe = int32_t(p0.z); o.light_tri = int32_t(p0.w);
        std::memcpy(&o.light_pdf, &p1.x, 4);
        std::memcpy(&o.bsdf_pdf_light, &p1.y, 4);
        std::memcpy(&o.roughness, &p1.z, 4);
        std::memcpy(&o.medium_throughput[0], &p2.x, 4); std::memcpy(&o.medium_throughput[1], &p2.y, 4); std::memcpy(&o.medium_throughput[2], &p2.z, 4);
    }
    return RT_OK;
}

static int bsdf_common(rt_ctx* c, uint32_t n, const rt_material* mats, const float* eta, const float* V, const float* N,
                       const float* L, const float* r, float* out_L, float* out_f, float* out_pdf) {
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    if (n == 0) return RT_OK;
    DevBuf dm, de, dv, dn, dl, dr, oL, of, op;
    auto put = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(bytes);
        if (e == hipSuccess && src) e = hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    HIPCHK(c, put(dm, mats, size_t(n) * sizeof(rt_material)));
    HIPCHK(c, put(de, eta, size_t(n) * 4));
    HIPCHK(c, put(dv, V, size_t(n) * 12));
    HIPCHK(c, put(dn, N, size_t(n) * 12));
    HIPCHK(c, put(dl, L, size_t(n) * 12));
    HIPCHK(c, put(dr, r, size_t(n) * 8));
    HIPCHK(c, put(oL, nullptr, size_t(n) * 12));
    HIPCHK(c, put(of, nullptr, size_t(n) * 12));
    HIPCHK(c, put(op, nullptr, size_t(n) * 4));
    BsdfIo io{dm.as<float4>(), de.as<float>(), dv.as<float>(), dn.as<float>(), dl.as<float>(), dr.as<float>(), oL.as<float>(), of.as<float>(), op.as<float>()};
    if (!c->exact_arith) rtint::launch_bsdf_fast((n + 63) / 64, 64, c->stream, io, n, r ? 1 : 0);
    else hipLaunchKernelGGL(k_bsdf, dim3((n + 63) / 64), dim3(64), 0, c->stream, io, n, r ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (out_L) HIPCHK(c, hipMemcpy(out_L, oL.p, size_t(n) * 12, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_f, of.p, size_t(n) * 12, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(out_pdf, op.p, size_t(n) * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_eval_bsdf(rt_ctx* c, uint32_t n, const rt_material* mats, const float* eta, const float* V, const float* N,
                            const float* L, float* out_f, float* out_pdf) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!mats || !eta || !V || !N || !L || !out_f || !out_pdf)) return fail(c, RT_ERR_INVALID, "rt_eval_bsdf: null argument");
    return bsdf_common(c, n, mats, eta, V, N, L, nullptr, nullptr, out_f, out_pdf);
}

extern "C" int rt_sample_bsdf(rt_ctx* c, uint32_t n, const rt_material* mats, const float* eta, const float* V, const float* N,
                              const float* r1r2, float* out_L, float* out_f, float* out_pdf) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!mats || !eta || !V || !N || !r1r2 || !out_L || !out_f || !out_pdf)) return fail(c, RT_ERR_INVALID, "rt_sample_bsdf: null argument");
    return bsdf_common(c, n, mats, eta, V, N, V, r1r2, out_L, out_f, out_pdf);
}

extern "C" int rt_sample_texture(rt_ctx* c, uint32_t tex, uint32_t n, const float* uvw, float* rgba) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!uvw || !rgba)) return fail(c, RT_ERR_INVALID, "rt_sample_texture: null argument");
    if (tex != 0xFFFFFFFFu && tex >= c->sd.tex.n_tex) return fail(c, RT_ERR_INVALID, "rt_sample_texture: texture index out of range");
    int rc;
    if ((rc = bind_device(c)) != RT_OK) return rc;
    if (n == 0) return RT_OK;
    DevBuf in, out;
    HIPCHK(c, in.ensure(size_t(n) * 12));
    HIPCHK(c, out.ensure(size_t(n) * 16));
    HIPCHK(c, hipMemcpy(in.p, uvw, size_t(n) * 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sample_texture, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->sd.tex, tex, n, in.as<float>(), out.as<float4>());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(rgba, out.p, size_t(n) * 16, hipMemcpyDeviceToHost));
    return RT_OK;
}
