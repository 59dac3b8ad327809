// Types and wavefront primitives shared by every kernel of the render core: scene and queue views, the append
// primitive, the map from a queue's item number to where the item lives.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_api.h"
#include "rt_device_math.h"

namespace rtdev {

constexpr uint32_t kBlock = 256;          // 4 wavefronts per workgroup
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 1024
#endif
constexpr uint32_t kTraceBlock = RT_TRACE_BLOCK;   // workgroup size of the traversal kernels
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 8
#endif
#ifndef RT_REFILL_AT
#define RT_REFILL_AT 8
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 1
#endif
#ifndef RT_CHUNK
#define RT_CHUNK 256
#endif
#ifndef RT_TAIL_EIGHTHS
#define RT_TAIL_EIGHTHS 4
#endif
#ifndef RT_TAIL_CHUNK
#define RT_TAIL_CHUNK 128
#endif
#ifndef RT_TAIL_EIGHTHS_ANYHIT
#define RT_TAIL_EIGHTHS_ANYHIT RT_TAIL_EIGHTHS
#endif
#ifndef RT_TAIL_CHUNK_ANYHIT
#define RT_TAIL_CHUNK_ANYHIT RT_TAIL_CHUNK
#endif
#ifndef RT_WAVE_INTERLEAVE
#define RT_WAVE_INTERLEAVE 0
#endif
#ifndef RT_STATIC_FLOOR
#define RT_STATIC_FLOOR 1
#endif
#ifndef RT_REFILL_AT_ANYHIT
#define RT_REFILL_AT_ANYHIT RT_REFILL_AT
#endif
#ifndef RT_LEAF_BATCH_ANYHIT
#define RT_LEAF_BATCH_ANYHIT 32
#endif
#ifndef RT_TAIL_CURSORS
#define RT_TAIL_CURSORS 1
#endif
#ifndef RT_SPECULATE
#define RT_SPECULATE 0
#endif
#ifndef RT_COOP_ANYHIT
#define RT_COOP_ANYHIT 0
#endif
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 16
#endif
#ifndef RT_HOT_LDS
#define RT_HOT_LDS 746
#endif
#ifndef RT_TRACE_WAVES_PER_EU
#define RT_TRACE_WAVES_PER_EU 8
#endif
constexpr uint32_t kStackLds = RT_STACK_LDS;  // traversal stack entries kept in LDS per lane (4 x kTraceBlock bytes per entry per workgroup)
constexpr uint32_t kHotLds = RT_HOT_LDS;      // node records (64 B each) every traversal workgroup keeps in LDS
constexpr uint32_t kStackMax = 65;        // MAX_BVH_DEPTH + 1, main.comp:6 / intersects.comp:130
#ifndef RT_STACK_MAX_WIDE
#define RT_STACK_MAX_WIDE (3 * 64 + 1)
#endif
constexpr uint32_t kStackMaxWide = RT_STACK_MAX_WIDE;   // the 4-wide tree stacks up to three entries per level
constexpr uint32_t kHotWide = (RT_HOT_LDS * 64u / 112u) & ~3u;   // 4-wide records (112 B used) in the same LDS bytes
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kRefNone = 0x7FFFFFFFu;
constexpr uint32_t kNoPrim = 0xFFFFFFFFu;
constexpr uint32_t kMaxBounces = 64;

struct SceneDev {
    const float4* __restrict__ nodes;     // 4 x float4 per interior record
    const float4* __restrict__ isect;     // 3 x float4 per triangle
    const float4* __restrict__ shade;     // 4 x float4 per triangle: {n0, matword} {n1, n2.x} {n2.yz, uv0} {uv1, uv2}
    const float4* __restrict__ mats;      // 6 x float4 per material
    const float4* __restrict__ lights;    // 3 x float4 per analytic light
    const float4* __restrict__ light_tris; // per emissive triangle: {v0, area} {v1, Le.r} {v2, Le.g} {unit normal, Le.b}
    const float* __restrict__ cdf;        // un-normalised running power
    uint32_t root_ref, n_tri, n_light, n_emissive, n_mat;
    uint32_t n_hot;                       // records [0, n_hot) of `nodes` are the largest-area nodes, largest first
    const float4* __restrict__ wide;      // 8 x float4 per record of the 4-wide tree the occlusion queries walk
    uint32_t wide_root, n_hot_wide;
    float origin_limit;                   // the compressed tree's boxes are conservative for rays starting within this distance of the origin (per axis)
    const float4* __restrict__ cw;        // 4 x float4 per record of the compressed 4-wide tree (same topology and record order as `wide`)
    uint32_t has_glass;                   // any material with spec_trans > 0.5 (decides how shadow rays may terminate)
    uint32_t cdf_len_seen;                // length the light selector searches (true length unless the padded-length quirk is on)
    TexView tex;
};

constexpr uint32_t kQueueShards = 64;
// The append counters of a queue sit kCounterStride words apart, one per 256-byte block: returning atomics on words of
// one cache line are served one after another by the same memory channel, whichever word they name (64 counters packed
// into two lines cost the shade stage 0.75 of its 1.2 ms per 8-frame launch, measured by leaving the atomic out).
#ifndef RT_COUNTER_STRIDE
#define RT_COUNTER_STRIDE 64
#endif
constexpr uint32_t kCounterStride = RT_COUNTER_STRIDE;
constexpr uint32_t kTailCursors = RT_TAIL_CURSORS;   // work cursors per traversal launch, kCounterStride words apart
constexpr uint32_t kCursorSetWords = kTailCursors * kCounterStride;

// State of the paths of one bounce, three 16-byte records per path, indexed by queue position.
struct PathDev {
    float4* __restrict__ o;   // {ray origin, pdf of the BSDF sample that produced the ray}
    float4* __restrict__ d;   // {ray direction, RNG state bits}
    float4* __restrict__ t;   // {throughput rgb, slot bits}
};

// Light samples of one bounce, indexed by their own queue position.
struct ShadowDev {
    float4* __restrict__ o;   // {origin, tmax}
    float4* __restrict__ d;   // {direction, slot bits}
    float4* __restrict__ c;   // {contribution if unoccluded, -}
};

// Where the items of a queue live: kQueueShards regions of `cap` records each, region k filled from its start with
// counts[k] records; item number v (0 <= v < sum of counts) is the (v - prefix[k])-th record of the region k it falls
// into. counts == nullptr describes the queue of bounce 0: n_imm items at positions 0 .. n_imm-1.
struct Regions {
    const uint32_t* __restrict__ counts;
    uint32_t cap;
    uint32_t n_imm;
};

struct TraceStats {               // all 64-bit, updated with one atomic per wavefront
    unsigned long long extend_rays, shadow_rays, camera_rays;
    unsigned long long extend_node_pops, extend_leaf_pops, extend_tri_tests;
    unsigned long long shadow_node_pops, shadow_leaf_pops, shadow_tri_tests;
    unsigned long long tail_paths;   // paths taken over by the per-path kernel
    // Per-record visit counters (null outside a tuning pass): [0] the binary tree's records, [1] the 4-wide trees' (the
    // compressed tree shares the uncompressed one's record order). Only the counting kernel variants look at them, and
    // only for every 16th work item.
    uint32_t* visits[2];
};
constexpr uint32_t kVisitSampleMask = 15u;

// ---- wavefront primitives (64 lanes) -------------------------------------------------------------------
RT_DEV uint32_t lane_id() { return __lane_id(); }
// Number of set bits of a wavefront mask below the calling lane (v_mbcnt: no per-lane 64-bit mask to keep in registers).
RT_DEV uint32_t bits_below_lane(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// Every lane of the wavefront must call this (predicate false for lanes without work).
// Returns the destination index for lanes with pred == true.
RT_DEV uint32_t wave_append(bool pred, uint32_t* counter) {
    unsigned long long mask = __ballot(pred);
    uint32_t n = uint32_t(__popcll(mask));
    uint32_t base = 0;
    if (n != 0 && lane_id() == 0) base = atomicAdd(counter, n);
    base = __shfl(base, 0);
    return base + bits_below_lane(mask);
}
RT_DEV uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;  // valid in lane 0
}

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Exclusive prefix of the region counts into s_prefix[0..64] (s_prefix[64] = number of items), by the first wavefront
// of the workgroup; ends with a barrier. Returns the number of items.
RT_DEV uint32_t load_region_prefix(const Regions& rg, uint32_t* s_prefix) {
    if (threadIdx.x < 64) {
        uint32_t c = rg.counts ? rg.counts[threadIdx.x * kCounterStride] : (threadIdx.x == 0 ? rg.n_imm : 0u);
        uint32_t incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t up = __shfl_up(incl, off);
            if (int(threadIdx.x) >= off) incl += up;
        }
        s_prefix[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) s_prefix[0] = 0u;
    }
    __syncthreads();
    return s_prefix[kQueueShards];
}

// Position of item v: which region it falls into (binary search over the 64 prefixes in LDS), then its offset there.
struct RegionMap {
    const lds_u32* prefix;
    uint32_t cap;
    bool identity;
    RT_DEV uint32_t operator()(uint32_t v) const {
        if (identity) return v;
        uint32_t k = 0;
        for (uint32_t step = kQueueShards >> 1; step != 0; step >>= 1)
            if (prefix[k + step] <= v) k += step;
        return k * cap + (v - prefix[k]);
    }
    // The run of items around item v that lies in one region: items [lo, hi) live at positions base + (item - lo). Meant
    // for a wave-uniform v: a wave that takes consecutive items looks its region up once and then places each item with
    // one subtraction and one addition instead of a six-step search through LDS per item.
    RT_DEV void run_of(uint32_t v, uint32_t& lo, uint32_t& hi, uint32_t& base) const {
        if (identity) { lo = 0u; hi = 0xFFFFFFFFu; base = 0u; return; }
        uint32_t k = 0;
        for (uint32_t step = kQueueShards >> 1; step != 0; step >>= 1)
            if (prefix[k + step] <= v) k += step;
        lo = prefix[k]; hi = prefix[k + 1]; base = k * cap;
    }
};

}  // namespace rtdev
