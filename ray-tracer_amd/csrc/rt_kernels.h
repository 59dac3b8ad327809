// Wavefront path-tracing kernels for gfx950 (MI355X): raygen, extend, shade, connect, accumulate.
//
// One work item = one path segment. A path owns a fixed "slot" (slot = frame_in_pass * n_pixels + pixel) only for its
// radiance sum; everything else it carries (ray, throughput, pdf, RNG state, its slot number) travels with it in QUEUE
// ORDER: shade reads the state of the paths of this bounce as dense records and writes the survivors' state as dense
// records for the next bounce, at a position obtained with a wave-level ballot/popcount prefix and ONE atomic per
// wavefront. The appends go to kQueueShards regions with a counter each (a single counter would serialise ~10^5 returning
// atomics per sample); the next kernel walks the regions one after another, so no compaction pass is needed. Queue
// lengths stay on the device: every kernel is a fixed-size grid of persistent wavefronts that reads its work counts
// from HBM, so a whole multi-bounce sample is queued without a host round trip.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_api.h"
#include "rt_device_math.h"
#include "rt_kernels_common.h"
#include "rt_shade.h"
#include "rt_traverse.h"

namespace rtdev {

// ---- traversal -----------------------------------------------------------------------------------------
// Per ray, visit order and tie rules are exactly those of hit_meshes (intersects.comp:124-175): the nearer child
// first, on equal entry distance the right child first, a child is entered only while its entry distance is
// strictly below the current hit distance, triangles of a leaf in storage order with a strict "closer" test, so the
// first triangle tested wins an exact tie. Only the scheduling across the 64 lanes differs from a per-thread loop:
//
//   * persistent wavefronts own chunks of consecutive work items (the tail of each list is handed out through a
//     device-side cursor) and refill lanes whose ray has finished, so a wave is not held hostage by its slowest ray;
//   * lanes walk interior nodes until enough of them are parked on a leaf (or none has a node left), then the parked
//     leaves are intersected together, closest-hit queries with the whole wavefront sharing the triangles. Box tests
//     and triangle tests are therefore each executed with most lanes active instead of being serialised inside every
//     iteration;
//   * the records of the largest nodes are read from a copy in LDS.
// Occlusion queries (any hit below a fixed distance) do not depend on visiting order and walk a 4-wide tree instead.
//
// The traversal stack keeps its first kStackLds - 1 entries per lane in LDS (entry-major, so lanes never conflict on a
// bank) and the rest, up to the reference's bound of 65, in per-lane scratch.

// The busiest node records live in LDS, one copy per workgroup, stored chunk-major (all first 16-byte chunks, then all
// second ones, ...): the 16 lanes a ds_read_b128 serves per cycle then spread over 16 bank groups by record index
// instead of piling onto the 4 a record-major layout would give them.
struct HotNodes {
    const lds_v4f* tab;     // kHotLds x 4 chunks
    uint32_t n;             // records [0, n) are resident
};
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2u lds_v2u;

RT_DEV void load_hot_nodes(const float4* __restrict__ recs, v4f* s_hot, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n * 4u; i += kTraceBlock) {
        float4 v = recs[i];
        s_hot[(i & 3u) * kHotLds + (i >> 2)] = v4f{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
}

// Fetches the 56 used bytes of node record `cur`: resident records from LDS, the others from global memory. Both kinds
// are requested before either is waited for. Written as one asm block because the two sources fill the same registers
// for disjoint sets of lanes: the compiler cannot know the sets are disjoint and would wait for the global data before
// issuing the LDS reads (adding the two latencies); inside the block the lane mask is split by hand and restored.
// `lds_a` / `lds_b` are the LDS byte addresses of chunk 0 and chunk 2 of the record (meaningful for resident records).
RT_DEV void fetch_node(const float4* __restrict__ nodes, uint32_t cur, uint32_t n_resident, uint32_t lds_a, uint32_t lds_b,
                       v4f& q0, v4f& q1, v4f& q2, v2u& refs) {
    const uint32_t goff = cur << 6;               // byte offset of the record: the node array stays below 4 GiB (2^26 records)
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_gt_u32 vcc, %[nres], %[cur]\n\t"
        "s_and_b64 exec, %[sv], vcc\n\t"
        "ds_read_b128 %[q0], %[la]\n\t"
        "ds_read_b128 %[q1], %[la] offset:%[c1]\n\t"
        "ds_read_b128 %[q2], %[lb]\n\t"
        "ds_read_b64 %[rf], %[lb] offset:%[c1]\n\t"
        "s_andn2_b64 exec, %[sv], vcc\n\t"
        "global_load_dwordx4 %[q0], %[go], %[base]\n\t"
        "global_load_dwordx4 %[q1], %[go], %[base] offset:16\n\t"
        "global_load_dwordx4 %[q2], %[go], %[base] offset:32\n\t"
        "global_load_dwordx2 %[rf], %[go], %[base] offset:48\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [rf] "=&v"(refs), [sv] "=&s"(saved)
        : [nres] "s"(n_resident), [cur] "v"(cur), [la] "v"(lds_a), [lb] "v"(lds_b), [go] "v"(goff), [base] "s"(nodes),
          [c1] "n"(kHotLds * 16)
        : "vcc", "memory");
}

// One interior-node visit: slab test of both children (intersects.comp:103-121), ordered descent.
RT_DEV void node_step(const SceneDev& sc, LaneRay& r, lds_u32* stack, uint32_t* spill, const HotNodes& hot) {
    float4 q0, q1, q2;
    uint32_t ref_l, ref_r;
    if (kHotLds != 0) {
        static_assert(kHotLds * 16 < 65536, "chunk stride must fit the 16-bit offset of an LDS instruction");
        v4f a, b, c;
        v2u refs;
        const uint32_t lds_a = uint32_t(uintptr_t(hot.tab + r.cur));
        fetch_node(sc.nodes, r.cur, hot.n, lds_a, lds_a + 2u * kHotLds * 16u, a, b, c, refs);
        q0 = make_float4(a.x, a.y, a.z, a.w); q1 = make_float4(b.x, b.y, b.z, b.w); q2 = make_float4(c.x, c.y, c.z, c.w);
        ref_l = refs.x; ref_r = refs.y;
    } else {
        const float4* np = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.nodes) + (r.cur << 6));
        q0 = np[0]; q1 = np[1]; q2 = np[2];
        uint2 refs = *reinterpret_cast<const uint2*>(np + 3);
        ref_l = refs.x; ref_r = refs.y;
    }
    f3 l1 = (f3{q0.x, q0.y, q0.z} - r.o) * r.inv, l2 = (f3{q0.w, q1.x, q1.y} - r.o) * r.inv;
    f3 r1 = (f3{q1.z, q1.w, q2.x} - r.o) * r.inv, r2 = (f3{q2.y, q2.z, q2.w} - r.o) * r.inv;
    float l_lo = fmaxf(fmaxf(fminf(l1.x, l2.x), fminf(l1.y, l2.y)), fminf(l1.z, l2.z));
    float l_hi = fminf(fminf(fmaxf(l1.x, l2.x), fmaxf(l1.y, l2.y)), fmaxf(l1.z, l2.z));
    float r_lo = fmaxf(fmaxf(fminf(r1.x, r2.x), fminf(r1.y, r2.y)), fminf(r1.z, r2.z));
    float r_hi = fminf(fminf(fmaxf(r1.x, r2.x), fmaxf(r1.y, r2.y)), fmaxf(r1.z, r2.z));
    float dl = (l_lo <= l_hi && l_hi > 0.0f) ? l_lo : INFINITY;
    float dr = (r_lo <= r_hi && r_hi > 0.0f) ? r_lo : INFINITY;
    // intersects.comp:164-170. dl and dr are never NaN, so the child visited first has the smaller-or-equal entry distance
    // and "far child passes the strict test against the hit distance" implies "near child passes": the shader's four pushes
    // reduce to {enter near, stack far} / {enter near} / {nothing to enter: pop}.
    bool left_first = dl < dr;
    float d_near = left_first ? dl : dr, d_far = left_first ? dr : dl;
    uint32_t near_ref = left_first ? ref_l : ref_r, far_ref = left_first ? ref_r : ref_l;
    const bool go_near = d_near < r.t, push = d_far < r.t;
    // wave-uniform choice: unless some lane's stack has outgrown its LDS slots (rare), pushes and pops are single LDS operations
    const bool lds_only = __builtin_amdgcn_ballot_w64(r.sp + 1u >= kStackLds) == 0ull;
    if (__builtin_expect(lds_only, 1)) {
        if (push) stack_push(stack, spill, r, far_ref, true);
        if (go_near) r.cur = near_ref;
        else r.cur = stack_pop(stack, spill, r, true);
    } else {
        if (push) stack_push(stack, spill, r, far_ref, false);
        if (go_near) r.cur = near_ref;
        else r.cur = stack_pop(stack, spill, r, false);
    }
}

// ---- 4-wide tree (occlusion queries) -------------------------------------------------------------------
// Record: 24 floats {min.xyz, max.xyz} x 4 entries in chunks 0-5, the four references in chunk 6. In LDS the first
// kHotWide records are stored chunk-major like the binary ones.
RT_DEV void load_hot_wide(const SceneDev& sc, v4f* s_hot, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n * 8u; i += kTraceBlock) {
        const uint32_t chunk = i & 7u;
        if (chunk == 7u) continue;
        float4 v = sc.wide[i];
        s_hot[chunk * kHotWide + (i >> 3)] = v4f{v.x, v.y, v.z, v.w};
    }
    __syncthreads();
}

// Fetch of a 4-wide record, resident lanes from LDS and the others from global memory in one block (see fetch_node).
RT_DEV void fetch_wide(const float4* __restrict__ wide, uint32_t cur, uint32_t n_resident, uint32_t lds_a,
                       v4f& c0, v4f& c1, v4f& c2, v4f& c3, v4f& c4, v4f& c5, v4f& refs) {
    const uint32_t goff = cur << 7;               // byte offset of the record: below 4 GiB (2^25 records)
    const uint32_t lds_b = lds_a + 4u * kHotWide * 16u;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_gt_u32 vcc, %[nres], %[cur]\n\t"
        "s_and_b64 exec, %[sv], vcc\n\t"
        "ds_read_b128 %[c0], %[la]\n\t"
        "ds_read_b128 %[c1], %[la] offset:%[s1]\n\t"
        "ds_read_b128 %[c2], %[la] offset:%[s2]\n\t"
        "ds_read_b128 %[c3], %[la] offset:%[s3]\n\t"
        "ds_read_b128 %[c4], %[lb]\n\t"
        "ds_read_b128 %[c5], %[lb] offset:%[s1]\n\t"
        "ds_read_b128 %[rf], %[lb] offset:%[s2]\n\t"
        "s_andn2_b64 exec, %[sv], vcc\n\t"
        "global_load_dwordx4 %[c0], %[go], %[base]\n\t"
        "global_load_dwordx4 %[c1], %[go], %[base] offset:16\n\t"
        "global_load_dwordx4 %[c2], %[go], %[base] offset:32\n\t"
        "global_load_dwordx4 %[c3], %[go], %[base] offset:48\n\t"
        "global_load_dwordx4 %[c4], %[go], %[base] offset:64\n\t"
        "global_load_dwordx4 %[c5], %[go], %[base] offset:80\n\t"
        "global_load_dwordx4 %[rf], %[go], %[base] offset:96\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [c4] "=&v"(c4), [c5] "=&v"(c5), [rf] "=&v"(refs),
          [sv] "=&s"(saved)
        : [nres] "s"(n_resident), [cur] "v"(cur), [la] "v"(lds_a), [lb] "v"(lds_b), [go] "v"(goff), [base] "s"(wide),
          [s1] "n"(kHotWide * 16), [s2] "n"(kHotWide * 32), [s3] "n"(kHotWide * 48)
        : "vcc", "memory");
}

// Entry distance of one box, +inf unless the box is hit and entered below `tmax` (intersects.comp:103-121 and the
// strict test of intersects.comp:164-170).
RT_DEV float box_entry(const LaneRay& r, f3 bmin, f3 bmax) {
    f3 t1 = (bmin - r.o) * r.inv, t2 = (bmax - r.o) * r.inv;
    float lo = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    float hi = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    return (lo <= hi && hi > 0.0f && lo < r.t) ? lo : INFINITY;
}

// One visit of a 4-wide record by an occlusion query: the nearest entered entry is visited next, the others are stacked
// (their order does not affect the answer).
RT_DEV void wide_step(const SceneDev& sc, LaneRay& r, lds_u32* stack, uint32_t* spill, const lds_v4f* hot_tab, uint32_t n_resident) {
    static_assert(kHotWide * 48 < 65536, "chunk offsets must fit the 16-bit offset of an LDS instruction");
    v4f c0, c1, c2, c3, c4, c5, rf;
    fetch_wide(sc.wide, r.cur, n_resident, uint32_t(uintptr_t(hot_tab + r.cur)), c0, c1, c2, c3, c4, c5, rf);
    const float d0 = box_entry(r, f3{c0.x, c0.y, c0.z}, f3{c0.w, c1.x, c1.y});
    const float d1 = box_entry(r, f3{c1.z, c1.w, c2.x}, f3{c2.y, c2.z, c2.w});
    const float d2 = box_entry(r, f3{c3.x, c3.y, c3.z}, f3{c3.w, c4.x, c4.y});
    const float d3 = box_entry(r, f3{c4.z, c4.w, c5.x}, f3{c5.y, c5.z, c5.w});
    const uint32_t e0 = __float_as_uint(rf.x), e1 = __float_as_uint(rf.y), e2 = __float_as_uint(rf.z), e3 = __float_as_uint(rf.w);
    // nearest entered entry (ties: the lower index)
    float dn = d0; uint32_t near = 0u;
    if (d1 < dn) { dn = d1; near = 1u; }
    if (d2 < dn) { dn = d2; near = 2u; }
    if (d3 < dn) { dn = d3; near = 3u; }
    const bool deep = __builtin_amdgcn_ballot_w64(r.sp + 4u >= kStackLds) != 0ull;    // wave-uniform
    if (d0 < INFINITY && near != 0u) stack_push(stack, spill, r, e0, !deep);
    if (d1 < INFINITY && near != 1u) stack_push(stack, spill, r, e1, !deep);
    if (d2 < INFINITY && near != 2u) stack_push(stack, spill, r, e2, !deep);
    if (d3 < INFINITY && near != 3u) stack_push(stack, spill, r, e3, !deep);
    if (dn < INFINITY) r.cur = near == 0u ? e0 : (near == 1u ? e1 : (near == 2u ? e2 : e3));
    else r.cur = stack_pop(stack, spill, r, !deep);
}

RT_DEV float wave_pull(uint32_t src_lane, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(int(src_lane << 2), __float_as_int(v)));
}
RT_DEV uint32_t wave_pull(uint32_t src_lane, uint32_t v) {
    return uint32_t(__builtin_amdgcn_ds_bpermute(int(src_lane << 2), int(v)));
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;

// Leaf phase, executed by the whole wavefront (all 64 lanes active) for the P lanes parked on a leaf. A leaf holds two
// or three triangles on average and only a quarter of the lanes are parked when the phase starts, so the lanes are
// regrouped: with P <= 16 the wave forms 4 columns of 16 slots, slot p of column c tests triangle c of the p-th parked
// lane's leaf (P <= 32: 2 columns, else 1), on ray data pulled from the owner with ds_bpermute. The owner then folds
// its columns in triangle order with the shader's strict "closer" test (intersects.comp:141-150), so the first triangle
// tested still wins an exact tie. `slot_owner` is this wave's 64-byte LDS strip.
// `my_leaf` is the leaf reference of a parked lane (`parked`); with `pop_after` the lane continues from its stack afterwards
// (the leaf was its current node), without it the lane's current node is left alone (the leaf was a postponed one).
template <bool COUNT, bool ANYHIT>
RT_DEV void leaf_phase(const SceneDev& sc, LaneRay& r, uint32_t my_leaf, bool parked, bool pop_after, unsigned long long parked_mask, lds_u8* slot_owner,
                       lds_u32* stack, uint32_t* spill, uint32_t& n_leaf, uint32_t& n_tri) {
    const uint32_t lane = lane_id();
    const uint32_t P = uint32_t(__popcll(parked_mask));
    const uint32_t rank = bits_below_lane(parked_mask);
    const uint32_t shift = P <= 16u ? 4u : (P <= 32u ? 5u : 6u);      // wave-uniform
    const uint32_t columns = 64u >> shift;
    const uint32_t slot = lane & ((1u << shift) - 1u), col = lane >> shift;
    if (parked) slot_owner[rank] = uint8_t(lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // same wave: LDS operations complete in order
    const bool slot_used = slot < P;
    const uint32_t owner = slot_used ? uint32_t(slot_owner[slot]) : lane;
    const f3 o = {wave_pull(owner, r.o.x), wave_pull(owner, r.o.y), wave_pull(owner, r.o.z)};
    const f3 d = {wave_pull(owner, r.d.x), wave_pull(owner, r.d.y), wave_pull(owner, r.d.z)};
    const uint32_t leaf = wave_pull(owner, my_leaf);
    const uint32_t first = leaf & 0x01FFFFFFu, count = ((leaf >> 25) & 63u) + 1u;
    const uint32_t my_first = my_leaf & 0x01FFFFFFu;
#ifdef RT_OCCUPANCY_PROBE
    if (COUNT && lane == 0) ++n_tri;                     // experiment: leaf phases in the triangle counter
#else
    if (COUNT && parked) { ++n_leaf; n_tri += ((my_leaf >> 25) & 63u) + 1u; }
#endif
    for (uint32_t base = 0;; base += columns) {
        const bool mine = slot_used && base + col < count;
        if (__builtin_amdgcn_ballot_w64(mine) == 0ull) break;
        float t = INFINITY, u = 0.0f, v = 0.0f;
        bool ok = false;
        if (mine) ok = triangle_test(sc, o, d, first + base + col, t, u, v);
        const float t_res = ok ? t : INFINITY;        // NaN stays NaN and loses the comparison below
        // owners: fold the columns in triangle order
        float best_t = r.t;
        uint32_t best_col = 0xFFFFFFFFu;
        for (uint32_t c = 0; c < columns; ++c) {
            float tc = wave_pull(rank + (c << shift), t_res);
            if (tc < best_t) { best_t = tc; best_col = c; }
        }
        const uint32_t win = rank + ((best_col & 3u) << shift);
        const float bu = wave_pull(win, u), bv = wave_pull(win, v);
        if (parked && best_col != 0xFFFFFFFFu) {
            r.t = best_t; r.u = bu; r.v = bv; r.prim = my_first + base + best_col;
        }
    }
    if (parked && pop_after) {
        // ANYHIT: the caller only asks whether anything lies nearer than the initial distance, so a hit ends the ray
        if (ANYHIT && r.prim != kNoPrim) { r.cur = kDone; r.sp = 0; }
        else r.cur = stack_pop(stack, spill, r, false);
    }
}

// Work description for the persistent tracer: how item i becomes a ray and where its result goes.
//   void run_of(uint32_t i, uint32_t& lo, uint32_t& hi, uint32_t& base)  -- items [lo, hi) around i live at handles base + (item - lo)
//   void load(uint32_t handle, f3& o, f3& d, float& tmax)  -- tmax is the initial "closest so far"; store() gets the
//                                                            same handle back
// ANYHIT turns the closest-hit search into an occlusion query (stop at the first hit below tmax).
//   void store(uint32_t handle, const LaneRay& r)
// TREE: which tree is walked. kTreeBinary: the reference's binary tree in the reference's order (bit-identical ids).
// kTreeWide (occlusion queries under RT_EXACT_ORDER only): the uncompressed 4-wide tree, the reference's box arithmetic.
// kTreeCw: the compressed 4-wide tree, conservative boxes (see cw_step).
constexpr int kTreeBinary = 0, kTreeWide = 1, kTreeCw = 2;
template <bool COUNT, bool ANYHIT, int TREE, class Work>
RT_DEV void trace_persistent(const SceneDev& sc, const Work& work, uint32_t n, uint32_t* cursor, lds_u32* stack, lds_u8* slot_owner,
                             const HotNodes& hot, uint32_t& n_node, uint32_t& n_leaf, uint32_t& n_tri, uint32_t* visits = nullptr) {
    constexpr uint32_t kRefillAt = ANYHIT ? RT_REFILL_AT_ANYHIT : RT_REFILL_AT;     // refill once this many lanes sit idle
    const uint32_t waves_total = (gridDim.x * kTraceBlock) >> 6;
    // intersect parked leaves once this many lanes wait on one: 16 where the leaf phase regroups the wavefront (closest hits),
    // 32 where every lane loops over its own leaf (queries that stop at the first hit): connect 0.573 -> 0.549 ms per sample,
    // 24: 0.558, 40: 0.566, 48: 0.634
    constexpr uint32_t kLeafBatch = ANYHIT ? RT_LEAF_BATCH_ANYHIT : RT_LEAF_BATCH;
    static_assert(TREE != kTreeWide || ANYHIT, "the uncompressed 4-wide tree serves order-independent queries only");
    constexpr bool WIDE = TREE != kTreeBinary;
    uint32_t spill[(WIDE ? kStackMaxWide : kStackMax) - kStackLds + 1];     // + 1: without a top-of-stack register the deepest entry takes a slot too
    LaneRay r;
    r.cur = kDone; r.sp = 0; r.item = 0; r.prim = kNoPrim; r.t = INFINITY; r.u = r.v = 0.0f;
    stack[0] = kDone;                                // slot 0 of this lane's stack: what a pop of the empty stack reads
    r.o = r.d = r.inv = f3{0.0f, 0.0f, 0.0f};
    bool live = false;
    bool drained = false;       // wave-uniform: the cursor has passed the end of the work list
    const uint32_t lane = lane_id();
    // Work is handed out in chunks of consecutive items (neighbouring pixels / queue entries): a wave refills its idle
    // lanes from its current chunk, so the rays it holds stay spatially close. The first part of the list is dealt out
    // statically (chunk k goes to wave k mod #waves: no shared counter); the last RT_TAIL_EIGHTHS/8 of it is handed out
    // through the device-wide cursor in small chunks, so waves that finish their share early take over work from slow
    // ones and the grid runs down together. Without a cursor the whole list is static.
    uint32_t chunk_pos = 0, chunk_end = 0;                                   // wave-uniform
    // readfirstlane: the compiler cannot prove these wave-uniform by itself, and as scalars the whole work-distribution state
    // (chunk, static share, cursor segment, run cache) stays out of the vector registers the traversal needs
    // Waves are numbered across the workgroups first (wave w of workgroup b is b + w * #workgroups), so a list shorter than
    // the grid is spread over all compute units instead of filling the first ones.
    const uint32_t wave_id = RT_WAVE_INTERLEAVE ? __builtin_amdgcn_readfirstlane(blockIdx.x + (threadIdx.x >> 6) * gridDim.x)
                                                : __builtin_amdgcn_readfirstlane((blockIdx.x * kTraceBlock + threadIdx.x) >> 6);
    constexpr uint32_t kTailChunk = ANYHIT ? RT_TAIL_CHUNK_ANYHIT : RT_TAIL_CHUNK;
    constexpr uint32_t kTailEighths = ANYHIT ? RT_TAIL_EIGHTHS_ANYHIT : RT_TAIL_EIGHTHS;
    // The static share is at least one 64-item chunk per wave: below that the cursor could not balance anything (a wave's
    // grain is one chunk) and every wave of the grid would queue up at it for nothing, about 90 atomics per microsecond.
    // The static chunk is the static share dealt evenly, between 64 and RT_CHUNK items.
    uint32_t static_target = cursor ? n - ((n >> 3) * kTailEighths) : n;
    if (RT_STATIC_FLOOR && static_target < waves_total * 64u) static_target = n < waves_total * 64u ? n : waves_total * 64u;
    uint32_t kChunk = (((RT_STATIC_FLOOR ? static_target : n) / waves_total) + 63u) & ~63u;
    kChunk = kChunk < 64u ? 64u : (kChunk > RT_CHUNK ? RT_CHUNK : kChunk);
    const uint32_t n_static = static_target == n ? n : (static_target / kChunk) * kChunk;
    if (n_static == n) cursor = nullptr;
    uint32_t next_static = wave_id * kChunk;
    const uint32_t static_stride = waves_total * kChunk;
    bool static_done = false;
    // The dynamic part [n_static, n) can be cut into kTailCursors segments, each with a cursor of its own in its own 256-byte
    // block (a wave starts at segment (wave mod kTailCursors) and moves on when a segment is used up). Returning atomics on
    // one address are served one at a time, about 90 per microsecond on this chip, and a 20-frame launch asks for nearly
    // that while it runs down; measured, though, 8 or 16 cursors make extend 2 % faster on its own and the whole frame 4 %
    // slower when extend and connect overlap (5315 vs 5110 Mrays/s, four alternating runs each), so the default is one.
    const uint32_t seg_len = cursor ? (((n - n_static) + kTailCursors * kTailChunk - 1u) / (kTailCursors * kTailChunk)) * kTailChunk : 0u;
    uint32_t seg = wave_id % kTailCursors, segs_empty = 0;
    uint32_t run_lo = 1u, run_hi = 0u, run_base = 0u;                        // wave-uniform: where the items around chunk_pos live (empty at first)
    // Postponed leaves (closest-hit queries on the compressed tree only; RT_SPECULATE): see phase A
    constexpr bool SPECULATE = RT_SPECULATE && TREE == kTreeCw && !ANYHIT;
    uint32_t pend = 0u;
    for (;;) {
        unsigned long long idle = __builtin_amdgcn_ballot_w64(!live);
        uint32_t n_idle = uint32_t(__popcll(idle));
        if (!drained && n_idle >= kRefillAt) {
            if (chunk_pos >= chunk_end) {
                if (!static_done) {
                    chunk_pos = next_static;
                    chunk_end = chunk_pos + kChunk < n_static ? chunk_pos + kChunk : n_static;
                    next_static += static_stride;
                    if (chunk_pos >= n_static) static_done = true;
                }
                if (static_done) {
                    chunk_pos = chunk_end = 0;
                    while (cursor && segs_empty < kTailCursors) {
                        uint32_t off = seg_len;
                        if (lane == 0) off = atomicAdd(cursor + seg * kCounterStride, kTailChunk);
                        off = __builtin_amdgcn_readfirstlane(off);
                        const uint32_t seg_begin = n_static + seg * seg_len;
                        const uint32_t seg_end = seg_begin + seg_len < n ? seg_begin + seg_len : n;
                        if (off < seg_len && seg_begin + off < seg_end) {
                            chunk_pos = seg_begin + off;
                            chunk_end = chunk_pos + kTailChunk < seg_end ? chunk_pos + kTailChunk : seg_end;
                            segs_empty = 0;
                            break;
                        }
                        seg = seg + 1u == kTailCursors ? 0u : seg + 1u;
                        ++segs_empty;
                    }
                    if (chunk_pos >= chunk_end) { drained = true; chunk_pos = chunk_end = 0; }
                }
            }
            uint32_t avail = chunk_end - chunk_pos;
            uint32_t rank = bits_below_lane(idle);
            if (avail != 0u && (chunk_pos < run_lo || chunk_pos >= run_hi)) {
                work.run_of(chunk_pos, run_lo, run_hi, run_base);
                run_lo = __builtin_amdgcn_readfirstlane(run_lo);
                run_hi = __builtin_amdgcn_readfirstlane(run_hi);
                run_base = __builtin_amdgcn_readfirstlane(run_base);
            }
            if (!live && rank < avail) {
                uint32_t mine = chunk_pos + rank;
                if (__builtin_expect(mine >= run_hi, 0)) {                   // the chunk runs on into the next region
                    uint32_t lo2, hi2, base2;
                    work.run_of(mine, lo2, hi2, base2);
                    r.item = base2 + (mine - lo2);
                } else {
                    r.item = run_base + (mine - run_lo);
                }
                float tmax;
                work.load(r.item, r.o, r.d, tmax);
                r.t = tmax;
                start_ray<TREE == kTreeCw>(r, WIDE ? sc.wide_root : sc.root_ref);
                live = true;
            }
            chunk_pos += n_idle < avail ? n_idle : avail;
        }
        if (__builtin_amdgcn_ballot_w64(live) == 0ull) {
            if (drained) break;
            continue;
        }
        // phase A: interior nodes, until enough lanes are parked on a leaf (or no lane has a node left)
        for (;;) {
            if (SPECULATE) {
                // a lane that reaches a leaf while it holds no postponed one keeps the leaf for later and goes on with its stack
                const bool take = int32_t(r.cur) < 0 && pend == 0u;
                if (__builtin_amdgcn_ballot_w64(take) != 0ull) {
                    if (take) { pend = r.cur; r.cur = stack_pop(stack, spill, r, false); }
                }
            }
            const bool at_node = r.cur < kDone;            // lanes without a ray hold kDone
            if (__builtin_amdgcn_ballot_w64(at_node) == 0ull) break;
            if (uint32_t(__popcll(__builtin_amdgcn_ballot_w64(SPECULATE ? pend != 0u : int32_t(r.cur) < 0))) >= kLeafBatch) break;
#ifdef RT_OCCUPANCY_PROBE
            if (COUNT && lane == 0) ++n_leaf;            // experiment: wave-level node steps in the leaf counter
#endif
            if (at_node) {
                if (COUNT) {
                    ++n_node;
                    if (visits && (r.item & kVisitSampleMask) == 0u) atomicAdd(visits + r.cur, 1u);
                }
                if (TREE == kTreeCw) cw_step(sc, r, stack, spill, hot.tab, hot.n);
                else if (TREE == kTreeWide) wide_step(sc, r, stack, spill, hot.tab, hot.n);
                else node_step(sc, r, stack, spill, hot);
            }
        }
        // phase B: the parked leaves
        {
            const unsigned long long parked_mask = __builtin_amdgcn_ballot_w64(int32_t(r.cur) < 0);
            if (SPECULATE) {
                // the postponed leaves of all lanes together; a lane that has run into a second leaf meanwhile keeps it as its
                // current node and postpones it in turn at the top of the next phase A
                const unsigned long long pend_mask = __builtin_amdgcn_ballot_w64(pend != 0u);
                if (pend_mask != 0ull) leaf_phase<COUNT, ANYHIT>(sc, r, pend, pend != 0u, false, pend_mask, slot_owner, stack, spill, n_leaf, n_tri);
                pend = 0u;
            } else if (RT_COOP_ANYHIT || !ANYHIT) {
                if (parked_mask != 0ull) leaf_phase<COUNT, ANYHIT>(sc, r, r.cur, int32_t(r.cur) < 0, true, parked_mask, slot_owner, stack, spill, n_leaf, n_tri);
            } else if (int32_t(r.cur) < 0) {
                // occlusion queries: each parked lane walks its own leaf and stops at the first hit
                if (COUNT) ++n_leaf;
                const uint32_t first = r.cur & 0x01FFFFFFu, count = ((r.cur >> 25) & 63u) + 1u;
                for (uint32_t k = 0; k < count; ++k) {
                    if (COUNT) ++n_tri;
                    float t, u, v;
                    if (triangle_test(sc, r.o, r.d, first + k, t, u, v) && t < r.t) { r.t = t; r.u = u; r.v = v; r.prim = first + k; break; }
                }
                if (r.prim != kNoPrim) { r.cur = kDone; r.sp = 0; }
                else r.cur = stack_pop(stack, spill, r, false);
            }
        }
        if (live && r.cur == kDone) {
            work.store(r.item, r);
            live = false;
        }
    }
}

// ---- kernels -------------------------------------------------------------------------------------------

// Which pixels a context renders. stride == 1: all of them, local pixel k = pixel k (row-major). stride > 1: the image is cut
// into 8x8 tiles (the reference's workgroup footprint, main.comp:13), numbered row by row, and the context owns tiles first,
// first + stride, ...; local pixel k is pixel (k & 7, (k >> 3) & 7) of its (k >> 6)-th tile. A tile that overhangs the right
// or bottom edge has local pixels without a pixel: they get a ray that hits nothing and carries no throughput, and the
// accumulation skips them.
struct TileSlice {
    uint32_t first, stride, tiles_x, width, height;
    RT_DEV bool pixel_of(uint32_t local, uint32_t& px, uint32_t& py) const {
        if (stride == 1u) { px = local % width; py = local / width; return true; }
        const uint32_t tile = first + (local >> 6) * stride;
        px = (tile % tiles_x) * 8u + (local & 7u);
        py = (tile / tiles_x) * 8u + ((local >> 3) & 7u);
        return px < width && py < height;
    }
};

// Seed rule + thin-lens camera ray for every slot (main.comp:251; generate_ray.comp:2-29). `n_pixels` = local pixels per frame.
__global__ __launch_bounds__(kBlock) void k_raygen(rt_camera cam, PathDev pd, float4* __restrict__ rad, uint32_t n_pixels, uint32_t n_slots,
                                                   uint32_t frame_first, uint32_t frame_stride, uint32_t* seeds_out, TileSlice tiles) {
    const uint32_t W = cam.resolution[0];
    const float* m = cam.rotation;
    for (uint32_t slot = blockIdx.x * kBlock + threadIdx.x; slot < n_slots; slot += gridDim.x * kBlock) {
        uint32_t pixel = slot % n_pixels;
        uint32_t frame = frame_first + (slot / n_pixels) * frame_stride;
        uint32_t px, py;
        if (!tiles.pixel_of(pixel, px, py)) {
            const float nan = __uint_as_float(0x7FC00000u);
            pd.o[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            pd.d[slot] = make_float4(nan, nan, nan, 0.0f);                      // a non-finite ray reports a miss at once
            pd.t[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(slot));
            rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (seeds_out) seeds_out[slot] = 0u;
            continue;
        }
        uint32_t seed = px + W * py + frame * py * px;
        float k = cam.focus_distance / cam.focal_length;
        float fpx = cam.view_port_uv[0] * k, fpy = cam.view_port_uv[1] * k;
        float dx = fpx / float(cam.resolution[0]);
        float dy = (fpy / float(cam.resolution[1])) * -1.0f;
        f3 origin = f3{-fpx / 2.0f, fpy / 2.0f, -cam.focus_distance} + f3{dx / 2.0f, dy / 2.0f, 0.0f};
        // pixel_offset with quadrant 0 (random.comp:16-27): jitter in [-0.5, 0)^2
        float jx = pcg_next(seed) * 0.5f;
        float jy = pcg_next(seed) * 0.5f;
        jx += 0.0f; jy += 0.0f;
        jx -= 0.5f; jy -= 0.5f;
        f3 pix = origin + f3{dx * (float(px) + jx), dy * (float(py) + jy), 0.0f};
        // random_disc (random.comp:30-35,54-57): two Box-Muller normals, normalised, times a uniform radius
        float a1 = sqrtf(-2.0f * logf(1.0f - pcg_next(seed)));
        float b1 = 2.0f * kPi * pcg_next(seed);
        float n1 = cosf(b1) * fminf(a1, 1000000000.0f);
        float a2 = sqrtf(-2.0f * logf(1.0f - pcg_next(seed)));
        float b2 = 2.0f * kPi * pcg_next(seed);
        float n2 = cosf(b2) * fminf(a2, 1000000000.0f);
        float rn = 1.0f / sqrtf(n1 * n1 + n2 * n2);
        float radius = pcg_next(seed);
        f3 ap = f3{n1 * rn * radius, n2 * rn * radius, 0.0f} * cam.aperture_radius;
        f3 dl = normalize(pix - ap);
        f3 dw = {m[0] * dl.x + m[4] * dl.y + m[8] * dl.z, m[1] * dl.x + m[5] * dl.y + m[9] * dl.z,
                 m[2] * dl.x + m[6] * dl.y + m[10] * dl.z};
        f3 ow = f3{m[0] * ap.x + m[4] * ap.y + m[8] * ap.z, m[1] * ap.x + m[5] * ap.y + m[9] * ap.z,
                   m[2] * ap.x + m[6] * ap.y + m[10] * ap.z} + f3{cam.location[0], cam.location[1], cam.location[2]};
        (void)pcg_next(seed);  // ray.time
        pd.o[slot] = make_float4(ow.x, ow.y, ow.z, 0.0f);                      // queue position of bounce 0 = slot
        pd.d[slot] = make_float4(dw.x, dw.y, dw.z, __uint_as_float(seed));
        pd.t[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(slot));
        rad[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (seeds_out) seeds_out[slot] = seed;
    }
}

// LDS of a traversal workgroup: the per-lane stack heads and the resident node records (filled before any ray is traced).
#define RT_TRACE_LDS(tree)                                                                                       \
    __shared__ uint32_t s_prefix[kQueueShards + 1];                                                               \
    __shared__ uint32_t s_stack[kStackLds * kTraceBlock];                                                         \
    __shared__ uint8_t s_owner[kTraceBlock];                                                                      \
    __shared__ v4f s_hot[kHotLds ? kHotLds * 4 : 1];                                                              \
    const HotNodes hot = (tree) == kTreeWide ? HotNodes{(const lds_v4f*)s_hot, sc.n_hot_wide < kHotWide ? sc.n_hot_wide : kHotWide} \
                       : (tree) == kTreeCw   ? HotNodes{(const lds_v4f*)s_hot, sc.n_hot_wide < kHotLds ? sc.n_hot_wide : kHotLds}   \
                                             : HotNodes{(const lds_v4f*)s_hot, sc.n_hot < kHotLds ? sc.n_hot : kHotLds};            \
    if (kHotLds) {                                                                                                \
        if ((tree) == kTreeWide) load_hot_wide(sc, s_hot, hot.n);                                                 \
        else load_hot_nodes((tree) == kTreeCw ? sc.cw : sc.nodes, s_hot, hot.n);                                  \
    }                                                                                                             \
    (void)s_prefix;

// Record i of an array of 16-byte records, addressed as scalar base + 32-bit byte offset: the lane keeps one register for
// the index instead of a zero-extended pair. Valid below 2^28 records (the path-slot limit of a context).
template <class T>
RT_DEV T* rec16(T* base, uint32_t i) {
    static_assert(sizeof(T) == 16, "16-byte records");
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + i * 16u);
}

struct ExtendWork {
    PathDev pd;
    float4* __restrict__ hit;     // {t, u, v, prim bits} per queue position
    RegionMap map;
    RT_DEV void run_of(uint32_t i, uint32_t& lo, uint32_t& hi, uint32_t& base) const { map.run_of(i, lo, hi, base); }
    RT_DEV void load(uint32_t p, f3& o, f3& d, float& tmax) const {
        o = xyz(*rec16(pd.o, p));
        d = xyz(*rec16(pd.d, p));
        tmax = INFINITY;
    }
    RT_DEV void store(uint32_t p, const LaneRay& r) const { *rec16(hit, p) = make_float4(r.t, r.u, r.v, __uint_as_float(r.prim)); }
};

// Occluded = a closest hit nearer than tmax whose material is not glass-like (disney.comp:697-701,
// intersects.comp:152-154). A path has at most one light sample per bounce, so the add to its slot needs no atomic.
struct ConnectWork {
    const float4* __restrict__ tri_shade;   // the triangles' shading records: chunk 0 carries the glass bit (only that pointer, not the whole scene view, stays live through the traversal)
    ShadowDev sq;
    float4* __restrict__ rad;     // per slot {radiance rgb, -}
    RegionMap map;
    RT_DEV void run_of(uint32_t i, uint32_t& lo, uint32_t& hi, uint32_t& base) const { map.run_of(i, lo, hi, base); }
    RT_DEV void load(uint32_t p, f3& o, f3& d, float& tmax) const {
        float4 o4 = *rec16(sq.o, p);
        o = xyz(o4);
        d = xyz(*rec16(sq.d, p));
        tmax = o4.w;
    }
    RT_DEV void store(uint32_t p, const LaneRay& r) const {
        bool blocked = false;
        if (r.prim != kNoPrim) blocked = (__float_as_uint(tri_shade[size_t(r.prim) * 4].w) & 0x80000000u) == 0u;
        if (!blocked) {
            const uint32_t slot = __float_as_uint(rec16(sq.d, p)->w);
            float4 c = *rec16(sq.c, p);
            float4* ap = rec16(rad, slot);
            float4 a = *ap;
            a.x += c.x; a.y += c.y; a.z += c.z;
            *ap = a;
        }
    }
};

template <bool COUNT>
RT_DEV void flush_counts(uint32_t n_node, uint32_t n_leaf, uint32_t n_tri, unsigned long long* node, unsigned long long* leaf,
                         unsigned long long* tri) {
    if (!COUNT) return;
    n_node = wave_sum(n_node); n_leaf = wave_sum(n_leaf); n_tri = wave_sum(n_tri);
    if (lane_id() == 0) {
        atomicAdd(node, (unsigned long long)n_node);
        atomicAdd(leaf, (unsigned long long)n_leaf);
        atomicAdd(tri, (unsigned long long)n_tri);
    }
}

// extend: closest hit for every path of the bounce (a persistent grid; `cursor` hands out the tail of the queue).
// FAST: walk the compressed 4-wide tree instead of the reference's binary tree in the reference's order.
template <bool COUNT, bool FAST>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_WAVES_PER_EU) void k_extend(SceneDev sc, PathDev pd, float4* __restrict__ hit, Regions rg,
                                                   uint32_t* cursor, TraceStats* stats) {
    constexpr int TREE = FAST ? kTreeCw : kTreeBinary;
    RT_TRACE_LDS(TREE)
    const uint32_t n = load_region_prefix(rg, s_prefix);
    uint32_t n_node = 0, n_leaf = 0, n_tri = 0;
    ExtendWork work{pd, hit, RegionMap{(const lds_u32*)s_prefix, rg.cap, rg.counts == nullptr}};
    trace_persistent<COUNT, false, TREE>(sc, work, n, cursor, (lds_u32*)s_stack + threadIdx.x, (lds_u8*)s_owner + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u), hot, n_node, n_leaf, n_tri,
                                         COUNT ? stats->visits[TREE == kTreeBinary ? 0 : 1] : nullptr);
    flush_counts<COUNT>(n_node, n_leaf, n_tri, &stats->extend_node_pops, &stats->extend_leaf_pops, &stats->extend_tri_tests);
}

// connect: visibility of every queued light sample; adds the pre-weighted contribution when unoccluded.
// When no material of the scene is glass-like (spec_trans > 0.5) "the closest hit is opaque" is the same predicate as
// "there is a hit", so the query stops at the first hit (ANYHIT); with glass present the closest hit decides.
template <bool COUNT, bool ANYHIT, bool FAST>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_WAVES_PER_EU) void k_connect(SceneDev sc, ShadowDev sq, float4* __restrict__ rad, Regions rg,
                                                    uint32_t* cursor, TraceStats* stats) {
    // By default every query walks the compressed tree: an occlusion query that may stop at the first hit finds a blocker
    // below tmax whenever the reference does (the compressed boxes contain the reference's), whatever the visiting order, and
    // the 64-byte records cost it less than the 112-byte ones of the uncompressed 4-wide tree (0.666 -> 0.637 ms per sample
    // on the 1080p atrium). Under RT_EXACT_ORDER such queries use the uncompressed 4-wide tree, which reproduces the
    // reference's box arithmetic, and closest-hit queries (scenes with glass) the binary tree.
    constexpr int TREE = FAST ? kTreeCw : (ANYHIT ? kTreeWide : kTreeBinary);
    RT_TRACE_LDS(TREE)
    const uint32_t n = load_region_prefix(rg, s_prefix);
    uint32_t n_node = 0, n_leaf = 0, n_tri = 0;
    ConnectWork work{sc.shade, sq, rad, RegionMap{(const lds_u32*)s_prefix, rg.cap, rg.counts == nullptr}};
    trace_persistent<COUNT, ANYHIT, TREE>(sc, work, n, cursor, (lds_u32*)s_stack + threadIdx.x, (lds_u8*)s_owner + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u), hot, n_node, n_leaf, n_tri,
                                          COUNT ? stats->visits[TREE == kTreeBinary ? 0 : 1] : nullptr);
    flush_counts<COUNT>(n_node, n_leaf, n_tri, &stats->shadow_node_pops, &stats->shadow_leaf_pops, &stats->shadow_tri_tests);
}

// ---- accumulate / display ------------------------------------------------------------------------------
RT_DEV f3 suppress_fireflies(f3 c, float max_value) {  // main.comp:114-120
    float lum = luminance(c);
    if (lum > max_value) c = c * (max_value / lum);
    return c;
}
RT_DEV float hable(float x) {
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
RT_DEV f3 filmic_tone_map(f3 c) {  // main.comp:215-240
    c = c * 1.5f;
    f3 r = {hable(c.x), hable(c.y), hable(c.z)};
    r = r / hable(11.2f);
    float l = dot(r, f3{0.2126f, 0.7152f, 0.0722f});
    r = mix3(mk3(l), r, 1.7f);
    const float g = 1.0f / 2.4f;
    return {powf(fmaxf(r.x, 0.0f), g), powf(fmaxf(r.y, 0.0f), g), powf(fmaxf(r.z, 0.0f), g)};
}
RT_DEV uint32_t unorm8(float x) {  // UNORM store: NaN and negatives give 0
    if (!(x > 0.0f)) return 0u;
    if (x >= 1.0f) return 255u;
    return uint32_t(floorf(x * 255.0f + 0.5f));
}
RT_DEV uint32_t pack_rgba8(float r, float g, float b, float a) {
    return unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | (unorm8(a) << 24);
}

// Folds the finished samples of one pass into the accumulator, in frame order (main.comp:260-267).
// HDR mode: fp32 RGBA sum with the sample count in alpha; a non-finite sample component contributes 0.
// Legacy mode: the shader's running mean stored in 8-bit UNORM every frame, plus the tone-mapped framebuffer.
// `n_pixels` = local pixels per frame (the layout of `rad`); `tiles` maps a local pixel to its place in the image.
__global__ __launch_bounds__(kBlock) void k_accumulate(const float4* __restrict__ rad, float4* __restrict__ sum, uint32_t* __restrict__ accum8,
                                                       uint32_t* __restrict__ framebuffer, uint32_t n_pixels,
                                                       uint32_t frames_in_pass, uint32_t frame_first, uint32_t frame_stride,
                                                       uint32_t legacy8, TileSlice tiles) {
    for (uint32_t lp = blockIdx.x * kBlock + threadIdx.x; lp < n_pixels; lp += gridDim.x * kBlock) {
        uint32_t px, py;
        if (!tiles.pixel_of(lp, px, py)) continue;
        const uint32_t p = py * tiles.width + px;
        if (legacy8) {
            uint32_t w = accum8[p];
            float old[4] = {float(w & 255u) / 255.0f, float((w >> 8) & 255u) / 255.0f, float((w >> 16) & 255u) / 255.0f,
                            float(w >> 24) / 255.0f};
            f3 fb = {0, 0, 0};
            float fa = 0.0f;
            for (uint32_t f = 0; f < frames_in_pass; ++f) {
                float4 r = rad[size_t(f) * n_pixels + lp];
                f3 c = suppress_fireflies(f3{r.x, r.y, r.z}, 5.0f);
                float alpha = 1.0f / float(frame_first + f * frame_stride + 1u);
                float nw[4] = {mixf(old[0], c.x, alpha), mixf(old[1], c.y, alpha), mixf(old[2], c.z, alpha), mixf(old[3], 1.0f, alpha)};
                fb = {nw[0], nw[1], nw[2]};
                fa = nw[3];
                w = pack_rgba8(nw[0], nw[1], nw[2], nw[3]);
                for (int k = 0; k < 4; ++k) old[k] = float((w >> (8 * k)) & 255u) / 255.0f;
            }
            accum8[p] = w;
            f3 tm = filmic_tone_map(fb);
            framebuffer[p] = pack_rgba8(tm.x, tm.y, tm.z, fa);
        } else {
            float4 s = sum[p];
            for (uint32_t f = 0; f < frames_in_pass; ++f) {
                float4 r = rad[size_t(f) * n_pixels + lp];
                f3 c = suppress_fireflies(f3{r.x, r.y, r.z}, 5.0f);
                s.x += isfinite(c.x) ? c.x : 0.0f;
                s.y += isfinite(c.y) ? c.y : 0.0f;
                s.z += isfinite(c.z) ? c.z : 0.0f;
                s.w += 1.0f;
            }
            sum[p] = s;
        }
    }
}

// Tone-maps the HDR mean into an RGBA8 framebuffer (what SaveFrame would write).
__global__ __launch_bounds__(kBlock) void k_tonemap(const float4* __restrict__ sum, uint32_t* __restrict__ framebuffer, uint32_t n_pixels) {
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pixels; p += gridDim.x * kBlock) {
        float4 s = sum[p];
        float inv = s.w > 0.0f ? 1.0f / s.w : 0.0f;
        f3 tm = filmic_tone_map(f3{s.x * inv, s.y * inv, s.z * inv});
        framebuffer[p] = pack_rgba8(tm.x, tm.y, tm.z, s.w > 0.0f ? 1.0f : 0.0f);
    }
}

__global__ void k_add_sum(float4* __restrict__ sum, const float4* __restrict__ other, uint32_t n_pixels) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_pixels; p += gridDim.x * blockDim.x) {
        float4 a = sum[p], b = other[p];
        sum[p] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

// End of a pass: fold the per-bounce queue lengths into the 64-bit totals (the counters are cleared by a memset
// at the start of the next pass). `shard_counts` holds, per bounce, kQueueShards counters of the paths that continued
// (= the rays extend traces in the next bounce) followed by kQueueShards counters of the light samples queued.
__global__ void k_finish_pass(const uint32_t* __restrict__ shard_counts, uint32_t n_bounces, uint32_t camera_rays, TraceStats* stats) {
    if (blockIdx.x != 0 || threadIdx.x >= kQueueShards) return;
    unsigned long long e = 0, s = 0;
    for (uint32_t b = 0; b < n_bounces; ++b) {
        if (b + 1 < n_bounces) e += shard_counts[(b * 2 * kQueueShards + threadIdx.x) * kCounterStride];      // survivors of the last bounce are never traced
        s += shard_counts[(b * 2 * kQueueShards + kQueueShards + threadIdx.x) * kCounterStride];
    }
    for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off); s += __shfl_down(s, off); }
    if (threadIdx.x == 0) {                 // atomics: the per-path kernel of a pass on another lane may be adding to the same totals
        atomicAdd(&stats->extend_rays, e + camera_rays);
        atomicAdd(&stats->shadow_rays, s);
        atomicAdd(&stats->camera_rays, (unsigned long long)camera_rays);
    }
}

// ---- stage-level kernels on caller-supplied inputs -----------------------------------------------------
struct RayListWork {
    const float4* __restrict__ tri_shade;
    const float4* __restrict__ ro;   // {origin, tmax when shadow}
    const float4* __restrict__ rd;
    float4* __restrict__ hits;
    int shadow;
    RT_DEV void run_of(uint32_t, uint32_t& lo, uint32_t& hi, uint32_t& base) const { lo = 0u; hi = 0xFFFFFFFFu; base = 0u; }
    RT_DEV void load(uint32_t i, f3& o, f3& d, float& tmax) const {
        float4 o4 = ro[i];
        o = xyz(o4);
        d = xyz(rd[i]);
        tmax = shadow ? o4.w : INFINITY;
    }
    RT_DEV void store(uint32_t i, const LaneRay& r) const {
        if (shadow) {
            bool blocked = r.prim != kNoPrim && (__float_as_uint(tri_shade[size_t(r.prim) * 4].w) & 0x80000000u) == 0u;
            hits[i] = make_float4(blocked ? 0.0f : 1.0f, 0.0f, 0.0f, __uint_as_float(r.prim));
        } else {
            hits[i] = make_float4(r.t, r.u, r.v, __uint_as_float(r.prim));
        }
    }
};

template <bool COUNT, bool FAST>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_WAVES_PER_EU) void k_trace_rays(SceneDev sc, const float4* __restrict__ ro, const float4* __restrict__ rd,
                                                       uint32_t n, int shadow, float4* __restrict__ hits, uint32_t* cursor,
                                                       TraceStats* stats) {
    const bool any_hit = shadow && !sc.has_glass;       // wave-uniform
    const int tree = FAST ? kTreeCw : (any_hit ? kTreeWide : kTreeBinary);
    RT_TRACE_LDS(tree)
    uint32_t n_node = 0, n_leaf = 0, n_tri = 0;
    RayListWork work{sc.shade, ro, rd, hits, shadow};
    lds_u32* st = (lds_u32*)s_stack + threadIdx.x;
    lds_u8* ow = (lds_u8*)s_owner + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u);
    if (any_hit && FAST) trace_persistent<COUNT, true, kTreeCw>(sc, work, n, cursor, st, ow, hot, n_node, n_leaf, n_tri);
    else if (any_hit) trace_persistent<COUNT, true, kTreeWide>(sc, work, n, cursor, st, ow, hot, n_node, n_leaf, n_tri);
    else if (FAST) trace_persistent<COUNT, false, kTreeCw>(sc, work, n, cursor, st, ow, hot, n_node, n_leaf, n_tri);
    else trace_persistent<COUNT, false, kTreeBinary>(sc, work, n, cursor, st, ow, hot, n_node, n_leaf, n_tri);
    if (shadow) flush_counts<COUNT>(n_node, n_leaf, n_tri, &stats->shadow_node_pops, &stats->shadow_leaf_pops, &stats->shadow_tri_tests);
    else flush_counts<COUNT>(n_node, n_leaf, n_tri, &stats->extend_node_pops, &stats->extend_leaf_pops, &stats->extend_tri_tests);
}

// Rays as the C ABI hands them over (n x 3 packed floats), traced without repacking.
struct PackedRayWork {
    const float* __restrict__ o3;
    const float* __restrict__ d3;
    float4* __restrict__ hits;
    RT_DEV void run_of(uint32_t, uint32_t& lo, uint32_t& hi, uint32_t& base) const { lo = 0u; hi = 0xFFFFFFFFu; base = 0u; }
    RT_DEV void load(uint32_t i, f3& o, f3& d, float& tmax) const {
        o = f3{o3[3 * size_t(i)], o3[3 * size_t(i) + 1], o3[3 * size_t(i) + 2]};
        d = f3{d3[3 * size_t(i)], d3[3 * size_t(i) + 1], d3[3 * size_t(i) + 2]};
        tmax = INFINITY;
    }
    RT_DEV void store(uint32_t i, const LaneRay& r) const { hits[i] = make_float4(r.t, r.u, r.v, __uint_as_float(r.prim)); }
};

template <bool COUNT, bool FAST>
__global__ __launch_bounds__(kTraceBlock, RT_TRACE_WAVES_PER_EU) void k_trace_packed(SceneDev sc, const float* __restrict__ o3, const float* __restrict__ d3,
                                                         uint32_t n, float4* __restrict__ hits, uint32_t* cursor, TraceStats* stats) {
    constexpr int TREE = FAST ? kTreeCw : kTreeBinary;
    RT_TRACE_LDS(TREE)
    uint32_t n_node = 0, n_leaf = 0, n_tri = 0;
    PackedRayWork work{o3, d3, hits};
    trace_persistent<COUNT, false, TREE>(sc, work, n, cursor, (lds_u32*)s_stack + threadIdx.x, (lds_u8*)s_owner + __builtin_amdgcn_readfirstlane(threadIdx.x & ~63u), hot, n_node, n_leaf, n_tri);
    flush_counts<COUNT>(n_node, n_leaf, n_tri, &stats->extend_node_pops, &stats->extend_leaf_pops, &stats->extend_tri_tests);
}

__global__ void k_sample_texture(TexView tv, uint32_t tex, uint32_t n, const float* __restrict__ uvw, float4* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f4 c = tex == 0xFFFFFFFFu ? sample_cube(tv, f3{uvw[3 * i], uvw[3 * i + 1], uvw[3 * i + 2]})
                                  : sample_2d(tv, tex, uvw[3 * i], uvw[3 * i + 1]);
        out[i] = make_float4(c.x, c.y, c.z, c.w);
    }
}

}  // namespace rtdev
