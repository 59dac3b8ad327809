// Traversal primitives shared by the wavefront kernels (rt_kernels.h) and the per-path kernel of the shade stage's
// translation units (rt_shade.h): the lane state of a ray, its stack, one visit of a compressed 4-wide record, the triangle
// test, and a plain one-ray-per-lane walk built from them.
//
// The arithmetic of everything in this header is the same in every translation unit, whatever arithmetic mode the shade
// stage around it is compiled in (rt_device_math.h): no FMA contraction, IEEE division, and none of the mode-dependent vector
// helpers, so a ray finds the same hit whichever kernel traces it.
#pragma once

#include "rt_kernels_common.h"

#pragma clang fp contract(off)

namespace rtdev {

RT_DEV float tv_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 tv_cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

constexpr uint32_t kDone = 0x7FFFFFFEu;       // lane has no node to visit; also the "empty" marker of the stack

struct LaneRay {
    f3 o, d, inv;
    float t, u, v;
    uint32_t prim;
    uint32_t cur;        // node record index, leaf reference, or kDone (lanes without a ray always hold kDone)
    uint32_t sp;         // entries on the stack; entry k (1-based) lives in memory slot k, slot 0 always reads kDone
    uint32_t item;       // handle of the work item this lane is tracing (what Work::load returned)
};

// The LDS part of the stack is addressed through an explicit LDS pointer type so pushes and pops stay ds_write /
// ds_read; with generic pointers the compiler folds the LDS and scratch cases into one flat access whose
// completion every iteration then has to wait for.
// Memory slots of the stack: slot 0 is a permanent kDone (so popping an empty stack needs no test), slots
// 1 .. kStackLds-1 are in LDS (STRIDE words apart: one entry of every lane of the workgroup side by side), higher ones in scratch.
template <uint32_t STRIDE = kTraceBlock>
RT_DEV void slot_store(lds_u32* stack, uint32_t* spill, uint32_t slot, uint32_t v) {
    if (__builtin_expect(slot < kStackLds, 1)) stack[slot * STRIDE] = v;
    else spill[slot - kStackLds] = v;
}
template <uint32_t STRIDE = kTraceBlock>
RT_DEV uint32_t slot_load(const lds_u32* stack, const uint32_t* spill, uint32_t slot) {
    if (__builtin_expect(slot < kStackLds, 1)) return stack[slot * STRIDE];
    return spill[slot - kStackLds];
}
// push / pop. The whole stack lives in memory (an earlier version kept the top entry in a register, which takes the LDS
// round trip out of a pop but costs every push three more vector-ALU instructions; the kernels are bound by
// vector-ALU issue, and the plain form is 1-2 % faster). Popping the empty stack reads slot 0, i.e. kDone.
template <uint32_t STRIDE = kTraceBlock>
RT_DEV void stack_push(lds_u32* stack, uint32_t* spill, LaneRay& r, uint32_t ref, bool lds_only) {
    r.sp += 1u;
    if (lds_only) stack[r.sp * STRIDE] = ref;
    else slot_store<STRIDE>(stack, spill, r.sp, ref);
}
template <uint32_t STRIDE = kTraceBlock>
RT_DEV uint32_t stack_pop(const lds_u32* stack, const uint32_t* spill, LaneRay& r, bool lds_only) {
    const uint32_t top = lds_only ? uint32_t(stack[r.sp * STRIDE]) : slot_load<STRIDE>(stack, spill, r.sp);
    r.sp -= r.sp != 0u ? 1u : 0u;
    return top;
}

// ---- compressed 4-wide tree (the default traversal) -------------------------------------------------------
// Same topology and record order as the 4-wide tree above, 64 bytes per record instead of 112: the four child boxes are
// stored as 8-bit offsets from a per-record origin `p` in units of a per-axis step `s`,
//   chunk 0 {p.x, p.y, p.z, s.x}   chunk 1 {lo.x, lo.y, lo.z, hi.x}   chunk 2 {hi.y, hi.z, s.y, s.z}   chunk 3 {ref 0..3}
// where each of the six plane words holds one byte per child. The host grows every box by 2^-17 of the scene's largest
// coordinate before rounding it outwards to the step grid, which exceeds the rounding error of the fused evaluation below
// for every ray that starts within 30 scene sizes of the origin: a box is entered whenever the reference's slab test
// (intersects.comp:103-121) enters it. Triangles are tested exactly as in the reference, so the hit found is the reference's
// hit except where another triangle lies within rounding distance of it (the order of visits decides such ties) or where
// the reference's own box test loses a hit to rounding. Half as many visits as the binary tree, and one 64-byte line per
// visit where the uncompressed 4-wide record takes two: the traversal kernels are bound by the number of scattered
// 16-byte requests, not by arithmetic.
RT_DEV void fetch_cw(const float4* __restrict__ recs, uint32_t cur, uint32_t n_resident, uint32_t lds_a, v4f& c0, v4f& c1, v4f& c2, v4f& c3) {
    const uint32_t goff = cur << 6;
    uint64_t saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_gt_u32 vcc, %[nres], %[cur]\n\t"
        "s_and_b64 exec, %[sv], vcc\n\t"
        "ds_read_b128 %[c0], %[la]\n\t"
        "ds_read_b128 %[c1], %[la] offset:%[s1]\n\t"
        "ds_read_b128 %[c2], %[la] offset:%[s2]\n\t"
        "ds_read_b128 %[c3], %[la] offset:%[s3]\n\t"
        "s_andn2_b64 exec, %[sv], vcc\n\t"
        "global_load_dwordx4 %[c0], %[go], %[base]\n\t"
        "global_load_dwordx4 %[c1], %[go], %[base] offset:16\n\t"
        "global_load_dwordx4 %[c2], %[go], %[base] offset:32\n\t"
        "global_load_dwordx4 %[c3], %[go], %[base] offset:48\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)"
        : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [sv] "=&s"(saved)
        : [nres] "s"(n_resident), [cur] "v"(cur), [la] "v"(lds_a), [go] "v"(goff), [base] "s"(recs),
          [s1] "n"(kHotLds * 16), [s2] "n"(kHotLds * 32), [s3] "n"(kHotLds * 48)
        : "vcc", "memory");
}

RT_DEV float ubyte_f32(uint32_t w, int k) { return float((w >> (8 * k)) & 0xFFu); }      // one v_cvt_f32_ubyte{k}

// Key of child K: its entry distance with the two lowest mantissa bits replaced by K (non-negative floats order like their
// bit patterns), or all ones when the child is not entered. `nx..fz` are the plane words already ordered by the ray's
// direction (near plane, far plane per axis), so no per-plane min/max is needed, and an unused entry (lo byte 255, hi byte
// 0 on every axis) comes out empty for either sign. Entered iff max(near planes, 0) <= min(far planes, t): the reference's
// lo <= hi && hi > 0 && lo < t with its two strict comparisons relaxed, which only ever enters more.
template <int K>
RT_DEV uint32_t cw_child_key(const LaneRay& r, f3 si, f3 pi, uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz) {
    const float ax = __builtin_fmaf(ubyte_f32(nx, K), si.x, pi.x), bx = __builtin_fmaf(ubyte_f32(fx, K), si.x, pi.x);
    const float ay = __builtin_fmaf(ubyte_f32(ny, K), si.y, pi.y), by = __builtin_fmaf(ubyte_f32(fy, K), si.y, pi.y);
    const float az = __builtin_fmaf(ubyte_f32(nz, K), si.z, pi.z), bz = __builtin_fmaf(ubyte_f32(fz, K), si.z, pi.z);
    const float lo = fmaxf(fmaxf(fmaxf(ax, ay), az), 0.0f);
    const float hi = fminf(fminf(fminf(bx, by), bz), r.t);
    return lo <= hi ? ((__float_as_uint(lo) & ~3u) | uint32_t(K)) : 0xFFFFFFFFu;
}

// One visit of a compressed record: the nearest entered child is visited next, the others are stacked.
template <uint32_t STRIDE = kTraceBlock>
RT_DEV void cw_step(const SceneDev& sc, LaneRay& r, lds_u32* stack, uint32_t* spill, const lds_v4f* hot_tab, uint32_t n_resident) {
    static_assert(kHotLds * 48 < 65536, "chunk offsets must fit the 16-bit offset of an LDS instruction");
    v4f c0, c1, c2, c3;
    fetch_cw(sc.cw, r.cur, n_resident, uint32_t(uintptr_t(hot_tab + r.cur)), c0, c1, c2, c3);
    const f3 si = {c0.w * r.inv.x, c2.z * r.inv.y, c2.w * r.inv.z};
    const f3 pi = {(c0.x - r.o.x) * r.inv.x, (c0.y - r.o.y) * r.inv.y, (c0.z - r.o.z) * r.inv.z};
    const uint32_t lx = __float_as_uint(c1.x), ly = __float_as_uint(c1.y), lz = __float_as_uint(c1.z);
    const uint32_t hx = __float_as_uint(c1.w), hy = __float_as_uint(c2.x), hz = __float_as_uint(c2.y);
    // the far plane comes first along a negative direction; the sign is taken from the reciprocal, which is what the products
    // below use (a direction component of -0 has a negative reciprocal)
    const bool bx = r.inv.x < 0.0f, by = r.inv.y < 0.0f, bz = r.inv.z < 0.0f;
    const uint32_t nx = bx ? hx : lx, fx = bx ? lx : hx, ny = by ? hy : ly, fy = by ? ly : hy, nz = bz ? hz : lz, fz = bz ? lz : hz;
    const uint32_t k0 = cw_child_key<0>(r, si, pi, nx, ny, nz, fx, fy, fz);
    const uint32_t k1 = cw_child_key<1>(r, si, pi, nx, ny, nz, fx, fy, fz);
    const uint32_t k2 = cw_child_key<2>(r, si, pi, nx, ny, nz, fx, fy, fz);
    const uint32_t k3 = cw_child_key<3>(r, si, pi, nx, ny, nz, fx, fy, fz);
    const uint32_t e0 = __float_as_uint(c3.x), e1 = __float_as_uint(c3.y), e2 = __float_as_uint(c3.z), e3 = __float_as_uint(c3.w);
    const uint32_t kn = min(min(k0, k1), min(k2, k3));                              // keys are distinct (their low bits differ) unless all ones
    const bool deep = __builtin_amdgcn_ballot_w64(r.sp + 4u >= kStackLds) != 0ull;    // wave-uniform
    if (k0 != 0xFFFFFFFFu && k0 != kn) stack_push<STRIDE>(stack, spill, r, e0, !deep);
    if (k1 != 0xFFFFFFFFu && k1 != kn) stack_push<STRIDE>(stack, spill, r, e1, !deep);
    if (k2 != 0xFFFFFFFFu && k2 != kn) stack_push<STRIDE>(stack, spill, r, e2, !deep);
    if (k3 != 0xFFFFFFFFu && k3 != kn) stack_push<STRIDE>(stack, spill, r, e3, !deep);
    const uint32_t in = kn & 3u;
    if (kn != 0xFFFFFFFFu) r.cur = in == 0u ? e0 : (in == 1u ? e1 : (in == 2u ? e2 : e3));
    else r.cur = stack_pop<STRIDE>(stack, spill, r, !deep);
}

// Moeller-Trumbore exactly as hit_triangle evaluates it (intersects.comp:37-64) on the 48-byte record
// {v0, n.x} {v1-v0, n.y} {v2-v0, n.z}. Returns false when the shader would have returned early; t may be NaN
// (degenerate triangle), which then loses every "closer than" comparison, as there.
RT_DEV bool triangle_test(const SceneDev& sc, f3 o, f3 d, uint32_t tri, float& t, float& u, float& v) {
    // 32-bit byte offset (the 48-byte records of at most 2^25 triangles stay below 4 GiB): scalar base + lane offset addressing
    const float4* tp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.isect) + tri * 48u);
    float4 A = tp[0], B = tp[1], C = tp[2];        // all 48 bytes requested at once: one round trip per triangle
    const f3 pn = {A.w, B.w, C.w};
    const float ndd = tv_dot(pn, d);
    const f3 ab = {B.x, B.y, B.z}, ac = {C.x, C.y, C.z};
    const f3 oo = {o.x - A.x, o.y - A.y, o.z - A.z};
    const f3 p = tv_cross(d, ac);
    const f3 q = tv_cross(oo, ab);
    const float inv_det = 1.0f / tv_dot(p, ab);
    t = tv_dot(q, ac) * inv_det;
    u = tv_dot(p, oo) * inv_det;
    v = tv_dot(q, d) * inv_det;
    float w = 1.0f - u - v;
    // the shader's chain of early returns (intersects.comp:42,52,61), evaluated without branches; every comparison is
    // written exactly as there, so NaNs fall through the same way
    bool reject = (fabsf(ndd) < kRayEps) | (t < kRayEps) | (u < 0.0f) | (v < 0.0f) | (w < 0.0f) | (u > 1.0f) | (v > 1.0f) | (w > 1.0f);
    return !reject;
}

// A new ray in lane state `r` (origin, direction and the initial "closest so far" already set): reciprocal direction, empty
// stack, the root as the first node to visit. CW: the ray is going to walk the compressed tree.
template <bool CW>
RT_DEV void start_ray(LaneRay& r, uint32_t root) {
    r.inv = f3{1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    r.u = 0.0f; r.v = 0.0f; r.prim = kNoPrim;
    r.sp = 0;
    r.cur = root == kRefNone ? kDone : root;
    if (CW) {
        // The compressed boxes are evaluated as fma(byte, s * inv, (p - o) * inv): an infinite reciprocal (a
        // direction component of exactly zero) would turn that into inf - inf = NaN, the axis would drop out of
        // the test and the ray would walk every box that overlaps it on the other two axes (measured: 2103
        // visits instead of 26 for one such camera ray, a 3 ms tail on a 6 ms launch). A huge finite reciprocal
        // keeps the arithmetic finite and still puts both planes of the axis at -huge / +huge when the origin
        // lies between them. A ray with a non-finite component hits nothing (every comparison of the
        // reference's slab test fails for it as well).
        constexpr float kBigInv = 1e30f;
        r.inv.x = copysignf(fminf(fabsf(r.inv.x), kBigInv), r.inv.x);
        r.inv.y = copysignf(fminf(fabsf(r.inv.y), kBigInv), r.inv.y);
        r.inv.z = copysignf(fminf(fabsf(r.inv.z), kBigInv), r.inv.z);
        const float probe = (r.o.x + r.o.y + r.o.z) * 0.0f + (r.d.x + r.d.y + r.d.z) * 0.0f;      // 0 for finite input, NaN otherwise
        if (!(probe == 0.0f)) r.cur = kDone;
    }
}

// One ray per lane through the compressed tree, every record read from global memory: what the persistent kernels of
// rt_kernels.h compute for the same ray (each ray visits the same records in the same order, tests the triangles of a leaf
// in storage order with the same strict "closer" test, and an occlusion query ends at the first leaf that holds a hit),
// without their scheduling across lanes. Must be called by all lanes of the wavefront together; lanes without a ray pass
// `active` == false. `r.o`, `r.d` and `r.t` are set by the caller.
template <bool ANYHIT, uint32_t STRIDE>
RT_DEV void trace_lane(const SceneDev& sc, LaneRay& r, bool active, lds_u32* stack, uint32_t* spill) {
    start_ray<true>(r, sc.wide_root);
    if (!active) r.cur = kDone;
    for (;;) {
        // interior records until every lane is at a leaf or done
        while (__builtin_amdgcn_ballot_w64(r.cur < kDone) != 0ull) {
            if (r.cur < kDone) cw_step<STRIDE>(sc, r, stack, spill, nullptr, 0u);
        }
        if (__builtin_amdgcn_ballot_w64(int32_t(r.cur) < 0) == 0ull) break;
        if (int32_t(r.cur) < 0) {
            const uint32_t first = r.cur & 0x01FFFFFFu, count = ((r.cur >> 25) & 63u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                float t, u, v;
                if (triangle_test(sc, r.o, r.d, first + k, t, u, v) && t < r.t) { r.t = t; r.u = u; r.v = v; r.prim = first + k; }
            }
            if (ANYHIT && r.prim != kNoPrim) { r.cur = kDone; r.sp = 0; }
            else r.cur = stack_pop<STRIDE>(stack, spill, r, false);
        }
    }
}

}  // namespace rtdev

#ifdef RT_FAST_ARITH
#pragma clang fp contract(on)
#endif
