// The shade stage: everything ray_color does at one path vertex (main.comp:137-209) except the two traversals, and the
// BSDF evaluated on caller-supplied inputs. This header is compiled twice, once per arithmetic mode (rt_device_math.h):
// into the library's main translation unit in the exact mode and into rt_shade_fast.hip in the fast mode; the inline
// namespace keeps the two sets of kernels apart.
#pragma once

#include "rt_kernels_common.h"
#include "rt_traverse.h"

namespace rtdev {

struct ShadeParams {
    uint32_t bounce, max_bounces, flags;
    Regions in;                    // where this bounce's paths live in `pd_in` / `hit`
    const float4* hit;             // {t, u, v, prim bits} per queue position, as extend left it
    // Survivors and light samples are appended to kQueueShards regions of `shard_cap` records each (region k takes the work
    // of block-iterations k, k + kQueueShards, ...), each with its own counter.
    PathDev out;
    uint32_t* next_shard_counts;
    ShadowDev shadow;
    uint32_t* shadow_shard_counts;
    uint32_t shard_cap;
    float4* rad;                   // per slot {radiance rgb, -}
    // k_shade<.., true> only: outputs are written at the INPUT position instead of being appended, and three records per
    // position report {alive | queued light sample << 1, light index, lobe, light-table entry},
    // {light pdf, BSDF pdf for the light direction, roughness used, -} and {throughput after absorption, -}
    uint4* probe;
};

constexpr uint32_t kLdsMats = 128;     // materials staged in LDS by k_shade<true> (12 KiB)
constexpr uint32_t kLdsLights = 256;   // emissive triangles (CDF + light table) staged in LDS by k_shade<true> (17 KiB)

// Inputs and outputs of the BSDF probe kernel (one material, frame and direction set per item).
struct BsdfIo {
    const float4* mats;   // 6 x float4 per input
    const float* eta; const float* V; const float* N; const float* L; const float* r;
    float* out_L; float* out_f; float* out_pdf;
};

// The per-path kernel (k_path_tail): which queue it may take over and where its results go.
struct TailParams {
    uint32_t bounce, max_bounces, flags;
    uint32_t* counts;              // the append counters of the queue of `bounce` (kQueueShards of them, kCounterStride apart)
    uint32_t cap;                  // records per region of that queue
    uint32_t max_paths;            // the kernel takes the queue over only when it holds at most this many paths
    float4* rad;                   // per slot {radiance rgb, -}
    uint32_t* done;                // zero at launch: counts the workgroups that have finished
    TraceStats* stats;
};
constexpr uint32_t kTailBlock = 256;

// Diagnostic builds only (tools/shade_ablation.sh): RT_SHADE_ABLATE is a bit mask of parts of the vertex the kernel leaves
// out, to measure what each costs at bounce 0: 1 = read shading record 0 instead of the hit triangle's, 2 = no radiance-slot
// update, 8 = no stores of the next path state / light sample, 16 = no light sample and no BSDF sample, 32 = a single
// wave_append, 64 = no atomics (survivors packed at the front of their wave's own input range), 128 = one constant material,
// 256 = no sky fetch on a miss, 512 = no frame / tint set-up (make_shade_point), 1024 = the atomics are performed but survivors are written
// as under 64, 2048 = no atomics, survivors written where their region would hold them if every path survived, 4096 / 8192 = every
// vertex samples the diffuse / the reflection lobe (rt_device_math.h). The pictures such builds produce are wrong.
#ifndef RT_SHADE_ABLATE
#define RT_SHADE_ABLATE 0
#endif

inline namespace RT_ARITH_NS {

// What the stage-level entry point records about a vertex (k_shade<.., true> only).
struct VertexProbe {
    int light = -1, lobe = -1, tri = -1;
    float lpdf = 0.0f, bpdf = 0.0f, rough = 0.0f;
    f3 thr = {0.0f, 0.0f, 0.0f};
};

// One path vertex: everything ray_color does between the two traversals (main.comp:137-209). In: the hit record extend left
// ({t, u, v, prim bits}), the ray that found it, the path's throughput, the pdf of the BSDF sample that produced the ray and
// the RNG state. Out: what the vertex adds to the path's radiance (`rad`), the light sample to test for visibility
// (`want_shadow`: origin + tmax, direction, contribution if unoccluded) and the path's continuation (`alive`: next ray,
// throughput, pdf). The l_* tables are the LDS copies when LDS_TABLES is set (otherwise unused, except l_srgb).
template <bool LDS_TABLES, bool PROBE>
RT_DEV void shade_vertex(const SceneDev& sc, uint32_t bounce, uint32_t max_bounces, uint32_t flags, int num_lights,
                         const lds_v4f* l_mats, const lds_v4f* l_lt, const lds_f32* l_cdf, const lds_f32* l_srgb,
                         float4 h4, f3 ro, f3 rd, f3 thr, float prev_pdf, uint32_t& seed, f3& rad,
                         bool& want_shadow, float4& sh_o, float4& sh_d, float4& sh_c,
                         bool& alive, f3& next_o, f3& next_d, f3& next_thr, float& next_pdf, VertexProbe& pr) {
    uint32_t prim = __float_as_uint(h4.w);
    if (prim == kNoPrim) {
        f4 bg = (RT_SHADE_ABLATE & 256) ? f4{1.0f, 1.0f, 1.0f, 1.0f} : sample_cube(sc.tex, rd, l_srgb);     // main.comp:137-140
        rad = f3{bg.x, bg.y, bg.z} * thr;
    } else {
        const float t = h4.x, u = h4.y, v = h4.z, w = 1.0f - h4.y - h4.z;
        // The kernel is bound by the rate at which scattered 16-byte loads return, so a vertex fetches only what
        // it uses: chunk 0 of the shading record always (normal 0, or the plane normal of a flat triangle, + material
        // word), the other normals for smooth-shaded triangles, the texture coordinates when the material binds a
        // texture, and the 48-byte geometry record only for tangent frames and the area of an emitter that was hit.
        const float4* ap = sc.shade + size_t((RT_SHADE_ABLATE & 1) ? (prim & 1023u) : prim) * 4;
        const float4 s0 = ap[0];
        const uint32_t matword = __float_as_uint(s0.w);
        Mat m;
        if (LDS_TABLES) {
            const lds_v4f* mp = l_mats + (matword & 0x3FFFFFFFu) * 6;
            if (RT_SHADE_ABLATE & 128) mp = l_mats;
            m = unpack_material(lds_load4(mp), lds_load4(mp + 1), lds_load4(mp + 2), lds_load4(mp + 3), lds_load4(mp + 4), lds_load4(mp + 5));
        } else {
            m = load_material(sc.mats, matword & 0x3FFFFFFFu);
        }
        const f3 untextured_emission = m.emission;
        const bool smooth = (matword & 0x40000000u) != 0u;
        const bool textured = (m.base_tex & m.mr_tex & m.emission_tex & m.normal_tex) != 0xFFFFFFFFu;
        const bool emitter_mis = bounce > 0 && sc.n_emissive != 0 && num_lights != 0 && luminance(untextured_emission) > 0.0f;
        const bool need_geometry = m.normal_tex != 0xFFFFFFFFu || emitter_mis;
        float4 s1 = {}, s2 = {}, s3 = {}, A = {}, B = {}, C = {};
        if (smooth || textured) s2 = ap[2];
        if (smooth) s1 = ap[1];
        if (textured) s3 = ap[3];
        if (need_geometry) {
            const float4* ip = sc.isect + size_t(prim) * 3;
            A = ip[0]; B = ip[1]; C = ip[2];
        }
        f3 ab = xyz(B), ac = xyz(C);
        f3 pos = ro + rd * t;
        float uvx = s2.z * w + s3.x * u + s3.z * v;
        float uvy = s2.w * w + s3.y * u + s3.w * v;
        f3 normal;
        if (smooth)
            normal = normalize(f3{s0.x, s0.y, s0.z} * w + f3{s1.x, s1.y, s1.z} * u + f3{s1.w, s2.x, s2.y} * v);
        else
            normal = f3{s0.x, s0.y, s0.z};               // the host stores the plane normal here for flat triangles
        // texture modulation, main.comp:141-159
        if (m.base_tex != 0xFFFFFFFFu) { f4 c = sample_2d(sc.tex, m.base_tex, uvx, uvy, l_srgb); m.base = m.base * f3{c.x, c.y, c.z}; }
        if (m.mr_tex != 0xFFFFFFFFu) { f4 c = sample_2d(sc.tex, m.mr_tex, uvx, uvy, l_srgb); m.metallic *= c.z; m.roughness *= c.y; }
        if (m.emission_tex != 0xFFFFFFFFu) { f4 c = sample_2d(sc.tex, m.emission_tex, uvx, uvy, l_srgb); m.emission = m.emission * f3{c.x, c.y, c.z}; }
        if (m.normal_tex != 0xFFFFFFFFu) {
            f4 c = sample_2d(sc.tex, m.normal_tex, uvx, uvy, l_srgb);
            f3 tn = f3{c.x, c.y, c.z} * 2.0f - mk3(1.0f);
            float d1x = s3.x - s2.z, d1y = s3.y - s2.w, d2x = s3.z - s2.z, d2y = s3.w - s2.w;  // intersects.comp:82-89
            float f = frcp(d1x * d2y - d2x * d1y);
            f3 tangent = normalize(f * (d2y * ab - d1y * ac));
            f3 bitangent = normalize(f * (-d2x * ab + d1x * ac));
            normal = normalize(tangent * tn.x + bitangent * tn.y + normal * tn.z);
        }
        f3 ffn = dot(normal, rd) <= 0.0f ? normal : -normal;      // main.comp:161
        m.roughness = fmaxf(m.roughness, 0.001f);
        float eta = dot(normal, ffn) > 0.0f ? frcp(m.ior) : m.ior;

        // emitter hit, MIS against light sampling (main.comp:170-178; MeshLightPdf disney.comp:297-330)
        float mis = 1.0f;
        if (bounce > 0 && sc.n_emissive != 0 && num_lights != 0) {
            float total_power = LDS_TABLES ? l_cdf[sc.n_emissive - 1] : sc.cdf[sc.n_emissive - 1];
            float lum = luminance(untextured_emission);
            if (total_power > 0.0f && lum > 0.0f) {
                f3 ln = cross(ab, ac);
                float area = 0.5f * length(ln);
                if (area > 0.0f) {
                    ln = ln / (2.0f * area);
                    float cos_theta = fabsf(dot(ln, rd));
                    if (cos_theta > 0.0f) {
                        float select = fdiv(float(sc.n_emissive), float(num_lights)) * fdiv(area * lum, total_power);
                        float light_pdf = fdiv(select * (t * t), area * cos_theta);
                        if (light_pdf > 0.0f) mis = power_heuristic(prev_pdf, light_pdf);
                    }
                }
            }
        }
        rad = mis * m.emission * thr;

        // "absorption" inside a transmissive medium, main.comp:182-184 (sign as upstream unless fixed)
        if (dot(normal, ffn) < 0.0f && m.spec_trans > 0.0f) {
            float sgn = (flags & RT_FIX_ABSORPTION) ? 1.0f : -1.0f;
            thr = thr * f3{fexp(sgn * flog(m.base.x) * t), fexp(sgn * flog(m.base.y) * t), fexp(sgn * flog(m.base.z) * t)};
        }

        ShadePoint pt = (RT_SHADE_ABLATE & 512) ? ShadePoint{} : make_shade_point(m, eta, -rd, ffn);
        if (PROBE) { pr.rough = m.roughness; pr.thr = thr; }

        // next-event estimation: DirectLight, disney.comp:623-720
        if (num_lights > 0 && !(RT_SHADE_ABLATE & 16)) {
            f3 surface = pos + normal * kSurfEps;
            float light_rand = pcg_next(seed);
            int light_index = int(light_rand * float(num_lights));
            if (PROBE) pr.light = light_index;
            f3 ldir, lnormal, lemit;
            float ldist, lpdf;
            bool delta = false, have = true;
            if (light_index < int(sc.n_light)) {
                const float4* lp = sc.lights + size_t(light_index) * 3;
                float4 P = lp[0], E = lp[1], Ty = lp[2];
                delta = __float_as_uint(Ty.x) == 0u;
                ldir = xyz(P) - surface;
                ldist = length(ldir);
                float d2 = ldist * ldist;
                ldir = ldir / ldist;
                lnormal = -ldir;
                lemit = xyz(E);
                lpdf = frcp(d2);
                lpdf *= float(num_lights);
            } else {
                // SampleMeshCDF (disney.comp:270-288): binary search in the un-normalised CDF
                float cdf_rand = pcg_next(seed);
                int len = int(sc.cdf_len_seen);
                auto cdf_at = [&](int k) { return uint32_t(k) < sc.n_emissive ? (LDS_TABLES ? float(l_cdf[k]) : sc.cdf[k]) : 0.0f; };
                float total = cdf_at(len - 1);
                float target = cdf_rand * total;
                int lo = 0, hi = len - 1;
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (cdf_at(mid) < target) lo = mid + 1; else hi = mid;
                }
                float prev = lo > 0 ? cdf_at(lo - 1) : 0.0f;
                float tri_prob = fdiv(cdf_at(lo) - prev, total);
                float select = fdiv(float(sc.n_emissive), float(num_lights)) * tri_prob;
                // SampleMeshLight (disney.comp:230-264) on the device light table
                float r1 = pcg_next(seed);
                float r2 = pcg_next(seed);
                if (PROBE) pr.tri = lo;
                if (uint32_t(lo) < sc.n_emissive) {
                    float4 L0, L1, L2, L3;
                    if (LDS_TABLES) {
                        const lds_v4f* lt = l_lt + uint32_t(lo) * 4;
                        L0 = lds_load4(lt); L1 = lds_load4(lt + 1); L2 = lds_load4(lt + 2); L3 = lds_load4(lt + 3);
                    } else {
                        const float4* lt = sc.light_tris + size_t(lo) * 4;
                        L0 = lt[0]; L1 = lt[1]; L2 = lt[2]; L3 = lt[3];
                    }
                    float sq1 = fsqrt(r1);
                    float bu = 1.0f - sq1, bv = r2 * sq1, bw = 1.0f - bu - bv;
                    f3 lpnt = xyz(L0) * bu + xyz(L1) * bv + xyz(L2) * bw;
                    ldir = lpnt - surface;
                    ldist = length(ldir);
                    float d2 = ldist * ldist;
                    ldir = ldir / ldist;
                    lnormal = xyz(L3);
                    lemit = f3{L1.w, L2.w, L3.w};
                    float cos_l = fabsf(dot(lnormal, -ldir));
                    float area_pdf = fdiv(d2, L0.w * fmaxf(cos_l, 0.0001f));
                    lpdf = select * area_pdf;
                } else {
                    have = false;
                }
            }
            if (have && dot(ldir, lnormal) < 0.0f) {
                float bpdf;
                f3 f = disney_eval(m, pt, ldir, bpdf);
                if (PROBE) { pr.lpdf = lpdf; pr.bpdf = bpdf; }
                if (bpdf > 0.0f) {
                    float weight = 1.0f;
                    if (!delta && lpdf > 0.0f) weight = power_heuristic(lpdf, bpdf);
                    f3 Li = weight * f * lemit / lpdf;
                    f3 contrib = Li * thr;
                    if (!(contrib.x == 0.0f && contrib.y == 0.0f && contrib.z == 0.0f)) {
                        want_shadow = true;
                        sh_o = make_float4(surface.x, surface.y, surface.z, ldist - kSurfEps);
                        sh_d = make_float4(ldir.x, ldir.y, ldir.z, 0.0f);
                        sh_c = make_float4(contrib.x, contrib.y, contrib.z, 0.0f);
                    }
                }
            }
        }

        // continuation: BSDF sample, throughput, Russian roulette, next ray (main.comp:188-209).
        // After the last bounce the shader's remaining work has no observable effect, so it is skipped.
        if ((RT_SHADE_ABLATE & 16) && bounce + 1 < max_bounces) {
            alive = true; next_o = pos; next_d = reflect3(rd, ffn); next_thr = thr * m.base; next_pdf = 1.0f;
        } else if (bounce + 1 < max_bounces) {
            float r1 = pcg_next(seed);
            float r2 = pcg_next(seed);
            f3 L;
            float pdf;
            f3 f = disney_sample(m, pt, r1, r2, L, pdf, PROBE ? &pr.lobe : nullptr);
            if (PROBE) { next_thr = thr; next_pdf = pdf; }                         // a path that ends here still reports its state
            if (pdf > 0.0f) {
                thr = thr * (f / pdf);
                bool go = true;
                if (bounce > 2) {
                    float p = fmaxf(0.05f, fminf(0.95f, fmaxf(thr.x, fmaxf(thr.y, thr.z))));
                    if (pcg_next(seed) > p) go = false;
                    else thr = thr / p;
                }
                // The next ray is written whether or not the path goes on (a path that ends here carries it nowhere): with
                // these assignments under `if (go)` hipcc 7.2 drops the one to next_d.x on the roulette branch.
                next_o = pos + L * kRayEps;
                next_d = L;
                next_thr = thr;
                next_pdf = pdf;
                alive = go;
            }
        }
    }
}

// shade: everything ray_color does at one path vertex (main.comp:137-209) except the two traversals.
// LDS_TABLES: the material table, the light CDF and the light table are small (<= kLdsMats / kLdsLights entries) and are
// copied into LDS once per workgroup, which removes three dependent global round trips (material fetch, the CDF binary
// search, the light record) from every vertex. Larger scenes use the same code reading the tables from global memory.
// PROBE: the same code additionally records the discrete decisions taken at each vertex (which light, which lobe) and the
// two pdfs of the light sample, for the stage-level entry point that checks this kernel vertex by vertex.
template <bool LDS_TABLES, bool PROBE = false>
__global__ __launch_bounds__(kBlock, RT_SHADE_WAVES) void k_shade(SceneDev sc, PathDev pd, ShadeParams sp) {
    __shared__ uint32_t s_prefix[kQueueShards + 1];
    __shared__ v4f s_mats[LDS_TABLES ? kLdsMats * 6 : 1];
    __shared__ v4f s_lt[LDS_TABLES ? kLdsLights * 4 : 1];
    __shared__ float s_cdf[LDS_TABLES ? kLdsLights : 1];
    __shared__ float s_srgb[256];
    s_srgb[threadIdx.x & 255u] = sc.tex.srgb[threadIdx.x & 255u];
    const lds_f32* l_srgb = (const lds_f32*)s_srgb;
    if (!LDS_TABLES) __syncthreads();
    if (LDS_TABLES) {
        for (uint32_t k = threadIdx.x; k < sc.n_mat * 6; k += kBlock) { float4 v = sc.mats[k]; s_mats[k] = v4f{v.x, v.y, v.z, v.w}; }
        for (uint32_t k = threadIdx.x; k < sc.n_emissive * 4; k += kBlock) { float4 v = sc.light_tris[k]; s_lt[k] = v4f{v.x, v.y, v.z, v.w}; }
        for (uint32_t k = threadIdx.x; k < sc.n_emissive; k += kBlock) s_cdf[k] = sc.cdf[k];
        __syncthreads();
    }
    const lds_v4f* l_mats = (const lds_v4f*)s_mats;
    const lds_v4f* l_lt = (const lds_v4f*)s_lt;
    const lds_f32* l_cdf = (const lds_f32*)s_cdf;
    const uint32_t n = load_region_prefix(sp.in, s_prefix);      // ends with a barrier: the tables above are complete too
    const RegionMap map{(const lds_u32*)s_prefix, sp.in.cap, sp.in.counts == nullptr};
    const uint32_t wave_base0 = (blockIdx.x * kBlock + threadIdx.x) & ~63u;
    const uint32_t stride = gridDim.x * kBlock;
    const int num_lights = int(sc.n_light + sc.n_emissive);
    for (uint32_t base = wave_base0; base < n; base += stride) {   // wave-uniform trip count
        const uint32_t i = base + lane_id();
        const bool valid = i < n;
        bool alive = false, want_shadow = false;
        uint32_t slot = 0, pin = 0, seed = 0;
        float4 sh_o = {}, sh_d = {}, sh_c = {};
        f3 next_o = {}, next_d = {}, next_thr = {};
        float next_pdf = 0.0f;
        VertexProbe pr;
        // the 64 consecutive items of this wave nearly always lie in one region: one look-up for the wave, a per-lane
        // search only for the lanes beyond the region's end
        uint32_t run_lo, run_hi, run_base;
        map.run_of(base, run_lo, run_hi, run_base);
        if (valid) {
            pin = __builtin_expect(i < run_hi, 1) ? run_base + (i - run_lo) : map(i);
            float4 h4 = sp.hit[pin], o4 = pd.o[pin], d4 = pd.d[pin], T4 = pd.t[pin];
            f3 ro = xyz(o4), rd = xyz(d4), thr = xyz(T4);
            f3 rad = {0.0f, 0.0f, 0.0f};                   // what this vertex adds to the path's radiance
            float prev_pdf = o4.w;
            seed = __float_as_uint(d4.w);
            slot = __float_as_uint(T4.w);
            shade_vertex<LDS_TABLES, PROBE>(sc, sp.bounce, sp.max_bounces, sp.flags, num_lights, l_mats, l_lt, l_cdf, l_srgb, h4, ro, rd, thr, prev_pdf,
                                            seed, rad, want_shadow, sh_o, sh_d, sh_c, alive, next_o, next_d, next_thr, next_pdf, pr);
            // The slot's radiance is touched only by vertices that add something (sky on a miss, an emitter that was hit);
            // a NaN term (0 * infinite throughput) passes the test and poisons the sum exactly as in the shader.
            if (!(RT_SHADE_ABLATE & 2) && !(rad.x == 0.0f && rad.y == 0.0f && rad.z == 0.0f)) {
                float4 a = sp.rad[slot];
                a.x += rad.x; a.y += rad.y; a.z += rad.z;
                sp.rad[slot] = a;
            }
        }
        // queue append: one atomic per wavefront and queue, spread over kQueueShards counters; the records themselves go
        // to the reserved position, so the next bounce reads them as a dense stream
        const uint32_t shard = (base / kBlock) & (kQueueShards - 1u);
        uint32_t dst = (RT_SHADE_ABLATE & (64 | 2048)) ? bits_below_lane(__ballot(alive)) : wave_append(alive, sp.next_shard_counts + shard * kCounterStride);
        uint32_t pout_ablate = 0;
        if (RT_SHADE_ABLATE & 1024) { pout_ablate = base + bits_below_lane(__ballot(alive)) + (dst == 0xFFFFFFFFu ? 1u : 0u); }
        if (RT_SHADE_ABLATE & 2048) { pout_ablate = shard * sp.shard_cap + (base / (kBlock * kQueueShards)) * kBlock + (threadIdx.x & ~63u) + dst; }
        uint32_t sdst = (RT_SHADE_ABLATE & 32) ? dst : wave_append(want_shadow, sp.shadow_shard_counts + shard * kCounterStride);
        const uint32_t pout = PROBE ? pin : ((RT_SHADE_ABLATE & (1024 | 2048)) ? pout_ablate : (RT_SHADE_ABLATE & 64) ? base + dst : shard * sp.shard_cap + dst);
        const uint32_t spos = PROBE ? pin : shard * sp.shard_cap + sdst;
        if ((alive && !(RT_SHADE_ABLATE & 8)) || (PROBE && valid)) {
            sp.out.o[pout] = make_float4(next_o.x, next_o.y, next_o.z, next_pdf);
            sp.out.d[pout] = make_float4(next_d.x, next_d.y, next_d.z, __uint_as_float(seed));
            sp.out.t[pout] = make_float4(next_thr.x, next_thr.y, next_thr.z, __uint_as_float(slot));
        }
        if (want_shadow && !(RT_SHADE_ABLATE & 8)) {
            sp.shadow.o[spos] = sh_o;
            sp.shadow.d[spos] = make_float4(sh_d.x, sh_d.y, sh_d.z, __uint_as_float(slot));
            sp.shadow.c[spos] = sh_c;
        }
        if (PROBE && valid) {
            slot = pin;
            sp.probe[3 * size_t(slot)] = make_uint4((alive ? 1u : 0u) | (want_shadow ? 2u : 0u), uint32_t(pr.light), uint32_t(pr.lobe), uint32_t(pr.tri));
            sp.probe[3 * size_t(slot) + 1] = make_uint4(__float_as_uint(pr.lpdf), __float_as_uint(pr.bpdf), __float_as_uint(pr.rough), 0u);
            sp.probe[3 * size_t(slot) + 2] = make_uint4(__float_as_uint(pr.thr.x), __float_as_uint(pr.thr.y), __float_as_uint(pr.thr.z), 0u);
        }
    }
}


// The last bounces of a short pass, one path per lane from its current vertex to its end: closest hit, vertex, shadow ray,
// next closest hit, ... without queues in between. A wavefront launch cannot finish before the longest of its rays has, which
// takes 0.1-0.2 ms however few rays it holds, and a bounce is two such launches; in a pass too small to fill the machine
// (a sample of a few 10^5 pixels) those floors are most of the time, while this kernel's critical path is the longest
// remaining *path*. Its traversal is a plain per-lane loop (no resident records, no regrouping of lanes, 3 waves per SIMD),
// so it only pays where the alternative is launch floors: the host offers it to small passes only. The kernel looks at the length of the queue of `tp.bounce` and leaves it alone unless it holds between 1
// and `tp.max_paths` paths; if it takes the queue over, its last workgroup empties the queue's counters, so the wavefront
// launches that follow in the stream find nothing to do. Every path does exactly what the wavefront kernels would have done
// with it, in the same order (trace_lane and shade_vertex are the code those kernels run; the adds to a path's radiance slot
// keep their order: vertex, then its light sample), so the image does not depend on whether the kernel took over.
// ANYHIT: no material of the scene is glass-like, a shadow ray is blocked by any hit (see k_connect).
template <bool LDS_TABLES, bool ANYHIT>
__global__ __launch_bounds__(kTailBlock) void k_path_tail(SceneDev sc, PathDev pd, TailParams tp) {
    __shared__ uint32_t s_prefix[kQueueShards + 1];
    __shared__ v4f s_mats[LDS_TABLES ? kLdsMats * 6 : 1];
    __shared__ v4f s_lt[LDS_TABLES ? kLdsLights * 4 : 1];
    __shared__ float s_cdf[LDS_TABLES ? kLdsLights : 1];
    __shared__ float s_srgb[256];
    __shared__ uint32_t s_stack[kStackLds * kTailBlock];
    __shared__ uint32_t s_last;
    const Regions in{tp.counts, tp.cap, 0u};
    const uint32_t n = load_region_prefix(in, s_prefix);
    if (n == 0u || n > tp.max_paths) return;                       // the same decision in every workgroup
    s_srgb[threadIdx.x & 255u] = sc.tex.srgb[threadIdx.x & 255u];
    if (LDS_TABLES) {
        for (uint32_t k = threadIdx.x; k < sc.n_mat * 6; k += kTailBlock) { float4 v = sc.mats[k]; s_mats[k] = v4f{v.x, v.y, v.z, v.w}; }
        for (uint32_t k = threadIdx.x; k < sc.n_emissive * 4; k += kTailBlock) { float4 v = sc.light_tris[k]; s_lt[k] = v4f{v.x, v.y, v.z, v.w}; }
        for (uint32_t k = threadIdx.x; k < sc.n_emissive; k += kTailBlock) s_cdf[k] = sc.cdf[k];
    }
    lds_u32* stack = (lds_u32*)s_stack + threadIdx.x;
    stack[0] = kDone;                                // slot 0 of this lane's stack: what a pop of the empty stack reads
    __syncthreads();
    const lds_f32* l_srgb = (const lds_f32*)s_srgb;
    const lds_v4f* l_mats = (const lds_v4f*)s_mats;
    const lds_v4f* l_lt = (const lds_v4f*)s_lt;
    const lds_f32* l_cdf = (const lds_f32*)s_cdf;
    const RegionMap map{(const lds_u32*)s_prefix, tp.cap, false};
    const int num_lights = int(sc.n_light + sc.n_emissive);
    uint32_t spill[kStackMaxWide - kStackLds + 1];
    uint32_t n_ext = 0, n_sh = 0;
    for (uint32_t base = (blockIdx.x * kTailBlock + threadIdx.x) & ~63u; base < n; base += gridDim.x * kTailBlock) {   // wave-uniform trip count
        const uint32_t i = base + lane_id();
        bool active = i < n;
        f3 ro = {}, rd = {}, thr = {};
        float prev_pdf = 0.0f;
        uint32_t seed = 0, slot = 0;
        if (active) {
            const uint32_t pin = map(i);
            const float4 o4 = pd.o[pin], d4 = pd.d[pin], T4 = pd.t[pin];
            ro = xyz(o4); rd = xyz(d4); thr = xyz(T4);
            prev_pdf = o4.w;
            seed = __float_as_uint(d4.w);
            slot = __float_as_uint(T4.w);
        }
        for (uint32_t b = tp.bounce; __builtin_amdgcn_ballot_w64(active) != 0ull; ++b) {
            // extend
            LaneRay r;
            r.o = ro; r.d = rd; r.t = INFINITY; r.item = 0;
            trace_lane<false, kTailBlock>(sc, r, active, stack, spill);
            // the vertex
            bool alive = false, want_shadow = false;
            float4 sh_o = {}, sh_d = {}, sh_c = {};
            f3 next_o = {}, next_d = {}, next_thr = {};
            float next_pdf = 0.0f;
            if (active) {
                ++n_ext;
                f3 rad = {0.0f, 0.0f, 0.0f};
                VertexProbe pr;
                shade_vertex<LDS_TABLES, false>(sc, b, tp.max_bounces, tp.flags, num_lights, l_mats, l_lt, l_cdf, l_srgb,
                                                make_float4(r.t, r.u, r.v, __uint_as_float(r.prim)), ro, rd, thr, prev_pdf, seed, rad,
                                                want_shadow, sh_o, sh_d, sh_c, alive, next_o, next_d, next_thr, next_pdf, pr);
                if (!(rad.x == 0.0f && rad.y == 0.0f && rad.z == 0.0f)) {
                    float4 a = tp.rad[slot];
                    a.x += rad.x; a.y += rad.y; a.z += rad.z;
                    tp.rad[slot] = a;
                }
            }
            // connect
            if (__builtin_amdgcn_ballot_w64(want_shadow) != 0ull) {
                LaneRay s;
                s.o = f3{sh_o.x, sh_o.y, sh_o.z}; s.d = f3{sh_d.x, sh_d.y, sh_d.z}; s.t = sh_o.w; s.item = 0;
                trace_lane<ANYHIT, kTailBlock>(sc, s, want_shadow, stack, spill);
                if (want_shadow) {
                    ++n_sh;
                    bool blocked = s.prim != kNoPrim;
                    if (!ANYHIT && blocked) blocked = (__float_as_uint(sc.shade[size_t(s.prim) * 4].w) & 0x80000000u) == 0u;
                    if (!blocked) {
                        float4 a = tp.rad[slot];
                        a.x += sh_c.x; a.y += sh_c.y; a.z += sh_c.z;
                        tp.rad[slot] = a;
                    }
                }
            }
            active = active && alive;
            ro = next_o; rd = next_d; thr = next_thr; prev_pdf = next_pdf;
        }
    }
    n_ext = wave_sum(n_ext); n_sh = wave_sum(n_sh);
    if (lane_id() == 0) {
        if (n_ext) atomicAdd(&tp.stats->extend_rays, (unsigned long long)n_ext);
        if (n_sh) atomicAdd(&tp.stats->shadow_rays, (unsigned long long)n_sh);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&tp.stats->tail_paths, (unsigned long long)n);
    // the last workgroup to finish empties the queue (every workgroup has read its length by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(tp.done, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last != 0u && threadIdx.x < kQueueShards) tp.counts[threadIdx.x * kCounterStride] = 0u;
}

__global__ void k_bsdf(BsdfIo io, uint32_t n, int do_sample) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Mat m = load_material(io.mats, i);
        ShadePoint pt = make_shade_point(m, io.eta[i], f3{io.V[3 * i], io.V[3 * i + 1], io.V[3 * i + 2]},
                                         f3{io.N[3 * i], io.N[3 * i + 1], io.N[3 * i + 2]});
        float pdf;
        f3 f;
        if (do_sample) {
            f3 L = {0, 0, 0};
            f = disney_sample(m, pt, io.r[2 * i], io.r[2 * i + 1], L, pdf);
            io.out_L[3 * i] = L.x; io.out_L[3 * i + 1] = L.y; io.out_L[3 * i + 2] = L.z;
        } else {
            f = disney_eval(m, pt, f3{io.L[3 * i], io.L[3 * i + 1], io.L[3 * i + 2]}, pdf);
        }
        io.out_f[3 * i] = f.x; io.out_f[3 * i + 1] = f.y; io.out_f[3 * i + 2] = f.z;
        io.out_pdf[i] = pdf;
    }
}
}  // inline namespace
}  // namespace rtdev

namespace rtint {
// Launchers of the fast-arithmetic instances of the two kernels above (rt_shade_fast.hip).
void launch_shade_fast(bool lds_tables, bool probe, uint32_t grid, hipStream_t st, const rtdev::SceneDev& sc, const rtdev::PathDev& pd,
                       const rtdev::ShadeParams& sp);
void launch_bsdf_fast(uint32_t grid, uint32_t block, hipStream_t st, const rtdev::BsdfIo& io, uint32_t n, int do_sample);
void launch_tail_fast(bool lds_tables, bool anyhit, uint32_t grid, hipStream_t st, const rtdev::SceneDev& sc, const rtdev::PathDev& pd,
                      const rtdev::TailParams& tp);
}  // namespace rtint
