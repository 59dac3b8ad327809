// The shade stage: everything ray_color does at one path vertex (main.comp:137-209) except the two traversals, and the
// BSDF evaluated on caller-supplied inputs. This header is compiled twice, once per arithmetic mode (rt_device_math.h):
// into the library's main translation unit in the exact mode and into rt_shade_fast.hip in the fast mode; the inline
// namespace keeps the two sets of kernels apart.
#pragma once

#include "rt_kernels_common.h"

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

// Diagnostic builds only (tools/shade_ablation.sh): RT_SHADE_ABLATE is a bit mask of parts of the vertex the kernel leaves
// out, to measure what each costs at bounce 0: 1 = read shading record 0 instead of the hit triangle's, 2 = no radiance-slot
// update, 8 = no stores of the next path state / light sample, 16 = no light sample and no BSDF sample, 32 = a single
// wave_append, 64 = no atomics (survivors packed at the front of their wave's own input range), 128 = one constant material,
// 256 = no sky fetch on a miss, 512 = no frame / tint set-up (make_shade_point), 1024 = the atomics are performed but survivors are written
// as under 64, 2048 = no atomics, survivors written where their region would hold them if every path survived. The pictures such builds produce are wrong.
#ifndef RT_SHADE_ABLATE
#define RT_SHADE_ABLATE 0
#endif

inline namespace RT_ARITH_NS {

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
        int probe_light = -1, probe_lobe = -1, probe_tri = -1;
        float probe_lpdf = 0.0f, probe_bpdf = 0.0f, probe_rough = 0.0f;
        f3 probe_thr = {0.0f, 0.0f, 0.0f};
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
                const bool emitter_mis = sp.bounce > 0 && sc.n_emissive != 0 && num_lights != 0 && luminance(untextured_emission) > 0.0f;
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
                if (sp.bounce > 0 && sc.n_emissive != 0 && num_lights != 0) {
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
                    float sgn = (sp.flags & RT_FIX_ABSORPTION) ? 1.0f : -1.0f;
                    thr = thr * f3{fexp(sgn * flog(m.base.x) * t), fexp(sgn * flog(m.base.y) * t), fexp(sgn * flog(m.base.z) * t)};
                }

                ShadePoint pt = (RT_SHADE_ABLATE & 512) ? ShadePoint{} : make_shade_point(m, eta, -rd, ffn);
                if (PROBE) { probe_rough = m.roughness; probe_thr = thr; }

                // next-event estimation: DirectLight, disney.comp:623-720
                if (num_lights > 0 && !(RT_SHADE_ABLATE & 16)) {
                    f3 surface = pos + normal * kSurfEps;
                    float light_rand = pcg_next(seed);
                    int light_index = int(light_rand * float(num_lights));
                    if (PROBE) probe_light = light_index;
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
                        if (PROBE) probe_tri = lo;
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
                        if (PROBE) { probe_lpdf = lpdf; probe_bpdf = bpdf; }
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
                if ((RT_SHADE_ABLATE & 16) && sp.bounce + 1 < sp.max_bounces) {
                    alive = true; next_o = pos; next_d = reflect3(rd, ffn); next_thr = thr * m.base; next_pdf = 1.0f;
                } else if (sp.bounce + 1 < sp.max_bounces) {
                    float r1 = pcg_next(seed);
                    float r2 = pcg_next(seed);
                    f3 L;
                    float pdf;
                    f3 f = disney_sample(m, pt, r1, r2, L, pdf, PROBE ? &probe_lobe : nullptr);
                    if (PROBE) { next_thr = thr; next_pdf = pdf; }                         // a path that ends here still reports its state
                    if (pdf > 0.0f) {
                        thr = thr * (f / pdf);
                        bool go = true;
                        if (sp.bounce > 2) {
                            float p = fmaxf(0.05f, fminf(0.95f, fmaxf(thr.x, fmaxf(thr.y, thr.z))));
                            if (pcg_next(seed) > p) go = false;
                            else thr = thr / p;
                        }
                        if (PROBE) next_thr = thr;
                        if (go) {
                            alive = true;
                            next_o = pos + L * kRayEps;
                            next_d = L;
                            next_thr = thr;
                            next_pdf = pdf;
                        }
                    }
                }
            }
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
            sp.probe[3 * size_t(slot)] = make_uint4((alive ? 1u : 0u) | (want_shadow ? 2u : 0u), uint32_t(probe_light), uint32_t(probe_lobe), uint32_t(probe_tri));
            sp.probe[3 * size_t(slot) + 1] = make_uint4(__float_as_uint(probe_lpdf), __float_as_uint(probe_bpdf), __float_as_uint(probe_rough), 0u);
            sp.probe[3 * size_t(slot) + 2] = make_uint4(__float_as_uint(probe_thr.x), __float_as_uint(probe_thr.y), __float_as_uint(probe_thr.z), 0u);
        }
    }
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
}  // namespace rtint
