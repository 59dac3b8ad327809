// Device-side vector algebra, RNG, texture fetch and the Disney principled BSDF for the shade stage.
//
// Two arithmetic modes, chosen per translation unit; every function below lives in an inline namespace named after the
// mode, so the two sets of device functions and kernels never share a symbol.
//   exact (default): chosen so a CPU evaluation of the reference shader can be compared value by value: IEEE binary32
//     with no FMA contraction (the translation unit is built with -ffp-contract=off), correctly rounded division and
//     square root, min/max = fminf/fmaxf, mix(x,y,a) = x*(1-a) + y*a, normalize(v) = v * (1/sqrt(v.v)), dot products
//     summed left to right. Everything but sin/cos/log/exp/pow then equals the CPU evaluation bit for bit.
//   fast (RT_FAST_ARITH, used by the shade stage unless the caller asks for exact arithmetic): the same formulas with
//     the hardware's reciprocal, reciprocal square root and square root (1 ulp each) in place of the correctly rounded
//     operations, FMA contraction within an expression, log/exp/pow through the hardware's base-2 instructions and a short-range sine/cosine.
//     Each operation is within 1-2 ulp of the exact one; results stay within 1e-5 relative of the exact mode wherever the
//     formula itself is well conditioned.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_DEV __device__ __forceinline__

#ifdef RT_FAST_ARITH
#define RT_ARITH_NS fast_arith
#else
#define RT_ARITH_NS exact_arith
#endif

#ifndef RT_SHADE_ABLATE
#define RT_SHADE_ABLATE 0
#endif

namespace rtdev {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

// ---- material in registers (wire layout: six float4, src/types/material.rs:6-30)
struct Mat {
    f3 base; f3 emission;
    float metallic, roughness, subsurface, specular_tint, sheen, sheen_tint, clearcoat, clearcoat_roughness, spec_trans, ior;
    uint32_t base_tex, mr_tex, normal_tex, emission_tex;
};

// LDS-resident copies of small scene tables are read through explicit LDS pointer types (ds_read_b128 / ds_read_b32).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) float lds_f32;

// ---- textures: RGBA8 holding sRGB values, bilinear, one mip (image.rs:307-318, sampler.rs:10-34)
struct TexView {
    const uint32_t* __restrict__ texels;   // all 2-D images back to back, one packed RGBA8 word per texel
    const uint4* __restrict__ table;       // per image {first texel, width, height, 0}
    uint32_t n_tex;
    const uint32_t* __restrict__ sky;      // 6 faces back to back
    uint32_t sky_w, sky_h;
    const float4* __restrict__ sky_f32;    // the cube map as linear fp32 radiance (rt_upload_skybox_f32), or null: use `sky`
    const float* __restrict__ srgb;        // 256-entry decode table
};

// Everything about a shading point that both the light-sampling evaluation and the BSDF sample need:
// the frame around the face-forward normal, the view vector in it, and the two tint colours.
struct ShadePoint {
    f3 T, B, N;      // Onb(N) — disney.comp:191-196
    f3 V;            // view vector, local
    f3 spec_col, sheen_col;
    float eta;
};
struct LobeWeights { float diffuse, refl, refr, coat; };

inline namespace RT_ARITH_NS {

#ifdef RT_FAST_ARITH
RT_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_DEV float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
RT_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RT_DEV float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
RT_DEV float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }          // v_log_f32 is log2
RT_DEV float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
RT_DEV float fpow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }   // x >= 0
// sin and cos of an angle in [0, 2 pi] (what every caller passes: 2 pi times a uniform number): the nearest multiple of
// pi/2 is removed in two steps (the float nearest pi/2 minus its error), then degree-7 / degree-8 polynomials on [-pi/4, pi/4].
RT_DEV void fsincos(float phi, float& s, float& c) {
    const float q = rintf(phi * 0.636619772367581343f);
    float t = __builtin_fmaf(q, -1.57079637050628662109375f, phi);
    t = __builtin_fmaf(q, 4.37113882867379116e-8f, t);
    const float t2 = t * t;
    float sp = __builtin_fmaf(t2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, t2, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(sp * t2, t, t);
    float cp = __builtin_fmaf(t2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, t2, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(cp * t2, t2, __builtin_fmaf(t2, -0.5f, 1.0f));
    const int k = int(q);
    const bool swap = (k & 1) != 0;
    const float sv = swap ? cs : sn, cv = swap ? sn : cs;
    s = (k & 2) ? -sv : sv;
    c = ((k + 1) & 2) ? -cv : cv;
}
#else
RT_DEV float frcp(float x) { return 1.0f / x; }
RT_DEV float fdiv(float a, float b) { return a / b; }
RT_DEV float fsqrt(float x) { return sqrtf(x); }
RT_DEV float frsq(float x) { return 1.0f / sqrtf(x); }
RT_DEV float flog(float x) { return logf(x); }
RT_DEV float fexp(float x) { return expf(x); }
RT_DEV float fpow(float x, float y) { return powf(x, y); }
RT_DEV void fsincos(float phi, float& s, float& c) { s = sinf(phi); c = cosf(phi); }
#endif

RT_DEV f3 mk3(float a) { return {a, a, a}; }
RT_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RT_DEV f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
#ifdef RT_FAST_ARITH
RT_DEV f3 operator/(f3 a, float s) { const float r = frcp(s); return {a.x * r, a.y * r, a.z * r}; }
#else
RT_DEV f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
#endif
RT_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
RT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 cross(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RT_DEV float length(f3 a) { return fsqrt(dot(a, a)); }
RT_DEV f3 normalize(f3 a) { float r = frsq(dot(a, a)); return a * r; }
RT_DEV float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RT_DEV f3 mix3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
RT_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
RT_DEV f3 xyz(float4 v) { return {v.x, v.y, v.z}; }

constexpr float kPi = 3.14159265358979323f;
constexpr float kInvPi = 0.31830988618379067f;
constexpr float kTwoPi = 6.28318530717958648f;
constexpr float kRayEps = 0.0001f;   // EPSILON, main.comp:5
constexpr float kSurfEps = 0.001f;   // EPS, global.comp:6

// ---- RNG: random.comp:6-12. The shader divides by the literal 4294967295.0, which as a float is 2^32.
RT_DEV float pcg_next(uint32_t& s) {
    s = s * 747796405u + 2891336453u;
    uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return float(word) * 2.3283064365386963e-10f;
}

RT_DEV float luminance(f3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }  // disney.comp:30-33
RT_DEV float power_heuristic(float a, float b) { float t = a * a; return fdiv(t, b * b + t); }    // disney.comp:185-189

RT_DEV Mat unpack_material(float4 a, float4 b, float4 c, float4 d, float4 e, float4 f) {
    Mat m;
    m.base = {a.x, a.y, a.z};
    m.emission = {b.x, b.y, b.z};
    m.metallic = c.x; m.roughness = c.y; m.subsurface = c.z;
    m.specular_tint = d.x; m.sheen = d.y; m.sheen_tint = d.z; m.clearcoat = d.w;
    m.clearcoat_roughness = e.x; m.spec_trans = e.y; m.ior = e.z;
    m.base_tex = __float_as_uint(f.x); m.mr_tex = __float_as_uint(f.y);
    m.normal_tex = __float_as_uint(f.z); m.emission_tex = __float_as_uint(f.w);
    return m;
}
RT_DEV Mat load_material(const float4* __restrict__ mats, uint32_t idx) {
    const float4* p = mats + size_t(idx) * 6;
    return unpack_material(p[0], p[1], p[2], p[3], p[4], p[5]);
}

RT_DEV float4 lds_load4(const lds_v4f* p) { v4f v = *p; return make_float4(v.x, v.y, v.z, v.w); }

// `srgb_lds`, when not null, is a copy of the decode table in LDS: three dependent table reads per texel, twelve per
// bilinear fetch, then stay off the vector-memory pipeline.
RT_DEV f4 decode_texel(const TexView& tv, const lds_f32* srgb_lds, uint32_t w) {
    if (srgb_lds) return {srgb_lds[w & 255u], srgb_lds[(w >> 8) & 255u], srgb_lds[(w >> 16) & 255u], float(w >> 24) / 255.0f};
    return {tv.srgb[w & 255u], tv.srgb[(w >> 8) & 255u], tv.srgb[(w >> 16) & 255u], float(w >> 24) / 255.0f};
}
RT_DEV f4 lerp4(f4 a, f4 b, float f) { return {mixf(a.x, b.x, f), mixf(a.y, b.y, f), mixf(a.z, b.z, f), mixf(a.w, b.w, f)}; }
RT_DEV int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
RT_DEV int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

RT_DEV f4 sample_2d(const TexView& tv, uint32_t tex, float u, float v, const lds_f32* srgb_lds = nullptr) {  // REPEAT addressing
    if (tex >= tv.n_tex) return {1.0f, 1.0f, 1.0f, 1.0f};
    uint4 e = tv.table[tex];
    int W = int(e.y), H = int(e.z);
    if (!isfinite(u)) u = 0.0f;
    if (!isfinite(v)) v = 0.0f;
    u = u - floorf(u);
    v = v - floorf(v);
    float x = u * float(W) - 0.5f, y = v * float(H) - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi(int(x0f), W), x1 = wrapi(int(x0f) + 1, W);
    int y0 = wrapi(int(y0f), H), y1 = wrapi(int(y0f) + 1, H);
    const uint32_t* base = tv.texels + e.x;
    f4 t00 = decode_texel(tv, srgb_lds, base[y0 * W + x0]), t10 = decode_texel(tv, srgb_lds, base[y0 * W + x1]);
    f4 t01 = decode_texel(tv, srgb_lds, base[y1 * W + x0]), t11 = decode_texel(tv, srgb_lds, base[y1 * W + x1]);
    return lerp4(lerp4(t00, t10, fx), lerp4(t01, t11, fx), fy);
}

// Cube fetch (main.comp:82-84): major axis picks the face (ties: z, then y, then x); clamp to the face edge.
RT_DEV f4 sample_cube(const TexView& tv, f3 d, const lds_f32* srgb_lds = nullptr) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face;
    float sc, tc, ma;
    if (az >= ax && az >= ay) { face = d.z < 0.0f ? 5 : 4; sc = d.z < 0.0f ? -d.x : d.x; tc = -d.y; ma = az; }
    else if (ay >= ax) { face = d.y < 0.0f ? 3 : 2; sc = d.x; tc = d.y < 0.0f ? -d.z : d.z; ma = ay; }
    else { face = d.x < 0.0f ? 1 : 0; sc = d.x < 0.0f ? d.z : -d.z; tc = -d.y; ma = ax; }
    float s = fdiv(0.5f * sc, ma) + 0.5f, t = fdiv(0.5f * tc, ma) + 0.5f;
    if (!isfinite(s)) s = 0.5f;
    if (!isfinite(t)) t = 0.5f;
    int W = int(tv.sky_w), H = int(tv.sky_h);
    float x = s * float(W) - 0.5f, y = t * float(H) - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = clampi(int(x0f), W), x1 = clampi(int(x0f) + 1, W);
    int y0 = clampi(int(y0f), H), y1 = clampi(int(y0f) + 1, H);
    f4 t00, t10, t01, t11;
    if (tv.sky_f32) {            // wave-uniform: linear radiance, no decode
        const float4* base = tv.sky_f32 + size_t(face) * W * H;
        const float4 a = base[y0 * W + x0], b = base[y0 * W + x1], c = base[y1 * W + x0], e = base[y1 * W + x1];
        t00 = f4{a.x, a.y, a.z, a.w}; t10 = f4{b.x, b.y, b.z, b.w}; t01 = f4{c.x, c.y, c.z, c.w}; t11 = f4{e.x, e.y, e.z, e.w};
    } else {
        const uint32_t* base = tv.sky + size_t(face) * W * H;
        t00 = decode_texel(tv, srgb_lds, base[y0 * W + x0]); t10 = decode_texel(tv, srgb_lds, base[y0 * W + x1]);
        t01 = decode_texel(tv, srgb_lds, base[y1 * W + x0]); t11 = decode_texel(tv, srgb_lds, base[y1 * W + x1]);
    }
    return lerp4(lerp4(t00, t10, fx), lerp4(t01, t11, fx), fy);
}

// ---- Disney principled BSDF (disney.comp:37-196, 357-619), isotropic subset the shader actually calls
RT_DEV float gtr1(float ndh, float a) {
    if (a >= 1.0f) return kInvPi;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * ndh * ndh;
    return fdiv(a2 - 1.0f, kPi * flog(a2) * t);
}
RT_DEV float gtr2(float ndh, float a) {
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * ndh * ndh;
    return fdiv(a2, kPi * t * t);
}
RT_DEV float smith_g(float ndv, float alpha) {
    float a = alpha * alpha;
    float b = ndv * ndv;
    return fdiv(2.0f * ndv, ndv + fsqrt(a + b - a * b));
}
RT_DEV float schlick(float u) {
    float m = clampf(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
RT_DEV float dielectric_fresnel(float cos_i, float eta) {
    float sin2_t = eta * eta * (1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return 1.0f;
    float cos_t = fsqrt(fmaxf(1.0f - sin2_t, 0.0f));
    float rs = fdiv(eta * cos_t - cos_i, eta * cos_t + cos_i);
    float rp = fdiv(eta * cos_i - cos_t, eta * cos_i + cos_t);
    return 0.5f * (rs * rs + rp * rp);
}
RT_DEV f3 sample_gtr1(float rgh, float r1) {
    float a = fmaxf(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * kTwoPi;
    float ct = fsqrt(fdiv(1.0f - fpow(a2, 1.0f - r1), 1.0f - a2));
    float st = clampf(fsqrt(1.0f - (ct * ct)), 0.0f, 1.0f);
    float sp, cp;
    fsincos(phi, sp, cp);
    return {st * cp, st * sp, ct};
}
RT_DEV f3 sample_ggx_vndf(f3 V, float rgh, float r1, float r2) {
    f3 Vh = normalize(f3{rgh * V.x, rgh * V.y, V.z});
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    f3 T1 = lensq > 0.0f ? f3{-Vh.y, Vh.x, 0.0f} * frsq(lensq) : f3{1.0f, 0.0f, 0.0f};
    f3 T2 = cross(Vh, T1);
    float r = fsqrt(r1);
    float phi = 2.0f * kPi * r2;
    float sphi, cphi;
    fsincos(phi, sphi, cphi);
    float t1 = r * cphi;
    float t2 = r * sphi;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * fsqrt(1.0f - t1 * t1) + s * t2;
    f3 Nh = t1 * T1 + t2 * T2 + fsqrt(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(f3{rgh * Nh.x, rgh * Nh.y, fmaxf(0.0f, Nh.z)});
}
RT_DEV f3 cosine_hemisphere(float r1, float r2) {
    float r = fsqrt(r1);
    float phi = kTwoPi * r2;
    float sphi, cphi;
    fsincos(phi, sphi, cphi);
    f3 d;
    d.x = r * cphi;
    d.y = r * sphi;
    d.z = fsqrt(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return d;
}
RT_DEV f3 reflect3(f3 I, f3 N) { return I - N * (2.0f * dot(N, I)); }
RT_DEV f3 refract3(f3 I, f3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f);
    return I * eta - N * (eta * d + fsqrt(k));
}

RT_DEV f3 to_local(const ShadePoint& sp, f3 v) { return {dot(v, sp.T), dot(v, sp.B), dot(v, sp.N)}; }
RT_DEV f3 to_world(const ShadePoint& sp, f3 v) { return v.x * sp.T + v.y * sp.B + v.z * sp.N; }

RT_DEV ShadePoint make_shade_point(const Mat& m, float eta, f3 Vworld, f3 N) {
    ShadePoint sp;
    f3 up = fabsf(N.z) < 0.999f ? f3{0.0f, 0.0f, 1.0f} : f3{1.0f, 0.0f, 0.0f};
    sp.T = normalize(cross(up, N));
    sp.B = cross(N, sp.T);
    sp.N = N;
    sp.V = to_local(sp, Vworld);
    sp.eta = eta;
    float lum = luminance(m.base);                                   // GetSpecColor, disney.comp:453-460
    f3 ctint = lum > 0.0f ? m.base / lum : mk3(1.0f);
    float f0 = fdiv(1.0f - eta, 1.0f + eta);
    sp.spec_col = mix3(f0 * f0 * mix3(mk3(1.0f), ctint, m.specular_tint), m.base, m.metallic);
    sp.sheen_col = mix3(mk3(1.0f), ctint, m.sheen_tint);
    return sp;
}

RT_DEV float fresnel_mix(const Mat& m, float eta, float vdh) {
    return mixf(dielectric_fresnel(vdh, eta), schlick(vdh), m.metallic);
}

RT_DEV LobeWeights lobe_weights(const Mat& m, f3 spec_col, float fresnel) {  // disney.comp:462-474
    LobeWeights w;
    w.diffuse = luminance(m.base) * (1.0f - m.metallic) * (1.0f - m.spec_trans);
    w.refl = luminance(mix3(spec_col, mk3(1.0f), fresnel));
    w.refr = (1.0f - fresnel) * (1.0f - m.metallic) * m.spec_trans * luminance(m.base);
    w.coat = m.clearcoat * (1.0f - m.metallic);
    float total = w.diffuse + w.refl + w.refr + w.coat;
#ifdef RT_FAST_ARITH
    const float inv_total = frcp(total);
    w.diffuse *= inv_total; w.refl *= inv_total; w.refr *= inv_total; w.coat *= inv_total;
#else
    w.diffuse /= total; w.refl /= total; w.refr /= total; w.coat /= total;
#endif
    return w;
}

RT_DEV f3 eval_diffuse(const Mat& m, f3 sheen_col, f3 V, f3 L, f3 H, float& pdf) {  // disney.comp:374-397
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    float FL = schlick(L.z), FV = schlick(V.z);
    float ldh = dot(L, H);
    float FH = schlick(ldh);
    float Fd90 = 0.5f + 2.0f * ldh * ldh * m.roughness;
    float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);
    float Fss90 = ldh * ldh * m.roughness;
    float Fss = mixf(1.0f, Fss90, FL) * mixf(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (frcp(L.z + V.z) - 0.5f) + 0.5f);
    f3 Fsheen = FH * m.sheen * sheen_col;
    pdf = L.z * kInvPi;
    return (kInvPi * mixf(Fd, ss, m.subsurface) * m.base + Fsheen) * (1.0f - m.metallic) * (1.0f - m.spec_trans);
}
RT_DEV f3 eval_spec_reflection(const Mat& m, float eta, f3 spec_col, f3 V, f3 L, f3 H, float& pdf) {  // :399-414
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    float FM = fresnel_mix(m, eta, dot(L, H));
    f3 F = mix3(spec_col, mk3(1.0f), FM);
    float D = gtr2(H.z, m.roughness);
    float G1 = smith_g(fabsf(V.z), m.roughness);
    float G2 = G1 * smith_g(fabsf(L.z), m.roughness);
    float vdh = dot(V, H);
    float jac = frcp(4.0f * vdh);
    pdf = fdiv(G1 * fmaxf(0.0f, vdh) * D * jac, V.z);
    return F * D * G2 / (4.0f * L.z * V.z);
}
RT_DEV f3 eval_spec_refraction(const Mat& m, float eta, f3 V, f3 L, f3 H, float& pdf) {  // :416-434
    pdf = 0.0f;
    if (L.z >= 0.0f) return mk3(0.0f);
    float vdh = dot(V, H), ldh = dot(L, H);
    float F = dielectric_fresnel(fabsf(vdh), eta);
    float D = gtr2(H.z, m.roughness);
    float denom = ldh + vdh * eta;
    denom *= denom;
    float G1 = smith_g(fabsf(V.z), m.roughness);
    float G2 = G1 * smith_g(fabsf(L.z), m.roughness);
    float jac = fdiv(fabsf(ldh), denom);
    pdf = fdiv(G1 * fmaxf(0.0f, vdh) * D * jac, V.z);
#ifdef RT_FAST_ARITH
    f3 tint = {fsqrt(m.base.x), fsqrt(m.base.y), fsqrt(m.base.z)};
#else
    f3 tint = {powf(m.base.x, 0.5f), powf(m.base.y, 0.5f), powf(m.base.z, 0.5f)};
#endif
    return tint * (1.0f - m.metallic) * m.spec_trans * (1.0f - F) * D * G2 * fabsf(vdh) * fabsf(ldh) * eta * eta /
           (denom * fabsf(L.z) * fabsf(V.z));
}
RT_DEV f3 eval_clearcoat(const Mat& m, f3 V, f3 L, f3 H, float& pdf) {  // :436-451
    pdf = 0.0f;
    if (L.z <= 0.0f) return mk3(0.0f);
    float vdh = dot(V, H);
    float FH = dielectric_fresnel(vdh, 1.0f / 1.5f);
    float F = mixf(0.04f, 1.0f, FH);
    float D = gtr1(H.z, m.clearcoat_roughness);
    float G = smith_g(L.z, m.clearcoat_roughness) * smith_g(V.z, m.clearcoat_roughness);
    float jac = frcp(4.0f * vdh);
    pdf = D * H.z * jac;
    return mk3(1.0f) * m.clearcoat * F * D * G / (4.0f * L.z * V.z);
}

// DisneyEval (disney.comp:560-619): f * |cos| and the lobe-mixture pdf for a given world-space L.
RT_DEV f3 disney_eval(const Mat& m, const ShadePoint& sp, f3 Lworld, float& out_pdf) {
    out_pdf = 0.0f;
    f3 f = mk3(0.0f);
    f3 V = sp.V;
    f3 L = to_local(sp, Lworld);
    f3 H = L.z > 0.0f ? normalize(L + V) : normalize(L + V * sp.eta);
    if (H.z < 0.0f) H = -H;
    LobeWeights w = lobe_weights(m, sp.spec_col, fresnel_mix(m, sp.eta, dot(V, H)));
    float pdf;
    if (w.diffuse > 0.0f && L.z > 0.0f) { f = f + eval_diffuse(m, sp.sheen_col, V, L, H, pdf); out_pdf += pdf * w.diffuse; }
    if (w.refl > 0.0f && L.z > 0.0f && V.z > 0.0f) { f = f + eval_spec_reflection(m, sp.eta, sp.spec_col, V, L, H, pdf); out_pdf += pdf * w.refl; }
    if (w.refr > 0.0f && L.z < 0.0f) { f = f + eval_spec_refraction(m, sp.eta, V, L, H, pdf); out_pdf += pdf * w.refr; }
    if (w.coat > 0.0f && L.z > 0.0f && V.z > 0.0f) { f = f + eval_clearcoat(m, V, L, H, pdf); out_pdf += pdf * w.coat; }
    return f * fabsf(L.z);
}

// DisneySample (disney.comp:476-558): picks a lobe with r1, samples it, returns f * |N.L|, the pdf and world-space L.
// `lobe_out`, when given, receives the lobe chosen: 0 diffuse, 1 specular reflection, 2 specular refraction, 3 clearcoat.
RT_DEV f3 disney_sample(const Mat& m, const ShadePoint& sp, float r1, float r2, f3& Lworld, float& out_pdf, int* lobe_out = nullptr) {
    out_pdf = 0.0f;
    f3 f = mk3(0.0f);
    f3 V = sp.V;
    LobeWeights w = lobe_weights(m, sp.spec_col, fresnel_mix(m, sp.eta, V.z));
    float c0 = w.diffuse, c1 = c0 + w.refl, c2 = c1 + w.refr;
    // diagnostic builds (see RT_SHADE_ABLATE in rt_shade.h): every vertex samples the diffuse lobe / the reflection lobe, so the
    // lanes of a wave never part ways here; the difference to the full kernel bounds what sorting vertices by lobe could save
    if (RT_SHADE_ABLATE & 4096) { c0 = 2.0f; }
    if (RT_SHADE_ABLATE & 8192) { c0 = -1.0f; c1 = 2.0f; }
    if (lobe_out) *lobe_out = r1 < c0 ? 0 : (r1 < c1 ? 1 : (r1 < c2 ? 2 : 3));
    f3 L;
    if (r1 < c0) {
        r1 = fdiv(r1, c0);
        L = cosine_hemisphere(r1, r2);
        f3 H = normalize(L + V);
        f = eval_diffuse(m, sp.sheen_col, V, L, H, out_pdf);
        out_pdf *= w.diffuse;
    } else {
        // The three specular lobes, each lane taking the branch the shader takes (disney.comp:499-552) with the shader's
        // arithmetic; what they share is written once so that a wavefront whose lanes picked different lobes runs it once:
        // the rescaling of r1, the GGX half-vector of reflection and refraction, the mirror direction of reflection and
        // clearcoat.
        const int lobe = r1 < c1 ? 1 : (r1 < c2 ? 2 : 3);
        const float lo = lobe == 1 ? c0 : (lobe == 2 ? c1 : c2);
        const float hi = lobe == 1 ? c1 : (lobe == 2 ? c2 : 1.0f);
        r1 = fdiv(r1 - lo, hi - lo);
        f3 H;
        if (lobe != 3) H = sample_ggx_vndf(V, m.roughness, r1, r2);
        else H = sample_gtr1(m.clearcoat_roughness, r1);
        if (H.z < 0.0f) H = -H;
        if (lobe != 2) L = normalize(reflect3(-V, H));
        else L = normalize(refract3(-V, H, sp.eta));
        if (lobe == 1) { f = eval_spec_reflection(m, sp.eta, sp.spec_col, V, L, H, out_pdf); out_pdf *= w.refl; }
        else if (lobe == 2) { f = eval_spec_refraction(m, sp.eta, V, L, H, out_pdf); out_pdf *= w.refr; }
        else { f = eval_clearcoat(m, V, L, H, out_pdf); out_pdf *= w.coat; }
    }
    Lworld = to_world(sp, L);
    return f * fabsf(dot(sp.N, Lworld));
}

}  // inline namespace
}  // namespace rtdev
