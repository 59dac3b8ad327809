// TEST INFRASTRUCTURE — CPU restatement of the reference renderer's hot path.
//
// This file is the parity oracle for the HIP kernels and the CPU baseline timed by bench.py. It is never
// linked into, loaded by, or called from the product library; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use it.
//
// PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path, and it cannot
// be executed here (GLSL through Vulkan from a Rust host; no Rust, no GLSL compiler, no Vulkan ICD in this
// image). The restatement is therefore pinned only by (a) known answers derived by hand from the shader
// text (tests/golden/), and (b) analytic identities checked in tests/test_oracle_*.py.
//
// It follows the reference's megakernel structure on the reference's own wire layout (32-byte nodes,
// 144-byte triangles, 96-byte materials): one loop per pixel, one path per pixel per frame. Every function
// cites the shader lines it restates (paths relative to the reference checkout, src/shaders/...).
//
// Arithmetic conventions (GLSL leaves these open; the HIP kernels adopt the same ones):
//   all math in IEEE binary32, no FMA contraction (build with -ffp-contract=off), denormals kept;
//   min/max = fminf/fmaxf; clamp(x,a,b) = min(max(x,a),b); mix(x,y,a) = x*(1-a) + y*a;
//   normalize(v) = v * (1/sqrt(v.v)); inversesqrt(x) = 1/sqrt(x); dot = ((x*x)+(y*y))+(z*z) left to right;
//   sin/cos/log/exp/pow = the C library's float functions.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/rt_api.h"   // declarations only (wire records and the stage-probe records); nothing is linked

namespace {

// ---------------------------------------------------------------- small vector algebra
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
inline vec3 V(float a) { return {a, a, a}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
inline vec3& operator/=(vec3& a, float s) { a = a / s; return a; }
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(vec3 a) { return sqrtf(dot(a, a)); }
inline vec3 normalize(vec3 a) { float r = 1.0f / sqrtf(dot(a, a)); return a * r; }
inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
inline vec3 reflect(vec3 I, vec3 N) { return I - N * (2.0f * dot(N, I)); }
inline vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V(0.0f);
    return I * eta - N * (eta * d + sqrtf(k));
}

constexpr float PI = 3.14159265358979323f;       // global.comp:1-5
constexpr float INV_PI = 0.31830988618379067f;
constexpr float TWO_PI = 6.28318530717958648f;
constexpr float EPSILON = 0.0001f;               // main.comp:5
constexpr float EPS = 0.001f;                    // global.comp:6
constexpr float BIG = 10000000.0f;               // main.comp:9 (INFINITY)
const float INF = INFINITY;

enum : uint32_t {  // mirrors the product's behaviour switches (include/rt_api.h)
    F_REF_RGBA8 = 1u << 0,
    F_REF_CDF_PAD = 1u << 1,
    F_FIX_ABSORPTION = 1u << 2,
};

// ---------------------------------------------------------------- scene
struct Image { uint32_t w = 1, h = 1; std::vector<uint8_t> px; };

struct Scene {
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_material> mats;
    std::vector<rt_triangle> tris;
    std::vector<rt_light> lights;
    std::vector<rt_emissive_tri> emissive;
    std::vector<float> cdf;       // true length
    uint32_t cdf_padded_len = 0;  // length the shader sees through the 16-byte padded binding
    std::vector<Image> textures;
    Image sky;                    // 6 faces, px.size() = 6*w*h*4
    std::vector<float> sky_f32;   // the build's fp32 cube map (6*w*h*4 floats of linear radiance, sizes in sky.w / sky.h); empty = use `sky`
    float srgb[256];
};

struct Counters { uint64_t rays = 0, shadow_rays = 0, node_pops = 0, leaf_pops = 0, tri_tests = 0,
                  shadow_node_pops = 0, shadow_leaf_pops = 0, shadow_tri_tests = 0; };

inline vec3 v3(const float* p) { return {p[0], p[1], p[2]}; }

// ---------------------------------------------------------------- RNG (random.comp:6-57)
inline float pcg(uint32_t& s) {
    s = s * 747796405u + 2891336453u;
    uint32_t word = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return float(word) / 4294967296.0f;  // the shader's literal 4294967295.0 is not representable; it is 2^32
}
inline vec2 pixel_offset(uint32_t& s) {  // random.comp:16-27 — `quadrant` restarts at 0 in every invocation
    float off = 1.0f / 2.0f;
    float x = pcg(s) * off;
    float y = pcg(s) * off;
    x += 0.0f * off;
    y += 0.0f * off;
    return {x - 0.5f, y - 0.5f};
}
inline float random_normal(uint32_t& s) {  // random.comp:30-35
    float a = sqrtf(-2.0f * logf(1.0f - pcg(s)));
    float b = 2.0f * PI * pcg(s);
    return cosf(b) * fminf(a, 1000000000.0f);
}
inline vec2 random_disc(uint32_t& s) {  // random.comp:54-57
    float x = random_normal(s);
    float y = random_normal(s);
    float r = 1.0f / sqrtf(x * x + y * y);
    float radius = pcg(s);
    return {x * r * radius, y * r * radius};
}

// ---------------------------------------------------------------- camera (generate_ray.comp:2-29, main.comp:251)
struct Ray { vec3 o, d; };

inline vec3 rot3(const float* m, vec3 v) {  // mat3(camera.rotation) * v, column-major
    return {m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
            m[2] * v.x + m[6] * v.y + m[10] * v.z};
}

inline uint32_t pixel_seed(const rt_camera& cam, uint32_t x, uint32_t y, uint32_t frame) {
    return x + cam.resolution[0] * y + frame * y * x;
}

inline Ray generate_ray(const rt_camera& cam, uint32_t px, uint32_t py, uint32_t& seed) {
    float k = cam.focus_distance / cam.focal_length;
    vec2 fp = {cam.view_port_uv[0] * k, cam.view_port_uv[1] * k};
    vec2 delta = {fp.x / float(cam.resolution[0]), fp.y / float(cam.resolution[1])};
    delta.y = delta.y * -1.0f;
    vec3 origin = vec3{-fp.x / 2.0f, fp.y / 2.0f, -cam.focus_distance} + vec3{delta.x / 2.0f, delta.y / 2.0f, 0.0f};
    vec2 j = pixel_offset(seed);
    vec3 pix = origin + vec3{delta.x * (float(px) + j.x), delta.y * (float(py) + j.y), 0.0f};
    vec2 disc = random_disc(seed);
    vec3 ap = vec3{disc.x, disc.y, 0.0f} * cam.aperture_radius;
    Ray r;
    r.d = rot3(cam.rotation, normalize(pix - ap));
    r.o = rot3(cam.rotation, ap) + v3(cam.location);
    (void)pcg(seed);  // ray.time
    return r;
}

// ---------------------------------------------------------------- intersection (intersects.comp)
struct Hit {
    bool did_hit = false;
    float t = INF;
    float u = 0, v = 0;
    uint32_t prim = 0xFFFFFFFFu;
    vec3 pos{}, normal{}, ffnormal{}, tangent{}, bitangent{};
    vec2 uv{};
    float eta = 1.0f;
    rt_material mat{};
    bool has_mat = false;
};

// hit_triangle, intersects.comp:3-92. Fills only what survives into the caller.
inline bool hit_triangle(const rt_triangle& tr, const Ray& ray, bool smooth, Hit& h) {
    vec3 a = v3(tr.v[0]), b = v3(tr.v[1]), c = v3(tr.v[2]);
    vec3 ab = b - a, ac = c - a;
    vec3 pn = normalize(cross(ab, ac));
    if (fabsf(dot(pn, ray.d)) < EPSILON) return false;
    vec3 o = ray.o - a;
    vec3 p = cross(ray.d, ac);
    vec3 q = cross(o, ab);
    float inv_det = 1.0f / dot(p, ab);
    float t = dot(q, ac) * inv_det;
    if (t < EPSILON) return false;
    float u = dot(p, o) * inv_det;
    float v = dot(q, ray.d) * inv_det;
    float w = 1.0f - u - v;
    if (u < 0.0f || v < 0.0f || w < 0.0f || u > 1.0f || v > 1.0f || w > 1.0f) return false;
    h.did_hit = true;
    h.t = t; h.u = u; h.v = v;
    h.pos = ray.o + ray.d * t;
    vec2 uv0{tr.uv[0][0], tr.uv[0][1]}, uv1{tr.uv[1][0], tr.uv[1][1]}, uv2{tr.uv[2][0], tr.uv[2][1]};
    h.uv = uv0 * w + uv1 * u + uv2 * v;
    if (smooth) h.normal = normalize(v3(tr.n[0]) * w + v3(tr.n[1]) * u + v3(tr.n[2]) * v);
    else h.normal = pn;
    vec2 d1 = uv1 - uv0, d2 = uv2 - uv0;
    float f = 1.0f / (d1.x * d2.y - d2.x * d1.y);
    h.tangent = normalize(f * (d2.y * ab - d1.y * ac));
    h.bitangent = normalize(f * (-d2.x * ab + d1.x * ac));
    return true;
}

inline float dist_aabb(const Ray& ray, const rt_bvh_node& n) {  // intersects.comp:103-121
    vec3 inv = V(1.0f) / ray.d;
    vec3 t1 = (v3(n.min) - ray.o) * inv;
    vec3 t2 = (v3(n.max) - ray.o) * inv;
    vec3 lo{fminf(t1.x, t2.x), fminf(t1.y, t2.y), fminf(t1.z, t2.z)};
    vec3 hi{fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z)};
    float tmin = fmaxf(fmaxf(lo.x, lo.y), lo.z);
    float tmax = fminf(fminf(hi.x, hi.y), hi.z);
    return (tmin <= tmax && tmax > 0.0f) ? tmin : INF;
}

// Optional per-node visit histogram (analysis aid for the BVH-layout tools; off unless a buffer is installed).
static std::atomic<uint32_t*> g_visit_hist{nullptr};

// hit_meshes, intersects.comp:124-175
Hit hit_meshes(const Scene& sc, const Ray& ray, bool shadow, Counters* cn) {
    Hit hit;
    if (sc.tris.empty()) return hit;  // empty scene: inverted root bounds, nothing can be hit
    uint32_t stack[65];
    uint32_t sp = 0;
    stack[sp++] = 0;
    uint64_t pops = 0, leafs = 0, tests = 0;
    uint32_t* const hist = g_visit_hist.load(std::memory_order_relaxed);
    while (sp > 0) {
        const uint32_t node_index = stack[--sp];
        const rt_bvh_node& node = sc.nodes[node_index];
        if (hist) __atomic_fetch_add(&hist[node_index], 1u, __ATOMIC_RELAXED);
        if (node.amt != 0) {
            ++leafs;
            for (uint32_t k = 0; k < node.amt; ++k) {
                ++tests;
                const rt_triangle& tr = sc.tris[node.idx + k];
                const rt_material& m = sc.mats[tr.material_idx];
                Hit cur;
                if (hit_triangle(tr, ray, m.shade_smooth != 0, cur) && cur.t < hit.t) {
                    cur.prim = node.idx + k;
                    cur.mat = m;
                    cur.has_mat = true;
                    hit = cur;
                }
                if (shadow && hit.has_mat && hit.mat.spec_trans > 0.5f) hit.did_hit = false;
            }
        } else {
            ++pops;
            uint32_t idx = node.idx;
            float dl = dist_aabb(ray, sc.nodes[idx]);
            float dr = dist_aabb(ray, sc.nodes[idx + 1]);
            if (dl < dr) {
                if (dr < hit.t) stack[sp++] = idx + 1;
                if (dl < hit.t) stack[sp++] = idx;
            } else {
                if (dl < hit.t) stack[sp++] = idx;
                if (dr < hit.t) stack[sp++] = idx + 1;
            }
        }
    }
    if (cn) {
        if (shadow) { cn->shadow_node_pops += pops; cn->shadow_leaf_pops += leafs; cn->shadow_tri_tests += tests; }
        else { cn->node_pops += pops; cn->leaf_pops += leafs; cn->tri_tests += tests; }
    }
    return hit;
}

// ---------------------------------------------------------------- textures
// R8G8B8A8_SRGB, bilinear, single mip, REPEAT (src/vulkan/core/image.rs:307-318, sampler.rs:10-34).
struct vec4 { float x, y, z, w; };
inline vec4 lerp4(vec4 a, vec4 b, float f) {
    return {mixf(a.x, b.x, f), mixf(a.y, b.y, f), mixf(a.z, b.z, f), mixf(a.w, b.w, f)};
}
inline vec4 texel(const Scene& sc, const uint8_t* p) {
    return {sc.srgb[p[0]], sc.srgb[p[1]], sc.srgb[p[2]], float(p[3]) / 255.0f};
}
inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
inline int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

vec4 sample2d(const Scene& sc, uint32_t tex, vec2 uv) {
    if (tex >= sc.textures.size()) return {1, 1, 1, 1};
    const Image& im = sc.textures[tex];
    float u = std::isfinite(uv.x) ? uv.x : 0.0f, v = std::isfinite(uv.y) ? uv.y : 0.0f;
    u = u - floorf(u);
    v = v - floorf(v);
    float x = u * float(im.w) - 0.5f, y = v * float(im.h) - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = wrapi(int(x0f), int(im.w)), x1 = wrapi(int(x0f) + 1, int(im.w));
    int y0 = wrapi(int(y0f), int(im.h)), y1 = wrapi(int(y0f) + 1, int(im.h));
    auto at = [&](int xx, int yy) { return texel(sc, &im.px[(size_t(yy) * im.w + xx) * 4]); };
    vec4 top = lerp4(at(x0, y0), at(x1, y0), fx);
    vec4 bot = lerp4(at(x0, y1), at(x1, y1), fx);
    return lerp4(top, bot, fy);
}

// samplerCube fetch (main.comp:82-84): face by major axis (ties: z, then y, then x), CLAMP_TO_EDGE inside the face.
vec4 sample_cube(const Scene& sc, vec3 d) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face;
    float scoord, tcoord, ma;
    if (az >= ax && az >= ay) { face = d.z < 0.0f ? 5 : 4; scoord = d.z < 0.0f ? -d.x : d.x; tcoord = -d.y; ma = az; }
    else if (ay >= ax) { face = d.y < 0.0f ? 3 : 2; scoord = d.x; tcoord = d.y < 0.0f ? -d.z : d.z; ma = ay; }
    else { face = d.x < 0.0f ? 1 : 0; scoord = d.x < 0.0f ? d.z : -d.z; tcoord = -d.y; ma = ax; }
    float s = 0.5f * scoord / ma + 0.5f, t = 0.5f * tcoord / ma + 0.5f;
    if (!std::isfinite(s)) s = 0.5f;
    if (!std::isfinite(t)) t = 0.5f;
    const Image& im = sc.sky;
    float x = s * float(im.w) - 0.5f, y = t * float(im.h) - 0.5f;
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int x0 = clampi(int(x0f), int(im.w)), x1 = clampi(int(x0f) + 1, int(im.w));
    int y0 = clampi(int(y0f), int(im.h)), y1 = clampi(int(y0f) + 1, int(im.h));
    const uint8_t* base = sc.sky_f32.empty() ? &im.px[size_t(face) * im.w * im.h * 4] : nullptr;
    const float* fbase = sc.sky_f32.empty() ? nullptr : &sc.sky_f32[size_t(face) * im.w * im.h * 4];
    auto at = [&](int xx, int yy) {
        if (fbase) { const float* q = &fbase[(size_t(yy) * im.w + xx) * 4]; return vec4{q[0], q[1], q[2], q[3]}; }    // linear radiance: no decode
        return texel(sc, &base[(size_t(yy) * im.w + xx) * 4]);
    };
    vec4 top = lerp4(at(x0, y0), at(x1, y0), fx);
    vec4 bot = lerp4(at(x0, y1), at(x1, y1), fx);
    return lerp4(top, bot, fy);
}

// ---------------------------------------------------------------- libm sensitivity probe
// GLSL leaves sin/cos/log/exp/pow to the implementation (a few ulp); two correct implementations of the shader therefore
// differ wherever a formula amplifies those last bits (a GGX peak at roughness 0.001, a cosine sample at grazing angle).
// To tell such vertices from real defects the tests re-evaluate a vertex with these five functions nudged by a few ulp and
// take the spread as that vertex's conditioning. With all biases 0 (the default, and the only setting any render uses)
// the wrappers return the library values untouched.
static int g_bias_sin = 0, g_bias_cos = 0, g_bias_other = 0;
inline float nudge(float y, int k) {
    for (; k > 0; --k) y = nextafterf(y, INFINITY);
    for (; k < 0; ++k) y = nextafterf(y, -INFINITY);
    return y;
}
inline float m_sin(float x) { return nudge(sinf(x), g_bias_sin); }
inline float m_cos(float x) { return nudge(cosf(x), g_bias_cos); }
inline float m_log(float x) { return nudge(logf(x), g_bias_other); }
inline float m_exp(float x) { return nudge(expf(x), g_bias_other); }
inline float m_pow(float x, float y) { return nudge(powf(x, y), g_bias_other); }

// ---------------------------------------------------------------- Disney BSDF (disney.comp)
inline float Luminance(vec3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }  // :30-33

inline float GTR1(float NDotH, float a) {  // :37-44
    if (a >= 1.0f) return INV_PI;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (PI * m_log(a2) * t);
}
inline vec3 SampleGTR1(float rgh, float r1, float r2) {  // :46-59
    (void)r2;
    float a = fmaxf(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * TWO_PI;
    float cosTheta = sqrtf((1.0f - m_pow(a2, 1.0f - r1)) / (1.0f - a2));
    float sinTheta = clampf(sqrtf(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
    float sinPhi = m_sin(phi), cosPhi = m_cos(phi);
    return {sinTheta * cosPhi, sinTheta * sinPhi, cosTheta};
}
inline float GTR2(float NDotH, float a) {  // :61-66
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return a2 / (PI * t * t);
}
inline vec3 SampleGGXVNDF(vec3 Vv, float rgh, float r1, float r2) {  // :82-100
    vec3 Vh = normalize(vec3{rgh * Vv.x, rgh * Vv.y, Vv.z});
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    vec3 T1 = lensq > 0.0f ? vec3{-Vh.y, Vh.x, 0.0f} * (1.0f / sqrtf(lensq)) : vec3{1, 0, 0};
    vec3 T2 = cross(Vh, T1);
    float r = sqrtf(r1);
    float phi = 2.0f * PI * r2;
    float t1 = r * m_cos(phi);
    float t2 = r * m_sin(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    vec3 Nh = t1 * T1 + t2 * T2 + sqrtf(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2)) * Vh;
    return normalize(vec3{rgh * Nh.x, rgh * Nh.y, fmaxf(0.0f, Nh.z)});
}
inline float SmithG(float NDotV, float alphaG) {  // :121-126
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return (2.0f * NDotV) / (NDotV + sqrtf(a + b - a * b));
}
inline float SchlickFresnel(float u) {  // :136-141
    float m = clampf(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
inline float DielectricFresnel(float cosThetaI, float eta) {  // :143-157
    float sinThetaTSq = eta * eta * (1.0f - cosThetaI * cosThetaI);
    if (sinThetaTSq > 1.0f) return 1.0f;
    float cosThetaT = sqrtf(fmaxf(1.0f - sinThetaTSq, 0.0f));
    float rs = (eta * cosThetaT - cosThetaI) / (eta * cosThetaT + cosThetaI);
    float rp = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
    return 0.5f * (rs * rs + rp * rp);
}
inline vec3 CosineSampleHemisphere(float r1, float r2) {  // :159-168
    float r = sqrtf(r1);
    float phi = TWO_PI * r2;
    vec3 d;
    d.x = r * m_cos(phi);
    d.y = r * m_sin(phi);
    d.z = sqrtf(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return d;
}
inline float PowerHeuristic(float a, float b) {  // :185-189
    float t = a * a;
    return t / (b * b + t);
}
inline void Onb(vec3 N, vec3& T, vec3& B) {  // :191-196
    vec3 up = fabsf(N.z) < 0.999f ? vec3{0, 0, 1} : vec3{1, 0, 0};
    T = normalize(cross(up, N));
    B = cross(N, T);
}
inline vec3 ToWorld(vec3 X, vec3 Y, vec3 Z, vec3 v) { return v.x * X + v.y * Y + v.z * Z; }  // :357-360
inline vec3 ToLocal(vec3 X, vec3 Y, vec3 Z, vec3 v) { return {dot(v, X), dot(v, Y), dot(v, Z)}; }  // :362-365

inline vec3 base(const rt_material& m) { return v3(m.base_colour); }

inline float FresnelMix(const rt_material& m, float eta, float VDotH) {  // :367-372
    float metallicFresnel = SchlickFresnel(VDotH);
    float dielectricFresnel = DielectricFresnel(VDotH, eta);
    return mixf(dielectricFresnel, metallicFresnel, m.metallic);
}
vec3 EvalDiffuse(const rt_material& m, vec3 Csheen, vec3 Vv, vec3 L, vec3 H, float& pdf) {  // :374-397
    pdf = 0.0f;
    if (L.z <= 0.0f) return V(0.0f);
    float FL = SchlickFresnel(L.z);
    float FV = SchlickFresnel(Vv.z);
    float FH = SchlickFresnel(dot(L, H));
    float Fd90 = 0.5f + 2.0f * dot(L, H) * dot(L, H) * m.roughness;
    float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);
    float Fss90 = dot(L, H) * dot(L, H) * m.roughness;
    float Fss = mixf(1.0f, Fss90, FL) * mixf(1.0f, Fss90, FV);
    float ss = 1.25f * (Fss * (1.0f / (L.z + Vv.z) - 0.5f) + 0.5f);
    vec3 Fsheen = FH * m.sheen * Csheen;
    pdf = L.z * INV_PI;
    return (INV_PI * mixf(Fd, ss, m.subsurface) * base(m) + Fsheen) * (1.0f - m.metallic) * (1.0f - m.spec_trans);
}
vec3 EvalSpecReflection(const rt_material& m, float eta, vec3 specCol, vec3 Vv, vec3 L, vec3 H, float& pdf) {  // :399-414
    pdf = 0.0f;
    if (L.z <= 0.0f) return V(0.0f);
    float FM = FresnelMix(m, eta, dot(L, H));
    vec3 F = mix(specCol, V(1.0f), FM);
    float D = GTR2(H.z, m.roughness);
    float G1 = SmithG(fabsf(Vv.z), m.roughness);
    float G2 = G1 * SmithG(fabsf(L.z), m.roughness);
    float jacobian = 1.0f / (4.0f * dot(Vv, H));
    pdf = G1 * fmaxf(0.0f, dot(Vv, H)) * D * jacobian / Vv.z;
    return F * D * G2 / (4.0f * L.z * Vv.z);
}
vec3 EvalSpecRefraction(const rt_material& m, float eta, vec3 Vv, vec3 L, vec3 H, float& pdf) {  // :416-434
    pdf = 0.0f;
    if (L.z >= 0.0f) return V(0.0f);
    float F = DielectricFresnel(fabsf(dot(Vv, H)), eta);
    float D = GTR2(H.z, m.roughness);
    float denom = dot(L, H) + dot(Vv, H) * eta;
    denom *= denom;
    float G1 = SmithG(fabsf(Vv.z), m.roughness);
    float G2 = G1 * SmithG(fabsf(L.z), m.roughness);
    float jacobian = fabsf(dot(L, H)) / denom;
    pdf = G1 * fmaxf(0.0f, dot(Vv, H)) * D * jacobian / Vv.z;
    vec3 bc = base(m);
    vec3 specColor{m_pow(bc.x, 0.5f), m_pow(bc.y, 0.5f), m_pow(bc.z, 0.5f)};
    return specColor * (1.0f - m.metallic) * m.spec_trans * (1.0f - F) * D * G2 * fabsf(dot(Vv, H)) * fabsf(dot(L, H)) *
           eta * eta / (denom * fabsf(L.z) * fabsf(Vv.z));
}
vec3 EvalClearcoat(const rt_material& m, vec3 Vv, vec3 L, vec3 H, float& pdf) {  // :436-451
    pdf = 0.0f;
    if (L.z <= 0.0f) return V(0.0f);
    float FH = DielectricFresnel(dot(Vv, H), 1.0f / 1.5f);
    float F = mixf(0.04f, 1.0f, FH);
    float D = GTR1(H.z, m.clearcoat_roughness);
    float G = SmithG(L.z, m.clearcoat_roughness) * SmithG(Vv.z, m.clearcoat_roughness);
    float jacobian = 1.0f / (4.0f * dot(Vv, H));
    pdf = D * H.z * jacobian;
    return V(1.0f) * m.clearcoat * F * D * G / (4.0f * L.z * Vv.z);
}
inline void GetSpecColor(const rt_material& m, float eta, vec3& specCol, vec3& sheenCol) {  // :453-460
    float lum = Luminance(base(m));
    vec3 ctint = lum > 0.0f ? base(m) / lum : V(1.0f);
    float F0 = (1.0f - eta) / (1.0f + eta);
    specCol = mix(F0 * F0 * mix(V(1.0f), ctint, m.specular_tint), base(m), m.metallic);
    sheenCol = mix(V(1.0f), ctint, m.sheen_tint);
}
inline void GetLobeProbabilities(const rt_material& m, vec3 specCol, float approxFresnel, float& diffuseWt,
                                 float& specReflectWt, float& specRefractWt, float& clearcoatWt) {  // :462-474
    diffuseWt = Luminance(base(m)) * (1.0f - m.metallic) * (1.0f - m.spec_trans);
    specReflectWt = Luminance(mix(specCol, V(1.0f), approxFresnel));
    specRefractWt = (1.0f - approxFresnel) * (1.0f - m.metallic) * m.spec_trans * Luminance(base(m));
    clearcoatWt = m.clearcoat * (1.0f - m.metallic);
    float totalWt = diffuseWt + specReflectWt + specRefractWt + clearcoatWt;
    diffuseWt /= totalWt;
    specReflectWt /= totalWt;
    specRefractWt /= totalWt;
    clearcoatWt /= totalWt;
}

// DisneySample, disney.comp:476-558, with the two random numbers passed in.
vec3 DisneySample(const rt_material& m, float eta, vec3 Vw, vec3 N, float r1, float r2, vec3& L, float& pdf, int* lobe = nullptr) {
    pdf = 0.0f;
    vec3 f = V(0.0f);
    vec3 T, B;
    Onb(N, T, B);
    vec3 Vv = ToLocal(T, B, N, Vw);
    vec3 specCol, sheenCol;
    GetSpecColor(m, eta, specCol, sheenCol);
    float diffuseWt, specReflectWt, specRefractWt, clearcoatWt;
    float approxFresnel = FresnelMix(m, eta, Vv.z);
    GetLobeProbabilities(m, specCol, approxFresnel, diffuseWt, specReflectWt, specRefractWt, clearcoatWt);
    float cdf0 = diffuseWt;
    float cdf1 = cdf0 + specReflectWt;
    float cdf2 = cdf1 + specRefractWt;
    if (lobe) *lobe = r1 < cdf0 ? 0 : (r1 < cdf1 ? 1 : (r1 < cdf2 ? 2 : 3));
    if (r1 < cdf0) {
        r1 /= cdf0;
        L = CosineSampleHemisphere(r1, r2);
        vec3 H = normalize(L + Vv);
        f = EvalDiffuse(m, sheenCol, Vv, L, H, pdf);
        pdf *= diffuseWt;
    } else if (r1 < cdf1) {
        r1 = (r1 - cdf0) / (cdf1 - cdf0);
        vec3 H = SampleGGXVNDF(Vv, m.roughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-Vv, H));
        f = EvalSpecReflection(m, eta, specCol, Vv, L, H, pdf);
        pdf *= specReflectWt;
    } else if (r1 < cdf2) {
        r1 = (r1 - cdf1) / (cdf2 - cdf1);
        vec3 H = SampleGGXVNDF(Vv, m.roughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(refract(-Vv, H, eta));
        f = EvalSpecRefraction(m, eta, Vv, L, H, pdf);
        pdf *= specRefractWt;
    } else {
        r1 = (r1 - cdf2) / (1.0f - cdf2);
        vec3 H = SampleGTR1(m.clearcoat_roughness, r1, r2);
        if (H.z < 0.0f) H = -H;
        L = normalize(reflect(-Vv, H));
        f = EvalClearcoat(m, Vv, L, H, pdf);
        pdf *= clearcoatWt;
    }
    L = ToWorld(T, B, N, L);
    return f * fabsf(dot(N, L));
}

// DisneyEval, disney.comp:560-619
vec3 DisneyEval(const rt_material& m, float eta, vec3 Vw, vec3 N, vec3 Lw, float& bsdfPdf) {
    bsdfPdf = 0.0f;
    vec3 f = V(0.0f);
    vec3 T, B;
    Onb(N, T, B);
    vec3 Vv = ToLocal(T, B, N, Vw);
    vec3 L = ToLocal(T, B, N, Lw);
    vec3 H;
    if (L.z > 0.0f) H = normalize(L + Vv);
    else H = normalize(L + Vv * eta);
    if (H.z < 0.0f) H = -H;
    vec3 specCol, sheenCol;
    GetSpecColor(m, eta, specCol, sheenCol);
    float diffuseWt, specReflectWt, specRefractWt, clearcoatWt;
    float fresnel = FresnelMix(m, eta, dot(Vv, H));
    GetLobeProbabilities(m, specCol, fresnel, diffuseWt, specReflectWt, specRefractWt, clearcoatWt);
    float pdf;
    if (diffuseWt > 0.0f && L.z > 0.0f) {
        f += EvalDiffuse(m, sheenCol, Vv, L, H, pdf);
        bsdfPdf += pdf * diffuseWt;
    }
    if (specReflectWt > 0.0f && L.z > 0.0f && Vv.z > 0.0f) {
        f += EvalSpecReflection(m, eta, specCol, Vv, L, H, pdf);
        bsdfPdf += pdf * specReflectWt;
    }
    if (specRefractWt > 0.0f && L.z < 0.0f) {
        f += EvalSpecRefraction(m, eta, Vv, L, H, pdf);
        bsdfPdf += pdf * specRefractWt;
    }
    if (clearcoatWt > 0.0f && L.z > 0.0f && Vv.z > 0.0f) {
        f += EvalClearcoat(m, Vv, L, H, pdf);
        bsdfPdf += pdf * clearcoatWt;
    }
    return f * fabsf(L.z);
}

// ---------------------------------------------------------------- lights (disney.comp:216-340, 623-720)
inline uint32_t cdf_len_seen(const Scene& sc, uint32_t flags) {
    if (sc.cdf.empty()) return 1;  // empty sections are bound with a 4-byte dummy range
    return (flags & F_REF_CDF_PAD) ? sc.cdf_padded_len : uint32_t(sc.cdf.size());
}
inline float cdf_at(const Scene& sc, uint32_t i) { return i < sc.cdf.size() ? sc.cdf[i] : 0.0f; }  // padding reads 0

int SampleMeshCDF(const Scene& sc, uint32_t flags, float r, float& triProb) {  // :270-288
    int n = int(cdf_len_seen(sc, flags));
    float total = cdf_at(sc, uint32_t(n - 1));
    float target = r * total;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cdf_at(sc, uint32_t(mid)) < target) lo = mid + 1;
        else hi = mid;
    }
    float prev = lo > 0 ? cdf_at(sc, uint32_t(lo - 1)) : 0.0f;
    triProb = (cdf_at(sc, uint32_t(lo)) - prev) / total;
    return lo;
}

float MeshLightPdf(const Scene& sc, const Hit& hit, vec3 rayDir, int numOfLights) {  // :297-330
    int num_mesh = int(sc.emissive.size());
    if (num_mesh == 0 || numOfLights == 0) return 0.0f;
    float total_power = cdf_at(sc, uint32_t(num_mesh - 1));
    if (total_power <= 0.0f) return 0.0f;
    const rt_triangle& tr = sc.tris[hit.prim];
    vec3 emission = v3(sc.mats[tr.material_idx].emission);
    float lum = Luminance(emission);
    if (lum <= 0.0f) return 0.0f;
    vec3 e1 = v3(tr.v[1]) - v3(tr.v[0]);
    vec3 e2 = v3(tr.v[2]) - v3(tr.v[0]);
    vec3 ln = cross(e1, e2);
    float area = 0.5f * length(ln);
    if (area <= 0.0f) return 0.0f;
    ln /= (2.0f * area);
    float cosTheta = fabsf(dot(ln, rayDir));
    if (cosTheta <= 0.0f) return 0.0f;
    float selectProb = (float(num_mesh) / float(numOfLights)) * ((area * lum) / total_power);
    float dist = hit.t;
    return selectProb * (dist * dist) / (area * cosTheta);
}

struct LightSample { vec3 normal{}, emission{}, direction{}; float dist = 0, pdf = 0; };

// Everything DirectLight (disney.comp:623-720) does except the visibility test: which light, which point on it, the
// shadow ray, and the contribution the sample makes if it turns out to be unoccluded. The shader traces the shadow ray
// first and evaluates the BSDF afterwards; no random number is drawn in between, so the order does not show.
struct LightPick {
    int light_index = -1;      // light_rand * numOfLights, -1 when the scene has no lights (no number drawn)
    int tri = -1;              // entry of the emissive-triangle table (mesh lights only)
    bool traced = false;       // the light faces the surface: the shader traces a shadow ray (disney.comp:697)
    vec3 origin{}, direction{};
    float dist = 0.0f;         // distance to the sampled point; occluded means a blocker nearer than dist - EPS
    float light_pdf = 0.0f, bsdf_pdf = 0.0f;
    vec3 Li{};                 // weight * f * Le / pdf if unoccluded (zero when the BSDF pdf is not positive)
};

LightPick SampleDirect(const Scene& sc, uint32_t flags, const Ray& r, const Hit& rec, uint32_t& seed, int numOfLights) {
    LightPick lp;
    vec3 surfacePos = rec.pos + rec.normal * EPS;
    int num_point = int(sc.lights.size());
    int num_mesh = int(sc.emissive.size());
    bool isDelta = false;
    if (numOfLights <= 0) return lp;
    LightSample ls;
    float light_rand = pcg(seed);
    int light_index = int(light_rand * float(numOfLights));
    lp.light_index = light_index;
    if (light_index < num_point) {
        const rt_light& lt = sc.lights[size_t(light_index)];
        isDelta = (int(lt.type) == 0);
        ls.direction = v3(lt.position) - surfacePos;  // SamplePointLight :216-225
        ls.dist = length(ls.direction);
        float distSq = ls.dist * ls.dist;
        ls.direction /= ls.dist;
        ls.normal = -ls.direction;
        ls.emission = v3(lt.emission);
        ls.pdf = 1.0f / distSq;
        ls.pdf *= float(numOfLights);
    } else {
        float cdf_rand = pcg(seed);
        float triProb;
        int tri_idx = SampleMeshCDF(sc, flags, cdf_rand, triProb);
        lp.tri = tri_idx;
        float selectProb = (float(num_mesh) / float(numOfLights)) * triProb;
        // reading past the emissive table (possible only with the padded-CDF quirk) yields zeros
        rt_emissive_tri et{0, 0, 0.0f, 0.0f};
        if (size_t(tri_idx) < sc.emissive.size()) et = sc.emissive[size_t(tri_idx)];
        const rt_triangle& tr = sc.tris[et.tri_index];  // SampleMeshLight :230-264
        float r1 = pcg(seed);
        float r2 = pcg(seed);
        float sq = sqrtf(r1);
        float u = 1.0f - sq;
        float v = r2 * sq;
        float w = 1.0f - u - v;
        vec3 lpnt = v3(tr.v[0]) * u + v3(tr.v[1]) * v + v3(tr.v[2]) * w;
        ls.direction = lpnt - surfacePos;
        ls.dist = length(ls.direction);
        float distSq = ls.dist * ls.dist;
        ls.direction /= ls.dist;
        vec3 e1 = v3(tr.v[1]) - v3(tr.v[0]);
        vec3 e2 = v3(tr.v[2]) - v3(tr.v[0]);
        ls.normal = normalize(cross(e1, e2));
        ls.emission = v3(sc.mats[tr.material_idx].emission);
        float cos_theta = fabsf(dot(ls.normal, -ls.direction));
        float areaPdf = distSq / (et.area * fmaxf(cos_theta, 0.0001f));
        ls.pdf = selectProb * areaPdf;
    }
    lp.origin = surfacePos;
    lp.direction = ls.direction;
    lp.dist = ls.dist;
    lp.light_pdf = ls.pdf;
    if (dot(ls.direction, ls.normal) < 0.0f) {
        lp.traced = true;
        float bpdf;
        vec3 f = DisneyEval(rec.mat, rec.eta, -r.d, rec.ffnormal, ls.direction, bpdf);
        lp.bsdf_pdf = bpdf;
        float weight = 1.0f;
        if (!isDelta && ls.pdf > 0.0f) weight = PowerHeuristic(ls.pdf, bpdf);
        if (bpdf > 0.0f) lp.Li = weight * f * ls.emission / ls.pdf;
    }
    return lp;
}

// ---------------------------------------------------------------- path (main.comp:122-213)
// State of a path when it arrives at a vertex: the ray of this bounce, what its closest-hit search found, and what the
// path carries (throughput, pdf of the BSDF sample that produced the ray, RNG state).
struct VertexState { Ray ray; Hit hit; vec3 throughput; float bsdfPdf; uint32_t seed; };

// Everything ray_color does at one vertex (main.comp:137-209) apart from tracing rays.
struct VertexResult {
    vec3 emitted{};            // added to the radiance before the light sample: sky on a miss, MIS-weighted emission on a hit
    LightPick light;           // the light sample; its contribution is light.Li * light_throughput when unoccluded
    vec3 light_throughput{};   // throughput after the absorption step (main.comp:182-184)
    bool alive = false;        // the path continues along `next`
    Ray next{};
    vec3 throughput{};         // after the BSDF sample and Russian roulette
    float pdf = 0.0f;          // bsdfPdf carried to the next vertex
    uint32_t seed = 0;         // RNG state after this vertex's draws
    int lobe = -1;             // 0 diffuse, 1 specular reflection, 2 specular refraction, 3 clearcoat; -1 on a miss
    float roughness = 0.0f;    // roughness the BSDF used (after texture modulation and the 0.001 clamp)
};

VertexResult shade_vertex(const Scene& sc, uint32_t flags, uint32_t bounce, VertexState st) {
    VertexResult out;
    const Ray& ray = st.ray;
    Hit& hit = st.hit;
    vec3 throughput = st.throughput;
    uint32_t seed = st.seed;
    int numOfLights = int(sc.lights.size() + sc.emissive.size());
    out.throughput = throughput; out.pdf = st.bsdfPdf; out.seed = seed; out.light_throughput = throughput;
    if (!hit.did_hit) {
        vec4 bg = sample_cube(sc, ray.d);
        out.emitted = vec3{bg.x, bg.y, bg.z} * throughput;
        return out;
    }
    rt_material& m = hit.mat;
    vec3 emission = v3(m.emission);
    if (m.base_color_tex != RT_NO_TEXTURE) {
        vec4 t = sample2d(sc, m.base_color_tex, hit.uv);
        m.base_colour[0] *= t.x; m.base_colour[1] *= t.y; m.base_colour[2] *= t.z; m.base_colour[3] *= t.w;
    }
    if (m.metallic_roughness_tex != RT_NO_TEXTURE) {
        vec4 t = sample2d(sc, m.metallic_roughness_tex, hit.uv);
        m.metallic *= t.z;
        m.roughness *= t.y;
    }
    if (m.emission_tex != RT_NO_TEXTURE) {
        vec4 t = sample2d(sc, m.emission_tex, hit.uv);
        emission = emission * vec3{t.x, t.y, t.z};
        m.emission[0] = emission.x; m.emission[1] = emission.y; m.emission[2] = emission.z;
    }
    if (m.normal_tex != RT_NO_TEXTURE) {
        vec4 t = sample2d(sc, m.normal_tex, hit.uv);
        vec3 tn = vec3{t.x, t.y, t.z} * 2.0f - V(1.0f);
        hit.normal = normalize(hit.tangent * tn.x + hit.bitangent * tn.y + hit.normal * tn.z);
    }
    hit.ffnormal = dot(hit.normal, ray.d) <= 0.0f ? hit.normal : -hit.normal;
    m.roughness = fmaxf(m.roughness, 0.001f);
    out.roughness = m.roughness;
    hit.eta = dot(hit.normal, hit.ffnormal) > 0.0f ? (1.0f / m.ior) : m.ior;

    float misWeight = 1.0f;
    if (bounce > 0) {
        float lightPdf = MeshLightPdf(sc, hit, ray.d, numOfLights);
        if (lightPdf > 0.0f) misWeight = PowerHeuristic(st.bsdfPdf, lightPdf);
    }
    out.emitted = misWeight * emission * throughput;

    if (dot(hit.normal, hit.ffnormal) < 0.0f && m.spec_trans > 0.0f) {
        vec3 bc = base(m);
        if (flags & F_FIX_ABSORPTION)
            throughput *= vec3{m_exp(m_log(bc.x) * hit.t), m_exp(m_log(bc.y) * hit.t), m_exp(m_log(bc.z) * hit.t)};
        else
            throughput *= vec3{m_exp(-m_log(bc.x) * hit.t), m_exp(-m_log(bc.y) * hit.t), m_exp(-m_log(bc.z) * hit.t)};
    }
    out.light_throughput = throughput;
    out.light = SampleDirect(sc, flags, ray, hit, seed, numOfLights);

    vec3 L;
    float pdf;
    float r1 = pcg(seed);
    float r2 = pcg(seed);
    vec3 f = DisneySample(m, hit.eta, -ray.d, hit.ffnormal, r1, r2, L, pdf, &out.lobe);
    out.seed = seed;
    out.throughput = throughput;
    out.pdf = pdf;
    if (!(pdf > 0.0f)) return out;
    throughput *= f / pdf;
    out.throughput = throughput;
    if (bounce > 2) {
        float p = fmaxf(0.05f, fminf(0.95f, fmaxf(throughput.x, fmaxf(throughput.y, throughput.z))));
        bool kill = pcg(seed) > p;
        out.seed = seed;
        if (kill) return out;
        throughput /= p;
        out.throughput = throughput;
    }
    out.alive = true;
    out.next.d = L;
    out.next.o = hit.pos + L * EPSILON;
    return out;
}

// Optional tap: the state of the path on arrival at vertex `bounce` (left untouched when the path ends earlier).
struct PathTap { uint32_t bounce; bool reached; VertexState state; };

vec3 ray_color(const Scene& sc, uint32_t flags, uint32_t max_bounces, Ray ray, uint32_t& seed, Counters* cn, PathTap* tap = nullptr) {
    vec3 radiance = V(0.0f), throughput = V(1.0f);
    float bsdfPdf = 0.0f;
    for (uint32_t bounce = 0; bounce < max_bounces; ++bounce) {
        if (cn) cn->rays++;
        Hit hit = hit_meshes(sc, ray, false, cn);
        VertexState st{ray, hit, throughput, bsdfPdf, seed};
        if (tap && tap->bounce == bounce) { tap->reached = true; tap->state = st; }
        VertexResult vr = shade_vertex(sc, flags, bounce, st);
        seed = vr.seed;
        radiance += vr.emitted;
        if (!hit.did_hit) break;
        // DirectLight (disney.comp:697-713): the sample counts unless something opaque lies in front of the light
        vec3 Ld = V(0.0f);
        if (vr.light.traced) {
            Ray sr{vr.light.origin, vr.light.direction};
            if (cn) cn->shadow_rays++;
            Hit sh = hit_meshes(sc, sr, true, cn);
            bool inShadow = sh.did_hit && sh.t < vr.light.dist - EPS;
            if (!inShadow) Ld = vr.light.Li;
        }
        radiance += Ld * vr.light_throughput;
        if (!vr.alive) break;
        throughput = vr.throughput;
        bsdfPdf = vr.pdf;
        ray = vr.next;
    }
    return radiance;
}

// ---------------------------------------------------------------- output (main.comp:114-120, 215-244, 260-267)
inline vec3 suppress_fireflies(vec3 c, float maxv) {
    float lum = Luminance(c);
    if (lum > maxv) c *= maxv / lum;
    return c;
}
inline vec3 filmic(vec3 c) {
    c *= 1.5f;
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    auto curve = [&](float x) { return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F; };
    vec3 r{curve(c.x), curve(c.y), curve(c.z)};
    const float white = 11.2f;
    r /= curve(white);
    float l = dot(r, vec3{0.2126f, 0.7152f, 0.0722f});
    r = mix(V(l), r, 1.7f);
    float g = 1.0f / 2.4f;
    return {powf(fmaxf(r.x, 0.0f), g), powf(fmaxf(r.y, 0.0f), g), powf(fmaxf(r.z, 0.0f), g)};
}
inline uint8_t unorm8(float x) {
    if (!(x > 0.0f)) return 0;  // NaN and negatives store 0
    if (x >= 1.0f) return 255;
    return uint8_t(floorf(x * 255.0f + 0.5f));
}

// Worker threads when the caller does not say: the machine's, but at most 16 (a GPU box gives a job 16 cores of a much
// larger host, and hardware_concurrency reports the whole host).
inline uint32_t default_threads() { return std::min(16u, std::max(1u, std::thread::hardware_concurrency())); }

template <class Fn>
void parallel_rows(uint32_t rows, uint32_t threads, Fn fn) {
    if (threads == 0) threads = default_threads();
    std::atomic<uint32_t> next{0};
    auto body = [&](uint32_t tid) {
        for (;;) {
            uint32_t r0 = next.fetch_add(4);
            if (r0 >= rows) break;
            for (uint32_t r = r0; r < std::min(rows, r0 + 4); ++r) fn(r, tid);
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(body, t);
    body(0);
    for (auto& th : pool) th.join();
}

}  // namespace

// =================================================================== C interface (ctypes)
extern "C" {

struct orc_counters { uint64_t rays, shadow_rays, node_pops, leaf_pops, tri_tests, shadow_node_pops, shadow_leaf_pops, shadow_tri_tests; };

void* orc_scene_create(const void* blob, const uint64_t sizes[6], uint32_t n_bvh, uint32_t n_mat, uint32_t n_tri,
                       uint32_t n_light, uint32_t n_emissive) {
    Scene* sc = new Scene;
    const uint8_t* p = static_cast<const uint8_t*>(blob);
    sc->nodes.assign(reinterpret_cast<const rt_bvh_node*>(p), reinterpret_cast<const rt_bvh_node*>(p) + n_bvh); p += sizes[0];
    sc->mats.assign(reinterpret_cast<const rt_material*>(p), reinterpret_cast<const rt_material*>(p) + n_mat); p += sizes[1];
    sc->tris.assign(reinterpret_cast<const rt_triangle*>(p), reinterpret_cast<const rt_triangle*>(p) + n_tri); p += sizes[2];
    sc->lights.assign(reinterpret_cast<const rt_light*>(p), reinterpret_cast<const rt_light*>(p) + n_light); p += sizes[3];
    sc->emissive.assign(reinterpret_cast<const rt_emissive_tri*>(p), reinterpret_cast<const rt_emissive_tri*>(p) + n_emissive); p += sizes[4];
    sc->cdf.assign(reinterpret_cast<const float*>(p), reinterpret_cast<const float*>(p) + n_emissive);
    sc->cdf_padded_len = uint32_t(sizes[5] / 4);
    sc->sky.w = sc->sky.h = 1;
    sc->sky.px.assign(6 * 4, 255);  // 1x1 white cube (src/scene/mod.rs:49-65)
    for (int i = 0; i < 256; ++i) {
        double c = i / 255.0;
        sc->srgb[i] = float(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
    }
    return sc;
}
void orc_scene_destroy(void* s) { delete static_cast<Scene*>(s); }

// Nudges the results of sin / cos / {log, exp, pow} inside the shading code by that many ulp (see the sensitivity probe
// above). Not thread-safe and meant for orc_shade_vertices only; (0, 0, 0) restores the library functions.
void orc_set_libm_bias(int sin_ulp, int cos_ulp, int other_ulp) { g_bias_sin = sin_ulp; g_bias_cos = cos_ulp; g_bias_other = other_ulp; }

// Installs (or, with nullptr, removes) a caller-owned array of one counter per reference-layout BVH node; every node
// popped by hit_meshes increments its counter.
void orc_set_visit_histogram(uint32_t* per_node) { g_visit_hist.store(per_node); }

void orc_scene_set_textures(void* s, const rt_texture* tex, uint32_t n, const rt_texture* sky) {
    Scene* sc = static_cast<Scene*>(s);
    sc->textures.clear();
    for (uint32_t i = 0; i < n; ++i) {
        Image im;
        im.w = tex[i].width; im.h = tex[i].height;
        im.px.assign(tex[i].pixels, tex[i].pixels + size_t(im.w) * im.h * 4);
        sc->textures.push_back(std::move(im));
    }
    if (sky) {
        sc->sky.w = sky->width; sc->sky.h = sky->height;
        sc->sky.px.assign(sky->pixels, sky->pixels + size_t(sky->width) * sky->height * 4 * 6);
    }
    sc->sky_f32.clear();
}

// The build's extension (rt_upload_skybox_f32): same addressing and filtering as the 8-bit cube, texels are linear floats.
void orc_scene_set_skybox_f32(void* s, uint32_t w, uint32_t h, const float* rgba) {
    Scene* sc = static_cast<Scene*>(s);
    sc->sky.w = w; sc->sky.h = h;
    sc->sky_f32.assign(rgba, rgba + size_t(w) * h * 4 * 6);
}

float orc_pcg(uint32_t* state) { return pcg(*state); }
uint32_t orc_pcg_word(uint32_t* state) {
    *state = *state * 747796405u + 2891336453u;
    return ((*state >> ((*state >> 28u) + 4u)) ^ *state) * 277803737u;
}

void orc_generate_rays(const rt_camera* cam, uint32_t frame, float* origins, float* dirs, uint32_t* seeds) {
    uint32_t W = cam->resolution[0], H = cam->resolution[1];
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            uint32_t seed = pixel_seed(*cam, x, y, frame);
            Ray r = generate_ray(*cam, x, y, seed);
            size_t i = size_t(y) * W + x;
            origins[i * 3] = r.o.x; origins[i * 3 + 1] = r.o.y; origins[i * 3 + 2] = r.o.z;
            dirs[i * 3] = r.d.x; dirs[i * 3 + 1] = r.d.y; dirs[i * 3 + 2] = r.d.z;
            if (seeds) seeds[i] = seed;
        }
}

// hits: n x {t,u,v,prim}
void orc_trace_closest(const void* s, uint32_t n, const float* o, const float* d, void* hits_out, orc_counters* cnt, uint32_t threads) {
    const Scene& sc = *static_cast<const Scene*>(s);
    struct H4 { float t, u, v; uint32_t prim; };
    H4* out = static_cast<H4*>(hits_out);
    uint32_t rows = (n + 1023) / 1024;
    std::vector<Counters> local(threads ? threads : default_threads());
    parallel_rows(rows, uint32_t(local.size()), [&](uint32_t r, uint32_t tid) {
        for (uint32_t i = r * 1024; i < std::min(n, (r + 1) * 1024); ++i) {
            Ray ray{v3(o + 3 * size_t(i)), v3(d + 3 * size_t(i))};
            local[tid].rays++;
            Hit h = hit_meshes(sc, ray, false, &local[tid]);
            if (h.did_hit) out[i] = {h.t, h.u, h.v, h.prim};
            else out[i] = {INF, 0.0f, 0.0f, 0xFFFFFFFFu};
        }
    });
    if (cnt) {
        std::memset(cnt, 0, sizeof(*cnt));
        for (auto& c : local) { cnt->rays += c.rays; cnt->node_pops += c.node_pops; cnt->leaf_pops += c.leaf_pops; cnt->tri_tests += c.tri_tests; }
    }
}

// hit_triangle (intersects.comp:3-92) of ray i against triangle prim[i] alone: what the shader computes for that pair,
// whatever the traversal. Misses (and prim = 0xFFFFFFFF) report {inf, 0, 0, 0xFFFFFFFF}.
void orc_test_triangles(const void* s, uint32_t n, const float* o, const float* d, const uint32_t* prim, void* hits_out) {
    const Scene& sc = *static_cast<const Scene*>(s);
    struct H4 { float t, u, v; uint32_t prim; };
    H4* out = static_cast<H4*>(hits_out);
    for (uint32_t i = 0; i < n; ++i) {
        out[i] = {INF, 0.0f, 0.0f, 0xFFFFFFFFu};
        if (prim[i] >= sc.tris.size()) continue;
        Ray ray{v3(o + 3 * size_t(i)), v3(d + 3 * size_t(i))};
        Hit h;
        if (hit_triangle(sc.tris[prim[i]], ray, false, h)) out[i] = {h.t, h.u, h.v, prim[i]};
    }
}

void orc_trace_shadow(const void* s, uint32_t n, const float* o, const float* d, const float* tmax, uint8_t* visible) {
    const Scene& sc = *static_cast<const Scene*>(s);
    for (uint32_t i = 0; i < n; ++i) {
        Ray ray{v3(o + 3 * size_t(i)), v3(d + 3 * size_t(i))};
        Hit h = hit_meshes(sc, ray, true, nullptr);
        visible[i] = (h.did_hit && h.t < tmax[i]) ? 0 : 1;
    }
}

void orc_eval_bsdf(uint32_t n, const rt_material* mats, const float* eta, const float* Vw, const float* N, const float* L,
                   float* out_f, float* out_pdf) {
    for (uint32_t i = 0; i < n; ++i) {
        float pdf;
        vec3 f = DisneyEval(mats[i], eta[i], v3(Vw + 3 * i), v3(N + 3 * i), v3(L + 3 * i), pdf);
        out_f[3 * i] = f.x; out_f[3 * i + 1] = f.y; out_f[3 * i + 2] = f.z;
        out_pdf[i] = pdf;
    }
}
void orc_sample_bsdf(uint32_t n, const rt_material* mats, const float* eta, const float* Vw, const float* N, const float* r,
                     float* out_L, float* out_f, float* out_pdf) {
    for (uint32_t i = 0; i < n; ++i) {
        float pdf;
        vec3 L{0, 0, 0};
        vec3 f = DisneySample(mats[i], eta[i], v3(Vw + 3 * i), v3(N + 3 * i), r[2 * i], r[2 * i + 1], L, pdf);
        out_L[3 * i] = L.x; out_L[3 * i + 1] = L.y; out_L[3 * i + 2] = L.z;
        out_f[3 * i] = f.x; out_f[3 * i + 1] = f.y; out_f[3 * i + 2] = f.z;
        out_pdf[i] = pdf;
    }
}
void orc_sample_texture(const void* s, uint32_t tex, uint32_t n, const float* uvw, float* rgba) {
    const Scene& sc = *static_cast<const Scene*>(s);
    for (uint32_t i = 0; i < n; ++i) {
        vec4 c = tex == 0xFFFFFFFFu ? sample_cube(sc, v3(uvw + 3 * i)) : sample2d(sc, tex, vec2{uvw[3 * i], uvw[3 * i + 1]});
        rgba[4 * i] = c.x; rgba[4 * i + 1] = c.y; rgba[4 * i + 2] = c.z; rgba[4 * i + 3] = c.w;
    }
}

// One sample per pixel for frame `frame`: ray_color before firefly clamp. rgb: W*H*3.
void orc_render_radiance(const void* s, const rt_camera* cam, uint32_t frame, uint32_t max_bounces, uint32_t flags,
                         float* rgb, orc_counters* cnt, uint32_t threads) {
    const Scene& sc = *static_cast<const Scene*>(s);
    uint32_t W = cam->resolution[0], H = cam->resolution[1];
    uint32_t nt = threads ? threads : default_threads();
    std::vector<Counters> local(nt);
    parallel_rows(H, nt, [&](uint32_t y, uint32_t tid) {
        for (uint32_t x = 0; x < W; ++x) {
            uint32_t seed = pixel_seed(*cam, x, y, frame);
            Ray r = generate_ray(*cam, x, y, seed);
            vec3 c = ray_color(sc, flags, max_bounces, r, seed, &local[tid]);
            size_t i = (size_t(y) * W + x) * 3;
            rgb[i] = c.x; rgb[i + 1] = c.y; rgb[i + 2] = c.z;
        }
    });
    if (cnt) {
        std::memset(cnt, 0, sizeof(*cnt));
        for (auto& c : local) {
            cnt->rays += c.rays; cnt->shadow_rays += c.shadow_rays; cnt->node_pops += c.node_pops; cnt->leaf_pops += c.leaf_pops;
            cnt->tri_tests += c.tri_tests; cnt->shadow_node_pops += c.shadow_node_pops; cnt->shadow_leaf_pops += c.shadow_leaf_pops;
            cnt->shadow_tri_tests += c.shadow_tri_tests;
        }
    }
}

// State of every pixel's path of frame `frame` on arrival at vertex `bounce` (after that bounce's closest-hit search),
// in the record the product's stage probe takes. reached[i] = 0 for paths that ended earlier (their record is zeroed).
void orc_path_states(const void* s, const rt_camera* cam, uint32_t frame, uint32_t max_bounces, uint32_t flags, uint32_t bounce,
                     rt_shade_vertex* out, uint8_t* reached, uint32_t threads) {
    const Scene& sc = *static_cast<const Scene*>(s);
    uint32_t W = cam->resolution[0], H = cam->resolution[1];
    uint32_t nt = threads ? threads : default_threads();
    parallel_rows(H, nt, [&](uint32_t y, uint32_t) {
        for (uint32_t x = 0; x < W; ++x) {
            uint32_t seed = pixel_seed(*cam, x, y, frame);
            Ray r = generate_ray(*cam, x, y, seed);
            PathTap tap{bounce, false, {}};
            (void)ray_color(sc, flags, max_bounces, r, seed, nullptr, &tap);
            size_t i = size_t(y) * W + x;
            std::memset(&out[i], 0, sizeof(out[i]));
            reached[i] = tap.reached ? 1 : 0;
            if (!tap.reached) continue;
            const VertexState& st = tap.state;
            rt_shade_vertex& o = out[i];
            o.ray_o[0] = st.ray.o.x; o.ray_o[1] = st.ray.o.y; o.ray_o[2] = st.ray.o.z; o.prev_pdf = st.bsdfPdf;
            o.ray_d[0] = st.ray.d.x; o.ray_d[1] = st.ray.d.y; o.ray_d[2] = st.ray.d.z; o.seed = st.seed;
            o.throughput[0] = st.throughput.x; o.throughput[1] = st.throughput.y; o.throughput[2] = st.throughput.z;
            if (st.hit.did_hit) o.hit = rt_hit{st.hit.t, st.hit.u, st.hit.v, st.hit.prim};
            else o.hit = rt_hit{INF, 0.0f, 0.0f, 0xFFFFFFFFu};
        }
    });
}

// shade_vertex on caller-supplied vertices: the twin of the product's rt_shade_vertices. The hit is rebuilt from
// (prim, ray) with hit_triangle, which reproduces t, u, v bit for bit for a hit that came from hit_meshes; a record whose
// triangle the ray does not hit at all is treated as a miss. The light sample is reported as the shader would trace it
// (has_shadow = the light faces the surface), with the contribution it makes if unoccluded.
void orc_shade_vertices(const void* s, uint32_t n, uint32_t bounce, uint32_t flags, const rt_shade_vertex* in, rt_shade_result* out) {
    const Scene& sc = *static_cast<const Scene*>(s);
    for (uint32_t i = 0; i < n; ++i) {
        const rt_shade_vertex& v = in[i];
        VertexState st;
        st.ray = Ray{v3(v.ray_o), v3(v.ray_d)};
        st.throughput = v3(v.throughput);
        st.bsdfPdf = v.prev_pdf;
        st.seed = v.seed;
        if (v.hit.prim != 0xFFFFFFFFu && v.hit.prim < sc.tris.size()) {
            const rt_triangle& tr = sc.tris[v.hit.prim];
            const rt_material& m = sc.mats[tr.material_idx];
            Hit h;
            if (hit_triangle(tr, st.ray, m.shade_smooth != 0, h)) { h.prim = v.hit.prim; h.mat = m; h.has_mat = true; st.hit = h; }
        }
        VertexResult r = shade_vertex(sc, flags, bounce, st);
        rt_shade_result& o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.emitted[0] = r.emitted.x; o.emitted[1] = r.emitted.y; o.emitted[2] = r.emitted.z; o.seed = r.seed;
        o.throughput[0] = r.throughput.x; o.throughput[1] = r.throughput.y; o.throughput[2] = r.throughput.z; o.pdf = r.pdf;
        o.next_o[0] = r.next.o.x; o.next_o[1] = r.next.o.y; o.next_o[2] = r.next.o.z; o.alive = r.alive ? 1u : 0u;
        o.next_d[0] = r.next.d.x; o.next_d[1] = r.next.d.y; o.next_d[2] = r.next.d.z; o.has_shadow = r.light.traced ? 1u : 0u;
        o.shadow_o[0] = r.light.origin.x; o.shadow_o[1] = r.light.origin.y; o.shadow_o[2] = r.light.origin.z;
        o.shadow_tmax = r.light.dist - EPS;
        o.shadow_d[0] = r.light.direction.x; o.shadow_d[1] = r.light.direction.y; o.shadow_d[2] = r.light.direction.z;
        o.light_index = r.light.light_index;
        vec3 c = r.light.Li * r.light_throughput;
        o.shadow_contrib[0] = c.x; o.shadow_contrib[1] = c.y; o.shadow_contrib[2] = c.z;
        o.lobe = r.lobe;
        o.light_tri = r.light.tri;
        o.light_pdf = r.light.light_pdf;
        o.bsdf_pdf_light = r.light.bsdf_pdf;
        o.roughness = r.roughness;
        o.medium_throughput[0] = r.light_throughput.x; o.medium_throughput[1] = r.light_throughput.y; o.medium_throughput[2] = r.light_throughput.z;
    }
}

// Progressive render of frames [first, first+n_frames*stride) step stride.
//   sum_rgba  : W*H*4 floats, in/out: running fp32 sum of firefly-clamped samples, alpha = count (HDR mode)
//   accum8    : W*H*4 bytes, in/out: the reference's 8-bit running mean (used when flags & F_REF_RGBA8)
// The 8-bit mode weights each new sample with 1/(frame+1) exactly like main.comp:262-266.
void orc_render(const void* s, const rt_camera* cam, uint32_t first, uint32_t stride, uint32_t n_frames, uint32_t max_bounces,
                uint32_t flags, float* sum_rgba, uint8_t* accum8, orc_counters* cnt, uint32_t threads) {
    const Scene& sc = *static_cast<const Scene*>(s);
    uint32_t W = cam->resolution[0], H = cam->resolution[1];
    uint32_t nt = threads ? threads : default_threads();
    std::vector<Counters> local(nt);
    for (uint32_t k = 0; k < n_frames; ++k) {
        uint32_t frame = first + k * stride;
        parallel_rows(H, nt, [&](uint32_t y, uint32_t tid) {
            for (uint32_t x = 0; x < W; ++x) {
                uint32_t seed = pixel_seed(*cam, x, y, frame);
                Ray r = generate_ray(*cam, x, y, seed);
                vec3 c = suppress_fireflies(ray_color(sc, flags, max_bounces, r, seed, &local[tid]), 5.0f);
                size_t i = (size_t(y) * W + x) * 4;
                if (flags & F_REF_RGBA8) {
                    float alpha = 1.0f / float(frame + 1u);
                    float nw[4] = {c.x, c.y, c.z, 1.0f};
                    for (int ch = 0; ch < 4; ++ch) accum8[i + ch] = unorm8(mixf(float(accum8[i + ch]) / 255.0f, nw[ch], alpha));
                } else {
                    sum_rgba[i] += c.x; sum_rgba[i + 1] += c.y; sum_rgba[i + 2] += c.z; sum_rgba[i + 3] += 1.0f;
                }
            }
        });
    }
    if (cnt) {
        std::memset(cnt, 0, sizeof(*cnt));
        for (auto& c : local) {
            cnt->rays += c.rays; cnt->shadow_rays += c.shadow_rays; cnt->node_pops += c.node_pops; cnt->leaf_pops += c.leaf_pops;
            cnt->tri_tests += c.tri_tests; cnt->shadow_node_pops += c.shadow_node_pops; cnt->shadow_leaf_pops += c.shadow_leaf_pops;
            cnt->shadow_tri_tests += c.shadow_tri_tests;
        }
    }
}

// Tone-map a mean image (W*H*4 floats, rgb used) to RGBA8 — filmic_tone_mapping + gamma_correction, main.comp:215-244.
void orc_tonemap(uint32_t n_pixels, const float* mean_rgba, uint8_t* out_rgba8) {
    for (uint32_t i = 0; i < n_pixels; ++i) {
        vec3 c = filmic(vec3{mean_rgba[4 * i], mean_rgba[4 * i + 1], mean_rgba[4 * i + 2]});
        out_rgba8[4 * i] = unorm8(c.x); out_rgba8[4 * i + 1] = unorm8(c.y); out_rgba8[4 * i + 2] = unorm8(c.z);
        out_rgba8[4 * i + 3] = unorm8(mean_rgba[4 * i + 3]);
    }
}
void orc_suppress_fireflies(uint32_t n, float* rgb) {
    for (uint32_t i = 0; i < n; ++i) {
        vec3 c = suppress_fireflies(vec3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]}, 5.0f);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}

// Brute-force closest hit over all triangles (no BVH): validates any BVH against the triangle soup.
void orc_trace_brute(const void* s, uint32_t n, const float* o, const float* d, void* hits_out) {
    const Scene& sc = *static_cast<const Scene*>(s);
    struct H4 { float t, u, v; uint32_t prim; };
    H4* out = static_cast<H4*>(hits_out);
    uint32_t rows = (n + 255) / 256;
    parallel_rows(rows, 0, [&](uint32_t r, uint32_t) {
        for (uint32_t i = r * 256; i < std::min(n, (r + 1) * 256); ++i) {
            Ray ray{v3(o + 3 * size_t(i)), v3(d + 3 * size_t(i))};
            Hit best;
            for (uint32_t k = 0; k < sc.tris.size(); ++k) {
                Hit cur;
                if (hit_triangle(sc.tris[k], ray, false, cur) && cur.t < best.t) { cur.prim = k; best = cur; }
            }
            if (best.did_hit) out[i] = {best.t, best.u, best.v, best.prim};
            else out[i] = {INF, 0.0f, 0.0f, 0xFFFFFFFFu};
        }
    });
}

}  // extern "C"
