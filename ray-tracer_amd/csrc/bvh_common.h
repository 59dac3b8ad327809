// Small vector / box helpers and the 16-bin mapping shared by the host-side BVH builders (object-split, spatial-split and
// sweep). Host code only.
#pragma once

#include <algorithm>
#include <cstdint>
#include <limits>

namespace rtbvh {

constexpr int kBins = 16;              // BinSah: 16 bins per axis (bin_sah.rs:115-125)
constexpr uint32_t kMaxDepth = 64;     // MAX_BVH_DEPTH, main.comp:6
constexpr float kInf = std::numeric_limits<float>::infinity();

struct V3 {
    float x, y, z;
    float operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }
    float& at(int a) { return a == 0 ? x : (a == 1 ? y : z); }
};
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 vmul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }

struct Box {
    V3 lo{kInf, kInf, kInf};
    V3 hi{-kInf, -kInf, -kInf};
    void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(V3 a, V3 b) { lo = vmin(lo, a); hi = vmax(hi, b); }
    float half_area() const {
        V3 d = vmax(vsub(hi, lo), V3{0, 0, 0});
        return d.x * d.y + d.y * d.z + d.z * d.x;
    }
};
inline Box intersect(const Box& a, const Box& b) { return Box{vmax(a.lo, b.lo), vmin(a.hi, b.hi)}; }

inline int widest_axis(V3 v) {
    if (v.x >= v.y && v.x >= v.z) return 0;
    return v.y >= v.z ? 1 : 2;
}

// Centroid -> bin index per axis: bins span the centroid bounds `c`, degenerate axes map everything to bin 0.
inline V3 bin_scale(const Box& c) {
    V3 ext = vmax(vsub(c.hi, c.lo), V3{0, 0, 0});
    const float k = float(kBins) * (1.0f - 1e-6f);
    return {ext.x > 1e-12f ? k / ext.x : 0.0f, ext.y > 1e-12f ? k / ext.y : 0.0f, ext.z > 1e-12f ? k / ext.z : 0.0f};
}
inline int clamp_bin(float v) {
    if (!(v > 0.0f)) v = 0.0f;
    if (v > float(kBins - 1)) v = float(kBins - 1);
    return int(v);
}
inline void bin_of(V3 cen, V3 origin, V3 scale, int out[3]) {
    V3 t = vsub(cen, origin);
    out[0] = clamp_bin(t.x * scale.x);
    out[1] = clamp_bin(t.y * scale.y);
    out[2] = clamp_bin(t.z * scale.z);
}

}  // namespace rtbvh
