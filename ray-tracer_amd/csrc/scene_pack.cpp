// Host-side scene packing: light table, scene blob, camera block and the wire-format -> device-layout converter.
// Pure CPU code (no HIP calls); compiled into librt_hip.so.

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "rt_api.h"
#include "rt_internal.h"

namespace rtint {

static thread_local std::string g_thread_error;

int fail(rt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) set_ctx_error(ctx, msg);
    else g_thread_error = msg;
    return code;
}

const char* thread_error() { return g_thread_error.c_str(); }

static inline uint64_t pad16(uint64_t n) { return (n + 15u) & ~uint64_t(15); }

int convert_scene(const rt_bvh_node* nodes, uint32_t n_nodes, const rt_triangle* tris, uint32_t n_tri,
                  const rt_material* mats, uint32_t n_mat, GpuScene& out, std::string& err) {
    out = GpuScene{};
    out.isect.resize(n_tri);
    out.shade.resize(n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        const rt_triangle& t = tris[i];
        if (t.material_idx >= n_mat) {
            err = "triangle " + std::to_string(i) + " references material " + std::to_string(t.material_idx) +
                  " but only " + std::to_string(n_mat) + " exist";
            return RT_ERR_INVALID;
        }
        float ab[3], ac[3];
        for (int k = 0; k < 3; ++k) { ab[k] = t.v[1][k] - t.v[0][k]; ac[k] = t.v[2][k] - t.v[0][k]; }
        // plane normal = normalize(cross(ab, ac)), evaluated exactly as hit_triangle does per ray
        // (intersects.comp:37-39): products and differences individually rounded, then v * (1/sqrt(v.v)).
        float cx = ab[1] * ac[2] - ac[1] * ab[2];
        float cy = ab[2] * ac[0] - ac[2] * ab[0];
        float cz = ab[0] * ac[1] - ac[0] * ab[1];
        float len2 = cx * cx + cy * cy + cz * cz;
        float inv = 1.0f / std::sqrt(len2);
        TriIsect& q = out.isect[i];
        q.a[0] = t.v[0][0]; q.a[1] = t.v[0][1]; q.a[2] = t.v[0][2]; q.a[3] = cx * inv;
        q.b[0] = ab[0];     q.b[1] = ab[1];     q.b[2] = ab[2];     q.b[3] = cy * inv;
        q.c[0] = ac[0];     q.c[1] = ac[1];     q.c[2] = ac[2];     q.c[3] = cz * inv;
        TriShade& s = out.shade[i];
        for (int k = 0; k < 3; ++k) { s.n0[k] = t.n[0][k]; s.n1[k] = t.n[1][k]; }
        s.n2x = t.n[2][0]; s.n2yz[0] = t.n[2][1]; s.n2yz[1] = t.n[2][2];
        for (int k = 0; k < 2; ++k) { s.uv0[k] = t.uv[0][k]; s.uv1[k] = t.uv[1][k]; s.uv2[k] = t.uv[2][k]; }
        const rt_material& m = mats[t.material_idx];
        if (t.material_idx >= (1u << 30)) { err = "material index too large"; return RT_ERR_UNSUPPORTED; }
        s.matword = t.material_idx | (m.spec_trans > 0.5f ? 0x80000000u : 0u) | (m.shade_smooth ? 0x40000000u : 0u);
        // a flat-shaded triangle never uses its vertex normals: chunk 0 carries the plane normal instead, so its vertices
        // read one 16-byte chunk and leave the geometry record alone
        if (!m.shade_smooth) { s.n0[0] = q.a[3]; s.n0[1] = q.b[3]; s.n0[2] = q.c[3]; }
    }

    if (n_tri == 0 || n_nodes == 0) { out.root_ref = kRefNone; return RT_OK; }

    auto leaf_ref = [&](const rt_bvh_node& nd, uint32_t& ref) -> bool {
        if (nd.amt > kLeafMaxCount || nd.idx >= kLeafMaxFirst || uint64_t(nd.idx) + nd.amt > n_tri) return false;
        ref = make_leaf_ref(nd.idx, nd.amt);
        return true;
    };

    const rt_bvh_node& root = nodes[0];
    if (root.amt != 0) {
        if (!leaf_ref(root, out.root_ref)) { err = "root leaf range invalid or larger than 64 triangles"; return RT_ERR_UNSUPPORTED; }
        return RT_OK;
    }

    // Pass 1: validate the tree (indices in range, every node reached once, depth within the traversal stack) and list
    // the interior nodes.
    std::vector<uint32_t> interior_nodes;
    {
        std::vector<uint32_t> stack{0}, depth_stack{0};
        std::vector<uint8_t> seen(n_nodes, 0);
        while (!stack.empty()) {
            uint32_t i = stack.back();
            stack.pop_back();
            const uint32_t depth = depth_stack.back();
            depth_stack.pop_back();
            // the traversal stack holds at most one entry per level: 65 entries serve trees of depth <= 64 (main.comp:6)
            if (depth > 64) { err = "BVH deeper than 64 levels"; return RT_ERR_UNSUPPORTED; }
            if (i >= n_nodes) { err = "node index out of range"; return RT_ERR_INVALID; }
            if (seen[i]) { err = "BVH is not a tree (node reached twice)"; return RT_ERR_INVALID; }
            seen[i] = 1;
            const rt_bvh_node& nd = nodes[i];
            if (nd.amt != 0) continue;
            if (uint64_t(nd.idx) + 1 >= n_nodes) { err = "interior node child index out of range"; return RT_ERR_INVALID; }
            interior_nodes.push_back(i);
            stack.push_back(nd.idx + 1);
            stack.push_back(nd.idx);
            depth_stack.push_back(depth + 1);
            depth_stack.push_back(depth + 1);
        }
    }

    // Record order. The first `n_hot` slots hold the interior nodes with the largest surface area, largest first: the
    // chance that a ray visits a node grows with its area (the premise of the surface-area heuristic), and the traversal
    // kernels keep a prefix of the record array in LDS, so the busiest records never touch the vector-memory pipeline
    // (on a 262k-triangle architectural scene the 1024 largest nodes take about two thirds of all node visits).
    // The remaining nodes follow in the order a depth-first walk meets them, which keeps a subtree's records close in
    // memory, with one twist: when both children of a node are interior, their two 64-byte records are given adjacent
    // slots inside one 128-byte cache line. A ray that enters one child and stacks the other (the common case) then
    // finds the second record already fetched when it pops it.
    const char* layout_env = std::getenv("RT_HIP_NODE_LAYOUT");
    const bool pair_layout = !(layout_env && std::strcmp(layout_env, "dfs") == 0);
    uint32_t hot_slots = kHotNodeSlots;
    if (const char* env = std::getenv("RT_HIP_HOT_NODES")) hot_slots = uint32_t(std::max(0, std::atoi(env)));
    std::vector<uint32_t> record_of(n_nodes, 0xFFFFFFFFu);
    std::vector<uint32_t> interior_order;          // record slot -> source node (0xFFFFFFFF = padding slot)
    {
        auto area_of = [&](uint32_t i) {
            const rt_bvh_node& b = nodes[i];
            float ex = b.max[0] - b.min[0], ey = b.max[1] - b.min[1], ez = b.max[2] - b.min[2];
            float a = ex * ey + ey * ez + ez * ex;
            return (a == a && ex >= 0.0f && ey >= 0.0f && ez >= 0.0f) ? a : -1.0f;      // inverted or NaN bounds sort last
        };
        std::vector<uint32_t> by_area(interior_nodes);
        const size_t n_hot = std::min<size_t>(hot_slots, by_area.size());
        auto hotter = [&](uint32_t a, uint32_t b) {
            float aa = area_of(a), ab = area_of(b);
            return aa != ab ? aa > ab : a < b;
        };
        std::partial_sort(by_area.begin(), by_area.begin() + n_hot, by_area.end(), hotter);
        for (size_t k = 0; k < n_hot; ++k) {
            record_of[by_area[k]] = uint32_t(interior_order.size());
            interior_order.push_back(by_area[k]);
        }
        out.n_hot = uint32_t(n_hot);
    }
    // Pass 2: place the remaining interior nodes (the tree is known to be valid).
    {
        std::vector<uint32_t> stack{0};
        while (!stack.empty()) {
            uint32_t i = stack.back();
            stack.pop_back();
            const rt_bvh_node& nd = nodes[i];
            if (nd.amt != 0) continue;
            if (record_of[i] == 0xFFFFFFFFu) {         // not placed by its parent or as a hot node: next free slot
                record_of[i] = uint32_t(interior_order.size());
                interior_order.push_back(i);
            }
            const bool li = nodes[nd.idx].amt == 0, ri = nodes[nd.idx + 1].amt == 0;
            if (pair_layout && li && ri && record_of[nd.idx] == 0xFFFFFFFFu && record_of[nd.idx + 1] == 0xFFFFFFFFu) {
                if (interior_order.size() & 1u) interior_order.push_back(0xFFFFFFFFu);
                record_of[nd.idx] = uint32_t(interior_order.size());
                interior_order.push_back(nd.idx);
                record_of[nd.idx + 1] = uint32_t(interior_order.size());
                interior_order.push_back(nd.idx + 1);
            }
            stack.push_back(nd.idx + 1);
            stack.push_back(nd.idx);
        }
    }
    // the traversal kernels address records with a 32-bit byte offset (64 bytes per record)
    if (interior_order.size() >= (1u << 26)) { err = "more than 2^26 interior node records"; return RT_ERR_UNSUPPORTED; }
    out.nodes.resize(interior_order.size());
    for (size_t r = 0; r < interior_order.size(); ++r) {
        if (interior_order[r] == 0xFFFFFFFFu) { out.nodes[r] = NodePair{}; continue; }
        const rt_bvh_node& nd = nodes[interior_order[r]];
        const rt_bvh_node& L = nodes[nd.idx];
        const rt_bvh_node& R = nodes[nd.idx + 1];
        NodePair& p = out.nodes[r];
        p.q0[0] = L.min[0]; p.q0[1] = L.min[1]; p.q0[2] = L.min[2]; p.q0[3] = L.max[0];
        p.q1[0] = L.max[1]; p.q1[1] = L.max[2]; p.q1[2] = R.min[0]; p.q1[3] = R.min[1];
        p.q2[0] = R.min[2]; p.q2[1] = R.max[0]; p.q2[2] = R.max[1]; p.q2[3] = R.max[2];
        const rt_bvh_node* kids[2] = {&L, &R};
        for (int k = 0; k < 2; ++k) {
            if (kids[k]->amt != 0) {
                if (!leaf_ref(*kids[k], p.ref[k])) {
                    err = "leaf range invalid, larger than 64 triangles, or beyond 2^25 triangles";
                    return RT_ERR_UNSUPPORTED;
                }
            } else {
                p.ref[k] = record_of[nd.idx + k];
            }
        }
        p.ref[2] = p.ref[3] = 0;
    }
    out.root_ref = record_of[0];

    // ---- 4-wide tree for occlusion queries (see NodeQuad) ----
    {
        auto nested = [&](uint32_t child, uint32_t parent) {       // false for NaN bounds and for boxes stored with min > max
            const rt_bvh_node& c = nodes[child];
            const rt_bvh_node& p = nodes[parent];
            for (int k = 0; k < 3; ++k)
                if (!(c.min[k] >= p.min[k] && c.max[k] <= p.max[k] && c.min[k] <= c.max[k])) return false;
            return true;
        };
        // entries of the wide node rooted at binary interior node i
        auto entries_of = [&](uint32_t i, uint32_t e[4]) {
            uint32_t n = 0;
            const uint32_t kid = nodes[i].idx;
            for (uint32_t c = kid; c < kid + 2; ++c) {
                const rt_bvh_node& cn = nodes[c];
                if (cn.amt == 0 && nested(cn.idx, c) && nested(cn.idx + 1, c)) { e[n++] = cn.idx; e[n++] = cn.idx + 1; }
                else e[n++] = c;
            }
            return n;
        };
        std::vector<uint32_t> wide_nodes{0};                        // binary nodes that head a wide record, in discovery order
        std::vector<uint32_t> wide_of(n_nodes, 0xFFFFFFFFu);
        wide_of[0] = 0;
        for (size_t w = 0; w < wide_nodes.size(); ++w) {
            uint32_t e[4];
            const uint32_t n = entries_of(wide_nodes[w], e);
            for (uint32_t k = 0; k < n; ++k)
                if (nodes[e[k]].amt == 0) { wide_of[e[k]] = uint32_t(wide_nodes.size()); wide_nodes.push_back(e[k]); }
        }
        // order: the largest-area heads first (kept in LDS by the kernels), the rest in discovery (breadth-first) order
        std::vector<uint32_t> order(wide_nodes.size());
        for (size_t k = 0; k < order.size(); ++k) order[k] = uint32_t(k);
        auto area = [&](uint32_t w) {
            const rt_bvh_node& b = nodes[wide_nodes[w]];
            float ex = b.max[0] - b.min[0], ey = b.max[1] - b.min[1], ez = b.max[2] - b.min[2];
            float a = ex * ey + ey * ez + ez * ex;
            return (a == a && ex >= 0.0f && ey >= 0.0f && ez >= 0.0f) ? a : -1.0f;
        };
        const size_t n_hot = std::min<size_t>(hot_slots, order.size());
        std::partial_sort(order.begin(), order.begin() + n_hot, order.end(), [&](uint32_t a, uint32_t b) {
            float aa = area(a), ab = area(b);
            return aa != ab ? aa > ab : a < b;
        });
        std::sort(order.begin() + n_hot, order.end());
        const char* wl = std::getenv("RT_HIP_WIDE_LAYOUT");
        if (!(wl && std::strcmp(wl, "bfs") == 0)) {
            // Records outside the hot prefix: the interior children of a record take consecutive slots (a visit stacks up
            // to three of them and pops them soon after), and the blocks follow each other in the order a depth-first walk
            // opens them, so a subtree's records share cache lines and pages.
            std::vector<uint8_t> placed(order.size(), 0);
            for (size_t k = 0; k < n_hot; ++k) placed[order[k]] = 1;
            std::vector<uint32_t> rest;
            rest.reserve(order.size() - n_hot);
            if (!placed[0]) { placed[0] = 1; rest.push_back(0); }
            std::vector<uint32_t> walk{0};
            while (!walk.empty()) {
                const uint32_t w = walk.back();
                walk.pop_back();
                uint32_t e[4], kids[4];
                const uint32_t n = entries_of(wide_nodes[w], e);
                uint32_t nk = 0;
                for (uint32_t k = 0; k < n; ++k)
                    if (nodes[e[k]].amt == 0) kids[nk++] = wide_of[e[k]];
                for (uint32_t k = 0; k < nk; ++k)
                    if (!placed[kids[k]]) { placed[kids[k]] = 1; rest.push_back(kids[k]); }
                for (uint32_t k = nk; k-- > 0;) walk.push_back(kids[k]);
            }
            std::copy(rest.begin(), rest.end(), order.begin() + n_hot);
        }
        std::vector<uint32_t> slot_of(order.size());
        for (size_t s = 0; s < order.size(); ++s) slot_of[order[s]] = uint32_t(s);
        if (order.size() >= (1u << 25)) { err = "more than 2^25 wide node records"; return RT_ERR_UNSUPPORTED; }
        out.wide.assign(order.size(), NodeQuad{});
        const float nan = std::nanf("");
        for (size_t s = 0; s < order.size(); ++s) {
            NodeQuad& q = out.wide[s];
            uint32_t e[4];
            const uint32_t n = entries_of(wide_nodes[order[s]], e);
            for (uint32_t k = 0; k < 4; ++k) {
                if (k >= n) {
                    for (int j = 0; j < 6; ++j) q.box[k][j] = nan;
                    q.ref[k] = kRefNone;
                    continue;
                }
                const rt_bvh_node& b = nodes[e[k]];
                for (int j = 0; j < 3; ++j) { q.box[k][j] = b.min[j]; q.box[k][3 + j] = b.max[j]; }
                if (b.amt != 0) {
                    if (!leaf_ref(b, q.ref[k])) { err = "leaf range invalid, larger than 64 triangles, or beyond 2^25 triangles"; return RT_ERR_UNSUPPORTED; }
                } else {
                    q.ref[k] = slot_of[wide_of[e[k]]];
                }
            }
        }
        out.wide_root = slot_of[0];
        out.n_hot_wide = uint32_t(n_hot);
    }

    // ---- compressed twin of the 4-wide tree (see NodeCw) ----
    {
        float size = 0.0f;
        for (int k = 0; k < 3; ++k) {
            if (std::isfinite(nodes[0].min[k])) size = std::max(size, std::fabs(nodes[0].min[k]));
            if (std::isfinite(nodes[0].max[k])) size = std::max(size, std::fabs(nodes[0].max[k]));
        }
        out.scene_size = size;
        out.box_margin = std::ldexp(size, -17);
        const double margin = std::max(double(out.box_margin), 1e-30);
        out.cw.assign(out.wide.size(), NodeCw{});
        bool overflow = false;
        for (size_t s = 0; s < out.wide.size(); ++s) {
            const NodeQuad& q = out.wide[s];
            NodeCw& w = out.cw[s];
            // The reference's slab test (intersects.comp:108-118) takes min(t1, t2) and max(t1, t2) per axis, so it does not
            // care which of a box's two planes is called min: a box stored with min > max on some axis is entered exactly like
            // the same box with the planes swapped, and is quantised as that box here. A NaN bound is never entered. A bound at
            // infinity is (a triangle with an infinite coordinate shares its leaf and its ancestors with ordinary ones), but
            // 8-bit offsets cannot express it: such a scene keeps the exact traversal.
            bool usable[4];
            float blo[4][3], bhi[4][3];
            for (int k = 0; k < 4; ++k) {
                usable[k] = q.ref[k] != kRefNone;
                for (int a = 0; a < 3; ++a) {
                    const float b0 = q.box[k][a], b1 = q.box[k][3 + a];
                    blo[k][a] = std::fmin(b0, b1); bhi[k][a] = std::fmax(b0, b1);
                    if (!usable[k]) continue;
                    if (std::isnan(b0) || std::isnan(b1)) usable[k] = false;
                    else if (!std::isfinite(b0) || !std::isfinite(b1)) { usable[k] = false; out.cw_complete = false; }
                }
                w.ref[k] = usable[k] ? q.ref[k] : kRefNone;
            }
            uint32_t lo_w[3] = {0, 0, 0}, hi_w[3] = {0, 0, 0};
            float step[3];
            for (int a = 0; a < 3; ++a) {
                double lo_min = 1e300, hi_max = -1e300;
                for (int k = 0; k < 4; ++k)
                    if (usable[k]) { lo_min = std::min(lo_min, double(blo[k][a])); hi_max = std::max(hi_max, double(bhi[k][a])); }
                if (lo_min > hi_max) { lo_min = hi_max = 0.0; }
                // origin one margin below the lowest box (rounded down to float), step so that the highest box plus a margin
                // fits into 253 steps (rounded up to float); at least one margin per step
                float p = float(lo_min - margin);
                if (double(p) > lo_min - margin) p = std::nextafter(p, -INFINITY);
                double ext = hi_max + margin - double(p);
                float st = float(std::max(ext / 253.0, margin));
                if (double(st) < std::max(ext / 253.0, margin)) st = std::nextafter(st, INFINITY);
                w.p[a] = p;
                step[a] = st;
                for (int k = 0; k < 4; ++k) {
                    uint32_t ql = 255u, qh = 0u;                                   // unused: never entered
                    if (usable[k]) {
                        long l = long(std::floor((double(blo[k][a]) - margin - double(p)) / double(st)));
                        long h = long(std::ceil((double(bhi[k][a]) + margin - double(p)) / double(st)));
                        overflow |= h > 255;
                        ql = uint32_t(std::min(255L, std::max(0L, l)));
                        qh = uint32_t(std::min(255L, std::max(0L, h)));
                    }
                    lo_w[a] |= ql << (8 * k);
                    hi_w[a] |= qh << (8 * k);
                }
            }
            w.sx = step[0]; w.sy = step[1]; w.sz = step[2];
            w.lo[0] = lo_w[0]; w.lo[1] = lo_w[1]; w.lo[2] = lo_w[2];
            w.hix = hi_w[0]; w.hiy = hi_w[1]; w.hiz = hi_w[2];
        }
        if (overflow) { err = "internal: a compressed child box does not fit its 8-bit range"; return RT_ERR_UNSUPPORTED; }
    }
    return RT_OK;
}

uint32_t order_by_visits(const uint32_t* visits, size_t n_records, uint32_t n_hot, std::vector<uint32_t>& new_of_old) {
    new_of_old.resize(n_records);
    for (size_t i = 0; i < n_records; ++i) new_of_old[i] = uint32_t(i);
    const size_t H = std::min<size_t>(n_hot, n_records);
    if (H == 0) return 0;
    std::vector<uint32_t> top(n_records);
    for (size_t i = 0; i < n_records; ++i) top[i] = uint32_t(i);
    std::partial_sort(top.begin(), top.begin() + H, top.end(), [&](uint32_t a, uint32_t b) {
        return visits[a] != visits[b] ? visits[a] > visits[b] : a < b;
    });
    std::vector<uint8_t> in_top(n_records, 0);
    for (size_t k = 0; k < H; ++k) in_top[top[k]] = 1;
    std::vector<uint32_t> vacated, displaced;            // both ascending
    for (size_t i = H; i < n_records; ++i) if (in_top[i]) vacated.push_back(uint32_t(i));
    for (size_t i = 0; i < H; ++i) if (!in_top[i]) displaced.push_back(uint32_t(i));
    for (size_t k = 0; k < H; ++k) new_of_old[top[k]] = uint32_t(k);
    for (size_t k = 0; k < displaced.size(); ++k) new_of_old[displaced[k]] = vacated[k];
    uint32_t moved = 0;
    for (size_t i = 0; i < n_records; ++i) moved += new_of_old[i] != i;
    return moved;
}

}  // namespace rtint

using rtint::fail;

extern "C" const char* rt_version(void) { return "rt_hip 0.1 (gfx950)"; }

extern "C" int rt_build_lights(const rt_triangle* tris, uint32_t n_tri, const rt_material* mats, uint32_t n_mat,
                               rt_emissive_tri* emissive, float* cdf, uint32_t* n_emissive, float* total_power) {
    if ((n_tri && (!tris || !emissive || !cdf)) || !n_emissive) return fail(nullptr, RT_ERR_INVALID, "rt_build_lights: null argument");
    uint32_t n = 0;
    float total = 0.0f, running = 0.0f;
    for (uint32_t i = 0; i < n_tri; ++i) {
        const rt_triangle& t = tris[i];
        if (t.material_idx >= n_mat) continue;  // upstream skips out-of-range material indices silently
        const float* e = mats[t.material_idx].emission;
        if (!(e[0] * e[0] + e[1] * e[1] + e[2] * e[2] > 0.0f)) continue;
        float e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = t.v[1][k] - t.v[0][k]; e2[k] = t.v[2][k] - t.v[0][k]; }
        float cx = e1[1] * e2[2] - e1[2] * e2[1];
        float cy = e1[2] * e2[0] - e1[0] * e2[2];
        float cz = e1[0] * e2[1] - e1[1] * e2[0];
        float area = std::sqrt(cx * cx + cy * cy + cz * cz) * 0.5f;
        float lum = e[0] * 0.2126f + e[1] * 0.7152f + e[2] * 0.0722f;
        float power = lum * area;
        if (!(power > 0.0f)) continue;
        emissive[n] = rt_emissive_tri{i, t.material_idx, area, power};
        total += power;
        running += power;
        cdf[n] = running;
        ++n;
    }
    *n_emissive = n;
    if (total_power) *total_power = total;
    return RT_OK;
}

extern "C" int rt_scene_buffer_sizes(uint32_t n_bvh, uint32_t n_mat, uint32_t n_tri, uint32_t n_light,
                                     uint32_t n_emissive, uint64_t sizes[6]) {
    if (!sizes) return fail(nullptr, RT_ERR_INVALID, "rt_scene_buffer_sizes: null sizes");
    sizes[0] = rtint::pad16(uint64_t(n_bvh) * sizeof(rt_bvh_node));
    sizes[1] = rtint::pad16(uint64_t(n_mat) * sizeof(rt_material));
    sizes[2] = rtint::pad16(uint64_t(n_tri) * sizeof(rt_triangle));
    sizes[3] = rtint::pad16(uint64_t(n_light) * sizeof(rt_light));
    sizes[4] = rtint::pad16(uint64_t(n_emissive) * sizeof(rt_emissive_tri));
    sizes[5] = rtint::pad16(uint64_t(n_emissive) * sizeof(float));
    return RT_OK;
}

extern "C" int rt_scene_write_buffers(const rt_bvh_node* nodes, uint32_t n_bvh, const rt_material* mats, uint32_t n_mat,
                                      const rt_triangle* tris, uint32_t n_tri, const rt_light* lights, uint32_t n_light,
                                      const rt_emissive_tri* emissive, const float* cdf, uint32_t n_emissive,
                                      void* blob, uint64_t blob_capacity) {
    uint64_t sz[6];
    rt_scene_buffer_sizes(n_bvh, n_mat, n_tri, n_light, n_emissive, sz);
    uint64_t total = sz[0] + sz[1] + sz[2] + sz[3] + sz[4] + sz[5];
    if (!blob || blob_capacity < total) return fail(nullptr, RT_ERR_INVALID, "rt_scene_write_buffers: blob too small");
    uint8_t* p = static_cast<uint8_t*>(blob);
    std::memset(p, 0, total);
    const void* src[6] = {nodes, mats, tris, lights, emissive, cdf};
    const uint64_t bytes[6] = {uint64_t(n_bvh) * sizeof(rt_bvh_node),      uint64_t(n_mat) * sizeof(rt_material),
                               uint64_t(n_tri) * sizeof(rt_triangle),      uint64_t(n_light) * sizeof(rt_light),
                               uint64_t(n_emissive) * sizeof(rt_emissive_tri), uint64_t(n_emissive) * sizeof(float)};
    for (int s = 0; s < 6; ++s) {
        if (bytes[s]) {
            if (!src[s]) return fail(nullptr, RT_ERR_INVALID, "rt_scene_write_buffers: null section with non-zero count");
            std::memcpy(p, src[s], bytes[s]);
        }
        p += sz[s];
    }
    return RT_OK;
}

extern "C" int rt_camera_look_at(rt_camera* cam, uint32_t width, uint32_t height, const float location[3],
                                 const float look_at[3], float focal_length, float focus_distance, float aperture_radius) {
    if (!cam || !location || !look_at || width == 0 || height == 0) return fail(nullptr, RT_ERR_INVALID, "rt_camera_look_at: bad argument");
    std::memset(cam, 0, sizeof(*cam));
    cam->resolution[0] = width;
    cam->resolution[1] = height;
    float ratio = float(width) / float(height);
    if (ratio > 1.0f) { cam->view_port_uv[0] = ratio; cam->view_port_uv[1] = 1.0f; }
    else { cam->view_port_uv[0] = 1.0f; cam->view_port_uv[1] = 1.0f / ratio; }
    cam->focal_length = focal_length;
    cam->focus_distance = focus_distance;
    cam->aperture_radius = aperture_radius;
    for (int k = 0; k < 3; ++k) cam->location[k] = location[k];

    auto norm = [](float v[3]) {
        float r = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        v[0] *= r; v[1] *= r; v[2] *= r;
    };
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - b[1] * a[2];
        o[1] = a[2] * b[0] - b[2] * a[0];
        o[2] = a[0] * b[1] - b[0] * a[1];
    };
    auto dot = [](const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    // right-handed look-at: f = forward, s = right, u = up; the stored matrix is its transpose, whose
    // upper-left 3x3 has columns s, u, -f (camera -> world).
    float f[3] = {look_at[0] - location[0], look_at[1] - location[1], look_at[2] - location[2]};
    norm(f);
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float s[3], u[3];
    cross(f, up, s);
    norm(s);
    cross(s, f, u);
    float* m = cam->rotation;  // column-major
    m[0] = s[0];  m[1] = s[1];  m[2] = s[2];   m[3] = -dot(location, s);
    m[4] = u[0];  m[5] = u[1];  m[6] = u[2];   m[7] = -dot(location, u);
    m[8] = -f[0]; m[9] = -f[1]; m[10] = -f[2]; m[11] = dot(location, f);
    m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f;  m[15] = 1.0f;
    return RT_OK;
}
