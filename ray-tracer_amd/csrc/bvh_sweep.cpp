// Host-side twin of the reference's first builder, `Bvh` (src/accelerators/bvh_based/bvh.rs:183-331): top-down, eight
// candidate planes per axis evenly spaced inside the node's bounds, cost = N_left * area_left + N_right * area_right with
// triangles assigned by the centre of their extent, a node stays a leaf when its best split does not beat the cost its
// parent was split at. It exists for the build-time baseline (benches/bvh.rs:20-27 times this builder next to BinSah) and
// as a second producer of trees that honour the traversal contract. Single-threaded, like the original.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <vector>

#include "rt_api.h"
#include "rt_internal.h"
#include "bvh_common.h"

namespace {

using namespace rtbvh;
constexpr int kSplitAttempts = 8;      // SPLIT_ATTEMPTS, bvh.rs:17

struct TriExtent { V3 lo, hi, mid; };  // per-axis extent of the three vertices and its centre (bvh.rs:338-350, :258)

float surface_area(const Box& b) {     // BvhNode::area, bvh.rs:100-110: 0 for a box nothing was added to
    if (b.lo.x == kInf || b.hi.x == -kInf) return 0.0f;
    V3 d = vsub(b.hi, b.lo);
    return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}

struct Sweep {
    const std::vector<TriExtent>& ext;
    std::vector<uint32_t>& idx;

    float split_cost(uint32_t lo, uint32_t hi, int axis, float pos) const {       // bvh.rs:314-330
        Box l, r;
        uint32_t n_left = 0;
        for (uint32_t k = lo; k < hi; ++k) {
            const TriExtent& e = ext[idx[k]];
            if (e.mid[axis] < pos) { l.grow(e.lo, e.hi); ++n_left; }
            else r.grow(e.lo, e.hi);
        }
        return float(n_left) * surface_area(l) + float(hi - lo - n_left) * surface_area(r);
    }
};

}  // namespace

extern "C" int rt_bvh_build_sweep(rt_triangle* tris, uint32_t n_tri, rt_bvh_node* nodes, uint32_t* n_nodes, rt_bvh_stats* stats) {
    if (!nodes || !n_nodes || (n_tri && !tris)) return rtint::fail(nullptr, RT_ERR_INVALID, "rt_bvh_build_sweep: null argument");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<TriExtent> ext(n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        V3 a{tris[i].v[0][0], tris[i].v[0][1], tris[i].v[0][2]}, b{tris[i].v[1][0], tris[i].v[1][1], tris[i].v[1][2]},
           c{tris[i].v[2][0], tris[i].v[2][1], tris[i].v[2][2]};
        ext[i].lo = vmin(a, vmin(b, c));
        ext[i].hi = vmax(a, vmax(b, c));
        ext[i].mid = V3{(ext[i].lo.x + ext[i].hi.x) / 2.0f, (ext[i].lo.y + ext[i].hi.y) / 2.0f, (ext[i].lo.z + ext[i].hi.z) / 2.0f};
    }
    std::vector<uint32_t> idx(n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) idx[i] = i;
    Sweep sw{ext, idx};

    rt_bvh_stats st{};
    uint32_t used = 1;
    struct Job { uint32_t node, lo, hi, depth; float parent_cost; };
    std::vector<Job> todo;
    todo.push_back({0u, 0u, n_tri, 0u, 1e30f});
    float leaf_items = 0.0f;
    while (!todo.empty()) {
        const Job j = todo.back();
        todo.pop_back();
        Box box;
        for (uint32_t k = j.lo; k < j.hi; ++k) box.grow(ext[idx[k]].lo, ext[idx[k]].hi);
        rt_bvh_node& nd = nodes[j.node];
        // an empty range keeps the inverted "nothing inside" bounds of BvhNode::default (bvh.rs:52-61)
        nd.min[0] = box.lo.x; nd.min[1] = box.lo.y; nd.min[2] = box.lo.z;
        nd.max[0] = box.hi.x; nd.max[1] = box.hi.y; nd.max[2] = box.hi.z;
        nd.idx = j.lo;
        nd.amt = j.hi - j.lo;
        st.max_depth = std::max(st.max_depth, j.depth);
        const uint32_t n = j.hi - j.lo;
        bool split = j.depth < kMaxDepth && n > 1;
        int axis = 0;
        float pos = 0.0f, best = 1e30f;
        if (split) {
            const V3 dims = vsub(box.hi, box.lo);
            for (int a = 0; a < 3; ++a) {                                          // find_best_split, bvh.rs:288-312
                const float step = dims[a] / float(kSplitAttempts + 1);
                for (int attempt = 1; attempt <= kSplitAttempts; ++attempt) {
                    const float p = box.lo[a] + step * float(attempt);
                    const float c = sw.split_cost(j.lo, j.hi, a, p);
                    if (c < best) { best = c; axis = a; pos = p; }
                }
            }
            best = sw.split_cost(j.lo, j.hi, axis, pos);
            split = best < j.parent_cost;
        }
        uint32_t left = 0;
        if (split) {                                                               // two-ended partition, bvh.rs:242-258
            uint32_t right = n - 1;
            while (left <= right) {
                if (ext[idx[j.lo + left]].mid[axis] < pos) ++left;
                else {
                    std::swap(idx[j.lo + left], idx[j.lo + right]);
                    if (right == 0) break;
                    --right;
                }
            }
            split = left != 0 && left != n;
        }
        if (!split) {
            st.leaf_count++;
            st.max_leaf_size = std::max(st.max_leaf_size, n);
            leaf_items += float(n);
            continue;
        }
        const uint32_t kids = used;
        used += 2;
        nd.idx = kids;
        nd.amt = 0;
        // left subtree first (it claims the next node slots), as the recursion does
        todo.push_back({kids + 1, j.lo + left, j.hi, j.depth + 1, best});
        todo.push_back({kids, j.lo, j.lo + left, j.depth + 1, best});
    }
    std::vector<rt_triangle> src(tris, tris + n_tri);
    for (uint32_t k = 0; k < n_tri; ++k) tris[k] = src[idx[k]];                    // apply_ordering, bvh.rs:332-336

    *n_nodes = used;
    st.node_count = used;
    st.avg_leaf_size = leaf_items / float(std::max(1u, st.leaf_count));
    // SAH cost with unit costs, relative to the root's area, for comparison with the binned builders' statistics
    {
        auto area = [&](const rt_bvh_node& b) {
            float ex = b.max[0] - b.min[0], ey = b.max[1] - b.min[1], ez = b.max[2] - b.min[2];
            return (ex >= 0.0f && ey >= 0.0f && ez >= 0.0f) ? ex * ey + ey * ez + ez * ex : 0.0f;
        };
        const float root = std::max(area(nodes[0]), 1e-30f);
        float sah = 0.0f;
        for (uint32_t i = 0; i < used; ++i) sah += area(nodes[i]) / root * (nodes[i].amt ? float(nodes[i].amt) : 1.0f);
        st.sah_cost = sah;
    }
    st.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return RT_OK;
}
