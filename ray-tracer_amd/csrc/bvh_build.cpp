// Host-side BVH construction for the render core: top-down binned SAH with object splits.
//
// Produces the node/triangle ordering the traversal kernels (and the reference shader) rely on:
// 32-byte nodes, root at 0, sibling pairs adjacent, leaves cover contiguous triangle ranges, nodes
// emitted depth-first, depth <= 64.  Behavioural twin of the reference's `BinSah` with
// `SplitPolicy::Object` (src/accelerators/bvh_based/bin_sah.rs:39-44,115-125,404-438,728-859,1462-1519):
// 16 bins per axis, leaf when n <= min_leaf, SAH termination test `split_cost >= n * intersect_cost`,
// centroid-median fallback when a range above max_leaf refuses to split.  The concurrency scheme
// (a shared work list drained by std::thread workers) is this library's own; the resulting tree does not
// depend on the thread count because every per-node decision is a pure function of its range.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "rt_api.h"
#include "rt_internal.h"
#include "bvh_common.h"

namespace {

using namespace rtbvh;
constexpr uint32_t kSharedWorkMin = 8 * 1024;  // ranges at least this big are split one level at a time and shared

struct BinSet {
    Box box[3][kBins];
    uint32_t count[3][kBins];
    void clear() {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < kBins; ++b) { box[a][b] = Box{}; count[a][b] = 0; }
    }
};

// Intermediate node: count != 0 -> leaf over refs [first, first+count); else children at slots first, first+1.
struct TmpNode {
    Box box;
    uint32_t first = 0, count = 0;
};

struct WorkItem {
    uint32_t slot, lo, hi, depth;
    Box geom, cent;
};

struct Builder {
    const rt_bvh_params cfg;
    const std::vector<V3>& pmin;
    const std::vector<V3>& pmax;
    std::vector<uint32_t>& refs;
    std::vector<TmpNode> arena;
    std::atomic<uint32_t> arena_used{1};

    std::mutex mu;
    std::condition_variable cv;
    std::vector<WorkItem> shared;
    int outstanding = 0;

    Builder(const rt_bvh_params& p, const std::vector<V3>& mn, const std::vector<V3>& mx, std::vector<uint32_t>& r)
        : cfg(p), pmin(mn), pmax(mx), refs(r), arena(2 * r.size() + 2) {}

    V3 centroid(uint32_t i) const { return vmul(vadd(pmin[i], pmax[i]), 0.5f); }

    void bounds_of(uint32_t lo, uint32_t hi, Box& g, Box& c) const {
        g = Box{}; c = Box{};
        for (uint32_t k = lo; k < hi; ++k) {
            uint32_t i = refs[k];
            g.grow(pmin[i], pmax[i]);
            c.grow(centroid(i));
        }
    }

    // Returns false for "make a leaf"; otherwise partitions refs[lo,hi) and fills the child descriptions.
    bool split(const WorkItem& it, BinSet& bins, uint32_t& mid, Box& lg, Box& lc, Box& rg, Box& rc) {
        const uint32_t n = it.hi - it.lo;
        if (n <= cfg.min_leaf || it.depth >= kMaxDepth) return false;

        auto forced = [&]() -> bool {
            if (n <= cfg.max_leaf) return false;
            median(it, mid, lg, lc, rg, rc);
            return true;
        };

        V3 scale = bin_scale(it.cent);
        if (!(scale.x > 0.0f || scale.y > 0.0f || scale.z > 0.0f)) return forced();

        bins.clear();
        for (uint32_t k = it.lo; k < it.hi; ++k) {
            uint32_t i = refs[k];
            int b[3];
            bin_of(centroid(i), it.cent.lo, scale, b);
            for (int a = 0; a < 3; ++a) {
                bins.box[a][b[a]].grow(pmin[i], pmax[i]);
                bins.count[a][b[a]]++;
            }
        }

        float best = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = 0;
        for (int a = 0; a < 3; ++a) {
            float right_h[kBins];
            uint32_t right_n[kBins];
            Box acc;
            uint32_t cnt = 0;
            for (int i = kBins - 1; i >= 1; --i) {
                acc.grow(bins.box[a][i].lo, bins.box[a][i].hi);
                cnt += bins.count[a][i];
                right_h[i] = acc.half_area();
                right_n[i] = cnt;
            }
            acc = Box{};
            cnt = 0;
            for (int i = 1; i < kBins; ++i) {
                acc.grow(bins.box[a][i - 1].lo, bins.box[a][i - 1].hi);
                cnt += bins.count[a][i - 1];
                if (cnt == 0 || right_n[i] == 0) continue;
                float cost = float(cnt) * acc.half_area() + float(right_n[i]) * right_h[i];
                if (cost < best) { best = cost; best_axis = a; best_bin = i; }
            }
        }
        if (best_axis < 0) return forced();

        float parent_h = std::max(it.geom.half_area(), 1e-30f);
        float split_cost = cfg.traversal_cost + cfg.intersect_cost * best / parent_h;
        if (split_cost >= float(n) * cfg.intersect_cost) return forced();

        // two-pointer partition by bin index
        lc = Box{}; rc = Box{};
        uint32_t l = it.lo, r = it.hi;
        while (l < r) {
            uint32_t i = refs[l];
            V3 cen = centroid(i);
            int b[3];
            bin_of(cen, it.cent.lo, scale, b);
            if (b[best_axis] < best_bin) { lc.grow(cen); ++l; }
            else { rc.grow(cen); --r; std::swap(refs[l], refs[r]); }
        }
        mid = l;
        if (mid == it.lo || mid == it.hi) { median(it, mid, lg, lc, rg, rc); return true; }
        lg = Box{}; rg = Box{};
        for (int b = 0; b < best_bin; ++b) lg.grow(bins.box[best_axis][b].lo, bins.box[best_axis][b].hi);
        for (int b = best_bin; b < kBins; ++b) rg.grow(bins.box[best_axis][b].lo, bins.box[best_axis][b].hi);
        return true;
    }

    void median(const WorkItem& it, uint32_t& mid, Box& lg, Box& lc, Box& rg, Box& rc) {
        const uint32_t n = it.hi - it.lo;
        mid = it.lo + n / 2;
        V3 cext = vmax(vsub(it.cent.hi, it.cent.lo), V3{0, 0, 0});
        int axis = std::max(cext.x, std::max(cext.y, cext.z)) > 0.0f
                       ? widest_axis(cext)
                       : widest_axis(vmax(vsub(it.geom.hi, it.geom.lo), V3{0, 0, 0}));
        auto key = [&](uint32_t i) { return (pmin[i][axis] + pmax[i][axis]) * 0.5f; };
        std::nth_element(refs.begin() + it.lo, refs.begin() + mid, refs.begin() + it.hi,
                         [&](uint32_t p, uint32_t q) {
                             float a = key(p), b = key(q);
                             return a < b || (a == b && p < q);
                         });
        bounds_of(it.lo, mid, lg, lc);
        bounds_of(mid, it.hi, rg, rc);
    }

    // Handles one item; children either go to `local` (depth-first in this thread) or to the shared list.
    void process(const WorkItem& it, BinSet& bins, std::vector<WorkItem>& local) {
        uint32_t mid;
        Box lg, lc, rg, rc;
        if (!split(it, bins, mid, lg, lc, rg, rc)) {
            arena[it.slot].box = it.geom;
            arena[it.slot].first = it.lo;
            arena[it.slot].count = it.hi - it.lo;
            return;
        }
        uint32_t pair = arena_used.fetch_add(2);
        arena[it.slot].box = it.geom;
        arena[it.slot].first = pair;
        arena[it.slot].count = 0;
        WorkItem right{pair + 1, mid, it.hi, it.depth + 1, rg, rc};
        WorkItem left{pair, it.lo, mid, it.depth + 1, lg, lc};
        auto route = [&](const WorkItem& w) {
            if (w.hi - w.lo >= kSharedWorkMin) {
                std::lock_guard<std::mutex> g(mu);
                shared.push_back(w);
                ++outstanding;
                cv.notify_one();
            } else {
                local.push_back(w);
            }
        };
        route(right);
        route(left);
    }

    void worker() {
        BinSet bins;
        std::vector<WorkItem> local;
        local.reserve(128);
        for (;;) {
            WorkItem it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !shared.empty() || outstanding == 0; });
                if (shared.empty()) return;
                it = shared.back();
                shared.pop_back();
            }
            local.push_back(it);
            while (!local.empty()) {
                WorkItem w = local.back();
                local.pop_back();
                process(w, bins, local);
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (--outstanding == 0) cv.notify_all();
            }
        }
    }

    void run(uint32_t n_threads, const Box& g, const Box& c) {
        shared.push_back(WorkItem{0, 0, uint32_t(refs.size()), 0, g, c});
        outstanding = 1;
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back([this] { worker(); });
        worker();
        for (auto& th : pool) th.join();
    }
};

}  // namespace

extern "C" int rt_bvh_build(rt_triangle* tris, uint32_t n_tri, const rt_bvh_params* params, rt_bvh_node* nodes,
                            uint32_t* n_nodes, rt_bvh_stats* stats) {
    if (!nodes || !n_nodes || (n_tri && !tris)) return rtint::fail(nullptr, RT_ERR_INVALID, "rt_bvh_build: null argument");
    auto t0 = std::chrono::steady_clock::now();
    rt_bvh_params cfg{1.0f, 1.0f, 2, 8, 0};
    if (params) cfg = *params;
    if (cfg.min_leaf == 0) cfg.min_leaf = 1;
    if (cfg.max_leaf < cfg.min_leaf) cfg.max_leaf = cfg.min_leaf;
    rt_bvh_stats st{};

    if (n_tri == 0) {
        // inverted bounds never intersect: traversal ends at the root (bin_sah.rs:138-142)
        rt_bvh_node root{{1e30f, 1e30f, 1e30f}, 0, {-1e30f, -1e30f, -1e30f}, 0};
        nodes[0] = root;
        *n_nodes = 1;
        st.node_count = 1;
        if (stats) *stats = st;
        return RT_OK;
    }

    std::vector<V3> pmin(n_tri), pmax(n_tri);
    Box g, c;
    for (uint32_t i = 0; i < n_tri; ++i) {
        const rt_triangle& t = tris[i];
        V3 a{t.v[0][0], t.v[0][1], t.v[0][2]}, b{t.v[1][0], t.v[1][1], t.v[1][2]}, d{t.v[2][0], t.v[2][1], t.v[2][2]};
        pmin[i] = vmin(a, vmin(b, d));
        pmax[i] = vmax(a, vmax(b, d));
        g.grow(pmin[i], pmax[i]);
        c.grow(vmul(vadd(pmin[i], pmax[i]), 0.5f));
    }
    std::vector<uint32_t> refs(n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) refs[i] = i;

    uint32_t n_threads = cfg.threads ? cfg.threads : std::max(1u, std::thread::hardware_concurrency());
    n_threads = std::min(n_threads, 64u);
    Builder b(cfg, pmin, pmax, refs);
    b.run(n_threads, g, c);

    // depth-first flatten: children adjacent, subtrees contiguous, leaf ranges re-emitted in visit order
    std::vector<uint32_t> order;
    order.reserve(n_tri);
    uint32_t out_n = 1;
    const float root_h = std::max(b.arena[0].box.half_area(), 1e-30f);
    float sah = 0.0f;
    struct Visit { uint32_t src, dst, depth; };
    std::vector<Visit> stack;
    stack.push_back({0, 0, 0});
    while (!stack.empty()) {
        Visit v = stack.back();
        stack.pop_back();
        const TmpNode& tn = b.arena[v.src];
        st.max_depth = std::max(st.max_depth, v.depth);
        float rel = tn.box.half_area() / root_h;
        rt_bvh_node& o = nodes[v.dst];
        o.min[0] = tn.box.lo.x; o.min[1] = tn.box.lo.y; o.min[2] = tn.box.lo.z;
        o.max[0] = tn.box.hi.x; o.max[1] = tn.box.hi.y; o.max[2] = tn.box.hi.z;
        if (tn.count != 0) {
            o.idx = uint32_t(order.size());
            o.amt = tn.count;
            for (uint32_t k = 0; k < tn.count; ++k) order.push_back(refs[tn.first + k]);
            st.leaf_count++;
            st.max_leaf_size = std::max(st.max_leaf_size, tn.count);
            sah += rel * float(tn.count) * cfg.intersect_cost;
        } else {
            uint32_t left = out_n;
            out_n += 2;
            o.idx = left;
            o.amt = 0;
            sah += rel * cfg.traversal_cost;
            stack.push_back({tn.first + 1, left + 1, v.depth + 1});
            stack.push_back({tn.first, left, v.depth + 1});
        }
    }
    if (st.max_depth > kMaxDepth) return rtint::fail(nullptr, RT_ERR_UNSUPPORTED, "rt_bvh_build: depth exceeds 64");

    std::vector<rt_triangle> src(tris, tris + n_tri);
    for (uint32_t k = 0; k < n_tri; ++k) tris[k] = src[order[k]];

    *n_nodes = out_n;
    st.node_count = out_n;
    st.sah_cost = sah;
    st.avg_leaf_size = float(order.size()) / float(std::max(1u, st.leaf_count));
    st.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return RT_OK;
}
