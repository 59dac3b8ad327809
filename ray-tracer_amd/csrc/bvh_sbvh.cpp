// Host-side BVH construction with spatial splits (SBVH, Stich et al. 2009) on top of binned SAH.
//
// Behavioural twin of the reference's `BinSah` with `SplitPolicy::Spatial` (what `Scene::build_bvh` uses,
// src/scene/mod.rs:84-115; builder at src/accelerators/bvh_based/bin_sah.rs:862-1450): per node a 16-bin
// object-split candidate over reference centroids, a 16-bin "chopped" spatial candidate (the triangle polygon
// is clipped against the bin planes) evaluated only when the object-split children overlap by more than
// alpha x root area, the cheaper of the two wins, straddling references are duplicated while a duplication
// budget lasts, leaves de-duplicate their primitive ids, and the DFS flatten re-emits leaf ranges so that
// duplicated triangles simply appear more than once in the output triangle array.
//
// Differences from the reference, on purpose: its budget is one atomic counter raced by rayon tasks, so which
// subtree gets the last duplications depends on timing; here every node carries its own share of the budget
// (children inherit the remainder in proportion to their reference counts), which makes the tree a pure
// function of the input whatever the thread count.  The thread pool is the same shared-work-list scheme the
// object-split builder of this library uses.

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
constexpr size_t kSharedWorkMin = 8 * 1024;
constexpr size_t kChunkParallelMin = 64 * 1024;

// What the builder sorts: a triangle id with the part of its bounds that lies in the node being split (spatial splits
// hand one triangle to both children, each with its own clipped part).
struct Fragment {
    V3 lo, hi;
    uint32_t prim;
    V3 centroid() const { return vmul(vadd(lo, hi), 0.5f); }
};

struct ObjBins {
    Box box[3][kBins];
    uint32_t count[3][kBins];
    ObjBins() { std::memset(count, 0, sizeof(count)); }
    void merge(const ObjBins& o) {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < kBins; ++b) {
                box[a][b].grow(o.box[a][b].lo, o.box[a][b].hi);
                count[a][b] += o.count[a][b];
            }
    }
};

struct SpatBins {
    Box box[kBins];
    uint32_t entry[kBins], exit[kBins];
    SpatBins() { std::memset(entry, 0, sizeof(entry)); std::memset(exit, 0, sizeof(exit)); }
    void merge(const SpatBins& o) {
        for (int b = 0; b < kBins; ++b) {
            box[b].grow(o.box[b].lo, o.box[b].hi);
            entry[b] += o.entry[b];
            exit[b] += o.exit[b];
        }
    }
};

// Convex polygon from a triangle chopped by parallel planes: at most 5 vertices per piece, 8 for on-plane cases.
struct Poly {
    V3 v[8];
    int n = 0;
    void push(V3 p) { if (n < 8) v[n++] = p; }
    // Sutherland-Hodgman against v[axis] = pos: (part <= pos, part >= pos)   (bin_sah.rs:1209-1238)
    void split(int axis, float pos, Poly& left, Poly& right) const {
        left.n = right.n = 0;
        for (int e = 0; e < n; ++e) {
            V3 p = v[e], q = v[(e + 1) % n];
            float dp = p[axis] - pos, dq = q[axis] - pos;
            if (dp <= 0.0f) left.push(p);
            if (dp >= 0.0f) right.push(p);
            if ((dp < 0.0f && dq > 0.0f) || (dp > 0.0f && dq < 0.0f)) {
                V3 m = vadd(p, vmul(vsub(q, p), dp / (dp - dq)));
                left.push(m);
                right.push(m);
            }
        }
    }
    // polygon bounds intersected with the parent reference box; false when empty   (bin_sah.rs:1254-1266)
    bool clip_box(V3 rlo, V3 rhi, V3& lo, V3& hi) const {
        if (n == 0) return false;
        lo = hi = v[0];
        for (int i = 1; i < n; ++i) { lo = vmin(lo, v[i]); hi = vmax(hi, v[i]); }
        lo = vmax(lo, rlo);
        hi = vmin(hi, rhi);
        return lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z;
    }
};

struct TmpNode {
    Box box;
    uint32_t first = 0, count = 0;  // count != 0: leaf over pool[first, first+count); else children at first, first+1
};

struct Task {
    uint32_t slot = 0, depth = 0;
    int64_t budget = 0;
    std::vector<Fragment> refs;
    Box geom, cent;
};

struct SpatialBuilder {
    rt_bvh_params cfg;
    float alpha;
    const rt_triangle* tris;
    uint32_t n_threads;
    float root_half = 1.0f;

    std::vector<TmpNode> arena;
    std::atomic<uint32_t> arena_used{1};
    std::vector<uint32_t> pool;
    std::atomic<size_t> pool_used{0};

    std::mutex mu;
    std::condition_variable cv;
    std::vector<Task> shared;
    int outstanding = 0;

    Poly tri_poly(uint32_t prim) const {
        const rt_triangle& t = tris[prim];
        Poly p;
        p.n = 3;
        for (int k = 0; k < 3; ++k) p.v[k] = V3{t.v[k][0], t.v[k][1], t.v[k][2]};
        return p;
    }

    // Runs fn(begin, end, chunk_index) over [0, n); big ranges are spread over short-lived helper threads (the pool's
    // other workers are idle while the first few huge nodes are processed). Chunk boundaries depend only on n.
    template <class F>
    void for_chunks(size_t n, size_t n_chunks, F fn) const {
        if (n_chunks <= 1) { fn(size_t(0), n, size_t(0)); return; }
        std::vector<std::thread> th;
        size_t per = (n + n_chunks - 1) / n_chunks;
        for (size_t c = 1; c < n_chunks; ++c) {
            size_t b = std::min(n, c * per), e = std::min(n, b + per);
            th.emplace_back([=] { fn(b, e, c); });
        }
        fn(size_t(0), std::min(n, per), size_t(0));
        for (auto& t : th) t.join();
    }
    size_t chunks_for(size_t n) const { return n >= kChunkParallelMin ? std::min<size_t>(n_threads, 16) : 1; }


    void emit_leaf(const Task& t) {
        // both halves of a split primitive can land in one leaf: test it once   (bin_sah.rs:1060-1073)
        std::vector<uint32_t> ids(t.refs.size());
        for (size_t i = 0; i < ids.size(); ++i) ids[i] = t.refs[i].prim;
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        size_t at = pool_used.fetch_add(ids.size());
        std::copy(ids.begin(), ids.end(), pool.begin() + at);
        arena[t.slot].box = t.geom;
        arena[t.slot].first = uint32_t(at);
        arena[t.slot].count = uint32_t(ids.size());
    }

    struct ObjectSplit { float cost; int axis, bin; Box left_bounds, right_bounds; };
    struct PlaneSplit { float cost; int axis; float plane; };

    bool object_candidate(const Task& t, V3 scale, ObjectSplit& out) const {
        size_t nc = chunks_for(t.refs.size());
        std::vector<ObjBins> part(nc);
        for_chunks(t.refs.size(), nc, [&](size_t b, size_t e, size_t c) {
            ObjBins& bins = part[c];
            for (size_t k = b; k < e; ++k) {
                const Fragment& r = t.refs[k];
                int bi[3];
                bin_of(r.centroid(), t.cent.lo, scale, bi);
                for (int a = 0; a < 3; ++a) { bins.box[a][bi[a]].grow(r.lo, r.hi); bins.count[a][bi[a]]++; }
            }
        });
        for (size_t c = 1; c < nc; ++c) part[0].merge(part[c]);
        const ObjBins& bins = part[0];

        float best = kInf;
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
        if (best_axis < 0) return false;
        out.cost = best;
        out.axis = best_axis;
        out.bin = best_bin;
        out.left_bounds = Box{};
        out.right_bounds = Box{};
        for (int b = 0; b < best_bin; ++b) out.left_bounds.grow(bins.box[best_axis][b].lo, bins.box[best_axis][b].hi);
        for (int b = best_bin; b < kBins; ++b) out.right_bounds.grow(bins.box[best_axis][b].lo, bins.box[best_axis][b].hi);
        return true;
    }

    // chopped binning along one axis   (bin_sah.rs:1268-1317)
    void spatial_bins(const Task& t, size_t b0, size_t e0, int axis, float base, float width, float inv, SpatBins& sb) const {
        const int top = kBins - 1;
        for (size_t k = b0; k < e0; ++k) {
            const Fragment& r = t.refs[k];
            int lo = clamp_bin((r.lo[axis] - base) * inv);
            int hi = std::max(clamp_bin((r.hi[axis] - base) * inv), lo);
            if (lo == hi) {
                sb.box[lo].grow(r.lo, r.hi);
                sb.entry[lo]++;
                sb.exit[lo]++;
                continue;
            }
            Poly poly = tri_poly(r.prim), lp, rp;
            V3 mn, mx;
            for (int b = lo; b < hi; ++b) {
                float plane = base + width * (float(b) + 1.0f);
                poly.split(axis, plane, lp, rp);
                if (lp.clip_box(r.lo, r.hi, mn, mx)) sb.box[b].grow(mn, mx);
                poly = rp;
                if (poly.n == 0) break;
            }
            if (poly.clip_box(r.lo, r.hi, mn, mx)) sb.box[hi].grow(mn, mx);
            sb.entry[lo]++;
            sb.exit[hi]++;
        }
        (void)top;
    }

    bool spatial_candidate(const Task& t, PlaneSplit& out) const {
        V3 ext = vmax(vsub(t.geom.hi, t.geom.lo), V3{0, 0, 0});
        bool found = false;
        for (int axis = 0; axis < 3; ++axis) {
            if (ext[axis] <= 1e-12f) continue;
            float width = ext[axis] / float(kBins);
            float inv = float(kBins) * (1.0f - 1e-6f) / ext[axis];
            float base = t.geom.lo[axis];
            size_t nc = chunks_for(t.refs.size());
            std::vector<SpatBins> part(nc);
            for_chunks(t.refs.size(), nc,
                       [&](size_t b, size_t e, size_t c) { spatial_bins(t, b, e, axis, base, width, inv, part[c]); });
            for (size_t c = 1; c < nc; ++c) part[0].merge(part[c]);
            const SpatBins& sb = part[0];

            float right_h[kBins];
            uint32_t right_n[kBins];
            Box acc;
            uint32_t cnt = 0;
            for (int i = kBins - 1; i >= 1; --i) {
                acc.grow(sb.box[i].lo, sb.box[i].hi);
                cnt += sb.exit[i];
                right_h[i] = acc.half_area();
                right_n[i] = cnt;
            }
            acc = Box{};
            cnt = 0;
            for (int i = 1; i < kBins; ++i) {
                acc.grow(sb.box[i - 1].lo, sb.box[i - 1].hi);
                cnt += sb.entry[i - 1];
                if (cnt == 0 || right_n[i] == 0) continue;
                float cost = float(cnt) * acc.half_area() + float(right_n[i]) * right_h[i];
                if (!found || cost < out.cost) {
                    found = true;
                    out.cost = cost;
                    out.axis = axis;
                    out.plane = base + width * float(i);
                }
            }
        }
        return found;
    }

    struct Side {
        std::vector<Fragment> refs;
        Box geom, cent;
        void push(V3 lo, V3 hi, uint32_t prim) {
            geom.grow(lo, hi);
            cent.grow(vmul(vadd(lo, hi), 0.5f));
            refs.push_back(Fragment{lo, hi, prim});
        }
    };

    // bin_sah.rs:1383-1450; `budget` is this node's own allowance and is decremented per duplication.
    bool distribute_spatial(const Task& t, int axis, float plane, int64_t& budget, Side& l, Side& r) const {
        l.refs.reserve(t.refs.size());
        r.refs.reserve(t.refs.size());
        for (const Fragment& rf : t.refs) {
            if (rf.hi[axis] <= plane) { l.push(rf.lo, rf.hi, rf.prim); continue; }
            if (rf.lo[axis] >= plane) { r.push(rf.lo, rf.hi, rf.prim); continue; }
            bool whole = true;
            if (budget > 0) {
                Poly poly = tri_poly(rf.prim), lp, rp;
                poly.split(axis, plane, lp, rp);
                V3 lmn, lmx, rmn, rmx;
                bool hl = lp.clip_box(rf.lo, rf.hi, lmn, lmx), hr = rp.clip_box(rf.lo, rf.hi, rmn, rmx);
                if (hl && hr) {
                    --budget;
                    l.push(lmn, lmx, rf.prim);
                    r.push(rmn, rmx, rf.prim);
                    whole = false;
                } else if (hl) {
                    l.push(lmn, lmx, rf.prim);
                    whole = false;
                } else if (hr) {
                    r.push(rmn, rmx, rf.prim);
                    whole = false;
                }
            }
            if (whole) {  // out of budget or degenerate clip: whole reference to the side of its centre
                float cen = (rf.lo[axis] + rf.hi[axis]) * 0.5f;
                if (cen < plane) l.push(rf.lo, rf.hi, rf.prim);
                else r.push(rf.lo, rf.hi, rf.prim);
            }
        }
        return !l.refs.empty() && !r.refs.empty();
    }

    static void bounds_of(const std::vector<Fragment>& refs, Box& g, Box& c) {
        g = Box{}; c = Box{};
        for (const Fragment& r : refs) { g.grow(r.lo, r.hi); c.grow(r.centroid()); }
    }

    void median_split(Task& t, Side& l, Side& r) const {
        size_t n = t.refs.size(), mid = n / 2;
        V3 cext = vmax(vsub(t.cent.hi, t.cent.lo), V3{0, 0, 0});
        int axis = std::max(cext.x, std::max(cext.y, cext.z)) > 0.0f
                       ? widest_axis(cext)
                       : widest_axis(vmax(vsub(t.geom.hi, t.geom.lo), V3{0, 0, 0}));
        std::nth_element(t.refs.begin(), t.refs.begin() + mid, t.refs.end(), [&](const Fragment& p, const Fragment& q) {
            float a = (p.lo[axis] + p.hi[axis]) * 0.5f, b = (q.lo[axis] + q.hi[axis]) * 0.5f;
            if (a != b) return a < b;
            if (p.prim != q.prim) return p.prim < q.prim;
            return p.lo[axis] < q.lo[axis];
        });
        l.refs.assign(t.refs.begin(), t.refs.begin() + mid);
        r.refs.assign(t.refs.begin() + mid, t.refs.end());
        bounds_of(l.refs, l.geom, l.cent);
        bounds_of(r.refs, r.geom, r.cent);
    }

    // One node (bin_sah.rs:952-1040). Children go to `local` or the shared list.
    void process(Task& t, std::vector<Task>& local) {
        const size_t n = t.refs.size();
        if (n <= cfg.min_leaf || t.depth >= kMaxDepth) { emit_leaf(t); return; }

        V3 scale = bin_scale(t.cent);
        ObjectSplit obj{};
        bool has_obj = (scale.x > 0.0f || scale.y > 0.0f || scale.z > 0.0f) && object_candidate(t, scale, obj);

        bool try_spatial = t.budget > 0 && (!has_obj || intersect(obj.left_bounds, obj.right_bounds).half_area() / root_half > alpha);
        PlaneSplit spa{};
        bool has_spa = try_spatial && spatial_candidate(t, spa);

        float parent_h = std::max(t.geom.half_area(), 1e-30f);
        auto norm = [&](float c) { return cfg.traversal_cost + cfg.intersect_cost * c / parent_h; };
        float obj_cost = has_obj ? norm(obj.cost) : kInf, spa_cost = has_spa ? norm(spa.cost) : kInf;
        bool prefer_spatial = has_spa && (!has_obj || spa_cost < obj_cost);
        float best = std::min(obj_cost, spa_cost);

        Side l, r;
        int64_t budget = t.budget;
        bool done = false;
        if (!((has_obj || has_spa) && best < float(n) * cfg.intersect_cost)) {
            if (n <= cfg.max_leaf) { emit_leaf(t); return; }
            median_split(t, l, r);
            done = true;
        }
        if (!done && prefer_spatial) {
            int64_t b = budget;
            if (distribute_spatial(t, spa.axis, spa.plane, b, l, r)) { budget = b; done = true; }
            else { l = Side{}; r = Side{}; }
        }
        if (!done && has_obj) {
            for (const Fragment& rf : t.refs) {
                int bi[3];
                bin_of(rf.centroid(), t.cent.lo, scale, bi);
                Side& s = bi[obj.axis] < obj.bin ? l : r;
                s.refs.push_back(rf);
                s.cent.grow(rf.centroid());
            }
            if (!l.refs.empty() && !r.refs.empty()) { l.geom = obj.left_bounds; r.geom = obj.right_bounds; done = true; }
            else { l = Side{}; r = Side{}; }
        }
        if (!done) median_split(t, l, r);

        std::vector<Fragment>().swap(t.refs);
        uint32_t pair = arena_used.fetch_add(2);
        arena[t.slot].box = t.geom;
        arena[t.slot].first = pair;
        arena[t.slot].count = 0;

        // the children inherit what is left of the allowance in proportion to their reference counts
        int64_t bl = budget > 0 ? int64_t((__int128)budget * (int64_t)l.refs.size() / (int64_t)(l.refs.size() + r.refs.size())) : 0;
        Task right, left;
        right.slot = pair + 1; right.depth = t.depth + 1; right.budget = std::max<int64_t>(budget, 0) - bl;
        right.refs = std::move(r.refs); right.geom = r.geom; right.cent = r.cent;
        left.slot = pair; left.depth = t.depth + 1; left.budget = bl;
        left.refs = std::move(l.refs); left.geom = l.geom; left.cent = l.cent;
        auto route = [&](Task&& w) {
            if (w.refs.size() >= kSharedWorkMin) {
                std::lock_guard<std::mutex> g(mu);
                shared.push_back(std::move(w));
                ++outstanding;
                cv.notify_one();
            } else {
                local.push_back(std::move(w));
            }
        };
        route(std::move(right));
        route(std::move(left));
    }

    void worker() {
        std::vector<Task> local;
        for (;;) {
            Task it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !shared.empty() || outstanding == 0; });
                if (shared.empty()) return;
                it = std::move(shared.back());
                shared.pop_back();
            }
            local.push_back(std::move(it));
            while (!local.empty()) {
                Task w = std::move(local.back());
                local.pop_back();
                process(w, local);
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (--outstanding == 0) cv.notify_all();
            }
        }
    }

    void run(Task&& root) {
        shared.push_back(std::move(root));
        outstanding = 1;
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < n_threads; ++t) th.emplace_back([this] { worker(); });
        worker();
        for (auto& x : th) x.join();
    }
};

}  // namespace

extern "C" int rt_bvh_build_spatial(const rt_triangle* tris, uint32_t n_tri, const rt_bvh_params* params, float alpha,
                                    float budget_fraction, rt_triangle* tris_out, uint32_t tri_capacity, uint32_t* n_tris_out,
                                    rt_bvh_node* nodes, uint32_t node_capacity, uint32_t* n_nodes, rt_bvh_stats* stats) {
    if (!nodes || !n_nodes || !n_tris_out || (n_tri && (!tris || !tris_out)))
        return rtint::fail(nullptr, RT_ERR_INVALID, "rt_bvh_build_spatial: null argument");
    if (tris == tris_out && n_tri) return rtint::fail(nullptr, RT_ERR_INVALID, "rt_bvh_build_spatial: input and output alias");
    auto t0 = std::chrono::steady_clock::now();
    rt_bvh_params cfg{1.0f, 1.0f, 2, 8, 0};
    if (params) cfg = *params;
    if (cfg.min_leaf == 0) cfg.min_leaf = 1;
    if (cfg.max_leaf < cfg.min_leaf) cfg.max_leaf = cfg.min_leaf;
    rt_bvh_stats st{};

    if (n_tri == 0) {
        if (node_capacity < 1) return rtint::fail(nullptr, RT_ERR_INVALID, "rt_bvh_build_spatial: node capacity");
        nodes[0] = rt_bvh_node{{1e30f, 1e30f, 1e30f}, 0, {-1e30f, -1e30f, -1e30f}, 0};
        *n_nodes = 1;
        *n_tris_out = 0;
        st.node_count = 1;
        if (stats) *stats = st;
        return RT_OK;
    }

    const size_t extra = size_t(double(n_tri) * double(std::max(budget_fraction, 0.0f)));   // bin_sah.rs:909
    const size_t ref_cap = size_t(n_tri) + extra;
    if (tri_capacity < ref_cap || size_t(node_capacity) < 2 * ref_cap)
        return rtint::fail(nullptr, RT_ERR_INVALID,
                           "rt_bvh_build_spatial: need room for n + floor(n*budget) triangles and twice that many nodes");

    SpatialBuilder b;
    b.cfg = cfg;
    b.alpha = alpha;
    b.tris = tris;
    b.n_threads = std::min(cfg.threads ? cfg.threads : std::max(1u, std::thread::hardware_concurrency()), 64u);
    b.arena.resize(2 * ref_cap + 2);
    b.pool.resize(ref_cap);

    Task root;
    root.refs.resize(n_tri);
    for (uint32_t i = 0; i < n_tri; ++i) {
        const rt_triangle& t = tris[i];
        V3 a{t.v[0][0], t.v[0][1], t.v[0][2]}, c{t.v[1][0], t.v[1][1], t.v[1][2]}, d{t.v[2][0], t.v[2][1], t.v[2][2]};
        Fragment r{vmin(a, vmin(c, d)), vmax(a, vmax(c, d)), i};
        root.refs[i] = r;
        root.geom.grow(r.lo, r.hi);
        root.cent.grow(r.centroid());
    }
    root.budget = int64_t(extra);
    b.root_half = std::max(root.geom.half_area(), 1e-30f);
    b.run(std::move(root));

    // depth-first flatten; a duplicated primitive is emitted once per leaf that holds it   (bin_sah.rs:1462-1526)
    uint32_t out_n = 1, out_t = 0;
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
            o.idx = out_t;
            o.amt = tn.count;
            for (uint32_t k = 0; k < tn.count; ++k) tris_out[out_t++] = tris[b.pool[tn.first + k]];
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
    if (st.max_depth > kMaxDepth) return rtint::fail(nullptr, RT_ERR_UNSUPPORTED, "rt_bvh_build_spatial: depth exceeds 64");

    *n_nodes = out_n;
    *n_tris_out = out_t;
    st.node_count = out_n;
    st.sah_cost = sah;
    st.avg_leaf_size = float(out_t) / float(std::max(1u, st.leaf_count));
    st.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    return RT_OK;
}
