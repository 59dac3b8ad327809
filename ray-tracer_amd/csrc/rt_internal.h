// Internal declarations shared by the translation units of librt_hip.so. Not part of the ABI.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "rt_api.h"

namespace rtint {

// Records `msg` for rt_last_error (in the context when there is one, else thread-local) and returns `code`.
int fail(rt_ctx* ctx, int code, const std::string& msg);
void set_ctx_error(rt_ctx* ctx, const std::string& msg);  // defined next to the context type
const char* thread_error();                               // message of the last context-less failure on this thread

// ---- device-side data layout, produced on the host from the wire-format scene ----

// 64-byte record for one interior node: the boxes of BOTH children plus how to continue into each,
// so one visit costs four 16-byte loads instead of two dependent 32-byte node fetches.
//   q0 = {l.min.x, l.min.y, l.min.z, l.max.x}   q1 = {l.max.y, l.max.z, r.min.x, r.min.y}
//   q2 = {r.min.z, r.max.x, r.max.y, r.max.z}   q3 = {left_ref, right_ref, 0, 0}
// A child reference is either the index of another record (bit 31 clear) or a leaf
// (bit 31 set, bits 30..25 = triangle count - 1, bits 24..0 = first triangle).
struct alignas(16) NodePair {
    float q0[4], q1[4], q2[4];
    uint32_t ref[4];
};
static_assert(sizeof(NodePair) == 64, "NodePair");

// 128-byte record of the 4-wide tree the occlusion queries walk (112 bytes used): an interior node of the binary tree
// with each interior child replaced by that child's own two children, when their boxes lie inside the child's box.
// A ray's plane distances are monotonic in the plane coordinate, so the slab interval of a box nested in another is
// nested in the other's interval in floating point as well: testing the grandchildren directly admits exactly the
// boxes the two-step test admits. An occlusion query ("is any triangle hit below tmax") prunes against a constant and
// does not depend on visiting order, so it gives the same answer on either tree.
//   box[k] = {min.xyz, max.xyz} of entry k (all NaN for an unused entry: never hit), ref[k] as in NodePair but
//   indexing this array.
struct alignas(16) NodeQuad {
    float box[4][6];
    uint32_t ref[4];
    uint32_t pad[4];
};
static_assert(sizeof(NodeQuad) == 128, "NodeQuad");

// 64-byte record of the compressed 4-wide tree (same topology, record order and references as NodeQuad): child boxes as
// 8-bit offsets from `p` in steps of `s`, grown by GpuScene::box_margin and then rounded outwards to the step grid.
//   box k spans [p + lo[a].byte(k) * s[a], p + hi[a].byte(k) * s[a]] on axis a; an unused entry has lo byte 255 and hi byte 0 (empty for either ray sign) and reference kRefNone.
struct alignas(16) NodeCw {
    float p[3];
    float sx;
    uint32_t lo[3];
    uint32_t hix;
    uint32_t hiy, hiz;
    float sy, sz;
    uint32_t ref[4];
};
static_assert(sizeof(NodeCw) == 64, "NodeCw");

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafMaxCount = 64;
constexpr uint32_t kLeafMaxFirst = 1u << 25;
constexpr uint32_t kRefNone = 0x7FFFFFFFu;  // "empty scene" root marker: nothing to traverse

inline uint32_t make_leaf_ref(uint32_t first, uint32_t count) { return kLeafBit | ((count - 1u) << 25) | first; }

// 48-byte intersection record: {v0, n.x} {v1-v0, n.y} {v2-v0, n.z}, n = unit plane normal.
struct alignas(16) TriIsect {
    float a[4], b[4], c[4];
};
// 64-byte shading record, ordered so that a vertex loads only the 16-byte chunks it needs:
//   chunk 0 {n0.xyz, matword}   chunk 1 {n1.xyz, n2.x}   chunk 2 {n2.yz, uv0.xy}   chunk 3 {uv1.xy, uv2.xy}
// flat untextured triangles need chunk 0 (n0 holds their plane normal), smooth ones chunks 0-2, textured ones all four.
// matword = material index | bit 31: spec_trans > 0.5 | bit 30: shade_smooth.
struct alignas(16) TriShade {
    float n0[3];
    uint32_t matword;
    float n1[3];
    float n2x;
    float n2yz[2];
    float uv0[2];
    float uv1[2];
    float uv2[2];
};
static_assert(sizeof(TriIsect) == 48 && sizeof(TriShade) == 64, "triangle streams");

// Node records placed at the front of the record array in order of decreasing surface area (the traversal kernels copy
// as many of them into LDS as their LDS budget allows).
constexpr uint32_t kHotNodeSlots = 2048;

struct GpuScene {
    std::vector<NodePair> nodes;
    uint32_t root_ref = kRefNone;
    uint32_t n_hot = 0;          // records [0, n_hot) are the largest-area interior nodes, largest first
    std::vector<NodeQuad> wide;  // 4-wide tree for occlusion queries (empty when the root is a leaf)
    uint32_t wide_root = kRefNone;
    uint32_t n_hot_wide = 0;     // wide records [0, n_hot_wide) are the largest-area ones, largest first
    std::vector<NodeCw> cw;      // compressed twin of `wide`, record for record
    float scene_size = 0.0f;     // largest absolute coordinate of the root bounds
    float box_margin = 0.0f;     // every compressed box contains its source box grown by this much: 2^-17 * scene_size
    bool cw_complete = true;     // false when a box with an infinite bound could not be represented: the compressed tree must not be used
    std::vector<TriIsect> isect;
    std::vector<TriShade> shade;
};

// Converts reference-layout nodes/triangles into the device layout. Returns RT_OK or an rt_status
// (with `err` filled) when the input violates the traversal contract or a device-layout limit.
int convert_scene(const rt_bvh_node* nodes, uint32_t n_nodes, const rt_triangle* tris, uint32_t n_tri,
                  const rt_material* mats, uint32_t n_mat, GpuScene& out, std::string& err);

// Record order by measured use. `visits[i]` = how often record i was visited; the `n_hot` most visited records move to
// slots 0 .. n_hot-1, most visited first (ties: lower old slot first), and the records they push out of that prefix take
// the slots the newcomers left; every other record stays where it is. Fills new_of_old (old slot -> new slot) and returns
// the number of records that moved.
uint32_t order_by_visits(const uint32_t* visits, size_t n_records, uint32_t n_hot, std::vector<uint32_t>& new_of_old);

// Applies such a permutation to a record array: records move to their new slots and every reference to a record
// (bit 31 clear, not kRefNone) in the first `refs_per_record` words of Rec::ref is renumbered. Leaf references are
// untouched.
template <class Rec>
void permute_records(std::vector<Rec>& recs, const std::vector<uint32_t>& new_of_old, int refs_per_record) {
    std::vector<Rec> out(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) {
        Rec r = recs[i];
        for (int k = 0; k < refs_per_record; ++k)
            if (!(r.ref[k] & kLeafBit) && r.ref[k] != kRefNone && r.ref[k] < recs.size()) r.ref[k] = new_of_old[r.ref[k]];
        out[new_of_old[i]] = r;
    }
    recs.swap(out);
}

}  // namespace rtint
