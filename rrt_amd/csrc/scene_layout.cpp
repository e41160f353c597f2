#include "scene_layout.h"

#include <algorithm>
#include <cmath>
#include <thread>
#include <cstring>

namespace rrt_amd {

namespace {
// fn(begin, end) over [0, n) on a few host threads when the range is big enough to pay for them
template <class Fn>
void parallel_ranges(size_t n, Fn fn) {
    const size_t n_threads = n < (size_t(1) << 16) ? 1 : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (n_threads == 1) { fn(size_t(0), n); return; }
    std::vector<std::thread> pool;
    const size_t step = (n + n_threads - 1) / n_threads;
    for (size_t b = 0; b < n; b += step) pool.emplace_back([=] { fn(b, std::min(n, b + step)); });
    for (std::thread &t : pool) t.join();
}
}  // namespace
namespace {

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline bool is_neg_zero(float f) { return f2u(f) == 0x80000000u; }
inline bool is_leaf(const rrt_bvh_node &n) { return (n.right_axis_leaf & RRT_NODE_LEAF_BIT) != 0; }

}  // namespace

int build_scene_layout(const rrt_sphere_soa *sp, const rrt_material_soa *mt, const rrt_bvh_node *nodes, uint32_t n_nodes,
                       SceneLayout &out, std::string &err) {
    if (!sp || !mt || !nodes) { err = "null scene argument"; return RRT_ERR_INVALID_ARGUMENT; }
    if (sp->n == 0 || n_nodes == 0) {
        // Bvh::hit indexes nodes[0] unconditionally (acceleration.rs:233,237): an empty scene panics there.
        err = "empty scene: the reference cannot render it either (Bvh::hit indexes nodes[0])";
        return RRT_ERR_BAD_SCENE;
    }
    if (n_nodes > kRefIndexMask) { err = "too many BVH nodes"; return RRT_ERR_BAD_SCENE; }
    if (sp->n > kInfoMatMask || mt->n > kInfoMatMask) { err = "too many spheres or materials"; return RRT_ERR_BAD_SCENE; }
    if (!sp->c0x || !sp->c0y || !sp->c0z || !sp->c1x || !sp->c1y || !sp->c1z || !sp->time0 || !sp->time1 || !sp->radius ||
        !sp->material_id) { err = "null sphere plane"; return RRT_ERR_INVALID_ARGUMENT; }
    if (mt->n == 0 || !mt->kind || !mt->albedo_r || !mt->albedo_g || !mt->albedo_b || !mt->fuzz || !mt->ref_idx) {
        err = "null material plane"; return RRT_ERR_INVALID_ARGUMENT;
    }

    const uint32_t ns = sp->n;
    out = SceneLayout();
    out.n_spheres = ns;
    out.n_nodes = n_nodes;

    // ---- materials ----
    out.mat.resize(mt->n);
    for (uint32_t m = 0; m < mt->n; m++) {
        if (mt->kind[m] > RRT_MAT_DIELECTRIC) { err = "unknown material kind"; return RRT_ERR_BAD_SCENE; }
        const float param = mt->kind[m] == RRT_MAT_METAL ? mt->fuzz[m] : (mt->kind[m] == RRT_MAT_DIELECTRIC ? mt->ref_idx[m] : 0.f);
        out.mat[m] = Float4{mt->albedo_r[m], mt->albedo_g[m], mt->albedo_b[m], param};
    }

    // ---- spheres ----
    out.sph.resize(ns); out.motion.resize(ns); out.motion_dt.resize(ns); out.sph_info.resize(ns); out.prim_att.resize(ns);
    bool any_moving = false, uniform = true, bounded = true, radii_normal = true;
    uint32_t ut0 = 0, ut1 = 0;
    for (uint32_t i = 0; i < ns; i++) {
        const uint32_t m = sp->material_id[i];
        if (m >= mt->n) { err = "sphere material_id out of range"; return RRT_ERR_BAD_SCENE; }
        const float c0[3] = {sp->c0x[i], sp->c0y[i], sp->c0z[i]};
        const float c1[3] = {sp->c1x[i], sp->c1y[i], sp->c1z[i]};
        // Sphere::center_at (sphere.rs:61-64) returns centre0 bit for bit when centre1 == centre0, the times
        // are the (0, 1) of Sphere::from (sphere.rs:57-59) and no component is -0.0 (which "+ 0" would turn
        // into +0.0).  Only those spheres skip the interpolation; every other sphere evaluates it in full.
        bool still = f2u(sp->time0[i]) == 0u && sp->time1[i] == 1.0f;
        for (int k = 0; k < 3; k++) still = still && f2u(c0[k]) == f2u(c1[k]) && !is_neg_zero(c0[k]);
        for (int k = 0; k < 3; k++) bounded = bounded && std::fabs(c0[k]) <= 0x1p29f && std::fabs(c1[k]) <= 0x1p29f;  // false for NaN
        bounded = bounded && std::fabs(sp->radius[i]) <= 0x1p30f;
        radii_normal = radii_normal && std::fabs(sp->radius[i]) >= 0x1p-30f && std::fabs(sp->radius[i]) <= 0x1p30f;
        out.sph[i] = Float4{c0[0], c0[1], c0[2], sp->radius[i]};
        out.motion[i] = Float4{c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2], sp->time0[i]};
        out.motion_dt[i] = sp->time1[i] - sp->time0[i];
        out.sph_info[i] = m | (mt->kind[m] << kInfoKindShift) | (still ? 0u : kInfoMovingBit);
        if (mt->kind[m] == RRT_MAT_DIELECTRIC) {
            // a glass sphere is never on a path's trail (its attenuation is 1), so the colour lanes of its row carry what
            // Dielectric::scatter derives from ref_idx alone (materials.rs:80, 66-67), in the same f32 operations
            const float ior = out.mat[m].w, inv = 1.0f / ior;
            const float qf = (1.0f - inv) / (1.0f + inv), qb = (1.0f - ior) / (1.0f + ior);
            out.prim_att[i] = Float4{inv, qf * qf, qb * qb, ior};
        }
        else out.prim_att[i] = Float4{mt->albedo_r[m], mt->albedo_g[m], mt->albedo_b[m], out.mat[m].w};
        if (!still) {
            if (!any_moving) { ut0 = f2u(sp->time0[i]); ut1 = f2u(sp->time1[i]); out.u_time0 = sp->time0[i]; out.u_dt = out.motion_dt[i]; }
            else if (ut0 != f2u(sp->time0[i]) || ut1 != f2u(sp->time1[i])) uniform = false;
            any_moving = true;
        }
    }
    if (any_moving) out.flags |= kSceneHasMoving;
    if (any_moving && uniform) out.flags |= kSceneUniformTimes;
    if (bounded && (!any_moving || uniform)) out.flags |= kSceneBounded;
    if (radii_normal) out.flags |= kSceneRadiiNormal;

    // ---- tree validation: every node reachable from the root exactly once, indices in range ----
    std::vector<uint8_t> seen(n_nodes, 0);
    struct Item { uint32_t node; uint32_t depth; };
    std::vector<Item> stack;
    stack.push_back({0u, 0u});
    uint32_t visited = 0;
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        if (it.node >= n_nodes) { err = "BVH child index out of range"; return RRT_ERR_BAD_SCENE; }
        if (seen[it.node]) { err = "BVH is not a tree (node reached twice)"; return RRT_ERR_BAD_SCENE; }
        seen[it.node] = 1; visited++;
        const rrt_bvh_node &n = nodes[it.node];
        if (is_leaf(n)) {
            if (n.left_or_obj >= ns) { err = "BVH leaf object index out of range"; return RRT_ERR_BAD_SCENE; }
            if (it.depth > out.max_leaf_depth) out.max_leaf_depth = it.depth;
        } else {
            const uint32_t axis = (n.right_axis_leaf >> RRT_NODE_AXIS_SHIFT) & 3u;
            if (axis > 2) { err = "BVH inner node axis out of range"; return RRT_ERR_BAD_SCENE; }
            stack.push_back({n.right_axis_leaf & RRT_NODE_INDEX_MASK, it.depth + 1});
            stack.push_back({n.left_or_obj, it.depth + 1});
        }
    }
    (void)visited;  // unreachable nodes are never visited by Bvh::hit either

    auto ref_of = [&](uint32_t idx) { return idx | (is_leaf(nodes[idx]) ? kRefLeafBit : 0u); };

    // ---- node words ----
    out.nodes.assign(static_cast<size_t>(n_nodes) * kNodeWords, 0u);
    out.node_kind.assign(n_nodes, 0);
    parallel_ranges(n_nodes, [&](size_t begin, size_t end) {
    for (size_t i = begin; i < end; i++) {
        if (!seen[i]) continue;
        uint32_t *w = &out.nodes[i * kNodeWords];
        const rrt_bvh_node &n = nodes[i];
        out.node_kind[i] = is_leaf(n) ? 2 : 1;
        if (is_leaf(n)) {
            const uint32_t p = n.left_or_obj;
            const float r = out.sph[p].w;
            w[0] = f2u(out.sph[p].x); w[1] = f2u(out.sph[p].y); w[2] = f2u(out.sph[p].z);
            w[3] = f2u(r * r);  // self.radius * self.radius (sphere.rs:73)
            w[4] = p | (((out.sph_info[p] >> kInfoKindShift) & 3u) << kLeafKindShift);  // the hit record carries the material kind along
            w[5] = (out.sph_info[p] & kInfoMovingBit) ? 1u : 0u;
        } else {
            for (int k = 0; k < 3; k++) { w[2 * k] = f2u(n.min[k]); w[2 * k + 1] = f2u(n.max[k]); }
            const uint32_t axis = (n.right_axis_leaf >> RRT_NODE_AXIS_SHIFT) & 3u;
            w[6] = ref_of(n.right_axis_leaf & RRT_NODE_INDEX_MASK) | (axis << RRT_NODE_AXIS_SHIFT);
            w[7] = ref_of(n.left_or_obj);
        }
    }
    });
    out.root_ref = ref_of(0);

    // ---- exit links: for octant o, the order of acceleration.rs:253-267 is fixed, so the node popped after a
    //      subtree is known ahead of time.  near = right when the ray is negative along the node's axis. ----
    //      One walk carries the links of all eight octants: a child that is visited first ("near") in octant o exits to
    //      its sibling, the other one inherits its parent's exit.
    struct Link { uint32_t node; uint32_t exit_ref[8]; };
    // (the subtrees left on the stack once it holds a few hundred entries are disjoint: on big trees they are shared out
    // to host threads)
    auto walk = [&](std::vector<Link> &ls, size_t stop_at) {
    while (!ls.empty() && ls.size() < stop_at) {
        const Link l = ls.back(); ls.pop_back();
        uint32_t *w = &out.nodes[static_cast<size_t>(l.node) * kNodeWords + 8];
        for (uint32_t oct = 0; oct < 8; oct++) w[oct] = l.exit_ref[oct];
        const rrt_bvh_node &n = nodes[l.node];
        if (is_leaf(n)) continue;
        const uint32_t axis = (n.right_axis_leaf >> RRT_NODE_AXIS_SHIFT) & 3u;
        const uint32_t left = n.left_or_obj, right = n.right_axis_leaf & RRT_NODE_INDEX_MASK;
        Link ll{left, {}}, lr{right, {}};
        for (uint32_t oct = 0; oct < 8; oct++) {
            const bool neg = (oct >> axis) & 1u;  // near = right when the ray is negative along the node's axis
            ll.exit_ref[oct] = neg ? l.exit_ref[oct] : ref_of(right);
            lr.exit_ref[oct] = neg ? ref_of(left) : l.exit_ref[oct];
        }
        ls.push_back(lr);
        ls.push_back(ll);
    }
    };
    std::vector<Link> ls;
    ls.push_back(Link{0u, {kRefEnd, kRefEnd, kRefEnd, kRefEnd, kRefEnd, kRefEnd, kRefEnd, kRefEnd}});
    const size_t kShareAt = 512;
    walk(ls, n_nodes >= (1u << 16) ? kShareAt : ~size_t(0));
    if (!ls.empty()) {
        const size_t n_threads = std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<std::vector<Link>> share(n_threads);
        for (size_t k = 0; k < ls.size(); k++) share[k % n_threads].push_back(ls[k]);
        std::vector<std::thread> pool;
        for (size_t t = 0; t < n_threads; t++) pool.emplace_back([&, t] { walk(share[t], ~size_t(0)); });
        for (std::thread &t : pool) t.join();
    }

    // ---- compact copy for LDS residency (layout in scene_layout.h) ----
    const bool moving = (out.flags & kSceneHasMoving) != 0;
    const bool times_ok = !moving || (out.flags & kSceneUniformTimes) != 0;
    uint32_t n_inner = 0, n_leaf = 0;
    for (uint32_t i = 0; i < n_nodes; i++) { if (out.node_kind[i] == 1) n_inner++; else if (out.node_kind[i] == 2) n_leaf++; }
    const uint32_t leaf_bytes = moving ? kSmallLeafBytesMoving : kSmallLeafBytes;
    const uint64_t end_off = static_cast<uint64_t>(n_inner) * kSmallInnerBytes;
    const uint64_t total = end_off + 4u + static_cast<uint64_t>(n_leaf) * leaf_bytes;
    if (times_ok && total <= kSmallLdsBudget) {
        // byte offset of every reachable node: inner nodes and leaves numbered separately, each kind in node order
        std::vector<uint32_t> off(n_nodes, 0u);
        uint32_t ii = 0, li = 0;
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (out.node_kind[i] == 1) off[i] = (ii++) * kSmallInnerBytes;
            else if (out.node_kind[i] == 2) off[i] = static_cast<uint32_t>(end_off) + 4u + (li++) * leaf_bytes;
        }
        auto ref16 = [&](uint32_t r) -> uint32_t { return r == kRefEnd ? static_cast<uint32_t>(end_off) : off[r & kRefIndexMask]; };
        // a link that octant `o` follows to an inner node points at the (near, far) pair of that node's x planes for the
        // octant: 8 bytes in for rays that are negative along x (scene_layout.h)
        auto link16 = [&](uint32_t r, uint32_t o) -> uint32_t {
            const uint32_t a = ref16(r);
            return a < end_off ? a + ((o & 1u) ? 8u : 0u) : a;
        };
        out.small_bytes = static_cast<uint32_t>(total);
        out.small_end = static_cast<uint32_t>(end_off);
        out.small_leaf_links = leaf_bytes - 16u;
        out.small_nodes.assign((static_cast<size_t>(total) + 15u) / 16u * 4u, 0u);  // the extend kernel copies it in 16-byte pieces
        for (uint32_t i = 0; i < n_nodes; i++) {
            if (out.node_kind[i] == 0) continue;
            const uint32_t *w = &out.nodes[static_cast<size_t>(i) * kNodeWords];
            uint32_t *s = &out.small_nodes[off[i] / 4u];
            if (out.node_kind[i] == 1) {
                for (int k = 0; k < 3; k++) { s[4 * k] = w[2 * k]; s[4 * k + 1] = w[2 * k + 1]; s[4 * k + 2] = w[2 * k + 1]; s[4 * k + 3] = w[2 * k]; }
                const uint32_t axis = (w[6] >> RRT_NODE_AXIS_SHIFT) & 3u;
                const uint32_t left = ref16(w[7]), right = ref16(w[6] & (kRefLeafBit | kRefIndexMask));
                for (uint32_t o = 0; o < 8; o++) {
                    const uint32_t near_ref = ((o >> axis) & 1u) ? right : left;  // acceleration.rs:253-267
                    s[kSmallLinkOffset / 4u + o] = (near_ref < end_off ? near_ref + ((o & 1u) ? 8u : 0u) : near_ref) | (link16(w[8 + o], o) << 16);
                }
            } else {
                for (int k = 0; k < 5; k++) s[k] = w[k];
                uint32_t *links = s + 5;
                if (moving) {
                    s[5] = w[5];
                    if (w[5] & 1u) {
                        const Float4 &m = out.motion[w[4] & kLeafPrimMask];
                        s[6] = f2u(m.x); s[7] = f2u(m.y); s[8] = f2u(m.z);
                    }
                    links = s + 9;
                }
                for (uint32_t o = 0; o < 8; o += 2) links[o / 2] = link16(w[8 + o], o) | (link16(w[8 + o + 1], o + 1u) << 16);
            }
        }
        out.small_root = ref16(out.root_ref);
    }
    return RRT_OK;
}

void make_top_tree(SceneLayout &L) {
    L.top_nodes.clear();
    if (!L.small_nodes.empty() || L.n_nodes == 0 || L.root_ref == kRefEnd) return;
    const uint32_t none = 0xFFFFFFFFu;
    std::vector<uint32_t> local(L.n_nodes, none);
    std::vector<uint32_t> order, level{L.root_ref & kRefIndexMask}, next;
    // whole levels, breadth first, while they fit
    while (!level.empty() && order.size() + level.size() <= kTopMaxNodes) {
        next.clear();
        for (uint32_t i : level) {
            local[i] = static_cast<uint32_t>(order.size());
            order.push_back(i);
            if (L.node_kind[i] != 1) continue;
            const uint32_t *w = &L.nodes[static_cast<size_t>(i) * kNodeWords];
            next.push_back(w[7] & kRefIndexMask);
            next.push_back(w[6] & kRefIndexMask);
        }
        level.swap(next);
    }
    if (order.empty()) return;
    auto remap = [&](uint32_t ref) -> uint32_t {
        if (ref == kRefEnd) return ref;
        const uint32_t at = local[ref & kRefIndexMask];
        return at == none ? ref : ((ref & ~kRefIndexMask) | kRefTopBit | at);
    };
    parallel_ranges(L.n_nodes, [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; i++) {
            if (L.node_kind[i] == 0) continue;
            uint32_t *w = &L.nodes[i * kNodeWords];
            if (L.node_kind[i] == 1) { w[6] = remap(w[6]); w[7] = remap(w[7]); }  // remap keeps the axis bits of word 6
            for (int o = 0; o < 8; o++) w[8 + o] = remap(w[8 + o]);
        }
    });
    L.root_ref = remap(L.root_ref);
    L.top_nodes.resize(order.size() * kNodeWords);
    for (size_t k = 0; k < order.size(); k++)
        for (uint32_t j = 0; j < kNodeWords; j++) L.top_nodes[k * kNodeWords + j] = L.nodes[static_cast<size_t>(order[k]) * kNodeWords + j];
}

void make_wide_nodes(SceneLayout &L) {
    L.wide.clear();
    if (L.top_nodes.empty() || L.root_ref == kRefEnd || L.max_leaf_depth >= kWideStackDepth) return;
    auto node_words = [&](uint32_t ref) -> const uint32_t * {
        const size_t idx = ref & kRefIndexMask;
        return (ref & kRefTopBit) ? &L.top_nodes[idx * kNodeWords] : &L.nodes[idx * kNodeWords];
    };
    // One array: the records of the nodes of the top-of-tree copy first (in its order), then one per node of the node array.
    // Inside the records an inner child is named by its record's index in that array (the walk's own reference to an inner
    // node: a shift away from the record's address, and `index < n` says whether the record is among the first n, which the
    // walk keeps in LDS); a leaf child keeps its node reference.
    const size_t n_top = L.top_nodes.size() / kNodeWords;
    auto wide_ref = [&](uint32_t ref) -> uint32_t {
        if (ref & kRefLeafBit) return ref;
        return (ref & kRefTopBit) ? (ref & kRefIndexMask) : static_cast<uint32_t>(n_top) + (ref & kRefIndexMask);
    };
    auto fill = [&](const uint32_t *w, uint32_t *rec) {  // w: an inner node's words, references already final
        const uint32_t left = w[7], right = w[6] & kRefMask;
        const uint32_t *lw = node_words(left), *rw = node_words(right);
        for (int k = 0; k < 6; k++) { rec[k] = lw[k]; rec[6 + k] = rw[k]; }
        rec[12] = wide_ref(left); rec[13] = wide_ref(right) | (w[6] & ~kRefMask);  // word 13 keeps the split axis
        rec[14] = rec[15] = 0u;
    };
    L.wide.assign((n_top + static_cast<size_t>(L.n_nodes)) * kNodeWords, 0u);
    parallel_ranges(L.n_nodes, [&](size_t begin, size_t end) {
        for (size_t i = begin; i < end; i++)
            if (L.node_kind[i] == 1) fill(&L.nodes[i * kNodeWords], &L.wide[(n_top + i) * kNodeWords]);
    });
    for (size_t k = 0; k < n_top; k++) {
        const uint32_t *w = &L.top_nodes[k * kNodeWords];
        // an inner node's child references are never zero (node 0 is the root, nobody's child); a leaf's words 6 and 7 are
        if (w[6] != 0u || w[7] != 0u) fill(w, &L.wide[k * kNodeWords]);
    }
    L.wide_root = wide_ref(L.root_ref);
    const uint32_t *rw = node_words(L.root_ref);
    for (int k = 0; k < 6; k++) { uint32_t u = rw[k]; std::memcpy(&L.root_box[k], &u, 4); }
    L.flags |= kSceneWideRecords;
}

}  // namespace rrt_amd
