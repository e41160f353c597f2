// Wavefront path tracing kernels for gfx950 (wave64).
//
//   raygen     fills every path slot of a batch with a camera ray            (lib.rs:93-101, camera.rs:56-66)
//   extend     closest hit of each live ray: BVH walk + sphere quadratic     (acceleration.rs:221-272, sphere.rs:68-101)
//   shade      hit record, material scatter, sky, path termination,          (lib.rs:50-71, materials.rs, sphere.rs:86-93)
//              in-place path regeneration and ballot/popcount compaction of the wave's live-slot list
//   accumulate per-pixel sums over the batch's samples in sample order       (lib.rs:95-103)
//   finalize   gamma 2 + 8-bit quantisation                                  (lib.rs:105-108)
//
// Arithmetic follows the reference operation for operation: compiled with -ffp-contract=off, fused
// multiply-adds only where ultraviolet's dot/mag use mul_add, IEEE division and square root.
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "kernels.h"
#include "scene_layout.h"

namespace rrt_amd {
namespace {

#define DEV __device__ __forceinline__

DEV float u2f(uint32_t u) { return __uint_as_float(u); }
DEV uint32_t f2u(float f) { return __float_as_uint(f); }
DEV uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// how many lanes below this one are set in a ballot (v_mbcnt_lo/hi: no per-lane mask to keep)
DEV uint32_t prefix_count(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

typedef float f32x2 __attribute__((ext_vector_type(2)));  // one v_pk_* operand

struct V3 { float x, y, z; };
DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 scale(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// ultraviolet Vec4::dot with both w lanes zero: x.mul_add(ox, y.mul_add(oy, z.mul_add(oz, 0)))
DEV float dot3(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, __builtin_fmaf(a.z, b.z, 0.0f))); }
// Correctly rounded square root in 5 instructions instead of the 20 of sqrtf: the hardware's reciprocal-root estimate y,
// the root estimate s = x y, and one Newton step on s with the exact fma residual x - s s and the slope y / 2.  Checked
// against sqrtf on every non-negative input (tools/micro/sqrt_exact.hip): the result is the same bit for bit for every x
// from 2^-102 up to the largest finite value; below that (zero and denormals included) and for +inf the result is NaN or
// a number below 2^-51.  Both show in the result: if any lane of the wave has one, the whole wave takes sqrtf.
DEV bool sqrt_suspect(float s) { return !(s >= 0x1p-48f); }
DEV float sqrt_rn(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y, s0);
    if (__builtin_expect(__ballot(sqrt_suspect(s)) != 0ull, 0)) s = sqrtf(x);
    return s;
}

// Correctly rounded 1 / x in 3 instructions instead of the 11 of an IEEE division: the hardware estimate and one
// Newton step.  Checked against 1.0f / x on every one of the 2^32 inputs (tools/micro/recip_exact.hip): the result
// is the same bit for bit unless it comes out NaN (x zero, denormal or infinite) or x is 2^126 or more, where the quotient
// is denormal.  Both cases show in the result (NaN, or magnitude below 2^-125): if any lane of the wave has one, the whole
// wave divides.
DEV bool recip_suspect(float q) { return !(fabsf(q) >= 0x1p-125f); }
DEV float recip_newton(float x) { const float r0 = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0); }
DEV float recip_rn(float x) {
    float q = recip_newton(x);
    if (__builtin_expect(__ballot(recip_suspect(q)) != 0ull, 0)) q = 1.0f / x;
    return q;
}

// n / a in three instructions where y is the correctly rounded 1 / a: q0 = n y is within an ulp of the quotient, the
// residual n - a q0 is exact in one fma, and q0 + residual y rounds to the correctly rounded quotient (Markstein's
// correction; checked against n / a on 10^11 random and near-tie cases of either sign, tools/micro/div_exact.hip).  Holds
// while no intermediate leaves the normal range; callers guard |a| in [2^-30, 2^30] and n either +0 with a > 0 or |n| in
// [2^-90, 2^90].
DEV float div_known(float n, float a, float y) { const float q0 = n * y; return __builtin_fmaf(__builtin_fmaf(-a, q0, n), y, q0); }

// ultraviolet normalized() is the reciprocal of the magnitude, recip_rn(sqrt_rn(dot3(a, a))), then one multiply per
// component: a * (1 / |a|).  This is the multiply half.
DEV V3 scale_r(V3 a, float r) { return mk(a.x * r, a.y * r, a.z * r); }
// ultraviolet reflected(): self - (2 * self.dot(n)) * n
DEV V3 reflected(V3 i, V3 n) { const float k = 2.0f * dot3(i, n); return sub(i, scale(k, n)); }

// 1 + 2*gamma_eb(3) of fp.rs:16-19 evaluated in f32 (acceleration.rs:78)
#define RRT_FAR_SCALE 0x1.000006p+0f
// scale of rand's Uniform::new_inclusive(-1., 1.) (rng.rs:33,43): (1 - -1) / (1 - 2^-23) rounded, no decrement needed
#define RRT_PM1_SCALE 0x1.000002p+1f

// ---- Xoshiro128Plus (rng.rs:14) ----
struct Rng { uint32_t s0, s1, s2, s3; };

DEV uint32_t next_u32(Rng &g) {
    const uint32_t r = g.s0 + g.s3;
    const uint32_t t = g.s1 << 9;
    g.s2 ^= g.s0; g.s3 ^= g.s1; g.s1 ^= g.s2; g.s0 ^= g.s3;
    g.s2 ^= t;
    g.s3 = (g.s3 << 11) | (g.s3 >> 21);
    return r;
}
DEV float gen_f32(Rng &g) { return (float)(next_u32(g) >> 8) * 0x1p-24f; }  // rand Standard
// rand's 23-bit uniform in [0, 1): (next_u32 >> 9) * 2^-23.  The 23 bits under the exponent of 1.0 are the float 1 + v, and
// subtracting 1.0 from it is exact: the same value in two instructions (v_alignbit_b32, v_sub_f32) instead of a shift, a
// conversion and a multiply.
DEV float unit23(Rng &g) { return u2f(__builtin_amdgcn_alignbit(0x7Fu, next_u32(g), 9u)) - 1.0f; }
DEV float sample_pm1(Rng &g) { return unit23(g) * RRT_PM1_SCALE + -1.0f; }

DEV uint64_t splitmix64(uint64_t &st) {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
DEV void seed_from_u64(Rng &g, uint64_t seed) {
    for (;;) {
        uint64_t st = seed;
        const uint64_t z1 = splitmix64(st), z2 = splitmix64(st);
        g.s0 = (uint32_t)z1; g.s1 = (uint32_t)(z1 >> 32); g.s2 = (uint32_t)z2; g.s3 = (uint32_t)(z2 >> 32);
        if ((g.s0 | g.s1 | g.s2 | g.s3) != 0u) return;
        seed = 0;  // an all-zero state is replaced by seed_from_u64(0)
    }
}

// Per-ray constants of the walk, computed once where the ray is made: 1/d (types.rs:21-23) and d.d (sphere.rs:71)
DEV float4 ray_constants(V3 d) {
    float ix = recip_newton(d.x), iy = recip_newton(d.y), iz = recip_newton(d.z);
    if (__builtin_expect(__ballot(recip_suspect(ix) || recip_suspect(iy) || recip_suspect(iz)) != 0ull, 0)) { ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z; }
    return make_float4(ix, iy, iz, dot3(d, d));
}

// ---- path generation (lib.rs:93-101, camera.rs:56-66) ----
DEV uint32_t fd_div(uint32_t n, const FastDiv &f) { const uint32_t t = __umulhi(f.magic, n); return (t + ((n - t) >> f.sh1)) >> f.sh2; }

// Leaves 1/direction and direction.direction (ray_constants) to the caller.
DEV void generate_path(const BatchDev &Bt, const CameraDev &C, uint32_t p, float4 &A, float4 &B, Rng &g) {
    const uint32_t sl = fd_div(p, Bt.div_npix);
    const uint32_t pl = p - sl * Bt.npix;
    const uint32_t idx = Bt.pixel_map ? Bt.pixel_map[pl] : pl;
    const uint32_t k = Bt.k_first + sl * Bt.k_step;
    seed_from_u64(g, Bt.seed ^ ((uint64_t)idx * 0x9E3779B97F4A7C15ull + (uint64_t)k * 0xD1B54A32D192ED03ull));
    const uint32_t row = fd_div(idx, Bt.div_nx);
    const float x = (float)(idx - row * Bt.nx);
    const float y = (float)(Bt.ny - 1u - row);
    // (x + r) / nx and (y + r) / ny (lib.rs:98-99) through the correctly rounded reciprocals the host supplies: the
    // numerators are +0 or in [2^-24, 2^28) and the divisors in [1, 2^28], so div_known is the IEEE quotient bit for bit
    const float s = div_known(x + gen_f32(g), (float)Bt.nx, Bt.inv_nx);
    const float t = div_known(y + gen_f32(g), (float)Bt.ny, Bt.inv_ny);
    float dx, dy;
    for (;;) {  // rng.rs:42-50
        dx = sample_pm1(g); dy = sample_pm1(g);
        if (__builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, 0.0f)) < 1.0f) break;
    }
    const float rdx = C.lens_radius * dx, rdy = C.lens_radius * dy;
    const V3 u = mk(C.u[0], C.u[1], C.u[2]), v = mk(C.v[0], C.v[1], C.v[2]);
    const V3 offset = add(scale(rdx, u), scale(rdy, v));
    const V3 origin = add(mk(C.org[0], C.org[1], C.org[2]), offset);
    const V3 llc = mk(C.llc[0], C.llc[1], C.llc[2]), hor = mk(C.hor[0], C.hor[1], C.hor[2]), ver = mk(C.ver[0], C.ver[1], C.ver[2]);
    const V3 dir = sub(add(add(llc, scale(s, hor)), scale(t, ver)), origin);
    const float tv = unit23(g);
    const float time = tv * C.time_scale + C.time_low;  // rand Uniform::sample (camera.rs:64)
    A = make_float4(origin.x, origin.y, origin.z, time);
    B = make_float4(dir.x, dir.y, dir.z, u2f(p));
}

// ---- the two layouts of the origin plane and of the depth / trail-window state (kernels.h: kLayout*) ----
struct F3 { float x, y, z; };
DEV float4 origin_load(const PathState &P, uint32_t slot, bool timed) {
    if (timed) return P.rayA[slot];
    const F3 v = reinterpret_cast<const F3 *>(P.rayA)[slot];
    return make_float4(v.x, v.y, v.z, 0.0f);
}
DEV void origin_store(const PathState &P, uint32_t slot, bool timed, float4 A) {
    if (timed) P.rayA[slot] = A;
    else { F3 v; v.x = A.x; v.y = A.y; v.z = A.z; reinterpret_cast<F3 *>(P.rayA)[slot] = v; }
}
// depth word: scatters so far (low 16 bits) | trail entries written (high 16 bits)
DEV uint32_t narrow_depth_word(uint32_t x) { return (x & 0xFFu) | ((x & 0xFF00u) << 8); }
DEV uint32_t depth_load(const PathState &P, uint32_t slot, bool narrow) {
    return narrow ? narrow_depth_word(reinterpret_cast<const uint2 *>(P.recent)[slot].x) : P.depth[slot];
}
DEV void meta_load(const PathState &P, uint32_t slot, bool narrow, uint32_t &depth_word, uint4 &recent) {
    if (narrow) {
        const uint2 m = reinterpret_cast<const uint2 *>(P.recent)[slot];
        depth_word = narrow_depth_word(m.x);
        recent = make_uint4((m.x >> 16) & 0x3FFu, m.y & 0x3FFu, (m.y >> 10) & 0x3FFu, (m.y >> 20) & 0x3FFu);
    } else {
        depth_word = P.depth[slot];
        recent = P.recent[slot];
    }
}
DEV void meta_store(const PathState &P, uint32_t slot, bool narrow, uint32_t depth_word, uint4 recent) {
    if (narrow) {
        reinterpret_cast<uint2 *>(P.recent)[slot] = make_uint2((depth_word & 0xFFu) | ((depth_word >> 8) & 0xFF00u) | (recent.x << 16),
                                                               recent.y | (recent.z << 10) | (recent.w << 20));
    } else {
        P.depth[slot] = depth_word;
        P.recent[slot] = recent;
    }
}

// ---- BVH traversal (acceleration.rs:221-272) over the linked layouts of scene_layout.h ----
//
// A node store hands out the words of a node; two exist: the 64-byte nodes in HBM (any scene) and the compact copy in
// LDS (small scenes).  References order as inner < kEnd < leaf, so `cur < kEnd` means "inner node".
struct RayCtx {
    f32x2 ox, oy, oz;  // per axis {origin, 1 / direction}: both ray operands of the slab arithmetic in one aligned register pair
    V3 d;
    float a;      // d.d (sphere.rs:71), the same for every sphere
    float ra;     // 1 / a, correctly rounded (where quick_div_ok says so)
    float time, f_ray;
    float tmin, tmax;
    uint32_t oct;   // bit k set: 1/d[k] < 0 (acceleration.rs:230-231)
    // for the compact LDS image: byte offset inside an inner node of the (near, far) pair of x planes (what the walk's reference
    // to an inner node points at), from there to the pairs of y and z planes and to the (near child | exit) word of the ray's
    // octant, and inside a leaf to the octant's exit link
    uint32_t cx, cy, cz, link, leaf_link;
    uint32_t best;  // store-specific handle of the closest leaf so far
};

DEV V3 origin(const RayCtx &r) { return mk(r.ox.x, r.oy.x, r.oz.x); }

DEV void ray_finish(RayCtx &r, float tmin, float tmax, const SceneDev &S, bool uniform_times) {
    const uint32_t nx = r.ox.y < 0.0f ? 1u : 0u, ny = r.oy.y < 0.0f ? 1u : 0u, nz = r.oz.y < 0.0f ? 1u : 0u;  // acceleration.rs:230-231
    r.oct = nx | (ny << 1) | (nz << 2);
    // the walk's reference to an inner node already points at the ray's pair of x planes (scene_layout.h): the other offsets
    // count from there
    r.cx = nx << 3; r.cy = 16u + (ny << 3) - r.cx; r.cz = 32u + (nz << 3) - r.cx;
    r.link = kSmallLinkOffset + (r.oct << 2) - r.cx;
    r.leaf_link = S.small_leaf_links + (r.oct << 1);
    r.f_ray = uniform_times ? (r.time - S.u_time0) / S.u_dt : 0.0f;  // center_at's factor when every moving sphere shares its shutter interval
    r.tmin = tmin;
    r.tmax = tmax;
    r.best = kPrimNone;
}

DEV void ray_load(RayCtx &r, float4 A, float4 B, float4 C, float tmin, float tmax, const SceneDev &S, bool uniform_times) {
    r.ox = f32x2{A.x, C.x}; r.oy = f32x2{A.y, C.y}; r.oz = f32x2{A.z, C.z};
    r.d = mk(B.x, B.y, B.z);
    r.time = A.w;
    r.a = C.w;
    r.ra = recip_newton(r.a);
    ray_finish(r, tmin, tmax, S, uniform_times);
}

// n / a for the ray's a = d.d in three instructions instead of the division's eleven: with y the correctly rounded
// 1 / a, q0 = n y is within an ulp of the quotient, the residual n - a q0 is exact in one fma, and q0 + residual y
// rounds to the correctly rounded quotient (Markstein's correction; checked against n / a on 1.4e11 random and near-tie
// cases, tools/micro/div_exact.hip).  That holds while no intermediate leaves the normal range.  Instead of looking at
// every numerator, the caller looks at every ray once (quick_div_ok) in a scene whose spheres are bounded
// (kSceneBounded) walked with t_min >= 2^-40:
//   * a in [2^-30, 2^30], so y is the exact reciprocal rounded (recip_newton) and no direction component exceeds 2^15;
//   * origin components up to 2^30 and centres up to 2^31 (a moving sphere's, with the ray's time inside the shutter
//     interval all moving spheres share: 2^29 + 1 * 2^30) put |o - c| below 2^32 per component, |half_b| below 2^49,
//     |c| below 2^66 and the numerators -half_b -+ sqrt(half_b^2 - a c) below 2^51: nothing overflows;
//   * a numerator below 2^-80, where the residual could go subnormal, gives a quotient below 2^-50 by either
//     route: under t_min, a rejected root whatever its last bits.
template <bool MOVING>
DEV bool quick_div_ok(const RayCtx &r) {
    return r.a >= 0x1p-30f && r.a <= 0x1p30f && fmaxf(fmaxf(fabsf(r.ox.x), fabsf(r.oy.x)), fabsf(r.oz.x)) <= 0x1p30f &&
           (!MOVING || (r.f_ray >= 0.0f && r.f_ray <= 1.0f));
}
DEV float div_by_a(float n, const RayCtx &r, uint32_t quick) {  // quick: 0 or 1, the same in every lane
    if (quick == 0u) return n / r.a;
    const float q0 = n * r.ra;
    return __builtin_fmaf(__builtin_fmaf(-r.a, q0, n), r.ra, q0);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 *HbmNodePtr;
typedef const __attribute__((address_space(3))) u32x4 *LdsNodePtr;  // spelled out so that the loads are ds_read, not flat_load
template <class Ptr>
DEV uint4 load4(Ptr p, uint32_t i) { const u32x4 v = p[i]; return make_uint4(v.x, v.y, v.z, v.w); }

constexpr uint32_t kLdsBlockThreads = 1024;  // 16 waves share one LDS copy of the scene (or of the top of its tree)

struct NoStack {};  // the walks along exit links keep nothing but the current node
DEV void bind_spill(NoStack &, uint2 *) {}

template <bool MOVING, class TopPtr = HbmNodePtr>
struct HbmStore {
    typedef NoStack Stack;
    static constexpr bool kCountsItself = false;
    static constexpr uint32_t kSpillEntries = 1;
    const SceneDev &S;
    TopPtr top;  // the copy of the top of the tree that references with kRefTopBit index: in HBM, or (k_extend_top) in LDS
    DEV explicit HbmStore(const SceneDev &s) : S(s), top(reinterpret_cast<HbmNodePtr>(s.top_nodes)) {}
    DEV HbmStore(const SceneDev &s, TopPtr top_copy) : S(s), top(top_copy) {}
    DEV uint32_t root() const { return S.root_ref; }
    DEV uint32_t end() const { return kRefEnd; }  // references order as inner < end < leaf
    // the first 32 bytes of the node and its exit link for octant `oct`
    DEV void fetch(uint32_t ref, uint32_t oct, uint4 &n0, uint4 &n1, uint32_t &exit_ref) const {
        const uint32_t idx = ref & kRefIndexMask;
        if (ref & kRefTopBit) {
            n0 = load4(top, idx * 4u); n1 = load4(top, idx * 4u + 1u);
            const uint4 links = load4(top, idx * 4u + 2u + (oct >> 2));
            const uint32_t k = oct & 3u;
            exit_ref = k == 0u ? links.x : (k == 1u ? links.y : (k == 2u ? links.z : links.w));
        } else {
            n0 = S.nodes[idx * 4u]; n1 = S.nodes[idx * 4u + 1u];
            exit_ref = reinterpret_cast<const uint32_t *>(S.nodes)[idx * kNodeWords + 8u + oct];
        }
    }
    DEV bool uniform_times() const { return MOVING && (S.flags & kSceneUniformTimes); }
    DEV uint32_t start(RayCtx &, NoStack &, uint32_t &) const { return root(); }
    // bx, by, bz: per axis the (near, far) planes of the node's box for this ray (acceleration.rs:66-71)
    DEV void inner(uint32_t cur, const RayCtx &r, f32x2 &bx, f32x2 &by, f32x2 &bz, uint32_t &near_ref, uint32_t &exit_ref) const {
        const uint32_t oct = r.oct;
        uint4 n0, n1;
        fetch(cur, oct, n0, n1, exit_ref);
        const bool ngx = oct & 1u, ngy = oct & 2u, ngz = oct & 4u;
        bx = ngx ? f32x2{u2f(n0.y), u2f(n0.x)} : f32x2{u2f(n0.x), u2f(n0.y)};
        by = ngy ? f32x2{u2f(n0.w), u2f(n0.z)} : f32x2{u2f(n0.z), u2f(n0.w)};
        bz = ngz ? f32x2{u2f(n1.y), u2f(n1.x)} : f32x2{u2f(n1.x), u2f(n1.y)};
        const uint32_t axis = (n1.z >> RRT_NODE_AXIS_SHIFT) & 3u;
        near_ref = ((oct >> axis) & 1u) ? (n1.z & kRefMask) : n1.w;  // acceleration.rs:253-267
    }
    DEV void leaf(uint32_t cur, const RayCtx &r, V3 &c, float &rr, uint32_t &handle, uint32_t &exit_ref) const {
        uint4 n0, n1;
        fetch(cur, r.oct, n0, n1, exit_ref);
        c = mk(u2f(n0.x), u2f(n0.y), u2f(n0.z));
        rr = u2f(n0.w);
        handle = n1.x;  // sphere index | material kind << 30
        if (MOVING && (n1.y & 1u)) {  // center_at (sphere.rs:61-64)
            const uint32_t prim = n1.x & kLeafPrimMask;
            const float4 m = S.motion[prim];
            const float f = (S.flags & kSceneUniformTimes) ? r.f_ray : (r.time - m.w) / S.motion_dt[prim];
            c = add(c, scale(f, mk(m.x, m.y, m.z)));
        }
    }
    DEV void prepare(RayCtx &) const {}
    DEV uint32_t prim_of(uint32_t handle) const { return handle; }
};

// The compact image in LDS (scene_layout.h): a reference is a node's LDS address, inner nodes lie below `end`, leaves above.
template <bool MOVING>
struct LdsStore {
    typedef NoStack Stack;
    static constexpr bool kCountsItself = false;
    static constexpr uint32_t kSpillEntries = 1;
    uint32_t root16, end16;
    DEV LdsStore(uint32_t root, uint32_t end) : root16(root), end16(end) {}
    DEV uint32_t root() const { return root16; }
    DEV uint32_t end() const { return end16; }
    DEV bool uniform_times() const { return MOVING; }  // the compact copy exists only for scenes with one shutter interval
    DEV uint32_t start(RayCtx &r, NoStack &, uint32_t &) const { return root16 < end16 ? root16 + r.cx : root16; }  // an inner root: at the ray's pair of x planes
    typedef const __attribute__((address_space(3))) uint32_t *LdsWordPtr;
    typedef const __attribute__((address_space(3))) float *LdsFloatPtr;
    typedef const __attribute__((address_space(3))) uint16_t *LdsHalfPtr;
    DEV static LdsWordPtr words(uint32_t lds_addr) { return reinterpret_cast<LdsWordPtr>(static_cast<uintptr_t>(lds_addr)); }
    DEV static LdsFloatPtr floats(uint32_t lds_addr) { return reinterpret_cast<LdsFloatPtr>(static_cast<uintptr_t>(lds_addr)); }
    typedef const __attribute__((address_space(3))) f32x2 *LdsPairPtr;
    DEV static f32x2 pair_at(uint32_t lds_addr) { return *reinterpret_cast<LdsPairPtr>(static_cast<uintptr_t>(lds_addr)); }  // 8-byte aligned: ds_read_b64
    DEV void prepare(RayCtx &) const {}
    // The box is stored per axis as (min, max, max, min): the second pair is (max, min), so a ray that is negative along
    // the axis reads (near, far) directly, with an address instead of a select.  `links` comes back as near child | exit
    // link << 16 for the ray's octant.
    DEV void inner(uint32_t cur, const RayCtx &r, f32x2 &bx, f32x2 &by, f32x2 &bz, uint32_t &links) const {
        bx = pair_at(cur); by = pair_at(cur + r.cy); bz = pair_at(cur + r.cz);
        links = *words(cur + r.link);
    }
    DEV void leaf(uint32_t cur, const RayCtx &r, V3 &c, float &rr, uint32_t &handle, uint32_t &exit_ref) const {
        const LdsFloatPtr f = floats(cur);
        exit_ref = *reinterpret_cast<LdsHalfPtr>(static_cast<uintptr_t>(cur + r.leaf_link));
        c = mk(f[0], f[1], f[2]);
        rr = f[3];
        handle = cur;
        if (MOVING) {  // centre1 - centre0 behind the flag word, zero words for a sphere at rest
            if (words(cur)[5] & 1u) c = add(c, scale(r.f_ray, mk(f[6], f[7], f[8])));
        }
    }
    DEV uint32_t prim_of(uint32_t handle) const { return words(handle)[4]; }
};

// Aabb::intersect (acceleration.rs:60-84) against the current t_max, then the step of acceleration.rs:250-267.
// The node store hands out, per axis, the (near, far) planes the reference's blend selects by the sign of 1/d; both go
// through one packed subtract and one packed multiply.
// (box - origin) * (1 / direction) for both planes of one axis.  `oi` is {origin, 1 / direction}: the subtract reads its
// low half for both results and the multiply its high half, so the ray needs no duplicated operands in registers.
DEV f32x2 slab_pair(f32x2 box, f32x2 oi) {
    f32x2 diff, t;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(diff) : "v"(box), "v"(oi));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(diff), "v"(oi));
    return t;
}

// f32::min (the operand that is not NaN, if one is) of two values that arithmetic has made: v_min_f32 as it is.  fminf is the same
// instruction behind two more that quiet a signalling NaN in either operand first, which an arithmetic result never is.
DEV float min_quiet(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// The same without the last step: t0 as below, f the scaled exit distance before it is limited by t_max
DEV void slab_entry_exit(f32x2 bx, f32x2 by, f32x2 bz, const RayCtx &r, float &t0, float &f) {
    const f32x2 tx = slab_pair(bx, r.ox), ty = slab_pair(by, r.oy), tz = slab_pair(bz, r.oz);
    float n3, f3;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(n3) : "v"(tx.x), "v"(ty.x), "v"(tz.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(t0) : "v"(n3), "v"(r.tmin));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(f3) : "v"(tx.y), "v"(ty.y), "v"(tz.y));
    f = f3 * RRT_FAR_SCALE;
}

// t0 <= t1 of the slab test, from the three (near, far) plane pairs
DEV void slab_interval(f32x2 bx, f32x2 by, f32x2 bz, const RayCtx &r, float &t0, float &t1) {
    const f32x2 tx = slab_pair(bx, r.ox), ty = slab_pair(by, r.oy), tz = slab_pair(bz, r.oz);
    const float tnx = tx.x, tny = ty.x, tnz = tz.x;
    const float tfx = tx.y, tfy = ty.y, tfz = tz.y;
    // each far value is scaled by 1 + 2*gamma(3) (acceleration.rs:60-84); rounding x -> x * k is monotonic, so scaling
    // the smallest of the three gives bit for bit the smallest of the three scaled values.  The min/max chains are
    // spelled as instructions because the compiler cannot know that the packed results above are already canonical
    // and would re-canonicalise each of them first.
    float n3, f3;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(n3) : "v"(tnx), "v"(tny), "v"(tnz));
    asm("v_max_f32 %0, %1, %2" : "=v"(t0) : "v"(n3), "v"(r.tmin));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(f3) : "v"(tfx), "v"(tfy), "v"(tfz));
    const float f3s = f3 * RRT_FAR_SCALE;
    asm("v_min_f32 %0, %1, %2" : "=v"(t1) : "v"(f3s), "v"(r.tmax));
}

template <bool MOVING, class TopPtr>
DEV uint32_t visit_inner(const HbmStore<MOVING, TopPtr> &st, uint32_t cur, RayCtx &r, NoStack &, uint32_t, uint32_t &, uint32_t &) {
    f32x2 bx, by, bz;
    uint32_t near_ref, exit_ref;
    st.inner(cur, r, bx, by, bz, near_ref, exit_ref);
    float t0, t1;
    slab_interval(bx, by, bz, r, t0, t1);
    return (t0 <= t1) ? near_ref : exit_ref;  // the far child is the near child's exit link
}

template <bool MOVING>
DEV uint32_t visit_inner(const LdsStore<MOVING> &st, uint32_t cur, RayCtx &r, NoStack &, uint32_t, uint32_t &, uint32_t &) {
    f32x2 bx, by, bz;
    uint32_t links;
    st.inner(cur, r, bx, by, bz, links);
    float t0, t1;
    slab_interval(bx, by, bz, r, t0, t1);
    // low half of `links` (the near child) if the box is hit, high half (the exit link) if not, in one instruction
    uint32_t next;
    asm("v_cmp_le_f32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0"
        : "=v"(next) : "v"(t0), "v"(t1), "v"(links) : "vcc");
    return next;
}

// Sphere::hit (sphere.rs:68-85); the Hit record itself is rebuilt in shade, for the final hit only.  (Deferring the
// square root and divisions until many lanes need them was tried: lanes that wait for roots cost the descent passes as
// many idle lanes as the batched root passes save.)
// quick: 0 or 1, the same in every lane (a number and not a bool: as a bool the compiler keeps it as a lane mask and spends two
// vector instructions per use on turning it over): every ray of the wave passes quick_div_ok
DEV void sphere_test(V3 c, float rr, uint32_t handle, RayCtx &r, uint32_t quick) {
    const V3 oc = sub(origin(r), c);
    const float half_b = dot3(oc, r.d);
    const float cc = dot3(oc, oc) - rr;
    const float disc = (half_b * half_b) - (r.a * cc);
    if (!(disc < 0.0f)) {
        const float sq = sqrt_rn(disc);
        float root = div_by_a(-half_b - sq, r, quick);
        bool outside = root < r.tmin || r.tmax < root;
        if (outside) {
            root = div_by_a(-half_b + sq, r, quick);
            outside = root < r.tmin || r.tmax < root;
        }
        if (!outside) { r.tmax = root; r.best = handle; }
    }
}

template <class Store>
DEV uint32_t visit_leaf(const Store &st, uint32_t cur, RayCtx &r, NoStack &, uint32_t quick, uint32_t &, uint32_t &) {
    V3 c;
    float rr;
    uint32_t handle, exit_ref;
    st.leaf(cur, r, c, rr, handle, exit_ref);
    sphere_test(c, rr, handle, r, quick);
    return exit_ref;
}

// ---- The two-children walk of scenes whose nodes stay in HBM ----
//
// A pass of the walk along exit links costs one dependent memory round trip per node, and a wave waits for the slowest
// of its 64 lanes: with a few hundred thousand nodes in play nearly every pass waits for a miss.  Here a record holds what
// the walk needs of BOTH children of an inner node (their boxes; for a leaf child its sphere), so a round trip settles two
// nodes, and a leaf reached from its parent costs none.  The order of acceleration.rs:236-267 is kept exactly: the near
// child is dealt with at once; the far child waits on a stack (the reference's) unless the near child turns out to be a
// miss or a leaf, in which case the far child is next anyway and is dealt with in the same visit.  The reference tests a
// waiting child's box when it pops it, against the closest hit so far; the entry distance t0 and the scaled exit
// distance do not depend on that, so the box is worked out when the record is at hand, the child is dropped if it misses
// whatever t_max is (entry beyond exit), and otherwise waits with its t0: popping it compares t0 with the t_max of that moment.
// Waiting children are kept by their position on the stack: the first kWideLdsEntries positions in LDS (entry k of the
// workgroup's lane l at (k * 1024 + l) * 8, no bank conflicts), deeper ones in scratch memory.  Scratch goes through the
// same address path as the record gathers, which is what bounds this walk (a first version with the whole stack in
// scratch was no faster than the walk along exit links), and a ray seldom has more than a few children waiting.
constexpr uint32_t kWideStack = kWideStackDepth;
constexpr uint32_t kWideLdsEntries = 6;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 *LdsStackPtr;
struct WideStack {
    LdsStackPtr l;  // this lane's entry 0 in LDS; entry k is kLdsBlockThreads entries further on
    uint2 *e;       // positions from kWideLdsEntries on: an array of the caller's
    uint32_t n;
    DEV void clear() { n = 0u; }
    DEV bool empty() const { return n == 0u; }
    DEV void push(uint2 v) {
        if (n < kWideLdsEntries) l[n * kLdsBlockThreads] = u32x2{v.x, v.y};
        else e[n - kWideLdsEntries] = v;
        n += 1u;
    }
    DEV uint2 pop() {  // not empty
        n -= 1u;
        if (n < kWideLdsEntries) { const u32x2 v = l[n * kLdsBlockThreads]; return make_uint2(v.x, v.y); }
        return e[n - kWideLdsEntries];
    }
};
DEV void bind_spill(WideStack &k, uint2 *e) { k.e = e; }

template <bool MOVING, bool COUNT>
struct WideStore {
    typedef WideStack Stack;
    static constexpr bool kCountsItself = true;
    static constexpr uint32_t kSpillEntries = kWideStack - kWideLdsEntries;
    const SceneDev &S;
    LdsNodePtr top;      // LDS copy of the first `n_top_lds` records of the top of the tree
    uint32_t n_top_lds;
    LdsStackPtr stack0;  // entry 0 of this lane's stack in LDS
    DEV WideStore(const SceneDev &s, LdsNodePtr top_copy, uint32_t n_top_copy, LdsStackPtr lane_stack) : S(s), top(top_copy), n_top_lds(n_top_copy), stack0(lane_stack) {}
    DEV bool in_top_copy(uint32_t ref) const { return ref < n_top_lds; }  // an inner node's reference is its record's index, and the first records are the copy
    DEV uint32_t end() const { return kRefEnd; }
    DEV bool uniform_times() const { return MOVING && (S.flags & kSceneUniformTimes); }
    DEV void prepare(RayCtx &) const {}
    DEV uint32_t prim_of(uint32_t handle) const { return handle; }
    // (near, far) planes of a box given as (min, max) pairs, for this ray (acceleration.rs:66-71)
    // t0: entry distance (not below t_min); f: exit distance, scaled (acceleration.rs:77-81).  The box is hit iff t0 <= min(f, t_max).
    DEV static void box_test(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, const RayCtx &r, float &t0, float &f) {
        const bool ngx = r.oct & 1u, ngy = r.oct & 2u, ngz = r.oct & 4u;
        const f32x2 bx = ngx ? f32x2{u2f(w1), u2f(w0)} : f32x2{u2f(w0), u2f(w1)};
        const f32x2 by = ngy ? f32x2{u2f(w3), u2f(w2)} : f32x2{u2f(w2), u2f(w3)};
        const f32x2 bz = ngz ? f32x2{u2f(w5), u2f(w4)} : f32x2{u2f(w4), u2f(w5)};
        slab_entry_exit(bx, by, bz, r, t0, f);
    }
    // a leaf's six words: centre at time0, radius^2, sphere word, moving flag
    DEV void leaf_test(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5, RayCtx &r, uint32_t quick) const {
        V3 c = mk(u2f(w0), u2f(w1), u2f(w2));
        if (MOVING && (w5 & 1u)) {  // center_at (sphere.rs:61-64)
            const uint32_t prim = w4 & kLeafPrimMask;
            const float4 m = S.motion[prim];
            const float f = (S.flags & kSceneUniformTimes) ? r.f_ray : (r.time - m.w) / S.motion_dt[prim];
            c = add(c, scale(f, mk(m.x, m.y, m.z)));
        }
        sphere_test(c, u2f(w3), w4, r, quick);
    }
    // The next node of the walk once everything at hand is done: the newest waiting child that still passes.  A waiting
    // inner child counts as a box test here, when the reference would pop and test it.
    DEV uint32_t pop(RayCtx &r, WideStack &k, uint32_t &n_inner) const {
        while (!k.empty()) {
            const uint2 e = k.pop();
            if (e.x & kRefLeafBit) return e.x;
            if (COUNT) n_inner++;
            // Aabb::intersect's t0 <= min(exit, t_max) (acceleration.rs:81-83), where f32::min drops a NaN operand.  t0 <= exit is
            // known since the push if the exit is a number; what is left is t0 <= min(+inf, t_max): a NaN t_max (a ray without a
            // direction accepts every sphere it tests at t = NaN, sphere.rs:74-81) leaves the exit, and the box is hit.  If the exit
            // is NaN itself (a NaN ray, which is what follows such a hit) the minimum is t_max, NaN or not: t0 <= min(NaN, t_max).
            // The entry's sign says which: t0 is t_min or more, and this walk runs with t_min > 0 only.
            const float t0 = fabsf(u2f(e.y));  // (the sign says that the exit distance is NaN; t0 itself is t_min or more, and t_min > 0 here)
            const float limit = u2f(0x7F800000u | (uint32_t)((int32_t)e.y >> 31));  // +inf, or with the sign set all ones: a NaN
            if (t0 <= min_quiet(limit, r.tmax)) return e.x;
        }
        return kRefEnd;
    }
    // Bvh::hit's first pop (acceleration.rs:236): the root, whose box no record holds
    DEV uint32_t start(RayCtx &r, WideStack &k, uint32_t &n_inner) const {
        k.clear();
        k.l = stack0;
        if (S.wide_root & kRefLeafBit) return S.wide_root;
        float t0, f;
        box_test(f2u(S.root_box[0]), f2u(S.root_box[1]), f2u(S.root_box[2]), f2u(S.root_box[3]), f2u(S.root_box[4]), f2u(S.root_box[5]), r, t0, f);
        if (COUNT) n_inner++;
        return t0 <= min_quiet(f, r.tmax) ? S.wide_root : kRefEnd;
    }
};

// `cur` is an inner node whose box the ray is known to hit: deal with its two children.  A leaf child's sphere is in the
// record and is tested here, unless the child has to wait for its sibling's subtree.  (Sending every leaf child to the
// wave's next sphere pass instead, as the walks along exit links do, was slower: 824 ms against 760 on the 1 M-sphere
// scene.  It saves instructions, but each such leaf then costs a dependent fetch of its own.)
template <bool MOVING, bool COUNT>
DEV uint32_t visit_inner(const WideStore<MOVING, COUNT> &st, uint32_t cur, RayCtx &r, WideStack &k, uint32_t quick, uint32_t &n_inner, uint32_t &n_leaf) {
    const uint32_t idx = cur;
    uint4 q0, q1, q2, q3;
    if (st.in_top_copy(cur)) { q0 = load4(st.top, idx * 4u); q1 = load4(st.top, idx * 4u + 1u); q2 = load4(st.top, idx * 4u + 2u); q3 = load4(st.top, idx * 4u + 3u); }
    else { const uint4 *src = st.S.wide + (size_t)idx * 4u; q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }
    const uint32_t left = q3.x, right = q3.y & kRefMask;
    const uint32_t axis = (q3.y >> RRT_NODE_AXIS_SHIFT) & 3u;
    const bool right_first = ((r.oct >> axis) & 1u) != 0u;  // acceleration.rs:253-267
    // entry distance and scaled exit distance of both boxes (of a leaf child: of whatever its words are, never looked at)
    float lt0, lf, rt0, rf;
    st.box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, lt0, lf);
    st.box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r, rt0, rf);
    const uint32_t near_ref = right_first ? right : left, far_ref = right_first ? left : right;
    const float near_t0 = right_first ? rt0 : lt0, near_f = right_first ? rf : lf;
    const float far_t0 = right_first ? lt0 : rt0, far_f = right_first ? lf : rf;
    const bool near_leaf = (near_ref & kRefLeafBit) != 0u, far_leaf = (far_ref & kRefLeafBit) != 0u;
    if (near_leaf) {
        if (COUNT) n_leaf++;
        // (one copy of the sphere test for either side: the words are picked, not the code)
        st.leaf_test(right_first ? q1.z : q0.x, right_first ? q1.w : q0.y, right_first ? q2.x : q0.z, right_first ? q2.y : q0.w,
                     right_first ? q2.z : q1.x, right_first ? q2.w : q1.y, r, quick);
    } else {
        if (COUNT) n_inner++;
        if (near_t0 <= min_quiet(near_f, r.tmax)) {  // Aabb::intersect's last step (acceleration.rs:82-83); the far child waits
            // A waiting inner child is dropped here only if the ray leaves its box before it enters it, which nothing changes.  How
            // its entry compares with t_max is looked at when it is popped, as the reference does: t_max can turn NaN in between
            // (and grow again after that: a NaN t_max accepts any root above t_min).
            if (far_leaf) k.push(make_uint2(far_ref, 0u));
            else if (!(far_t0 > far_f)) {
                k.push(make_uint2(far_ref, f2u(far_f != far_f ? -far_t0 : far_t0)));  // (a NaN exit distance takes a NaN ray)
            }
            else if (COUNT) n_inner++;  // the reference pops it later, and it misses then too
            return near_ref;
        }
    }
    // the near child is done with (a miss, or a sphere): the far child is the next pop
    if (far_leaf) {
        if (COUNT) n_leaf++;
        st.leaf_test(right_first ? q0.x : q1.z, right_first ? q0.y : q1.w, right_first ? q0.z : q2.x, right_first ? q0.w : q2.y,
                     right_first ? q1.x : q2.z, right_first ? q1.y : q2.w, r, quick);
        return st.pop(r, k, n_inner);
    }
    if (COUNT) n_inner++;
    if (far_t0 <= min_quiet(far_f, r.tmax)) return far_ref;
    return st.pop(r, k, n_inner);
}

// a leaf that waited on the stack (or the root of a one-sphere scene): its words come from its own node
template <bool MOVING, bool COUNT>
DEV uint32_t visit_leaf(const WideStore<MOVING, COUNT> &st, uint32_t cur, RayCtx &r, WideStack &k, uint32_t quick, uint32_t &n_inner, uint32_t &n_leaf) {
    const uint32_t idx = cur & kRefIndexMask;
    const uint4 *src = (cur & kRefTopBit) ? st.S.top_nodes : st.S.nodes;
    const uint4 a = src[idx * 4u], b = src[idx * 4u + 1u];
    if (COUNT) n_leaf++;
    st.leaf_test(a.x, a.y, a.z, a.w, b.x, b.y, r, quick);
    return st.pop(r, k, n_inner);
}

// A wave's statistics are added to its row by atomics that return nothing: a plain `+=` is a load the wave has to wait for
// before it can store and end, a memory round trip at the very end of every wave of every launch.
DEV void stat_add(uint64_t *word, uint64_t v) { (void)__hip_atomic_fetch_add(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

DEV uint64_t wave_sum_u64(uint64_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wave's share of an extend launch.  Rays come from a pool (below): whenever lanes finish their ray they are given
// the next rays of the pool.  Inside a ray the walk alternates "descend through inner nodes" and "test one sphere" (the
// latter once `leaf_min` lanes wait for it), so that lanes run the same kind of step together.  Finished lanes wait
// until at least `retire_min` of them can be retired and restarted in one go (or nothing else is left).
template <class Store, bool MOVING, bool COUNT>
DEV void extend_wave(const Store &st, const SceneDev &S, const PathState &P, uint32_t wave, uint32_t lane, uint32_t retire_min,
                     uint32_t leaf_min, uint64_t *wave_stats, uint32_t own_count, uint32_t *pool_head, uint32_t pool_base, uint32_t pool_total,
                     bool pool_listed = false) {
    // Where the wave's rays come from.  Usually (pool_head != null) from a pool shared by the waves of the workgroup:
    // when every live-slot list of the workgroup is full, the lists are one run of consecutive slots, and idle lanes of
    // any wave take the next ones off a counter in LDS.  All waves then stay busy until the run is used up, instead of
    // each wave trailing off at the end of its own 256 rays.  Otherwise from the wave's own list.  `pool_listed`: the pool's
    // stretch of the list plane says in which order the pool's slots are taken (order_pool has written it).
    const bool pooled = pool_head != nullptr;
    const uint32_t base = pooled ? pool_base : wave * kWaveSlots;
    const uint32_t count = pooled ? pool_total : own_count;
    if (count == 0) return;
    uint32_t n_taken = 0;
    const bool uniform_times = st.uniform_times();
    const uint32_t kEnd = st.end();  // wave-uniform: references below it are inner nodes, above it leaves
    uint32_t n_inner = 0, n_leaf = 0;
    uint32_t d_desc_iters = 0, d_desc_lanes = 0, d_leaf_iters = 0, d_leaf_lanes = 0, d_retires = 0, d_retire_lanes = 0;  // COUNT only
    // (own list that is not full) the lane's column of the list, read up front so that no ray load waits for a list load
    static_assert(kWaveSlots == 256, "four list entries per lane");
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    if (!pooled && count != kWaveSlots) {
        q0 = lane < count ? P.queue[base + lane] : 0u;
        q1 = lane + 64u < count ? P.queue[base + lane + 64u] : 0u;
        q2 = lane + 128u < count ? P.queue[base + lane + 128u] : 0u;
        q3 = lane + 192u < count ? P.queue[base + lane + 192u] : 0u;
    }
    uint32_t slot = 0;
    bool has = false;
    uint32_t head = 0;  // first list position not handed out yet (wave-uniform)
    RayCtx r;
    typename Store::Stack stk;
    uint2 stack_spill[Store::kSpillEntries];  // (two-children walk) stack positions beyond those that LDS holds
    bind_spill(stk, stack_spill);
    ray_load(r, make_float4(0, 0, 0, 0), make_float4(1, 1, 1, 0), make_float4(1, 1, 1, 3), 0.001f, __FLT_MAX__, S, false);
    uint32_t cur = kEnd;
    const bool may_quick = (S.flags & kSceneBounded) != 0u && P.t_min >= 0x1p-40f;
    uint32_t quick = 0u;  // wave-uniform, 0 or 1: every ray the wave holds may be divided the short way (div_by_a)
    for (;;) {
        const bool idle = cur == kEnd;
        const uint64_t idle_m = __ballot(idle);
        const uint64_t ready = __ballot(idle && (has || head < count));
        if (~idle_m == 0ull || (uint32_t)__popcll(ready) >= retire_min) {
            if (COUNT) { d_retires++; d_retire_lanes += (uint32_t)__popcll(ready); }
            // every idle lane takes the next ray, in lane order
            const uint32_t n_idle = (uint32_t)__popcll(idle_m);
            if (pooled && head < count) {  // `head` doubles as "the pool is not known to be empty"
                uint32_t got = 0;
                if (lane == 0) got = atomicAdd(pool_head, n_idle);
                head = min(rfl(got), count);
            }
            const uint32_t pos = head + prefix_count(idle_m);
            uint32_t n_slot = base + pos;  // the pool, or a full list, is a run of consecutive slots
            if (!pooled && count != kWaveSlots) {  // list entry p sits in lane p % 64, register p / 64
                const uint32_t col = pos >> 6;
                const int src = (int)(pos & 63u);
                const uint32_t s0 = (uint32_t)__shfl((int)q0, src, 64), s1 = (uint32_t)__shfl((int)q1, src, 64);
                const uint32_t s2 = (uint32_t)__shfl((int)q2, src, 64), s3 = (uint32_t)__shfl((int)q3, src, 64);
                n_slot = col == 0u ? s0 : (col == 1u ? s1 : (col == 2u ? s2 : s3));
            }
            const bool take = idle && pos < count;
            if (pool_listed && take) n_slot = P.queue[base + pos];
            if (idle && has) {
                P.hit[slot] = make_uint2(f2u(r.tmax), r.best == kPrimNone ? kPrimNone : st.prim_of(r.best));
                has = false;
            }
            if (take) {
                slot = n_slot;
                const float4 nA = origin_load(P, n_slot, MOVING), nB = P.rayB[n_slot];  // MOVING scenes keep the time in the origin plane
                const float4 nC = ray_constants(mk(nB.x, nB.y, nB.z));  // 1/d and d.d, once per ray (types.rs:21-23, sphere.rs:71)
                ray_load(r, nA, nB, nC, P.t_min, P.t_max, S, uniform_times);  // lib.rs:57
                st.prepare(r);
                cur = st.start(r, stk, n_inner);
                has = true;
            }
            quick = (may_quick && __ballot(has && !quick_div_ok<MOVING>(r)) == 0ull) ? 1u : 0u;
            n_taken += (uint32_t)__popcll(__ballot(take));
            head += min(n_idle, count - head);  // pooled: reaches `count` once a grab runs past the end of the pool
            if (__ballot(has) == 0ull) break;
        }
        // descend until no lane is at an inner node, or enough lanes wait at a sphere for a test to be worth a pass
        for (;;) {  // two box tests per look at the number of lanes waiting at a sphere
            bool at_inner = cur < kEnd;
            uint64_t im = __ballot(at_inner);
            if (im == 0ull) break;
            if (COUNT) { d_desc_iters++; d_desc_lanes += (uint32_t)__popcll(im); }
#ifdef RRT_DRAIN_STATS  // (tools/drain_stats.py) the sphere-pass counters count the descent passes made with nothing left to take instead
            if (COUNT && head >= count) { d_leaf_iters++; d_leaf_lanes += (uint32_t)__popcll(im); }
#endif
            if (at_inner) {
                if (COUNT && !Store::kCountsItself) n_inner++;
                cur = visit_inner(st, cur, r, stk, quick, n_inner, n_leaf);
            }
            at_inner = cur < kEnd;
            im = __ballot(at_inner);
            if (im == 0ull) break;
            if (COUNT) { d_desc_iters++; d_desc_lanes += (uint32_t)__popcll(im); }
#ifdef RRT_DRAIN_STATS
            if (COUNT && head >= count) { d_leaf_iters++; d_leaf_lanes += (uint32_t)__popcll(im); }
#endif
            if (at_inner) {
                if (COUNT && !Store::kCountsItself) n_inner++;
                cur = visit_inner(st, cur, r, stk, quick, n_inner, n_leaf);
            }
            if ((uint32_t)__popcll(__ballot(cur > kEnd)) >= leaf_min) break;
        }
#ifndef RRT_DRAIN_STATS
        if (COUNT) { const uint64_t lm = __ballot(cur > kEnd); if (lm) { d_leaf_iters++; d_leaf_lanes += (uint32_t)__popcll(lm); } }
#endif
        if (cur > kEnd) {
            if (COUNT && !Store::kCountsItself) n_leaf++;
            cur = visit_leaf(st, cur, r, stk, quick, n_inner, n_leaf);
        }
    }
    uint64_t *stt = wave_stats + (size_t)wave * kWaveStatWords;
    if (COUNT) {
        const uint64_t si = wave_sum_u64(n_inner), sl = wave_sum_u64(n_leaf);
        if (lane == 0) {
            stat_add(&stt[ST_INNER], si); stat_add(&stt[ST_LEAF], sl);
#ifndef RRT_SHADE_TIMING
            stt[10] += d_desc_iters; stt[11] += d_desc_lanes; stt[12] += d_leaf_iters; stt[13] += d_leaf_lanes; stt[14] += d_retires;
            stt[15] += d_retire_lanes;
#endif
        }
    }
    if (lane == 0) stat_add(&stt[ST_RAYS], n_taken);
}

// (kLayoutFreshApart) The order in which a workgroup's pool of full lists is taken: the slots whose paths are under way, list
// by list, then the freshly started ones, list by list.  Every list arrives with its fresh slots first (`own_word`: the
// list's word of P.qcount); the order is written over the workgroup's stretch of the list plane, which nobody reads as lists
// while the lists are full (the shade kernel takes a full list to be every slot in order, and rewrites it).  `fresh`: one
// LDS word per wave of the workgroup.  Called by every wave of the workgroup; the last barrier makes the order visible (one
// workgroup runs on one CU: its waves read through the same cache).
DEV void order_pool(const PathState &P, uint32_t *fresh, uint32_t wave0, uint32_t n_lists, uint32_t own_word) {
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const bool mine = w < n_lists;
    const uint32_t n_fresh = own_word >> kListFreshShift;
    if (lane == 0) fresh[w] = mine ? n_fresh : 0u;
    uint32_t q[4] = {0u, 0u, 0u, 0u};
    if (mine) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) q[k] = P.queue[(wave0 + w) * kWaveSlots + lane + 64u * k];
    }
    __syncthreads();  // the counts are there
    uint32_t fresh_before = 0, fresh_total = 0;
    for (uint32_t b = 0; b < n_lists; b++) {
        const uint32_t v = rfl(fresh[b]);
        fresh_total += v;
        if (b < w) fresh_before += v;
    }
    __syncthreads();  // every list has been read
    if (mine) {
        const uint32_t at_fresh = n_lists * kWaveSlots - fresh_total + fresh_before, at_rest = w * kWaveSlots - fresh_before;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t e = lane + 64u * k;
            P.queue[wave0 * kWaveSlots + (e < n_fresh ? at_fresh + e : at_rest + (e - n_fresh))] = q[k];
        }
    }
    __syncthreads();
}

template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void k_extend(SceneDev S, PathState P, uint32_t retire_min, uint32_t leaf_min, uint64_t *wave_stats) {
    __shared__ uint32_t pool[2];  // next position of the workgroup's ray pool; sum of the workgroup's list lengths
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = P.wave_begin + blockIdx.x * (kBlockThreads / 64u);
    const uint32_t wave = wave0 + (threadIdx.x >> 6);
    const uint32_t own_count = wave < P.wave_end ? rfl(P.qcount[wave]) & kListCountMask : 0u;
    if (threadIdx.x < 2) pool[threadIdx.x] = 0u;
    __syncthreads();
    if (lane == 0 && own_count) atomicAdd(&pool[1], own_count);
    __syncthreads();
    if (wave >= P.wave_end) return;
    const uint32_t n_lists = min(kBlockThreads / 64u, P.wave_end - wave0);
    const uint32_t total = rfl(pool[1]);  // the same for every lane: say so, and the list bookkeeping stays in scalar registers
    const bool pooled = total == n_lists * kWaveSlots;
    const HbmStore<MOVING> st(S);
    extend_wave<HbmStore<MOVING>, MOVING, COUNT>(st, S, P, wave, lane, retire_min, leaf_min, wave_stats, own_count, pooled ? &pool[0] : nullptr,
                                                 wave0 * kWaveSlots, total);
}


template <bool MOVING, bool COUNT>
DEV void extend_lds_block(const SceneDev &S, const PathState &P, uint32_t retire_min, uint32_t leaf_min, uint64_t *wave_stats) {
    extern __shared__ uint4 lds4[];
#ifdef RRT_WAVE_TIMING
    const uint64_t wt0 = wall_clock64();
#endif
    const uint32_t n_vec = (S.small_bytes + 15u) / 16u;  // 16-byte pieces; the host pads the image to a whole number of them
    // next position of the workgroup's ray pool; sum of the workgroup's list lengths.  Behind the nodes, so that the nodes
    // start at LDS address 0 and a node reference needs no base added
    // (then a word per wave for order_pool)
    uint32_t *pool = reinterpret_cast<uint32_t *>(lds4 + n_vec);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = P.wave_begin + blockIdx.x * (kLdsBlockThreads / 64u);
    const uint32_t wave = wave0 + (threadIdx.x >> 6);
    const uint32_t own_word = wave < P.wave_end ? rfl(P.qcount[wave]) : 0u;
    const uint32_t own_count = own_word & kListCountMask;
    if (threadIdx.x < 2) pool[threadIdx.x] = 0u;
    for (uint32_t i = threadIdx.x; i < n_vec; i += kLdsBlockThreads) lds4[i] = S.small_nodes[i];
    __syncthreads();
    if (lane == 0 && own_count) atomicAdd(&pool[1], own_count);
    __syncthreads();
    const uint32_t n_lists = min(kLdsBlockThreads / 64u, P.wave_end - wave0);
    const uint32_t total = rfl(pool[1]);  // the same for every lane: say so, and the list bookkeeping stays in scalar registers
    const bool pooled = total == n_lists * kWaveSlots;
    const bool listed = pooled && (P.layout & kLayoutFreshApart) != 0u;  // the same for every wave of the workgroup
    if (listed) order_pool(P, pool + 4, wave0, n_lists, own_word);
    if (wave >= P.wave_end) return;
    const LdsStore<MOVING> st(S.small_root, S.small_end);  // the image starts at LDS address 0: a reference is an LDS address
#ifdef RRT_WAVE_TIMING
    const uint64_t wt1 = wall_clock64();
#endif
    extend_wave<LdsStore<MOVING>, MOVING, COUNT>(st, S, P, wave, lane, retire_min, leaf_min, wave_stats, own_count, pooled ? &pool[0] : nullptr,
                                                 wave0 * kWaveSlots, total, listed);
#ifdef RRT_WAVE_TIMING
    {   // time from the end of this wave's work to the end of the workgroup's: the wave's slot stays taken that long
        const uint64_t wt2 = wall_clock64();
        __syncthreads();
        if (lane == 0) { uint64_t *w = wave_stats + (size_t)wave * kWaveStatWords; w[10] += wt2 - wt0; w[11] += wt1 - wt0; w[12] += 1; w[15] += wall_clock64() - wt2; }
    }
#endif
}

// The kernel a render runs is held to 48 scalar registers: with more (the same code, nothing but the allocation) its
// launches share the chip worse with the shade kernel's, 6 % of the step time with two streams; alone it runs the same.
// The counting build does not fit in 48 without spilling and is not timed: it is a kernel of its own without the limit.
template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kLdsBlockThreads, 8) __attribute__((amdgpu_num_sgpr(48))) void k_extend_lds(SceneDev S, PathState P, uint32_t retire_min, uint32_t leaf_min,
                                                                  uint64_t *wave_stats) {
    static_assert(!COUNT, "the counting variant is k_count_extend_lds");
    extend_lds_block<MOVING, false>(S, P, retire_min, leaf_min, wave_stats);
}
template <bool MOVING>
__global__ __launch_bounds__(kLdsBlockThreads, 8) void k_count_extend_lds(SceneDev S, PathState P, uint32_t retire_min, uint32_t leaf_min,
                                                                        uint64_t *wave_stats) {
    extend_lds_block<MOVING, true>(S, P, retire_min, leaf_min, wave_stats);
}

// Scenes whose nodes stay in HBM: the top levels of the tree (every ray starts there) are walked in an LDS copy.
template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kLdsBlockThreads, 8) void k_extend_top(SceneDev S, PathState P, uint32_t retire_min, uint32_t leaf_min,
                                                                  uint64_t *wave_stats) {
    extern __shared__ uint4 lds4[];
    __shared__ uint32_t pool[2];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = P.wave_begin + blockIdx.x * (kLdsBlockThreads / 64u);
    const uint32_t wave = wave0 + (threadIdx.x >> 6);
    const uint32_t own_count = wave < P.wave_end ? rfl(P.qcount[wave]) & kListCountMask : 0u;
    if (threadIdx.x < 2) pool[threadIdx.x] = 0u;
    for (uint32_t i = threadIdx.x; i < S.n_top * 4u; i += kLdsBlockThreads) lds4[i] = S.top_nodes[i];
    __syncthreads();
    if (lane == 0 && own_count) atomicAdd(&pool[1], own_count);
    __syncthreads();
    const uint32_t n_lists = min(kLdsBlockThreads / 64u, P.wave_end - wave0);
    const uint32_t total = rfl(pool[1]);  // the same for every lane: say so, and the list bookkeeping stays in scalar registers
    const bool pooled = total == n_lists * kWaveSlots;
    if (wave >= P.wave_end) return;
    const HbmStore<MOVING, LdsNodePtr> st(S, (LdsNodePtr)lds4);
    extend_wave<HbmStore<MOVING, LdsNodePtr>, MOVING, COUNT>(st, S, P, wave, lane, retire_min, leaf_min, wave_stats, own_count, pooled ? &pool[0] : nullptr,
                                                 wave0 * kWaveSlots, total);
}

// The same with the two-children records (WideStore).  LDS: the first kWideLdsEntries stack positions of every lane
// (48 KB), then the records of the top seven levels of the tree (8 KB; a larger copy measured no faster).
constexpr uint32_t kWideTopLds = 127;
template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kLdsBlockThreads, 8) void k_extend_wide(SceneDev S, PathState P, uint32_t retire_min, uint32_t leaf_min,
                                                                   uint64_t *wave_stats) {
    extern __shared__ uint4 lds4[];
    __shared__ uint32_t pool[2];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave0 = P.wave_begin + blockIdx.x * (kLdsBlockThreads / 64u);
    const uint32_t wave = wave0 + (threadIdx.x >> 6);
    const uint32_t own_count = wave < P.wave_end ? rfl(P.qcount[wave]) & kListCountMask : 0u;
    if (threadIdx.x < 2) pool[threadIdx.x] = 0u;
    const uint32_t n_top_lds = min(S.n_top, kWideTopLds);
    uint4 *top_copy = lds4 + kWideLdsEntries * kLdsBlockThreads / 2u;  // behind the stacks (8 bytes per entry)
    for (uint32_t i = threadIdx.x; i < n_top_lds * 4u; i += kLdsBlockThreads) top_copy[i] = S.wide[i];
    __syncthreads();
    if (lane == 0 && own_count) atomicAdd(&pool[1], own_count);
    __syncthreads();
    const uint32_t n_lists = min(kLdsBlockThreads / 64u, P.wave_end - wave0);
    const uint32_t total = rfl(pool[1]);
    const bool pooled = total == n_lists * kWaveSlots;
    if (wave >= P.wave_end) return;
    const WideStore<MOVING, COUNT> st(S, (LdsNodePtr)top_copy, n_top_lds, (LdsStackPtr)lds4 + threadIdx.x);
    extend_wave<WideStore<MOVING, COUNT>, MOVING, COUNT>(st, S, P, wave, lane, retire_min, leaf_min, wave_stats, own_count, pooled ? &pool[0] : nullptr,
                                                        wave0 * kWaveSlots, total);
}

// (1 - cos)^5 of materials.rs:68, as a double-precision product rounded once to f32
DEV float pow5(float x) { const double d = (double)x, d2 = d * d, d4 = d2 * d2; return (float)(d4 * d); }
// materials.rs:65-69 with r0 = ((1 - ratio) / (1 + ratio))^2 taken from the sphere's table row (same f32 operations, done once on the host)
DEV float reflectance(float cosine, float r0) { return r0 + (1.0f - r0) * pow5(1.0f - cosine); }

// Path ids are dealt in groups of 64 consecutive ids, group g belonging to cursor shard g % n_shards; a wave draws
// from shard wave % n_shards.  `n` is the wave's running count of claims from its shard.
DEV uint64_t path_of_claim(const BatchDev &Bt, uint32_t shard, uint32_t n) { return ((uint64_t)(n >> 6) * Bt.n_shards + shard) * 64ull + (n & 63u); }

DEV bool start_path(const PathState &P, const BatchDev &Bt, const CameraDev &C, uint32_t shard, uint32_t n, uint32_t slot, bool timed, bool narrow) {
    const uint64_t p64 = path_of_claim(Bt, shard, n);
    if (p64 >= (uint64_t)Bt.n_paths) return false;
    float4 A, B;
    Rng g;
    generate_path(Bt, C, (uint32_t)p64, A, B, g);
    origin_store(P, slot, timed, A);
    P.rayB[slot] = B;
    P.rng[slot] = make_uint4(g.s0, g.s1, g.s2, g.s3);
    if (narrow) reinterpret_cast<uint2 *>(P.recent)[slot] = make_uint2(0u, 0u);
    else P.depth[slot] = 0u;
    return true;
}

// One claim of `cnt` path ids from the shard's cursor by lane 0, broadcast to the wave
DEV uint32_t claim_paths(const BatchDev &Bt, uint32_t lane, uint32_t shard, uint32_t cnt) {
    uint32_t base_n = 0;
    if (lane == 0 && cnt) base_n = atomicAdd(&Bt.cursors[shard * kCursorStride], cnt);
    return (uint32_t)__shfl((int)base_n, 0, 64);
}

__global__ __launch_bounds__(kBlockThreads) void k_raygen(PathState P, BatchDev Bt, CameraDev C, uint64_t *wave_stats) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = P.wave_begin + ((blockIdx.x * kBlockThreads + threadIdx.x) >> 6);
    const uint32_t base = wave * kWaveSlots;
    if (wave >= P.wave_end) return;
    const uint32_t shard = wave % Bt.n_shards;
    const uint32_t base_n = claim_paths(Bt, lane, shard, kWaveSlots);
    uint32_t out_count = 0;
    for (uint32_t t0 = 0; t0 < kWaveSlots; t0 += 64u) {
        const uint32_t slot = base + t0 + lane;
        const bool alive = start_path(P, Bt, C, shard, base_n + t0 + lane, slot, (P.layout & kLayoutTimed) != 0u, (P.layout & kLayoutNarrow) != 0u);
        const uint64_t am = __ballot(alive);
        if (alive) P.queue[base + out_count + prefix_count(am)] = slot;
        out_count += (uint32_t)__popcll(am);
    }
    if (lane == 0) {
        P.qcount[wave] = out_count | (out_count << kListFreshShift);  // every path of the list has just started
        if (out_count && Bt.alive_flag) Bt.alive_flag[(wave & 31u) * 32u] = 1u;  // 32 lines share the traffic
        stat_add(&wave_stats[(size_t)wave * kWaveStatWords + ST_PATHS], out_count);
    }
}

// Trail addressing: the levels of 64 neighbouring slots are kept together (level-major inside the group), so that a
// wave's trail traffic stays inside a few kilobytes whatever the depths of its paths.
DEV size_t trail_index(const PathState &P, uint32_t level, uint32_t slot) {
    return ((size_t)(slot >> 6) * P.trail_levels + level) * 64u + (slot & 63u);
}

// ---- shade ----
//
// A wave shades the (up to) 256 live slots of its list, four tiles of 64, in these steps:
//   1. classify every slot by what happened to its ray (the extend kernel packs the material kind of the sphere into
//      the hit record), park the generator states in LDS, and claim path ids for every path that is known to end
//      (sky, depth limit) with one cursor update;
//   2. run the in-unit-sphere rejection loop of every Lambertian and metal slot (rng.rs:32-40).  The slots that want a
//      sample are one work list of the wave: a lane whose sample has been accepted takes the next slot nobody has taken
//      (the generator states wait in LDS by list position, so any lane can make any slot's draws), and the loop runs as
//      long as the wave's total number of draws asks for, not as long as the unluckiest column of four slots.  What
//      waits in LDS is the generator state just before the accepted draw: step 3 repeats that one draw;
//   3. per tile: hit record, scatter, sky, trail bookkeeping; an ended path's slot starts its next path right there,
//      so that each array of the path state gets one store covering (nearly) all 64 neighbouring slots;
//   4. the few slots whose path was absorbed by a metal sphere (materials.rs:57-60) are refilled, compacted;
//   5. the live-slot list is rewritten in slot order (ballot + prefix popcount), so that the 64 lanes of a tile keep
//      touching neighbouring slots.
// (Sorting the list by class so that tiles run one material's code was tried: the scattered 16-byte stores that follow
// cost more HBM traffic than the saved instructions are worth.)
enum : uint32_t { CL_LAMB = 0, CL_METAL = 1, CL_DIEL = 2, CL_SKY = 3, CL_CUT = 4, CL_NONE = 5 };
static_assert(RRT_MAT_LAMBERTIAN == CL_LAMB && RRT_MAT_METAL == CL_METAL && RRT_MAT_DIELECTRIC == CL_DIEL, "material kinds double as classes");

// LDS words per wave: one 16-byte entry per list position
constexpr uint32_t kShadeLdsWords = kWaveSlots * 4u;
constexpr uint32_t kShadeListBytes = kWaveSlots;  // and one byte per list position: the positions that want a sample, compacted (step 2)
#ifndef RRT_SHADE_TABLE_THREADS
#define RRT_SHADE_TABLE_THREADS 512
#endif
#ifndef RRT_SHADE_TABLE_WAVES
#define RRT_SHADE_TABLE_WAVES 4
#endif
constexpr uint32_t kShadeTableThreads = RRT_SHADE_TABLE_THREADS;   // block size of the variant that keeps the per-sphere tables in LDS
static_assert(kShadeTableSpheres <= 1024, "the narrow trail entries have 10 bits");  // kernels.h: the largest scene the table variant takes, 2 x 16 bytes per sphere

// What step 3 reads of one path slot
struct SlotIn {
    float4 A, B;
    uint2 h;
    uint32_t depth;
    uint4 recent;  // the four most recent trail entries, newest first
    float4 sph;  // (tables in HBM only) the hit sphere's row of the sphere table, fetched with the rest of the state
};
// `hit_word` is the sphere word of the slot's hit record as step 1 saw it, or kPrimNone: with it the sphere's table row can
// be asked for at once, a tile ahead, instead of by a dependent gather in the middle of the tile.
template <bool PREFETCH_ROWS, bool TIMED, bool NARROW>
DEV void load_slot(const SceneDev &S, const PathState &P, uint32_t slot, uint32_t hit_word, SlotIn &in) {
    // Every lane loads, whether its list position is taken or not (`slot` is always a slot of this wave, a miss reads
    // row 0): a load under a branch makes the compiler wait for everything in flight where the branch joins, and these
    // loads are meant to stay in flight while the tile before them is shaded.
    __asm__ volatile("" : "+v"(slot));  // addresses are worked out here, from the slot number, not kept in registers since the classification
    in.A = origin_load(P, slot, TIMED); in.B = P.rayB[slot]; in.h = P.hit[slot]; meta_load(P, slot, NARROW, in.depth, in.recent);
    if (PREFETCH_ROWS) {
        const uint32_t row = hit_word != kPrimNone ? hit_word & kLeafPrimMask : 0u;
        in.sph = S.sph[row];
    }
}

// Where shade reads the per-sphere tables from: HBM through the caches, or (small scenes) a copy in the block's LDS.  An
// LDS read does not queue behind the kernel's outstanding stores the way a global load does.
struct GlobalTables {
    static constexpr bool kInLds = false;
    const float4 *sph, *att;
    DEV float4 sphere(uint32_t i) const { return sph[i]; }
    DEV float4 material(uint32_t i) const { return att[i]; }
};
struct LdsTables {
    static constexpr bool kInLds = true;
    const float4 *sph, *att;  // both in LDS
    DEV float4 sphere(uint32_t i) const { return sph[i]; }
    DEV float4 material(uint32_t i) const { return att[i]; }
};

// A freshly generated path: camera ray and generator state (lib.rs:93-101)
struct NewPath { float4 A, B; uint4 rng; bool ok; };

// The product of attenuations of the trail levels below `j`, innermost first (lib.rs:62), onto a radiance: read back
// eight levels per round so that a long path costs few memory round trips.
template <class Tables>
DEV void fold_trail(const PathState &P, const Tables &T, uint32_t slot, uint32_t j, float &rx, float &ry, float &rz) {
    const uint32_t *tr = P.trail + trail_index(P, 0u, slot);
    const size_t ns = 64u;  // level stride inside a 64-slot group
    while (j > 0u) {
        const uint32_t m = j < 8u ? j : 8u;
        uint32_t pr[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) pr[k] = k < m ? tr[(size_t)(j - 1u - k) * ns] : 0u;
#pragma unroll
        for (uint32_t half = 0; half < 8u; half += 4u) {
            float4 at[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) at[k] = T.material(pr[half + k]);
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++)
                if (half + k < m) { rx = at[k].x * rx; ry = at[k].y * ry; rz = at[k].z * rz; }
        }
        j -= m;
    }
}

// Step 3 for one lane.  `gw` is the slot's LDS entry: the generator state (before the accepted draw, for Lambertian and
// metal).  `np` is the path that takes over the slot if this one ends here.
template <bool MOVING, class Tables>
DEV void shade_slot(const SceneDev &S, const Tables &T, const PathState &P, const BatchDev &Bt, uint32_t cls, uint32_t slot,  // Tables::kInLds: the narrow state layout
                    const SlotIn &in, uint4 gw, const NewPath &np, bool &alive, bool &absorbed, bool &started, uint4 *ent, uint32_t &older) {
    const float4 A = in.A, B = in.B;
    const uint32_t depth = in.depth & 0xFFFFu;       // scatters so far
    const uint32_t trail_len = in.depth >> 16;       // of which recorded on the trail
    const uint32_t pid = f2u(B.w);
    const V3 o = mk(A.x, A.y, A.z), d = mk(B.x, B.y, B.z);
    const float time = A.w;
    float rx = 0.0f, ry = 0.0f, rz = 0.0f;  // radiance of a finished path
    // what the slot holds afterwards
    float4 oA = A, oB = B;
    Rng g;
    g.s0 = gw.x; g.s1 = gw.y; g.s2 = gw.z; g.s3 = gw.w;
    uint32_t o_depth = 0u;
    uint4 o_recent = in.recent;
    bool ended = false;
    // the draw step 2 accepted (Lambertian and metal), repeated from the generator state it left in the slot's entry
    V3 smp = mk(0.0f, 0.0f, 0.0f);
    if (cls <= CL_METAL) { smp.x = sample_pm1(g); smp.y = sample_pm1(g); smp.z = sample_pm1(g); }
    // Every class but the depth cut normalises exactly one vector: the sample (Lambertian, rng.rs:52-55) or the direction
    // (sky lib.rs:67, metal materials.rs:47, glass materials.rs:83).  One reciprocal magnitude serves whichever it is.
    float inv_len = 0.0f;
    if (cls <= CL_SKY) {
        const V3 v = cls == CL_LAMB ? smp : d;
        inv_len = recip_rn(sqrt_rn(dot3(v, v)));
    }
    if (cls == CL_SKY) {
        // sky (lib.rs:66-70), then the product of attenuations innermost first (lib.rs:62).  Dielectric hits are not on
        // the trail: their attenuation is (1,1,1) and x * 1 is x.
        const float uy = d.y * inv_len;
        const float t = 0.5f * (uy + 1.0f);
        const float w = 1.0f - t;
        rx = w + t * 0.5f;  // white.lerp(sky_blue, t) = white*(1-t) + sky_blue*t
        ry = w + t * 0.7f;
        rz = w + t * 1.0f;
        // the newest four levels travel with the path state; older ones are in the trail planes
        {
            const uint32_t w4[4] = {in.recent.x, in.recent.y, in.recent.z, in.recent.w};
            float4 at[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) at[k] = T.material(k < trail_len ? w4[k] : 0u);
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++)
                if (k < trail_len) { rx = at[k].x * rx; ry = at[k].y * ry; rz = at[k].z * rz; }
        }
        // Levels older than those four are in the trail planes, a memory round trip away with the whole tile waiting.
        // Where the caller can take them later (narrow state: a level count fits a byte) the path's radiance so far and
        // its id wait in the slot's LDS entry, and the caller folds the older levels of all such paths of the wave in
        // one go after the tiles; the planes keep them, the slot's next path writes none before its fifth scatter.
        const uint32_t j = trail_len > 4u ? trail_len - 4u : 0u;
        if (Tables::kInLds && j != 0u) {
            older = j;
            *ent = make_uint4(f2u(rx), f2u(ry), f2u(rz), pid);
        } else {
            fold_trail(P, T, slot, j, rx, ry, rz);
        }
        ended = true;
    } else if (cls == CL_CUT) {  // lib.rs:58-60
        ended = true;
    } else if (cls != CL_NONE) {
        const uint32_t prim = in.h.y & kLeafPrimMask;
        const float t = u2f(in.h.x);
        const float4 sp = Tables::kInLds ? T.sphere(prim) : in.sph;
        // (tables in HBM: the material row is gathered here, not a tile ahead with the sphere row: the eight registers that
        // holding it in both tile buffers takes are the difference between three and four waves per SIMD)
        const float4 mrow = T.material(prim);  // metal: .w = fuzz; glass: 1/ior, r0 seen from outside, r0 from inside, ior
        const float mat_param = mrow.w;
        V3 c = mk(sp.x, sp.y, sp.z);
        if (MOVING) {
            if (S.sph_info[prim] & kInfoMovingBit) {  // sphere.rs:61-64
                const float4 m = S.motion[prim];
                const float f = (S.flags & kSceneUniformTimes) ? (time - S.u_time0) / S.u_dt : (time - m.w) / S.motion_dt[prim];
                c = add(c, scale(f, mk(m.x, m.y, m.z)));
            }
        }
        // Hit record (sphere.rs:86-100)
        const V3 p = add(o, scale(t, d));
        const V3 pc = sub(p, c);
        // outward normal (p - c) / radius (sphere.rs:88): through the reciprocal of the radius where every lane's operands
        // are in div_known's range (the scene's radii by kSceneRadiiNormal, the numerators looked at here: a zero, a
        // denormal, a NaN or a huge component sends the wave to the division)
        V3 on;
        {
            const float lo = 0x1p-90f, hi = 0x1p90f;
            const bool in_range = fabsf(pc.x) >= lo && fabsf(pc.x) <= hi && fabsf(pc.y) >= lo && fabsf(pc.y) <= hi && fabsf(pc.z) >= lo && fabsf(pc.z) <= hi;
            float y = recip_newton(sp.w);
            if (__builtin_expect((S.flags & kSceneRadiiNormal) != 0u && __ballot(!in_range || recip_suspect(y)) == 0ull, 1))
                on = mk(div_known(pc.x, sp.w, y), div_known(pc.y, sp.w, y), div_known(pc.z, sp.w, y));
            else
                on = mk(pc.x / sp.w, pc.y / sp.w, pc.z / sp.w);
        }
        const bool front = dot3(d, on) < 0.0f;
        const V3 n = front ? on : neg(on);
        V3 dir;
        bool scattered = true;
        if (cls == CL_LAMB) {  // materials.rs:19-35
            const V3 unit = scale_r(smp, inv_len);  // ultraviolet normalized(): each component times 1 / |smp|
            dir = add(n, unit);
            if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = n;
        } else if (cls == CL_METAL) {  // materials.rs:44-61
            const float fuzz = mat_param;
            const V3 refl = reflected(scale_r(d, inv_len), n);
            dir = add(refl, scale(fuzz, smp));
            scattered = dot3(dir, n) > 0.0f;
        } else {  // materials.rs:75-104
            const float ratio = front ? mrow.x : mrow.w;  // 1.0 / ref_idx : ref_idx (materials.rs:80)
            const float r0 = front ? mrow.y : mrow.z;
            const V3 ud = scale_r(d, inv_len);
            const float cos_theta = fminf(dot3(neg(ud), n), 1.0f);
            const float sin_theta = sqrt_rn(1.0f - (cos_theta * cos_theta));
            bool reflect = ratio * sin_theta > 1.0f;
            if (!reflect) reflect = reflectance(cos_theta, r0) > gen_f32(g);  // the draw happens only here (materials.rs:89)
            if (reflect) {
                dir = reflected(ud, n);
            } else {  // ultraviolet refracted()
                const float ndi = dot3(n, ud);
                const float k = 1.0f - ratio * ratio * (1.0f - ndi * ndi);
                if (k < 0.0f) dir = mk(0.0f, 0.0f, 0.0f);
                else dir = sub(scale(ratio, ud), scale(ratio * ndi + sqrt_rn(k), n));
            }
        }
        if (scattered) {
            oA = make_float4(p.x, p.y, p.z, time);
            oB = make_float4(dir.x, dir.y, dir.z, B.w);
            const bool on_trail = cls != CL_DIEL;
            o_depth = (depth + 1u) | ((trail_len + (on_trail ? 1u : 0u)) << 16);
            if (on_trail) {
                // the entry pushed out of the four-deep window goes to the trail planes
                if (trail_len >= 4u) P.trail[trail_index(P, trail_len - 4u, slot)] = in.recent.w;
                o_recent = make_uint4(prim, in.recent.x, in.recent.y, in.recent.z);
            }
            alive = true;
        } else {  // lib.rs:64
            absorbed = true;
        }
    }
    if ((ended || absorbed) && older == 0u) {
        Bt.rad[pid] = make_float4(rx, ry, rz, 0.0f);  // one 16-byte store per finished path
    }
    if (ended && np.ok) {  // the slot's next path, generated beforehand (see k_shade)
        oA = np.A; oB = np.B;
        g.s0 = np.rng.x; g.s1 = np.rng.y; g.s2 = np.rng.z; g.s3 = np.rng.w;
        o_depth = 0u;
        alive = true;
        started = true;
    }
    if (alive) {
        origin_store(P, slot, MOVING, oA);
        P.rayB[slot] = oB;
        P.rng[slot] = make_uint4(g.s0, g.s1, g.s2, g.s3);
        meta_store(P, slot, Tables::kInLds, o_depth, o_recent);  // also where nothing changed: a store that covers whole lines costs the memory system less than a sparse one
    }
}

// Lane `src`'s new path, for every lane (12 ds_bpermute)
DEV NewPath fetch_path(const NewPath &g, int src) {
    NewPath r;
    r.A = make_float4(__shfl(g.A.x, src, 64), __shfl(g.A.y, src, 64), __shfl(g.A.z, src, 64), __shfl(g.A.w, src, 64));
    r.B = make_float4(__shfl(g.B.x, src, 64), __shfl(g.B.y, src, 64), __shfl(g.B.z, src, 64), __shfl(g.B.w, src, 64));
    r.rng = make_uint4((uint32_t)__shfl((int)g.rng.x, src, 64), (uint32_t)__shfl((int)g.rng.y, src, 64), (uint32_t)__shfl((int)g.rng.z, src, 64),
                       (uint32_t)__shfl((int)g.rng.w, src, 64));
    r.ok = __shfl(g.ok ? 1 : 0, src, 64) != 0;
    return r;
}

// One wave shades its list of `count` (> 0) live slots; returns the length of the list it leaves behind.  `l_ent` is the
// wave's 4 KB of LDS (one 16-byte entry per list position: the slot's generator state; after step 3 the same memory
// holds step 4's list of slots to refill), `l_list` its kShadeListBytes of LDS for step 2's work list.
template <bool MOVING, bool COUNT, class Tables>
DEV uint32_t shade_wave(const SceneDev &S, const PathState &P, const BatchDev &Bt, const CameraDev &C, const Tables &T, uint4 *l_ent, uint8_t *l_list,
                        uint32_t wave, uint32_t lane, uint32_t count, uint64_t *wave_stats) {
    constexpr bool TABLES = Tables::kInLds;
    const uint32_t base = wave * kWaveSlots;
    uint32_t *l_slot = reinterpret_cast<uint32_t *>(l_ent);
    const uint32_t shard = wave % Bt.n_shards;
    static_assert(kWaveSlots == 256, "four list entries per lane");

#ifdef RRT_SHADE_TIMING
    const uint64_t tk0 = __builtin_readcyclecounter();
#endif
    // ---- 1. classify ----
    uint32_t sq[4];        // the slots at list positions lane + 64 t
    uint32_t hw[4] = {kPrimNone, kPrimNone, kPrimNone, kPrimNone};  // (tables in HBM only) sphere word of their hit records
    uint32_t classes = 0;  // class of each of them, 4 bits apiece
    uint32_t end_before[4];  // wave-uniform: paths known to end (sky, depth limit) in the tiles before tile t
    uint32_t n_ending = 0;
    uint32_t c_ev[6] = {0, 0, 0, 0, 0, 0};  // HL, HM, HD, SKY, DEPTHCUT, ABSORBED
    SlotIn buf[2];  // tile t reads buf[t & 1] while tile t + 1 loads into the other
    {
        uint2 h[4];
        uint32_t dep[4];
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++) sq[t] = base + lane + 64u * t;  // a full list is every slot of the wave, in order
        if (count != kWaveSlots) {
            // entries past the end of the list are stale: their lanes keep the slot of their own position, which is one
            // of this wave's, so that every load below can go out whether the position is taken or not
            uint32_t q[4];
#pragma unroll
            for (uint32_t t = 0; t < 4u; t++) q[t] = P.queue[base + lane + 64u * t];
#pragma unroll
            for (uint32_t t = 0; t < 4u; t++) sq[t] = lane + 64u * t < count ? q[t] : sq[t];
        }
        // the first tile's ray and trail window travel together with everything the classification needs: all of it
        // is asked for back to back, with no branch in between (see load_slot), and costs one memory round trip
        // The generator states go from memory straight into the wave's LDS entries (lane i of column t lands in entry
        // 64 t + i), through no registers.
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P.rng + sq[t]),
                                             (__attribute__((address_space(3))) void *)(l_ent + 64u * t), 16, 0, 0);
        buf[0].A = origin_load(P, sq[0], MOVING); buf[0].B = P.rayB[sq[0]]; meta_load(P, sq[0], TABLES, buf[0].depth, buf[0].recent);
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++) { h[t] = P.hit[sq[t]]; dep[t] = depth_load(P, sq[t], TABLES); }
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the entries have landed
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++) {
            uint32_t c;
            if (!(lane + 64u * t < count)) c = CL_NONE;
            else if (h[t].y == kPrimNone) c = CL_SKY;
            else if ((dep[t] & 0xFFFFu) >= Bt.max_depth) c = CL_CUT;
            else c = h[t].y >> kLeafKindShift;
            classes |= c << (4u * t);
            if (!TABLES && c <= CL_DIEL) hw[t] = h[t].y;
            end_before[t] = n_ending;
            n_ending += (uint32_t)__popcll(__ballot(c == CL_SKY || c == CL_CUT));
            if (COUNT) {
#pragma unroll
                for (uint32_t k = 0; k < 5u; k++) c_ev[k] += (uint32_t)__popcll(__ballot(c == k));
            }
        }
        buf[0].h = h[0]; buf[0].depth = dep[0];
        if (!TABLES) { const uint32_t row = hw[0] != kPrimNone ? hw[0] & kLeafPrimMask : 0u; buf[0].sph = S.sph[row]; }
    }
    const uint32_t n_tiles = (count + 63u) >> 6;
    // one cursor update for every path known to end; the answer is not needed before step 3
    uint32_t claim_lane0 = 0;
    if (lane == 0 && n_ending) claim_lane0 = atomicAdd(&Bt.cursors[shard * kCursorStride], n_ending);

#ifdef RRT_SHADE_TIMING
    const uint64_t tk1 = __builtin_readcyclecounter();
#endif
    // ---- 2. in-unit-sphere rejection ----
    {
        // The list positions that want a sample, compacted (one byte each, in LDS).  A lane that has had a sample accepted
        // takes the next position nobody has taken, whichever lane's column it is in: the generator states wait in LDS by
        // position, so any lane can run any slot's draws.
        uint32_t n_want = 0;
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++) {
            const bool want = ((classes >> (4u * t)) & 15u) <= CL_METAL;
            const uint64_t wm = __ballot(want);
            if (want) l_list[n_want + prefix_count(wm)] = (uint8_t)(lane + 64u * t);
            n_want += (uint32_t)__popcll(wm);
        }
        uint32_t next = min(64u, n_want);  // first position of the compacted list not taken yet (wave-uniform)
        bool have = lane < n_want;
        uint32_t idx = have ? (uint32_t)l_list[lane] : 0u;
        Rng g = {0u, 0u, 0u, 0u};
        if (have) { const uint4 w = l_ent[idx]; g.s0 = w.x; g.s1 = w.y; g.s2 = w.z; g.s3 = w.w; }
        while (__ballot(have) != 0ull) {
            bool done = false;
            if (have) {
                const Rng before = g;
                const float x = sample_pm1(g), y = sample_pm1(g), z = sample_pm1(g);
                if (dot3(mk(x, y, z), mk(x, y, z)) < 1.0f) {
                    l_ent[idx] = make_uint4(before.s0, before.s1, before.s2, before.s3);
                    done = true;
                }
            }
            const uint64_t dm = __ballot(done);
            if (dm != 0ull) {
                const uint32_t k = next + prefix_count(dm);
                if (done) {
                    have = k < n_want;
                    if (have) { idx = (uint32_t)l_list[k]; const uint4 w = l_ent[idx]; g.s0 = w.x; g.s1 = w.y; g.s2 = w.z; g.s3 = w.w; }
                }
                next = min(next + (uint32_t)__popcll(dm), n_want);
            }
        }
    }

#ifdef RRT_SHADE_TIMING
    const uint64_t tk2 = __builtin_readcyclecounter();
#endif
    // ---- 3. tiles ----
    uint32_t absorbed_bits = 0, alive_bits = 0;
    uint32_t n_started = 0;
    uint32_t older_levels = 0;  // per column, a byte: trail levels still to be folded into the radiance parked in the slot's LDS entry
    const uint32_t claim0 = rfl(claim_lane0);
    // New paths for the slots whose path ends are generated 64 at a time with every lane busy (lane i makes the path of
    // the wave's ending slot gen_first + i) and stay in registers; the lane that owns an ending slot fetches its path
    // from there (ds_bpermute) when its tile comes up.
    NewPath gen;  // ray and generator state are meaningful only where .ok is set
    gen.ok = false;
    uint32_t gen_first = 0, gen_end = 0;  // ending slots [gen_first, gen_end) are in `gen`
#pragma unroll
    for (uint32_t t = 0; t < 4u; t++) {
        if (t < n_tiles) {
            if (t + 1u < 4u) load_slot<!TABLES, MOVING, TABLES>(S, P, sq[(t + 1u) & 3u], hw[(t + 1u) & 3u], buf[(t + 1u) & 1u]);
            const uint32_t pos = lane + 64u * t;
            const uint32_t cls = (classes >> (4u * t)) & 15u;
            bool alive = false, absorbed = false, started = false;
            // ending slots take the claimed path ids in list order
            const bool ending = cls == CL_SKY || cls == CL_CUT;
            const uint64_t end_m = __ballot(ending);
            const uint32_t end_rank = end_before[t] + prefix_count(end_m);
            const uint32_t end_hi = end_before[t] + (uint32_t)__popcll(end_m);
            NewPath np;  // read by the lanes of ending slots only, and those always fetch one below
            np.ok = false;
            bool first_fetch = true;
            for (uint32_t have = end_before[t]; have < end_hi;) {
                if (gen_end <= have) {  // generate the next 64
                    gen_first = gen_end;
                    gen_end = gen_first + 64u;
                    const uint64_t p64 = path_of_claim(Bt, shard, claim0 + gen_first + lane);
                    gen.ok = gen_first + lane < n_ending && p64 < (uint64_t)Bt.n_paths;
                    if (gen.ok) {
                        Rng g;
                        generate_path(Bt, C, (uint32_t)p64, gen.A, gen.B, g);
                        gen.rng = make_uint4(g.s0, g.s1, g.s2, g.s3);
                    }
                }
                const int src = (int)((end_rank - gen_first) & 63u);
                const bool mine = ending && end_rank >= have && end_rank < gen_end;
                const NewPath got = fetch_path(gen, src);
                if (first_fetch) np = got;  // lanes it is not meant for overwrite it in the next round, or never look at it
                else if (mine) np = got;
                first_fetch = false;
                have = min(gen_end, end_hi);
            }
            uint32_t slot = sq[t];
            __asm__ volatile("" : "+v"(slot));  // as in load_slot: the store addresses are worked out from the slot number when they are needed
            uint32_t older = 0u;
            if (pos < count) shade_slot<MOVING, Tables>(S, T, P, Bt, cls, slot, buf[t & 1u], l_ent[pos], np, alive, absorbed, started, &l_ent[pos], older);
            older_levels |= older << (8u * t);
            absorbed_bits |= absorbed ? (1u << t) : 0u;
            alive_bits |= (alive ? (1u << t) : 0u) | (started ? (16u << t) : 0u);  // bits 4..7: the slot's path has just started
            n_started += (uint32_t)__popcll(__ballot(started));
        }
    }

#ifdef RRT_SHADE_TIMING
    const uint64_t tk3 = __builtin_readcyclecounter();
#endif
    // ---- 3b. the paths that ended in the sky with more than four trail levels: the older levels, one pass for all ----
    if (TABLES) {
        while (__ballot(older_levels != 0u) != 0ull) {
            if (older_levels != 0u) {
                const uint32_t t = (uint32_t)(__ffs((int)older_levels) - 1) >> 3;  // the lane's first column with levels left
                const uint32_t j = (older_levels >> (8u * t)) & 0xFFu;
                const uint32_t slot = t == 0u ? sq[0] : (t == 1u ? sq[1] : (t == 2u ? sq[2] : sq[3]));
                const uint4 parked = l_ent[lane + 64u * t];
                float rx = u2f(parked.x), ry = u2f(parked.y), rz = u2f(parked.z);
                fold_trail(P, T, slot, j, rx, ry, rz);
                Bt.rad[parked.w] = make_float4(rx, ry, rz, 0.0f);
                older_levels &= ~(0xFFu << (8u * t));
            }
        }
    }

    // ---- 4. slots whose path was absorbed take the next paths of the batch, if any are left ----
    uint32_t n_need = 0;
    uint32_t rank[4];  // position of an absorbed slot in the wave's refill list
#pragma unroll
    for (uint32_t t = 0; t < 4u; t++) {
        const bool need = (absorbed_bits >> t) & 1u;
        const uint64_t nm = __ballot(need);
        rank[t] = n_need + prefix_count(nm);
        if (need) l_slot[rank[t]] = sq[t];
        n_need += (uint32_t)__popcll(nm);
    }
    uint32_t n_refilled = 0;
    if (n_need) {
        const uint32_t n = claim_paths(Bt, lane, shard, n_need);
        for (uint32_t i0 = 0; i0 < n_need; i0 += 64u) {
            const uint32_t i = i0 + lane;
            bool started = false;
            if (i < n_need) started = start_path(P, Bt, C, shard, n + i, l_slot[i], MOVING, TABLES);
            n_refilled += (uint32_t)__popcll(__ballot(started));  // path ids grow with i: the paths that still exist are a prefix of the list
        }
    }

#ifdef RRT_SHADE_TIMING
    const uint64_t tk4 = __builtin_readcyclecounter();
#endif
    // ---- 5. the live-slot list, in slot order (ballot + prefix popcount); kLayoutFreshApart: the slots whose path has just
    // started first, then the others, both in slot order ----
    const bool fresh_apart = (P.layout & kLayoutFreshApart) != 0u;
    uint32_t out_count = 0, n_fresh = 0;
    uint64_t am[4], fm[4];  // per column: live slots listed behind the fresh ones / fresh ones
#pragma unroll
    for (uint32_t t = 0; t < 4u; t++) {
        const bool refilled = ((absorbed_bits >> t) & 1u) && rank[t] < n_refilled;
        const bool alive = ((alive_bits >> t) & 1u) || refilled;
        const bool fresh = fresh_apart && alive && (((alive_bits >> (4u + t)) & 1u) || refilled);
        fm[t] = __ballot(fresh);
        am[t] = __ballot(alive && !fresh);
        n_fresh += (uint32_t)__popcll(fm[t]);
        out_count += (uint32_t)__popcll(fm[t]) + (uint32_t)__popcll(am[t]);
    }
    // a full list is every slot of the wave in order; if it was full and stays full, what is in memory is already right
    if (fresh_apart || !(count == kWaveSlots && out_count == kWaveSlots)) {
        uint32_t at_f = 0, at = n_fresh;
#pragma unroll
        for (uint32_t t = 0; t < 4u; t++) {
            const bool f = (fm[t] >> lane) & 1ull;
            if (f || ((am[t] >> lane) & 1ull)) P.queue[base + (f ? at_f + prefix_count(fm[t]) : at + prefix_count(am[t]))] = sq[t];
            at_f += (uint32_t)__popcll(fm[t]);
            at += (uint32_t)__popcll(am[t]);
        }
    }
    if (lane == 0) {
        P.qcount[wave] = out_count | (n_fresh << kListFreshShift);
        if (out_count && Bt.alive_flag) Bt.alive_flag[(wave & 31u) * 32u] = 1u;  // 32 lines share the traffic
        uint64_t *st = wave_stats + (size_t)wave * kWaveStatWords;
        stat_add(&st[ST_PATHS], n_started + n_refilled);
#ifdef RRT_SHADE_TIMING
        { const uint64_t tk5 = __builtin_readcyclecounter(); st[10] += tk1 - tk0; st[11] += tk2 - tk1; st[12] += tk3 - tk2; st[13] += tk4 - tk3; st[14] += tk5 - tk4; st[15] += 1; }
#endif
        if (COUNT) {
            stat_add(&st[ST_HL], c_ev[0]); stat_add(&st[ST_HM], c_ev[1]); stat_add(&st[ST_HD], c_ev[2]); stat_add(&st[ST_SKY], c_ev[3]);
            stat_add(&st[ST_DEPTHCUT], c_ev[4]); stat_add(&st[ST_ABSORBED], (uint32_t)n_need);
        }
    }
    return out_count;
}

template <bool MOVING, bool COUNT, bool TABLES>
__global__ __launch_bounds__(TABLES ? kShadeTableThreads : kBlockThreads, TABLES ? RRT_SHADE_TABLE_WAVES : (!MOVING ? 4 : 3)) void k_shade(SceneDev S, PathState P, BatchDev Bt, CameraDev C,
                                                                                      uint64_t *wave_stats) {
    constexpr uint32_t kThreads = TABLES ? kShadeTableThreads : kBlockThreads;
    extern __shared__ uint4 lds4[];  // per wave kShadeLdsWords words, then (TABLES) the sphere and material tables, then per wave kShadeListBytes
#ifdef RRT_WAVE_TIMING
    const uint64_t wt0 = wall_clock64();
#endif
    using Tables = typename std::conditional<TABLES, LdsTables, GlobalTables>::type;
    Tables T;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = P.wave_begin + ((blockIdx.x * kThreads + threadIdx.x) >> 6);
    // asked for before the tables so that the two round trips overlap (a wave past the end of the part reads the last list's)
    const uint32_t count_word = P.qcount[min(wave, P.wave_end - 1u)];
    if (TABLES) {
        uint4 *tab = lds4 + (kThreads / 64u) * (kShadeLdsWords / 4u);
        const uint4 *src_s = reinterpret_cast<const uint4 *>(S.sph), *src_a = reinterpret_cast<const uint4 *>(S.prim_att);
        for (uint32_t i = threadIdx.x; i < S.n_spheres; i += kThreads) { tab[i] = src_s[i]; tab[S.n_spheres + i] = src_a[i]; }
        __syncthreads();
        T.sph = reinterpret_cast<const float4 *>(tab);
        T.att = reinterpret_cast<const float4 *>(tab + S.n_spheres);
    } else {
        T.sph = S.sph;
        T.att = S.prim_att;
    }
    if (wave >= P.wave_end) return;
    const uint32_t count = rfl(count_word) & kListCountMask;
    if (count == 0) return;
    uint8_t *lists = reinterpret_cast<uint8_t *>(lds4 + (kThreads / 64u) * (kShadeLdsWords / 4u) + (TABLES ? 2u * S.n_spheres : 0u));  // behind the tables
    shade_wave<MOVING, COUNT, Tables>(S, P, Bt, C, T, lds4 + (threadIdx.x >> 6) * (kShadeLdsWords / 4u), lists + (threadIdx.x >> 6) * kShadeListBytes, wave, lane, count,
                                      wave_stats);
#ifdef RRT_WAVE_TIMING
    if (lane == 0) { uint64_t *w = wave_stats + (size_t)wave * kWaveStatWords; w[13] += wall_clock64() - wt0; w[14] += 1; }
#endif
}

// The end of a batch in one launch (scenes whose BVH image and tables fit in LDS together).  Once every path of the
// batch has been dealt, rounds get ever shorter: the surviving paths thin out over up to max_depth more bounces, and a
// round of two launches costs its launch latency however little it carries.  Here a wave runs extend and shade on its own
// list, turn by turn, until the list is empty: the lists of different waves share nothing but the path cursor.  (Also the
// whole of a render that is smaller than the path pool.  Before the paths are all dealt it would still be right, only
// slower than the two kernels: extend wants twice the waves that shade's registers leave room for.)
constexpr uint32_t kFinishThreads = 1024;
template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kFinishThreads, 4) void k_finish(SceneDev S, PathState P, BatchDev Bt, CameraDev C, uint32_t max_rounds,
                                                              uint32_t retire_min, uint32_t leaf_min, uint32_t *alive_flag,
                                                              uint64_t *wave_stats) {
    extern __shared__ uint4 lds4[];  // the BVH image (at LDS address 0, as LdsStore wants it), 4 KB per wave, the tables
    const uint32_t n_vec = (S.small_bytes + 15u) / 16u;
    uint4 *ents = lds4 + n_vec;
    uint4 *tab = ents + (kFinishThreads / 64u) * (kShadeLdsWords / 4u);
    const uint4 *src_s = reinterpret_cast<const uint4 *>(S.sph), *src_a = reinterpret_cast<const uint4 *>(S.prim_att);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = P.wave_begin + blockIdx.x * (kFinishThreads / 64u) + (threadIdx.x >> 6);
    uint32_t count = P.qcount[min(wave, P.wave_end - 1u)] & kListCountMask;
    {   // nothing left in this workgroup's lists: leave before the copies (no static LDS: the image must start at address 0)
        uint32_t *any = reinterpret_cast<uint32_t *>(ents);
        if (threadIdx.x == 0) *any = 0u;
        __syncthreads();
        if (wave < P.wave_end && count != 0u && lane == 0) *any = 1u;
        __syncthreads();
        if (*any == 0u) return;
    }
    for (uint32_t i = threadIdx.x; i < n_vec; i += kFinishThreads) lds4[i] = S.small_nodes[i];
    for (uint32_t i = threadIdx.x; i < S.n_spheres; i += kFinishThreads) { tab[i] = src_s[i]; tab[S.n_spheres + i] = src_a[i]; }
    __syncthreads();
    if (wave >= P.wave_end) return;
    count = rfl(count);
    LdsTables T;
    T.sph = reinterpret_cast<const float4 *>(tab);
    T.att = reinterpret_cast<const float4 *>(tab + S.n_spheres);
    const LdsStore<MOVING> st(S.small_root, S.small_end);
    uint4 *l_ent = ents + (threadIdx.x >> 6) * (kShadeLdsWords / 4u);
    for (uint32_t round = 0; round < max_rounds && count != 0u; round++) {
        extend_wave<LdsStore<MOVING>, MOVING, COUNT>(st, S, P, wave, lane, retire_min, leaf_min, wave_stats, count, nullptr, 0u, 0u);
        // what one lane stored the next step may read in another lane: every store has to have reached the cache the
        // wave reads through
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        count = shade_wave<MOVING, COUNT, LdsTables>(S, P, Bt, C, T, l_ent, reinterpret_cast<uint8_t *>(tab + 2u * S.n_spheres) + (threadIdx.x >> 6) * kShadeListBytes, wave, lane,
                                                     count, wave_stats);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (count != 0u && lane == 0 && alive_flag) alive_flag[(wave & 31u) * 32u] = 1u;
}

// ---- Merging the live lists before the finishing launch ----
//
// When the last path of a batch has been dealt, every list thins out from round to round, and a launch that gives every
// list a wave of its own keeps every workgroup of the pool busy for as long as its longest path lasts.  A list is nothing
// but the slot numbers a wave works on: the lists' entries are concatenated (in wave order) and cut into lists of
// `list_len` entries for the first waves, in a second list plane, and the finishing launch runs on those: a fraction of
// the workgroups, full of work.  No path state moves.  list_len is below kWaveSlots: a list of kWaveSlots entries is taken
// to be the wave's own slots in order.
__global__ __launch_bounds__(1024) void k_list_offsets(const uint32_t *qcount, uint32_t n_waves, uint32_t *offsets) {
    __shared__ uint32_t part[1024];  // one workgroup: offsets[w] = entries in the lists before list w, offsets[n_waves] = all
    const uint32_t per = (n_waves + 1023u) / 1024u;
    const uint32_t lo = min(n_waves, threadIdx.x * per), hi = min(n_waves, lo + per);
    uint32_t sum = 0;
    for (uint32_t w = lo; w < hi; w++) sum += qcount[w] & kListCountMask;
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t w = lo; w < hi; w++) { offsets[w] = run; run += qcount[w] & kListCountMask; }
    if (threadIdx.x == 1023u) offsets[n_waves] = part[1023];
}

__global__ __launch_bounds__(kBlockThreads) void k_merge_lists(PathState P, uint32_t *merged_queue, uint32_t *merged_qcount, const uint32_t *offsets, uint32_t list_len) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
    const uint32_t n_waves = P.wave_end;
    if (wave >= n_waves) return;
    const uint32_t count = rfl(P.qcount[wave]) & kListCountMask, before = rfl(offsets[wave]), total = rfl(offsets[n_waves]);
    // Lists of list_len entries hold a little less than the pool: while the lists are still (nearly) full there is nothing to
    // merge, and every list goes into the second plane as it is.
    const bool merge = (uint64_t)total <= (uint64_t)n_waves * list_len;
#pragma unroll
    for (uint32_t k = 0; k < kWaveSlots / 64u; k++) {
        const uint32_t pos = lane + 64u * k;
        if (pos < count) {
            const uint32_t slot = count == kWaveSlots ? wave * kWaveSlots + pos : P.queue[wave * kWaveSlots + pos];  // a full list: the wave's slots in order
            const uint32_t at = before + pos;
            const uint32_t list = at / list_len;
            merged_queue[merge ? list * kWaveSlots + (at - list * list_len) : wave * kWaveSlots + pos] = slot;
        }
    }
    if (lane == 0) {  // the length of merged list number `wave`
        const uint64_t begin = (uint64_t)wave * list_len;
        merged_qcount[wave] = !merge ? count : (total > begin ? (uint32_t)min<uint64_t>(list_len, total - begin) : 0u);
    }
}

__global__ void k_accumulate(float *accum, uint32_t npix, const float4 *rad, uint32_t n_samples, int first) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= npix) return;
    float ax = first ? 0.0f : accum[l], ay = first ? 0.0f : accum[(size_t)npix + l], az = first ? 0.0f : accum[2 * (size_t)npix + l];
    const float4 *r = rad + l;
    for (uint32_t s = 0; s < n_samples; s++) {  // c += trace(..) (lib.rs:102), sample by sample
        const float4 v = r[(size_t)s * npix];
        ax += v.x; ay += v.y; az += v.z;
    }
    accum[l] = ax; accum[(size_t)npix + l] = ay; accum[2 * (size_t)npix + l] = az;
}

// Rust `as u8`: saturating, NaN -> 0, truncation toward zero (lib.rs:106-108)
DEV uint8_t as_u8(float x) {
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)(int)x;
}
DEV uint8_t quantise(float sum, float fns) { return as_u8(sqrtf(sum / fns) * 255.99f); }  // lib.rs:105

__global__ void k_finalize(const float *accum, uint32_t npix, const uint32_t *pixel_map, uint32_t ns, uint8_t *rgb8, float *accum_out) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= npix) return;
    const uint32_t idx = pixel_map ? pixel_map[l] : l;
    const float fns = (float)ns;
    for (uint32_t c = 0; c < 3u; c++) {
        const float v = accum[(size_t)c * npix + l];
        if (rgb8) rgb8[(size_t)idx * 3u + c] = quantise(v, fns);
        if (accum_out) accum_out[(size_t)idx * 3u + c] = v;
    }
}

__global__ void k_finalize_interleaved(const float *accum_rgb, uint32_t n_values, uint32_t ns, uint8_t *rgb8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    rgb8[i] = quantise(accum_rgb[i], (float)ns);
}

// Bvh::hit (acceleration.rs:221-272) alone, for caller-supplied rays: ray i takes path slot i, every wave's live list is
// the identity over its slots (the last one shorter), and the extend kernel of the render loop does the walk.
__global__ void k_intersect_setup(PathState P, uint32_t n, const float *org, const float *dir, const float *time) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 d = mk(dir[i * 3u], dir[i * 3u + 1u], dir[i * 3u + 2u]);
    origin_store(P, i, (P.layout & kLayoutTimed) != 0u, make_float4(org[i * 3u], org[i * 3u + 1u], org[i * 3u + 2u], time ? time[i] : 0.0f));
    P.rayB[i] = make_float4(d.x, d.y, d.z, 0.0f);
    P.queue[i] = i;
    if ((i & (kWaveSlots - 1u)) == 0u) P.qcount[i / kWaveSlots] = min(kWaveSlots, n - i);
}

__global__ void k_intersect_gather(PathState P, uint32_t n, float *out_t, uint32_t *out_obj) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 h = P.hit[i];
    out_t[i] = u2f(h.x);
    out_obj[i] = h.y == kPrimNone ? kPrimNone : (h.y & kLeafPrimMask);
}

template <class T>
__global__ void k_scatter_pixels(const T *packed, const uint32_t *map, uint32_t n_values, T *image) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    const uint32_t l = i / 3u, c = i - l * 3u;
    image[(size_t)map[l] * 3u + c] = packed[i];
}

__global__ void k_add_f32(float *dst, const float *src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

__global__ void k_reduce_stats(const uint64_t *wave_stats, uint32_t n_waves, uint64_t *out16) {
    __shared__ uint64_t part[256];
    const uint32_t k = blockIdx.x;  // one counter per block
    uint64_t s = 0;
    for (uint32_t w = threadIdx.x; w < n_waves; w += 256u) s += wave_stats[(size_t)w * kWaveStatWords + k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out16[k] = part[0];
}

inline uint32_t wave_grid(const PathState &P) { return (P.wave_end - P.wave_begin + 3u) / 4u; }

// A kernel whose only purpose is its private segment: 256 bytes per lane, more than any kernel of the render path that the
// no-allocation entry point can launch reserves (the moving-sphere shade variant's descriptor asks for 36, the frame slots
// of scalar registers it keeps in vector-register lanes).  The runtime sizes a queue's scratch memory when a kernel that
// needs more than the queue has is first dispatched on it; launched once per stream at context creation, this one makes
// that happen there.
__global__ void k_prime_scratch(uint32_t *out, uint32_t n) {
    volatile uint32_t a[64];
    for (uint32_t i = 0; i < 64u; i++) a[i] = i * n;
    uint32_t v = 0;
    for (uint32_t i = 0; i < n; i++) v += a[(v + i * 7u) & 63u];
    if (v == 0xFFFFFFFFu) out[0] = v;
}

}  // namespace

void launch_prime_scratch(hipStream_t s, uint32_t *any_device_word) { hipLaunchKernelGGL(k_prime_scratch, dim3(1), dim3(64), 0, s, any_device_word, 3u); }

void launch_raygen(hipStream_t s, const PathState &P, const BatchDev &B, const CameraDev &C, uint64_t *wave_stats) {
    hipLaunchKernelGGL(k_raygen, dim3(wave_grid(P)), dim3(kBlockThreads), 0, s, P, B, C, wave_stats);
}

void launch_extend(hipStream_t s, const SceneDev &S, const PathState &P, bool moving, bool count, uint32_t retire_min,
                   uint32_t leaf_min, uint64_t *wave_stats) {
    if (S.small_nodes) {  // scene fits in LDS
        const uint32_t waves_per_block = kLdsBlockThreads / 64u;
        const dim3 g((P.wave_end - P.wave_begin + waves_per_block - 1u) / waves_per_block), b(kLdsBlockThreads);
        const size_t lds = (((size_t)S.small_bytes + 15u) & ~(size_t)15u) + 16u + 4u * waves_per_block;  // the image, the pool counters, a word per wave
        if (moving) {
            if (count) hipLaunchKernelGGL((k_count_extend_lds<true>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
            else hipLaunchKernelGGL((k_extend_lds<true, false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
        } else {
            if (count) hipLaunchKernelGGL((k_count_extend_lds<false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
            else hipLaunchKernelGGL((k_extend_lds<false, false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
        }
        return;
    }
    if (S.top_nodes && S.n_top && S.wide) {
        const uint32_t waves_per_block = kLdsBlockThreads / 64u;
        const dim3 g((P.wave_end - P.wave_begin + waves_per_block - 1u) / waves_per_block), b(kLdsBlockThreads);
        const size_t lds = (size_t)kWideLdsEntries * kLdsBlockThreads * 8u + (size_t)std::min(S.n_top, kWideTopLds) * kNodeWords * 4u;
        if (moving) {
            if (count) hipLaunchKernelGGL((k_extend_wide<true, true>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
            else hipLaunchKernelGGL((k_extend_wide<true, false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
        } else {
            if (count) hipLaunchKernelGGL((k_extend_wide<false, true>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
            else hipLaunchKernelGGL((k_extend_wide<false, false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
        }
        return;
    }
    if (S.top_nodes && S.n_top) {
        const uint32_t waves_per_block = kLdsBlockThreads / 64u;
        const dim3 g((P.wave_end - P.wave_begin + waves_per_block - 1u) / waves_per_block), b(kLdsBlockThreads);
        const size_t lds = (size_t)S.n_top * kNodeWords * 4u;
        if (moving) {
            if (count) hipLaunchKernelGGL((k_extend_top<true, true>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
            else hipLaunchKernelGGL((k_extend_top<true, false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
        } else {
            if (count) hipLaunchKernelGGL((k_extend_top<false, true>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
            else hipLaunchKernelGGL((k_extend_top<false, false>), g, b, lds, s, S, P, retire_min, leaf_min, wave_stats);
        }
        return;
    }
    const dim3 g(wave_grid(P)), b(kBlockThreads);
    if (moving) {
        if (count) hipLaunchKernelGGL((k_extend<true, true>), g, b, 0, s, S, P, retire_min, leaf_min, wave_stats);
        else hipLaunchKernelGGL((k_extend<true, false>), g, b, 0, s, S, P, retire_min, leaf_min, wave_stats);
    } else {
        if (count) hipLaunchKernelGGL((k_extend<false, true>), g, b, 0, s, S, P, retire_min, leaf_min, wave_stats);
        else hipLaunchKernelGGL((k_extend<false, false>), g, b, 0, s, S, P, retire_min, leaf_min, wave_stats);
    }
}

void launch_shade(hipStream_t s, const SceneDev &S, const PathState &P, const BatchDev &B, const CameraDev &C, bool moving,
                  bool count, uint64_t *wave_stats) {
    const uint32_t n_waves = P.wave_end - P.wave_begin;
    if (P.layout & kLayoutNarrow) {  // set by the host for scenes of at most kShadeTableSpheres spheres (and a depth limit the narrow state holds)
        const uint32_t wpb = kShadeTableThreads / 64u;
        const dim3 g((n_waves + wpb - 1u) / wpb), b(kShadeTableThreads);
        const size_t lds = (size_t)wpb * (kShadeLdsWords * 4u + kShadeListBytes) + (size_t)S.n_spheres * 32u;
        if (moving) {
            if (count) hipLaunchKernelGGL((k_shade<true, true, true>), g, b, lds, s, S, P, B, C, wave_stats);
            else hipLaunchKernelGGL((k_shade<true, false, true>), g, b, lds, s, S, P, B, C, wave_stats);
        } else {
            if (count) hipLaunchKernelGGL((k_shade<false, true, true>), g, b, lds, s, S, P, B, C, wave_stats);
            else hipLaunchKernelGGL((k_shade<false, false, true>), g, b, lds, s, S, P, B, C, wave_stats);
        }
        return;
    }
    const dim3 g(wave_grid(P)), b(kBlockThreads);
    const size_t lds = (size_t)(kBlockThreads / 64u) * (kShadeLdsWords * 4u + kShadeListBytes);
    if (moving) {
        if (count) hipLaunchKernelGGL((k_shade<true, true, false>), g, b, lds, s, S, P, B, C, wave_stats);
        else hipLaunchKernelGGL((k_shade<true, false, false>), g, b, lds, s, S, P, B, C, wave_stats);
    } else {
        if (count) hipLaunchKernelGGL((k_shade<false, true, false>), g, b, lds, s, S, P, B, C, wave_stats);
        else hipLaunchKernelGGL((k_shade<false, false, false>), g, b, lds, s, S, P, B, C, wave_stats);
    }
}

bool finish_fits(const SceneDev &S, const PathState &P) {
    return S.small_nodes != nullptr && (P.layout & kLayoutNarrow) != 0u && S.n_spheres <= kShadeTableSpheres;
}

int launch_finish(hipStream_t s, const SceneDev &S, const PathState &P, const BatchDev &B, const CameraDev &C, bool moving,
                  bool count, uint32_t max_rounds, uint32_t retire_min, uint32_t leaf_min, uint32_t *alive_flag, uint64_t *wave_stats) {
    const uint32_t wpb = kFinishThreads / 64u;
    const dim3 g((P.wave_end - P.wave_begin + wpb - 1u) / wpb), b(kFinishThreads);
    const size_t lds = (((size_t)S.small_bytes + 15u) & ~(size_t)15u) + (size_t)wpb * (kShadeLdsWords * 4u + kShadeListBytes) + (size_t)S.n_spheres * 32u;
    // more than 64 KB of LDS per workgroup has to be asked for once per kernel and device
    static std::mutex mu;
    static std::set<std::pair<int, int>> allowed;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    const int variant = (moving ? 2 : 0) | (count ? 1 : 0);
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!allowed.count({dev, variant})) {
            const void *f = variant == 3 ? reinterpret_cast<const void *>(&k_finish<true, true>)
                          : variant == 2 ? reinterpret_cast<const void *>(&k_finish<true, false>)
                          : variant == 1 ? reinterpret_cast<const void *>(&k_finish<false, true>)
                                         : reinterpret_cast<const void *>(&k_finish<false, false>);
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
            allowed.insert({dev, variant});
        }
    }
    BatchDev Bq = B;
    Bq.alive_flag = nullptr;  // a wave reports once, when it stops, not after every round
    if (moving) {
        if (count) hipLaunchKernelGGL((k_finish<true, true>), g, b, lds, s, S, P, Bq, C, max_rounds, retire_min, leaf_min, alive_flag, wave_stats);
        else hipLaunchKernelGGL((k_finish<true, false>), g, b, lds, s, S, P, Bq, C, max_rounds, retire_min, leaf_min, alive_flag, wave_stats);
    } else {
        if (count) hipLaunchKernelGGL((k_finish<false, true>), g, b, lds, s, S, P, Bq, C, max_rounds, retire_min, leaf_min, alive_flag, wave_stats);
        else hipLaunchKernelGGL((k_finish<false, false>), g, b, lds, s, S, P, Bq, C, max_rounds, retire_min, leaf_min, alive_flag, wave_stats);
    }
    return 0;
}

void launch_merge_lists(hipStream_t s, const PathState &P, uint32_t *merged_queue, uint32_t *merged_qcount, uint32_t *offsets, uint32_t list_len) {
    hipLaunchKernelGGL(k_list_offsets, dim3(1), dim3(1024), 0, s, P.qcount, P.wave_end, offsets);
    hipLaunchKernelGGL(k_merge_lists, dim3((P.wave_end + 3u) / 4u), dim3(kBlockThreads), 0, s, P, merged_queue, merged_qcount, offsets, list_len);
}

void launch_accumulate(hipStream_t s, float *accum, uint32_t npix, const float4 *rad, uint32_t n_samples,
                       bool first_batch) {
    hipLaunchKernelGGL(k_accumulate, dim3((npix + 255u) / 256u), dim3(256), 0, s, accum, npix, rad, n_samples,
                       first_batch ? 1 : 0);
}

void launch_finalize(hipStream_t s, const float *accum, uint32_t npix, const uint32_t *pixel_map, uint32_t ns, uint8_t *rgb8,
                     float *accum_out) {
    hipLaunchKernelGGL(k_finalize, dim3((npix + 255u) / 256u), dim3(256), 0, s, accum, npix, pixel_map, ns, rgb8, accum_out);
}

void launch_finalize_interleaved(hipStream_t s, const float *accum_rgb, uint32_t n_pixels, uint32_t ns, uint8_t *rgb8) {
    const uint32_t n = n_pixels * 3u;
    hipLaunchKernelGGL(k_finalize_interleaved, dim3((n + 255u) / 256u), dim3(256), 0, s, accum_rgb, n, ns, rgb8);
}

void launch_intersect_setup(hipStream_t s, const PathState &P, uint32_t n, const float *org, const float *dir, const float *time) {
    hipLaunchKernelGGL(k_intersect_setup, dim3((n + 255u) / 256u), dim3(256), 0, s, P, n, org, dir, time);
}

void launch_intersect_gather(hipStream_t s, const PathState &P, uint32_t n, float *out_t, uint32_t *out_obj) {
    hipLaunchKernelGGL(k_intersect_gather, dim3((n + 255u) / 256u), dim3(256), 0, s, P, n, out_t, out_obj);
}

void launch_scatter_rgb8(hipStream_t s, const uint8_t *packed, const uint32_t *map, uint32_t n_pixels, uint8_t *image) {
    const uint32_t n = n_pixels * 3u;
    if (n) hipLaunchKernelGGL((k_scatter_pixels<uint8_t>), dim3((n + 255u) / 256u), dim3(256), 0, s, packed, map, n, image);
}

void launch_scatter_f32(hipStream_t s, const float *packed, const uint32_t *map, uint32_t n_pixels, float *image) {
    const uint32_t n = n_pixels * 3u;
    if (n) hipLaunchKernelGGL((k_scatter_pixels<float>), dim3((n + 255u) / 256u), dim3(256), 0, s, packed, map, n, image);
}

void launch_add_f32(hipStream_t s, float *dst, const float *src, size_t n) {
    if (n) hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, s, dst, src, n);
}

void launch_reduce_stats(hipStream_t s, const uint64_t *wave_stats, uint32_t n_waves, uint64_t *out16) {
    hipLaunchKernelGGL(k_reduce_stats, dim3(kWaveStatWords), dim3(256), 0, s, wave_stats, n_waves, out16);
}

}  // namespace rrt_amd
