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
#include "kernels.h"
#include "scene_layout.h"

namespace rrt_amd {
namespace {

#define DEV __device__ __forceinline__

DEV float u2f(uint32_t u) { return __uint_as_float(u); }
DEV uint32_t f2u(float f) { return __float_as_uint(f); }
DEV uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

struct V3 { float x, y, z; };
DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 scale(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// ultraviolet Vec4::dot with both w lanes zero: x.mul_add(ox, y.mul_add(oy, z.mul_add(oz, 0)))
DEV float dot3(V3 a, V3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, __builtin_fmaf(a.z, b.z, 0.0f))); }
// ultraviolet normalized(): reciprocal of the magnitude, then one multiply per lane
DEV V3 normalized(V3 a) { const float r = 1.0f / sqrtf(dot3(a, a)); return mk(a.x * r, a.y * r, a.z * r); }
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
DEV float sample_pm1(Rng &g) { const float v = (float)(next_u32(g) >> 9) * 0x1p-23f; return v * RRT_PM1_SCALE + -1.0f; }

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

DEV V3 in_unit_sphere(Rng &g) {  // rng.rs:32-40
    for (;;) {
        const float x = sample_pm1(g), y = sample_pm1(g), z = sample_pm1(g);
        const V3 v = mk(x, y, z);
        if (dot3(v, v) < 1.0f) return v;
    }
}

// ---- path generation (lib.rs:93-101, camera.rs:56-66) ----
DEV void generate_path(const BatchDev &Bt, const CameraDev &C, uint32_t p, float4 &A, float4 &B, Rng &g) {
    const uint32_t sl = p / Bt.npix;
    const uint32_t pl = p - sl * Bt.npix;
    const uint32_t idx = Bt.pixel_map ? Bt.pixel_map[pl] : pl;
    const uint32_t k = Bt.k_first + sl * Bt.k_step;
    seed_from_u64(g, Bt.seed ^ ((uint64_t)idx * 0x9E3779B97F4A7C15ull + (uint64_t)k * 0xD1B54A32D192ED03ull));
    const float x = (float)(idx % Bt.nx);
    const float y = (float)(Bt.ny - 1u - idx / Bt.nx);
    const float s = (x + gen_f32(g)) / (float)Bt.nx;
    const float t = (y + gen_f32(g)) / (float)Bt.ny;
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
    const float tv = (float)(next_u32(g) >> 9) * 0x1p-23f;
    const float time = tv * C.time_scale + C.time_low;  // rand Uniform::sample (camera.rs:64)
    A = make_float4(origin.x, origin.y, origin.z, time);
    B = make_float4(dir.x, dir.y, dir.z, u2f(p));
}

// ---- BVH traversal (acceleration.rs:221-272) over the linked layout of scene_layout.h ----
template <bool MOVING, bool COUNT>
DEV void traverse(const SceneDev &S, V3 o, V3 d, float time, float tmin, float &tmax, uint32_t &best, uint32_t &n_inner,
                  uint32_t &n_leaf) {
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // types.rs:21-23
    const bool ngx = inv.x < 0.0f, ngy = inv.y < 0.0f, ngz = inv.z < 0.0f;  // acceleration.rs:230
    const uint32_t oct = (ngx ? 1u : 0u) | (ngy ? 2u : 0u) | (ngz ? 4u : 0u);
    const float a = dot3(d, d);  // sphere.rs:71, the same for every sphere
    float f_ray = 0.0f;
    if (MOVING && (S.flags & kSceneUniformTimes)) f_ray = (time - S.u_time0) / S.u_dt;
    const uint32_t *nw = reinterpret_cast<const uint32_t *>(S.nodes);
    uint32_t cur = S.root_ref;
    while (cur != kRefEnd) {
        const uint32_t idx = cur & kRefIndexMask;
        const uint4 n0 = S.nodes[idx * 4u];
        const uint4 n1 = S.nodes[idx * 4u + 1u];
        const uint32_t exit_ref = nw[idx * kNodeWords + 8u + oct];
        if (cur & kRefLeafBit) {
            // Sphere::hit (sphere.rs:68-85); the Hit record itself is rebuilt in shade for the final hit only
            if (COUNT) n_leaf++;
            V3 c = mk(u2f(n0.x), u2f(n0.y), u2f(n0.z));
            const float rr = u2f(n0.w);
            const uint32_t prim = n1.x;
            if (MOVING && (n1.y & 1u)) {  // center_at (sphere.rs:61-64)
                const float4 m = S.motion[prim];
                const float f = (S.flags & kSceneUniformTimes) ? f_ray : (time - m.w) / S.motion_dt[prim];
                c = add(c, scale(f, mk(m.x, m.y, m.z)));
            }
            const V3 oc = sub(o, c);
            const float half_b = dot3(oc, d);
            const float cc = dot3(oc, oc) - rr;
            const float disc = (half_b * half_b) - (a * cc);
            if (!(disc < 0.0f)) {
                const float sq = sqrtf(disc);
                float root = (-half_b - sq) / a;
                bool outside = root < tmin || tmax < root;
                if (outside) {
                    root = (-half_b + sq) / a;
                    outside = root < tmin || tmax < root;
                }
                if (!outside) { tmax = root; best = prim; }
            }
            cur = exit_ref;
        } else {
            // Aabb::intersect (acceleration.rs:60-84) against the current t_max
            if (COUNT) n_inner++;
            const float lox = u2f(n0.x), loy = u2f(n0.y), loz = u2f(n0.z), hix = u2f(n0.w), hiy = u2f(n1.x), hiz = u2f(n1.y);
            const float tnx = ((ngx ? hix : lox) - o.x) * inv.x;
            const float tny = ((ngy ? hiy : loy) - o.y) * inv.y;
            const float tnz = ((ngz ? hiz : loz) - o.z) * inv.z;
            const float tfx = (((ngx ? lox : hix) - o.x) * inv.x) * RRT_FAR_SCALE;
            const float tfy = (((ngy ? loy : hiy) - o.y) * inv.y) * RRT_FAR_SCALE;
            const float tfz = (((ngz ? loz : hiz) - o.z) * inv.z) * RRT_FAR_SCALE;
            const float t0 = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), tmin);
            const float t1 = fminf(fminf(fminf(tfx, tfy), tfz), tmax);
            if (t0 <= t1) {
                // children are visited near side first (acceleration.rs:253-267); the far one is the near one's exit
                const uint32_t axis = (n1.z >> RRT_NODE_AXIS_SHIFT) & 3u;
                const bool neg_axis = (oct >> axis) & 1u;
                cur = neg_axis ? (n1.z & (kRefLeafBit | kRefIndexMask)) : n1.w;
            } else {
                cur = exit_ref;
            }
        }
    }
}

DEV uint64_t wave_sum_u64(uint64_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void k_extend(SceneDev S, PathState P, uint64_t *wave_stats) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
    const uint32_t base = wave * kWaveSlots;
    if (base >= P.n_slots) return;
    const uint32_t count = rfl(P.qcount[wave]);
    if (count == 0) return;
    uint32_t n_inner = 0, n_leaf = 0;
    for (uint32_t i = lane; i < count; i += 64u) {
        const uint32_t slot = P.queue[base + i];
        const float4 A = P.rayA[slot];
        const float4 B = P.rayB[slot];
        float tmax = __FLT_MAX__;
        uint32_t best = kPrimNone;
        traverse<MOVING, COUNT>(S, mk(A.x, A.y, A.z), mk(B.x, B.y, B.z), A.w, 0.001f, tmax, best, n_inner, n_leaf);  // lib.rs:57
        P.hit[slot] = make_uint2(f2u(tmax), best);
    }
    uint64_t *st = wave_stats + (size_t)wave * kWaveStatWords;
    if (COUNT) {
        const uint64_t si = wave_sum_u64(n_inner), sl = wave_sum_u64(n_leaf);
        if (lane == 0) { st[ST_INNER] += si; st[ST_LEAF] += sl; }
    }
    if (lane == 0) st[ST_RAYS] += count;
}

// The same walk over the compact node copy held in LDS (16-bit references, scene_layout.h).  `best` receives the
// node index of the closest leaf; the caller turns it into a sphere index with one more LDS read.
template <bool MOVING, bool COUNT>
DEV void traverse_lds(const uint4 *lds4, uint32_t root16, const SceneDev &S, V3 o, V3 d, float time, float tmin, float &tmax,
                      uint32_t &best_node, uint32_t &n_inner, uint32_t &n_leaf) {
    constexpr uint32_t W = MOVING ? kSmallWordsMoving : kSmallWordsStatic;
    const uint16_t *lds16 = reinterpret_cast<const uint16_t *>(lds4);
    const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds4);
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool ngx = inv.x < 0.0f, ngy = inv.y < 0.0f, ngz = inv.z < 0.0f;
    const uint32_t oct = (ngx ? 1u : 0u) | (ngy ? 2u : 0u) | (ngz ? 4u : 0u);
    const float a = dot3(d, d);
    float f_ray = 0.0f;
    if (MOVING) f_ray = (time - S.u_time0) / S.u_dt;  // the compact copy exists only when all moving spheres share their shutter interval
    uint32_t cur = root16;
    while (cur != kRef16End) {
        const uint32_t idx = cur & (kRef16LeafBit - 1u);
        const uint4 n0 = lds4[idx * (W / 4u)];
        if (cur & kRef16LeafBit) {
            if (COUNT) n_leaf++;
            V3 c = mk(u2f(n0.x), u2f(n0.y), u2f(n0.z));
            const float rr = u2f(n0.w);
            if (MOVING) {
                if (lds32[idx * W + 5u] & 1u) {
                    const uint4 m = lds4[idx * (W / 4u) + 3u];
                    c = add(c, scale(f_ray, mk(u2f(m.x), u2f(m.y), u2f(m.z))));
                }
            }
            const V3 oc = sub(o, c);
            const float half_b = dot3(oc, d);
            const float cc = dot3(oc, oc) - rr;
            const float disc = (half_b * half_b) - (a * cc);
            if (!(disc < 0.0f)) {
                const float sq = sqrtf(disc);
                float root = (-half_b - sq) / a;
                bool outside = root < tmin || tmax < root;
                if (outside) {
                    root = (-half_b + sq) / a;
                    outside = root < tmin || tmax < root;
                }
                if (!outside) { tmax = root; best_node = idx; }
            }
            cur = lds16[idx * (W * 2u) + 16u + oct];
        } else {
            if (COUNT) n_inner++;
            const uint4 n1 = lds4[idx * (W / 4u) + 1u];
            const float lox = u2f(n0.x), loy = u2f(n0.y), loz = u2f(n0.z), hix = u2f(n0.w), hiy = u2f(n1.x), hiz = u2f(n1.y);
            const float tnx = ((ngx ? hix : lox) - o.x) * inv.x;
            const float tny = ((ngy ? hiy : loy) - o.y) * inv.y;
            const float tnz = ((ngz ? hiz : loz) - o.z) * inv.z;
            const float tfx = (((ngx ? lox : hix) - o.x) * inv.x) * RRT_FAR_SCALE;
            const float tfy = (((ngy ? loy : hiy) - o.y) * inv.y) * RRT_FAR_SCALE;
            const float tfz = (((ngz ? loz : hiz) - o.z) * inv.z) * RRT_FAR_SCALE;
            const float t0 = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), tmin);
            const float t1 = fminf(fminf(fminf(tfx, tfy), tfz), tmax);
            if (t0 <= t1) {
                const bool neg_axis = (oct >> n1.w) & 1u;
                cur = neg_axis ? (n1.z >> 16) : (n1.z & 0xFFFFu);
            } else {
                cur = lds16[idx * (W * 2u) + 16u + oct];
            }
        }
    }
}

constexpr uint32_t kLdsBlockThreads = 1024;  // 16 waves share one LDS copy of the scene

template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kLdsBlockThreads) void k_extend_lds(SceneDev S, PathState P, uint64_t *wave_stats) {
    extern __shared__ uint4 lds4[];
    constexpr uint32_t W = MOVING ? kSmallWordsMoving : kSmallWordsStatic;
    const uint32_t n_vec = S.n_nodes * (W / 4u);
    for (uint32_t i = threadIdx.x; i < n_vec; i += kLdsBlockThreads) lds4[i] = S.small_nodes[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kLdsBlockThreads + threadIdx.x) >> 6;
    const uint32_t base = wave * kWaveSlots;
    if (base >= P.n_slots) return;
    const uint32_t count = rfl(P.qcount[wave]);
    if (count == 0) return;
    const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds4);
    uint32_t n_inner = 0, n_leaf = 0;
    for (uint32_t i = lane; i < count; i += 64u) {
        const uint32_t slot = P.queue[base + i];
        const float4 A = P.rayA[slot];
        const float4 B = P.rayB[slot];
        float tmax = __FLT_MAX__;
        uint32_t best_node = kPrimNone;
        traverse_lds<MOVING, COUNT>(lds4, S.small_root, S, mk(A.x, A.y, A.z), mk(B.x, B.y, B.z), A.w, 0.001f, tmax, best_node, n_inner,
                                    n_leaf);
        const uint32_t best = best_node == kPrimNone ? kPrimNone : lds32[best_node * W + 4u];
        P.hit[slot] = make_uint2(f2u(tmax), best);
    }
    uint64_t *st = wave_stats + (size_t)wave * kWaveStatWords;
    if (COUNT) {
        const uint64_t si = wave_sum_u64(n_inner), sl = wave_sum_u64(n_leaf);
        if (lane == 0) { st[ST_INNER] += si; st[ST_LEAF] += sl; }
    }
    if (lane == 0) st[ST_RAYS] += count;
}

// (1 - cos)^5 of materials.rs:68, as a double-precision product rounded once to f32
DEV float pow5(float x) { const double d = (double)x, d2 = d * d, d4 = d2 * d2; return (float)(d4 * d); }
DEV float reflectance(float cosine, float ref_idx) {  // materials.rs:65-69
    const float q = (1.0f - ref_idx) / (1.0f + ref_idx);
    const float r0 = q * q;
    return r0 + (1.0f - r0) * pow5(1.0f - cosine);
}

// Claims path ids for the lanes in `need` from this wave's cursor shard and starts those paths.
// Path ids are dealt in groups of 64 consecutive ids, group g belonging to shard g % n_shards.
DEV void regenerate(const PathState &P, const BatchDev &Bt, const CameraDev &C, uint32_t lane, uint32_t shard, bool need,
                    uint32_t slot, bool &alive, uint32_t &n_started) {
    const uint64_t nm = __ballot(need);
    if (nm == 0ull) return;
    const uint32_t cnt = (uint32_t)__popcll(nm);
    const uint32_t leader = (uint32_t)__ffsll((long long)nm) - 1u;
    uint32_t base_n = 0;
    if (lane == leader) base_n = atomicAdd(&Bt.cursors[shard * kCursorStride], cnt);
    base_n = __shfl(base_n, (int)leader, 64);
    bool started = false;
    if (need) {
        const uint32_t n = base_n + (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
        const uint64_t p64 = ((uint64_t)(n >> 6) * Bt.n_shards + shard) * 64ull + (n & 63u);
        if (p64 < (uint64_t)Bt.n_paths) {
            float4 A, B;
            Rng g;
            generate_path(Bt, C, (uint32_t)p64, A, B, g);
            P.rayA[slot] = A;
            P.rayB[slot] = B;
            P.rng[slot] = make_uint4(g.s0, g.s1, g.s2, g.s3);
            P.depth[slot] = 0u;
            alive = true;
            started = true;
        }
    }
    n_started += (uint32_t)__popcll(__ballot(started));
}

__global__ __launch_bounds__(kBlockThreads) void k_raygen(PathState P, BatchDev Bt, CameraDev C, uint64_t *wave_stats) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
    const uint32_t base = wave * kWaveSlots;
    if (base >= P.n_slots) return;
    const uint32_t shard = wave % Bt.n_shards;
    uint32_t out_count = 0, n_started = 0;
    for (uint32_t t0 = 0; t0 < kWaveSlots; t0 += 64u) {
        const uint32_t slot = base + t0 + lane;
        bool alive = false;
        regenerate(P, Bt, C, lane, shard, true, slot, alive, n_started);
        const uint64_t am = __ballot(alive);
        if (alive) P.queue[base + out_count + (uint32_t)__popcll(am & ((1ull << lane) - 1ull))] = slot;
        out_count += (uint32_t)__popcll(am);
    }
    if (lane == 0) {
        P.qcount[wave] = out_count;
        if (out_count) *Bt.alive_flag = 1u;
        wave_stats[(size_t)wave * kWaveStatWords + ST_PATHS] += n_started;
    }
}

template <bool MOVING, bool COUNT>
__global__ __launch_bounds__(kBlockThreads) void k_shade(SceneDev S, PathState P, BatchDev Bt, CameraDev C, uint64_t *wave_stats) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * kBlockThreads + threadIdx.x) >> 6;
    const uint32_t base = wave * kWaveSlots;
    if (base >= P.n_slots) return;
    const uint32_t count = rfl(P.qcount[wave]);
    if (count == 0) return;
    const uint32_t shard = wave % Bt.n_shards;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t out_count = 0, n_started = 0;
    uint32_t c_hl = 0, c_hm = 0, c_hd = 0, c_sky = 0, c_cut = 0, c_abs = 0;

    for (uint32_t t0 = 0; t0 < count; t0 += 64u) {
        const uint32_t i = t0 + lane;
        const bool active = i < count;
        uint32_t slot = 0;
        bool alive = false, need = false;
        bool e_hl = false, e_hm = false, e_hd = false, e_sky = false, e_cut = false, e_abs = false;
        if (active) {
            slot = P.queue[base + i];
            const float4 A = P.rayA[slot];
            const float4 B = P.rayB[slot];
            const uint2 h = P.hit[slot];
            const uint32_t depth = P.depth[slot];
            const uint32_t pid = f2u(B.w);
            const V3 o = mk(A.x, A.y, A.z), d = mk(B.x, B.y, B.z);
            const float time = A.w;
            float rx = 0.0f, ry = 0.0f, rz = 0.0f;  // radiance of a finished path
            if (h.y == kPrimNone) {
                // sky (lib.rs:66-70), then the product of attenuations innermost first (lib.rs:62)
                e_sky = true;
                const float r = 1.0f / sqrtf(dot3(d, d));
                const float uy = d.y * r;
                const float t = 0.5f * (uy + 1.0f);
                const float w = 1.0f - t;
                rx = w + t * 0.5f;  // white.lerp(sky_blue, t) = white*(1-t) + sky_blue*t
                ry = w + t * 0.7f;
                rz = w + t * 1.0f;
                for (uint32_t j = depth; j-- > 0u;) {
                    const float4 att = S.prim_att[P.trail[(size_t)j * P.n_slots + slot]];
                    rx = att.x * rx; ry = att.y * ry; rz = att.z * rz;
                }
                need = true;
            } else if (depth >= Bt.max_depth) {  // lib.rs:58-60
                e_cut = true;
                need = true;
            } else {
                const uint32_t prim = h.y;
                const float t = u2f(h.x);
                const float4 sp = S.sph[prim];
                const uint32_t info = S.sph_info[prim];
                V3 c = mk(sp.x, sp.y, sp.z);
                if (MOVING && (info & kInfoMovingBit)) {  // sphere.rs:61-64
                    const float4 m = S.motion[prim];
                    const float f = (S.flags & kSceneUniformTimes) ? (time - S.u_time0) / S.u_dt : (time - m.w) / S.motion_dt[prim];
                    c = add(c, scale(f, mk(m.x, m.y, m.z)));
                }
                // Hit record (sphere.rs:86-100)
                const V3 p = add(o, scale(t, d));
                const V3 pc = sub(p, c);
                const V3 on = mk(pc.x / sp.w, pc.y / sp.w, pc.z / sp.w);
                const bool front = dot3(d, on) < 0.0f;
                const V3 n = front ? on : neg(on);
                const uint32_t kind = (info >> kInfoKindShift) & 3u;
                const float4 mt = S.mat[info & kInfoMatMask];
                Rng g;
                { const uint4 r = P.rng[slot]; g.s0 = r.x; g.s1 = r.y; g.s2 = r.z; g.s3 = r.w; }
                V3 dir;
                bool scattered = true;
                if (kind == RRT_MAT_LAMBERTIAN) {  // materials.rs:19-35
                    e_hl = true;
                    const V3 unit = normalized(in_unit_sphere(g));
                    dir = add(n, unit);
                    if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = n;
                } else if (kind == RRT_MAT_METAL) {  // materials.rs:44-61
                    e_hm = true;
                    const V3 refl = reflected(normalized(d), n);
                    const V3 s = in_unit_sphere(g);
                    dir = add(refl, scale(mt.w, s));
                    scattered = dot3(dir, n) > 0.0f;
                } else {  // materials.rs:75-104
                    e_hd = true;
                    const float ratio = front ? 1.0f / mt.w : mt.w;
                    const V3 ud = normalized(d);
                    const float cos_theta = fminf(dot3(neg(ud), n), 1.0f);
                    const float sin_theta = sqrtf(1.0f - (cos_theta * cos_theta));
                    bool reflect = ratio * sin_theta > 1.0f;
                    if (!reflect) reflect = reflectance(cos_theta, ratio) > gen_f32(g);  // the draw happens only here (materials.rs:89)
                    if (reflect) {
                        dir = reflected(ud, n);
                    } else {  // ultraviolet refracted()
                        const float ndi = dot3(n, ud);
                        const float k = 1.0f - ratio * ratio * (1.0f - ndi * ndi);
                        if (k < 0.0f) dir = mk(0.0f, 0.0f, 0.0f);
                        else dir = sub(scale(ratio, ud), scale(ratio * ndi + sqrtf(k), n));
                    }
                }
                if (scattered) {
                    P.rayA[slot] = make_float4(p.x, p.y, p.z, time);
                    P.rayB[slot] = make_float4(dir.x, dir.y, dir.z, B.w);
                    P.rng[slot] = make_uint4(g.s0, g.s1, g.s2, g.s3);
                    P.depth[slot] = depth + 1u;
                    P.trail[(size_t)depth * P.n_slots + slot] = prim;
                    alive = true;
                } else {  // lib.rs:64
                    e_abs = true;
                    need = true;
                }
            }
            if (need) {
                Bt.rad[pid] = rx;
                Bt.rad[(size_t)Bt.rad_plane + pid] = ry;
                Bt.rad[2 * (size_t)Bt.rad_plane + pid] = rz;
            }
        }
        if (COUNT) {
            c_hl += (uint32_t)__popcll(__ballot(e_hl)); c_hm += (uint32_t)__popcll(__ballot(e_hm));
            c_hd += (uint32_t)__popcll(__ballot(e_hd)); c_sky += (uint32_t)__popcll(__ballot(e_sky));
            c_cut += (uint32_t)__popcll(__ballot(e_cut)); c_abs += (uint32_t)__popcll(__ballot(e_abs));
        }
        // a finished path's slot takes the next path of the batch, if any is left
        regenerate(P, Bt, C, lane, shard, need, slot, alive, n_started);
        // compaction of the wave's live-slot list, in place: writes never pass the entries still to be read
        const uint64_t am = __ballot(alive);
        if (alive) P.queue[base + out_count + (uint32_t)__popcll(am & lt_mask)] = slot;
        out_count += (uint32_t)__popcll(am);
    }
    if (lane == 0) {
        P.qcount[wave] = out_count;
        if (out_count) *Bt.alive_flag = 1u;
        uint64_t *st = wave_stats + (size_t)wave * kWaveStatWords;
        st[ST_PATHS] += n_started;
        if (COUNT) {
            st[ST_HL] += c_hl; st[ST_HM] += c_hm; st[ST_HD] += c_hd; st[ST_SKY] += c_sky;
            st[ST_DEPTHCUT] += c_cut; st[ST_ABSORBED] += c_abs;
        }
    }
}

__global__ void k_accumulate(float *accum, uint32_t npix, const float *rad, uint32_t rad_plane, uint32_t n_samples, int first) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= npix) return;
    for (uint32_t c = 0; c < 3u; c++) {
        float acc = first ? 0.0f : accum[(size_t)c * npix + l];
        const float *r = rad + (size_t)c * rad_plane + l;
        for (uint32_t s = 0; s < n_samples; s++) acc += r[(size_t)s * npix];  // c += trace(..) (lib.rs:102)
        accum[(size_t)c * npix + l] = acc;
    }
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

template <bool MOVING>
__global__ void k_intersect(SceneDev S, uint32_t n, const float *org, const float *dir, const float *time, float t_min, float t_max,
                            float *out_t, uint32_t *out_obj) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float tmax = t_max;
    uint32_t best = kPrimNone, ni = 0, nl = 0;
    traverse<MOVING, false>(S, mk(org[i * 3u], org[i * 3u + 1u], org[i * 3u + 2u]), mk(dir[i * 3u], dir[i * 3u + 1u], dir[i * 3u + 2u]),
                            time ? time[i] : 0.0f, t_min, tmax, best, ni, nl);
    out_t[i] = tmax;
    out_obj[i] = best;
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

inline uint32_t wave_grid(const PathState &P) { return (P.n_slots / kWaveSlots + 3u) / 4u; }

}  // namespace

void launch_raygen(hipStream_t s, const PathState &P, const BatchDev &B, const CameraDev &C, uint64_t *wave_stats) {
    hipLaunchKernelGGL(k_raygen, dim3(wave_grid(P)), dim3(kBlockThreads), 0, s, P, B, C, wave_stats);
}

void launch_extend(hipStream_t s, const SceneDev &S, const PathState &P, bool moving, bool count, uint64_t *wave_stats) {
    if (S.small_nodes) {  // scene fits in LDS
        const uint32_t waves_per_block = kLdsBlockThreads / 64u;
        const dim3 g((P.n_slots / kWaveSlots + waves_per_block - 1u) / waves_per_block), b(kLdsBlockThreads);
        const size_t lds = (size_t)S.n_nodes * S.small_words * 4u;
        if (moving) {
            if (count) hipLaunchKernelGGL((k_extend_lds<true, true>), g, b, lds, s, S, P, wave_stats);
            else hipLaunchKernelGGL((k_extend_lds<true, false>), g, b, lds, s, S, P, wave_stats);
        } else {
            if (count) hipLaunchKernelGGL((k_extend_lds<false, true>), g, b, lds, s, S, P, wave_stats);
            else hipLaunchKernelGGL((k_extend_lds<false, false>), g, b, lds, s, S, P, wave_stats);
        }
        return;
    }
    const dim3 g(wave_grid(P)), b(kBlockThreads);
    if (moving) {
        if (count) hipLaunchKernelGGL((k_extend<true, true>), g, b, 0, s, S, P, wave_stats);
        else hipLaunchKernelGGL((k_extend<true, false>), g, b, 0, s, S, P, wave_stats);
    } else {
        if (count) hipLaunchKernelGGL((k_extend<false, true>), g, b, 0, s, S, P, wave_stats);
        else hipLaunchKernelGGL((k_extend<false, false>), g, b, 0, s, S, P, wave_stats);
    }
}

void launch_shade(hipStream_t s, const SceneDev &S, const PathState &P, const BatchDev &B, const CameraDev &C, bool moving,
                  bool count, uint64_t *wave_stats) {
    const dim3 g(wave_grid(P)), b(kBlockThreads);
    if (moving) {
        if (count) hipLaunchKernelGGL((k_shade<true, true>), g, b, 0, s, S, P, B, C, wave_stats);
        else hipLaunchKernelGGL((k_shade<true, false>), g, b, 0, s, S, P, B, C, wave_stats);
    } else {
        if (count) hipLaunchKernelGGL((k_shade<false, true>), g, b, 0, s, S, P, B, C, wave_stats);
        else hipLaunchKernelGGL((k_shade<false, false>), g, b, 0, s, S, P, B, C, wave_stats);
    }
}

void launch_accumulate(hipStream_t s, float *accum, uint32_t npix, const float *rad, uint32_t rad_plane, uint32_t n_samples,
                       bool first_batch) {
    hipLaunchKernelGGL(k_accumulate, dim3((npix + 255u) / 256u), dim3(256), 0, s, accum, npix, rad, rad_plane, n_samples,
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

void launch_intersect(hipStream_t s, const SceneDev &S, bool moving, uint32_t n, const float *org, const float *dir,
                      const float *time, float t_min, float t_max, float *out_t, uint32_t *out_obj) {
    const dim3 g((n + 255u) / 256u), b(256);
    if (moving) hipLaunchKernelGGL((k_intersect<true>), g, b, 0, s, S, n, org, dir, time, t_min, t_max, out_t, out_obj);
    else hipLaunchKernelGGL((k_intersect<false>), g, b, 0, s, S, n, org, dir, time, t_min, t_max, out_t, out_obj);
}

void launch_reduce_stats(hipStream_t s, const uint64_t *wave_stats, uint32_t n_waves, uint64_t *out16) {
    hipLaunchKernelGGL(k_reduce_stats, dim3(kWaveStatWords), dim3(256), 0, s, wave_stats, n_waves, out16);
}

}  // namespace rrt_amd
