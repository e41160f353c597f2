/*
 * rrt_oracle.c — CPU restatement of the jfroy/rrt render path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under rrt_amd/ links or calls it.
 *
 * PARITY UNPINNED: the reference (Rust nightly) cannot be compiled in this environment and ships
 * no tests, golden images or known-answer vectors, so nothing ties this restatement to outputs of
 * the reference itself.  What pins it: the published xoshiro128+/SplitMix64 known answers, the
 * brute-force closest_hit semantics (lib.rs:32-47) against the BVH traversal, and analytic checks.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 * Third-party arithmetic that is not in the reference tree (pinned in Cargo.lock) is restated from
 * the published algorithms: rand_xoshiro 0.6.0 (Xoshiro128Plus, SplitMix64 seeding, jump),
 * rand 0.8.5 (Standard f32, Uniform<f32>::new_inclusive / sample), ultraviolet 0.9.2 (Vec4 dot /
 * mag / normalized / reflected / refracted / lerp with their fused-multiply-add placement),
 * wide 0.7.28 (f32x4 blend / cmp_lt / move_mask in the slab test).
 *
 * Floating point discipline: compile with -ffp-contract=off and no fast-math; every fused
 * operation is an explicit fmaf().  Rust never contracts a*b+c on its own.
 *
 * Deviation from libm, on purpose and switchable: reflectance() raises (1-cos) to the 5th power.
 * Rust calls the platform powf there.  pow5_mode 0 (default, what the GPU computes) evaluates it
 * as a double-precision product chain rounded once to f32; pow5_mode 1 calls powf like the
 * reference.  tests/ quantify the difference.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Basic types (types.rs:6-37)                                                                  */
/* ------------------------------------------------------------------------------------------ */

typedef struct { float x, y, z, w; } v4;

static inline v4 v4_new(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v4 v4_splat(float s) { return v4_new(s, s, s, s); }
static inline v4 v4_add(v4 a, v4 b) { return v4_new(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline v4 v4_sub(v4 a, v4 b) { return v4_new(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline v4 v4_mul(v4 a, v4 b) { return v4_new(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline v4 v4_scale(float s, v4 a) { return v4_new(s * a.x, s * a.y, s * a.z, s * a.w); }
static inline v4 v4_divs(v4 a, float s) { return v4_new(a.x / s, a.y / s, a.z / s, a.w / s); }
static inline v4 v4_neg(v4 a) { return v4_new(-a.x, -a.y, -a.z, -a.w); }
static inline float v4_get(v4 a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }

/* ultraviolet 0.9.2 Vec4::dot: x.mul_add(ox, y.mul_add(oy, z.mul_add(oz, w*ow))) */
static inline float v4_dot(v4 a, v4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
static inline float v4_mag(v4 a) { return sqrtf(v4_dot(a, a)); }
/* ultraviolet normalize: r = 1/mag, then multiply each lane */
static inline v4 v4_normalized(v4 a) {
    float r = 1.0f / v4_mag(a);
    return v4_new(a.x * r, a.y * r, a.z * r, a.w * r);
}
/* ultraviolet reflected: self - (2 * self.dot(n)) * n */
static inline v4 v4_reflected(v4 i, v4 n) {
    float k = 2.0f * v4_dot(i, n);
    return v4_sub(i, v4_scale(k, n));
}
/* ultraviolet refracted (GLSL form) */
static inline v4 v4_refracted(v4 i, v4 n, float eta) {
    float ndi = v4_dot(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return v4_splat(0.0f);
    return v4_sub(v4_scale(eta, i), v4_scale(eta * ndi + sqrtf(k), n));
}

/* Rust f32::max / f32::min: the non-NaN operand wins */
static inline float rs_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline float rs_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }

typedef struct { v4 origin, direction; float time; } ray_t; /* types.rs:9-24 */

static inline v4 ray_point_at(const ray_t *r, float t) { /* types.rs:17-19 */
    return v4_add(r->origin, v4_scale(t, r->direction));
}

/* ------------------------------------------------------------------------------------------ */
/* RNG (rng.rs:14; rand_xoshiro 0.6.0; rand 0.8.5)                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct { uint32_t s[4]; } xrng;

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

static inline uint32_t rng_next_u32(xrng *g) { /* xoshiro128+ */
    uint32_t r = g->s[0] + g->s[3];
    uint32_t t = g->s[1] << 9;
    g->s[2] ^= g->s[0];
    g->s[3] ^= g->s[1];
    g->s[1] ^= g->s[2];
    g->s[0] ^= g->s[3];
    g->s[2] ^= t;
    g->s[3] = rotl32(g->s[3], 11);
    return r;
}

static inline uint64_t splitmix64_next(uint64_t *st) {
    *st += 0x9E3779B97F4A7C15ull;
    uint64_t z = *st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* SeedableRng::seed_from_u64 for Xoshiro128Plus: two SplitMix64 outputs, little-endian fill */
static void rng_seed_from_u64(xrng *g, uint64_t seed) {
    uint64_t st = seed;
    uint64_t z1 = splitmix64_next(&st), z2 = splitmix64_next(&st);
    g->s[0] = (uint32_t)z1; g->s[1] = (uint32_t)(z1 >> 32);
    g->s[2] = (uint32_t)z2; g->s[3] = (uint32_t)(z2 >> 32);
    if ((g->s[0] | g->s[1] | g->s[2] | g->s[3]) == 0) rng_seed_from_u64(g, 0);
}

/* Xoshiro128Plus::jump — 2^64 calls to next_u32 (threadpool.rs:18-20) */
static void rng_jump(xrng *g) {
    static const uint32_t J[4] = {0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu};
    uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 32; b++) {
            if (J[i] & (1u << b)) { a0 ^= g->s[0]; a1 ^= g->s[1]; a2 ^= g->s[2]; a3 ^= g->s[3]; }
            rng_next_u32(g);
        }
    g->s[0] = a0; g->s[1] = a1; g->s[2] = a2; g->s[3] = a3;
}

/* rand 0.8.5 Standard for f32: 24 high bits scaled by 2^-24, in [0,1) */
static inline float rng_gen_f32(xrng *g) { return (float)(rng_next_u32(g) >> 8) * 0x1p-24f; }

/* rand 0.8.5 UniformFloat<f32>::new_inclusive / sample */
typedef struct { float low, scale; } uniform_f32;

static inline float f32_decr(float x) { uint32_t b; memcpy(&b, &x, 4); b -= 1; memcpy(&x, &b, 4); return x; }

static uniform_f32 uniform_new_inclusive(float low, float high) {
    const float max_rand = 1.0f - 0x1p-23f;
    float scale = (high - low) / max_rand;
    while (scale * max_rand + low > high) scale = f32_decr(scale);
    uniform_f32 u = {low, scale};
    return u;
}
static inline float uniform_sample(uniform_f32 u, xrng *g) {
    float v01 = (float)(rng_next_u32(g) >> 9) * 0x1p-23f; /* == value1_2 - 1.0, exact */
    return v01 * u.scale + u.low;
}

/* rng.rs:32-40 */
static v4 gen_in_unit_sphere(xrng *g) {
    uniform_f32 d = uniform_new_inclusive(-1.0f, 1.0f);
    for (;;) {
        float x = uniform_sample(d, g), y = uniform_sample(d, g), z = uniform_sample(d, g);
        v4 v = v4_new(x, y, z, 0.0f);
        if (v4_dot(v, v) < 1.0f) return v;
    }
}
/* rng.rs:42-50 */
static v4 gen_in_unit_disc(xrng *g) {
    uniform_f32 d = uniform_new_inclusive(-1.0f, 1.0f);
    for (;;) {
        float x = uniform_sample(d, g), y = uniform_sample(d, g);
        v4 v = v4_new(x, y, 0.0f, 0.0f);
        if (v4_dot(v, v) < 1.0f) return v;
    }
}
/* rng.rs:52-55 */
static v4 gen_unit(xrng *g) { return v4_normalized(gen_in_unit_sphere(g)); }

/* ------------------------------------------------------------------------------------------ */
/* fp.rs:12-19                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static float gamma_eb(int n) {
    const float eps = FLT_EPSILON * 0.5f;
    float fn = (float)n;
    return (fn * eps) / (1.0f - fn * eps);
}

/* ------------------------------------------------------------------------------------------ */
/* Materials, spheres, AABB, BVH                                                                */
/* ------------------------------------------------------------------------------------------ */
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2 };
typedef struct { int kind; v4 albedo; float fuzz; float ref_idx; } material_t;

typedef struct { v4 minimum, maximum; } aabb_t; /* acceleration.rs:10-13 */

typedef struct { /* sphere.rs:9-17 */
    v4 center0, center1;
    float time0, time1, radius;
    material_t material;
    aabb_t aabb;
} sphere_t;

static aabb_t aabb_default(void) { /* acceleration.rs:87-94 */
    aabb_t a = {v4_splat(FLT_MAX), v4_splat(-FLT_MAX)};
    return a;
}
static v4 v4_min_c(v4 a, v4 b) { return v4_new(rs_min(a.x, b.x), rs_min(a.y, b.y), rs_min(a.z, b.z), rs_min(a.w, b.w)); }
static v4 v4_max_c(v4 a, v4 b) { return v4_new(rs_max(a.x, b.x), rs_max(a.y, b.y), rs_max(a.z, b.z), rs_max(a.w, b.w)); }
static aabb_t aabb_union(aabb_t a, aabb_t b) { /* acceleration.rs:16-20 */
    aabb_t r = {v4_min_c(a.minimum, b.minimum), v4_max_c(a.maximum, b.maximum)};
    return r;
}
static aabb_t aabb_union_p(aabb_t a, v4 p) { /* acceleration.rs:22-26 */
    aabb_t r = {v4_min_c(a.minimum, p), v4_max_c(a.maximum, p)};
    return r;
}
static v4 aabb_centroid(aabb_t a) { /* acceleration.rs:28-30 */
    return v4_add(v4_scale(0.5f, a.minimum), v4_scale(0.5f, a.maximum));
}
static int aabb_max_extent(aabb_t a) { /* acceleration.rs:36-45 */
    v4 d = v4_sub(a.maximum, a.minimum);
    if (d.x > d.y && d.x > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}

/* sphere.rs:25-35 — literal: a negative radius yields an inverted box */
static aabb_t sphere_aabb(v4 c0, v4 c1, float radius) {
    aabb_t b0 = {v4_sub(c0, v4_splat(radius)), v4_add(c0, v4_splat(radius))};
    aabb_t b1 = {v4_sub(c1, v4_splat(radius)), v4_add(c1, v4_splat(radius))};
    return aabb_union(b0, b1);
}
static sphere_t sphere_new(v4 c0, v4 c1, float t0, float t1, float radius, material_t m) { /* sphere.rs:38-55 */
    sphere_t s;
    s.center0 = c0; s.center1 = c1; s.time0 = t0; s.time1 = t1; s.radius = radius; s.material = m;
    s.aabb = sphere_aabb(c0, c1, radius);
    return s;
}
static sphere_t sphere_stationary(v4 c, float radius, material_t m) { /* sphere.rs:57-59 */
    return sphere_new(c, c, 0.0f, 1.0f, radius, m);
}
static inline v4 sphere_center_at(const sphere_t *s, float t) { /* sphere.rs:61-64 */
    float f = (t - s->time0) / (s->time1 - s->time0);
    return v4_add(s->center0, v4_scale(f, v4_sub(s->center1, s->center0)));
}

typedef struct { /* hittable.rs:10-16 */
    v4 p, normal;
    float t;
    const material_t *material;
    int front_face;
    uint32_t obj; /* index of the sphere, for tests */
} hit_t;

/* sphere.rs:68-101 */
static int sphere_hit(const sphere_t *s, const ray_t *r, float t_min, float t_max, hit_t *out) {
    v4 center = sphere_center_at(s, r->time);
    v4 oc = v4_sub(r->origin, center);
    float a = v4_dot(r->direction, r->direction);
    float half_b = v4_dot(oc, r->direction);
    float c = v4_dot(oc, oc) - (s->radius * s->radius);
    float disc = (half_b * half_b) - (a * c);
    if (disc < 0.0f) return 0;
    float sq = sqrtf(disc);
    float root = (-half_b - sq) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sq) / a;
        if (root < t_min || t_max < root) return 0;
    }
    v4 p = ray_point_at(r, root);
    v4 outward = v4_divs(v4_sub(p, center), s->radius);
    int front = v4_dot(r->direction, outward) < 0.0f;
    out->p = p;
    out->normal = front ? outward : v4_neg(outward);
    out->t = root;
    out->material = &s->material;
    out->front_face = front;
    return 1;
}

enum { NODE_LEAF = 0, NODE_INNER = 1 };
typedef struct { /* acceleration.rs:106-116 */
    int kind;
    uint32_t obj_i;
    aabb_t aabb;
    int axis;
    int64_t left_i, right_i; /* -1 == None */
} bvh_node_t;

typedef struct {
    sphere_t *spheres; uint32_t n_spheres;
    bvh_node_t *nodes; uint32_t n_nodes, cap_nodes;
} scene_t;

typedef struct { uint64_t rays, inner, leaf, h_l, h_m, h_d, sky, paths, depth_cut, absorbed; } ctrs_t;

typedef struct { int max_depth; int pow5_mode; } opts_t;

/* acceleration.rs:60-84.  dirneg[i] = (inv_dir[i] < 0) as in the cmp_lt mask of :230. */
static inline int aabb_intersect(const aabb_t *b, const ray_t *r, const float inv_d[3], const int dirneg[3],
                                 float t0, float t1, float far_scale) {
    float tn[3], tf[3];
    for (int i = 0; i < 3; i++) {
        float lo = v4_get(b->minimum, i), hi = v4_get(b->maximum, i), o = v4_get(r->origin, i);
        float nearp = dirneg[i] ? hi : lo;
        float farp = dirneg[i] ? lo : hi;
        tn[i] = (nearp - o) * inv_d[i];
        tf[i] = ((farp - o) * inv_d[i]) * far_scale;
    }
    t0 = rs_max(rs_max(rs_max(tn[0], tn[1]), tn[2]), t0);
    t1 = rs_min(rs_min(rs_min(tf[0], tf[1]), tf[2]), t1);
    return t0 <= t1;
}

static int g_stack_overflow = 0;

/* acceleration.rs:221-272 */
static int bvh_hit(const scene_t *sc, const ray_t *r, float t_min, float t_max, hit_t *out, ctrs_t *C) {
    float inv_d[3] = {1.0f / r->direction.x, 1.0f / r->direction.y, 1.0f / r->direction.z};
    int dirneg[3] = {inv_d[0] < 0.0f, inv_d[1] < 0.0f, inv_d[2] < 0.0f};
    const float far_scale = 1.0f + 2.0f * gamma_eb(3);
    uint32_t stack[64];
    int sp = 0, found = 0;
    if (sc->n_nodes == 0) return 0; /* the reference would panic indexing nodes[0] */
    stack[sp++] = 0;
    while (sp > 0) {
        const bvh_node_t *n = &sc->nodes[stack[--sp]];
        if (n->kind == NODE_LEAF) {
            hit_t h;
            if (C) C->leaf++;
            if (sphere_hit(&sc->spheres[n->obj_i], r, t_min, t_max, &h)) {
                t_max = h.t;
                h.obj = n->obj_i;
                *out = h;
                found = 1;
            }
        } else {
            if (C) C->inner++;
            if (!aabb_intersect(&n->aabb, r, inv_d, dirneg, t_min, t_max, far_scale)) continue;
            if (sp + 2 > 64) { g_stack_overflow = 1; return found; } /* ArrayVec<_,64>::push would panic */
            if (dirneg[n->axis]) {
                if (n->left_i >= 0) stack[sp++] = (uint32_t)n->left_i;
                if (n->right_i >= 0) stack[sp++] = (uint32_t)n->right_i;
            } else {
                if (n->right_i >= 0) stack[sp++] = (uint32_t)n->right_i;
                if (n->left_i >= 0) stack[sp++] = (uint32_t)n->left_i;
            }
        }
    }
    return found;
}

/* lib.rs:32-47 (dead code in the reference; its semantics are the BVH cross-check) */
static int closest_hit_brute(const scene_t *sc, const ray_t *r, float t_min, float t_max, hit_t *out) {
    int found = 0;
    for (uint32_t i = 0; i < sc->n_spheres; i++) {
        hit_t h;
        if (sphere_hit(&sc->spheres[i], r, t_min, t_max, &h)) {
            t_max = h.t; h.obj = i; *out = h; found = 1;
        }
    }
    return found;
}

/* ---- BVH build (acceleration.rs:131-217) ---- */
typedef struct { uint32_t index; aabb_t aabb; v4 centroid; } obj_info_t;

static uint32_t bvh_push(scene_t *sc, bvh_node_t n) {
    if (sc->n_nodes == sc->cap_nodes) {
        sc->cap_nodes = sc->cap_nodes ? sc->cap_nodes * 2 : 64;
        sc->nodes = (bvh_node_t *)realloc(sc->nodes, (size_t)sc->cap_nodes * sizeof(bvh_node_t));
    }
    sc->nodes[sc->n_nodes] = n;
    return sc->n_nodes++;
}

/* stable merge sort of infos by centroid[axis]; incomparable (NaN) pairs compare Equal */
static void sort_infos(obj_info_t *a, obj_info_t *tmp, size_t n, int axis) {
    if (n < 2) return;
    size_t h = n / 2;
    sort_infos(a, tmp, h, axis);
    sort_infos(a + h, tmp, n - h, axis);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        float ci = v4_get(a[i].centroid, axis), cj = v4_get(a[j].centroid, axis);
        if (cj < ci) tmp[k++] = a[j++]; else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(obj_info_t));
}

enum { PART_MIDDLE = 0, PART_EQUAL_COUNT = 1 };

/* Returns node index or -1.  The reference's partition (itertools::partition) and selection
 * (select_nth_unstable_by) leave the order inside each side unspecified; this restatement uses a
 * stable partition and a stable sort, which are valid instances of both. */
static int64_t bvh_build_r(scene_t *sc, obj_info_t *infos, obj_info_t *tmp, size_t n, int method) {
    if (n == 0) return -1;
    if (n == 1) {
        bvh_node_t leaf; memset(&leaf, 0, sizeof leaf);
        leaf.kind = NODE_LEAF; leaf.obj_i = infos[0].index; leaf.left_i = leaf.right_i = -1;
        return bvh_push(sc, leaf);
    }
    aabb_t box = aabb_default(), cb = aabb_default();
    for (size_t i = 0; i < n; i++) box = aabb_union(box, infos[i].aabb);
    for (size_t i = 0; i < n; i++) cb = aabb_union_p(cb, infos[i].centroid);
    int axis = aabb_max_extent(cb);
    size_t pivot = 0;
    for (;;) {
        if (method == PART_MIDDLE) {
            float mid = (v4_get(cb.minimum, axis) + v4_get(cb.maximum, axis)) / 2.0f;
            size_t nl = 0, nr = 0;
            for (size_t i = 0; i < n; i++) {
                if (v4_get(infos[i].centroid, axis) < mid) tmp[nl++] = infos[i];
            }
            nr = nl;
            for (size_t i = 0; i < n; i++) {
                if (!(v4_get(infos[i].centroid, axis) < mid)) tmp[nr++] = infos[i];
            }
            if (nl == 0 || nl == n) { method = PART_EQUAL_COUNT; continue; }
            memcpy(infos, tmp, n * sizeof(obj_info_t));
            pivot = nl;
        } else {
            pivot = n / 2;
            sort_infos(infos, tmp, n, axis);
        }
        break;
    }
    bvh_node_t inner; memset(&inner, 0, sizeof inner);
    inner.kind = NODE_INNER; inner.aabb = box; inner.axis = axis; inner.left_i = inner.right_i = -1;
    uint32_t me = bvh_push(sc, inner);
    int64_t l = bvh_build_r(sc, infos, tmp, pivot, method);
    int64_t rr = bvh_build_r(sc, infos + pivot, tmp + pivot, n - pivot, method);
    sc->nodes[me].left_i = l;
    sc->nodes[me].right_i = rr;
    return me;
}

static void scene_build_bvh(scene_t *sc, int method) { /* scene.rs:32-35, acceleration.rs:131-146 */
    size_t n = sc->n_spheres;
    obj_info_t *infos = (obj_info_t *)malloc((n + 1) * sizeof(obj_info_t));
    obj_info_t *tmp = (obj_info_t *)malloc((n + 1) * sizeof(obj_info_t));
    for (size_t i = 0; i < n; i++) {
        infos[i].index = (uint32_t)i;
        infos[i].aabb = sc->spheres[i].aabb;
        infos[i].centroid = aabb_centroid(infos[i].aabb);
    }
    free(sc->nodes); sc->nodes = NULL; sc->n_nodes = sc->cap_nodes = 0;
    bvh_build_r(sc, infos, tmp, n, method);
    free(infos); free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* Materials (materials.rs)                                                                     */
/* ------------------------------------------------------------------------------------------ */
typedef struct { ray_t r; v4 attenuation; } scattered_t; /* types.rs:26-30 */

static int near_zero(v4 v) { /* materials.rs:15-17 */
    return fabsf(v.x) < 1e-8f && fabsf(v.y) < 1e-8f && fabsf(v.z) < 1e-8f && fabsf(v.w) < 1e-8f;
}

static float pow5(float x, int mode) {
    if (mode == 1) return powf(x, 5.0f);
    double d = (double)x, d2 = d * d, d4 = d2 * d2;
    return (float)(d4 * d);
}

static float reflectance(float cosine, float ref_idx, int pow5_mode) { /* materials.rs:65-69 */
    float q = (1.0f - ref_idx) / (1.0f + ref_idx);
    float r0 = q * q; /* powf(2.) == exact square rounded once */
    return r0 + (1.0f - r0) * pow5(1.0f - cosine, pow5_mode);
}

static int material_scatter(const material_t *m, const ray_t *r, const hit_t *h, xrng *g, scattered_t *out,
                            const opts_t *O) {
    switch (m->kind) {
    case MAT_LAMBERTIAN: { /* materials.rs:19-35 */
        v4 dir = v4_add(h->normal, gen_unit(g));
        out->r.origin = h->p;
        out->r.direction = near_zero(dir) ? h->normal : dir;
        out->r.time = r->time;
        out->attenuation = m->albedo;
        return 1;
    }
    case MAT_METAL: { /* materials.rs:44-61 */
        v4 refl = v4_reflected(v4_normalized(r->direction), h->normal);
        v4 dir = v4_add(refl, v4_scale(m->fuzz, gen_in_unit_sphere(g)));
        if (v4_dot(dir, h->normal) > 0.0f) {
            out->r.origin = h->p; out->r.direction = dir; out->r.time = r->time;
            out->attenuation = m->albedo;
            return 1;
        }
        return 0;
    }
    default: { /* Dielectric, materials.rs:75-104 */
        float ratio = h->front_face ? 1.0f / m->ref_idx : m->ref_idx;
        v4 unit = v4_normalized(r->direction);
        float cos_theta = rs_min(v4_dot(v4_neg(unit), h->normal), 1.0f);
        float sin_theta = sqrtf(1.0f - (cos_theta * cos_theta));
        int cannot_refract = ratio * sin_theta > 1.0f;
        v4 dir;
        if (cannot_refract || reflectance(cos_theta, ratio, O->pow5_mode) > rng_gen_f32(g))
            dir = v4_reflected(unit, h->normal);
        else
            dir = v4_refracted(unit, h->normal, ratio);
        out->r.origin = h->p; out->r.direction = dir; out->r.time = r->time;
        out->attenuation = v4_new(1.0f, 1.0f, 1.0f, 0.0f);
        return 1;
    }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Camera (camera.rs)                                                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    v4 lower_left_corner, horizontal, vertical, origin, u, v, w;
    float lens_radius;
    uniform_f32 time_dist;
} camera_t;

typedef struct { v4 look_from, look_at, up; float fov, aspect, aperture, focus_dist, time0, time1; } camera_ci_t;

typedef struct { float x, y, z; } v3;
static v3 v3_cross(v3 a, v3 b) { /* ultraviolet Vec3::cross */
    v3 r = {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
    return r;
}
static v3 v3_normalized(v3 a) {
    float m = sqrtf(fmaf(a.x, a.x, fmaf(a.y, a.y, a.z * a.z)));
    float r = 1.0f / m;
    v3 o = {a.x * r, a.y * r, a.z * r};
    return o;
}

static camera_t camera_from(camera_ci_t ci) { /* camera.rs:31-54 */
    camera_t c;
    float theta = ci.fov * 3.14159265358979323846f / 180.0f;
    float half_height = tanf(theta / 2.0f);
    float half_width = ci.aspect * half_height;
    v4 w = v4_normalized(v4_sub(ci.look_from, ci.look_at));
    v3 up3 = {ci.up.x, ci.up.y, ci.up.z}, w3 = {w.x, w.y, w.z};
    v3 u3 = v3_normalized(v3_cross(up3, w3));
    v3 v3v = v3_cross(w3, u3);
    v4 u = v4_new(u3.x, u3.y, u3.z, 0.0f), v = v4_new(v3v.x, v3v.y, v3v.z, 0.0f);
    c.lower_left_corner = v4_sub(v4_sub(v4_sub(ci.look_from, v4_scale(half_width * ci.focus_dist, u)),
                                        v4_scale(half_height * ci.focus_dist, v)),
                                 v4_scale(ci.focus_dist, w));
    c.horizontal = v4_scale(2.0f * half_width * ci.focus_dist, u);
    c.vertical = v4_scale(2.0f * half_height * ci.focus_dist, v);
    c.origin = ci.look_from;
    c.u = u; c.v = v; c.w = w;
    c.lens_radius = ci.aperture / 2.0f;
    c.time_dist = uniform_new_inclusive(ci.time0, ci.time1);
    return c;
}

static ray_t camera_gen_ray(const camera_t *c, float s, float t, xrng *g) { /* camera.rs:56-66 */
    v4 rd = v4_scale(c->lens_radius, gen_in_unit_disc(g));
    v4 offset = v4_add(v4_scale(rd.x, c->u), v4_scale(rd.y, c->v));
    ray_t r;
    r.origin = v4_add(c->origin, offset);
    r.direction = v4_sub(v4_add(v4_add(c->lower_left_corner, v4_scale(s, c->horizontal)), v4_scale(t, c->vertical)),
                         r.origin);
    r.time = uniform_sample(c->time_dist, g);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Integrator (lib.rs:50-113)                                                                   */
/* ------------------------------------------------------------------------------------------ */
/* Debugging aid (orc_trace_path_rays): when set, every ray trace() is given is appended here, 7 floats each. */
static __thread float *g_ray_log = NULL;
static __thread uint64_t g_ray_log_cap = 0, g_ray_log_n = 0;

static v4 trace(const scene_t *sc, const ray_t *r, int depth, xrng *g, ctrs_t *C, const opts_t *O) { /* lib.rs:50-71 */
    hit_t h;
    C->rays++;
    if (g_ray_log) {
        if (g_ray_log_n < g_ray_log_cap) {
            float *o = g_ray_log + g_ray_log_n * 7;
            o[0] = r->origin.x; o[1] = r->origin.y; o[2] = r->origin.z;
            o[3] = r->direction.x; o[4] = r->direction.y; o[5] = r->direction.z; o[6] = r->time;
        }
        g_ray_log_n++;
    }
    if (bvh_hit(sc, r, 0.001f, FLT_MAX, &h, C)) {
        if (depth >= O->max_depth) { C->depth_cut++; return v4_splat(0.0f); }
        scattered_t s;
        switch (h.material->kind) { case MAT_LAMBERTIAN: C->h_l++; break; case MAT_METAL: C->h_m++; break; default: C->h_d++; }
        if (material_scatter(h.material, r, &h, g, &s, O)) {
            v4 c = trace(sc, &s.r, depth + 1, g, C, O);
            return v4_mul(s.attenuation, c);
        }
        C->absorbed++;
        return v4_splat(0.0f);
    }
    C->sky++;
    v4 white = v4_splat(1.0f), sky_blue = v4_new(0.5f, 0.7f, 1.0f, 0.0f);
    v4 unit = v4_normalized(r->direction);
    float t = 0.5f * (unit.y + 1.0f);
    /* ultraviolet Lerp: a*(1-t) + b*t */
    return v4_add(v4_scale(1.0f - t, white), v4_scale(t, sky_blue));
}

/* Rust `as u8`: saturating, NaN -> 0, truncation toward zero */
static uint8_t f32_as_u8(float x) {
    if (x != x) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

/* The per-(pixel,sample) stream used by the GPU renderer: a fresh Xoshiro128Plus seeded through
 * seed_from_u64 from a 64-bit mix of (seed, pixel index, sample index). */
static uint64_t sample_stream_seed(uint64_t seed, uint64_t idx, uint64_t k) {
    return seed ^ (idx * 0x9E3779B97F4A7C15ull + k * 0xD1B54A32D192ED03ull);
}

static void pixel_finish(v4 c, uint32_t ns, uint8_t *rgb, float *acc) { /* lib.rs:105-108 */
    if (acc) { acc[0] = c.x; acc[1] = c.y; acc[2] = c.z; }
    float fns = (float)ns;
    rgb[0] = f32_as_u8(sqrtf(c.x / fns) * 255.99f);
    rgb[1] = f32_as_u8(sqrtf(c.y / fns) * 255.99f);
    rgb[2] = f32_as_u8(sqrtf(c.z / fns) * 255.99f);
}

/* One sample of pixel idx drawn from stream g (lib.rs:93-102) */
static v4 pixel_sample(const scene_t *sc, const camera_t *cam, uint32_t nx, uint32_t ny, uint32_t idx, xrng *g,
                       ctrs_t *C, const opts_t *O) {
    float x = (float)(idx % nx);
    float y = (float)(ny - 1 - idx / nx);
    float s = (x + rng_gen_f32(g)) / (float)nx;
    float t = (y + rng_gen_f32(g)) / (float)ny;
    ray_t r = camera_gen_ray(cam, s, t, g);
    C->paths++;
    return trace(sc, &r, 0, g, C, O);
}

enum { MODE_REF_SERIAL = 0, MODE_PER_SAMPLE = 1, MODE_REF_POOL = 2 };

typedef struct {
    const scene_t *sc; const camera_t *cam;
    uint32_t nx, ny, ns; opts_t O; int mode; uint64_t seed; xrng base;
    uint8_t *rgb; float *acc; float *per_sample; /* optional [ns][npix][3] */
    uint32_t k_begin, k_step;
    volatile uint64_t next; uint64_t npix; int n_threads;
    ctrs_t *thread_ctrs;
} job_t;

typedef struct { job_t *job; int tid; } worker_arg_t;

static void render_pixel(job_t *J, uint32_t idx, xrng *stream, ctrs_t *C) {
    v4 c = v4_splat(0.0f);
    if (J->mode == MODE_PER_SAMPLE) {
        for (uint32_t k = J->k_begin; k < J->ns; k += J->k_step) {
            xrng g; rng_seed_from_u64(&g, sample_stream_seed(J->seed, idx, k));
            v4 s = pixel_sample(J->sc, J->cam, J->nx, J->ny, idx, &g, C, &J->O);
            if (J->per_sample) {
                float *ps = J->per_sample + ((size_t)k * J->npix + idx) * 3;
                ps[0] = s.x; ps[1] = s.y; ps[2] = s.z;
            }
            c = v4_add(c, s);
        }
    } else {
        for (uint32_t k = 0; k < J->ns; k++) c = v4_add(c, pixel_sample(J->sc, J->cam, J->nx, J->ny, idx, stream, C, &J->O));
    }
    pixel_finish(c, J->ns, J->rgb + (size_t)idx * 3, J->acc ? J->acc + (size_t)idx * 3 : NULL);
}

static void *worker(void *p) {
    worker_arg_t *a = (worker_arg_t *)p;
    job_t *J = a->job;
    ctrs_t *C = &J->thread_ctrs[a->tid];
    xrng stream = J->base;
    /* threadpool.rs:14-24: worker i clones the base generator and jumps i times */
    if (J->mode == MODE_REF_POOL) for (int i = 0; i < a->tid; i++) rng_jump(&stream);
    for (;;) {
        /* one work item per pixel, claimed dynamically (rayon's par_chunks_mut(3) over pixels, lib.rs:85-89) */
        uint64_t lo = __atomic_fetch_add(&J->next, 16, __ATOMIC_RELAXED);
        if (lo >= J->npix) break;
        uint64_t hi = lo + 16 < J->npix ? lo + 16 : J->npix;
        for (uint64_t idx = lo; idx < hi; idx++) render_pixel(J, (uint32_t)idx, &stream, C);
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* Scene builders                                                                               */
/* ------------------------------------------------------------------------------------------ */
static void scene_push(scene_t *sc, sphere_t s, uint32_t *cap) {
    if (sc->n_spheres == *cap) { *cap = *cap ? *cap * 2 : 64; sc->spheres = (sphere_t *)realloc(sc->spheres, (size_t)*cap * sizeof(sphere_t)); }
    sc->spheres[sc->n_spheres++] = s;
}
static material_t mat_lambertian(float r, float g, float b) { material_t m = {MAT_LAMBERTIAN, {r, g, b, 1.0f}, 0.0f, 0.0f}; return m; }
static material_t mat_metal(float r, float g, float b, float fuzz) { material_t m = {MAT_METAL, {r, g, b, 1.0f}, fuzz, 0.0f}; return m; }
static material_t mat_dielectric(float ior) { material_t m = {MAT_DIELECTRIC, {1.0f, 1.0f, 1.0f, 0.0f}, 0.0f, ior}; return m; }

static void build_chap11(scene_t *sc, camera_t *cam, uint32_t nx, uint32_t ny) { /* chap11.rs:8-63 */
    uint32_t cap = 0;
    scene_push(sc, sphere_stationary(v4_new(0.0f, 0.0f, -1.0f, 0.0f), 0.5f, mat_lambertian(0.1f, 0.2f, 0.5f)), &cap);
    scene_push(sc, sphere_stationary(v4_new(0.0f, -100.5f, -1.0f, 0.0f), 100.0f, mat_lambertian(0.8f, 0.8f, 0.0f)), &cap);
    scene_push(sc, sphere_stationary(v4_new(1.0f, 0.0f, -1.0f, 0.0f), 0.5f, mat_metal(0.8f, 0.6f, 0.2f, 0.3f)), &cap);
    scene_push(sc, sphere_stationary(v4_new(-1.0f, 0.0f, -1.0f, 0.0f), 0.5f, mat_dielectric(1.5f)), &cap);
    scene_push(sc, sphere_stationary(v4_new(-1.0f, 0.0f, -1.0f, 0.0f), -0.45f, mat_dielectric(1.5f)), &cap);
    scene_build_bvh(sc, PART_MIDDLE);
    camera_ci_t ci;
    ci.look_from = v4_new(3.0f, 3.0f, 2.0f, 0.0f);
    ci.look_at = v4_new(0.0f, 0.0f, -1.0f, 0.0f);
    ci.up = v4_new(0.0f, 1.0f, 0.0f, 0.0f);
    ci.fov = 20.0f; ci.aspect = (float)nx / (float)ny; ci.aperture = 2.0f;
    ci.focus_dist = v4_mag(v4_sub(ci.look_from, ci.look_at));
    ci.time0 = 0.0f; ci.time1 = 0.0f;
    *cam = camera_from(ci);
}

/* chap12.rs:9-115 and book2chap2.rs:10-126 share one body; `moving` selects the latter.
 * half = 11 reproduces the reference; a larger half gives the synthetic stress scene, whose
 * camera is the chap12 camera scaled by half/11. */
static void build_random_spheres(scene_t *sc, camera_t *cam, uint32_t nx, uint32_t ny, xrng *g, int moving, int half) {
    uint32_t cap = 0;
    xrng g2 = *g; /* book2chap2.rs:13 */
    uniform_f32 dy = uniform_new_inclusive(0.0f, 0.5f);
    scene_push(sc, sphere_stationary(v4_new(0.0f, -1000.0f, 0.0f, 0.0f), 1000.0f, mat_lambertian(0.5f, 0.5f, 0.5f)), &cap);
    for (int a = -half; a < half; a++)
        for (int b = -half; b < half; b++) {
            float cx = (float)a + 0.9f * rng_gen_f32(g);
            float cz = (float)b + 0.9f * rng_gen_f32(g);
            v4 center = v4_new(cx, 0.2f, cz, 0.0f);
            if (v4_mag(v4_sub(center, v4_new(4.0f, 0.2f, 0.0f, 0.0f))) > 0.9f) {
                float choose = rng_gen_f32(g);
                if (choose < 0.8f) {
                    v4 c1 = center;
                    if (moving) c1 = v4_add(center, v4_new(0.0f, uniform_sample(dy, &g2), 0.0f, 0.0f));
                    float r0 = rng_gen_f32(g), r1 = rng_gen_f32(g), g0 = rng_gen_f32(g), g1 = rng_gen_f32(g),
                          b0 = rng_gen_f32(g), b1 = rng_gen_f32(g);
                    scene_push(sc, sphere_new(center, c1, 0.0f, 1.0f, 0.2f, mat_lambertian(r0 * r1, g0 * g1, b0 * b1)), &cap);
                } else if (choose < 0.95f) {
                    float r = 0.5f * (1.0f + rng_gen_f32(g)), gg = 0.5f * (1.0f + rng_gen_f32(g)),
                          bb = 0.5f * (1.0f + rng_gen_f32(g)), fz = 0.5f * rng_gen_f32(g);
                    scene_push(sc, sphere_stationary(center, 0.2f, mat_metal(r, gg, bb, fz)), &cap);
                } else {
                    scene_push(sc, sphere_stationary(center, 0.2f, mat_dielectric(1.5f)), &cap);
                }
            }
        }
    scene_push(sc, sphere_stationary(v4_new(0.0f, 1.0f, 0.0f, 0.0f), 1.0f, mat_dielectric(1.5f)), &cap);
    scene_push(sc, sphere_stationary(v4_new(-4.0f, 1.0f, 0.0f, 0.0f), 1.0f, mat_lambertian(0.4f, 0.2f, 0.1f)), &cap);
    scene_push(sc, sphere_stationary(v4_new(4.0f, 1.0f, 0.0f, 0.0f), 1.0f, mat_metal(0.7f, 0.6f, 0.5f, 0.0f)), &cap);
    scene_build_bvh(sc, PART_MIDDLE);
    float k = (float)half / 11.0f;
    camera_ci_t ci;
    ci.look_from = v4_new(13.0f * k, 2.0f * k, 3.0f * k, 0.0f);
    ci.look_at = v4_new(0.0f, 0.0f, 0.0f, 0.0f);
    ci.up = v4_new(0.0f, 1.0f, 0.0f, 0.0f);
    ci.fov = 20.0f; ci.aspect = (float)nx / (float)ny; ci.aperture = 0.1f; ci.focus_dist = 10.0f * k;
    ci.time0 = 0.0f; ci.time1 = moving ? 1.0f : 0.0f;
    *cam = camera_from(ci);
}

/* ------------------------------------------------------------------------------------------ */
/* Exported C surface (ctypes)                                                                  */
/* ------------------------------------------------------------------------------------------ */
#define ORC_API __attribute__((visibility("default")))

ORC_API void orc_rng_seed(uint64_t seed, uint32_t st[4]) { xrng g; rng_seed_from_u64(&g, seed); memcpy(st, g.s, 16); }
ORC_API uint32_t orc_rng_next_u32(uint32_t st[4]) { xrng g; memcpy(g.s, st, 16); uint32_t r = rng_next_u32(&g); memcpy(st, g.s, 16); return r; }
ORC_API float orc_rng_gen_f32(uint32_t st[4]) { xrng g; memcpy(g.s, st, 16); float r = rng_gen_f32(&g); memcpy(st, g.s, 16); return r; }
ORC_API void orc_rng_jump(uint32_t st[4]) { xrng g; memcpy(g.s, st, 16); rng_jump(&g); memcpy(st, g.s, 16); }
ORC_API uint64_t orc_splitmix64(uint64_t *st) { return splitmix64_next(st); }
ORC_API void orc_uniform_new_inclusive(float lo, float hi, float *low, float *scale) { uniform_f32 u = uniform_new_inclusive(lo, hi); *low = u.low; *scale = u.scale; }
ORC_API float orc_uniform_sample(float low, float scale, uint32_t st[4]) { xrng g; memcpy(g.s, st, 16); uniform_f32 u = {low, scale}; float r = uniform_sample(u, &g); memcpy(st, g.s, 16); return r; }
ORC_API float orc_gamma_eb(int n) { return gamma_eb(n); }
ORC_API float orc_far_scale(void) { return 1.0f + 2.0f * gamma_eb(3); }
ORC_API uint64_t orc_sample_stream_seed(uint64_t seed, uint64_t idx, uint64_t k) { return sample_stream_seed(seed, idx, k); }
ORC_API float orc_reflectance(float cosine, float ref_idx, int pow5_mode) { return reflectance(cosine, ref_idx, pow5_mode); }
ORC_API uint8_t orc_f32_as_u8(float x) { return f32_as_u8(x); }
ORC_API void orc_gen_in_unit_sphere(uint32_t st[4], float out[3]) { xrng g; memcpy(g.s, st, 16); v4 v = gen_in_unit_sphere(&g); memcpy(st, g.s, 16); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
ORC_API void orc_gen_in_unit_disc(uint32_t st[4], float out[2]) { xrng g; memcpy(g.s, st, 16); v4 v = gen_in_unit_disc(&g); memcpy(st, g.s, 16); out[0] = v.x; out[1] = v.y; }

typedef struct { scene_t sc; camera_t cam; xrng rng_after; } orc_scene;

/* kind: 0 chap11, 1 chap12, 2 book2chap2, 3 synthetic (chap12 body over -half..half) */
ORC_API orc_scene *orc_scene_build(int kind, uint32_t nx, uint32_t ny, uint64_t seed, int half) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    rng_seed_from_u64(&s->rng_after, seed);
    switch (kind) {
    case 0: build_chap11(&s->sc, &s->cam, nx, ny); break;
    case 1: build_random_spheres(&s->sc, &s->cam, nx, ny, &s->rng_after, 0, 11); break;
    case 2: build_random_spheres(&s->sc, &s->cam, nx, ny, &s->rng_after, 1, 11); break;
    case 3: build_random_spheres(&s->sc, &s->cam, nx, ny, &s->rng_after, 0, half); break;
    default: free(s); return NULL;
    }
    return s;
}

/* Scene from flattened arrays: spheres as 9 float planes + per-sphere material, nodes in the
 * reference's pre-order as 8 x u32/f32 words {min[3], max[3], left_or_obj, right|axis<<29|leaf<<31}. */
ORC_API orc_scene *orc_scene_from_arrays(uint32_t n, const float *c0x, const float *c0y, const float *c0z,
                                         const float *c1x, const float *c1y, const float *c1z, const float *time0,
                                         const float *time1, const float *radius, const uint32_t *material_id,
                                         uint32_t n_mat, const uint32_t *mat_kind, const float *mat_r, const float *mat_g,
                                         const float *mat_b, const float *mat_fuzz, const float *mat_ior,
                                         uint32_t n_nodes, const uint32_t *node_words) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->sc.spheres = (sphere_t *)malloc((size_t)(n ? n : 1) * sizeof(sphere_t));
    s->sc.n_spheres = n;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t m = material_id[i];
        material_t mt;
        if (m >= n_mat) { free(s->sc.spheres); free(s); return NULL; }
        mt.kind = (int)mat_kind[m];
        mt.albedo = (mt.kind == MAT_DIELECTRIC) ? v4_new(1.0f, 1.0f, 1.0f, 0.0f) : v4_new(mat_r[m], mat_g[m], mat_b[m], 1.0f);
        mt.fuzz = mat_fuzz[m]; mt.ref_idx = mat_ior[m];
        s->sc.spheres[i] = sphere_new(v4_new(c0x[i], c0y[i], c0z[i], 0.0f), v4_new(c1x[i], c1y[i], c1z[i], 0.0f), time0[i],
                                      time1[i], radius[i], mt);
    }
    s->sc.nodes = (bvh_node_t *)malloc((size_t)(n_nodes ? n_nodes : 1) * sizeof(bvh_node_t));
    s->sc.n_nodes = s->sc.cap_nodes = n_nodes;
    for (uint32_t i = 0; i < n_nodes; i++) {
        const uint32_t *w = node_words + (size_t)i * 8;
        bvh_node_t nd; memset(&nd, 0, sizeof nd);
        float f[6]; memcpy(f, w, 24);
        if (w[7] >> 31) { nd.kind = NODE_LEAF; nd.obj_i = w[6]; nd.left_i = nd.right_i = -1; }
        else {
            nd.kind = NODE_INNER;
            nd.aabb.minimum = v4_new(f[0], f[1], f[2], 0.0f); /* lane w is never read by the slab test */
            nd.aabb.maximum = v4_new(f[3], f[4], f[5], 0.0f);
            nd.axis = (int)((w[7] >> 29) & 3u);
            nd.left_i = w[6]; nd.right_i = w[7] & 0x1FFFFFFFu;
        }
        s->sc.nodes[i] = nd;
    }
    return s;
}

/* Scene::build_bvh (scene.rs:32-35) on a scene made from arrays; method 0 = Middle, 1 = EqualCount */
ORC_API void orc_scene_build_bvh(orc_scene *s, int method) { scene_build_bvh(&s->sc, method); }

ORC_API void orc_scene_free(orc_scene *s) { if (!s) return; free(s->sc.spheres); free(s->sc.nodes); free(s); }
ORC_API uint32_t orc_scene_num_spheres(const orc_scene *s) { return s->sc.n_spheres; }
ORC_API uint32_t orc_scene_num_nodes(const orc_scene *s) { return s->sc.n_nodes; }
ORC_API void orc_scene_rng_after(const orc_scene *s, uint32_t st[4]) { memcpy(st, s->rng_after.s, 16); }

/* spheres: n x 9 floats {c0xyz, c1xyz, time0, time1, radius}; mats: n x 6 floats {kind, r, g, b, fuzz, ior} */
ORC_API void orc_scene_export_spheres(const orc_scene *s, float *spheres, float *mats) {
    for (uint32_t i = 0; i < s->sc.n_spheres; i++) {
        const sphere_t *p = &s->sc.spheres[i];
        float *o = spheres + (size_t)i * 9;
        o[0] = p->center0.x; o[1] = p->center0.y; o[2] = p->center0.z;
        o[3] = p->center1.x; o[4] = p->center1.y; o[5] = p->center1.z;
        o[6] = p->time0; o[7] = p->time1; o[8] = p->radius;
        float *m = mats + (size_t)i * 6;
        m[0] = (float)p->material.kind; m[1] = p->material.albedo.x; m[2] = p->material.albedo.y; m[3] = p->material.albedo.z;
        m[4] = p->material.fuzz; m[5] = p->material.ref_idx;
    }
}
/* nodes: n_nodes x 8 words in the interchange layout described at orc_scene_from_arrays */
ORC_API void orc_scene_export_nodes(const orc_scene *s, uint32_t *words) {
    for (uint32_t i = 0; i < s->sc.n_nodes; i++) {
        const bvh_node_t *n = &s->sc.nodes[i];
        uint32_t *w = words + (size_t)i * 8;
        if (n->kind == NODE_LEAF) { memset(w, 0, 32); w[6] = n->obj_i; w[7] = 1u << 31; }
        else {
            float f[6] = {n->aabb.minimum.x, n->aabb.minimum.y, n->aabb.minimum.z, n->aabb.maximum.x, n->aabb.maximum.y, n->aabb.maximum.z};
            memcpy(w, f, 24);
            w[6] = (uint32_t)n->left_i; w[7] = ((uint32_t)n->right_i & 0x1FFFFFFFu) | ((uint32_t)n->axis << 29);
        }
    }
}
/* camera: 7 x vec3 {llc, horizontal, vertical, origin, u, v, w} + lens_radius, time_low, time_scale = 24 floats */
ORC_API void orc_scene_export_camera(const orc_scene *s, float *out) {
    const camera_t *c = &s->cam;
    const v4 *vs[7] = {&c->lower_left_corner, &c->horizontal, &c->vertical, &c->origin, &c->u, &c->v, &c->w};
    for (int i = 0; i < 7; i++) { out[i * 3] = vs[i]->x; out[i * 3 + 1] = vs[i]->y; out[i * 3 + 2] = vs[i]->z; }
    out[21] = c->lens_radius; out[22] = c->time_dist.low; out[23] = c->time_dist.scale;
}
ORC_API void orc_scene_set_camera(orc_scene *s, const float *in) {
    camera_t *c = &s->cam;
    v4 *vs[7] = {&c->lower_left_corner, &c->horizontal, &c->vertical, &c->origin, &c->u, &c->v, &c->w};
    for (int i = 0; i < 7; i++) *vs[i] = v4_new(in[i * 3], in[i * 3 + 1], in[i * 3 + 2], 0.0f);
    c->lens_radius = in[21]; c->time_dist.low = in[22]; c->time_dist.scale = in[23];
}
/* camera from a create-info (camera.rs:18-54): 9 floats look_from, look_at, up then fov, aspect, aperture, focus, t0, t1 */
ORC_API void orc_camera_from(const float *ci15, float *out24) {
    camera_ci_t ci;
    ci.look_from = v4_new(ci15[0], ci15[1], ci15[2], 0.0f); ci.look_at = v4_new(ci15[3], ci15[4], ci15[5], 0.0f);
    ci.up = v4_new(ci15[6], ci15[7], ci15[8], 0.0f);
    ci.fov = ci15[9]; ci.aspect = ci15[10]; ci.aperture = ci15[11]; ci.focus_dist = ci15[12]; ci.time0 = ci15[13]; ci.time1 = ci15[14];
    orc_scene tmp; memset(&tmp, 0, sizeof tmp);
    tmp.cam = camera_from(ci);
    orc_scene_export_camera(&tmp, out24);
}

/* Closest hits for a batch of rays.  method 0 = BVH traversal, 1 = brute force (lib.rs:32-47).
 * out_obj[i] = 0xFFFFFFFF on a miss.  normals/points optional (n x 3). */
ORC_API int orc_hit_rays(const orc_scene *s, uint32_t n, const float *org, const float *dir, const float *time,
                         float t_min, float t_max, int method, float *out_t, uint32_t *out_obj, float *out_p,
                         float *out_n, uint8_t *out_front, uint64_t *visits /* [2]: inner, leaf */) {
    ctrs_t C; memset(&C, 0, sizeof C);
    g_stack_overflow = 0;
    for (uint32_t i = 0; i < n; i++) {
        ray_t r; hit_t h;
        r.origin = v4_new(org[i * 3], org[i * 3 + 1], org[i * 3 + 2], 0.0f);
        r.direction = v4_new(dir[i * 3], dir[i * 3 + 1], dir[i * 3 + 2], 0.0f);
        r.time = time ? time[i] : 0.0f;
        int f = method == 0 ? bvh_hit(&s->sc, &r, t_min, t_max, &h, &C) : closest_hit_brute(&s->sc, &r, t_min, t_max, &h);
        if (f) {
            out_t[i] = h.t; out_obj[i] = h.obj;
            if (out_p) { out_p[i * 3] = h.p.x; out_p[i * 3 + 1] = h.p.y; out_p[i * 3 + 2] = h.p.z; }
            if (out_n) { out_n[i * 3] = h.normal.x; out_n[i * 3 + 1] = h.normal.y; out_n[i * 3 + 2] = h.normal.z; }
            if (out_front) out_front[i] = (uint8_t)h.front_face;
        } else {
            out_t[i] = t_max; out_obj[i] = 0xFFFFFFFFu;
            if (out_front) out_front[i] = 0;
        }
    }
    if (visits) { visits[0] = C.inner; visits[1] = C.leaf; }
    return g_stack_overflow ? -1 : 0;
}

/* Primary rays exactly as lib.rs:93-101 + camera.rs:56-66 would generate them in per-sample mode.
 * out: n x 7 floats {origin xyz, direction xyz, time}, and the generator state after generation. */
ORC_API void orc_gen_primary(const orc_scene *s, uint32_t nx, uint32_t ny, uint64_t seed, uint32_t n, const uint32_t *idx,
                             const uint32_t *k, float *out7, uint32_t *out_state) {
    for (uint32_t i = 0; i < n; i++) {
        xrng g; rng_seed_from_u64(&g, sample_stream_seed(seed, idx[i], k[i]));
        float x = (float)(idx[i] % nx), y = (float)(ny - 1 - idx[i] / nx);
        float sx = (x + rng_gen_f32(&g)) / (float)nx;
        float ty = (y + rng_gen_f32(&g)) / (float)ny;
        ray_t r = camera_gen_ray(&s->cam, sx, ty, &g);
        float *o = out7 + (size_t)i * 7;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z; o[6] = r.time;
        if (out_state) memcpy(out_state + (size_t)i * 4, g.s, 16);
    }
}

/* counters out: rays, inner, leaf, h_l, h_m, h_d, sky, paths, depth_cut, absorbed */
#define ORC_NCTRS 10

/* The render entry (lib.rs:73-113).
 *  mode 0 ref-serial : one stream `state` consumed over pixels in index order — the reference run on one thread.
 *  mode 1 per-sample : stream per (pixel, sample) from `seed`; scheduling-independent; what the GPU computes.
 *                      Samples k_begin, k_begin+k_step, ... are rendered (k_step 1 = all).
 *  mode 2 ref-pool   : n_threads workers, worker i owns `state` jumped i times, pixels claimed dynamically
 *                      (threadpool.rs:11-27 + lib.rs:85-92); not reproducible across runs, used for timing.
 * rgb: nx*ny*3 bytes, row 0 = top.  acc (optional): nx*ny*3 floats, the un-normalised sums.
 * per_sample (optional, mode 1): ns*nx*ny*3 floats. */
ORC_API int orc_render(const orc_scene *s, uint32_t nx, uint32_t ny, uint32_t ns, int max_depth, int mode, uint64_t seed,
                       const uint32_t state[4], int n_threads, int pow5_mode, uint32_t k_begin, uint32_t k_step,
                       uint8_t *rgb, float *acc, float *per_sample, uint64_t *ctrs_out) {
    job_t J; memset(&J, 0, sizeof J);
    if (n_threads < 1) n_threads = 1;
    if (mode == MODE_REF_SERIAL) n_threads = 1;
    if (k_step == 0) k_step = 1;
    J.sc = &s->sc; J.cam = &s->cam; J.nx = nx; J.ny = ny; J.ns = ns;
    J.O.max_depth = max_depth; J.O.pow5_mode = pow5_mode; J.mode = mode; J.seed = seed;
    if (state) memcpy(J.base.s, state, 16); else rng_seed_from_u64(&J.base, seed);
    J.rgb = rgb; J.acc = acc; J.per_sample = (mode == MODE_PER_SAMPLE) ? per_sample : NULL;
    J.k_begin = k_begin; J.k_step = k_step;
    J.npix = (uint64_t)nx * ny; J.n_threads = n_threads;
    J.thread_ctrs = (ctrs_t *)calloc((size_t)n_threads, sizeof(ctrs_t));
    g_stack_overflow = 0;
    if (n_threads == 1) {
        worker_arg_t a = {&J, 0};
        worker(&a);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        worker_arg_t *args = (worker_arg_t *)malloc(sizeof(worker_arg_t) * (size_t)n_threads);
        pthread_attr_t attr; pthread_attr_init(&attr); pthread_attr_setstacksize(&attr, 16u << 20);
        for (int i = 0; i < n_threads; i++) { args[i].job = &J; args[i].tid = i; pthread_create(&th[i], &attr, worker, &args[i]); }
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
        pthread_attr_destroy(&attr);
        free(th); free(args);
    }
    if (ctrs_out) {
        memset(ctrs_out, 0, sizeof(uint64_t) * ORC_NCTRS);
        for (int i = 0; i < n_threads; i++) {
            const ctrs_t *c = &J.thread_ctrs[i];
            ctrs_out[0] += c->rays; ctrs_out[1] += c->inner; ctrs_out[2] += c->leaf; ctrs_out[3] += c->h_l; ctrs_out[4] += c->h_m;
            ctrs_out[5] += c->h_d; ctrs_out[6] += c->sky; ctrs_out[7] += c->paths; ctrs_out[8] += c->depth_cut; ctrs_out[9] += c->absorbed;
        }
    }
    free(J.thread_ctrs);
    return g_stack_overflow ? -1 : 0;
}

/* Chosen pixels only, per-sample mode: acc n x 3 floats (sums), rgb n x 3 bytes.  Lets a full-size
 * image be spot-checked without rendering all of it on the CPU. */
ORC_API void orc_render_pixels(const orc_scene *s, uint32_t nx, uint32_t ny, uint32_t ns, uint64_t seed, int max_depth,
                               int pow5_mode, uint32_t n, const uint32_t *idx, float *acc, uint8_t *rgb, uint64_t *rays) {
    opts_t O = {max_depth, pow5_mode};
    ctrs_t C; memset(&C, 0, sizeof C);
    for (uint32_t i = 0; i < n; i++) {
        v4 c = v4_splat(0.0f);
        for (uint32_t k = 0; k < ns; k++) {
            xrng g; rng_seed_from_u64(&g, sample_stream_seed(seed, idx[i], k));
            c = v4_add(c, pixel_sample(&s->sc, &s->cam, nx, ny, idx[i], &g, &C, &O));
        }
        pixel_finish(c, ns, rgb + (size_t)i * 3, acc + (size_t)i * 3);
    }
    if (rays) *rays = C.rays;
}

/* The same for one sample shard: samples k = k_begin, k_begin + k_step, ... < ns of the chosen pixels, added in that
 * order (what a rank of a sample-sharded render accumulates); rgb is quantised with the shard's own sample count. */
ORC_API void orc_render_pixels_shard(const orc_scene *s, uint32_t nx, uint32_t ny, uint32_t ns, uint64_t seed, int max_depth,
                                     int pow5_mode, uint32_t k_begin, uint32_t k_step, uint32_t n, const uint32_t *idx, float *acc,
                                     uint8_t *rgb, uint64_t *rays) {
    opts_t O = {max_depth, pow5_mode};
    ctrs_t C; memset(&C, 0, sizeof C);
    if (k_step == 0) k_step = 1;
    for (uint32_t i = 0; i < n; i++) {
        v4 c = v4_splat(0.0f);
        uint32_t n_local = 0;
        for (uint32_t k = k_begin; k < ns; k += k_step) {
            xrng g; rng_seed_from_u64(&g, sample_stream_seed(seed, idx[i], k));
            c = v4_add(c, pixel_sample(&s->sc, &s->cam, nx, ny, idx[i], &g, &C, &O));
            n_local++;
        }
        pixel_finish(c, n_local ? n_local : 1, rgb + (size_t)i * 3, acc + (size_t)i * 3);
    }
    if (rays) *rays = C.rays;
}

/* The rays of one path in the order trace() meets them: (origin, direction, time) x up to `cap`; returns how many the path has. */
ORC_API uint64_t orc_trace_path_rays(const orc_scene *s, uint32_t nx, uint32_t ny, uint64_t seed, uint32_t idx, uint32_t k,
                                     int max_depth, int pow5_mode, float *rays7, uint64_t cap) {
    opts_t O = {max_depth, pow5_mode};
    ctrs_t C; memset(&C, 0, sizeof C);
    xrng g; rng_seed_from_u64(&g, sample_stream_seed(seed, idx, k));
    g_ray_log = rays7; g_ray_log_cap = cap; g_ray_log_n = 0;
    (void)pixel_sample(&s->sc, &s->cam, nx, ny, idx, &g, &C, &O);
    g_ray_log = NULL;
    return g_ray_log_n;
}

/* One path, for debugging and fine-grained parity: radiance of (pixel idx, sample k) in per-sample mode. */
ORC_API void orc_trace_path(const orc_scene *s, uint32_t nx, uint32_t ny, uint64_t seed, uint32_t idx, uint32_t k,
                            int max_depth, int pow5_mode, float out_rgb[3], uint64_t *rays) {
    opts_t O = {max_depth, pow5_mode};
    ctrs_t C; memset(&C, 0, sizeof C);
    xrng g; rng_seed_from_u64(&g, sample_stream_seed(seed, idx, k));
    v4 c = pixel_sample(&s->sc, &s->cam, nx, ny, idx, &g, &C, &O);
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
    if (rays) *rays = C.rays;
}
