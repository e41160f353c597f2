// Host side of the render path: scene, camera and BVH construction as the reference does them,
// flattening, and tracescene() over the C ABI.  Compiled with -ffp-contract=off: Rust never
// contracts a*b+c, and ultraviolet's explicit mul_add calls are std::fmaf here.
#include "rrt.hpp"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace rrt {

// ------------------------------------------------------------------ Vec4f (ultraviolet 0.9.2)
Vec4f operator+(const Vec4f &a, const Vec4f &b) { return Vec4f(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
Vec4f operator-(const Vec4f &a, const Vec4f &b) { return Vec4f(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
Vec4f operator*(float s, const Vec4f &a) { return Vec4f(s * a.x, s * a.y, s * a.z, s * a.w); }

float Vec4f::dot(const Vec4f &o) const { return std::fmaf(x, o.x, std::fmaf(y, o.y, std::fmaf(z, o.z, w * o.w))); }
float Vec4f::mag() const { return std::sqrt(dot(*this)); }
Vec4f Vec4f::normalized() const {
    const float r = 1.0f / mag();
    return Vec4f(x * r, y * r, z * r, w * r);
}
namespace {
// f32::min / f32::max: the operand that is not NaN wins
inline float fmin_rs(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
inline float fmax_rs(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
}  // namespace
Vec4f Vec4f::min_by_component(const Vec4f &o) const { return Vec4f(fmin_rs(x, o.x), fmin_rs(y, o.y), fmin_rs(z, o.z), fmin_rs(w, o.w)); }
Vec4f Vec4f::max_by_component(const Vec4f &o) const { return Vec4f(fmax_rs(x, o.x), fmax_rs(y, o.y), fmax_rs(z, o.z), fmax_rs(w, o.w)); }

// ------------------------------------------------------------------ RttRng (rand_xoshiro 0.6.0)
namespace {
inline uint64_t splitmix64(uint64_t &st) {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
}  // namespace

RttRng RttRng::seed_from_u64(uint64_t seed) {
    RttRng r;
    uint64_t st = seed;
    const uint64_t a = splitmix64(st), b = splitmix64(st);
    r.s_[0] = static_cast<uint32_t>(a); r.s_[1] = static_cast<uint32_t>(a >> 32);
    r.s_[2] = static_cast<uint32_t>(b); r.s_[3] = static_cast<uint32_t>(b >> 32);
    if ((r.s_[0] | r.s_[1] | r.s_[2] | r.s_[3]) == 0) return seed_from_u64(0);
    return r;
}
RttRng RttRng::from_state(const uint32_t s[4]) {
    RttRng r;
    std::memcpy(r.s_, s, 16);
    if ((r.s_[0] | r.s_[1] | r.s_[2] | r.s_[3]) == 0) return seed_from_u64(0);
    return r;
}
uint32_t RttRng::next_u32() {
    const uint32_t out = s_[0] + s_[3];
    const uint32_t t = s_[1] << 9;
    s_[2] ^= s_[0]; s_[3] ^= s_[1]; s_[1] ^= s_[2]; s_[0] ^= s_[3];
    s_[2] ^= t;
    s_[3] = (s_[3] << 11) | (s_[3] >> 21);
    return out;
}
uint64_t RttRng::next_u64() {
    const uint64_t lo = next_u32();
    const uint64_t hi = next_u32();
    return lo | (hi << 32);
}
float RttRng::gen_f32() { return static_cast<float>(next_u32() >> 8) * 0x1p-24f; }
void RttRng::jump() {
    static const uint32_t kJump[4] = {0x8764000bu, 0xf542d2d3u, 0x6fa035c3u, 0x77f2db5bu};
    uint32_t acc[4] = {0, 0, 0, 0};
    for (uint32_t word : kJump)
        for (int b = 0; b < 32; b++) {
            if (word & (1u << b)) for (int k = 0; k < 4; k++) acc[k] ^= s_[k];
            next_u32();
        }
    std::memcpy(s_, acc, 16);
}

Uniform Uniform::new_inclusive(float low, float high) {  // rand 0.8.5 UniformFloat::new_inclusive
    const float max_rand = 1.0f - 0x1p-23f;
    float scale = (high - low) / max_rand;
    while (scale * max_rand + low > high) {
        uint32_t bits; std::memcpy(&bits, &scale, 4); bits -= 1; std::memcpy(&scale, &bits, 4);
    }
    Uniform u; u.low = low; u.scale = scale;
    return u;
}
float Uniform::sample(RttRng &rng) const {
    const float v01 = static_cast<float>(rng.next_u32() >> 9) * 0x1p-23f;
    return v01 * scale + low;
}

// ------------------------------------------------------------------ materials
MaterialDesc Lambertian::desc() const { return MaterialDesc{RRT_MAT_LAMBERTIAN, {albedo.x, albedo.y, albedo.z}, 0.f, 0.f}; }
MaterialDesc Metal::desc() const { return MaterialDesc{RRT_MAT_METAL, {albedo.x, albedo.y, albedo.z}, fuzz, 0.f}; }
MaterialDesc Dielectric::desc() const { return MaterialDesc{RRT_MAT_DIELECTRIC, {1.f, 1.f, 1.f}, 0.f, ref_idx}; }

// ------------------------------------------------------------------ Aabb (acceleration.rs:15-45)
Aabb Aabb::union_(const Aabb &b) const { Aabb r; r.minimum = minimum.min_by_component(b.minimum); r.maximum = maximum.max_by_component(b.maximum); return r; }
Aabb Aabb::union_p(const Vec4f &p) const { Aabb r; r.minimum = minimum.min_by_component(p); r.maximum = maximum.max_by_component(p); return r; }
Vec4f Aabb::centroid() const { return 0.5f * minimum + 0.5f * maximum; }
Vec4f Aabb::diagonal() const { return maximum - minimum; }
Axis Aabb::max_extent() const {
    const Vec4f d = diagonal();
    if (d.x > d.y && d.x > d.z) return Axis::X;
    if (d.y > d.z) return Axis::Y;
    return Axis::Z;
}

// ------------------------------------------------------------------ Sphere (sphere.rs:25-59)
namespace {
Aabb sphere_aabb(const Vec4f &c0, const Vec4f &c1, float radius) {
    Aabb b0; b0.minimum = c0 - Vec4f::broadcast(radius); b0.maximum = c0 + Vec4f::broadcast(radius);
    Aabb b1; b1.minimum = c1 - Vec4f::broadcast(radius); b1.maximum = c1 + Vec4f::broadcast(radius);
    return b0.union_(b1);
}
}  // namespace

Sphere::Sphere(Vec4f center0, Vec4f center1, float time0, float time1, float radius, std::shared_ptr<Material> material)
    : center0_(center0), center1_(center1), time0_(time0), time1_(time1), radius_(radius), material_(std::move(material)),
      aabb_(sphere_aabb(center0, center1, radius)) {}

Sphere Sphere::from(StationarySphere s) { return Sphere(s.center, s.center, 0.f, 1.f, s.radius, std::move(s.material)); }

// ------------------------------------------------------------------ Bvh (acceleration.rs:131-217)
void Bvh::build(const std::vector<const Hittable *> &objects, BvhPartitionMethod method) {
    std::vector<Info> infos(objects.size());
    for (size_t i = 0; i < objects.size(); i++) {
        infos[i].index = static_cast<uint32_t>(i);
        infos[i].aabb = objects[i]->aabb();
        infos[i].centroid = infos[i].aabb.centroid();
    }
    nodes_.clear();
    nodes_.reserve(objects.empty() ? 0 : 2 * objects.size() - 1);
    build_r(infos.data(), infos.size(), method);
}

int64_t Bvh::build_r(Info *infos, size_t n, BvhPartitionMethod method) {
    if (n == 0) return -1;
    if (n == 1) {
        rrt_bvh_node leaf{};
        leaf.left_or_obj = infos[0].index;
        leaf.right_axis_leaf = RRT_NODE_LEAF_BIT;
        nodes_.push_back(leaf);
        return static_cast<int64_t>(nodes_.size() - 1);
    }
    Aabb box, centroid_bounds;
    for (size_t i = 0; i < n; i++) box = box.union_(infos[i].aabb);
    for (size_t i = 0; i < n; i++) centroid_bounds = centroid_bounds.union_p(infos[i].centroid);
    const int axis = static_cast<int>(centroid_bounds.max_extent());
    size_t pivot = 0;
    for (;;) {
        if (method == BvhPartitionMethod::Middle) {
            const float midpoint = (centroid_bounds.minimum[axis] + centroid_bounds.maximum[axis]) / 2.0f;
            // itertools::partition leaves the order inside each side unspecified; a stable partition is one valid outcome
            Info *mid = std::stable_partition(infos, infos + n, [&](const Info &o) { return o.centroid[axis] < midpoint; });
            pivot = static_cast<size_t>(mid - infos);
            if (pivot == 0 || pivot == n) { method = BvhPartitionMethod::EqualCount; continue; }  // sticks for the subtree (:181,205-206)
        } else {
            pivot = n / 2;
            // select_nth_unstable_by only promises the partition around the pivot; a stable sort is one valid outcome
            std::stable_sort(infos, infos + n, [&](const Info &a, const Info &b) { return a.centroid[axis] < b.centroid[axis]; });
        }
        break;
    }
    rrt_bvh_node inner{};
    inner.min[0] = box.minimum.x; inner.min[1] = box.minimum.y; inner.min[2] = box.minimum.z;
    inner.max[0] = box.maximum.x; inner.max[1] = box.maximum.y; inner.max[2] = box.maximum.z;
    nodes_.push_back(inner);
    const size_t me = nodes_.size() - 1;
    const int64_t l = build_r(infos, pivot, method);
    const int64_t r = build_r(infos + pivot, n - pivot, method);
    nodes_[me].left_or_obj = static_cast<uint32_t>(l);
    nodes_[me].right_axis_leaf = (static_cast<uint32_t>(r) & RRT_NODE_INDEX_MASK) | (static_cast<uint32_t>(axis) << RRT_NODE_AXIS_SHIFT);
    return static_cast<int64_t>(me);
}

// ------------------------------------------------------------------ Scene (scene.rs)
std::vector<const Hittable *> Scene::objects() const {
    std::vector<const Hittable *> v;
    v.reserve(spheres.size());
    for (const Sphere &s : spheres) v.push_back(&s);
    return v;
}
namespace {
// One context for tree builds, made on first use on the current device and kept (null if there is no device).
rrt_ctx *build_context() {
    static std::mutex mu;
    static bool tried = false;
    static rrt_ctx *ctx = nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        if (rrt_create(-1, &ctx) != RRT_OK) ctx = nullptr;
    }
    return ctx;
}
}  // namespace

void Scene::build_bvh(BvhPartitionMethod method) {
    const char *host_only = std::getenv("RRT_HOST_BVH");
    if (spheres.size() >= kGpuBuildMinSpheres && !(host_only && host_only[0] == '1')) {
        if (rrt_ctx *ctx = build_context()) {
            const size_t n = spheres.size();
            std::vector<float> plane[7];
            for (auto &v : plane) v.resize(n);
            for (size_t i = 0; i < n; i++) {
                const Sphere &s = spheres[i];
                plane[0][i] = s.center0().x; plane[1][i] = s.center0().y; plane[2][i] = s.center0().z;
                plane[3][i] = s.center1().x; plane[4][i] = s.center1().y; plane[5][i] = s.center1().z;
                plane[6][i] = s.radius();
            }
            rrt_sphere_soa sp{};
            sp.c0x = plane[0].data(); sp.c0y = plane[1].data(); sp.c0z = plane[2].data();
            sp.c1x = plane[3].data(); sp.c1y = plane[4].data(); sp.c1z = plane[5].data();
            sp.radius = plane[6].data();
            sp.n = static_cast<uint32_t>(n);
            std::vector<rrt_bvh_node> nodes(2 * n - 1);
            uint32_t n_nodes = 0;
            const int rc = rrt_build_bvh_method(ctx, &sp, method == BvhPartitionMethod::EqualCount ? RRT_BVH_EQUAL_COUNT : RRT_BVH_MIDDLE, nodes.data(),
                                                static_cast<uint32_t>(nodes.size()), &n_nodes);
            if (rc == RRT_OK && n_nodes == nodes.size()) { bvh.adopt(std::move(nodes)); return; }
            // (a failed device build — out of device memory, say — is not an error of the scene: the host builds the same tree)
        }
    }
    bvh.build(objects(), method);
}

// ------------------------------------------------------------------ Camera (camera.rs:31-54)
namespace {
struct V3 { float x, y, z; };
V3 cross(V3 a, V3 b) {  // ultraviolet Vec3::cross
    return V3{std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x))};
}
V3 normalized3(V3 a) {
    const float m = std::sqrt(std::fmaf(a.x, a.x, std::fmaf(a.y, a.y, a.z * a.z)));
    const float r = 1.0f / m;
    return V3{a.x * r, a.y * r, a.z * r};
}
}  // namespace

Camera Camera::from(const CameraCreateInfo &ci) {
    Camera c;
    const float theta = ci.fov * 3.14159265358979323846f / 180.0f;
    const float half_height = std::tan(theta / 2.0f);
    const float half_width = ci.aspect * half_height;
    const Vec4f w = (ci.look_from - ci.look_at).normalized();
    const V3 u3 = normalized3(cross(V3{ci.up.x, ci.up.y, ci.up.z}, V3{w.x, w.y, w.z}));
    const V3 v3 = cross(V3{w.x, w.y, w.z}, u3);
    const Vec4f u(u3.x, u3.y, u3.z, 0.f), v(v3.x, v3.y, v3.z, 0.f);
    c.lower_left_corner_ = ci.look_from - ((half_width * ci.focus_dist) * u) - ((half_height * ci.focus_dist) * v) - (ci.focus_dist * w);
    c.horizontal_ = ((2.0f * half_width) * ci.focus_dist) * u;
    c.vertical_ = ((2.0f * half_height) * ci.focus_dist) * v;
    c.origin_ = ci.look_from;
    c.u_ = u; c.v_ = v; c.w_ = w;
    c.lens_radius_ = ci.aperture / 2.0f;
    c.time_dist_ = Uniform::new_inclusive(ci.time0, ci.time1);
    return c;
}

rrt_camera Camera::flatten() const {
    rrt_camera f{};
    const Vec4f *src[6] = {&lower_left_corner_, &horizontal_, &vertical_, &origin_, &u_, &v_};
    float *dst[6] = {f.lower_left_corner, f.horizontal, f.vertical, f.origin, f.u, f.v};
    for (int i = 0; i < 6; i++) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    f.lens_radius = lens_radius_;
    f.time_low = time_dist_.low;
    f.time_scale = time_dist_.scale;
    return f;
}

// ------------------------------------------------------------------ scene builders
std::pair<Scene, Camera> chap11_scene(size_t nx, size_t ny) {  // chap11.rs:8-63
    Scene scene;
    scene.spheres.push_back(Sphere::from({Vec4f(0.f, 0.f, -1.f, 0.f), 0.5f, std::make_shared<Lambertian>(Vec4f(0.1f, 0.2f, 0.5f, 1.f))}));
    scene.spheres.push_back(Sphere::from({Vec4f(0.f, -100.5f, -1.f, 0.f), 100.f, std::make_shared<Lambertian>(Vec4f(0.8f, 0.8f, 0.f, 1.f))}));
    scene.spheres.push_back(Sphere::from({Vec4f(1.f, 0.f, -1.f, 0.f), 0.5f, std::make_shared<Metal>(Vec4f(0.8f, 0.6f, 0.2f, 1.f), 0.3f)}));
    scene.spheres.push_back(Sphere::from({Vec4f(-1.f, 0.f, -1.f, 0.f), 0.5f, std::make_shared<Dielectric>(1.5f)}));
    scene.spheres.push_back(Sphere::from({Vec4f(-1.f, 0.f, -1.f, 0.f), -0.45f, std::make_shared<Dielectric>(1.5f)}));
    scene.build_bvh(BvhPartitionMethod::Middle);
    CameraCreateInfo ci;
    ci.look_from = Vec4f(3.f, 3.f, 2.f, 0.f);
    ci.look_at = Vec4f(0.f, 0.f, -1.f, 0.f);
    ci.up = Vec4f(0.f, 1.f, 0.f, 0.f);
    ci.fov = 20.f;
    ci.aspect = static_cast<float>(nx) / static_cast<float>(ny);
    ci.aperture = 2.f;
    ci.focus_dist = (ci.look_from - ci.look_at).mag();
    ci.time0 = 0.f; ci.time1 = 0.f;
    return {std::move(scene), Camera::from(ci)};
}

namespace {
// chap12.rs:9-115 and book2chap2.rs:10-126 are one generator; `moving` adds the second centre of
// the diffuse spheres drawn from a clone of the generator (book2chap2.rs:13,37-39) and opens the shutter.
std::pair<Scene, Camera> random_spheres(size_t nx, size_t ny, RttRng &rng, bool moving, int half) {
    RttRng rng2 = rng;
    const Uniform rise = Uniform::new_inclusive(0.f, 0.5f);
    Scene scene;
    scene.spheres.push_back(Sphere::from({Vec4f(0.f, -1000.f, 0.f, 0.f), 1000.f, std::make_shared<Lambertian>(Vec4f(0.5f, 0.5f, 0.5f, 1.f))}));
    for (int a = -half; a < half; a++) {
        for (int b = -half; b < half; b++) {
            const float cx = static_cast<float>(a) + 0.9f * rng.gen_f32();
            const float cz = static_cast<float>(b) + 0.9f * rng.gen_f32();
            const Vec4f center(cx, 0.2f, cz, 0.f);
            if ((center - Vec4f(4.f, 0.2f, 0.f, 0.f)).mag() > 0.9f) {
                const float choose_mat = rng.gen_f32();
                if (choose_mat < 0.8f) {
                    Vec4f center1 = center;
                    if (moving) center1 = center + Vec4f(0.f, rise.sample(rng2), 0.f, 0.f);
                    const float r0 = rng.gen_f32(), r1 = rng.gen_f32();
                    const float g0 = rng.gen_f32(), g1 = rng.gen_f32();
                    const float b0 = rng.gen_f32(), b1 = rng.gen_f32();
                    scene.spheres.push_back(Sphere(center, center1, 0.f, 1.f, 0.2f, std::make_shared<Lambertian>(Vec4f(r0 * r1, g0 * g1, b0 * b1, 1.f))));
                } else if (choose_mat < 0.95f) {
                    const float r = 0.5f * (1.f + rng.gen_f32());
                    const float g = 0.5f * (1.f + rng.gen_f32());
                    const float bl = 0.5f * (1.f + rng.gen_f32());
                    const float fuzz = 0.5f * rng.gen_f32();
                    scene.spheres.push_back(Sphere::from({center, 0.2f, std::make_shared<Metal>(Vec4f(r, g, bl, 1.f), fuzz)}));
                } else {
                    scene.spheres.push_back(Sphere::from({center, 0.2f, std::make_shared<Dielectric>(1.5f)}));
                }
            }
        }
    }
    scene.spheres.push_back(Sphere::from({Vec4f(0.f, 1.f, 0.f, 0.f), 1.f, std::make_shared<Dielectric>(1.5f)}));
    scene.spheres.push_back(Sphere::from({Vec4f(-4.f, 1.f, 0.f, 0.f), 1.f, std::make_shared<Lambertian>(Vec4f(0.4f, 0.2f, 0.1f, 1.f))}));
    scene.spheres.push_back(Sphere::from({Vec4f(4.f, 1.f, 0.f, 0.f), 1.f, std::make_shared<Metal>(Vec4f(0.7f, 0.6f, 0.5f, 1.f), 0.f)}));
    scene.build_bvh(BvhPartitionMethod::Middle);
    const float k = static_cast<float>(half) / 11.0f;
    CameraCreateInfo ci;
    ci.look_from = Vec4f(13.f * k, 2.f * k, 3.f * k, 0.f);
    ci.look_at = Vec4f(0.f, 0.f, 0.f, 0.f);
    ci.up = Vec4f(0.f, 1.f, 0.f, 0.f);
    ci.fov = 20.f;
    ci.aspect = static_cast<float>(nx) / static_cast<float>(ny);
    ci.aperture = 0.1f;
    ci.focus_dist = 10.f * k;
    ci.time0 = 0.f;
    ci.time1 = moving ? 1.f : 0.f;
    return {std::move(scene), Camera::from(ci)};
}
}  // namespace

std::pair<Scene, Camera> chap12_scene(size_t nx, size_t ny, RttRng &rng) { return random_spheres(nx, ny, rng, false, 11); }
std::pair<Scene, Camera> book2_chap2_scene(size_t nx, size_t ny, RttRng &rng) { return random_spheres(nx, ny, rng, true, 11); }
std::pair<Scene, Camera> random_spheres_scene(size_t nx, size_t ny, RttRng &rng, int half) { return random_spheres(nx, ny, rng, false, half); }

// ------------------------------------------------------------------ flattening
rrt_sphere_soa FlatScene::spheres() const {
    rrt_sphere_soa s{};
    s.c0x = c0x.data(); s.c0y = c0y.data(); s.c0z = c0z.data();
    s.c1x = c1x.data(); s.c1y = c1y.data(); s.c1z = c1z.data();
    s.time0 = time0.data(); s.time1 = time1.data(); s.radius = radius.data();
    s.material_id = material_id.data();
    s.n = static_cast<uint32_t>(radius.size());
    return s;
}
rrt_material_soa FlatScene::materials() const {
    rrt_material_soa m{};
    m.kind = kind.data(); m.albedo_r = albedo_r.data(); m.albedo_g = albedo_g.data(); m.albedo_b = albedo_b.data();
    m.fuzz = fuzz.data(); m.ref_idx = ref_idx.data();
    m.n = static_cast<uint32_t>(kind.size());
    return m;
}

FlatScene flatten(const Scene &scene) {
    FlatScene f;
    const size_t n = scene.spheres.size();
    for (auto *v : {&f.c0x, &f.c0y, &f.c0z, &f.c1x, &f.c1y, &f.c1z, &f.time0, &f.time1, &f.radius, &f.albedo_r, &f.albedo_g, &f.albedo_b,
                    &f.fuzz, &f.ref_idx})
        v->reserve(n);
    for (size_t i = 0; i < n; i++) {
        const Sphere &s = scene.spheres[i];
        f.c0x.push_back(s.center0().x); f.c0y.push_back(s.center0().y); f.c0z.push_back(s.center0().z);
        f.c1x.push_back(s.center1().x); f.c1y.push_back(s.center1().y); f.c1z.push_back(s.center1().z);
        f.time0.push_back(s.time0()); f.time1.push_back(s.time1()); f.radius.push_back(s.radius());
        const MaterialDesc d = s.material().desc();  // one material per sphere, as each sphere owns its Box<dyn Material>
        f.material_id.push_back(static_cast<uint32_t>(i));
        f.kind.push_back(d.kind);
        f.albedo_r.push_back(d.albedo[0]); f.albedo_g.push_back(d.albedo[1]); f.albedo_b.push_back(d.albedo[2]);
        f.fuzz.push_back(d.fuzz); f.ref_idx.push_back(d.ref_idx);
    }
    return f;
}

// ------------------------------------------------------------------ pool + tracescene
ThreadPool::ThreadPool(RttRng rng, int device, int n_gpus) {
    seed_ = rng.next_u64();
    std::vector<int> devices;
    if (device >= 0 || n_gpus == 1) devices.push_back(device);
    else {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw std::runtime_error("init_pool_with_rng: no HIP device");
        if (n_gpus > 0 && n_gpus < n) n = n_gpus;
        for (int d = 0; d < n; d++) devices.push_back(d);
    }
    for (int d : devices) {
        rrt_ctx *c = nullptr;
        if (rrt_create(d, &c) != RRT_OK) {
            const std::string msg = rrt_last_error();
            for (rrt_ctx *made : ctxs_) rrt_destroy(made);
            ctxs_.clear();
            throw std::runtime_error("init_pool_with_rng: " + msg);
        }
        ctxs_.push_back(c);
    }
}
ThreadPool::~ThreadPool() { for (rrt_ctx *c : ctxs_) rrt_destroy(c); }
ThreadPool::ThreadPool(ThreadPool &&o) noexcept : ctxs_(std::move(o.ctxs_)), seed_(o.seed_) { o.ctxs_.clear(); }

ThreadPool init_pool_with_rng(RttRng rng, int device, int n_gpus) { return ThreadPool(rng, device, n_gpus); }

std::vector<uint8_t> tracescene(size_t nx, size_t ny, size_t ns, const Scene &scene, const Camera &camera, const ThreadPool &pool,
                                std::atomic<size_t> &pdc) {
    const FlatScene flat = flatten(scene);
    const rrt_sphere_soa sp = flat.spheres();
    const rrt_material_soa mt = flat.materials();
    const std::vector<rrt_bvh_node> &nodes = scene.bvh.flatten();
    const rrt_camera cam = camera.flatten();
    for (rrt_ctx *c : pool.contexts()) {  // the scene and the camera are replicated on every GPU of the pool
        if (rrt_upload_scene(c, &sp, &mt, nodes.data(), static_cast<uint32_t>(nodes.size())) != RRT_OK)
            throw std::runtime_error(std::string("tracescene: ") + rrt_last_error());
        if (rrt_set_camera(c, &cam) != RRT_OK) throw std::runtime_error(std::string("tracescene: ") + rrt_last_error());
    }
    rrt_render_params p{};
    p.nx = static_cast<uint32_t>(nx); p.ny = static_cast<uint32_t>(ny); p.ns = static_cast<uint32_t>(ns);
    p.max_depth = 50;  // lib.rs:58
    p.seed = pool.stream_seed();
    std::vector<uint8_t> pixels(ny * nx * 3, 0);  // lib.rs:82-83
    // lib.rs:109 bumps `pdc` as pixels finish and main.rs:78-94 polls it from another thread.  The library raises `done`
    // while it renders; a helper thread forwards the increments to `pdc` so that a poller of `pdc` sees them.
    volatile uint64_t done = 0;
    uint64_t forwarded = 0;
    bool finished = false;
    std::mutex mu;
    std::condition_variable wake;
    auto forward = [&] {
        const uint64_t now = __atomic_load_n(const_cast<const uint64_t *>(&done), __ATOMIC_RELAXED);  // written by the render's thread(s)
        if (now > forwarded) { pdc.fetch_add(static_cast<size_t>(now - forwarded), std::memory_order_relaxed); forwarded = now; }
    };
    std::thread relay([&] {  // looks every 20 ms, and leaves at once when the render returns
        std::unique_lock<std::mutex> lock(mu);
        while (!finished) {
            forward();
            wake.wait_for(lock, std::chrono::milliseconds(20));
        }
    });
    // one GPU: rrt_render; several: tile shards, combined over RCCL on the first GPU (bit-identical either way)
    const int rc = rrt_render_multi(pool.contexts().data(), static_cast<uint32_t>(pool.contexts().size()), &p, pixels.data(), nullptr, nullptr, &done);
    {
        std::lock_guard<std::mutex> lock(mu);
        finished = true;
    }
    wake.notify_one();
    relay.join();
    if (rc != RRT_OK) throw std::runtime_error(std::string("tracescene: ") + rrt_last_error());
    forward();  // the rest, up to nx*ny
    return pixels;
}

}  // namespace rrt
