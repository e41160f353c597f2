// C++ mirror of the reference's public surface for the render path, over the C ABI of
// include/rrt_amd.h.  Names follow the Rust crate one for one so that a caller of
//
//     let (scene, camera) = chap12_scene(nx, ny, &mut rng);        (src/chap12.rs:9)
//     let pool = init_pool_with_rng(rng);                           (src/threadpool.rs:11)
//     let pixels = tracescene(nx, ny, ns, &scene, &camera, &pool, &pdc);   (src/lib.rs:73)
//
// reads the same here.  The host builds scenes, cameras and the BVH exactly as the reference does
// (same draw order from the same generator, same arithmetic); ray generation, traversal,
// intersection and shading run on the GPU.  There is no CPU render path in this library.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rrt_amd.h"

namespace rrt {

// ---- types.rs:6-37 (ultraviolet::Vec4 semantics) ----
struct Vec4f {
    float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
    Vec4f() = default;
    Vec4f(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    static Vec4f broadcast(float s) { return Vec4f(s, s, s, s); }
    static Vec4f zero() { return broadcast(0.f); }
    static Vec4f one() { return broadcast(1.f); }
    float operator[](int i) const { return i == 0 ? x : i == 1 ? y : i == 2 ? z : w; }
    float dot(const Vec4f &o) const;
    float mag() const;
    Vec4f normalized() const;
    Vec4f min_by_component(const Vec4f &o) const;
    Vec4f max_by_component(const Vec4f &o) const;
};
Vec4f operator+(const Vec4f &a, const Vec4f &b);
Vec4f operator-(const Vec4f &a, const Vec4f &b);
Vec4f operator*(float s, const Vec4f &a);

enum class Axis : uint32_t { X = 0, Y = 1, Z = 2 };

// ---- rng.rs:14 — rand_xoshiro::Xoshiro128Plus with rand's SeedableRng / Rng conversions ----
class RttRng {
public:
    static RttRng seed_from_u64(uint64_t seed);
    static RttRng from_state(const uint32_t s[4]);
    uint32_t next_u32();
    uint64_t next_u64();
    float gen_f32();  // rng.gen::<f32>()
    void jump();
    const uint32_t *state() const { return s_; }
private:
    uint32_t s_[4] = {0, 0, 0, 0};
};

// rand::distributions::Uniform<f32>
struct Uniform {
    float low = 0.f, scale = 0.f;
    static Uniform new_inclusive(float low, float high);
    float sample(RttRng &rng) const;
};

// ---- materials.rs ----
struct MaterialDesc { uint32_t kind; float albedo[3]; float fuzz; float ref_idx; };
struct Material {
    virtual ~Material() = default;
    virtual MaterialDesc desc() const = 0;  // what the GPU needs of a `dyn Material`
};
struct Lambertian : Material { Vec4f albedo; explicit Lambertian(Vec4f a) : albedo(a) {} MaterialDesc desc() const override; };
struct Metal : Material { Vec4f albedo; float fuzz; Metal(Vec4f a, float f) : albedo(a), fuzz(f) {} MaterialDesc desc() const override; };
struct Dielectric : Material { float ref_idx; explicit Dielectric(float r) : ref_idx(r) {} MaterialDesc desc() const override; };

// ---- acceleration.rs:10-94 ----
struct Aabb {
    Vec4f minimum = Vec4f::broadcast(3.402823466e+38f), maximum = Vec4f::broadcast(-3.402823466e+38f);
    Aabb union_(const Aabb &b) const;
    Aabb union_p(const Vec4f &p) const;
    Vec4f centroid() const;
    Vec4f diagonal() const;
    Axis max_extent() const;
};

// ---- hittable.rs:5-8 ----
struct Hittable {
    virtual ~Hittable() = default;
    virtual Aabb aabb() const = 0;
};

// ---- sphere.rs ----
struct StationarySphere { Vec4f center; float radius; std::shared_ptr<Material> material; };

class Sphere : public Hittable {
public:
    Sphere(Vec4f center0, Vec4f center1, float time0, float time1, float radius, std::shared_ptr<Material> material);
    static Sphere from(StationarySphere s);
    Aabb aabb() const override { return aabb_; }
    // flatten(): the fields the Rust struct keeps private
    const Vec4f &center0() const { return center0_; }
    const Vec4f &center1() const { return center1_; }
    float time0() const { return time0_; }
    float time1() const { return time1_; }
    float radius() const { return radius_; }
    const Material &material() const { return *material_; }
private:
    Vec4f center0_, center1_;
    float time0_, time1_, radius_;
    std::shared_ptr<Material> material_;
    Aabb aabb_;
};

// ---- acceleration.rs:96-217 ----
enum class BvhPartitionMethod { Middle, EqualCount };

class Bvh {
public:
    void build(const std::vector<const Hittable *> &objects, BvhPartitionMethod method);
    const std::vector<rrt_bvh_node> &flatten() const { return nodes_; }  // push order of build_r
    void adopt(std::vector<rrt_bvh_node> nodes) { nodes_ = std::move(nodes); }  // the same vector from another builder (rrt_build_bvh_method)
private:
    struct Info { uint32_t index; Aabb aabb; Vec4f centroid; };
    int64_t build_r(Info *infos, size_t n, BvhPartitionMethod method);
    std::vector<rrt_bvh_node> nodes_;
};

// ---- scene.rs:5-36 ----
struct Scene {
    std::vector<Sphere> spheres;
    Bvh bvh;
    std::vector<const Hittable *> objects() const;
    // scene.rs:32-35.  Scenes of kGpuBuildMinSpheres spheres or more are built by rrt_build_bvh_method on the current HIP
    // device when there is one (the same tree, node for node; 50 ms instead of 1.4 s for a million spheres); set
    // RRT_HOST_BVH=1 to keep every build on the host.
    void build_bvh(BvhPartitionMethod method);
    static constexpr size_t kGpuBuildMinSpheres = 1u << 15;
};

// ---- camera.rs ----
struct CameraCreateInfo {
    Vec4f look_from, look_at, up;
    float fov, aspect, aperture, focus_dist, time0, time1;
};
class Camera {
public:
    static Camera from(const CameraCreateInfo &ci);
    rrt_camera flatten() const;
private:
    Vec4f lower_left_corner_, horizontal_, vertical_, origin_, u_, v_, w_;
    float lens_radius_ = 0.f;
    Uniform time_dist_;
};

// ---- scene builders ----
std::pair<Scene, Camera> chap11_scene(size_t nx, size_t ny);                       // chap11.rs:8
std::pair<Scene, Camera> chap12_scene(size_t nx, size_t ny, RttRng &rng);         // chap12.rs:9
std::pair<Scene, Camera> book2_chap2_scene(size_t nx, size_t ny, RttRng &rng);    // book2chap2.rs:10
// Not in the reference: the chap12 generator over cells -half..half (half = 11 is chap12 itself),
// camera scaled by half/11.  half = 500 gives the one-million-sphere traversal stress scene.
std::pair<Scene, Camera> random_spheres_scene(size_t nx, size_t ny, RttRng &rng, int half);

// ---- threadpool.rs:11-27: what runs the per-pixel work.  Here: the GPUs of the node (or one of them) plus the base
//      generator from which every (pixel, sample) stream is derived. ----
class ThreadPool {
public:
    // device >= 0: that GPU alone.  device < 0: the first n_gpus GPUs of the node, all of them when n_gpus is 0.
    ThreadPool(RttRng rng, int device, int n_gpus = 1);
    ~ThreadPool();
    ThreadPool(ThreadPool &&o) noexcept;
    ThreadPool(const ThreadPool &) = delete;
    rrt_ctx *ctx() const { return ctxs_.empty() ? nullptr : ctxs_[0]; }
    const std::vector<rrt_ctx *> &contexts() const { return ctxs_; }
    uint64_t stream_seed() const { return seed_; }
private:
    std::vector<rrt_ctx *> ctxs_;
    uint64_t seed_ = 0;
};
// device < 0 and n_gpus == 1: the current GPU, as a one-GPU machine would
ThreadPool init_pool_with_rng(RttRng rng, int device = -1, int n_gpus = 1);

// Flattened scene in the interchange form of include/rrt_amd.h; owns the planes it points at.
struct FlatScene {
    std::vector<float> c0x, c0y, c0z, c1x, c1y, c1z, time0, time1, radius;
    std::vector<uint32_t> material_id, kind;
    std::vector<float> albedo_r, albedo_g, albedo_b, fuzz, ref_idx;
    rrt_sphere_soa spheres() const;
    rrt_material_soa materials() const;
};
FlatScene flatten(const Scene &scene);

// lib.rs:73-113.  Throws std::runtime_error with rrt_last_error() on failure (the reference panics).
std::vector<uint8_t> tracescene(size_t nx, size_t ny, size_t ns, const Scene &scene, const Camera &camera, const ThreadPool &pool,
                                std::atomic<size_t> &pdc);

}  // namespace rrt
