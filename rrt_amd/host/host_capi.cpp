// C shims of include/rrt_amd_host.h over the C++ host mirror.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>

#include "../../include/rrt_amd_host.h"
#include "rrt.hpp"

struct rrt_host_scene {
    rrt::Scene scene;
    rrt::Camera camera;
    rrt::RttRng rng_after;
};

extern "C" {

rrt_host_scene *rrt_host_scene_build(int kind, uint32_t nx, uint32_t ny, uint64_t seed, int half) {
    try {
        rrt::RttRng rng = rrt::RttRng::seed_from_u64(seed);
        std::pair<rrt::Scene, rrt::Camera> sc;
        switch (kind) {
        case RRT_SCENE_CHAP11: sc = rrt::chap11_scene(nx, ny); break;
        case RRT_SCENE_CHAP12: sc = rrt::chap12_scene(nx, ny, rng); break;
        case RRT_SCENE_BOOK2CHAP2: sc = rrt::book2_chap2_scene(nx, ny, rng); break;
        case RRT_SCENE_RANDOM_SPHERES: if (half < 1) return nullptr; sc = rrt::random_spheres_scene(nx, ny, rng, half); break;
        default: return nullptr;
        }
        return new rrt_host_scene{std::move(sc.first), std::move(sc.second), rng};
    } catch (const std::exception &) {
        return nullptr;
    }
}

void rrt_host_scene_free(rrt_host_scene *s) { delete s; }
uint32_t rrt_host_scene_num_spheres(const rrt_host_scene *s) { return static_cast<uint32_t>(s->scene.spheres.size()); }
uint32_t rrt_host_scene_num_nodes(const rrt_host_scene *s) { return static_cast<uint32_t>(s->scene.bvh.flatten().size()); }
void rrt_host_scene_rng_after(const rrt_host_scene *s, uint32_t state[4]) { std::memcpy(state, s->rng_after.state(), 16); }

void rrt_host_scene_export(const rrt_host_scene *s, float *spheres9, float *mats6, uint32_t *nodes8, rrt_camera *camera) {
    const rrt::FlatScene f = rrt::flatten(s->scene);
    const size_t n = f.radius.size();
    for (size_t i = 0; i < n; i++) {
        if (spheres9) {
            float *o = spheres9 + i * 9;
            o[0] = f.c0x[i]; o[1] = f.c0y[i]; o[2] = f.c0z[i]; o[3] = f.c1x[i]; o[4] = f.c1y[i]; o[5] = f.c1z[i];
            o[6] = f.time0[i]; o[7] = f.time1[i]; o[8] = f.radius[i];
        }
        if (mats6) {
            float *m = mats6 + i * 6;
            const uint32_t id = f.material_id[i];
            m[0] = static_cast<float>(f.kind[id]); m[1] = f.albedo_r[id]; m[2] = f.albedo_g[id]; m[3] = f.albedo_b[id];
            m[4] = f.fuzz[id]; m[5] = f.ref_idx[id];
        }
    }
    if (nodes8) {
        const auto &nodes = s->scene.bvh.flatten();
        static_assert(sizeof(rrt_bvh_node) == 32, "interchange node is 32 bytes");
        std::memcpy(nodes8, nodes.data(), nodes.size() * sizeof(rrt_bvh_node));
    }
    if (camera) *camera = s->camera.flatten();
}

int rrt_host_scene_upload(const rrt_host_scene *s, rrt_ctx *ctx) {
    const rrt::FlatScene f = rrt::flatten(s->scene);
    const rrt_sphere_soa sp = f.spheres();
    const rrt_material_soa mt = f.materials();
    const auto &nodes = s->scene.bvh.flatten();
    int rc = rrt_upload_scene(ctx, &sp, &mt, nodes.data(), static_cast<uint32_t>(nodes.size()));
    if (rc != RRT_OK) return rc;
    const rrt_camera cam = s->camera.flatten();
    return rrt_set_camera(ctx, &cam);
}

int rrt_host_tracescene(const rrt_host_scene *s, uint32_t nx, uint32_t ny, uint32_t ns, int device, uint8_t *pixels, uint64_t *pdc,
                        uint64_t *stream_seed) {
    try {
        rrt::ThreadPool pool = rrt::init_pool_with_rng(s->rng_after, device);
        std::atomic<size_t> count{0};
        const std::vector<uint8_t> px = rrt::tracescene(nx, ny, ns, s->scene, s->camera, pool, count);
        std::memcpy(pixels, px.data(), px.size());
        if (pdc) *pdc = count.load();
        if (stream_seed) *stream_seed = pool.stream_seed();
        return RRT_OK;
    } catch (const std::exception &) {
        return RRT_ERR_HIP;  // message stays in rrt_last_error()
    }
}

void rrt_host_camera_from(const float *ci, rrt_camera *out) {
    rrt::CameraCreateInfo c;
    c.look_from = rrt::Vec4f(ci[0], ci[1], ci[2], 0.f);
    c.look_at = rrt::Vec4f(ci[3], ci[4], ci[5], 0.f);
    c.up = rrt::Vec4f(ci[6], ci[7], ci[8], 0.f);
    c.fov = ci[9]; c.aspect = ci[10]; c.aperture = ci[11]; c.focus_dist = ci[12]; c.time0 = ci[13]; c.time1 = ci[14];
    *out = rrt::Camera::from(c).flatten();
}

void rrt_host_rng_seed(uint64_t seed, uint32_t state[4]) { std::memcpy(state, rrt::RttRng::seed_from_u64(seed).state(), 16); }
uint32_t rrt_host_rng_next_u32(uint32_t state[4]) {
    rrt::RttRng r = rrt::RttRng::from_state(state);
    const uint32_t v = r.next_u32();
    std::memcpy(state, r.state(), 16);
    return v;
}
float rrt_host_rng_gen_f32(uint32_t state[4]) {
    rrt::RttRng r = rrt::RttRng::from_state(state);
    const float v = r.gen_f32();
    std::memcpy(state, r.state(), 16);
    return v;
}
void rrt_host_rng_jump(uint32_t state[4]) {
    rrt::RttRng r = rrt::RttRng::from_state(state);
    r.jump();
    std::memcpy(state, r.state(), 16);
}
void rrt_host_uniform_new_inclusive(float low, float high, float *out_low, float *out_scale) {
    const rrt::Uniform u = rrt::Uniform::new_inclusive(low, high);
    *out_low = u.low; *out_scale = u.scale;
}
float rrt_host_uniform_sample(float low, float scale, uint32_t state[4]) {
    rrt::RttRng r = rrt::RttRng::from_state(state);
    rrt::Uniform u; u.low = low; u.scale = scale;
    const float v = u.sample(r);
    std::memcpy(state, r.state(), 16);
    return v;
}

int rrt_host_write_ppm(const char *path, const uint8_t *rgb8, uint32_t nx, uint32_t ny) {
    if (!path || !rgb8) return RRT_ERR_INVALID_ARGUMENT;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RRT_ERR_INVALID_ARGUMENT;
    std::fprintf(f, "P6\n%u %u\n255\n", nx, ny);
    const size_t n = static_cast<size_t>(nx) * ny * 3;
    const size_t w = std::fwrite(rgb8, 1, n, f);
    std::fclose(f);
    return w == n ? RRT_OK : RRT_ERR_INVALID_ARGUMENT;
}

}  // extern "C"
