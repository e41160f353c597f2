/*
 * rrt_amd_host.h — C shims over the C++ host mirror (rrt_amd/host/rrt.hpp) so that the reference's
 * host-side steps can be driven from languages without a C++ ABI (the Python test-suite, bench.py).
 * Each function stands for the Rust call named next to it.
 */
#ifndef RRT_AMD_HOST_H
#define RRT_AMD_HOST_H

#include "rrt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrt_host_scene rrt_host_scene;

enum { RRT_SCENE_CHAP11 = 0, RRT_SCENE_CHAP12 = 1, RRT_SCENE_BOOK2CHAP2 = 2, RRT_SCENE_RANDOM_SPHERES = 3 };

/* RttRng::seed_from_u64(seed) followed by chap11_scene(nx, ny) (src/chap11.rs:8), chap12_scene(nx, ny, &mut rng)
 * (src/chap12.rs:9), book2_chap2_scene (src/book2chap2.rs:10) or the chap12 generator over cells -half..half.
 * Keeps the scene, the camera and the generator state left behind (what main.rs:96-97 hands to the pool). */
RRT_API rrt_host_scene *rrt_host_scene_build(int kind, uint32_t nx, uint32_t ny, uint64_t seed, int half);
RRT_API void rrt_host_scene_free(rrt_host_scene *s);
RRT_API uint32_t rrt_host_scene_num_spheres(const rrt_host_scene *s);
RRT_API uint32_t rrt_host_scene_num_nodes(const rrt_host_scene *s);
RRT_API void rrt_host_scene_rng_after(const rrt_host_scene *s, uint32_t state[4]);

/* Flattened views: spheres n x 9 floats {c0 xyz, c1 xyz, time0, time1, radius}; mats n x 6 floats
 * {kind, r, g, b, fuzz, ref_idx}; nodes n_nodes x 8 words (rrt_bvh_node); camera: rrt_camera. */
RRT_API void rrt_host_scene_export(const rrt_host_scene *s, float *spheres9, float *mats6, uint32_t *nodes8, rrt_camera *camera);

/* Scene::flatten + rrt_upload_scene + Camera::flatten + rrt_set_camera on ctx. */
RRT_API int rrt_host_scene_upload(const rrt_host_scene *s, rrt_ctx *ctx);

/* init_pool_with_rng(rng) (src/threadpool.rs:11) + tracescene(nx, ny, ns, &scene, &camera, &pool, &pdc) (src/lib.rs:73):
 * pixels nx*ny*3 bytes; *pdc receives the finished-pixel count; *stream_seed the 64-bit seed the pool derived. */
RRT_API int rrt_host_tracescene(const rrt_host_scene *s, uint32_t nx, uint32_t ny, uint32_t ns, int device, uint8_t *pixels,
                                uint64_t *pdc, uint64_t *stream_seed);

/* Camera::from(CameraCreateInfo) (src/camera.rs:31-54): ci = look_from xyz, look_at xyz, up xyz, fov, aspect, aperture,
 * focus_dist, time0, time1 (15 floats). */
RRT_API void rrt_host_camera_from(const float *ci15, rrt_camera *out);

/* Generator known-answer hooks: RttRng::seed_from_u64, next_u32, gen::<f32>(), jump(), Uniform::new_inclusive + sample. */
RRT_API void rrt_host_rng_seed(uint64_t seed, uint32_t state[4]);
RRT_API uint32_t rrt_host_rng_next_u32(uint32_t state[4]);
RRT_API float rrt_host_rng_gen_f32(uint32_t state[4]);
RRT_API void rrt_host_rng_jump(uint32_t state[4]);
RRT_API void rrt_host_uniform_new_inclusive(float low, float high, float *out_low, float *out_scale);
RRT_API float rrt_host_uniform_sample(float low, float scale, uint32_t state[4]);

/* Binary PPM (P6) of an RGB8 buffer — the format image::save_buffer("o.ppm", ..) writes (src/main.rs:100-107). */
RRT_API int rrt_host_write_ppm(const char *path, const uint8_t *rgb8, uint32_t nx, uint32_t ny);

#ifdef __cplusplus
}
#endif
#endif /* RRT_AMD_HOST_H */
