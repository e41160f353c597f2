// Command-line driver with the reference's flags (src/main.rs:12-108): renders the book2chap2 scene on one GPU and
// saves `o.ppm`.
//
//   rrt [-r|--resolution WxH] [-s|--samples N] [-e|--seed SEED] [-m|--random] [--scene chap11|chap12|book2chap2]
//       [-o FILE] [--device N] [--gpus N]
//
// Same defaults (1200x800, 10 samples, seed 0), same messages on stderr, same order of events: generator, scene,
// pool, render, image.  The progress line is printed by a side thread that polls the pixel counter once a second like
// the reference's progress bar (main.rs:80-94); the counter jumps to nx*ny when the render call returns.
// --scene, -o, --device and --gpus are additions; the reference hard-wires book2chap2 (main.rs:96) and o.ppm (main.rs:101).
// Without --device the render is shared by all GPUs of the node (--gpus N: the first N), as the reference uses all cores.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <random>
#include <string>
#include <thread>

#include "../../include/rrt_amd_host.h"
#include "rrt.hpp"

namespace {

// main.rs:110-128
bool parse_resolution(const char *s, size_t &w, size_t &h) {
    if (!s) return false;
    const char *x = std::strchr(s, 'x');
    if (!x || std::strchr(x + 1, 'x')) return false;
    char *end = nullptr;
    const std::string a(s, x), b(x + 1);
    if (a.empty() || b.empty()) return false;
    const unsigned long long wv = std::strtoull(a.c_str(), &end, 10);
    if (*end) return false;
    const unsigned long long hv = std::strtoull(b.c_str(), &end, 10);
    if (*end) return false;
    w = static_cast<size_t>(wv);
    h = static_cast<size_t>(hv);
    return true;
}

bool parse_u64(const char *s, unsigned long long &v) {
    if (!s || !*s || *s == '-') return false;
    char *end = nullptr;
    v = std::strtoull(s, &end, 10);
    return *end == '\0';
}

void usage() {
    std::fprintf(stderr,
                 "Ray Tracing In One Weekend\n\nUsage: rrt [OPTIONS]\n\nOptions:\n"
                 "  -r, --resolution <resolution>  output resolution in pixels [default: 1200x800]\n"
                 "  -s, --samples <samples>        samples per pixel [default: 10]\n"
                 "  -e, --seed <seed>              rng seed [default: 0]\n"
                 "  -m, --random                   use a random rng seed\n"
                 "      --scene <name>             chap11, chap12 or book2chap2 [default: book2chap2]\n"
                 "  -o <file>                      output image [default: o.ppm]\n"
                 "      --device <n>               render on this GPU alone\n"
                 "      --gpus <n>                 render on the first n GPUs of the node [default: all]\n");
}

}  // namespace

int main(int argc, char **argv) {
    const char *resolution = "1200x800", *samples = "10", *seed = "0", *scene_name = "book2chap2", *out = "o.ppm";
    bool random_seed = false;
    int device = -1, n_gpus = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&](const char *&dst) {
            if (i + 1 >= argc) { std::fprintf(stderr, "error: a value is required for '%s'\n", a.c_str()); std::exit(2); }
            dst = argv[++i];
        };
        if (a == "-r" || a == "--resolution") value(resolution);
        else if (a == "-s" || a == "--samples") value(samples);
        else if (a == "-e" || a == "--seed") value(seed);
        else if (a == "-m" || a == "--random") random_seed = true;
        else if (a == "--scene") value(scene_name);
        else if (a == "-o") value(out);
        else if (a == "--device") { const char *d = nullptr; value(d); device = std::atoi(d); }
        else if (a == "--gpus") { const char *d = nullptr; value(d); n_gpus = std::atoi(d); }
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { std::fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(); return 2; }
    }
    size_t w = 0, h = 0;
    if (!parse_resolution(resolution, w, h)) { std::fprintf(stderr, "invalid resolution\n"); return 0; }  // main.rs:49-52 returns normally
    unsigned long long spp = 0, seed_v = 0;
    if (!parse_u64(samples, spp)) { std::fprintf(stderr, "invalid sample count\n"); return 0; }
    if (!random_seed && !parse_u64(seed, seed_v)) { std::fprintf(stderr, "invalid rng seed\n"); return 0; }
    if (random_seed) {  // RttRng::from_entropy() (main.rs:62)
        std::random_device rd;
        seed_v = (static_cast<unsigned long long>(rd()) << 32) | rd();
    }
    try {
        rrt::RttRng rng = rrt::RttRng::seed_from_u64(seed_v);
        std::fprintf(stderr, "Rendering %zu x %zu image using %llu samples per pixel.\n", w, h, spp);

        std::atomic<size_t> pxcount{0};
        std::atomic<bool> stop{false};
        const size_t total = w * h;
        std::thread ui([&] {
            for (;;) {
                const size_t x = pxcount.load(std::memory_order_relaxed);
                std::fprintf(stderr, "\r%zu/%zu", x, total);
                if (x >= total || stop.load()) break;
                for (int k = 0; k < 10 && pxcount.load(std::memory_order_relaxed) < total && !stop.load(); k++)
                    std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            std::fprintf(stderr, "\n");
        });
        struct Join { std::thread &t; std::atomic<bool> &s; ~Join() { s = true; if (t.joinable()) t.join(); } } join{ui, stop};

        const std::string sn = scene_name;
        auto sc = sn == "chap11" ? rrt::chap11_scene(w, h) : sn == "chap12" ? rrt::chap12_scene(w, h, rng)
                : sn == "book2chap2" ? rrt::book2_chap2_scene(w, h, rng)
                : throw std::runtime_error("unknown scene '" + sn + "'");
        rrt::ThreadPool pool = rrt::init_pool_with_rng(rng, device, device >= 0 ? 1 : n_gpus);
        const auto t0 = std::chrono::steady_clock::now();
        const std::vector<uint8_t> pixels = rrt::tracescene(w, h, static_cast<size_t>(spp), sc.first, sc.second, pool, pxcount);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stop = true;
        ui.join();
        std::fprintf(stderr, "rendered in %.3f s\n", sec);
        if (rrt_host_write_ppm(out, pixels.data(), static_cast<uint32_t>(w), static_cast<uint32_t>(h)) != 0)
            throw std::runtime_error(std::string("cannot write ") + out);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rrt: %s\n", e.what());
        return 1;
    }
    return 0;
}
