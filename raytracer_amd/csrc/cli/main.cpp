// `raytracer` — command line of the reference (clap derive, main.rs:23-66) over the C ABI in include/rt.h.
// Same flag names, defaults and conflict rule; same stderr "Done in …" line (main.rs:190) and RGB8 PNG
// (main.rs:192-200).  Extra flags: --gpus N, --reduce rccl|host, --progressive SECS, --kernel persistent|wavefront|pool,
// --sampler direct|reference, --dump-primary FILE, --json.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../../include/rt.h"

namespace {

struct Opt {
    size_t num_samples = 4;
    std::string output = "output.png";
    size_t threads = 0;
    size_t width = 400;
    bool has_height = false, has_aspect = false, has_seed = false, has_scene = false, has_example = false;
    size_t height = 0;
    double aspect_ratio = 1.5;
    uint64_t seed = 0;
    std::string scene, example, dump_primary;
    bool no_progress = false, no_bvh = false, json = false, device_build = false;
    int gpus = 1;
    int kernel = RT_KERNEL_PERSISTENT;
    int sampler = RT_SAMPLER_DIRECT;
    bool reduce_rccl = true;    // --gpus N: sum the per-GPU framebuffers with one RCCL reduce (else over PCIe on the host)
    double progressive = -1.0;  // >= 0: write the running image every so many seconds
    bool keep_partials = false;  // with --progressive: every partial image is also kept, as OUTPUT.partial<k>.png
};

[[noreturn]] void usage_error(const std::string& m) {
    std::fprintf(stderr, "error: %s\n\nUsage: raytracer [OPTIONS]\n\nFor more information, try '--help'.\n", m.c_str());
    std::exit(2);
}

void help() {
    std::puts(
        "A simple ray tracer implementation (MI355X-native render path).\n\n"
        "Usage: raytracer [OPTIONS]\n\nOptions:\n"
        "  -n, --num-samples <NUM_SAMPLES>    Number of samples to take per pixel [default: 4]\n"
        "  -o, --output <OUTPUT>              Destination of the output image (png) [default: output.png]\n"
        "  -t, --threads <THREADS>            Accepted for compatibility; the render runs on the GPU\n"
        "      --width <WIDTH>                Output image width [default: 400]\n"
        "      --height <HEIGHT>              Output image height\n"
        "      --aspect-ratio <ASPECT_RATIO>  Output image aspect ratio [default: 1.5]\n"
        "      --seed <SEED>                  Seed to use for RNG\n"
        "      --scene <SCENE>                A scene file to load from configuration\n"
        "      --example <EXAMPLE>            An example scene from the book: one-weekend (default without --scene),\n"
        "                                     random-spheres, two-spheres, two-perlin, earth, cornell, cornell-smoke,\n"
        "                                     final-scene; the random ones are drawn from --seed\n"
        "      --no-progress                  Disables the progress block (shown for long renders, once a second)\n"
        "      --no-bvh                       Disables the use of the bvh acceleration structure\n"
        "      --device-build                 Builds the tree the render kernels traverse on the GPU (default: host)\n"
        "      --gpus <N>                     Shard the strata over N GPUs [default: 1]\n"
        "      --reduce <rccl|host>           How the GPUs' framebuffers are summed: one RCCL reduce over xGMI\n"
        "                                     [default], or copies over PCIe added on the host\n"
        "      --progressive <SECS>           Write the image so far to --output every SECS seconds while rendering\n"
        "      --keep-partials                With --progressive: also keep every partial image, as OUTPUT.partial<k>.png\n"
        "      --kernel <persistent|wavefront|pool>\n"
        "      --sampler <direct|reference>   Unit-sphere/disk sampling: loop-free maps or the reference's rejection loops\n"
        "      --dump-primary <FILE>          Write the f64 primary-hit AOV (int32 ids, then f64 t) and exit\n"
        "      --json                         Print a JSON metrics line on stdout\n"
        "  -h, --help                         Print help");
}

Opt parse(int argc, char** argv) {
    Opt o;
    auto need = [&](int& i, const char* flag) -> std::string {
        std::string a = argv[i];
        size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) return a.substr(eq + 1);
        if (i + 1 >= argc) usage_error(std::string("a value is required for '") + flag + "' but none was supplied");
        return argv[++i];
    };
    auto to_size = [&](const std::string& s, const char* flag) -> size_t {
        char* end = nullptr;
        if (s.empty() || s[0] == '-') usage_error("invalid value '" + s + "' for '" + flag + "'");
        unsigned long long v = std::strtoull(s.c_str(), &end, 10);
        if (*end) usage_error("invalid value '" + s + "' for '" + flag + "'");
        return (size_t)v;
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        std::string key = a.substr(0, a.find('='));
        if (key == "-h" || key == "--help") {
            help();
            std::exit(0);
        } else if (key == "-n" || key == "--num-samples") o.num_samples = to_size(need(i, "--num-samples"), "--num-samples");
        else if (key == "-o" || key == "--output") o.output = need(i, "--output");
        else if (key == "-t" || key == "--threads") o.threads = to_size(need(i, "--threads"), "--threads");
        else if (key == "--width") o.width = to_size(need(i, "--width"), "--width");
        else if (key == "--height") o.height = to_size(need(i, "--height"), "--height"), o.has_height = true;
        else if (key == "--aspect-ratio") {
            std::string v = need(i, "--aspect-ratio");
            char* end = nullptr;
            o.aspect_ratio = std::strtod(v.c_str(), &end);
            if (*end || v.empty()) usage_error("invalid value '" + v + "' for '--aspect-ratio'");
            o.has_aspect = true;
        } else if (key == "--seed") o.seed = to_size(need(i, "--seed"), "--seed"), o.has_seed = true;
        else if (key == "--scene") o.scene = need(i, "--scene"), o.has_scene = true;
        else if (key == "--example") o.example = need(i, "--example"), o.has_example = true;
        else if (key == "--no-progress") o.no_progress = true;
        else if (key == "--no-bvh") o.no_bvh = true;
        else if (key == "--device-build") o.device_build = true;
        else if (key == "--json") o.json = true;
        else if (key == "--gpus") o.gpus = (int)to_size(need(i, "--gpus"), "--gpus");
        else if (key == "--dump-primary") o.dump_primary = need(i, "--dump-primary");
        else if (key == "--reduce") {
            std::string v = need(i, "--reduce");
            if (v == "rccl") o.reduce_rccl = true;
            else if (v == "host") o.reduce_rccl = false;
            else usage_error("invalid value '" + v + "' for '--reduce'");
        } else if (key == "--keep-partials") {
            o.keep_partials = true;
        } else if (key == "--progressive") {
            std::string v = need(i, "--progressive");
            char* end = nullptr;
            o.progressive = std::strtod(v.c_str(), &end);
            if (*end || v.empty() || !(o.progressive >= 0.0)) usage_error("invalid value '" + v + "' for '--progressive'");
        }
        else if (key == "--kernel") {
            std::string v = need(i, "--kernel");
            if (v == "persistent") o.kernel = RT_KERNEL_PERSISTENT;
            else if (v == "wavefront") o.kernel = RT_KERNEL_WAVEFRONT;
            else if (v == "pool") o.kernel = RT_KERNEL_POOL;
            else usage_error("invalid value '" + v + "' for '--kernel'");
        } else if (key == "--sampler") {
            std::string v = need(i, "--sampler");
            if (v == "direct") o.sampler = RT_SAMPLER_DIRECT;
            else if (v == "reference") o.sampler = RT_SAMPLER_REFERENCE;
            else usage_error("invalid value '" + v + "' for '--sampler'");
        } else usage_error("unexpected argument '" + a + "' found");
    }
    if (o.has_height && o.has_aspect)  // #[arg(long, conflicts_with = "height")], main.rs:44
        usage_error("the argument '--aspect-ratio <ASPECT_RATIO>' cannot be used with '--height <HEIGHT>'");
    return o;
}

int die(const char* ctx) {
    std::fprintf(stderr, "Error: %s%s%s\n", ctx, *ctx ? ": " : "", rt_last_error());
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    Opt o = parse(argc, argv);
    // TracerOpt::height / aspect_ratio, main.rs:77-95
    const size_t width = o.width;
    const size_t height = o.has_height ? o.height : (size_t)((double)width / o.aspect_ratio);
    const double aspect = o.has_height ? (double)width / (double)o.height : o.aspect_ratio;
    const uint32_t max_depth = 40;  // main.rs:118
    if (o.gpus < 1) o.gpus = 1;
    const auto start = std::chrono::steady_clock::now();
    rt_set_verbose(1);

    rt_scene_desc* desc = nullptr;
    // TracerOpt::scene, main.rs:96-107: a scene file if given, else the example (default one-weekend)
    const uint64_t scene_seed = o.has_seed ? o.seed : 0x5eed5eedull;
    const int bvh_mode = o.no_bvh ? 0 : (o.device_build ? 2 : 1);
    if (o.has_scene) {
        if (rt_load_scene(o.scene.c_str(), aspect, bvh_mode, &desc)) return die("");
    } else if (rt_example_scene(o.has_example ? o.example.c_str() : "one-weekend", aspect, scene_seed, bvh_mode,
                                &desc)) {
        return die("");
    }
    std::vector<rt_scene*> scenes((size_t)o.gpus, nullptr);
    for (int g = 0; g < o.gpus; ++g)
        if (rt_scene_create(desc, g, &scenes[(size_t)g])) return die("create device scene");

    if (!o.dump_primary.empty()) {
        std::vector<int32_t> ids(width * height);
        std::vector<double> t(width * height);
        if (rt_primary_hits(scenes[0], (uint32_t)width, (uint32_t)height, 64, ids.data(), t.data())) return die("primary hits");
        FILE* f = std::fopen(o.dump_primary.c_str(), "wb");
        if (!f) {
            std::perror(o.dump_primary.c_str());
            return 1;
        }
        std::fwrite(ids.data(), sizeof(int32_t), ids.size(), f);
        std::fwrite(t.data(), sizeof(double), t.size(), f);
        std::fclose(f);
        return 0;
    }

    // main.rs:142: stratification needs a perfect square
    const uint32_t sn = (uint32_t)std::floor(std::sqrt((double)o.num_samples));
    if (sn == 0 || width == 0 || height == 0) {
        std::fprintf(stderr, "Error: nothing to render (num-samples, width and height must be positive)\n");
        return 1;
    }
    // The reference seeds from OS entropy without --seed; any fixed default is as good, and reproducible.
    const uint64_t seed = o.has_seed ? o.seed : 0x5eed5eedull;
    const auto render_start = std::chrono::steady_clock::now();

    // Long jobs are rendered in batches of strata so that the progress block of progress.rs:63-97 (elapsed,
    // remaining, samples/sec, remaining pixels; once a second) can be shown and, with --progressive, the image so
    // far written; short ones go in one launch.
    const double total_samples = (double)width * (double)height * (double)sn * (double)sn;
    const uint32_t n_strata = sn * sn;
    const bool batched = o.progressive >= 0.0 || (!o.no_progress && total_samples > 8e9);
    // Batch b renders the strata k = b (mod step), step = batches x GPUs, and stratum k sits at (k / sn, k % sn) of
    // the pixel's sn x sn grid (main.rs:165-171): a step that shares a factor with sn confines a batch to a few
    // columns of that grid — with sn = 32 and 32 batches to ONE column, a strip a 32nd of the pixel high, and the
    // partial images (--progressive) would be estimates of a shifted image.  Up to 32 batches, then: the count whose
    // step has the smallest common factor with sn (1 where there is one), not congruent to 0 or +-1 modulo sn (the
    // grid's rows and diagonals), the largest such — the batch's strata then form a lattice spread over the pixel.
    uint32_t batches = 1u;
    if (batched) {
        const uint32_t most = std::min<uint32_t>(32u, std::max(1u, n_strata / (uint32_t)o.gpus));
        auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b, b = t; } return a; };
        uint32_t best_score = 0xffffffffu;
        for (uint32_t c = most; c >= 1u && c + 8u > most; --c) {
            const uint32_t step = c * (uint32_t)o.gpus, r = step % sn;
            const uint32_t score = 4u * gcd(step, sn) + ((sn > 2u && (r == 0u || r == 1u || r == sn - 1u)) ? 2u : 0u);
            if (score < best_score) best_score = score, batches = c;
        }
    }
    const size_t n_values = 3 * width * height;
    std::vector<float> sum(n_values, 0.f), part(n_values);
    // every GPU renders its strata of the batch into a framebuffer in its own memory
    std::vector<float*> d_fb((size_t)o.gpus, nullptr);
    for (int g = 0; g < o.gpus; ++g)
        if (hipSetDevice(g) != hipSuccess || hipMalloc((void**)&d_fb[(size_t)g], n_values * sizeof(float)) != hipSuccess) {
            std::fprintf(stderr, "Error: could not allocate the framebuffer on GPU %d\n", g);
            return 1;
        }
    rt_group* group = nullptr;
    if (o.gpus > 1 && o.reduce_rccl) {
        std::vector<int> devices((size_t)o.gpus);
        for (int g = 0; g < o.gpus; ++g) devices[(size_t)g] = g;
        if (rt_group_create(devices.data(), o.gpus, &group)) return die("");
    }
    std::vector<rt_stats> stats((size_t)o.gpus);
    double kernel_ms_total = 0;
    uint64_t samples_done = 0;  // per pixel
    auto last_report = start, last_image = start;
    bool reported = false;
    uint32_t partial_images = 0;
    for (uint32_t b = 0; b < batches; ++b) {
        std::vector<int> rc((size_t)o.gpus, 0);
        std::vector<std::string> errs((size_t)o.gpus);
        std::vector<std::thread> pool;
        for (int g = 0; g < o.gpus; ++g) {
            pool.emplace_back([&, g]() {
                rt_render_params p{};
                p.width = (uint32_t)width, p.height = (uint32_t)height, p.sn = sn, p.max_depth = max_depth;
                p.seed = seed, p.kernel = o.kernel, p.sampler = o.sampler;
                p.stratum_begin = b * (uint32_t)o.gpus + (uint32_t)g, p.stratum_step = batches * (uint32_t)o.gpus;
                rc[(size_t)g] = rt_render_device(scenes[(size_t)g], &p, d_fb[(size_t)g], nullptr, &stats[(size_t)g]);
                if (rc[(size_t)g]) errs[(size_t)g] = rt_last_error();
            });
        }
        for (auto& th : pool) th.join();
        double batch_ms = 0;
        for (int g = 0; g < o.gpus; ++g) {
            if (rc[(size_t)g]) {
                std::fprintf(stderr, "Error: render on GPU %d: %s\n", g, errs[(size_t)g].c_str());
                return 1;
            }
            batch_ms = std::fmax(batch_ms, stats[(size_t)g].kernel_ms);
            samples_done += stats[(size_t)g].samples / ((uint64_t)width * height);
        }
        kernel_ms_total += batch_ms;
        // the exchange step: the GPUs' sums onto GPU 0 with one RCCL reduce, or one copy per GPU over PCIe
        if (group && rt_group_reduce_sum(group, d_fb.data(), n_values, 0, nullptr)) return die("");
        for (int g = 0; g < (group ? 1 : o.gpus); ++g) {
            if (hipSetDevice(g) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(part.data(), d_fb[(size_t)g], n_values * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
                std::fprintf(stderr, "Error: could not read the framebuffer of GPU %d\n", g);
                return 1;
            }
            for (size_t i = 0; i < n_values; ++i) sum[i] += part[i];
        }
        const auto now = std::chrono::steady_clock::now();
        if (o.progressive >= 0.0 && b + 1 < batches && std::chrono::duration<double>(now - last_image).count() >= o.progressive) {
            // the image so far: the strata rendered up to here are a stratified subset of every pixel's
            last_image = now;
            std::vector<uint8_t> rgb(n_values);
            if (rt_tonemap_rgb8(sum.data(), (uint32_t)width, (uint32_t)height, samples_done, rgb.data())) return die("tonemap");
            const std::string tmp = o.output + ".part";
            if (rt_write_png(tmp.c_str(), rgb.data(), (uint32_t)width, (uint32_t)height) || std::rename(tmp.c_str(), o.output.c_str()))
                return die("could not write image");
            ++partial_images;
            if (o.keep_partials &&
                rt_write_png((o.output + ".partial" + std::to_string(partial_images) + ".png").c_str(), rgb.data(), (uint32_t)width, (uint32_t)height))
                return die("could not write image");
        }
        if (!o.no_progress && batches > 1 && b + 1 < batches && std::chrono::duration<double>(now - last_report).count() >= 1.0) {
            last_report = now;
            const double elapsed = std::chrono::duration<double>(now - start).count();
            const double done = (double)(b + 1) / (double)batches;
            const double render_elapsed = std::chrono::duration<double>(now - render_start).count();
            const double remaining = render_elapsed * (1.0 - done) / done;
            auto fmt = [](double sec, char* buf, size_t n) {  // format_duration, progress.rs:100-110
                const unsigned long s = (unsigned long)sec, hh = s / 3600, mm = (s - hh * 3600) / 60, ss = s - mm * 60 - hh * 3600;
                if (hh > 0) std::snprintf(buf, n, "%02lu:%02lu:%02lu", hh, mm, ss);
                else std::snprintf(buf, n, "%02lu:%02lu", mm, ss);
            };
            char e[32], r[32];
            fmt(elapsed, e, sizeof e), fmt(remaining, r, sizeof r);
            if (!reported) std::fprintf(stderr, "\n\n\n\n");
            reported = true;
            std::fprintf(stderr, "\x1b[4A    Elapsed Time: %s    \n  Remaining Time: %s    \n   Samples / sec: %.0f    \nRemaining Pixels: %.0f    \n",
                         e, r, total_samples * done / render_elapsed, (double)width * (double)height * (1.0 - done));
        }
    }
    if (o.progressive >= 0.0) std::fprintf(stderr, "wrote %u partial image%s to %s\n", partial_images, partial_images == 1 ? "" : "s", o.output.c_str());
    rt_group_destroy(group);
    for (int g = 0; g < o.gpus; ++g)
        if (hipSetDevice(g) == hipSuccess) (void)hipFree(d_fb[(size_t)g]);

    std::vector<uint8_t> rgb(3 * width * height);
    if (rt_tonemap_rgb8(sum.data(), (uint32_t)width, (uint32_t)height, (uint64_t)sn * sn, rgb.data())) return die("tonemap");

    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    const double rays = (double)width * (double)height * (double)sn * (double)sn;  // primary samples, main.rs:188
    std::fprintf(stderr, "\nDone in %.2fs (%.0f rays/s)\n", elapsed, rays / elapsed);  // main.rs:190
    if (rt_write_png(o.output.c_str(), rgb.data(), (uint32_t)width, (uint32_t)height)) return die("could not write image");
    if (o.json) {
        const double kernel_ms = kernel_ms_total;
        std::printf("{\"samples\": %.0f, \"wall_s\": %.4f, \"kernel_ms\": %.3f, \"msamples_per_s\": %.3f, \"gpus\": %d, \"reduce\": \"%s\"}\n",
                    rays, elapsed, kernel_ms, rays / (kernel_ms * 1e3), o.gpus, o.gpus > 1 ? (o.reduce_rccl ? "rccl" : "host") : "none");
    }
    for (rt_scene* s : scenes) rt_scene_destroy(s);
    rt_scene_desc_free(desc);
    return 0;
}
