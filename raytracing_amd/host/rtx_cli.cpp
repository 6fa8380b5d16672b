// rtx_cli.cpp -- the command line of the reference (RayTracing.cpp:42-180) with the GPU mode added.
// Keeps -a -b -d -f -m -r -t; adds -g (HIP backend, the default and only renderer here), -w / -h (image size, the
// reference's COLS / ROWS are compile-time, RayTracing.cpp:23-24), -s (seed), -n book1|glass|scaled (scene variant),
// -G <number of GPUs> (samples split over devices 0 .. N-1, one RCCL all-reduce), -D <first device>, -p <spp per pass> (progressive: the image file is rewritten after every pass), -k <file> (checkpoint of the
// float accumulation buffer after every pass), -R (resume from the -k file), -6 (binary P6 instead of P3).
// -c (CUDA) and -i (ISPC) name back ends this build does not contain and are refused.
#include "scene_types.h"
#include "render_scene_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace pk;

class ArgsParser {   // same behaviour as argsparser.h:10-37
public:
    ArgsParser(int argc, char** argv) { for (int i = 1; i < argc; ++i) tokens.push_back(argv[i]); }
    const std::string& getCmdOption(const std::string& option) const
    {
        static const std::string empty;
        auto it = std::find(tokens.begin(), tokens.end(), option);
        if (it != tokens.end() && ++it != tokens.end()) return *it;
        return empty;
    }
    bool cmdOptionExists(const std::string& option) const { return std::find(tokens.begin(), tokens.end(), option) != tokens.end(); }
private:
    std::vector<std::string> tokens;
};

static Scene* randomScene(uint64_t seed, const std::string& variant)   // RayTracing.cpp:182-213, seeded
{
    int grid_half = 11; float p_diffuse = 0.8f, p_metal = 0.95f;
    if (variant == "glass") p_diffuse = 0.1f;
    if (variant == "scaled") grid_half = 71;
    const uint32_t cap = rtx_scene_capacity(grid_half);
    std::vector<float> f[9]; for (auto& v : f) v.resize(cap);
    std::vector<uint32_t> type(cap);
    uint32_t n = 0;
    rtx_random_scene(seed, grid_half, p_diffuse, p_metal, &n, f[0].data(), f[1].data(), f[2].data(), f[3].data(), type.data(), f[4].data(), f[5].data(), f[6].data(), f[7].data(), f[8].data());
    Scene* scene = new Scene();
    for (uint32_t i = 0; i < n; ++i)
        scene->objects.push_back(new Sphere(vector3(f[0][i], f[1][i], f[2][i]), f[3][i], new material_t((material_type_t)type[i], vector3(f[4][i], f[5][i], f[6][i]), f[7][i], f[8][i])));
    return scene;
}

int main(int argc, char** argv)
{
    ArgsParser args(argc, argv);
    if (args.cmdOptionExists("-c") || args.cmdOptionExists("-i")) { printf("Error: this build contains the HIP backend only (-g); -c and -i are not available.\n"); return -1; }
    int aaSamples = 24;                                            // NUM_AA_SAMPLES, RayTracing.cpp:31
    if (args.cmdOptionExists("-a")) aaSamples = std::stoi(args.getCmdOption("-a"));
    if (aaSamples <= 0) { printf("Error: -a needs a positive sample count.\n"); return -1; }
    int blockSize = 64;
    if (args.cmdOptionExists("-b")) blockSize = std::stoi(args.getCmdOption("-b"));
    bool debug = args.cmdOptionExists("-d");
    std::string filename = "foo.ppm";
    if (args.cmdOptionExists("-f")) filename = args.getCmdOption("-f");
    int maxBounce = 5;                                             // MAX_RAY_DEPTH, RayTracing.cpp:34
    if (args.cmdOptionExists("-m")) maxBounce = std::stoi(args.getCmdOption("-m"));
    bool recursive = args.cmdOptionExists("-r");
    int numThreads = (int)std::thread::hardware_concurrency() - 1;
    if (args.cmdOptionExists("-t")) numThreads = std::stoi(args.getCmdOption("-t"));
    unsigned cols = 2000, rows = 1000;                             // COLS, ROWS, RayTracing.cpp:23-24
    if (args.cmdOptionExists("-w")) cols = (unsigned)std::stoi(args.getCmdOption("-w"));
    if (args.cmdOptionExists("-h")) rows = (unsigned)std::stoi(args.getCmdOption("-h"));
    uint64_t seed = 1;
    if (args.cmdOptionExists("-s")) seed = std::stoull(args.getCmdOption("-s"));
    std::string variant = "book1";
    if (args.cmdOptionExists("-n")) variant = args.getCmdOption("-n");
    int gpus = 1, device = 0;
    if (args.cmdOptionExists("-G")) gpus = std::stoi(args.getCmdOption("-G"));
    if (args.cmdOptionExists("-D")) device = std::stoi(args.getCmdOption("-D"));
    if (gpus < 1 || device < 0) { printf("Error: -G needs a positive GPU count and -D a device index.\n"); return -1; }
    if (args.cmdOptionExists("-G") && !args.cmdOptionExists("-D")) printf("note: -G %d is a GPU COUNT (devices 0 .. %d); -D <index> chooses the first device.\n", gpus, gpus - 1);
    unsigned passSpp = 0;
    if (args.cmdOptionExists("-p")) passSpp = (unsigned)std::stoi(args.getCmdOption("-p"));
    std::string checkpoint;
    if (args.cmdOptionExists("-k")) checkpoint = args.getCmdOption("-k");
    bool resume = args.cmdOptionExists("-R");
    if (resume && checkpoint.empty()) { printf("Error: -R needs -k <checkpoint file>.\n"); return -1; }
    bool binary = args.cmdOptionExists("-6");
    int filter = -1;                                               // -F: how _sceneHit's candidates are found (results are identical)
    if (args.cmdOptionExists("-F")) {
        const std::string f = args.getCmdOption("-F");
        filter = f == "valu" ? RTX_FILTER_VALU : f == "mfma" ? RTX_FILTER_MFMA : f == "grid" ? RTX_FILTER_GRID : f == "auto" ? RTX_FILTER_AUTO : -2;
        if (filter == -2) { printf("Error: -F takes valu, mfma, grid or auto.\n"); return -1; }
    }

    Scene* scene = randomScene(seed, variant);
    Camera camera(20, float(cols) / float(rows), 0.1f, 10.0f, vector3(13, 2, 3), vector3(0, 1, 0), vector3(0, 0, 0));   // RayTracing.cpp:107-117

    std::vector<uint32_t> frameBuffer((size_t)rows * cols, 0xCCCCCCCCu);
    FILE* probe = fopen(filename.c_str(), "w");
    if (!probe) { printf("Error: failed to open [%s] for writing.\n", filename.c_str()); delete scene; return -1; }
    fclose(probe);

    rtx_stats st{};
    static rtx_stats tot{};                 // all passes (opt.stats holds the last pass only)
    static unsigned  tot_passes = 0;
    struct PassCtx { std::string file; unsigned cols, rows, total; bool binary, write; const rtx_stats* last; } pc{ filename, cols, rows, (unsigned)aaSamples, binary, passSpp != 0, &st };
    HipRenderOptions opt;
    opt.seed = seed; opt.device = device; opt.n_devices = gpus; opt.stats = &st; opt.pass_spp = passSpp; opt.resume = resume; opt.filter = filter;
    opt.checkpoint = checkpoint.empty() ? nullptr : checkpoint.c_str();
    opt.user = &pc;
    opt.on_pass = [](const uint32_t* fb, unsigned done, void* user) -> bool {
        const PassCtx* p = (const PassCtx*)user;
        tot.samples += p->last->samples; tot.rays += p->last->rays; tot.kernel_ms += p->last->kernel_ms; tot_passes++;
        if (!p->write) return true;             // one-shot render: main() writes the file
        printf("pass done: %u of %u samples\n", done, p->total);
        fflush(stdout);
        return (p->binary ? rtx_write_ppm_p6(p->file.c_str(), p->cols, p->rows, fb) : rtx_write_ppm(p->file.c_str(), p->cols, p->rows, fb)) == 0;
    };
    auto t0 = std::chrono::steady_clock::now();
    int  rc = renderSceneHIP(*scene, camera, rows, cols, frameBuffer.data(), aaSamples, maxBounce, numThreads, blockSize, debug, recursive, opt);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != 0) { delete scene; return -1; }
    printf("renderSceneHIP: %f s (%zu spheres, %u x %u, %d spp, depth %d)\n", sec, scene->objects.size(), cols, rows, aaSamples, maxBounce);
    if (st.kernel_ms > 0.0) {    // (no pass ran -- -R on a finished checkpoint -- : there is nothing to report)
        // one machine-readable line: whole-job figures over all passes and GPUs; kernel_ms sums, pass by pass, the
        // slowest GPU's trace + reduce kernels.  mfma_frac: f16 matrix-core flop the linear list's filter issued (64 per ray and
        // sphere of the table padded to 128) against the dense f16 peak of the GPUs used; 0 when another candidate search ran.
        const size_t n_pad = (scene->objects.size() + 127) / 128 * 128;
        const char*  envf = getenv("RTX_FILTER");
        const int    eff = filter >= 0 ? filter : (envf ? (std::string(envf) == "mfma" ? RTX_FILTER_MFMA : std::string(envf) == "auto" ? RTX_FILTER_AUTO : RTX_FILTER_GRID) : RTX_FILTER_AUTO);
        const bool   linear = eff == RTX_FILTER_MFMA || (eff == RTX_FILTER_AUTO && scene->objects.size() < 257);
        const double mfma_frac = linear ? 64.0 * (double)tot.rays * (double)n_pad / (tot.kernel_ms * 1e-3) / (2.5e15 * gpus) : 0.0;
        printf("{\"msamples_per_s\": %.3f, \"mrays_per_s\": %.3f, \"kernel_ms\": %.3f, \"rays\": %llu, \"samples\": %llu, \"n_gpus\": %d, \"passes\": %u, \"mfma_frac_of_f16_peak\": %.4f, \"wall_s\": %.3f}\n",
               tot.samples / (tot.kernel_ms * 1e3), tot.rays / (tot.kernel_ms * 1e3), tot.kernel_ms, (unsigned long long)tot.rays, (unsigned long long)tot.samples, gpus, tot_passes, mfma_frac, sec);
    }
    rc = binary ? rtx_write_ppm_p6(filename.c_str(), cols, rows, frameBuffer.data()) : rtx_write_ppm(filename.c_str(), cols, rows, frameBuffer.data());
    delete scene;
    return rc == 0 ? 0 : -1;
}
