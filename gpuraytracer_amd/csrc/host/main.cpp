// gpuPT - headless counterpart of the reference's application (RayTracer.cpp): loads a Wavefront
// scene, renders `spp` samples per pixel with the fixed camera of the reference's start-up state and
// writes the mean image as OpenEXR, printing samples per second like finalRender (RayTracer.cpp:29-53).
//
//   gpuPT [--devices=0,1,...] [--preview=K] <model base path> <width> <height> [spp=64] [out.exr=final.exr] [env.exr|-]
//         [<material id>=<texture base> ...]
//
// Without an environment file (or with "-") a constant (0.5, 0.7, 1.0) sky is used (the reference's environment
// textures are not part of its repository).  "<id>=<base>" loads <base>_albedo.exr, <base>_normal.exr and
// <base>_arm.exr into the atlas layers of material <id>, as the loadMaterial calls of RayTracer.cpp:161-162 do.
//
// --preview=K (one GPU) plays the reference's interactive sequence (RayTracer.cpp:122-143): the still image is traced
// sample by sample, after half of its samples the camera "moves" for K frames (prepare(moving = true): the low-resolution
// target, RECURSION 3, a camera one step further back, SAMPLE 0 .. K-1), the preview is written to <out>.preview.exr, and
// the still image then carries on from where it was.
//
// --devices names the GPUs to render on (the same ordinal may appear more than once).  The scene is replicated,
// GPU r of N renders SAMPLE = r, r + N, r + 2N, ... (all GPUs are driven from this one thread: tracing only enqueues
// work), and the first GPU then adds the sums of the others in the order given, over the peer links.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "Scene.hpp"

int main(int argc, char *argv[]) {
    std::vector<int> devices;
    std::vector<std::string> args;
    int preview_frames = 0;
    for (int k = 1; k < argc; k++) {
        const std::string a(argv[k]);
        if (a.rfind("--devices=", 0) == 0) {
            size_t pos = 10;
            while (pos <= a.size()) {
                const size_t comma = a.find(',', pos);
                const std::string item = a.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                if (!item.empty()) devices.push_back(atoi(item.c_str()));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
        } else if (a.rfind("--preview=", 0) == 0) {
            preview_frames = atoi(a.c_str() + 10);
        } else {
            args.push_back(a);
        }
    }
    if (devices.empty()) devices.push_back(0);
    if (args.size() < 3) {
        fprintf(stderr, "usage: %s [--devices=0,1,...] <model base path> <width> <height> [spp] [out.exr] [env.exr|-] "
                        "[<material id>=<texture base> ...]\n", argv[0]);
        return EXIT_FAILURE;
    }
    const std::string name(args[0]);
    int width = atoi(args[1].c_str()), height = atoi(args[2].c_str());
    const uint32_t spp = args.size() > 3 ? (uint32_t)atoi(args[3].c_str()) : 64u;      // the reference stops at 64 (RayTracer.cpp:45)
    const std::string out = args.size() > 4 ? args[4] : "final.exr";
    const uint32_t n_dev = (uint32_t)devices.size();

    const double translation[3] = {0.0, 1.5, -3.0};                    // RayTracer.cpp:25
    const double rot[2] = {-0.78539816339744830962, 0.0};              // RayTracer.cpp:26
    float camera[16];
    host::cameraMatrix(translation, rot, camera);

    std::vector<std::unique_ptr<host::Scene>> scenes;
    for (uint32_t r = 0; r < n_dev; r++) {
        scenes.emplace_back(new host::Scene(devices[r]));
        host::Scene &scene = *scenes.back();
        const bool loaded = scene.addWavefrontModel(name);
        if (r == 0) std::cout << (loaded ? "Scene Loaded" : "Scene does not exist") << '\n';
        if (!loaded || !scene.finalizeObjects()) {
            fprintf(stderr, "%s\n", scene.lastError());
            return 1;
        }
        bool env_ok = false;
        if (args.size() > 5 && args[5] != "-") env_ok = scene.loadEnvironmentTexture(args[5]);
        if (!env_ok) {
            std::vector<float> env(8 * 4 * 4);
            for (int i = 0; i < 32; i++) { env[i * 4] = 0.5f; env[i * 4 + 1] = 0.7f; env[i * 4 + 2] = 1.0f; env[i * 4 + 3] = 1.0f; }
            if (!scene.setEnvironment(8, 4, env.data())) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
        }
        for (size_t k = 6; k < args.size(); k++) {
            const std::string &spec = args[k];
            const size_t eq = spec.find('=');
            if (eq == std::string::npos || eq == 0) { fprintf(stderr, "bad material argument %s\n", spec.c_str()); return EXIT_FAILURE; }
            const uint32_t id = (uint32_t)atoi(spec.substr(0, eq).c_str());
            const bool textured = scene.loadMaterial(spec.substr(eq + 1), id);
            if (r == 0) std::cout << "Material " << id << (textured ? " textured" : " not textured") << '\n';
        }
        if (!scene.prepare(width, height, false, camera, 95834783u)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
    }

    const auto t0 = std::chrono::steady_clock::now();
    if (preview_frames > 0 && n_dev == 1) {
        host::Scene &scene = *scenes[0];
        uint32_t sample = 0;
        for (; sample < spp / 2; sample++)
            if (!scene.traceScene((uint32_t)width, (uint32_t)height, sample)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
        const double moved[3] = {0.0, 1.5, -3.5};
        float camera_moved[16];
        host::cameraMatrix(moved, rot, camera_moved);
        int low_w = width, low_h = height;
        if (!scene.prepare(low_w, low_h, true, camera_moved, 95834783u)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
        for (int k = 0; k < preview_frames; k++)
            if (!scene.traceScene((uint32_t)low_w, (uint32_t)low_h, (uint32_t)k)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
        if (!scene.exportEXR((out + ".preview.exr").c_str())) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
        std::cout << "preview " << low_w << 'x' << low_h << ", " << preview_frames << " frames" << '\n';
        if (!scene.prepare(width, height, false, camera, 95834783u)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
        for (; sample < spp; sample++)
            if (!scene.traceScene((uint32_t)width, (uint32_t)height, sample)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
    } else
    for (uint32_t r = 0; r < n_dev; r++) {
        const uint32_t count = spp > r ? (spp - r + n_dev - 1) / n_dev : 0u;       // samples r, r + N, ... below spp
        if (count && !scenes[r]->traceSamples(r, count, n_dev)) { fprintf(stderr, "%s\n", scenes[r]->lastError()); return 1; }
    }
    for (uint32_t r = 1; r < n_dev; r++)
        if (!scenes[0]->addSumsOf(*scenes[r])) { fprintf(stderr, "%s\n", scenes[0]->lastError()); return 1; }
    if (!scenes[0]->exportEXR(out.c_str())) { fprintf(stderr, "%s\n", scenes[0]->lastError()); return 1; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t rays = 0;
    for (auto &s : scenes) rays += s->rays();
    std::cout << spp / dt << " samples/s, " << rays / dt / 1e6 << " Mrays/s on " << n_dev << " GPU" << (n_dev > 1 ? "s" : "")
              << ", wrote " << out << std::endl;
    scenes[0]->exportPNG8((out + ".ppm").c_str());
    return 0;
}
