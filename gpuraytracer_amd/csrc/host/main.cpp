// gpuPT - headless counterpart of the reference's application (RayTracer.cpp): loads a Wavefront
// scene, renders `spp` samples per pixel with the fixed camera of the reference's start-up state and
// writes the mean image as OpenEXR, printing samples per second like finalRender (RayTracer.cpp:29-53).
//
//   gpuPT <model base path> <width> <height> [spp=64] [out.exr=final.exr] [env.exr|-] [<material id>=<texture base> ...]
//
// Without an environment file (or with "-") a constant (0.5, 0.7, 1.0) sky is used (the reference's environment
// textures are not part of its repository).  "<id>=<base>" loads <base>_albedo.exr, <base>_normal.exr and
// <base>_arm.exr into the atlas layers of material <id>, as the loadMaterial calls of RayTracer.cpp:161-162 do.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "Scene.hpp"

int main(int argc, char *args[]) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <model base path> <width> <height> [spp] [out.exr] [env.exr|-] [<material id>=<texture base> ...]\n", args[0]);
        return EXIT_FAILURE;
    }
    const std::string name(args[1]);
    int width = atoi(args[2]), height = atoi(args[3]);
    const uint32_t spp = argc > 4 ? (uint32_t)atoi(args[4]) : 64u;      // the reference stops at 64 (RayTracer.cpp:45)
    const std::string out = argc > 5 ? args[5] : "final.exr";

    host::Scene scene(0);
    const bool loaded = scene.addWavefrontModel(name);
    std::cout << (loaded ? "Scene Loaded" : "Scene does not exist") << '\n';
    if (!loaded || !scene.finalizeObjects()) {
        fprintf(stderr, "%s\n", scene.lastError());
        return 1;
    }
    bool env_ok = false;
    if (argc > 6 && std::string(args[6]) != "-") env_ok = scene.loadEnvironmentTexture(args[6]);
    for (int k = 7; k < argc; k++) {
        const std::string spec(args[k]);
        const size_t eq = spec.find('=');
        if (eq == std::string::npos || eq == 0) { fprintf(stderr, "bad material argument %s\n", args[k]); return EXIT_FAILURE; }
        const uint32_t id = (uint32_t)atoi(spec.substr(0, eq).c_str());
        const bool textured = scene.loadMaterial(spec.substr(eq + 1), id);
        std::cout << "Material " << id << (textured ? " textured" : " not textured") << '\n';
    }
    if (!env_ok) {
        std::vector<float> env(8 * 4 * 4);
        for (int i = 0; i < 32; i++) { env[i * 4] = 0.5f; env[i * 4 + 1] = 0.7f; env[i * 4 + 2] = 1.0f; env[i * 4 + 3] = 1.0f; }
        if (!scene.setEnvironment(8, 4, env.data())) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
    }
    const double translation[3] = {0.0, 1.5, -3.0};                    // RayTracer.cpp:25
    const double rot[2] = {-0.78539816339744830962, 0.0};              // RayTracer.cpp:26
    float camera[16];
    host::cameraMatrix(translation, rot, camera);
    if (!scene.prepare(width, height, false, camera, 95834783u)) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    if (!scene.traceSamples(0, spp, 1) || !scene.exportEXR(out.c_str())) { fprintf(stderr, "%s\n", scene.lastError()); return 1; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << spp / dt << " samples/s, " << scene.rays() / dt / 1e6 << " Mrays/s, wrote " << out << std::endl;
    scene.exportPNG8((out + ".ppm").c_str());
    return 0;
}
