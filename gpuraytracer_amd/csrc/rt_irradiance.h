// rt_irradiance.h - irradiance pre-filter of the environment map.
//
// Scene::loadEnvironmentTexture (Scene.cpp:150-186) dispatches irradiance.glsl once per radiance row; every
// texel of the (w/4) x (h/4) irradiance image accumulates, in row order and left to right within a row,
//     weight * radiance,   weight = max(uvDotUV(uv, UV), 0) * 2 / (w * h)        (irradiance.glsl:16-18, :27-38)
// with uv = (i, row) / size and UV = 4 * texel / size (corner coordinates, as in the reference).  The sum of a
// texel is a chain of w * h float additions in a fixed order, so one lane owns one texel and walks the whole
// radiance image; the radiance texel of a step is the same for every lane and comes through scalar loads.
// Arithmetic as Mesa evaluates the shader: products left to right, no fused multiply-add, gallivm's sin/cos.
#pragma once

#include "rt_device.h"

namespace rt {

__global__ __launch_bounds__(64) void k_irradiance(const float4 *__restrict__ env, uint32_t w, uint32_t h,
                                                   float4 *__restrict__ out) {
    const uint32_t iw = w / 4u, ih = h / 4u;
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= iw * ih) return;
    const float PI = 3.141592653589793f, TWO_PI = 2.0f * PI;
    const float isx = 1.0f / (float)w, isy = 1.0f / (float)h;
    const float renorm = (2.0f / (float)w) / (float)h;      // Mesa: these two divisions stay divisions
    const uint32_t X = t % iw, Y = t / iw;
    const float Ux = ((float)X * isx) * 4.0f, Uy = ((float)Y * isy) * 4.0f;
    float sU, cU;
    gm::sincos(PI * Uy, sU, cU);
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t row = 0; row < h; row++) {
        const float uy = (float)row * isy;
        float sy, cy;
        gm::sincos(PI * uy, sy, cy);
        const float polar = cy * cU;
        const float4 *line = env + (size_t)row * w;
        for (uint32_t i = 0; i < w; i++) {
            const float ux = (float)i * isx;
            const float c = gm::cos(TWO_PI * (ux - Ux));
            const float weight = gm::fmax(polar + (c * sy) * sU, 0.0f) * renorm;
            const float4 px = line[i];
            r = r + weight * px.x;
            g = g + weight * px.y;
            b = b + weight * px.z;
        }
    }
    out[t] = make_float4(r, g, b, 1.0f);
}

}  // namespace rt
