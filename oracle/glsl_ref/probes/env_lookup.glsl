#version 450
// Probe of the GL implementation's bilinear REPEAT lookup in an RGBA32F texture: input (s, t) pairs,
// output texture(RADIANCE, vec2(s, t)).rgb
layout(local_size_x = 64) in;
layout(std430, binding = 7) restrict readonly buffer inBuffer { float xs[]; };
layout(std430, binding = 8) restrict writeonly buffer outBuffer { float ys[]; };
uniform layout(location = 20) uint N;
uniform layout(location = 2) sampler2D RADIANCE;
void main(void) {
    uint i = gl_GlobalInvocationID.x;
    if (i >= N) return;
    vec3 c = texture(RADIANCE, vec2(xs[2u*i], xs[2u*i+1u])).rgb;
    ys[3u*i] = c.r; ys[3u*i+1u] = c.g; ys[3u*i+2u] = c.b;
}
