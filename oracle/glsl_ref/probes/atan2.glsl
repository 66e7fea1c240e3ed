#version 450
// Probe of the GL implementation's two-argument arctangent: input (y, x) pairs, output atan(y, x)
layout(local_size_x = 64) in;
layout(std430, binding = 7) restrict readonly buffer inBuffer { float xs[]; };
layout(std430, binding = 8) restrict writeonly buffer outBuffer { float ys[]; };
uniform layout(location = 20) uint N;
void main(void) {
    uint i = gl_GlobalInvocationID.x;
    if (i >= N) return;
    ys[i] = atan(xs[2u*i], xs[2u*i+1u]);
}
