#version 450
// Second probe of elementary functions: log2(x), exp2(x), pow(x, 2.4), pow(1.0 - x, 5)
layout(local_size_x = 64) in;
layout(std430, binding = 7) restrict readonly buffer inBuffer { float xs[]; };
layout(std430, binding = 8) restrict writeonly buffer outBuffer { float ys[]; };
uniform layout(location = 20) uint N;
void main(void) {
    uint i = gl_GlobalInvocationID.x;
    if (i >= N) return;
    float x = xs[i];
    ys[4u*i+0u] = log2(x);
    ys[4u*i+1u] = exp2(x);
    ys[4u*i+2u] = pow(x, 2.4);
    ys[4u*i+3u] = pow(1.0 - x, 5);
}
