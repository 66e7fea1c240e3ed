#version 450
// Probe of the GL implementation's elementary functions: for each input x writes
// sin(x), cos(x), pow(x, 5), tan(x), acos(x), asin(x), atan(x, 0.37), sqrt(x), inversesqrt(x), 1/x, exp(x), log(x)
layout(local_size_x = 64) in;
layout(std430, binding = 7) restrict readonly buffer inBuffer { float xs[]; };
layout(std430, binding = 8) restrict writeonly buffer outBuffer { float ys[]; };
uniform layout(location = 20) uint N;
void main(void) {
    uint i = gl_GlobalInvocationID.x;
    if (i >= N) return;
    float x = xs[i];
    ys[12u*i+0u] = sin(x);
    ys[12u*i+1u] = cos(x);
    ys[12u*i+2u] = pow(x, 5);
    ys[12u*i+3u] = tan(x);
    ys[12u*i+4u] = acos(x);
    ys[12u*i+5u] = asin(x);
    ys[12u*i+6u] = atan(x, 0.37);
    ys[12u*i+7u] = sqrt(x);
    ys[12u*i+8u] = inversesqrt(x);
    ys[12u*i+9u] = 1.0 / x;
    ys[12u*i+10u] = exp(x);
    ys[12u*i+11u] = log(x);
}
