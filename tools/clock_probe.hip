// Diagnostics only (not part of the product library): measures the shader clock the GPU is actually running at.
// One wave runs a dependent FMA chain of known length; wall_clock64() ticks at a constant 100 MHz.
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(long long* out, int iters) {
    const long long w0 = wall_clock64();
    float x = (float)threadIdx.x;
#pragma unroll 16
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = (long long)x; }
}
extern "C" int clock_probe(long long* out, int iters, void* stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
