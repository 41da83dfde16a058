// Does a wave's VALU work issue in the shadow of its own MFMAs?  One wave per SIMD (256 threads per block, one block per CU) loops over
// { 1 x v_mfma_f32_32x32x16_bf16 ; N independent VALU } and reports cycles per iteration.  KIND: 0 v_fma_f32, 1 v_exp_f32, 2 v_max3_f32 on
// values produced by an older MFMA, 3 v_cvt_pk_bf16_f32.  ACC: 0 accumulators in VGPRs, 1 in AGPRs.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int KIND, int ACC>
__global__ __launch_bounds__(256) void probe(long long* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r + i);
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x & 7); b[r] = (__bf16)1.0f; }
    float x[12];
    for (int r = 0; r < 12; ++r) x[r] = 0.001f * (float)(threadIdx.x + r);
    const float c = 1.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int r = 0; r < N; ++r) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(c));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[r]));
                else if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[r]) : "v"(c));
                else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c), "v"(c));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 12; ++r) s += x[r];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 123.456f) out[1] = 1;
}

template <int KIND, int ACC> static void sweep(long long* out, int iters, long long* host) {
#define RUN(N) { hipLaunchKernelGGL((probe<N, KIND, ACC>), dim3(256), dim3(256), 0, 0, out, iters); hipDeviceSynchronize(); hipMemcpy(host + N, out, 8, hipMemcpyDeviceToHost); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
#undef RUN
}

// dependent accumulation chains: DEPTH accumulators used round-robin (1 = every MFMA waits for the previous one)
template <int DEPTH, int N>
__global__ __launch_bounds__(256) void chain_probe(long long* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + r + i);
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x & 7); b[r] = (__bf16)1.0f; }
    float x[12];
    for (int r = 0; r < 12; ++r) x[r] = 0.001f * (float)(threadIdx.x + r);
    const float c = 1.0001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % DEPTH]) : "v"(a), "v"(b));
#pragma unroll
            for (int r = 0; r < N; ++r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[r]) : "v"(c));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 12; ++r) s += x[r];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 123.456f) out[1] = 1;
}
extern "C" int mfma_chain_probe(int iters, long long* host6) {
    long long* out;
    hipMalloc(&out, 16);
    hipMemset(out, 0, 16);
#define RUNC(D, N, K) { hipLaunchKernelGGL((chain_probe<D, N>), dim3(256), dim3(256), 0, 0, out, iters); hipDeviceSynchronize(); hipMemcpy(host6 + K, out, 8, hipMemcpyDeviceToHost); }
    RUNC(1, 0, 0) RUNC(2, 0, 1) RUNC(4, 0, 2) RUNC(1, 4, 3) RUNC(2, 4, 4) RUNC(4, 4, 5)
#undef RUNC
    hipFree(out);
    return (int)hipGetLastError();
}

extern "C" int mfma_valu_probe(int kind, int acc, int iters, long long* host13) {
    long long* out;
    hipMalloc(&out, 16);
    hipMemset(out, 0, 16);
    if (kind == 0 && acc == 0) sweep<0, 0>(out, iters, host13);
    else if (kind == 0) sweep<0, 1>(out, iters, host13);
    else if (kind == 1 && acc == 0) sweep<1, 0>(out, iters, host13);
    else if (kind == 1) sweep<1, 1>(out, iters, host13);
    else if (kind == 2 && acc == 0) sweep<2, 0>(out, iters, host13);
    else if (kind == 2) sweep<2, 1>(out, iters, host13);
    else if (acc == 0) sweep<3, 0>(out, iters, host13);
    else sweep<3, 1>(out, iters, host13);
    hipFree(out);
    return (int)hipGetLastError();
}
