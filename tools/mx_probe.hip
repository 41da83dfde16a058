// v_mfma_scale_f32_32x32x64_f8f6f4 semantics probe (fp8 e4m3 x fp8 e4m3, E8M0 block scales): one wave; lane l supplies 32 bytes of A
// (row l & 31, bytes [32 (l >> 5), +32) of the row's 64-byte K slice), the same of B, and one scale dword per operand.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int OPA, int OPB>
__global__ void mx_probe_kernel(const uint8_t* A, const uint8_t* B, const uint32_t* sa, const uint32_t* sb, float* out) {
    const int l = threadIdx.x;
    v8i a, b;
    const int* ap = (const int*)(A + (l & 31) * 64 + (l >> 5) * 32);
    const int* bp = (const int*)(B + (l & 31) * 64 + (l >> 5) * 32);
    for (int r = 0; r < 8; ++r) { a[r] = ap[r]; b[r] = bp[r]; }
    v16f c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPA, (int)sa[l], OPB, (int)sb[l]);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
extern "C" int mx_probe(const void* A, const void* B, const void* sa, const void* sb, float* out, int opa, int opb, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define L(OA, OB) hipLaunchKernelGGL((mx_probe_kernel<OA, OB>), dim3(1), dim3(64), 0, s, (const uint8_t*)A, (const uint8_t*)B, (const uint32_t*)sa, (const uint32_t*)sb, out)
    if (opa == 0 && opb == 0) L(0, 0);
    else if (opa == 1 && opb == 0) L(1, 0);
    else if (opa == 0 && opb == 2) L(0, 2);
    else if (opa == 3 && opb == 3) L(3, 3);
    else return -1;
    return (int)hipGetLastError();
}
