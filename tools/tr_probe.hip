// ds_read_b64_tr_b16 semantics probe: LDS holds lds[i] = i (16-bit words); every lane supplies a byte address and gets 4 words back.
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr_probe_kernel(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
extern "C" int tr_probe(const int* addr, unsigned short* out, void* stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
    return (int)hipGetLastError();
}
