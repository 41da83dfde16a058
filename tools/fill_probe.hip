// Diagnostics only (not part of the product library): the rate at which ONE CU can stage operand tiles, by path.
// Every block (512 threads = 8 waves, one block per CU) streams `rows` x 128-byte row slices per iteration out of an L2-/MALL-resident region:
//   mode 0: buffer_load_dwordx4 ... lds  (the GEMM's LDS-DMA pieces: 8 lanes per 128-byte slice, 8 rows per wave instruction)
//   mode 1: global_load_dwordx4 into registers (xor-reduced), same addresses
//   mode 2: global_load_dwordx4 into registers + ds_write_b128 into LDS (register-staged tile)
// `depth` batches of `pieces` wave instructions stay in flight (s_waitcnt vmcnt(pieces * (depth - 1)) per iteration).
// Row pitch `ld` bytes: 128 = contiguous, 640 = a K-slice of a [M][320] bf16 matrix, ...
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

template <int MODE, int PIECES, int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, long long block_stride, int region_rows, int ld, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const char* base = src + (long long)blockIdx.x * block_stride;
    const int rows_per_iter = 64 * PIECES;             // 8 waves x 8 rows x PIECES
    const int r0 = wave * 8 + (lane >> 3), slot = lane & 7;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)region_rows * (unsigned)ld, 0x00020000);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int row = 0;
    for (int it = 0; it < iters; ++it) {
        char* stage = smem + (it % DEPTH) * (rows_per_iter * 128);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const int off = (row + q * 64 + r0) * ld + slot * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(stage + (q * 8 + wave_u) * 1024), 16, off, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (DEPTH - 1)) : "memory");
        } else {
            // (register paths: the PIECES loads of a wave are issued together, then waited for together; the other waves overlap)
            u32x4_t v[PIECES];
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                const char* ptr = base + (long long)(row + q * 64 + r0) * ld + slot * 16;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[q]) : "v"(ptr) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < PIECES; ++q) {
                if constexpr (MODE == 2) *(u32x4_t*)(stage + (q * 64 + r0) * 128 + ((slot ^ (r0 & 7)) << 4)) = v[q];
                else acc ^= v[q];
            }
        }
        row += rows_per_iter;
        if (row + rows_per_iter > region_rows) row = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 1) acc[0] ^= *(const uint32_t*)(smem + tid * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE, int PIECES, int DEPTH>
static int launch(const char* src, long long bs, int rr, int ld, int iters, int* sink, int blocks, hipStream_t st) {
    auto k = fill_kernel<MODE, PIECES, DEPTH>;
    const int smem = DEPTH * 64 * PIECES * 128;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), smem, st, src, bs, rr, ld, iters, sink);
    return (int)hipGetLastError();
}

extern "C" int fill_probe(int mode, int pieces, int depth, const void* src, long long block_stride, int region_rows, int ld, int iters, int* sink, int blocks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const char* s = (const char*)src;
#define C(M, P, D) if (mode == M && pieces == P && depth == D) return launch<M, P, D>(s, block_stride, region_rows, ld, iters, sink, blocks, st);
    C(0, 4, 1) C(0, 4, 2) C(0, 4, 4) C(0, 9, 1) C(0, 9, 2) C(0, 2, 4) C(0, 2, 8)
    C(1, 4, 1) C(1, 4, 2) C(1, 4, 4) C(1, 9, 2)
    C(2, 4, 1) C(2, 4, 2)
#undef C
    return -1;
}
