// Diagnostics only (not part of the product library): what does the ACCESS PATTERN of a GEMM epilogue's global stores / loads cost?
// Every wave owns a [32 rows][RB bytes] bf16 block of a row-major [M][ld] matrix and moves it with 16-byte accesses, 64 lanes per instruction:
//   mode 0  row-per-lane (the direct epilogue of gemm.hip): lane (m = l & 31, h = l >> 5) touches row m, bytes [32 (2 j + ...) ...] -- per
//           instruction 64 isolated 16-byte pieces in 32 rows (lane pair m / m + 32 is 32 bytes apart)
//   mode 1  pair-coalesced: lanes m and m + 32 touch adjacent 16-byte pieces (32 contiguous bytes per row and instruction)
//   mode 2  quad-coalesced: lanes 4k .. 4k+3 touch 64 contiguous bytes of one row (16 rows x 64 B per instruction)
//   mode 3  line-coalesced: lanes 8k .. 8k+7 touch 128 contiguous bytes of one row (8 full cache lines per instruction)
// op 0 = stores, 1 = loads (summed into a sink).  The block geometry mimics a 128 x 160 bf16 tile per 4-wave block (RB = 320 bytes per row).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int OP>
__global__ __launch_bounds__(256) void store_probe_kernel(char* base, long long ld_bytes, int rows_total, int RB, u32x4* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (row0 + 32 > rows_total) return;
    const int pieces = RB / 16;          // 16-byte pieces per row
    u32x4 acc = {0u, 0u, 0u, 0u};
    const u32x4 val = {(uint32_t)lane, (uint32_t)wave, (uint32_t)blockIdx.x, 7u};
    // instruction count is the same in every mode: 32 rows * pieces / 64 lanes
    const int ninst = 32 * pieces / 64;
    for (int it = 0; it < ninst; ++it) {
        int row, pc;
        if (MODE == 0) {            // lane = (m, h): piece index inside the row = 2 * (it) + ... as the direct epilogue walks: block j = it / 2, half h, piece it % 2
            const int m = lane & 31, h = lane >> 5;
            row = m;
            pc = (it >> 1) * 4 + h * 2 + (it & 1);
        } else if (MODE == 1) {     // lanes m, m + 32 adjacent pieces
            const int m = lane & 31, h = lane >> 5;
            row = m;
            pc = it * 2 + h;
        } else if (MODE == 2) {     // quads: 16 rows per instruction, 4 pieces each
            const int grp = lane >> 2, i = lane & 3;
            const int per_row = pieces / 4;                   // quad-slots per row
            const int slot = it * 16 + grp;                   // global quad-slot
            row = slot / per_row;
            pc = (slot % per_row) * 4 + i;
        } else {                    // 8 lanes = one 128-byte line
            const int grp = lane >> 3, i = lane & 7;
            const int per_row = pieces / 8;
            const int slot = it * 8 + grp;
            row = per_row > 0 ? slot / per_row : 0;
            pc = per_row > 0 ? (slot % per_row) * 8 + i : 0;
        }
        if (row < 32 && pc < pieces) {
            u32x4* p = (u32x4*)(base + (row0 + row) * ld_bytes + (long long)pc * 16);
            if (OP == 0) *p = val;
            else { const u32x4 v = *p; acc += v; }
        }
    }
    if (OP == 1 && acc[0] == 0x12345678u) sink[0] = acc;
}

extern "C" int store_probe(int mode, int op, void* base, long long ld_bytes, int rows_total, int RB, void* sink, void* stream) {
    const int blocks = rows_total / 128;
#define L(M, O) hipLaunchKernelGGL((store_probe_kernel<M, O>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (char*)base, ld_bytes, rows_total, RB, (u32x4*)sink)
    if (op == 0) { if (mode == 0) L(0, 0); else if (mode == 1) L(1, 0); else if (mode == 2) L(2, 0); else L(3, 0); }
    else { if (mode == 0) L(0, 1); else if (mode == 1) L(1, 1); else if (mode == 2) L(2, 1); else L(3, 1); }
#undef L
    return (int)hipGetLastError();
}
