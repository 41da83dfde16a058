// Fused transformer feed-forward  out = (GEGLU(x W1^T + b1)) W2^T + b2 + residual   (attention.py:40-76: FeedForward with GEGLU)
// for C = 320 (the 64x64 level, M = 65536 tokens per CFG batch of 16): the [M, 4C] hidden tensor -- 168 MB written by the GEGLU GEMM
// and read back by ff.net.2 in the unfused form -- never leaves the CU.
//
// Everything is computed TRANSPOSED, tokens on lanes (the flash-attention trick of attention.hip):
//   S^T[hidden, tok]  = W1c[hidden, :] . X^T[:, tok]     MFMA A = W1 rows (LDS, streamed), B = X rows (registers, resident)
//   H^T               = value * gelu(gate)               in registers: value / gate blocks have the same (lane, register) map
//   O^T[n, tok]      += W2[n, hidden] . H^T[hidden, tok]  MFMA A = W2 rows (LDS, streamed), B = H^T straight from registers
// The K order a register-fed B operand implies (rows 0-3, 8-11 | 4-7, 12-15 of each 16) is absorbed by the host-side column order
// of W2; the W2 rows of a 32-row block are read in the permuted order of gemm.hip's direct epilogue, so a lane ends with 16
// contiguous output columns of its token: bias + residual + 16-byte stores straight from registers.
//
// Block = 4 waves = 128 tokens, ONE wave per SIMD (up to 512 VGPRs: 160 output + 64 hidden accumulators + the wave's 80 registers of
// X^T fragments, resident for the whole block).  LDS holds weights only: a 4-stage ring of W1 K tiles (64 KB), two W2 chunks (80 KB),
// per-wave bias slots -- 148 KB.  Hidden dimension in chunks of 64 (20 chunks).
#include "common.h"

#ifndef RF_FFN_SPREAD
#define RF_FFN_SPREAD 2          // 0: bursts behind the barriers (rounds 3-4) | 1: the W2 pieces two per K tile | 2: every LDS-DMA piece between MFMAs (profiles/r05n_ffn_dma_spread.txt)
#endif

namespace rf {

// (the 16-bit operands are addressed as uint16_t: the kernel's element type T -- bf16_t or f16_t -- only selects conversions and the MFMA)
struct FfnParams {
    const bf16_t* x; int ldx;
    const bf16_t* w1; const float* b1;       // [8C][C] GEGLU-packed rows (32 value | 32 gate), [8C]
    const bf16_t* w2; const float* b2;       // [C][4C] hidden columns permuted inside every 16-group, [C]
    const bf16_t* res; int ldr;
    bf16_t* out; int ldo;
    int M;
    float ln_eps;          // > 0: x rows are LayerNorm-ed in registers first (no affine: gamma / beta are folded into w1 / b1 by the host)
    // SpatialTransformer.proj_out behind the feed-forward (attention.py:268-272, 288-289; round 5): the block's output tokens never leave the CU --
    //   out = ((GEGLU ...) W2^T + b2 + residual) Wpo^T + bpo + res2[row % res2_rows]
    // wpo [C][C] plain rows (no column permutation: the B operand is built in natural K order, see the epilogue); NULL = no projection.
    const bf16_t* wpo; const float* bpo;
    const bf16_t* res2; int ldr2; int res2_rows;
    // attn1.to_out IN FRONT of the feed-forward (attention.py:239-243; round 6, FRONT kernels): `x` is then the attention's output [front_rows][ldx] and the block first forms
    //   x1 = x Wo^T + bo + ctx[row / rows_per_sample0] + res0[row % front_rows]            (out-projection + cross-attention vector + residual tok)
    // rounds it to the storage type, stores it to `x1` (the feed-forward's residual, read back through `res`) and goes on with it as the feed-forward's input.
    const bf16_t* wo; const float* bo;       // [C][C] plain rows, [C]
    const float* ctx; int ldc; int rows_per_sample0;
    const bf16_t* res0; int ldr0; int front_rows;
    bf16_t* x1; int ldx1;
    // GroupNorm(32) partial sums of `out` for up to two consumers (the layout rf_conv_gemm's epilogue writes: one slot per 128-row block and sample)
    int gn_rows;
    double* gn_part[2];
    int gn_cpg[2], gn_coff[2], gn_slot[2], gn_nch[2];
};

__device__ __forceinline__ int ffn_lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

template <int C, bool PROJ, typename T = bf16_t, bool FRONT = false>
__global__ __launch_bounds__(256, 1) void ffn_geglu_kernel(const FfnParams p) {
    static_assert(sizeof(T) == 2, "16-bit operands (bf16 / fp16)");
    static_assert(!FRONT || PROJ, "the out-projection in front exists in the whole-block kernel only");
    constexpr int CK = C / 64;               // K tiles of GEMM 1
    constexpr int NB = C / 32;               // 32-row blocks of the output (transposed)
    constexpr int F = 4 * C;                 // hidden width
    constexpr int NCH = F / 64;              // hidden chunks of 64
    constexpr int NS1 = 4;                   // W1 ring: three tiles in flight behind the one being multiplied (one wave per SIMD: nobody
                                             // else covers an L2 round trip, a 2-stage ring stalls ~2k cycles per K tile)
    constexpr int W1B = 16384, W2B = C * 128;
    constexpr int OFF_W2 = NS1 * W1B, OFF_B = OFF_W2 + 2 * W2B;                 // bias slots: [2 stages][4 waves][128 floats]
    constexpr int NPW1 = 4, NPW2 = C / 8 / 4, NPB = 2;                          // DMA pieces per wave
    static_assert(OFF_B + 2 * 4 * 512 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    // XCD-aware block order (as gemm.hip): XCD x owns a contiguous run of token tiles
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = bid * 128;
    const int tok = wave * 32 + lrow;                     // this lane's token inside the block
    const int row = m0 + tok;

    // ---- X^T fragments of this wave's 32 tokens: the B operand of GEMM 1, resident in registers (20 k-steps x 16 bytes per lane)
    u32x4_t xq[CK * 4];
    const int xrow = (FRONT && p.front_rows > 0) ? row % p.front_rows : row;          // (FRONT under CFG sharing: both batch halves read the same attention output)
#pragma unroll
    for (int s_ = 0; s_ < CK * 4; ++s_) {
        xq[s_] = u32x4_t{0u, 0u, 0u, 0u};
        if (row < p.M) xq[s_] = *(const u32x4_t*)(p.x + (long long)xrow * p.ldx + s_ * 16 + lhalf * 8);
    }

    if constexpr (FRONT) {
        // ---- attn1.to_out in front: x1^T = Wo . att^T over five [C x 64] tiles of Wo streamed through the (still idle) W2 buffers + 40 KB of the W1 ring, exactly as the
        // proj_out contraction at the other end of the kernel; + bo + the sample's cross-attention vector + the residual tok; rounded, stored (the feed-forward's residual),
        // re-laid-out by v_permlane32_swap into the X^T fragments the rest of the kernel expects.  The kernel's own prologue starts behind it.
        const __amdgpu_buffer_rsrc_t rsWo = __builtin_amdgcn_make_buffer_rsrc((void*)p.wo, 0, (unsigned)(C * C * 2), 0x00020000);
        const int prow_ = lane >> 3;
        auto fbuf = [&](int kt) -> char* { return kt % 3 == 2 ? smem + W1B : smem + OFF_W2 + (kt % 3) * W2B; };
        auto issue_wo = [&](int kt, int q0, int q1) {
            char* base = fbuf(kt);
#pragma unroll
            for (int q = 0; q < NPW2; ++q) {
                if (q >= q0 && q < q1) {
                    const int rg = wave + 4 * q, r = rg * 8 + prow_;
                    const int off = r * C * 2 + (((lane & 7) ^ ((r >> 1) & 7)) * 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWo, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
                }
            }
        };
        issue_wo(0, 0, NPW2);
        issue_wo(1, 0, NPW2);
        float* const cl = (float*)smem;                          // [C] bo + ctx of this block's sample (the first bytes of the idle W1 ring)
        {
            const float* const cv = p.ctx ? p.ctx + (long long)(m0 / p.rows_per_sample0) * p.ldc : nullptr;
            for (int i = tid; i < C; i += 256) cl[i] = p.bo[i] + (cv ? cv[i] : 0.f);
        }
        const int brow0 = 16 * ((lrow >> 2) & 1) + 4 * (lrow >> 3) + (lrow & 3), bsw0 = (brow0 >> 1) & 7;
        f32x16_t a0[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) a0[nb][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < CK; ++kt) {
            if (kt + 1 < CK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW2) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const char* const wb = fbuf(kt) + brow0 * 128;
            u32x4_t af[2];
            af[0] = *(const u32x4_t*)(wb + (((0 + lhalf) ^ bsw0) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int it = kk * NB + nb, cur = it & 1;
                    if (it + 1 < 4 * NB) {
                        const int nkk = (it + 1) / NB, nnb = (it + 1) % NB;
                        af[cur ^ 1] = *(const u32x4_t*)(wb + nnb * 4096 + (((nkk * 2 + lhalf) ^ bsw0) << 4));
                    }
                    mma16<T>(a0[nb], af[cur], xq[kt * 4 + kk]);
                }
                if (kt + 2 < CK) issue_wo(kt + 2, kk * 3, kk * 3 + 3 < NPW2 ? kk * 3 + 3 : NPW2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // + column constants + residual, rounded as the unfused out-projection stores it; the row goes out (x1) and stays (xq)
        constexpr int PF0 = 4;
        u32x4_t r0[PF0][2];
        auto load_r0 = [&](int nb, u32x4_t* r) {
            if (p.res0 && row < p.M) {
                const u32x4_t* rp = (const u32x4_t*)(p.res0 + (long long)xrow * p.ldr0 + nb * 32 + lhalf * 16);
                r[0] = rp[0];
                r[1] = rp[1];
            }
        };
#pragma unroll
        for (int nb = 0; nb < PF0; ++nb) load_r0(nb, r0[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = nb * 32 + lhalf * 16;
            u32x4_t pk[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float f[8], v[8];
                const f32x4_t c0 = *(const f32x4_t*)(cl + col + 8 * h), c1 = *(const f32x4_t*)(cl + col + 8 * h + 4);
                if (p.res0) unpack16<T>(r0[nb % PF0][h], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (row < p.M) ? a0[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]) + (p.res0 ? f[e] : 0.f) : 0.f;
                pk[h] = pack16<T>(v);
                if (row < p.M) ((u32x4_t*)(p.x1 + (long long)row * p.ldx1 + col))[h] = pk[h];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
                xq[2 * nb][e] = sw[0];
                xq[2 * nb + 1][e] = sw[1];
            }
            if (nb + PF0 < NB) load_r0(nb + PF0, r0[nb % PF0]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // every wave is done with the Wo buffers and the column constants: the kernel's own prologue (W1 ring, W2 chunk 0, bias) may overwrite them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // LayerNorm of the token's row in registers (attention.py:231-233 `norm3` in front of the feed-forward): the lane pair (tok, half 0 / 1)
    // holds all C values of the row -- two-pass statistics in fp32 as rf_layernorm computes them, the normalised values rounded to bf16
    // as that pass stores them.  gamma rides in W1's columns, beta in b1 (host), so the separate pass (one read + one write of [M, C]) and
    // its launch are gone.
    if (p.ln_eps > 0.f) {
        float sum = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < CK * 4; ++s_) {
            float f[8];
            unpack16<T>(xq[s_], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += f[e];
        }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(sum), as_u32(sum), false, false);
            sum = as_f32(sw[0]) + as_f32(sw[1]);
        }
        const float mu = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < CK * 4; ++s_) {
            float f[8];
            unpack16<T>(xq[s_], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - mu; sq += d * d; }
        }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(sq), as_u32(sq), false, false);
            sq = as_f32(sw[0]) + as_f32(sw[1]);
        }
        const float rstd = 1.0f / sqrtf(sq / (float)C + p.ln_eps);
#pragma unroll
        for (int s_ = 0; s_ < CK * 4; ++s_) {
            float f[8];
            unpack16<T>(xq[s_], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (f[e] - mu) * rstd;
            xq[s_] = pack16<T>(f);
        }
    }

    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (unsigned)(2 * F * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (unsigned)(C * F * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.b1, 0, (unsigned)(2 * F * 4), 0x00020000);
    // a DMA piece = 8 rows x 128 B written lane-linearly; lane l lands on row (l >> 3), 16-byte position (l & 7) and fetches k-slot
    // (l & 7) ^ swizzle(row): the XOR swizzle lives on the source side
    const int prow = lane >> 3;
    auto kslot = [&](int r) { return ((lane & 7) ^ ((r >> 1) & 7)) * 16; };

    // ---- DMA issue helpers (every wave issues the same number of pieces: the counted vmcnt waits are wave-uniform)
    auto issue_w1 = [&](int g) {                          // global W1 tile index g = chunk * CK + kt
        const int c = g / CK, kt = g - c * CK;
        char* base = smem + (g % NS1) * W1B;
#pragma unroll
        for (int q = 0; q < NPW1; ++q) {
            const int rg = wave + 4 * q, r = rg * 8 + prow;                       // row inside the 128-row chunk
            const int off = (c * 128 + r) * C * 2 + kslot(r);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
        }
    };
    auto issue_w2 = [&](int c) {
        char* base = smem + OFF_W2 + (c & 1) * W2B;
#pragma unroll
        for (int q = 0; q < NPW2; ++q) {
            const int rg = wave + 4 * q, r = rg * 8 + prow;
            const int off = r * F * 2 + kslot(r);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, c * 128, 0, 0);
        }
    };
    auto issue_w1_piece = [&](int g, int q) {                  // piece q of W1 tile g (RF_FFN_SPREAD 2)
        const int c = g / CK, kt = g - c * CK;
        char* base = smem + (g % NS1) * W1B;
        const int rg = wave + 4 * q, r = rg * 8 + prow;
        const int off = (c * 128 + r) * C * 2 + kslot(r);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
    };
    auto issue_w2_part = [&](int c, int q0, int q1) {          // pieces [q0, q1) of W2 chunk c (RF_FFN_SPREAD)
        char* base = smem + OFF_W2 + (c & 1) * W2B;
#pragma unroll
        for (int q = 0; q < NPW2; ++q) {
            if (q >= q0 && q < q1) {
                const int rg = wave + 4 * q, r = rg * 8 + prow;
                const int off = r * F * 2 + kslot(r);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, c * 128, 0, 0);
            }
        }
    };
    auto issue_b1 = [&](int c) {                          // this wave's private copy of the chunk's 128 bias values (2 x 64 floats)
        char* base = smem + OFF_B + ((c & 1) * 4 + wave) * 512;
#pragma unroll
        for (int q = 0; q < NPB; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB1, (__attribute__((address_space(3))) void*)(base + q * 256), 4, (c * 128 + q * 64 + lane) * 4, 0, 0, 0);
    };

    // ---- prologue: W1 tiles 0 .. NS1-1, W2 chunk 0, bias chunk 0
#pragma unroll
    for (int g = 0; g < (RF_FFN_SPREAD == 2 ? NS1 - 1 : NS1); ++g) issue_w1(g);
    issue_w2(0);
    issue_b1(0);

    f32x16_t acc2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;

    // fragment addressing
    const int wsw = (lrow >> 1) & 7;
    const int brow = 16 * ((lrow >> 2) & 1) + 4 * (lrow >> 3) + (lrow & 3);     // permuted W2 row (see gemm.hip EPI = 1)
    const int bsw = (brow >> 1) & 7;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int c = 0; c < NCH; ++c) {
        f32x16_t acc1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[b][r] = 0.f;
        const bool last = c == NCH - 1;
        // ---- GEMM 1: S^T (128 packed rows x 32 tokens per wave) over K = C
#pragma unroll
        for (int kt = 0; kt < CK; ++kt) {
            const int g = c * CK + kt;
            const char* const w1base = smem + (g % NS1) * W1B + lrow * 128;
            u32x4_t wf[2][4];
#pragma unroll
            for (int b = 0; b < 4; ++b) wf[0][b] = *(const u32x4_t*)(w1base + b * 4096 + (((0 + lhalf) ^ wsw) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cur = kk & 1, nx = cur ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) wf[nx][b] = *(const u32x4_t*)(w1base + b * 4096 + ((((kk + 1) * 2 + lhalf) ^ wsw) << 4));
                }
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    mma16<T>(acc1[b], wf[cur][b], xq[kt * 4 + kk]);
#if RF_FFN_SPREAD == 2
                // one LDS-DMA piece behind every k-step's MFMAs (their issue slots hide under the matrix pipe): W1 tile g + 3 into the stage tile g - 1
                // left at the last barrier, the next chunk's W2 pieces behind k-steps 1 and 3, its bias behind the last k-step of K tile 0
                static_assert(NPW1 == 4, "one W1 piece per k-step");
                if (g + NS1 - 1 < NCH * CK) issue_w1_piece(g + NS1 - 1, kk);
                if (!last && (kk & 1)) issue_w2_part(c + 1, kt * (NPW2 / CK) + (kk >> 1), kt * (NPW2 / CK) + (kk >> 1) + 1);
                if (!last && kt == 0 && kk == 3) issue_b1(c + 1);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            // Tile g+1 must have landed.  Loads complete in order, so it suffices that at most the pieces issued AFTER it are still in
            // flight: W1(g+2), W1(g+3) (8 pieces) and, for kt = 1..3, the W2 / bias group of the next chunk issued behind W1(g0+4).
            // The last chunk drains everything (the ring is running empty there).
#if RF_FFN_SPREAD == 2
            // Behind the last piece of W1(g+1) -- issued at k-step 3 of iteration g - 2 -- lie that iteration's second W2 piece (+ bias) and two whole
            // iterations of 6 pieces (+ 2 for a bias): 13, + 2 when one of the three iterations was a K tile 0 (kt in {0, 1, 2})
            static_assert(NPW2 / CK == 2 && NPW2 % CK == 0, "two W2 pieces per K tile");
            if (last) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (kt <= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(13 + NPB) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(13) : "memory");
            __builtin_amdgcn_s_barrier();                 // every wave is done with tile g: its stage is refilled during the next iteration
#elif RF_FFN_SPREAD
            // The next chunk's W2 pieces go out two per K tile instead of ten behind K tile 0 (one wave per SIMD: a burst of 16 LDS-DMA pieces stalls
            // the wave's issue for ~1.5 k cycles with nothing else to feed the matrix pipe).  Behind W1(g+1) -- issued first thing three iterations
            // ago -- lie that iteration's W2 part (+ bias) and two whole iterations: 3 x 2 + 2 x 4 pieces, + 2 when one of them was a K tile 0.
            constexpr int PW2 = NPW2 / CK;
            static_assert(NPW2 % CK == 0, "W2 pieces spread evenly over the K tiles");
            if (last) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (kt >= 1 && kt <= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NPW1 + 3 * PW2 + NPB) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NPW1 + 3 * PW2) : "memory");
            __builtin_amdgcn_s_barrier();                 // every wave is done with tile g: its stage may be refilled
            if (g + NS1 < NCH * CK) issue_w1(g + NS1);
            if (!last) {
                issue_w2_part(c + 1, kt * PW2, (kt + 1) * PW2);          // buffer (c+1)&1 was last read by GEMM 2 of chunk c-1 (barrier at its end)
                if (kt == 0) issue_b1(c + 1);
            }
#else
            if (last) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (kt >= 1 && kt <= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NPW1 + NPW2 + NPB) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NPW1) : "memory");
            __builtin_amdgcn_s_barrier();                 // every wave is done with tile g: its stage may be refilled
            if (g + NS1 < NCH * CK) issue_w1(g + NS1);
            if (kt == 0 && !last) {
                issue_w2(c + 1);                          // buffer (c+1)&1 was last read by GEMM 2 of chunk c-1 (barrier at its end)
                issue_b1(c + 1);
            }
#endif
        }
        // ---- GEGLU in registers: h = (value + bv) * gelu(gate + bg), packed to the B fragments of GEMM 2
        u32x4_t hb[4];
        {
            const float* const bl = (const float*)(smem + OFF_B + ((c & 1) * 4 + wave) * 512);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float h[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4_t bv = *(const f32x4_t*)(bl + pr * 64 + 8 * q4 + 4 * lhalf), bg = *(const f32x4_t*)(bl + pr * 64 + 32 + 8 * q4 + 4 * lhalf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q4 + e;
                        h[r] = (acc1[2 * pr][r] + bv[e]) * gelu_sigmoid5(acc1[2 * pr + 1][r] + bg[e]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hb[2 * pr + s2][e] = pack2<T>(h[8 * s2 + 2 * e], h[8 * s2 + 2 * e + 1]);
            }
        }
        // ---- GEMM 2: O^T += W2c . H^T  (the W2 chunk was issued a whole chunk ago; the waits of K tiles 0 and 4 covered it)
        {
            const char* const w2base = smem + OFF_W2 + (c & 1) * W2B + brow * 128;
            u32x4_t af[2];
            af[0] = *(const u32x4_t*)(w2base + (((0 + lhalf) ^ bsw) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int it = kk * NB + nb, cur = it & 1;
                    if (it + 1 < 4 * NB) {
                        const int nkk = (it + 1) / NB, nnb = (it + 1) % NB;
                        af[cur ^ 1] = *(const u32x4_t*)(w2base + nnb * 4096 + (((nkk * 2 + lhalf) ^ bsw) << 4));
                    }
                    mma16<T>(acc2[nb], af[cur], hb[kk]);
                }
            }
        }
        // this chunk's W2 buffer and bias slot are free once every wave is here (they are refilled at K tile 0 of chunk c+1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: lane (token, half) holds output columns 32*nb + 16*half + r; b2 (and proj_out's bias) parked in LDS (the operand stages are dead)
    float* const b2l = (float*)smem;
    constexpr bool proj = PROJ;
    for (int i = tid; i < C; i += 256) {
        b2l[i] = p.b2[i];
        if (proj) b2l[C + i] = p.bpo[i];
    }
    // proj_out: K tile kt of Wpo (C rows x 64 k) has the geometry of a W2 chunk and takes its buffers; tiles 0, 1 and 2 go out now
    const __amdgpu_buffer_rsrc_t rsWp = __builtin_amdgcn_make_buffer_rsrc((void*)(proj ? p.wpo : p.w2), 0, (unsigned)(C * C * 2), 0x00020000);
    // (three buffers: the two W2 buffers and 40 KB of the dead W1 ring behind the bias / statistics scratch -- three tiles in flight: a K tile's 40
    //  MFMAs per wave are shorter than an L2 round trip)
#ifndef RF_WP_BUFS
#define RF_WP_BUFS (RF_FFN_SPREAD == 2 ? 3 : 2)          // (bursts through two or three buffers: measured equal, profiles/r05i_fused_tail_ab.txt)
#endif
    constexpr int WPB = RF_WP_BUFS;
    auto wp_buf = [&](int kt) -> char* { return kt % WPB == 2 ? smem + W1B : smem + OFF_W2 + (kt % WPB) * W2B; };
    static_assert(W1B >= 4096 + 2 * 4 * C * 4 && W1B + W2B <= NS1 * W1B, "proj_out: third buffer inside the W1 ring, behind b2l / gcs");
    auto issue_wp = [&](int kt) {
        char* base = wp_buf(kt);
#pragma unroll
        for (int q = 0; q < NPW2; ++q) {
            const int rg = wave + 4 * q, r = rg * 8 + prow;
            const int off = r * C * 2 + kslot(r);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWp, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
        }
    };
    if (proj) {
#pragma unroll
        for (int kt = 0; kt < (RF_FFN_SPREAD == 2 ? 2 : WPB); ++kt) issue_wp(kt);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!proj) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    constexpr int PFD = 4;
    u32x4_t rq[PFD][2];
    auto load_res = [&](int nb, u32x4_t* r) {
        if (p.res && row < p.M) {
            const u32x4_t* rp = (const u32x4_t*)(p.res + (long long)row * p.ldr + nb * 32 + lhalf * 16);
            r[0] = rp[0];
            r[1] = rp[1];
        }
    };
    if constexpr (!proj) {
#pragma unroll
        for (int nb = 0; nb < PFD; ++nb) load_res(nb, rq[nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = nb * 32 + lhalf * 16;
            if (row < p.M) {
                bf16_t* dst = p.out + (long long)row * p.ldo + col;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float f[8], v[8];
                    const f32x4_t c0 = *(const f32x4_t*)(b2l + col + 8 * h), c1 = *(const f32x4_t*)(b2l + col + 8 * h + 4);
                    if (p.res) unpack16<T>(rq[nb % PFD][h], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc2[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]) + (p.res ? f[e] : 0.f);
                    ((u32x4_t*)dst)[h] = pack16<T>(v);
                }
            }
            if (nb + PFD < NB) load_res(nb + PFD, rq[nb % PFD]);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    } else {

    // ---- proj_out fused.  (a) the feed-forward's output row (bias + residual, rounded to bf16 as the unfused path stores it) becomes the B operand of
    // one more GEMM: lane (token, half h) holds columns 32 nb + 16 h + r of block nb; the fragment of k-step s = 2 nb + t wants columns 16 s + 8 h' .. + 8
    // in half h' -- one v_permlane32_swap per packed register pair moves the two middle quarters between the halves (natural K order: Wpo needs no
    // column permutation)
    u32x4_t hb3[2 * NB];
#pragma unroll
    for (int nb = 0; nb < PFD; ++nb) load_res(nb, rq[nb]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = nb * 32 + lhalf * 16;
        u32x4_t pk[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float f[8], v[8];
            const f32x4_t c0 = *(const f32x4_t*)(b2l + col + 8 * h), c1 = *(const f32x4_t*)(b2l + col + 8 * h + 4);
            if (p.res) unpack16<T>(rq[nb % PFD][h], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (row < p.M) ? acc2[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]) + (p.res ? f[e] : 0.f) : 0.f;
            pk[h] = pack16<T>(v);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(pk[0][e], pk[1][e], false, false);
            hb3[2 * nb][e] = sw[0];
            hb3[2 * nb + 1][e] = sw[1];
        }
        if (nb + PFD < NB) load_res(nb + PFD, rq[nb % PFD]);
        __builtin_amdgcn_sched_barrier(0);
    }
    // (b) Z^T = Wpo . X2^T : CK K tiles of 64
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
    auto issue_wp_part = [&](int kt, int q0, int q1) {
        char* base = wp_buf(kt);
#pragma unroll
        for (int q = 0; q < NPW2; ++q) {
            if (q >= q0 && q < q1) {
                const int rg = wave + 4 * q, r = rg * 8 + prow;
                const int off = r * C * 2 + kslot(r);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWp, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
            }
        }
    };
#pragma unroll
    for (int kt = 0; kt < CK; ++kt) {
#if RF_FFN_SPREAD == 2
        // three buffers, the pieces of tile kt + 2 between the MFMAs of tile kt (buffer (kt + 2) % 3 was last read by tile kt - 1: every wave has left it
        // at the barrier below) -- one barrier per tile, no burst
        static_assert(WPB == 3, "interleaved Wpo stream: three buffers");
        if (kt + 1 < CK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW2) : "memory");          // tile kt landed; tile kt + 1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#else
        // tile kt landed; the (up to two) tiles issued behind it may still fly
        if (kt == 0 || kt + 1 >= CK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt + WPB - 1 < CK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WPB - 1) * NPW2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW2) : "memory");
        __builtin_amdgcn_s_barrier();
#endif
        {
            const char* const w2base = wp_buf(kt) + brow * 128;
            u32x4_t af[2];
            af[0] = *(const u32x4_t*)(w2base + (((0 + lhalf) ^ bsw) << 4));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int it = kk * NB + nb, cur = it & 1;
                    if (it + 1 < 4 * NB) {
                        const int nkk = (it + 1) / NB, nnb = (it + 1) % NB;
                        af[cur ^ 1] = *(const u32x4_t*)(w2base + nnb * 4096 + (((nkk * 2 + lhalf) ^ bsw) << 4));
                    }
                    mma16<T>(acc2[nb], af[cur], hb3[kt * 4 + kk]);
                }
#if RF_FFN_SPREAD == 2
                if (kt + 2 < CK) issue_wp_part(kt + 2, kk * 3, kk * 3 + 3 < NPW2 ? kk * 3 + 3 : NPW2);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
#if RF_FFN_SPREAD != 2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // every wave is done with this buffer
        if (kt + WPB < CK) issue_wp(kt + WPB);
#endif
    }
    // (c) + bpo + the transformer's input (attention.py:289 `return x + x_in`; the CFG-shared first block: both batch halves read the same rows),
    // 16-byte stores, GroupNorm partial sums of the values as stored
    const float* const bpl = b2l + C;
    const int rrow = p.res2_rows > 0 ? row % p.res2_rows : row;
    auto load_res2 = [&](int nb, u32x4_t* r) {
        if (p.res2 && row < p.M) {
            const u32x4_t* rp = (const u32x4_t*)(p.res2 + (long long)rrow * p.ldr2 + nb * 32 + lhalf * 16);
            r[0] = rp[0];
            r[1] = rp[1];
        }
    };
    const bool gn_on = p.gn_rows > 0;
    float* const gcs = (float*)(smem + 4096);                 // [2][4 waves][C] column sums / sums of squares per wave
#pragma unroll
    for (int nb = 0; nb < PFD; ++nb) load_res2(nb, rq[nb]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = nb * 32 + lhalf * 16;
        float y[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) y[e] = 0.f;
        if (row < p.M) {
            bf16_t* dst = p.out + (long long)row * p.ldo + col;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float f[8], v[8];
                const f32x4_t c0 = *(const f32x4_t*)(bpl + col + 8 * h), c1 = *(const f32x4_t*)(bpl + col + 8 * h + 4);
                if (p.res2) unpack16<T>(rq[nb % PFD][h], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc2[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]) + (p.res2 ? f[e] : 0.f);
                const u32x4_t wv = pack16<T>(v);
                ((u32x4_t*)dst)[h] = wv;
                if (gn_on) {
                    unpack16<T>(wv, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[8 * h + e] = f[e];
                }
            }
        }
        if (nb + PFD < NB) load_res2(nb + PFD, rq[nb % PFD]);
        if (gn_on) {
            float gx[32];
#pragma unroll
            for (int k = 0; k < 16; ++k) { gx[k] = y[k]; gx[16 + k] = y[k] * y[k]; }
            halfwave_reduce_scatter32(gx, lrow);          // lane lrow: total over the wave's 32 tokens of column lrow (sums) / lrow - 16 (squares)
            gcs[((lrow >> 4) * 4 + wave) * C + nb * 32 + lhalf * 16 + (lrow & 15)] = gx[0];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (gn_on) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 64) {
            const int c = tid >> 5, g = tid & 31;
            if (p.gn_part[c] && m0 < p.M) {
                const int cpg = p.gn_cpg[c], base = p.gn_coff[c];            // consumer channel of output column 0
                const int lo = max(0, g * cpg - base), hi = min(C, (g + 1) * cpg - base);
                double sa = 0.0, sq = 0.0;
                for (int k = lo; k < hi; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa += (double)gcs[r * C + k]; sq += (double)gcs[(4 + r) * C + k]; }
                const int b = m0 / p.gn_rows, mt = (m0 - b * p.gn_rows) / 128;
                double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + mt) * 32 + g) * 2;
                o[0] = sa;
                o[1] = sq;
            }
        }
    }
    }          // (PROJ)
}

}  // namespace rf

static int ffn_launch(const rf::FfnParams& p, int C, void* stream, int dtype = RF_BF16) {
    using namespace rf;
    RF_CHECK(p.x && p.w1 && p.b1 && p.w2 && p.b2 && p.out && p.M > 0, "rf_ffn_geglu: bad arguments");
    RF_CHECK(C == 320, "rf_ffn_geglu: built for C = 320 (the 64x64 level), got %d", C);
    RF_CHECK(dtype == RF_BF16 || dtype == RF_F16, "rf_ffn_block: dtype %d (RF_BF16 or RF_F16)", dtype);
    RF_CHECK(p.ldx % 8 == 0 && p.ldo % 8 == 0 && (!p.res || p.ldr % 8 == 0) && (!p.res2 || p.ldr2 % 8 == 0), "rf_ffn_geglu: row pitches must be multiples of 8");
    RF_CHECK(((uintptr_t)p.x | (uintptr_t)p.w1 | (uintptr_t)p.w2 | (uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.b1 | (uintptr_t)p.b2 | (uintptr_t)p.wpo | (uintptr_t)p.bpo |
              (uintptr_t)p.res2) % 16 == 0, "rf_ffn_geglu: operands must be 16-byte aligned");
    RF_CHECK((long long)p.M * p.ldx * 2 < 0x7fff0000LL, "rf_ffn_geglu: x too large for 31-bit byte offsets");
    RF_CHECK(!p.wpo || p.bpo, "rf_ffn_block: proj_out needs its bias");
    RF_CHECK(p.wpo || (!p.res2 && p.gn_rows == 0), "rf_ffn_block: res2 / GroupNorm statistics belong to the fused proj_out");
    if (p.gn_rows > 0) {
        RF_CHECK(p.gn_rows % 128 == 0 && p.M % p.gn_rows == 0, "rf_ffn_block: fused GroupNorm statistics need gn_rows (%d) to be a multiple of the 128-token block and M a multiple of it", p.gn_rows);
        for (int c = 0; c < 2; ++c)
            RF_CHECK(!p.gn_part[c] || (p.gn_cpg[c] > 0 && p.gn_slot[c] >= 0 && p.gn_slot[c] + p.gn_rows / 128 <= p.gn_nch[c]),
                     "rf_ffn_block: GroupNorm consumer %d: cpg=%d slot=%d needs %d slots of %d", c, p.gn_cpg[c], p.gn_slot[c], p.gn_rows / 128, p.gn_nch[c]);
    }
    constexpr int smem = 4 * 16384 + 2 * 320 * 128 + 2 * 4 * 512;
    RF_CHECK(!p.wo || (p.wpo && p.bo && p.x1 && p.res == p.x1 && p.rows_per_sample0 > 0 && p.rows_per_sample0 % 128 == 0 && p.ldx1 % 8 == 0 && (!p.res0 || p.ldr0 % 8 == 0) &&
                       ((uintptr_t)p.wo | (uintptr_t)p.x1 | (uintptr_t)p.res0 | (uintptr_t)p.bo | (uintptr_t)p.ctx) % 16 == 0 && p.ldc % 4 == 0 && (p.front_rows == 0 || p.front_rows % 128 == 0)),
             "rf_ffn_block: the out-projection in front needs the fused tail (wpo), bo, x1 == residual, rows_per_sample0 / front_rows multiples of the 128-token block, aligned operands");
#define RF_FFN_LAUNCH(PROJ_, T_, FRONT_)                                                               \
    {                                                                                                  \
        auto k = ffn_geglu_kernel<320, PROJ_, T_, FRONT_>;                                             \
        RF_RAISE_LDS(k, smem, "rf_ffn_geglu");                                                         \
        hipLaunchKernelGGL(k, dim3((p.M + 127) / 128), dim3(256), smem, (hipStream_t)stream, p);       \
    }
    if (dtype == RF_F16) { if (p.wo) RF_FFN_LAUNCH(true, f16_t, true) else if (p.wpo) RF_FFN_LAUNCH(true, f16_t, false) else RF_FFN_LAUNCH(false, f16_t, false) }
    else { if (p.wo) RF_FFN_LAUNCH(true, bf16_t, true) else if (p.wpo) RF_FFN_LAUNCH(true, bf16_t, false) else RF_FFN_LAUNCH(false, bf16_t, false) }
#undef RF_FFN_LAUNCH
    RF_LAUNCH_CHECK("rf_ffn_geglu");
    return 0;
}

extern "C" int rf_ffn_geglu(const void* x, int ldx, const void* w1p, const float* b1p, const void* w2q, const float* b2, const void* residual,
                            int ldr, void* out, int ldo, int M, int C, float ln_eps, void* stream) {
    using namespace rf;
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)x; p.ldx = ldx; p.w1 = (const bf16_t*)w1p; p.b1 = b1p; p.w2 = (const bf16_t*)w2q; p.b2 = b2;
    p.res = (const bf16_t*)residual; p.ldr = ldr; p.out = (bf16_t*)out; p.ldo = ldo; p.M = M; p.ln_eps = ln_eps;
    return ffn_launch(p, C, stream);
}

extern "C" int rf_ffn_block(const rf_ffn_desc* d, void* stream) {
    using namespace rf;
    RF_CHECK(d != nullptr, "rf_ffn_block: null descriptor");
    FfnParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16_t*)d->x; p.ldx = d->ldx; p.w1 = (const bf16_t*)d->w1p; p.b1 = d->b1p; p.w2 = (const bf16_t*)d->w2q; p.b2 = d->b2;
    p.res = (const bf16_t*)d->residual; p.ldr = d->ldr; p.out = (bf16_t*)d->out; p.ldo = d->ldo; p.M = d->M; p.ln_eps = d->ln_eps;
    p.wpo = (const bf16_t*)d->wpo; p.bpo = d->bpo; p.res2 = (const bf16_t*)d->res2; p.ldr2 = d->ldr2; p.res2_rows = d->res2_rows;
    p.gn_rows = (d->gn_part0 || d->gn_part1) ? d->gn_rows : 0;
    p.gn_part[0] = d->gn_part0; p.gn_cpg[0] = d->gn_cpg0; p.gn_coff[0] = d->gn_coff0; p.gn_slot[0] = d->gn_slot0; p.gn_nch[0] = d->gn_nchunks0;
    p.gn_part[1] = d->gn_part1; p.gn_cpg[1] = d->gn_cpg1; p.gn_coff[1] = d->gn_coff1; p.gn_slot[1] = d->gn_slot1; p.gn_nch[1] = d->gn_nchunks1;
    p.wo = (const bf16_t*)d->wo; p.bo = d->bo; p.ctx = d->ctx; p.ldc = d->ldc; p.rows_per_sample0 = d->rows_per_sample0;
    p.res0 = (const bf16_t*)d->res0; p.ldr0 = d->ldr0; p.front_rows = d->front_rows; p.x1 = (bf16_t*)d->x1; p.ldx1 = d->ldx1;
    return ffn_launch(p, d->C, stream, d->dtype == 0 ? RF_BF16 : d->dtype);
}
