// The token-resident FRONT of a SpatialTransformer block at C = 320 (attention.py:262-266, 276-279, 231-233, 239: `norm` -> `proj_in` -> `norm1` -> to_q / to_k / to_v):
//
//     tok = x W'_s^T + r_s                      proj_in with the GroupNorm `norm` folded into per-sample weights (rf_groupnorm_fold_linear)
//     qkv = LN(tok) Wqkv'^T + b'                norm1 in registers (gamma folded into Wqkv's columns, Wqkv beta into b': ops.fold_layernorm_geglu's scheme)
//
// in ONE kernel: the K = 320 launches of the 64x64 level (proj_in 65536 x 320 x 320, qkv 65536 x 960 x 320) spend 75 % of their time in launch, prologue and a
// memory-bound epilogue around 5 K tiles of work; here tok is written once (the attention's out-projection needs it as its residual) and never re-read,
// the LayerNorm row statistics come from the registers that hold the row, and the four [320 x 320] weight panels stream through one LDS ring behind each other.
//
// Everything is computed transposed, tokens on lanes, exactly as the tail kernel (ffn.hip): block = 4 waves = 128 tokens, one wave per SIMD;
//   T^T[n, tok] = W'[n, :] . X^T[:, tok]       A = weight rows from LDS (read in the permuted row order of gemm.hip's direct epilogue: a lane ends with 16 contiguous
//                                              output columns of its token), B = the token's row in registers
// X^T fragments come straight from global memory (natural K order); the normalised tok row is re-laid-out as the B operand of the second contraction with one
// v_permlane32_swap per packed register pair (ffn.hip, proj_out).  LDS: a ring of three [320 rows x 64 k] weight tiles (40 KB each), two tiles in flight behind
// the one being multiplied, one barrier per tile, the pieces of tile g + 2 issued between the MFMAs of tile g; the per-sample vector and the qkv bias behind it.
#include "common.h"

namespace rf {

struct AttnInParams {
    const uint16_t* x; int ldx;              // [M][ldx] block input (un-normalised residual stream)
    const uint16_t* wpi; long long w_ps;     // [S][C][C] per-sample folded proj_in weights, sample stride in elements (0: one W)
    const float* rv; int ldv;                // [S][C] per-sample vector (bias + W beta - W' mean), fp32
    int rows_per_sample;
    uint16_t* tok; int ldt;                  // [M][ldt] proj_in output (stored: the out-projection's residual)
    const uint16_t* wqkv; const float* bqkv; // [3C][C] with norm1's gamma folded in, [3C] = Wqkv beta
    uint16_t* qkv; int ldq;                  // [M][ldq], columns [0, 3C)
    int M;
    float ln_eps;
};

template <int C, typename T>
__global__ __launch_bounds__(256, 1) void attn_in_kernel(const AttnInParams p) {
    static_assert(sizeof(T) == 2 && C % 64 == 0, "16-bit operands, whole K tiles");
    constexpr int CK = C / 64;               // K tiles per contraction
    constexpr int NB = C / 32;               // 32-row blocks of a [C]-wide output
    constexpr int NG = 4 * CK;               // weight tiles of the whole block: proj_in, then q / k / v
    constexpr int WB = C * 128;              // bytes of one [C rows x 64 k] tile
    constexpr int NPW = C / 8 / 4;           // one-KiB DMA pieces per wave and tile
    constexpr int OFF_V = 3 * WB;            // [C] per-sample vector, [3C] qkv bias (floats)
    static_assert(OFF_V + 4 * C * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = bid * 128;
    const int row = m0 + wave * 32 + lrow;
    const int smp = m0 / p.rows_per_sample;               // (host: rows_per_sample is a multiple of the 128-token block)

    // ---- X^T fragments of this wave's 32 tokens (B operand of the first contraction)
    u32x4_t xq[CK * 4];
#pragma unroll
    for (int s_ = 0; s_ < CK * 4; ++s_) {
        xq[s_] = u32x4_t{0u, 0u, 0u, 0u};
        if (row < p.M) xq[s_] = *(const u32x4_t*)(p.x + (long long)row * p.ldx + s_ * 16 + lhalf * 8);
    }

    const uint16_t* const wpi = p.wpi + (long long)smp * p.w_ps;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)wpi, 0, (unsigned)(C * C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.wqkv, 0, (unsigned)(3 * C * C * 2), 0x00020000);
    const int prow = lane >> 3;
    auto kslot = [&](int r) { return ((lane & 7) ^ ((r >> 1) & 7)) * 16; };
    // weight tile g: g < CK -> K tile g of W'_s; else chunk (g - CK) / CK (q, k, v) and K tile (g - CK) % CK of Wqkv'
    auto issue_part = [&](int g, int q0, int q1) {
        char* const base = smem + (g % 3) * WB;
        const bool first = g < CK;
        const int ch = first ? 0 : (g - CK) / CK, kt = first ? g : (g - CK) % CK;
#pragma unroll
        for (int q = 0; q < NPW; ++q) {
            if (q >= q0 && q < q1) {
                const int rg = wave + 4 * q, r = rg * 8 + prow;
                const int off = (ch * C + r) * C * 2 + kslot(r);
                if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + rg * 1024), 16, off, kt * 128, 0, 0);
            }
        }
    };
    issue_part(0, 0, NPW);
    issue_part(1, 0, NPW);
    // per-sample vector and qkv bias -> LDS (plain loads: tiny)
    float* const vl = (float*)(smem + OFF_V);
    for (int i = tid; i < 4 * C; i += 256) vl[i] = i < C ? p.rv[(long long)smp * p.ldv + i] : p.bqkv[i - C];

    const int brow = 16 * ((lrow >> 2) & 1) + 4 * (lrow >> 3) + (lrow & 3);     // permuted weight row (gemm.hip EPI = 1)
    const int bsw = (brow >> 1) & 7;
    f32x16_t acc[NB];
    u32x4_t hb[CK * 4];                      // LN(tok)^T fragments: the B operand of the q / k / v contractions

    // one [C x 64] tile against B fragments bq[kt * 4 + kk]; the pieces of tile g + 2 go out between the MFMAs (buffer (g + 2) % 3 was last read by tile g - 1:
    // every wave has left it at this tile's barrier)
    // `stores`: this wave issued 2 NB output stores since tile g + 1's pieces (the tile follows an epilogue, every lane of the wave in range): they are YOUNGER than the
    // pieces, so "at most NPW + 2 NB outstanding" already says tile g has landed -- without it the wait would sit out half of the stores' trip to L2
    const bool wave_full = m0 + wave * 32 + 31 < p.M;          // (wave-uniform; a ragged wave may have skipped stores: the strict count is always safe)
    auto tile_mma = [&](int g, const u32x4_t* bq, int kt, bool stores) {
        if (g + 1 >= NG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (stores && wave_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + 2 * NB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");          // tile g landed; tile g + 1 may still fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* const wbase = smem + (g % 3) * WB + brow * 128;
        u32x4_t af[2];
        af[0] = *(const u32x4_t*)(wbase + (((0 + lhalf) ^ bsw) << 4));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int it = kk * NB + nb, cur = it & 1;
                if (it + 1 < 4 * NB) {
                    const int nkk = (it + 1) / NB, nnb = (it + 1) % NB;
                    af[cur ^ 1] = *(const u32x4_t*)(wbase + nnb * 4096 + (((nkk * 2 + lhalf) ^ bsw) << 4));
                }
                mma16<T>(acc[nb], af[cur], bq[kt * 4 + kk]);
            }
            if (g + 2 < NG) issue_part(g + 2, kk * 3, kk * 3 + 3 < NPW ? kk * 3 + 3 : NPW);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    };

    // ---- contraction 1: tok^T = W'_s . X^T
    zero_acc();
#pragma unroll
    for (int kt = 0; kt < CK; ++kt) tile_mma(kt, xq, kt, false);
    // ---- + r_s, round to the storage type (what the out-projection will read back as its residual), store; LayerNorm of the STORED row in registers
    {
        u32x4_t pk[NB][2];
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = nb * 32 + lhalf * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
                const f32x4_t c0 = *(const f32x4_t*)(vl + col + 8 * h), c1 = *(const f32x4_t*)(vl + col + 8 * h + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]);
                pk[nb][h] = pack16<T>(v);
                if (row < p.M) ((u32x4_t*)(p.tok + (long long)row * p.ldt + col))[h] = pk[nb][h];
                unpack16<T>(pk[nb][h], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { acc[nb][8 * h + e] = v[e]; sum += v[e]; }          // the values as stored replace the (dead) accumulators
            }
        }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(sum), as_u32(sum), false, false);
            sum = as_f32(sw[0]) + as_f32(sw[1]);
        }
        const float mu = sum / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = acc[nb][e] - mu; sq += d * d; }
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(sq), as_u32(sq), false, false);
            sq = as_f32(sw[0]) + as_f32(sw[1]);
        }
        const float rstd = 1.0f / sqrtf(sq / (float)C + p.ln_eps);
        // normalised values rounded as rf_layernorm stores them, then re-laid-out as B fragments: lane (token, half h) holds columns 32 nb + 16 h + r; the fragment
        // of k-step s = 2 nb + t wants columns 16 s + 8 h' .. + 8 in half h' -- one v_permlane32_swap per packed register pair (ffn.hip, proj_out)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            u32x4_t q2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (row < p.M) ? (acc[nb][8 * h + e] - mu) * rstd : 0.f;
                q2[h] = pack16<T>(v);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(q2[0][e], q2[1][e], false, false);
                hb[2 * nb][e] = sw[0];
                hb[2 * nb + 1][e] = sw[1];
            }
        }
    }
    // ---- contractions 2-4: q / k / v chunks of qkv^T = Wqkv' . LN(tok)^T, + b', 16-byte stores
#pragma unroll 1
    for (int ch = 0; ch < 3; ++ch) {
        zero_acc();
#pragma unroll
        for (int kt = 0; kt < CK; ++kt) tile_mma(CK + ch * CK + kt, hb, kt, kt == 0);
        const float* const bl = vl + C + ch * C;
        if (row < p.M) {
            uint16_t* const dst = p.qkv + (long long)row * p.ldq + ch * C;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = nb * 32 + lhalf * 16;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8];
                    const f32x4_t c0 = *(const f32x4_t*)(bl + col + 8 * h), c1 = *(const f32x4_t*)(bl + col + 8 * h + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]);
                    ((u32x4_t*)(dst + col))[h] = pack16<T>(v);
                }
            }
        }
    }
}

}  // namespace rf

extern "C" int rf_attn_in(const rf_attn_in_desc* d, void* stream) {
    using namespace rf;
    RF_CHECK(d != nullptr, "rf_attn_in: null descriptor");
    RF_CHECK(d->x && d->wpi && d->rowvec && d->tok && d->wqkv && d->bqkv && d->qkv && d->M > 0, "rf_attn_in: bad arguments");
    RF_CHECK(d->C == 320, "rf_attn_in: built for C = 320 (the 64x64 level), got %d", d->C);
    const int dtype = d->dtype == 0 ? RF_BF16 : d->dtype;
    RF_CHECK(dtype == RF_BF16 || dtype == RF_F16, "rf_attn_in: dtype %d (RF_BF16 or RF_F16)", d->dtype);
    RF_CHECK(d->rows_per_sample > 0 && d->rows_per_sample % 128 == 0 && d->M % d->rows_per_sample == 0,
             "rf_attn_in: rows_per_sample (%d) must be a multiple of the 128-token block and divide M (%d)", d->rows_per_sample, d->M);
    RF_CHECK(d->ldx % 8 == 0 && d->ldt % 8 == 0 && d->ldq % 8 == 0 && d->ldx >= d->C && d->ldt >= d->C && d->ldq >= 3 * d->C && d->ldv >= d->C && d->w_sample_stride % 8 == 0,
             "rf_attn_in: row pitches must be multiples of 8 and cover the rows");
    RF_CHECK(((uintptr_t)d->x | (uintptr_t)d->wpi | (uintptr_t)d->tok | (uintptr_t)d->wqkv | (uintptr_t)d->qkv | (uintptr_t)d->rowvec | (uintptr_t)d->bqkv) % 16 == 0,
             "rf_attn_in: operands must be 16-byte aligned");
    RF_CHECK(d->ln_eps > 0.f, "rf_attn_in: ln_eps must be positive");
    AttnInParams p;
    p.x = (const uint16_t*)d->x; p.ldx = d->ldx; p.wpi = (const uint16_t*)d->wpi; p.w_ps = d->w_sample_stride; p.rv = d->rowvec; p.ldv = d->ldv;
    p.rows_per_sample = d->rows_per_sample; p.tok = (uint16_t*)d->tok; p.ldt = d->ldt; p.wqkv = (const uint16_t*)d->wqkv; p.bqkv = d->bqkv;
    p.qkv = (uint16_t*)d->qkv; p.ldq = d->ldq; p.M = d->M; p.ln_eps = d->ln_eps;
    constexpr int smem = 3 * 320 * 128 + 4 * 320 * 4;
    const dim3 grid((d->M + 127) / 128);
    if (dtype == RF_F16) {
        auto k = attn_in_kernel<320, f16_t>;
        RF_RAISE_LDS(k, smem, "rf_attn_in");
        hipLaunchKernelGGL(k, grid, dim3(256), smem, (hipStream_t)stream, p);
    } else {
        auto k = attn_in_kernel<320, bf16_t>;
        RF_RAISE_LDS(k, smem, "rf_attn_in");
        hipLaunchKernelGGL(k, grid, dim3(256), smem, (hipStream_t)stream, p);
    }
    RF_LAUNCH_CHECK("rf_attn_in");
    return 0;
}
