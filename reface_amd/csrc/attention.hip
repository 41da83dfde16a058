// Fused multi-head attention  out = softmax(scale * q k^T) v  (flash-style, scores never leave the CU).
//
// Formulated "transposed" so that every per-query quantity is lane-local on a 64-wide wavefront:
//   S^T[kv, q]  = K[kv, :] . Q[q, :]      (MFMA A = K rows from LDS, B = Q fragments held in VGPRs)
//   O^T[dv, q] += V^T[dv, kv] . P^T[kv, q] (MFMA A = V^T rows from LDS, B = P straight from the S accumulators)
// In the 32x32 accumulator layout lane l owns query column q = l & 31, so the running max / sum /
// rescale factors are per-lane scalars, the softmax needs one cross-half shuffle per KV tile, and
// P feeds the second MFMA without leaving registers (the K-order permutation this implies is
// absorbed by the order in which V^T is written to LDS).
// Same byte geometry for both storage types (16-byte fragments):
//   bf16: v_mfma_f32_32x32x16_bf16, fp16: v_mfma_f32_32x32x16_f16 (RF_F16: the same kernels for f16_t), fp32: 4 x v_mfma_f32_32x32x2_f32 (exact fp32).
// Block = 4 waves = 128 queries of one (batch, head); KV tile = 64 keys.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace rf {
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <typename T> struct AttnMma;
template <> struct AttnMma<bf16_t> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
    // D = A B + C with C in registers of its own (C stays live: no copy in front of the MFMA)
    __device__ static __forceinline__ f32x16_t mma_c(const u32x4_t& a, const u32x4_t& b, const f32x16_t& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct AttnMma<f16_t> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) { mma16<f16_t>(acc, a, b); }
    __device__ static __forceinline__ f32x16_t mma_c(const u32x4_t& a, const u32x4_t& b, const f32x16_t& c) {
        f32x16_t r = c;
        mma16<f16_t>(r, a, b);
        return r;
    }
};
template <> struct AttnMma<float> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as_f32( a[q]), as_f32( b[q]), acc, 0, 0, 0);
    }
};

struct AttnParams {
    const void* q; const void* k; const void* v; void* out;
    int heads, d, Nq, Nk, ldq, ldk, ldv, ldo;
    long long sq, sk, sv, so;
    float scale_log2e;
    int nqb;          // q-blocks (of 128 queries) per (batch, head)
};

constexpr int KV_SUB = 64;          // keys per softmax / MFMA pass (the S^T accumulators of one pass: two 32-key blocks)

// position of key j (0..15) inside its 16-key group in the V^T LDS row, such that lane-half h
// reads one contiguous 16-byte fragment holding exactly the keys its P registers cover
template <typename T> __device__ __forceinline__ int vt_pos(int j);
template <> __device__ __forceinline__ int vt_pos<bf16_t>(int j) { return (j < 4 || j >= 12) ? j : (j < 8 ? j + 4 : j - 4); }
template <> __device__ __forceinline__ int vt_pos<f16_t>(int j) { return vt_pos<bf16_t>(j); }
template <> __device__ __forceinline__ int vt_pos<float>(int j) { return j; }

// Smallest T-representable value >= x (the d = 40 kernel keeps its softmax reference point exactly representable in the operand type: -m rides
// in a Q element).  bf16: fp32's exponent range, the low 16 pattern bits dropped toward +inf.  fp16: 10 mantissa bits -> the low 13 pattern bits;
// below fp16's normal range (2^-14) the grid is coarser than that mask assumes -> 0 for the negatives, 2^-14 for the positives; clamped to +-65504
// (scores of that size in the exp2 domain do not occur: the logits would have to exceed 45 000).
template <typename T> __device__ __forceinline__ float ceil16(float x);
template <> __device__ __forceinline__ float ceil16<bf16_t>(float x) {
    const uint32_t tb = as_u32(x);
    return as_f32(x >= 0.f ? ((tb + 0xffffu) & 0xffff0000u) : (tb & 0xffff0000u));
}
template <> __device__ __forceinline__ float ceil16<f16_t>(float x) {
    const uint32_t tb = as_u32(x);
    float r = as_f32(x >= 0.f ? ((tb + 0x1fffu) & 0xffffe000u) : (tb & 0xffffe000u));
    if (fabsf(r) < 6.103515625e-05f) r = x > 0.f ? 6.103515625e-05f : 0.f;
    return fminf(fmaxf(r, -65504.f), 65504.f);
}
// the 16-bit pattern of a T-representable fp32 value
template <typename T> __device__ __forceinline__ uint32_t bits16(float x);
template <> __device__ __forceinline__ uint32_t bits16<bf16_t>(float x) { return as_u32(x) >> 16; }
template <> __device__ __forceinline__ uint32_t bits16<f16_t>(float x) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)x); }

// D = head dim (multiple of 8).  STEPS = 16-byte k-steps over D per lane-half pair.  QB = 32-query blocks per wave
// (QB = 2: each K / V^T fragment read from LDS feeds two MFMAs and the staging / barrier cost per query halves).
// KV_TILE = keys staged per barrier: 64, or 128 (two passes per stage: half the barriers and staging rounds; the loop is latency-bound --
// ~6k cycles per wave and 64-key pass against ~1.5k of issued work -- so what is synchronised less often is won).
// NW = waves per block (4 or 8): with 8, the two waves of every SIMD share ONE staged K / V tile -- the staging work per wave halves.
// Two query blocks per wave at 4 waves: without the second launch-bounds argument the compiler takes 241 + 128 registers (S and O in
// AGPRs, ~750 v_accvgpr moves per 128 keys, ONE wave per SIMD); capped at 256 it needs 237 and two blocks share a CU.
template <typename T, int D, int QB, int KV_TILE = 64, int NW = 4>
__global__ __launch_bounds__(NW * 64, (QB == 2 && NW == 4) ? 2 : 1) void attention_kernel(const AttnParams p) {
    constexpr int NT = NW * 64;
    static_assert(KV_TILE % KV_SUB == 0, "stage = whole passes");
    constexpr int VEC = elem<T>::VEC;                 // elements per 16 B
    constexpr int KSTEP = 2 * VEC;                    // d consumed per fragment pair (two lane halves)
    constexpr int STEPS = (D + KSTEP - 1) / KSTEP;
    constexpr int DVB = (D + 31) / 32;                // 32-row blocks of O^T
    constexpr int KROW = STEPS * 32 + 16;             // K row bytes, (KROW/16) odd
    constexpr int VROW = KV_TILE * (int)sizeof(T) + 16;               // V^T row bytes, (VROW/16) odd
    constexpr int VPR = D / VEC;                      // 16-byte vectors per K/V row
    constexpr int QW = 32 * QB;                       // queries per wave
    constexpr int TILE_BYTES = KV_TILE * KROW + DVB * 32 * VROW;      // one stage: K [64][KROW] + V^T [DVB*32][VROW]
    constexpr int NS = (2 * TILE_BYTES <= 160 * 1024) ? 2 : 1;        // double-buffer when it fits (fp32 d=160 does not)
    // When D leaves a spare (zero-pad) row in the V^T tile, that row is set to all ones: the P V MFMA then also produces
    // the softmax denominator sum_k P[k, q] (row D of O^T) and the 32 per-tile VALU adds per lane disappear.
    constexpr bool ONES = (D % 32) != 0;
    constexpr int L_I = D / 32, L_R = ((D % 32) / 8) * 4;             // accumulator block / register that holds row D (lane half 0)

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    // XCD-aware block order: XCD x (= dispatch id mod 8) owns a contiguous run of (batch*head, q-block) pairs with the
    // q-blocks of one head adjacent, so the K/V of the heads in flight on an XCD stay resident in its 4 MiB L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bh = bid / p.nqb, qblk = bid - bh * p.nqb;
    const int b = bh / p.heads, h = bh % p.heads;
    const int q0 = qblk * (NW * QW) + wave * QW;
    const T* Q = (const T*)p.q + b * p.sq + h * D;
    const T* K = (const T*)p.k + b * p.sk + h * D;
    const T* V = (const T*)p.v + b * p.sv + h * D;
    T* O = (T*)p.out + b * p.so + h * D;

    // ---- Q fragments (B operand): lane (q, half) holds Q[q][s*KSTEP + half*VEC .. +VEC)
    u32x4_t qf[QB][STEPS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 32 + lq;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * KSTEP + lh * VEC;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (qi < p.Nq && c < D) v = *(const u32x4_t*)(Q + (long long)qi * p.ldq + c);
            qf[qb][s] = v;
        }
    }
    // zero the stages once: the pad columns of K / pad rows of V^T are never rewritten
    for (int i = tid; i < NS * TILE_BYTES / 16; i += NT) ((u32x4_t*)smem)[i] = u32x4_t{0u, 0u, 0u, 0u};
    if constexpr (ONES) {
        __syncthreads();
        if (tid < NS * KV_TILE) {
            const int stg = tid / KV_TILE, kcol = tid - stg * KV_TILE;
            char* const op = smem + stg * TILE_BYTES + KV_TILE * KROW + D * VROW + kcol * (int)sizeof(T);
            if constexpr (sizeof(T) == 2) *(uint16_t*)op = one16<T>(); else *(float*)op = 1.0f;
        }
    }

    f32x16_t o[QB][DVB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < DVB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    float m_run[QB], l_run[QB];          // running max of the RAW scores, running sum (this lane's keys)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) { m_run[qb] = -INFINITY; l_run[qb] = 0.f; }
    const float c2 = p.scale_log2e;

    // K / V tile staging through registers: the loads of tile t+1 are issued before the MFMAs of tile t and written to
    // the other LDS stage afterwards (one barrier per tile).
    //   K : work item = (row r, 16-byte chunk c), row-major image.
    //   V : work item = (key pair pr, chunk c): the two keys 2pr, 2pr+1 are adjacent in the V^T row, so each transposed
    //       element pair is ONE 32-bit LDS store; lanes of a half-wave hold distinct pairs of one chunk => conflict-free.
    constexpr int NKV = (KV_TILE * VPR + NT - 1) / NT;
    constexpr int NVP = ((KV_TILE / 2) * VPR + NT - 1) / NT;
    u32x4_t rk[NKV], rv0[NVP], rv1[NVP];
    auto load_kv = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NKV; ++u) {
            const int idx = tid + u * NT;
            const int r = idx / VPR, c = idx - r * VPR;
            u32x4_t kk = {0u, 0u, 0u, 0u};
            if (idx < KV_TILE * VPR && kv0 + r < p.Nk) kk = *(const u32x4_t*)(K + (long long)(kv0 + r) * p.ldk + c * VEC);
            rk[u] = kk;
        }
#pragma unroll
        for (int u = 0; u < NVP; ++u) {
            const int idx = tid + u * NT;
            const int c = idx / (KV_TILE / 2), pr = idx - c * (KV_TILE / 2);
            u32x4_t a = {0u, 0u, 0u, 0u}, bq = {0u, 0u, 0u, 0u};
            if (idx < (KV_TILE / 2) * VPR) {
                const int kv = kv0 + 2 * pr;
                if (kv < p.Nk) a = *(const u32x4_t*)(V + (long long)kv * p.ldv + c * VEC);
                if (kv + 1 < p.Nk) bq = *(const u32x4_t*)(V + (long long)(kv + 1) * p.ldv + c * VEC);
            }
            rv0[u] = a;
            rv1[u] = bq;
        }
    };
    auto store_kv = [&](int stage) {
        char* ldsK = smem + stage * TILE_BYTES;
        char* ldsV = ldsK + KV_TILE * KROW;
#pragma unroll
        for (int u = 0; u < NKV; ++u) {
            const int idx = tid + u * NT;
            if (idx < KV_TILE * VPR) {
                const int r = idx / VPR, c = idx - r * VPR;
                *(u32x4_t*)(ldsK + r * KROW + c * 16) = rk[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NVP; ++u) {
            const int idx = tid + u * NT;
            if (idx < (KV_TILE / 2) * VPR) {
                const int c = idx / (KV_TILE / 2), pr = idx - c * (KV_TILE / 2);
                const int r = 2 * pr;
                const int pos = (r & ~15) + vt_pos<T>(r & 15);          // even; key r+1 sits at pos+1
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t x0 = rv0[u][e], x1 = rv1[u][e];
                        *(uint32_t*)(ldsV + (c * 8 + 2 * e) * VROW + pos * 2) = (x0 & 0xffffu) | (x1 << 16);
                        *(uint32_t*)(ldsV + (c * 8 + 2 * e + 1) * VROW + pos * 2) = (x0 >> 16) | (x1 & 0xffff0000u);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        u32x2_t w; w[0] = rv0[u][e]; w[1] = rv1[u][e];
                        *(u32x2_t*)(ldsV + (c * 4 + e) * VROW + pos * 4) = w;
                    }
                }
            }
        }
    };

    const int ntiles = (p.Nk + KV_TILE - 1) / KV_TILE;
    load_kv(0);
    __syncthreads();            // zero fill done
    store_kv(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * KV_TILE;
        const char* ldsK = smem + (NS == 2 ? (t & 1) : 0) * TILE_BYTES;
        const char* ldsV = ldsK + KV_TILE * KROW;
        if (t + 1 < ntiles) load_kv(kv0 + KV_TILE);

#pragma unroll
        for (int sub = 0; sub < KV_TILE / KV_SUB; ++sub) {
        const int kvs = kv0 + sub * KV_SUB;               // first key of this pass
        if (sub > 0 && kvs >= p.Nk) break;
        // ---- S^T = K Q^T for the two 32-key blocks of this pass (raw, unscaled scores)
        f32x16_t s[QB][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const u32x4_t kf = *(const u32x4_t*)(ldsK + (sub * KV_SUB + kb * 32 + lq) * KROW + st * 32 + lh * 16);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (st == 0) s[qb][kb] = f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // C = inline 0
                    AttnMma<T>::mma(s[qb][kb], kf, qf[qb][st]);
                }
            }
        // ---- online softmax on raw scores (scale > 0): p = exp2(c2 * s - c2 * m); keys of this lane: kb*32 + 8*(r>>2) + 4*lh + (r&3)
        if (kvs + KV_SUB > p.Nk) {        // tail pass only: mask keys beyond Nk
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kvs + kb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3) >= p.Nk) s[qb][kb][r] = -INFINITY;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = s[qb][0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qb], mx);
            const float mc = m_new * c2;
            if (__any(m_new != m_run[qb])) {       // rescale only when some query's running max moved (wave-uniform branch)
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] * c2 - mc);      // m_run = -inf on the first tile -> 0
                l_run[qb] *= alpha;
#pragma unroll
                for (int i = 0; i < DVB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
                m_run[qb] = m_new;
            }
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // two scores per v_pk_fma_f32 (the softmax is VALU-bound at d = 40: every instruction counts)
                    const f32x2_t a = __builtin_elementwise_fma(f32x2_t{s[qb][kb][r], s[qb][kb][r + 1]}, f32x2_t{c2, c2}, f32x2_t{-mc, -mc});
                    const float e0 = __builtin_amdgcn_exp2f(a[0]), e1 = __builtin_amdgcn_exp2f(a[1]);
                    s[qb][kb][r] = e0;
                    s[qb][kb][r + 1] = e1;
                    if constexpr (!ONES) psum += e0 + e1;
                }
            if constexpr (!ONES) l_run[qb] += psum;
        }

        // ---- O^T += V^T P^T   (each V^T fragment feeds the QB query blocks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {          // 16 keys per MFMA
                    u32x4_t pf[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pf[qb][e] = pack2<T>(s[qb][kb][8 * g + 2 * e], s[qb][kb][8 * g + 2 * e + 1]);
#pragma unroll
                    for (int i = 0; i < DVB; ++i) {
                        const u32x4_t vf = *(const u32x4_t*)(ldsV + (i * 32 + lq) * VROW + (sub * KV_SUB + kb * 32 + g * 16) * 2 + lh * 16);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) AttnMma<T>::mma(o[qb][i], vf, pf[qb]);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {          // 8 keys per 4-MFMA group
                    u32x4_t pf[QB];
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) pf[qb][e] = as_u32(s[qb][kb][4 * g + e]);
#pragma unroll
                    for (int i = 0; i < DVB; ++i) {
                        const u32x4_t vf = *(const u32x4_t*)(ldsV + (i * 32 + lq) * VROW + (sub * KV_SUB + kb * 32 + g * 8) * 4 + lh * 16);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) AttnMma<T>::mma(o[qb][i], vf, pf[qb]);
                    }
                }
            }
        }
        }      // passes of this stage
        if (t + 1 < ntiles) {
            if (NS == 1) __syncthreads();                  // single stage: every wave must be done reading tile t
            store_kv(NS == 2 ? ((t + 1) & 1) : 0);          // NS == 2: that stage was last read in iteration t-1
        }
        __syncthreads();
    }
    // ---- normalise and store: lane holds O[q][dv = i*32 + 8*(r>>2) + 4*lh + (r&3)]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l_tot;
        if constexpr (ONES) l_tot = __shfl(o[qb][L_I][L_R], lq, 64);        // row D of O^T lives in lane half 0
        else l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + qb * 32 + lq;
        if (qi < p.Nq) {
#pragma unroll
            for (int i = 0; i < DVB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = i * 32 + 8 * g + 4 * lh;
                    if (dv < D) {
                        T* dst = O + (long long)qi * p.ldo + dv;
                        if constexpr (sizeof(T) == 2) {
                            u32x2_t w;
                            w[0] = pack2<T>(o[qb][i][4 * g] * inv, o[qb][i][4 * g + 1] * inv);
                            w[1] = pack2<T>(o[qb][i][4 * g + 2] * inv, o[qb][i][4 * g + 3] * inv);
                            *(u32x2_t*)dst = w;
                        } else {
                            f32x4_t w = {o[qb][i][4 * g] * inv, o[qb][i][4 * g + 1] * inv, o[qb][i][4 * g + 2] * inv, o[qb][i][4 * g + 3] * inv};
                            *(f32x4_t*)dst = w;
                        }
                    }
                }
        }
    }
}


// ---- fp32 attention on split-bf16 operand pairs (dtype RF_BF16X3: the "f32x3" parity mode of the UNet) ----
// q / k / v / out are fp32 in memory.  Every operand value x enters the matrix pipe as hi = bf16(x), lo = bf16(x - hi) (16 significant
// bits) and a product is accumulated in fp32 as hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16 -- the operand form of rf_conv_gemm's
// RF_BF16X3 mode -- for BOTH contractions:  S^T = Khi Qhi + Khi Qlo + Klo Qhi  and  O^T += Vhi Phi + Vhi Plo + Vlo Phi  with the softmax
// probabilities split in registers (P = exp2(..) in fp32, Phi = bf16(P), Plo = bf16(P - Phi)).  Softmax state, rescaling and the output
// stay fp32.  42 bf16 MFMAs (1344 matrix-pipe cycles) per 64 keys x 32 queries at d = 40 against 104 exact-fp32 MFMAs (6656 cycles) of
// attention_kernel<float>; relative error ~2^-16 per product instead of 2^-24.  Same transposed formulation, staging and LDS geometry as
// attention_kernel<bf16_t> (two images per operand: hi, lo); the spare V^T row of ones (softmax denominator) lives in the hi image only.
template <int D, int KV_TILE = 64>
__global__ __launch_bounds__(256, 1) void attention_x3_kernel(const AttnParams p) {
    constexpr int NT = 256;
    constexpr int KSTEP = 16, STEPS = (D + KSTEP - 1) / KSTEP, DVB = (D + 31) / 32;
    constexpr int KROW = STEPS * 32 + 16, VROW = KV_TILE * 2 + 16, VPR = D / 8;
    constexpr int K_IMG = KV_TILE * KROW, V_IMG = DVB * 32 * VROW;
    constexpr int TILE_BYTES = 2 * (K_IMG + V_IMG);          // one stage: K hi, K lo, V^T hi, V^T lo
    constexpr int NS = (2 * TILE_BYTES <= 160 * 1024) ? 2 : 1;
    constexpr bool ONES = (D % 32) != 0;
    constexpr int L_I = D / 32, L_R = ((D % 32) / 8) * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bh = bid / p.nqb, qblk = bid - bh * p.nqb;
    const int b = bh / p.heads, h = bh % p.heads;
    const int q0 = qblk * 128 + wave * 32;
    const float* Q = (const float*)p.q + b * p.sq + h * D;
    const float* K = (const float*)p.k + b * p.sk + h * D;
    const float* V = (const float*)p.v + b * p.sv + h * D;
    float* O = (float*)p.out + b * p.so + h * D;

    // 8 fp32 -> (hi, lo) fragments of 8 bf16
    auto split8 = [](const f32x4_t& a, const f32x4_t& c, u32x4_t& hi, u32x4_t& lo) {
        const float f0[4] = {a[0], a[1], a[2], a[3]}, f1[4] = {c[0], c[1], c[2], c[3]};
        u32x2_t h0, l0, h1, l1;
        split4_bf16(f0, h0, l0);
        split4_bf16(f1, h1, l1);
        hi = u32x4_t{h0[0], h0[1], h1[0], h1[1]};
        lo = u32x4_t{l0[0], l0[1], l1[0], l1[1]};
    };
    const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
    u32x4_t qh[STEPS], ql[STEPS];
    {
        const int qi = q0 + lq;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * KSTEP + lh * 8;
            f32x4_t a = z4, c4 = z4;
            if (qi < p.Nq && c < D) {
                const float* src = Q + (long long)qi * p.ldq + c;
                a = *(const f32x4_t*)src;
                c4 = *(const f32x4_t*)(src + 4);
                if (p.scale_log2e != 1.0f) { a *= p.scale_log2e; c4 *= p.scale_log2e; }          // scores in the exp2 domain (fp32: no extra rounding issue)
            }
            split8(a, c4, qh[s], ql[s]);
        }
    }
    for (int i = tid; i < NS * TILE_BYTES / 16; i += NT) ((u32x4_t*)smem)[i] = u32x4_t{0u, 0u, 0u, 0u};
    if constexpr (ONES) {
        __syncthreads();
        if (tid < NS * KV_TILE) {
            const int stg = tid / KV_TILE, kcol = tid - stg * KV_TILE;
            *(bf16_t*)(smem + stg * TILE_BYTES + 2 * K_IMG + D * VROW + kcol * 2) = (bf16_t)0x3f80;          // hi image only
        }
    }
    f32x16_t o[DVB];
#pragma unroll
    for (int i = 0; i < DVB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    constexpr int NKV = (KV_TILE * VPR + NT - 1) / NT, NVP = ((KV_TILE / 2) * VPR + NT - 1) / NT;
    f32x4_t rk[NKV][2], rv0[NVP][2], rv1[NVP][2];
    auto load_kv = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NKV; ++u) {
            const int idx = tid + u * NT;
            const int r = idx / VPR, c = idx - r * VPR;
            rk[u][0] = rk[u][1] = z4;
            if (idx < KV_TILE * VPR && kv0 + r < p.Nk) {
                const float* src = K + (long long)(kv0 + r) * p.ldk + c * 8;
                rk[u][0] = *(const f32x4_t*)src;
                rk[u][1] = *(const f32x4_t*)(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < NVP; ++u) {
            const int idx = tid + u * NT;
            const int c = idx / (KV_TILE / 2), pr = idx - c * (KV_TILE / 2);
            rv0[u][0] = rv0[u][1] = rv1[u][0] = rv1[u][1] = z4;
            if (idx < (KV_TILE / 2) * VPR) {
                const int kv = kv0 + 2 * pr;
                if (kv < p.Nk) {
                    const float* src = V + (long long)kv * p.ldv + c * 8;
                    rv0[u][0] = *(const f32x4_t*)src;
                    rv0[u][1] = *(const f32x4_t*)(src + 4);
                }
                if (kv + 1 < p.Nk) {
                    const float* src = V + (long long)(kv + 1) * p.ldv + c * 8;
                    rv1[u][0] = *(const f32x4_t*)src;
                    rv1[u][1] = *(const f32x4_t*)(src + 4);
                }
            }
        }
    };
    auto store_kv = [&](int stage) {
        char* const kH = smem + stage * TILE_BYTES;
        char* const kL = kH + K_IMG;
        char* const vH = kH + 2 * K_IMG;
        char* const vL = vH + V_IMG;
#pragma unroll
        for (int u = 0; u < NKV; ++u) {
            const int idx = tid + u * NT;
            if (idx < KV_TILE * VPR) {
                const int r = idx / VPR, c = idx - r * VPR;
                u32x4_t hi, lo;
                split8(rk[u][0], rk[u][1], hi, lo);
                *(u32x4_t*)(kH + r * KROW + c * 16) = hi;
                *(u32x4_t*)(kL + r * KROW + c * 16) = lo;
            }
        }
#pragma unroll
        for (int u = 0; u < NVP; ++u) {
            const int idx = tid + u * NT;
            if (idx < (KV_TILE / 2) * VPR) {
                const int c = idx / (KV_TILE / 2), pr = idx - c * (KV_TILE / 2);
                const int r = 2 * pr;
                const int pos = (r & ~15) + vt_pos<bf16_t>(r & 15);          // even; key r+1 sits at pos+1
                u32x4_t h0, l0, h1, l1;
                split8(rv0[u][0], rv0[u][1], h0, l0);
                split8(rv1[u][0], rv1[u][1], h1, l1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    *(uint32_t*)(vH + (c * 8 + 2 * e) * VROW + pos * 2) = (h0[e] & 0xffffu) | (h1[e] << 16);
                    *(uint32_t*)(vH + (c * 8 + 2 * e + 1) * VROW + pos * 2) = (h0[e] >> 16) | (h1[e] & 0xffff0000u);
                    *(uint32_t*)(vL + (c * 8 + 2 * e) * VROW + pos * 2) = (l0[e] & 0xffffu) | (l1[e] << 16);
                    *(uint32_t*)(vL + (c * 8 + 2 * e + 1) * VROW + pos * 2) = (l0[e] >> 16) | (l1[e] & 0xffff0000u);
                }
            }
        }
    };
    auto mma = [](f32x16_t& acc, const u32x4_t& a, const u32x4_t& bq) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bq), acc, 0, 0, 0);
    };

    const int ntiles = (p.Nk + KV_TILE - 1) / KV_TILE;
    load_kv(0);
    __syncthreads();
    store_kv(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * KV_TILE;
        const char* const kH = smem + (NS == 2 ? (t & 1) : 0) * TILE_BYTES;
        const char* const kL = kH + K_IMG;
        const char* const vH = kH + 2 * K_IMG;
        const char* const vL = vH + V_IMG;
        if (t + 1 < ntiles) load_kv(kv0 + KV_TILE);
#pragma unroll
        for (int sub = 0; sub < KV_TILE / KV_SUB; ++sub) {
            const int kvs = kv0 + sub * KV_SUB;
            if (sub > 0 && kvs >= p.Nk) break;
            f32x16_t s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                s[kb] = f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < STEPS; ++st) {
                    const int off = (sub * KV_SUB + kb * 32 + lq) * KROW + st * 32 + lh * 16;
                    const u32x4_t kfh = *(const u32x4_t*)(kH + off), kfl = *(const u32x4_t*)(kL + off);
                    mma(s[kb], kfl, qh[st]);          // small terms first
                    mma(s[kb], kfh, ql[st]);
                    mma(s[kb], kfh, qh[st]);
                }
            }
            if (kvs + KV_SUB > p.Nk) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kvs + kb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3) >= p.Nk) s[kb][r] = -INFINITY;
            }
            float mx = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            if (__any(m_new != m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DVB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m_run = m_new;
            }
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[kb][r] - m_new);
                    s[kb][r] = e;
                    if constexpr (!ONES) psum += e;
                }
            if constexpr (!ONES) l_run += psum;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int g = 0; g < 2; ++g) {          // 16 keys per MFMA triple
                    u32x4_t ph, pl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = s[kb][8 * g + 2 * e], c = s[kb][8 * g + 2 * e + 1];
                        ph[e] = pack_bf2(a, c);
                        pl[e] = pack_bf2(a - as_f32(ph[e] << 16), c - as_f32(ph[e] & 0xffff0000u));
                    }
#pragma unroll
                    for (int i = 0; i < DVB; ++i) {
                        const int off = (i * 32 + lq) * VROW + (sub * KV_SUB + kb * 32 + g * 16) * 2 + lh * 16;
                        const u32x4_t vfh = *(const u32x4_t*)(vH + off), vfl = *(const u32x4_t*)(vL + off);
                        mma(o[i], vfl, ph);
                        mma(o[i], vfh, pl);
                        mma(o[i], vfh, ph);
                    }
                }
        }
        if (t + 1 < ntiles) {
            if (NS == 1) __syncthreads();
            store_kv(NS == 2 ? ((t + 1) & 1) : 0);
        }
        __syncthreads();
    }
    {
        float l_tot;
        if constexpr (ONES) l_tot = __shfl(o[L_I][L_R], lq, 64);
        else l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        const int qi = q0 + lq;
        if (qi < p.Nq) {
#pragma unroll
            for (int i = 0; i < DVB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = i * 32 + 8 * g + 4 * lh;
                    if (dv < D) *(f32x4_t*)(O + (long long)qi * p.ldo + dv) = f32x4_t{o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
                }
        }
    }
}

template <int D>
static int launch_attn_x3(const AttnParams& p, int B, hipStream_t st) {
    constexpr int KV_TILE = 64, STEPS = (D + 15) / 16, DVB = (D + 31) / 32;
    constexpr int tile_bytes = 2 * (KV_TILE * (STEPS * 32 + 16) + DVB * 32 * (KV_TILE * 2 + 16));
    constexpr int smem = (2 * tile_bytes <= 160 * 1024 ? 2 : 1) * tile_bytes;
    auto k = attention_x3_kernel<D, KV_TILE>;
    RF_RAISE_LDS(k, smem, "rf_attention");
    AttnParams pp = p;
    pp.nqb = (p.Nq + 127) / 128;
    hipLaunchKernelGGL(k, dim3(pp.nqb * B * p.heads), dim3(256), smem, st, pp);
    RF_LAUNCH_CHECK("rf_attention");
    return 0;
}

static int dispatch_attn_x3(const AttnParams& p, int B, hipStream_t st) {
    switch (p.d) {
        case 40: return launch_attn_x3<40>(p, B, st);
        case 64: return launch_attn_x3<64>(p, B, st);
        case 80: return launch_attn_x3<80>(p, B, st);
        case 160: return launch_attn_x3<160>(p, B, st);
        case 8: return launch_attn_x3<8>(p, B, st);
        case 16: return launch_attn_x3<16>(p, B, st);
        case 32: return launch_attn_x3<32>(p, B, st);
        default: break;
    }
    set_error("rf_attention: head dim %d not instantiated (have 8,16,32,40,64,80,160)", p.d);
    return 1;
}

// max of a value with its partner lane in the other half of the wave (lane ^ 32): v_permlane32_swap, a VALU operation -- the
// ds_bpermute of __shfl_xor waits for every LDS read in flight (the prefetched fragments) on the softmax's critical path.
__device__ __forceinline__ float max_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(as_u32(v), as_u32(v), false, false);
    return fmaxf(as_f32(r[0]), as_f32(r[1]));
}

// ---- d = 40 self-attention over long sequences (bf16): the same transposed formulation, software-pipelined INSIDE every wave ----
// The generic kernel runs {12 QK^T MFMAs} -> {~150 softmax VALU} -> {16 PV MFMAs} one after the other: matrix pipe and VALU never work
// for the same wave at the same time (PMC: 31 % matrix busy, 50 % VALU busy, nothing overlapped).  Here the unit of work is 32 keys x
// 64 queries and every step issues, in ONE basic block each,
//     region 1:  O^T += V^T P^T of unit u-1   (8 MFMAs)   beside   the running max of unit u          (VALU)
//     region 2:  S^T  = K Q^T   of unit u+1   (6 MFMAs)   beside   exp2 / pack of unit u -> P^T       (VALU)
// with two S accumulator sets alternating (tools/mfma_valu_probe.py: up to 5 VALU operations between two MFMAs of a wave are free,
// beyond that the group costs its VALU time + ~21 cycles per MFMA).  A first version staged V through registers into a transposed
// image (3 stages of 128 keys): 543 us at N = 4096 against 607 us for the generic kernel and 533 us for the kernel below.

// ds_read_b64_tr_b16 as inline asm: through the builtin the compiler cannot tell the read from the LDS-DMA writes in flight and puts
// an s_waitcnt vmcnt(0) in front of it (every tile then waits for the DMA issued a moment earlier).  The asm is invisible to the
// waitcnt pass, so whoever uses the result waits with tr_wait() first (LDS operations complete in order: the compiler's own counted
// lgkmcnt waits only become stricter).
template <int OFF> __device__ __forceinline__ u32x2_t ds_read_tr16(const char* p) {
    u32x2_t r;
    const uint32_t a = (uint32_t)(uintptr_t)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
    return r;
}

// ---- the same in-wave pipeline with BOTH operands staged by LDS-DMA ----
// K and V tiles (64 keys) arrive as unpadded row-major images (80-byte rows) through `buffer_load ... lds`: no staging registers, no
// VALU, no LDS stores, and a ring of 7 stages (70 KB, two blocks per CU) puts every tile 6 iterations (~2.5 us) ahead of its first use.
// V^T fragments come out of the row-major image with ds_read_b64_tr_b16 (lane i of a 16-lane group
// receives column i of the [4 keys][16 columns] block the group addresses; tools/tr_probe.py), two reads per MFMA operand; the 24
// rows of O^T beyond the head dim multiply whatever follows the row and are never stored -- except rows 48..63, whose lanes read a
// run of ones, so the MFMA also delivers the softmax denominator.  Every wave issues two or three of a tile's ten one-KiB pieces.
// The running max moves only when a score exceeds it by 2^8 in the exp2 domain (P <= 256; the O rescale becomes rare).
// d = 80 (round 6): the same stream with five whole k-steps and three O^T row blocks.  Nothing is padded in the head dim, so the softmax
// reference point enters as the C operand of the first QK^T MFMA of a unit (a 16-register splat of -m_run per query block, rewritten only when
// the reference moves) instead of riding in a padded k-slot; rows 80..95 of O^T read the run of ones (the denominator).  S twice + O + Q + the
// splats are ~300 registers: ONE wave per SIMD (a 142 KB ring, one block per CU) -- the in-wave pipeline is what overlaps the two pipes here.
template <typename T, int D, int KT>
__global__ __launch_bounds__(256, (D < 48 ? 2 : 1)) void attention_dma_kernel(const AttnParams p) {
    static_assert(sizeof(T) == 2, "16-bit operands (bf16 / fp16)");
    static_assert((D > 32 && D < 48 && D % 8 == 0) || D == 80, "three 16-wide k-steps with a padded slot, or five whole ones");
    static_assert(D % 32 == 8 || D % 32 == 16, "the upper 16 rows of the last O^T row block are free for the denominator");
    constexpr int QSTEPS = (D + 15) / 16;               // 16-wide k-steps of QK^T
    constexpr bool PADK = D % 16 != 0;                  // the last k-step has a zero-padded lane half: -m_run rides there
    constexpr int DVB = (D + 31) / 32;                  // 32-row blocks of O^T
    // KT keys per stage and barrier: 64 (ring of 7) or 128 (ring of 3; half the barriers) -- both ~70 KB at d = 40, two blocks per CU
    constexpr int QB = 2, NU = KT / 32, NSTG = KT == 64 ? 7 : 3;
    constexpr int VPR = D / 8, KROW = D * 2;
    constexpr int K_BYTES = KT * KROW, STAGE = 2 * K_BYTES;
    constexpr int ONES_OFF = NSTG * STAGE, ONES_BYTES = (KT - 8) * KROW + 128;     // a run of bf16 ones (see the V^T fragments below)
    constexpr int PIECES = K_BYTES / 1024;              // one-KiB DMA pieces per operand tile
    static_assert(K_BYTES % 1024 == 0 && (NU == 2 || NU == 4), "whole DMA pieces; two or four units per tile");
    constexpr int NP = 2 * PIECES, SLOTS = (NP + 3) / 4;          // pieces of a tile (K then V, contiguous in the stage), per wave

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bh = bid / p.nqb, qblk = bid - bh * p.nqb;
    const int b = bh / p.heads, h = bh % p.heads;
    const int q0 = qblk * (4 * 32 * QB) + wave * (32 * QB);
#ifdef RF_ATTN_STAMP          // (experiment builds: cycle stamps of the kernel's phases, written over the first output row of every wave -- tools/archive/attn_stamp.py)
    long long stamp_[5];
    stamp_[0] = __builtin_readcyclecounter();
#endif
    const T* Q = (const T*)p.q + b * p.sq + h * D;
    const T* K = (const T*)p.k + b * p.sk + h * D;
    const T* V = (const T*)p.v + b * p.sv + h * D;
    T* O = (T*)p.out + b * p.so + h * D;

    u32x4_t qf[QB][QSTEPS];
    auto load_q = [&]() {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qi = q0 + qb * 32 + lq;
#pragma unroll
        for (int st = 0; st < QSTEPS; ++st) {
            const int c = st * 16 + lh * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (qi < p.Nq && c < D) v = *(const u32x4_t*)(Q + (long long)qi * p.ldq + c);
            // Q carries scale * log2(e): the scores come out of the MFMA in the exp2 domain.  That is one more bf16 rounding of
            // q * c (a 2^-9-relative perturbation of the logits) unless the caller already folded the factor into q (scale = ln 2,
            // what the UNet does through the to_q weights): then this is a multiplication by exactly 1 and is skipped.
            if (p.scale_log2e != 1.0f) {
                float f[8];
                unpack16<T>(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= p.scale_log2e;
                v = pack16<T>(f);
            }
            qf[qb][st] = v;
        }
    }
    };
    if constexpr (PADK) load_q();

    if constexpr (PADK) {
        for (int i = tid; i < ONES_BYTES / 4; i += 256) ((uint32_t*)(smem + ONES_OFF))[i] = (uint32_t)one16<T>() * 0x10001u;
    } else {
        static_assert(ONES_BYTES % 16 == 0, "whole 16-byte writes");
        const uint32_t o2 = (uint32_t)one16<T>() * 0x10001u;
        for (int i = tid; i < ONES_BYTES / 16; i += 256) ((u32x4_t*)(smem + ONES_OFF))[i] = u32x4_t{o2, o2, o2, o2};
    }
    // ---- staging: tile X is issued by wave X % 4
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (unsigned)((((long long)p.Nk - 1) * p.ldk + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (unsigned)((((long long)p.Nk - 1) * p.ldv + D) * 2), 0x00020000);
    // Every wave issues its share of every tile: pieces wave, wave + 4, ... of the NP one-KiB pieces (K pieces first, V pieces
    // behind them, in stage order).  The per-lane source offsets do not depend on the tile.
    int off[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int pc = wave + 4 * i, g = (pc % PIECES) * 64 + lane, row = g / VPR, ch = g - row * VPR;
        off[i] = row * (pc < PIECES ? p.ldk : p.ldv) * 2 + ch * 16;
    }
    auto dma_tile = [&](int tile, int stage) {
        const int sk = tile * (KT * p.ldk * 2), sv = tile * (KT * p.ldv * 2);
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int pc = wave + 4 * i;                  // wave-uniform
            if (pc < NP) {
                char* const dst = smem + stage * STAGE + pc * 1024;
                const int vo = off[i];
                if (pc < PIECES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (__attribute__((address_space(3))) void*)dst, 16, vo, sk, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (__attribute__((address_space(3))) void*)dst, 16, vo, sv, 0, 0);
            }
        }
    };

    // ---- per-lane LDS offsets of the fragments (unit 0 of a stage)
    int kfo[QSTEPS];
#pragma unroll
    for (int st = 0; st < QSTEPS; ++st) {
        const int slot = st * 2 + lh;
        kfo[st] = lq * KROW + (slot < VPR ? slot : VPR - 1) * 16;
    }
    // V^T fragment = two transposing reads of [4 keys][16 columns]: keys 4 lh + (l16 >> 2) (+ 8), columns 16 * ((lane >> 4) & 1) + 4 (l16 & 3)
    const int l16 = lane & 15;
    const int vfo = K_BYTES + (4 * lh + (l16 >> 2)) * KROW + (((lane >> 4) & 1) * 16 + (l16 & 3) * 4) * 2;
    // Rows 48..63 of O^T (lanes 16..31 of the second row block) have no V column: their reads go to the run of ones instead, so that
    // those rows of the MFMA accumulate sum_k P[k, q] -- the softmax denominator, at no VALU cost.
    const bool ones_lane = ((lane >> 4) & 1) != 0;
    const char* const ones_ptr = smem + ONES_OFF - (DVB - 1) * 64;
    const char* const ones_k = smem + ONES_OFF;          // 16 bytes of ones: the K fragment of the padded k-slot (see m_run)

    f32x16_t o[QB][DVB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int i = 0; i < DVB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    // Softmax reference point m_run (exp2 domain; set from the first unit, afterwards only raised, and only when a score exceeds it
    // by `thr`: P = 2^(s - m_run) <= 2^8).  The subtraction rides in the MFMA: the zero-padded k-slot of the head dim (d = 40..47, lane
    // half 1 of the third k-step) multiplies a K fragment of ONES with a Q fragment whose first element is -m_run, so the accumulators
    // hold s - m_run and the exponentials need neither a subtraction nor a scale.  m_run is kept bf16-representable for that.
    float m_run[QB] = {0.f, 0.f};
    constexpr float thr = 8.0f;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16_t negm[PADK ? 1 : QB];                       // !PADK: -m_run of every query block as the C operand of a unit's first QK^T MFMA
    negm[0] = zero16;
    if constexpr (!PADK) negm[QB - 1] = zero16;
    const int ntiles = p.Nk / KT;                       // >= NSTG (dispatch)
    if constexpr (PADK) {
#pragma unroll
        for (int x = 0; x < NSTG - 1; ++x) dma_tile(x, x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // d = 80: one block per CU, nobody else covers the fill.  Tile 0 goes out first, the Q rows behind it, tile 1 last: "at most tile 1's pieces
        // outstanding" then says tile 0 AND Q have arrived (whether or not this wave had Q rows to load); the loop's own wait covers tile 1.
        static_assert(NSTG == 3 && NP % 4 == 0, "two tiles in the prologue, whole pieces per wave");
        dma_tile(0, 0);
        load_q();
        dma_tile(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP / 4) : "memory");
    }
    __syncthreads();

    f32x16_t S[2][QB];
    u32x4_t pf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) pf[qb][0] = pf[qb][1] = u32x4_t{0u, 0u, 0u, 0u};
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    // Fragments are read one region ahead of the MFMAs that use them (the LDS latency would otherwise sit in front of every group).
    u32x4_t kf[QSTEPS], vf[2][DVB];
#define RF_LOAD_KF(sb, un)                                                                                      \
    {                                                                                                           \
        _Pragma("unroll") for (int st = 0; st < QSTEPS - (PADK ? 1 : 0); ++st) kf[st] = *(const u32x4_t*)((sb) + (un) * (32 * KROW) + kfo[st]);             \
        if constexpr (PADK) kf[QSTEPS - 1] = *(const u32x4_t*)(lh ? ones_k : (sb) + (un) * (32 * KROW) + kfo[QSTEPS - 1]);                            \
    }
#define RF_LOAD_VF1(sb, un, g, i)                                                                               \
    {                                                                                                           \
        const char* const vb = ((i) == DVB - 1 && ones_lane) ? ones_ptr : (sb) + vfo;                           \
        const u32x2_t l2 = ds_read_tr16<((un) * 32 + (g) * 16) * KROW + (i) * 64>(vb);                          \
        const u32x2_t h2 = ds_read_tr16<((un) * 32 + (g) * 16 + 8) * KROW + (i) * 64>(vb);                      \
        vf[g][i] = u32x4_t{l2[0], l2[1], h2[0], h2[1]};                                                         \
    }
#define RF_LOAD_VF(sb, un)                                                                                      \
    {                                                                                                           \
        RF_LOAD_VF1(sb, un, 0, 0) RF_LOAD_VF1(sb, un, 0, 1) RF_LOAD_VF1(sb, un, 1, 0) RF_LOAD_VF1(sb, un, 1, 1)   \
        if constexpr (DVB == 3) { RF_LOAD_VF1(sb, un, 0, DVB - 1) RF_LOAD_VF1(sb, un, 1, DVB - 1) }             \
    }
#define RF_WAIT_VF()                                                                                            \
    {                                                                                                           \
        if constexpr (DVB == 3)                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[0][DVB - 1]), "+v"(vf[1][DVB - 1]));   \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[1][0]), "+v"(vf[1][1]));                    \
    }
#define RF_WAIT_KF()                                                                                            \
    {                                                                                                           \
        if constexpr (QSTEPS == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[QSTEPS - 2]), "+v"(kf[QSTEPS - 1]));   \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]));                      \
    }
#define RF_QK(dst)                                                                                              \
    {                                                                                                           \
        _Pragma("unroll") for (int st = 0; st < QSTEPS; ++st)                                                   \
            _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) {                                                 \
                if (st == 0 && !PADK) dst[qb] = AttnMma<T>::mma_c(kf[st], qf[qb][st], negm[PADK ? 0 : qb]);      \
                else {                                                                                          \
                    if (st == 0) dst[qb] = zero16;                                                              \
                    AttnMma<T>::mma(dst[qb], kf[st], qf[qb][st]);                                               \
                }                                                                                               \
            }                                                                                                   \
    }
#define RF_PV()                                                                                                 \
    {                                                                                                           \
        _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                           \
            _Pragma("unroll") for (int i = 0; i < DVB; ++i)                                                     \
                _Pragma("unroll") for (int qb = 0; qb < QB; ++qb) AttnMma<T>::mma(o[qb][i], vf[g][i], pf[qb][g]);                    \
    }
#ifdef RF_ATTN_STAMP
    stamp_[1] = __builtin_readcyclecounter();
#endif
    RF_LOAD_KF(smem, 0)
    RF_WAIT_KF()
    RF_QK(S[0])
    RF_LOAD_VF(smem, 0)                                  // the first step multiplies P = 0 with tile 0's own V rows
    int sc = 0;                                          // stage of tile t
    for (int t = 0; t < ntiles; ++t) {
        const int sn = sc == NSTG - 1 ? 0 : sc + 1;
        const int sp = sc == 0 ? NSTG - 1 : sc - 1;      // stage of tile t-1: takes tile t + NSTG - 1
        const char* const sb = smem + sc * STAGE;
        const char* const sbn = smem + sn * STAGE;
        auto unit = [&](auto UU) {
            constexpr int uu = decltype(UU)::value;           // compile-time: it selects immediates of the inline-asm reads
            f32x16_t* const cur = S[uu & 1];
            f32x16_t* const nxt = S[(uu + 1) & 1];
            // ---- region 1: PV of the previous unit beside the max of this one
            RF_WAIT_VF()                                   // issued a region ago: no stall -- and nothing younger is in flight yet
            if (uu == NU - 1) RF_LOAD_KF(sbn, 0) else RF_LOAD_KF(sb, uu + 1)
            __builtin_amdgcn_sched_barrier(0);
            RF_PV()
            float mx[QB];                                   // max of the relative scores s - m_run of this unit
            bool moved = false;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float m = cur[qb][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, cur[qb][r]);
                mx[qb] = max_xor32(m);
                moved |= mx[qb] > thr;
            }
            moved |= (t == 0 && uu == 0);
            if (uu == 0) {
                // tile t+1 (read from this iteration's last unit on) must have landed: behind it this wave has tiles t+2 .. t+NSTG-2 in
                // flight, NP / 4 or NP / 4 + 1 pieces each -- "at most (NSTG - 3) * (NP / 4) outstanding" covers both
                if (t + NSTG - 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 3) * (NP / 4)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                         // every wave is past the last read of tile t-1: its stage takes tile t+NSTG-1
                if (t + NSTG - 1 < ntiles) dma_tile(t + NSTG - 1, sp);
            }
            if (__builtin_expect(__any(moved), 0)) {      // out of line: the common path falls through
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    // new reference: this unit's max (the first unit may also lower it), rounded UP to a value of the operand type
                    const float target = m_run[qb] + ((t == 0 && uu == 0) ? mx[qb] : fmaxf(mx[qb], 0.f));
                    const float m_new = PADK ? ceil16<T>(target) : target;
                    const float delta = m_new - m_run[qb];
                    // (delta < 0 only when the first unit lowers the reference: O is still zero there, and 2^-delta overflows for scores below -128)
                    const float alpha = __builtin_amdgcn_exp2f(fminf(-delta, 0.f));
#pragma unroll
                    for (int i = 0; i < DVB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
                    m_run[qb] = m_new;
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[qb][r] -= delta;          // this unit's scores were taken against the old reference
                    if constexpr (PADK) {
                        if (lh) qf[qb][QSTEPS - 1][0] = bits16<T>(-m_new);       // lane half 1 of the last k-step: [-m_run, 0, ...] (exact: a value of T)
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) negm[PADK ? 0 : qb][r] = -m_new;
                    }
                }
            }
            // ---- region 2: QK^T of the next unit beside exp2 / pack of this one
            RF_WAIT_KF()                                   // before the (asm) V reads go out: a counted wait would otherwise cover them too
            RF_LOAD_VF(sb, uu)
            __builtin_amdgcn_sched_barrier(0);
            RF_QK(nxt)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 8 * g + 2 * e;
                        const float e0 = __builtin_amdgcn_exp2f(cur[qb][r]), e1 = __builtin_amdgcn_exp2f(cur[qb][r + 1]);
                        pf[qb][g][e] = pack2<T>(e0, e1);
                    }
            }
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
        if constexpr (NU == 4) {
            unit(std::integral_constant<int, 2>{});
            unit(std::integral_constant<int, 3>{});
        }
        sc = sn;
#ifdef RF_ATTN_STAMP
        if (t == 0) stamp_[2] = __builtin_readcyclecounter();
#endif
    }
#ifdef RF_ATTN_STAMP
    stamp_[3] = __builtin_readcyclecounter();
#endif
    RF_WAIT_VF()
    RF_PV()
#undef RF_WAIT_VF
#undef RF_WAIT_KF
#undef RF_LOAD_KF
#undef RF_LOAD_VF
#undef RF_LOAD_VF1
#undef RF_QK
#undef RF_PV
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = __shfl(o[qb][DVB - 1][8], lq, 64);          // row 48 (d = 80: 80) of O^T: lane half 0, register 8 of the last row block
        const float inv = 1.0f / l_tot;
        const int qi = q0 + qb * 32 + lq;
        if constexpr (!PADK) {
            // 16-byte stores: the two lane halves of a query hold columns 8 g + {0..3} and 8 g + {4..7}; one v_permlane32_swap per dword hands lane half 0
            // the whole group g (even), lane half 1 the whole group g + 1 -- half the store instructions, 32 contiguous bytes per row and instruction
#pragma unroll
            for (int i = 0; i < DVB; ++i)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    uint32_t w[2][2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        w[k][0] = pack2<T>(o[qb][i][4 * (g + k)] * inv, o[qb][i][4 * (g + k) + 1] * inv);
                        w[k][1] = pack2<T>(o[qb][i][4 * (g + k) + 2] * inv, o[qb][i][4 * (g + k) + 3] * inv);
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);          // [0]: group g (lh 0) / g + 1's low half from below (lh 1)
                    const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                    const int dv = i * 32 + 8 * (g + lh);
                    if (qi < p.Nq && dv < D) *(u32x4_t*)(O + (long long)qi * p.ldo + dv) = u32x4_t{s0[0], s1[0], s0[1], s1[1]};
                }
        } else if (qi < p.Nq) {
#pragma unroll
            for (int i = 0; i < DVB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dv = i * 32 + 8 * g + 4 * lh;
                    if (dv < D) {
                        u32x2_t w;
                        w[0] = pack2<T>(o[qb][i][4 * g] * inv, o[qb][i][4 * g + 1] * inv);
                        w[1] = pack2<T>(o[qb][i][4 * g + 2] * inv, o[qb][i][4 * g + 3] * inv);
                        *(u32x2_t*)(O + (long long)qi * p.ldo + dv) = w;
                    }
                }
        }
    }
#ifdef RF_ATTN_STAMP
    stamp_[4] = __builtin_readcyclecounter();
    if (lane == 0 && q0 < p.Nq) {
        int* const dst = (int*)(O + (long long)q0 * p.ldo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = (int)(stamp_[i + 1] - stamp_[i]);
        dst[4] = (int)(stamp_[0] & 0x7fffffff);
    }
#endif
}

template <typename T, int D, int KT>
static int launch_attn_dma(const AttnParams& p, int B, hipStream_t st) {
    constexpr int smem = (KT == 64 ? 7 : 3) * (2 * KT * D * 2) + (KT - 8) * D * 2 + 128;      // stages + the run of ones
    auto k = attention_dma_kernel<T, D, KT>;
    RF_RAISE_LDS(k, smem, "rf_attention");
    AttnParams pp = p;
    pp.nqb = (p.Nq + 255) / 256;
    static const int pad = tune_env("RF_ATTN_SMEM_PAD", 0);      // experiment: one block per CU
    if (pad) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem + pad);
    hipLaunchKernelGGL(k, dim3(pp.nqb * B * p.heads), dim3(256), smem + pad, st, pp);
    RF_LAUNCH_CHECK("rf_attention");
    return 0;
}

template <typename T, int D, int QB, int KV_TILE = 64, int NW = 4>
static int launch_attn_qb(const AttnParams& p, int B, hipStream_t st) {
    constexpr int VEC = elem<T>::VEC, KSTEP = 2 * VEC, STEPS = (D + KSTEP - 1) / KSTEP, DVB = (D + 31) / 32;
    constexpr int KROW = STEPS * 32 + 16;
    constexpr int VROW = KV_TILE * (int)sizeof(T) + 16;
    constexpr int tile_bytes = KV_TILE * KROW + DVB * 32 * VROW;
    constexpr int smem = (2 * tile_bytes <= 160 * 1024 ? 2 : 1) * tile_bytes;
    auto k = attention_kernel<T, D, QB, KV_TILE, NW>;
    RF_RAISE_LDS(k, smem, "rf_attention");
    AttnParams pp = p;
    pp.nqb = (p.Nq + 32 * NW * QB - 1) / (32 * NW * QB);
    dim3 grid(pp.nqb * B * p.heads);
    hipLaunchKernelGGL(k, grid, dim3(NW * 64), smem, st, pp);
    RF_LAUNCH_CHECK("rf_attention");
    return 0;
}

template <typename T, int D>
static int launch_attn(const AttnParams& p, int B, hipStream_t st) {
    // QB = 2 (two query blocks per wave, every K / V^T fragment feeds two MFMAs) pays for the small head dim when the
    // grid still fills the chip: d=40, N=4096: 687 us vs 739 us; it loses at d=80 (N=1024: 99 us vs 85 us).
    if constexpr (sizeof(T) == 2 && D <= 40) {
        static const int kt = tune_env("RF_ATTN_KT", 128);
        if ((long long)((p.Nq + 255) / 256) * B * p.heads >= 512) {
            // long sequences of whole 64-key tiles: the in-wave software-pipelined kernel (RF_ATTN_PIPE=0: the generic one)
            static const int pipe = tune_env("RF_ATTN_PIPE", 1);
            if constexpr (D == 40) {
                static const int kt128 = tune_env("RF_ATTN_KT128", 1);
                if (pipe && kt128 && p.Nk % 128 == 0 && p.Nk >= 1024) return launch_attn_dma<T, D, 128>(p, B, st);
                if (pipe && p.Nk % 64 == 0 && p.Nk >= 1024) return launch_attn_dma<T, D, 64>(p, B, st);
            }
            // 8-wave blocks (both waves of a SIMD on one staged tile) measured 608 vs 586 us at N = 4096: opt-in only (RF_ATTN_NW=8)
            static const int nw = tune_env("RF_ATTN_NW", 4);
            if (kt == 128 && p.Nk >= 1024 && nw == 8 && (long long)((p.Nq + 511) / 512) * B * p.heads >= 512)
                return launch_attn_qb<T, D, 2, 128, 8>(p, B, st);
            if (kt == 128 && p.Nk >= 1024) return launch_attn_qb<T, D, 2, 128>(p, B, st);
            return launch_attn_qb<T, D, 2>(p, B, st);
        }
    }
    // d = 80 (the 32x32 / 48x48 levels): 128 keys per stage and 8 waves per block -- both waves of a SIMD share one staged K / V tile, half the barriers;
    // 73 us against 85 at N = 1024 (tools/archive/run_r04x.sh: 64-key stages with 8 waves 85, 128-key stages with 4 waves 105, two query blocks per wave 81-99).
    // d = 160 keeps the 4-wave / 64-key form (N = 256: four stages of 64 keys already cover the sequence).
    if constexpr (sizeof(T) == 2 && D == 80) {
        // whole 128-key tiles: the in-wave software-pipelined kernel, one wave per SIMD (RF_ATTN_PIPE80=0 in experiment builds: the generic one)
        static const int pipe80 = tune_env("RF_ATTN_PIPE80", 1);
        if (pipe80 && p.Nk % 128 == 0 && p.Nk >= 384) return launch_attn_dma<T, D, 128>(p, B, st);
        if (p.Nk >= 512 && (long long)((p.Nq + 255) / 256) * B * p.heads >= 256) return launch_attn_qb<T, D, 1, 128, 8>(p, B, st);
    }
    return launch_attn_qb<T, D, 1>(p, B, st);
}

template <typename T>
static int dispatch_attn(const AttnParams& p, int B, hipStream_t st) {
    switch (p.d) {
        case 40: return launch_attn<T, 40>(p, B, st);
        case 64: return launch_attn<T, 64>(p, B, st);
        case 80: return launch_attn<T, 80>(p, B, st);
        case 160: return launch_attn<T, 160>(p, B, st);
        case 8: return launch_attn<T, 8>(p, B, st);
        case 16: return launch_attn<T, 16>(p, B, st);
        case 32: return launch_attn<T, 32>(p, B, st);
        default: break;
    }
    set_error("rf_attention: head dim %d not instantiated (have 8,16,32,40,64,80,160)", p.d);
    return 1;
}

}  // namespace rf

extern "C" int rf_attention(int dtype, const void* q, const void* k, const void* v, void* out, int B, int heads, int d, int Nq, int Nk,
                            int ldq, int ldk, int ldv, int ldo, int64_t sq, int64_t sk, int64_t sv, int64_t so, float scale, void* stream) {
    using namespace rf;
    RF_CHECK(dtype == RF_F32 || dtype == RF_BF16 || dtype == RF_F16 || dtype == RF_BF16X3, "rf_attention: bad dtype %d", dtype);
    RF_CHECK(q && k && v && out && B > 0 && heads > 0 && Nq > 0 && Nk > 0, "rf_attention: bad arguments");
    const int vec = (dtype == RF_BF16 || dtype == RF_F16) ? 8 : 4;
    RF_CHECK(d % 8 == 0 && ldq % vec == 0 && ldk % vec == 0 && ldv % vec == 0 && ldo % 4 == 0, "rf_attention: d/ld alignment (d=%d)", d);
    RF_CHECK(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "rf_attention: operands must be 16-byte aligned");
    RF_CHECK((long long)B * heads * ((Nq + 127) / 128) < (1LL << 31), "rf_attention: grid too large");
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.heads = heads; p.d = d; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.sq = sq; p.sk = sk; p.sv = sv; p.so = so;
    p.scale_log2e = scale * 1.4426950408889634f;
    if (fabsf(p.scale_log2e - 1.0f) < 1e-6f) p.scale_log2e = 1.0f;       // scale = ln 2: the caller's scores are already in the exp2 domain
    if (dtype == RF_BF16X3) return dispatch_attn_x3(p, B, (hipStream_t)stream);          // fp32 in memory, split-bf16 operand pairs on the bf16 MFMA
    if (dtype == RF_F32) return dispatch_attn<float>(p, B, (hipStream_t)stream);
    if (dtype == RF_F16) return dispatch_attn<f16_t>(p, B, (hipStream_t)stream);
    return dispatch_attn<bf16_t>(p, B, (hipStream_t)stream);
}
