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
//   bf16: v_mfma_f32_32x32x16_bf16, fp32: 4 x v_mfma_f32_32x32x2_f32 (exact fp32).
// Block = 4 waves = 128 queries of one (batch, head); KV tile = 64 keys.
#include "common.h"

namespace rf {

template <typename T> struct AttnMma;
template <> struct AttnMma<bf16_t> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
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
};

constexpr int KV_TILE = 64;

// position of key j (0..15) inside its 16-key group in the V^T LDS row, such that lane-half h
// reads one contiguous 16-byte fragment holding exactly the keys its P registers cover
template <typename T> __device__ __forceinline__ int vt_pos(int j);
template <> __device__ __forceinline__ int vt_pos<bf16_t>(int j) { return (j < 4 || j >= 12) ? j : (j < 8 ? j + 4 : j - 4); }
template <> __device__ __forceinline__ int vt_pos<float>(int j) { return j; }

// D = head dim (multiple of 8).  STEPS = 16-byte k-steps over D per lane-half pair.
template <typename T, int D>
__global__ __launch_bounds__(256) void attention_kernel(const AttnParams p) {
    constexpr int VEC = elem<T>::VEC;                 // elements per 16 B
    constexpr int KSTEP = 2 * VEC;                    // d consumed per fragment pair (two lane halves)
    constexpr int STEPS = (D + KSTEP - 1) / KSTEP;
    constexpr int DVB = (D + 31) / 32;                // 32-row blocks of O^T
    constexpr int KROW = STEPS * 32 + (((STEPS * 2) & 1) ? 0 : 16);   // K row bytes, (KROW/16) odd
    constexpr int VROW = KV_TILE * (int)sizeof(T) + 16;               // V^T row bytes, (VROW/16) odd
    constexpr int VPR = D / VEC;                      // 16-byte vectors per K/V row

    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_BYTES = KV_TILE * KROW + DVB * 32 * VROW;      // one stage: K [64][KROW] + V^T [DVB*32][VROW]
    constexpr int NS = (2 * TILE_BYTES <= 160 * 1024) ? 2 : 1;        // double-buffer when it fits (fp32 d=160 does not)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh % p.heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const T* Q = (const T*)p.q + b * p.sq + h * D;
    const T* K = (const T*)p.k + b * p.sk + h * D;
    const T* V = (const T*)p.v + b * p.sv + h * D;
    T* O = (T*)p.out + b * p.so + h * D;

    // ---- Q fragments (B operand): lane (q, half) holds Q[q][s*KSTEP + half*VEC .. +VEC)
    u32x4_t qf[STEPS];
    {
        const int qi = q0 + lq;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * KSTEP + lh * VEC;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (qi < p.Nq && c < D) v = *(const u32x4_t*)(Q + (long long)qi * p.ldq + c);
            qf[s] = v;
        }
    }
    // zero both stages once: the pad columns of K / pad rows of V^T are never rewritten
    for (int i = tid; i < NS * TILE_BYTES / 16; i += 256) ((u32x4_t*)smem)[i] = u32x4_t{0u, 0u, 0u, 0u};

    f32x16_t o[DVB];
#pragma unroll
    for (int i = 0; i < DVB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;          // running max of the RAW scores, running sum (this lane's keys)
    const float c2 = p.scale_log2e;

    // K / V tile staging through registers: the loads of tile t+1 are issued before the MFMAs of tile t and written to
    // the other LDS stage afterwards (one barrier per tile)
    constexpr int NV = (KV_TILE * VPR + 255) / 256;               // 16-byte vectors per thread per operand
    u32x4_t rk[NV], rv[NV];
    auto load_kv = [&](int kv0) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = tid + u * 256;
            const int r = idx / VPR, c = idx - r * VPR;
            const int kv = kv0 + r;
            u32x4_t kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (idx < KV_TILE * VPR && kv < p.Nk) {
                kk = *(const u32x4_t*)(K + (long long)kv * p.ldk + c * VEC);
                vv = *(const u32x4_t*)(V + (long long)kv * p.ldv + c * VEC);
            }
            rk[u] = kk;
            rv[u] = vv;
        }
    };
    auto store_kv = [&](int stage) {
        char* ldsK = smem + stage * TILE_BYTES;
        char* ldsV = ldsK + KV_TILE * KROW;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = tid + u * 256;
            if (idx < KV_TILE * VPR) {
                const int r = idx / VPR, c = idx - r * VPR;
                *(u32x4_t*)(ldsK + r * KROW + c * 16) = rk[u];
                const int pos = (r & ~15) + vt_pos<T>(r & 15);
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        *(bf16_t*)(ldsV + (c * 8 + 2 * e) * VROW + pos * 2) = (bf16_t)(rv[u][e] & 0xffffu);
                        *(bf16_t*)(ldsV + (c * 8 + 2 * e + 1) * VROW + pos * 2) = (bf16_t)(rv[u][e] >> 16);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(uint32_t*)(ldsV + (c * 4 + e) * VROW + pos * 4) = rv[u][e];
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

        // ---- S^T = K Q^T for the two 32-key blocks of this tile (raw, unscaled scores)
        f32x16_t s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const u32x4_t kf = *(const u32x4_t*)(ldsK + (kb * 32 + lq) * KROW + st * 32 + lh * 16);
                AttnMma<T>::mma(s[kb], kf, qf[st]);
            }
        }
        // ---- online softmax on raw scores (scale > 0): p = exp2(c2 * s - c2 * m); keys of this lane: kb*32 + 8*(r>>2) + 4*lh + (r&3)
        if (kv0 + KV_TILE > p.Nk) {       // tail tile only: mask keys beyond Nk
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + 8 * (r >> 2) + 4 * lh + (r & 3) >= p.Nk) s[kb][r] = -INFINITY;
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float mc = m_new * c2;
        if (__any(m_new != m_run)) {       // rescale only when some query's running max moved (wave-uniform branch)
            const float alpha = __builtin_amdgcn_exp2f(m_run * c2 - mc);      // m_run = -inf on the first tile -> 0
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
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c2, -mc));
                s[kb][r] = e;
                psum += e;
            }
        l_run += psum;

        // ---- O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {          // 16 keys per MFMA
                    u32x4_t pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf[e] = pack_bf2(s[kb][8 * g + 2 * e], s[kb][8 * g + 2 * e + 1]);
#pragma unroll
                    for (int i = 0; i < DVB; ++i) {
                        const u32x4_t vf = *(const u32x4_t*)(ldsV + (i * 32 + lq) * VROW + (kb * 32 + g * 16) * 2 + lh * 16);
                        AttnMma<T>::mma(o[i], vf, pf);
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {          // 8 keys per 4-MFMA group
                    u32x4_t pf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf[e] = as_u32(s[kb][4 * g + e]);
#pragma unroll
                    for (int i = 0; i < DVB; ++i) {
                        const u32x4_t vf = *(const u32x4_t*)(ldsV + (i * 32 + lq) * VROW + (kb * 32 + g * 8) * 4 + lh * 16);
                        AttnMma<T>::mma(o[i], vf, pf);
                    }
                }
            }
        }
        if (t + 1 < ntiles) {
            if (NS == 1) __syncthreads();                  // single stage: every wave must be done reading tile t
            store_kv(NS == 2 ? ((t + 1) & 1) : 0);          // NS == 2: that stage was last read in iteration t-1
        }
        __syncthreads();
    }
    // ---- normalise and store: lane holds O[q][dv = i*32 + 8*(r>>2) + 4*lh + (r&3)]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + lq;
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
                        w[0] = pack_bf2(o[i][4 * g] * inv, o[i][4 * g + 1] * inv);
                        w[1] = pack_bf2(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
                        *(u32x2_t*)dst = w;
                    } else {
                        f32x4_t w = {o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
                        *(f32x4_t*)dst = w;
                    }
                }
            }
    }
}

template <typename T, int D>
static int launch_attn(const AttnParams& p, int B, hipStream_t st) {
    constexpr int VEC = elem<T>::VEC, KSTEP = 2 * VEC, STEPS = (D + KSTEP - 1) / KSTEP, DVB = (D + 31) / 32;
    constexpr int KROW = STEPS * 32 + (((STEPS * 2) & 1) ? 0 : 16);
    constexpr int VROW = KV_TILE * (int)sizeof(T) + 16;
    constexpr int tile_bytes = KV_TILE * KROW + DVB * 32 * VROW;
    constexpr int smem = (2 * tile_bytes <= 160 * 1024 ? 2 : 1) * tile_bytes;
    auto k = attention_kernel<T, D>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
    dim3 grid((p.Nq + 127) / 128, B * p.heads);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, st, p);
    RF_LAUNCH_CHECK("rf_attention");
    return 0;
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
    RF_CHECK(dtype == RF_F32 || dtype == RF_BF16, "rf_attention: bad dtype %d", dtype);
    RF_CHECK(q && k && v && out && B > 0 && heads > 0 && Nq > 0 && Nk > 0, "rf_attention: bad arguments");
    const int vec = dtype == RF_F32 ? 4 : 8;
    RF_CHECK(d % 8 == 0 && ldq % vec == 0 && ldk % vec == 0 && ldv % vec == 0 && ldo % 4 == 0, "rf_attention: d/ld alignment (d=%d)", d);
    RF_CHECK(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0, "rf_attention: operands must be 16-byte aligned");
    RF_CHECK((long long)B * heads <= 65535, "rf_attention: B*heads too large");
    AttnParams p;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.heads = heads; p.d = d; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.sq = sq; p.sk = sk; p.sv = sv; p.so = so;
    p.scale_log2e = scale * 1.4426950408889634f;
    if (dtype == RF_F32) return dispatch_attn<float>(p, B, (hipStream_t)stream);
    return dispatch_attn<bf16_t>(p, B, (hipStream_t)stream);
}
