// GroupNorm(32) (+SiLU), LayerNorm and row softmax for channels-last activations (HBM-bound
// side kernels: 16-byte vector loads, wavefront reductions, fp64 cross-block sums).
#include <stdarg.h>

#include "common.h"

namespace rf {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int GN_SLOTS = 4;     // channel vectors per thread (C <= 4 * 256 * VEC)
constexpr int GN_THREADS = 256;

template <typename T> struct gn_acc { typedef float type; };
template <> struct gn_acc<float> { typedef double type; };   // fp32 (parity) path: fp64 partial sums throughout

// thread t owns channel vectors {tv + q*TV}; pixel lanes stride over the chunk's pixels
template <typename T>
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const T* __restrict__ x, int HW, int C, int ldx, int nchunks,
                                                              double* __restrict__ partial) {
    typedef typename gn_acc<T>::type acc_t;
    constexpr int VEC = elem<T>::VEC;
    const int nvec = C / VEC;
    const int TV = nvec < GN_THREADS ? nvec : GN_THREADS;
    const int PL = GN_THREADS / TV;                 // pixel lanes
    const int tv = threadIdx.x % TV, pl = threadIdx.x / TV;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int ppc = (HW + nchunks - 1) / nchunks;
    const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
    const int cpg = C / 32;

    acc_t s[GN_SLOTS][VEC], ss[GN_SLOTS][VEC];
#pragma unroll
    for (int q = 0; q < GN_SLOTS; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[q][e] = ss[q][e] = 0;

    if (pl < PL) {
        for (int p = p0 + pl; p < p1; p += PL) {
            const T* row = x + ((long long)b * HW + p) * ldx;
#pragma unroll
            for (int q = 0; q < GN_SLOTS; ++q) {
                const int v = tv + q * TV;
                if (v < nvec) {
                    float f[VEC];
                    unpack16<T>(*(const u32x4_t*)(row + v * VEC), f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { s[q][e] += f[e]; ss[q][e] += (acc_t)f[e] * f[e]; }
                }
            }
        }
    }
    // deterministic two-stage reduction through LDS (no atomics): per-(pixel lane, channel) sums,
    // then per group over (pixel lanes x channels-per-group)
    extern __shared__ __attribute__((aligned(16))) char gn_smem[];
    double* chs = (double*)gn_smem;             // [PL][C]
    double* chq = chs + (size_t)PL * C;         // [PL][C]
    if (pl < PL) {
#pragma unroll
        for (int q = 0; q < GN_SLOTS; ++q) {
            const int v = tv + q * TV;
            if (v < nvec) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    chs[pl * C + v * VEC + e] = (double)s[q][e];
                    chq[pl * C + v * VEC + e] = (double)ss[q][e];
                }
            }
        }
    }
    __syncthreads();
    __shared__ double red[GN_THREADS][2];
    {
        const int g = threadIdx.x & 31, part = threadIdx.x >> 5;
        double a = 0.0, qq = 0.0;
        const int n = PL * cpg;
        for (int e = part; e < n; e += GN_THREADS / 32) {
            const int pp = e / cpg, cc = e - pp * cpg;
            a += chs[pp * C + g * cpg + cc];
            qq += chq[pp * C + g * cpg + cc];
        }
        red[threadIdx.x][0] = a;
        red[threadIdx.x][1] = qq;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double sa = 0.0, sq = 0.0;
        for (int k = 0; k < GN_THREADS / 32; ++k) { sa += red[threadIdx.x + 32 * k][0]; sq += red[threadIdx.x + 32 * k][1]; }
        double* o = partial + (((long long)b * nchunks + chunk) * 32 + threadIdx.x) * 2;
        o[0] = sa;
        o[1] = sq;
    }
}

// NS = channel vectors per thread (1 for C <= 256 * VEC: every UNet / VAE layer but the widest concats): sizing the per-channel scale /
// shift registers for the worst case (4 slots = 128 registers) halved the occupancy of this HBM-bound pass.
template <typename T, typename TO, int NS>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const T* __restrict__ x, int HW, int C, int ldx, int nchunks,
                                                              const double* __restrict__ partial, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int do_silu,
                                                              TO* __restrict__ out, int ldo, int achunks, int split,
                                                              fp8_t* __restrict__ sc_out = nullptr, int lds_sc = 0) {
    constexpr int VEC = elem<T>::VEC;
    const int nvec = C / VEC;
    const int TV = nvec < GN_THREADS ? nvec : GN_THREADS;
    const int PL = GN_THREADS / TV;
    const int tv = threadIdx.x % TV, pl = threadIdx.x / TV;
    const int b = blockIdx.y;
    const int cpg = C / 32;

    __shared__ float mean_s[32], rstd_s[32];
    __shared__ double red[GN_THREADS][2];
    // Everything that does not depend on the statistics is fetched first -- affine parameters and this thread's first pixel --
    // so that the small low-resolution launches pay ONE global-memory latency, not three in a row (partials -> gamma/beta -> x)
    const int ppc = (HW + achunks - 1) / achunks;
    const int p0 = blockIdx.x * ppc, p1 = min(HW, p0 + ppc);
    float gm[NS][VEC], bt[NS][VEC];
    u32x4_t raw0[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int v = tv + q * TV;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { gm[q][e] = 0.f; bt[q][e] = 0.f; }
        raw0[q] = u32x4_t{0u, 0u, 0u, 0u};
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
                const f32x4_t g4 = *(const f32x4_t*)(gamma + v * VEC + e), b4 = *(const f32x4_t*)(beta + v * VEC + e);
                gm[q][e] = g4[0]; gm[q][e + 1] = g4[1]; gm[q][e + 2] = g4[2]; gm[q][e + 3] = g4[3];
                bt[q][e] = b4[0]; bt[q][e + 1] = b4[1]; bt[q][e + 2] = b4[2]; bt[q][e + 3] = b4[3];
            }
            if (pl < PL && p0 + pl < p1) raw0[q] = *(const u32x4_t*)(x + ((long long)b * HW + p0 + pl) * ldx + v * VEC);
        }
    }
    {   // reduce the per-chunk partials of sample b: thread -> (group, part)
        const int g = threadIdx.x & 31, part = threadIdx.x >> 5;
        double a = 0.0, q = 0.0;
#pragma unroll 4
        for (int c = part; c < nchunks; c += GN_THREADS / 32) {
            const double* pp = partial + (((long long)b * nchunks + c) * 32 + g) * 2;
            a += pp[0];
            q += pp[1];
        }
        red[threadIdx.x][0] = a;
        red[threadIdx.x][1] = q;
        __syncthreads();
        if (threadIdx.x < 32) {
            double sa = 0.0, sq = 0.0;
            for (int k = 0; k < GN_THREADS / 32; ++k) { sa += red[threadIdx.x + 32 * k][0]; sq += red[threadIdx.x + 32 * k][1]; }
            const double n = (double)HW * cpg;
            const double mean = sa / n;
            double var = sq / n - mean * mean;
            if (var < 0.0) var = 0.0;
            mean_s[threadIdx.x] = (float)mean;
            rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    float sc[NS][VEC], sh[NS][VEC];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int v = tv + q * TV;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sc[q][e] = sh[q][e] = 0.f;
            if (v < nvec) {
                const int c = v * VEC + e, g = c / cpg;
                const float a = rstd_s[g] * gm[q][e];
                sc[q][e] = a;
                sh[q][e] = bt[q][e] - mean_s[g] * a;
            }
        }
    }
    if (pl >= PL) return;
    // one pixel ahead: the next pixel's vectors are in flight while the current ones are normalised
    u32x4_t cur[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) cur[q] = raw0[q];
    for (int p = p0 + pl; p < p1; p += PL) {
        const T* nrow = x + ((long long)b * HW + p + PL) * ldx;
        TO* orow = out + ((long long)b * HW + p) * ldo;
        u32x4_t nxt[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int v = tv + q * TV;
            nxt[q] = u32x4_t{0u, 0u, 0u, 0u};
            if (v < nvec && p + PL < p1) nxt[q] = *(const u32x4_t*)(nrow + v * VEC);
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int v = tv + q * TV;
            if (v < nvec) {
                float f[VEC];
                unpack16<T>(cur[q], f);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float y = f[e] * sc[q][e] + sh[q][e];
                    f[e] = do_silu ? silu_exact(y) : y;
                }
                if constexpr (sizeof(TO) == 1) {             // bf16 in -> fp8 out + one E8M0 scale per 32-channel block (4 consecutive threads)
                    quant_block8(f, orow + v * VEC, sc_out + ((long long)b * HW + p) * lds_sc + (v >> 2), (v & 3) == 0);
                } else if constexpr (sizeof(TO) == sizeof(T)) {
                    *(u32x4_t*)(orow + v * VEC) = pack16<TO>(f);
                } else if constexpr (sizeof(TO) == 4) {      // bf16 in -> fp32 out: two vectors
                    *(u32x4_t*)(orow + v * VEC) = pack16<float>(f);
                    *(u32x4_t*)(orow + v * VEC + 4) = pack16<float>(f + 4);
                } else {                                     // fp32 in -> bf16 out: half vector
                    u32x2_t h, l;
                    split4_bf16(f, h, l);
                    *(u32x2_t*)(orow + v * VEC) = h;
                    if (split) *(u32x2_t*)(orow + C + v * VEC) = l;      // RF_BF16X3: the lo plane of the pixel lies C elements behind
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) cur[q] = nxt[q];
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 6;
constexpr int LN_RPW = 1;          // rows per wave.  4 (loads of four rows in flight per lane) measured -3 % on the 65536-row launches alone and
                                   // +14 % on the step's 32 launches in situ (the 1024 / 4096-row levels lose their parallelism): 1 it stays
// NV = 16-byte vectors per lane (C <= NV * 64 * VEC); two-pass statistics in registers, fp32.
template <typename T, typename TO, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int M, int C, int ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, TO* __restrict__ out, int ldo,
                                                        fp8_t* __restrict__ sc_out = nullptr, int lds_sc = 0, int split = 0) {
    constexpr int VEC = elem<T>::VEC;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
    const int lane = threadIdx.x & 63;
    if (row0 >= M) return;
    const int nvec = C / VEC;
    float f[LN_RPW][NV][VEC];
    u32x4_t raw[LN_RPW][NV];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r)
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int v = lane + q * 64;
            raw[r][q] = u32x4_t{0u, 0u, 0u, 0u};
            if (v < nvec && row0 + r < M) raw[r][q] = *(const u32x4_t*)(x + (long long)(row0 + r) * ldx + v * VEC);
        }
    // affine parameters of this lane's columns (the same for every row)
    float g[NV][VEC], bt[NV][VEC];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int v = lane + q * 64;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {                 // 16-byte loads (element-wise conditional loads are 2 x VEC load instructions per row)
            f32x4_t g4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if (v < nvec) { g4 = *(const f32x4_t*)(gamma + v * VEC + e); b4 = *(const f32x4_t*)(beta + v * VEC + e); }
            g[q][e] = g4[0]; g[q][e + 1] = g4[1]; g[q][e + 2] = g4[2]; g[q][e + 3] = g4[3];
            bt[q][e] = b4[0]; bt[q][e + 1] = b4[1]; bt[q][e + 2] = b4[2]; bt[q][e + 3] = b4[3];
        }
    }
    float mean[LN_RPW], rstd[LN_RPW];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            unpack16<T>(raw[r][q], f[r][q]);
            if (lane + q * 64 < nvec) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) sum += f[r][q][e];
            }
        }
        mean[r] = wave_sum(sum) / (float)C;
    }
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        float sq = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (lane + q * 64 < nvec) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { const float d = f[r][q][e] - mean[r]; sq += d * d; }
            }
        }
        rstd[r] = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
    }
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
        if (row0 + r >= M) break;
        TO* orow = out + (long long)(row0 + r) * ldo;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int v = lane + q * 64;
            if (v < nvec) {
                float y[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) y[e] = (f[r][q][e] - mean[r]) * rstd[r] * g[q][e] + bt[q][e];
                if constexpr (sizeof(TO) == 1) {
                    quant_block8(y, orow + v * VEC, sc_out + (long long)(row0 + r) * lds_sc + (v >> 2), (v & 3) == 0);
                } else if constexpr (sizeof(TO) == sizeof(T)) {
                    *(u32x4_t*)(orow + v * VEC) = pack16<TO>(y);
                } else if constexpr (sizeof(TO) == 4) {
                    *(u32x4_t*)(orow + v * VEC) = pack16<float>(y);
                    *(u32x4_t*)(orow + v * VEC + 4) = pack16<float>(y + 4);
                } else {
                    u32x2_t h, l;
                    split4_bf16(y, h, l);
                    *(u32x2_t*)(orow + v * VEC) = h;
                    if (split) *(u32x2_t*)(orow + C + v * VEC) = l;          // RF_BF16X3: [C hi | C lo] per row
                }
            }
        }
    }
}

// one block per row; three sweeps (max, sum, write) -- rows are L2-resident between sweeps
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int cols, int ld) {
    float* r = x + (long long)blockIdx.x * ld;
    __shared__ float red[4];
    float m = -INFINITY;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        const f32x4_t v = *(const f32x4_t*)(r + c);
        m = fmaxf(fmaxf(fmaxf(m, v[0]), fmaxf(v[1], v[2])), v[3]);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        const f32x4_t v = *(const f32x4_t*)(r + c);
        s += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        f32x4_t v = *(const f32x4_t*)(r + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = expf(v[e] - m) * inv;
        *(f32x4_t*)(r + c) = v;
    }
}

}  // namespace rf

using namespace rf;

extern "C" const char* rf_last_error(void) { return rf::g_err; }
extern "C" int rf_version(void) { return 101; }          // 101: RF_F16 (the fp16 throughput mode), `dtype` in rf_ffn_desc / rf_stem_desc / rf_gn_silu_conv3x3_small

static int gn_check(const char* name, int dtype, int C, int ldx, int nchunks) {
    const int vec = dtype == RF_F32 ? 4 : 8;
    RF_CHECK(dtype == RF_F32 || dtype == RF_BF16 || dtype == RF_F16, "%s: bad dtype %d", name, dtype);
    RF_CHECK(C % 32 == 0 && C % vec == 0 && ldx % vec == 0, "%s: C=%d ld=%d must be multiples of 32 and %d", name, C, ldx, vec);
    RF_CHECK(C / vec <= GN_SLOTS * GN_THREADS, "%s: C=%d too large", name, C);
    RF_CHECK(nchunks >= 1, "%s: nchunks=%d", name, nchunks);
    return 0;
}

extern "C" int rf_groupnorm_stats(int dtype, const void* x, int B, int HW, int C, int ldx, int nchunks, double* partial, void* stream) {
    if (gn_check("rf_groupnorm_stats", dtype, C, ldx, nchunks)) return 1;
    RF_CHECK(x && partial && B > 0 && HW > 0, "rf_groupnorm_stats: bad arguments");
    dim3 grid(nchunks, B);
    const int vec = dtype == RF_F32 ? 4 : 8;
    const int nvec = C / vec, TV = nvec < GN_THREADS ? nvec : GN_THREADS, PL = GN_THREADS / TV;
    const size_t smem = (size_t)2 * PL * C * sizeof(double);
    RF_CHECK(smem <= 64 * 1024, "rf_groupnorm_stats: C=%d needs %zu B of LDS", C, smem);
    if (dtype == RF_F32) hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(GN_THREADS), smem, (hipStream_t)stream, (const float*)x, HW, C, ldx, nchunks, partial);
    else if (dtype == RF_F16) hipLaunchKernelGGL(gn_stats_kernel<f16_t>, grid, dim3(GN_THREADS), smem, (hipStream_t)stream, (const f16_t*)x, HW, C, ldx, nchunks, partial);
    else hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(GN_THREADS), smem, (hipStream_t)stream, (const bf16_t*)x, HW, C, ldx, nchunks, partial);
    RF_LAUNCH_CHECK("rf_groupnorm_stats");
    return 0;
}

// sum the chunk partials of every (sample, group) in a fixed order: [B][nchunks][32][2] -> [B][1][32][2]
__global__ __launch_bounds__(GN_THREADS) void gn_finalize_kernel(const double* __restrict__ in, int nchunks, double* __restrict__ out) {
    const int b = blockIdx.x, g = threadIdx.x & 31, part = threadIdx.x >> 5;
    __shared__ double red[GN_THREADS][2];
    double a = 0.0, q = 0.0;
    for (int c = part; c < nchunks; c += GN_THREADS / 32) {
        const double* pp = in + (((long long)b * nchunks + c) * 32 + g) * 2;
        a += pp[0];
        q += pp[1];
    }
    red[threadIdx.x][0] = a;
    red[threadIdx.x][1] = q;
    __syncthreads();
    if (threadIdx.x < 32) {
        double sa = 0.0, sq = 0.0;
        for (int k = 0; k < GN_THREADS / 32; ++k) { sa += red[threadIdx.x + 32 * k][0]; sq += red[threadIdx.x + 32 * k][1]; }
        out[((long long)b * 32 + threadIdx.x) * 2] = sa;
        out[((long long)b * 32 + threadIdx.x) * 2 + 1] = sq;
    }
}

extern "C" int rf_groupnorm_finalize(const double* partial_in, int B, int nchunks, double* partial_out, void* stream) {
    RF_CHECK(partial_in && partial_out && B > 0 && nchunks >= 1, "rf_groupnorm_finalize: bad arguments");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(GN_THREADS), 0, (hipStream_t)stream, partial_in, nchunks, partial_out);
    RF_LAUNCH_CHECK("rf_groupnorm_finalize");
    return 0;
}

extern "C" int rf_groupnorm_apply(int dtype, const void* x, int B, int HW, int C, int ldx, int nchunks, const double* partial,
                                  const float* gamma, const float* beta, float eps, int silu, int out_dtype, void* out, int ldo, void* stream) {
    if (gn_check("rf_groupnorm_apply", dtype, C, ldx, nchunks)) return 1;
    RF_CHECK(x && partial && gamma && beta && out && B > 0 && HW > 0, "rf_groupnorm_apply: bad arguments");
    RF_CHECK((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "rf_groupnorm_apply: gamma / beta must be 16-byte aligned");
    const int split = out_dtype == RF_BF16X3 ? 1 : 0;      // [C hi | C lo] bf16 pairs per pixel (ldo >= 2C), fp32 input
    RF_CHECK(out_dtype == RF_F32 || out_dtype == RF_BF16 || out_dtype == RF_F16 || split, "rf_groupnorm_apply: bad out_dtype");
    RF_CHECK((dtype == RF_F16) == (out_dtype == RF_F16) || (dtype == RF_F16 && out_dtype == RF_F32), "rf_groupnorm_apply: fp16 goes to fp16 or fp32 (dtype %d, out_dtype %d)", dtype, out_dtype);
    RF_CHECK(!split || (dtype == RF_F32 && ldo >= 2 * C), "rf_groupnorm_apply: split-bf16 output needs fp32 input and ldo >= 2C");
    RF_CHECK(ldo % 8 == 0, "rf_groupnorm_apply: ldo=%d must be a multiple of 8", ldo);
    // about 4 blocks per CU in total, down to 2 pixels per block: the launches of the 16x16 / 8x8 levels are latency chains (statistics -> scale / shift -> pixels
    // one after the other), 16 pixels per block left them at ~10 us whatever the tensor's size (2 pixels: -0.85 % per batch, profiles/r04ab_gn_apply_blocks.txt)
    int achunks = (1024 + B - 1) / B;
    const int maxc = (HW + 1) / 2;
    if (achunks > maxc) achunks = maxc;
    if (achunks < 1) achunks = 1;
    dim3 grid(achunks, B);
    hipStream_t st = (hipStream_t)stream;
    const int ns = (C / (dtype == RF_F32 ? 4 : 8) + GN_THREADS - 1) / GN_THREADS;      // channel vectors per thread
#define GN_APPLY_(T, TO, NS_) hipLaunchKernelGGL((gn_apply_kernel<T, TO, NS_>), grid, dim3(GN_THREADS), 0, st, (const T*)x, HW, C, ldx, nchunks, partial, gamma, beta, eps, silu, (TO*)out, ldo, achunks, split)
#define GN_APPLY(T, TO) { if (ns <= 1) GN_APPLY_(T, TO, 1); else if (ns <= 2) GN_APPLY_(T, TO, 2); else GN_APPLY_(T, TO, GN_SLOTS); }
    if (dtype == RF_F32 && out_dtype == RF_F32) GN_APPLY(float, float)
    else if (dtype == RF_F16 && out_dtype == RF_F32) GN_APPLY(f16_t, float)
    else if (dtype == RF_F16) GN_APPLY(f16_t, f16_t)
    else if (dtype == RF_F32) GN_APPLY(float, bf16_t)
    else if (out_dtype == RF_F32) GN_APPLY(bf16_t, float)
    else GN_APPLY(bf16_t, bf16_t)
#undef GN_APPLY
#undef GN_APPLY_
    RF_LAUNCH_CHECK("rf_groupnorm_apply");
    return 0;
}

// ---- GroupNorm folded into the weights of the Linear / 1x1 conv behind it (SpatialTransformer `norm` -> `proj_in`, attention.py:262-266, 276-279).
// Block (x, s): the statistics of sample s (the preamble of gn_apply_kernel) and GN_FOLD_RPW rows n of W per wave: lane pairs
// of columns k, W'[s][n][k] = W[n][k] rstd[g(k)] gamma[k] rounded to the GEMM's operand type, and the row's per-sample constant
// r[s][n] = bias[n] + sum_k W[n][k] beta[k] - sum_k W'[s][n][k] mean[g(k)]   (with the ROUNDED W': what the matrix pipe will accumulate against x).
constexpr int GN_FOLD_RPW = 2;            // rows per wave: 8 rows per block -- the launch is latency-bound (a W row is 1.3-2.5 KB), so many small blocks
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// NIT = iterations of 8 columns per lane (C <= 512 NIT): 16-byte stores of W'.  Everything that does not depend on the statistics -- the rows of W,
// gamma, beta -- is in flight before the statistics are reduced (one global-memory latency for the block, not three in a row).
template <typename TO, int NIT>
__global__ __launch_bounds__(GN_THREADS) void gn_fold_linear_kernel(const float* __restrict__ W, int N, int C, int HW, int nchunks,
                                                                    const double* __restrict__ partial, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, const float* __restrict__ bias, float eps,
                                                                    TO* __restrict__ wout, float* __restrict__ rvout) {
    const int b = blockIdx.y, cpg = C / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * (GN_THREADS / 64) + wave) * GN_FOLD_RPW;
    f32x4_t w8[GN_FOLD_RPW][NIT][2], g8[NIT][2], b8[NIT][2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = 8 * lane + 512 * it;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            g8[it][h] = b8[it][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (k < C) { g8[it][h] = *(const f32x4_t*)(gamma + k + 4 * h); b8[it][h] = *(const f32x4_t*)(beta + k + 4 * h); }
#pragma unroll
            for (int r = 0; r < GN_FOLD_RPW; ++r) {
                w8[r][it][h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (k < C && n0 + r < N) w8[r][it][h] = *(const f32x4_t*)(W + (long long)(n0 + r) * C + k + 4 * h);
            }
        }
    }
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ double red[GN_THREADS][2];
    {
        const int g = threadIdx.x & 31, part = threadIdx.x >> 5;
        double a = 0.0, q = 0.0;
#pragma unroll 4
        for (int c = part; c < nchunks; c += GN_THREADS / 32) {
            const double* pp = partial + (((long long)b * nchunks + c) * 32 + g) * 2;
            a += pp[0];
            q += pp[1];
        }
        red[threadIdx.x][0] = a;
        red[threadIdx.x][1] = q;
        __syncthreads();
        if (threadIdx.x < 32) {
            double sa = 0.0, sq = 0.0;
            for (int k = 0; k < GN_THREADS / 32; ++k) { sa += red[threadIdx.x + 32 * k][0]; sq += red[threadIdx.x + 32 * k][1]; }
            const double n = (double)HW * cpg;
            const double mean = sa / n;
            double var = sq / n - mean * mean;
            if (var < 0.0) var = 0.0;
            mean_s[threadIdx.x] = (float)mean;
            rstd_s[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();
    }
    float t[GN_FOLD_RPW];
#pragma unroll
    for (int r = 0; r < GN_FOLD_RPW; ++r) t[r] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int k = 8 * lane + 512 * it;
        if (k < C) {
            float a[8], m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = (k + e) / cpg;
                a[e] = rstd_s[g] * g8[it][e >> 2][e & 3];
                m[e] = mean_s[g];
            }
#pragma unroll
            for (int r = 0; r < GN_FOLD_RPW; ++r) {
                if (n0 + r < N) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = w8[r][it][e >> 2][e & 3] * a[e];
                    TO* const wo = wout + ((long long)b * N + n0 + r) * C + k;
                    if constexpr (sizeof(TO) == 2) {
                        const u32x4_t pk = pack16<TO>(v);
                        *(u32x4_t*)wo = pk;
                        unpack16<TO>(pk, v);          // the values as rounded: what the matrix pipe will multiply x with
                    } else {
                        *(f32x4_t*)wo = f32x4_t{v[0], v[1], v[2], v[3]};
                        *(f32x4_t*)(wo + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[r] += w8[r][it][e >> 2][e & 3] * b8[it][e >> 2][e & 3] - v[e] * m[e];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < GN_FOLD_RPW; ++r) {
        float v = t[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && n0 + r < N) rvout[(long long)b * N + n0 + r] = v + (bias ? bias[n0 + r] : 0.f);
    }
}

extern "C" int rf_groupnorm_fold_linear(const float* W, int N, int C, int B, int HW, int nchunks, const double* partial, const float* gamma,
                                        const float* beta, const float* bias, float eps, int out_dtype, void* w_out, float* rowvec_out, void* stream) {
    RF_CHECK(W && partial && gamma && beta && w_out && rowvec_out && N > 0 && B > 0 && HW > 0 && nchunks >= 1, "rf_groupnorm_fold_linear: bad arguments");
    RF_CHECK(C > 0 && C % 32 == 0 && C <= 1536, "rf_groupnorm_fold_linear: C=%d must be a multiple of 32, at most 1536", C);
    RF_CHECK(out_dtype == RF_BF16 || out_dtype == RF_F16 || out_dtype == RF_F32, "rf_groupnorm_fold_linear: bad out_dtype %d", out_dtype);
    RF_CHECK((((uintptr_t)W | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w_out) & 15) == 0, "rf_groupnorm_fold_linear: W / gamma / beta / w_out must be 16-byte aligned");
    constexpr int RPB = (GN_THREADS / 64) * GN_FOLD_RPW;
    dim3 grid((N + RPB - 1) / RPB, B);
    hipStream_t st = (hipStream_t)stream;
#define GN_FOLD_(TO, NIT) hipLaunchKernelGGL((gn_fold_linear_kernel<TO, NIT>), grid, dim3(GN_THREADS), 0, st, W, N, C, HW, nchunks, partial, gamma, beta, bias, eps, (TO*)w_out, rowvec_out)
#define GN_FOLD(TO) { if (C <= 512) GN_FOLD_(TO, 1); else if (C <= 1024) GN_FOLD_(TO, 2); else GN_FOLD_(TO, 3); }
    if (out_dtype == RF_BF16) GN_FOLD(bf16_t) else if (out_dtype == RF_F16) GN_FOLD(f16_t) else GN_FOLD(float)
#undef GN_FOLD
#undef GN_FOLD_
    RF_LAUNCH_CHECK("rf_groupnorm_fold_linear");
    return 0;
}

extern "C" int rf_groupnorm_apply_fp8(const void* x, int B, int HW, int C, int ldx, int nchunks, const double* partial, const float* gamma,
                                      const float* beta, float eps, int silu, void* q, int ldq, void* scale, int lds, void* stream) {
    if (gn_check("rf_groupnorm_apply_fp8", RF_BF16, C, ldx, nchunks)) return 1;
    RF_CHECK(x && partial && gamma && beta && q && scale && B > 0 && HW > 0, "rf_groupnorm_apply_fp8: bad arguments");
    RF_CHECK((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)q & 7) == 0, "rf_groupnorm_apply_fp8: gamma / beta / q alignment");
    RF_CHECK(ldq >= C && ldq % 8 == 0 && lds >= C / 32, "rf_groupnorm_apply_fp8: ldq=%d lds=%d for C=%d", ldq, lds, C);
    int achunks = (1024 + B - 1) / B;
    const int maxc = (HW + 1) / 2;          // (as rf_groupnorm_apply)
    if (achunks > maxc) achunks = maxc;
    if (achunks < 1) achunks = 1;
    dim3 grid(achunks, B);
    const int ns = (C / 8 + GN_THREADS - 1) / GN_THREADS;
#define GNQ_(NS_) hipLaunchKernelGGL((gn_apply_kernel<bf16_t, fp8_t, NS_>), grid, dim3(GN_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x, HW, C, ldx, nchunks, partial, gamma, beta, eps, silu, (fp8_t*)q, ldq, achunks, 0, (fp8_t*)scale, lds)
    if (ns <= 1) GNQ_(1); else if (ns <= 2) GNQ_(2); else GNQ_(GN_SLOTS);
#undef GNQ_
    RF_LAUNCH_CHECK("rf_groupnorm_apply_fp8");
    return 0;
}

extern "C" int rf_layernorm_fp8(const void* x, int M, int C, int ldx, const float* gamma, const float* beta, float eps, void* q, int ldq, void* scale,
                                int lds, void* stream) {
    RF_CHECK(x && gamma && beta && q && scale && M > 0, "rf_layernorm_fp8: bad arguments");
    RF_CHECK((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && ((uintptr_t)q & 7) == 0, "rf_layernorm_fp8: gamma / beta / q alignment");
    RF_CHECK(C % 32 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && ldq >= C && lds >= C / 32 && C / 8 <= 64 * LN_MAXV, "rf_layernorm_fp8: C=%d ldx=%d ldq=%d lds=%d unsupported", C, ldx, ldq, lds);
    dim3 grid((M + 4 * LN_RPW - 1) / (4 * LN_RPW));
    const int nv = (C / 8 + 63) / 64;
#define LNQ_(NV) hipLaunchKernelGGL((layernorm_kernel<bf16_t, fp8_t, NV>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, M, C, ldx, gamma, beta, eps, (fp8_t*)q, ldq, (fp8_t*)scale, lds)
    if (nv <= 1) LNQ_(1); else if (nv <= 2) LNQ_(2); else if (nv <= 3) LNQ_(3); else LNQ_(LN_MAXV);
#undef LNQ_
    RF_LAUNCH_CHECK("rf_layernorm_fp8");
    return 0;
}

// x [M, C] bf16 (row pitch ldx) -> e4m3fn bytes q [M][ldq] + one E8M0 scale byte per 32-channel block, scale [M][lds]
__global__ __launch_bounds__(256) void quantize_fp8_act_kernel(const bf16_t* __restrict__ x, long long M, int C, int ldx, fp8_t* __restrict__ q, int ldq,
                                                               fp8_t* __restrict__ sc, int lds) {
    const int vpr = C >> 3;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < M * vpr;            // (vpr is a multiple of 4: the lanes of a quad are on or off together)
    const long long r = on ? i / vpr : 0;
    const int v = on ? (int)(i - r * vpr) : 0;
    float f[8];
    u32x4_t raw = {0u, 0u, 0u, 0u};
    if (on) raw = *(const u32x4_t*)(x + r * ldx + v * 8);
    unpack16<bf16_t>(raw, f);
    if (on) quant_block8(f, q + r * ldq + v * 8, sc + r * lds + (v >> 2), (v & 3) == 0);
}

extern "C" int rf_quantize_fp8_act(const void* x, int64_t M, int C, int ldx, void* q, int ldq, void* scale, int lds, void* stream) {
    RF_CHECK(x && q && scale && M > 0 && C > 0 && C % 32 == 0 && ldx % 8 == 0 && ldq % 8 == 0 && ldq >= C && lds >= C / 32,
             "rf_quantize_fp8_act: bad arguments M=%lld C=%d ldx=%d ldq=%d lds=%d", (long long)M, C, ldx, ldq, lds);
    RF_CHECK((((uintptr_t)x & 15) | ((uintptr_t)q & 7)) == 0, "rf_quantize_fp8_act: operand alignment");
    const long long n = (long long)M * (C / 8);
    hipLaunchKernelGGL(quantize_fp8_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (long long)M, C, ldx,
                       (fp8_t*)q, ldq, (fp8_t*)scale, lds);
    RF_LAUNCH_CHECK("rf_quantize_fp8_act");
    return 0;
}

extern "C" int rf_layernorm(int dtype, const void* x, int M, int C, int ldx, const float* gamma, const float* beta, float eps,
                            int out_dtype, void* out, int ldo, void* stream) {
    const int vec = dtype == RF_F32 ? 4 : 8;
    RF_CHECK(dtype == RF_F32 || dtype == RF_BF16 || dtype == RF_F16, "rf_layernorm: bad dtype %d", dtype);
    const int split = out_dtype == RF_BF16X3 ? 1 : 0;          // split-bf16 pairs [C hi | C lo] per row (ldo >= 2C), fp32 input
    RF_CHECK(out_dtype == RF_F32 || out_dtype == RF_BF16 || out_dtype == RF_F16 || split, "rf_layernorm: bad out_dtype");
    RF_CHECK((dtype == RF_F16) == (out_dtype == RF_F16) || (dtype == RF_F16 && out_dtype == RF_F32), "rf_layernorm: fp16 goes to fp16 or fp32 (dtype %d, out_dtype %d)", dtype, out_dtype);
    RF_CHECK(!split || (dtype == RF_F32 && ldo >= 2 * C), "rf_layernorm: split-bf16 output needs fp32 input and ldo >= 2C");
    RF_CHECK(x && gamma && beta && out && M > 0, "rf_layernorm: bad arguments");
    RF_CHECK((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "rf_layernorm: gamma / beta must be 16-byte aligned");
    RF_CHECK(C % vec == 0 && ldx % vec == 0 && ldo % 8 == 0 && C / vec <= 64 * LN_MAXV, "rf_layernorm: C=%d ldx=%d ldo=%d unsupported", C, ldx, ldo);
    dim3 grid((M + 4 * LN_RPW - 1) / (4 * LN_RPW));
    hipStream_t st = (hipStream_t)stream;
    const int nv = (C / vec + 63) / 64;          // 16-byte vectors per lane
#define LN_(T, TO, NV) hipLaunchKernelGGL((layernorm_kernel<T, TO, NV>), grid, dim3(256), 0, st, (const T*)x, M, C, ldx, gamma, beta, eps, (TO*)out, ldo, (fp8_t*)nullptr, 0, split)
#define LN(T, TO) { if (nv <= 1) LN_(T, TO, 1); else if (nv <= 2) LN_(T, TO, 2); else if (nv <= 3) LN_(T, TO, 3); else LN_(T, TO, LN_MAXV); }
    if (dtype == RF_F32 && out_dtype == RF_F32) LN(float, float)
    else if (dtype == RF_F16 && out_dtype == RF_F32) LN(f16_t, float)
    else if (dtype == RF_F16) LN(f16_t, f16_t)
    else if (dtype == RF_F32) LN(float, bf16_t)
    else if (out_dtype == RF_F32) LN(bf16_t, float)
    else LN(bf16_t, bf16_t)
#undef LN
#undef LN_
    RF_LAUNCH_CHECK("rf_layernorm");
    return 0;
}

extern "C" int rf_softmax_rows(float* x, int rows, int cols, int ld, void* stream) {
    RF_CHECK(x && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0, "rf_softmax_rows: bad arguments rows=%d cols=%d ld=%d", rows, cols, ld);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, cols, ld);
    RF_LAUNCH_CHECK("rf_softmax_rows");
    return 0;
}
