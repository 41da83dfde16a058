// GroupNorm(32) + SiLU + 3x3 convolution (stride 1, pad 1) to a HANDFUL of output channels, fused: the UNet's `out` head
//   openaimodel.py:737-741  self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
// (320 -> 4 channels on the full-resolution tensor).  As an implicit GEMM this layer stages the activation nine times (one tap-shifted copy per
// filter tap: 364 MB through the fabric for a 42 MB tensor, 55 us at 27 TFLOP/s on a 128 x 64 tile that is 94 % padding) behind a normalisation
// pass that writes and re-reads the tensor (13 us).  Here the raw tensor is read ONCE:
//   kernel 1 (per 32 pixels and wave, pixels on lanes as in ffn.hip): the pixel's C raw values -> registers, scale / shift / SiLU per (sample, channel)
//            from the fused GroupNorm statistics, rounded to bf16 (what the normalisation pass would have stored), then
//            Y^T[9 No, pixel] = Wt[9 No, C] . act(X)^T on the matrix pipe -- the per-TAP partial products of every pixel (row = tap * No + o);
//   kernel 2 (one thread per output pixel): out[p, o] = bias[o] + sum over the taps whose source pixel lies inside the image of Y[p + offset(tap), tap, o].
// Zero padding costs nothing: a tap whose source pixel is outside the image is simply not added (act(0) is never formed).
// The products are the ones the unfused pair forms (bf16 activation x bf16 weight, fp32 accumulation); only the order of the fp32 additions differs.
#include "common.h"

namespace rf {

struct SmallConvParams {
    const bf16_t* x; int ldx;          // [B * HW][ldx] raw (un-normalised) activations
    int B, H, W, C, No;
    int nchunks; const double* partial;          // GroupNorm(32) partial sums of x: [B][nchunks][32][2] (rf_groupnorm_stats layout)
    const float* gamma; const float* beta; float eps; int silu;
    const bf16_t* w;                   // [No][9 C] conv weight, k = tap * C + c (ops.pack_conv_weight)
    float* y;                          // [B * HW][YP] per-tap partial products (workspace)
};
constexpr int SC_YP = 40;              // floats per pixel in y (9 No <= 36 used; 160-byte rows keep the 16-byte vectors aligned)

// kernel 1: block = 4 waves = 128 pixels of one sample
// (T = bf16_t or f16_t: the 16-bit operands are addressed as uint16_t, T selects the conversions and the MFMA)
template <int C, typename T = bf16_t>
__global__ __launch_bounds__(256) void gn_silu_taps_kernel(const SmallConvParams p) {
    constexpr int KS = C / 16;                                   // k-steps of 16 channels
    constexpr int WP = C * 2 + 16;                               // row pitch of the weight table in LDS (bytes): 16-byte skew against bank conflicts
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;                                       // [36][WP]  Wt rows (tap * No + o), bf16
    float* const scl = (float*)(smem + 36 * WP);                 // [C] scale, [C] shift of this block's sample
    float* const shl = scl + C;
    __shared__ float mean_s[32], rstd_s[32];
    __shared__ double red[256][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int HW = p.H * p.W;
    const int blocks_per_sample = (HW + 127) / 128;
    const int b = blockIdx.x / blocks_per_sample, pb = blockIdx.x - b * blocks_per_sample;
    const int pix = pb * 128 + wave * 32 + lrow;                 // pixel inside the sample
    const bool live = pix < HW;
    const long long grow = (long long)b * HW + pix;

    // ---- the pixel's raw values: issued first (one memory latency for the block: statistics and weight table follow while they fly)
    u32x4_t xq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        xq[s] = u32x4_t{0u, 0u, 0u, 0u};
        if (live) xq[s] = *(const u32x4_t*)(p.x + grow * p.ldx + s * 16 + lhalf * 8);
    }
    // ---- everything the scale / shift need is fetched NOW, beside the pixel values: the statistics partials of this thread's (group, part) and the
    // affine parameters of its channels (one memory latency for the block, not three in a row)
    double st_a = 0.0, st_q = 0.0;
    {
        const int g = tid & 31, part = tid >> 5;
        for (int c = part; c < p.nchunks; c += 8) {
            const double* pp = p.partial + (((long long)b * p.nchunks + c) * 32 + g) * 2;
            st_a += pp[0];
            st_q += pp[1];
        }
    }
    float gm0 = 0.f, bt0 = 0.f, gm1 = 0.f, bt1 = 0.f;
    if (tid < C) { gm0 = p.gamma[tid]; bt0 = p.beta[tid]; }
    if (tid + 256 < C) { gm1 = p.gamma[tid + 256]; bt1 = p.beta[tid + 256]; }
    static_assert(C <= 512, "two channels per thread");
    // ---- weight table: row r = tap * No + o  <-  w[o][tap * C .. + C)
    const int rows = 9 * p.No;
    for (int i = tid; i < rows * (C / 8); i += 256) {
        const int r = i / (C / 8), v = i - r * (C / 8);
        const int tap = r / p.No, o = r - tap * p.No;
        *(u32x4_t*)(wl + r * WP + v * 16) = *(const u32x4_t*)(p.w + ((long long)o * 9 + tap) * C + v * 8);
    }
    // ---- statistics of sample b -> per-channel scale / shift (the preamble of gn_apply_kernel, norm.hip)
    {
        red[tid][0] = st_a;
        red[tid][1] = st_q;
        __syncthreads();
        if (tid < 32) {
            double sa = 0.0, sq = 0.0;
            for (int k = 0; k < 8; ++k) { sa += red[tid + 32 * k][0]; sq += red[tid + 32 * k][1]; }
            const double n = (double)HW * (C / 32);
            const double mean = sa / n;
            double var = sq / n - mean * mean;
            if (var < 0.0) var = 0.0;
            mean_s[tid] = (float)mean;
            rstd_s[tid] = (float)(1.0 / sqrt(var + (double)p.eps));
        }
        __syncthreads();
        if (tid < C) {
            const int g2 = tid / (C / 32);
            const float a2 = rstd_s[g2] * gm0;
            scl[tid] = a2;
            shl[tid] = bt0 - mean_s[g2] * a2;
        }
        if (tid + 256 < C) {
            const int c = tid + 256, g2 = c / (C / 32);
            const float a2 = rstd_s[g2] * gm1;
            scl[c] = a2;
            shl[c] = bt1 - mean_s[g2] * a2;
        }
        __syncthreads();
    }
    // ---- normalise + SiLU in registers, rounded to bf16 as the normalisation pass stores them
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float f[8];
        unpack16<T>(xq[s], f);
        const int c0 = s * 16 + lhalf * 8;
        const f32x4_t a0 = *(const f32x4_t*)(scl + c0), a1 = *(const f32x4_t*)(scl + c0 + 4), h0 = *(const f32x4_t*)(shl + c0), h1 = *(const f32x4_t*)(shl + c0 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = f[e] * (e < 4 ? a0[e] : a1[e - 4]) + (e < 4 ? h0[e] : h1[e - 4]);
            if (p.silu) v = silu_exact(v);
            f[e] = v;
        }
        xq[s] = pack16<T>(f);
    }
    // ---- Y^T = Wt . act(X)^T : two 32-row blocks (rows >= 9 No read a clamped row: their results are never used)
    f32x16_t acc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const int r0 = min(lrow, rows - 1), r1 = min(32 + lrow, rows - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const u32x4_t w0 = *(const u32x4_t*)(wl + r0 * WP + (s * 16 + lhalf * 8) * 2), w1 = *(const u32x4_t*)(wl + r1 * WP + (s * 16 + lhalf * 8) * 2);
        mma16<T>(acc[0], w0, xq[s]);
        mma16<T>(acc[1], w1, xq[s]);
    }
    // accumulator register r of block k = row 32 k + (r & 3) + 8 (r >> 2) + 4 half of this lane's pixel: four 16-byte row runs per block
    if (live) {
        float* const yr = p.y + grow * SC_YP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + 4 * lhalf;
            if (row < rows) *(f32x4_t*)(yr + row) = f32x4_t{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
        }
        if (lhalf == 0 && 32 < rows) *(f32x4_t*)(yr + 32) = f32x4_t{acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
        if (lhalf == 1 && 36 < rows) *(f32x4_t*)(yr + 36) = f32x4_t{acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
    }
}

// kernel 2: out[p, o] = bias[o] + sum_{taps inside the image} y[p + (dy - 1) W + (dx - 1)][tap * No + o]
template <typename TO>
__global__ __launch_bounds__(256) void gather_taps_kernel(const float* __restrict__ y, int B, int H, int W, int No, const float* __restrict__ bias,
                                                          TO* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int HW = H * W;
    if (i >= (long long)B * HW) return;
    const int pix = (int)(i % HW), py = pix / W, px = pix - py * W;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = (bias && o < No) ? bias[o] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int sy = py + dy - 1, sx = px + dx - 1;
            if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
                const float* src = y + (i + (dy - 1) * W + (dx - 1)) * SC_YP + (dy * 3 + dx) * No;
                if (No == 4) {          // (16-byte aligned: 160-byte rows, 16-byte tap groups)
                    const f32x4_t v = *(const f32x4_t*)src;
                    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
                } else {
                    for (int o = 0; o < No; ++o) acc[o] += src[o];
                }
            }
        }
    for (int o = 0; o < No; ++o) elem<TO>::store(out + i * ldo + o, acc[o]);
}


// ---- The UNet's stem: 3x3 convolution (stride 1, pad 1) from a HANDFUL of input channels (9 stored in 16) to C = 32 NB channels
//   openaimodel.py:666-671  TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))
// As an implicit GEMM it is K = 144 of 128 x 64 tiles behind a statistics pass over its output (its two GroupNorm consumers -- the first ResBlock and
// the last decoder block's concat -- read different row sets of it, which the GEMM's fused statistics cannot serve).  Pixels on lanes: a lane pair
// holds the 9 x 16 input values of its pixel (one 16-byte load per tap and lane: the B fragments of nine k-steps, zero for taps outside the image),
// the whole weight table sits in LDS, O^T[C, pixel] = W[C, 144] X^T on the matrix pipe, bias + bf16 rounding + 16-byte stores from the accumulators.
// `dup_off`: under classifier-free guidance both batch halves of the input are the same latent -- the block computes one and stores it twice.
// GroupNorm(32) partial sums of the values as stored for up to three consumers (one chunk slot per 128-pixel block), as ffn.hip's tail epilogue.
struct StemParams {
    const bf16_t* x; int ldx;          // [B * HW][ldx], channels 0..15 of a pixel (pad channels zero)
    int B, H, W;
    const bf16_t* w;                   // [C][144], k = tap * 16 + c (ops.pack_conv_weight with cin_pad = 16)
    const float* bias;
    bf16_t* out; int ldo;
    long long dup_off;                 // elements from a row of `out` to its duplicate (0: none)
    double* gn_part[3]; int gn_cpg[3], gn_coff[3], gn_slot[3], gn_nch[3];
};

template <int NB, typename T = bf16_t>
__global__ __launch_bounds__(256) void stem_conv3x3_kernel(const StemParams p) {
    constexpr int C = 32 * NB;
    constexpr int WP = 144 * 2 + 16;                             // row pitch of the weight table in LDS (bytes), 16-byte skew
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem;                                       // [C][WP]
    float* const bl = (float*)(smem + C * WP);                   // [C] bias
    float* const gcs = bl + C;                                   // [2][4 waves][C] column sums / sums of squares per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int HW = p.H * p.W;
    const int blocks_per_sample = HW / 128;
    const int b = blockIdx.x / blocks_per_sample, pb = blockIdx.x - b * blocks_per_sample;
    const int pix = pb * 128 + wave * 32 + lrow;
    const int py = pix / p.W, px = pix - py * p.W;
    const long long grow = (long long)b * HW + pix;

    // ---- the pixel's nine taps: issued first
    u32x4_t xq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int sy = py + t / 3 - 1, sx = px + t % 3 - 1;
        xq[t] = u32x4_t{0u, 0u, 0u, 0u};
        if ((unsigned)sy < (unsigned)p.H && (unsigned)sx < (unsigned)p.W)
            xq[t] = *(const u32x4_t*)(p.x + (grow + (t / 3 - 1) * p.W + (t % 3 - 1)) * p.ldx + lhalf * 8);
    }
    // ---- weight table + bias: every load in flight before the first LDS write (one memory latency for the block, beside the taps')
    constexpr int NV = C * 18, NIT = (NV + 255) / 256;
    u32x4_t wv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        wv[it] = u32x4_t{0u, 0u, 0u, 0u};
        if (i < NV) wv[it] = *(const u32x4_t*)(p.w + (long long)i * 8);
    }
    float bv[(C + 255) / 256];
#pragma unroll
    for (int it = 0; it < (C + 255) / 256; ++it) bv[it] = (p.bias && tid + 256 * it < C) ? p.bias[tid + 256 * it] : 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it, r = i / 18, v = i - r * 18;
        if (i < NV) *(u32x4_t*)(wl + r * WP + v * 16) = wv[it];
    }
#pragma unroll
    for (int it = 0; it < (C + 255) / 256; ++it)
        if (tid + 256 * it < C) bl[tid + 256 * it] = bv[it];
    __syncthreads();

    f32x16_t acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    // MFMA row i of a 32-row block reads weight row brow(i): accumulator register r of lane (pixel, half) is then channel 32 nb + 16 half + r
    const int brow = 16 * ((lrow >> 2) & 1) + 4 * (lrow >> 3) + (lrow & 3);
    const char* const wb = wl + brow * WP + lhalf * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const u32x4_t af = *(const u32x4_t*)(wb + nb * 32 * WP + t * 32);
            mma16<T>(acc[nb], af, xq[t]);
        }

    const bool gn_on = p.gn_part[0] || p.gn_part[1] || p.gn_part[2];
    bf16_t* const orow = p.out + grow * p.ldo;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = nb * 32 + lhalf * 16;
        float y[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[8], f[8];
            const f32x4_t c0 = *(const f32x4_t*)(bl + col + 8 * h), c1 = *(const f32x4_t*)(bl + col + 8 * h + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[nb][8 * h + e] + (e < 4 ? c0[e] : c1[e - 4]);
            const u32x4_t wv = pack16<T>(v);
            *(u32x4_t*)(orow + col + 8 * h) = wv;
            if (p.dup_off) *(u32x4_t*)(orow + p.dup_off + col + 8 * h) = wv;
            unpack16<T>(wv, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[8 * h + e] = f[e];
        }
        if (gn_on) {
            float gx[32];
#pragma unroll
            for (int k = 0; k < 16; ++k) { gx[k] = y[k]; gx[16 + k] = y[k] * y[k]; }
            halfwave_reduce_scatter32(gx, lrow);          // lane lrow: total over the wave's 32 pixels of column lrow (sums) / lrow - 16 (squares)
            gcs[((lrow >> 4) * 4 + wave) * C + nb * 32 + lhalf * 16 + (lrow & 15)] = gx[0];
        }
    }
    if (gn_on) {
        __syncthreads();
        if (tid < 96) {
            const int c = tid >> 5, g = tid & 31;
            if (p.gn_part[c]) {
                const int cpg = p.gn_cpg[c], base = p.gn_coff[c];            // consumer channel of output column 0
                const int lo = max(0, g * cpg - base), hi = min(C, (g + 1) * cpg - base);
                double sa = 0.0, sq = 0.0;
                for (int k = lo; k < hi; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa += (double)gcs[r * C + k]; sq += (double)gcs[(4 + r) * C + k]; }
                double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + pb) * 32 + g) * 2;
                o[0] = sa;
                o[1] = sq;
            }
        }
    }
}

}  // namespace rf

extern "C" int rf_gn_silu_conv3x3_small(int dtype, const void* x, int B, int H, int W, int C, int ldx, int nchunks, const double* partial, const float* gamma,
                                        const float* beta, float eps, int silu, const void* w, const float* bias, int No, int out_dtype, void* out, int ldo,
                                        float* workspace, long long workspace_bytes, void* stream) {
    using namespace rf;
    RF_CHECK(x && partial && gamma && beta && w && out && workspace, "rf_gn_silu_conv3x3_small: null argument");
    RF_CHECK(B > 0 && H > 0 && W > 0 && nchunks > 0 && No >= 1 && No <= 4, "rf_gn_silu_conv3x3_small: bad sizes (1 <= No <= 4: 9 No rows fit SC_YP), got No=%d", No);
    RF_CHECK(C == 320 || C == 128 || C == 64, "rf_gn_silu_conv3x3_small: built for C = 320 (REFace), 128, 64 (reduced-width tests), got %d", C);
    RF_CHECK(ldx % 8 == 0 && ((uintptr_t)x | (uintptr_t)w) % 16 == 0 && (uintptr_t)workspace % 16 == 0, "rf_gn_silu_conv3x3_small: operands must be 16-byte aligned, ldx a multiple of 8");
    RF_CHECK(dtype == RF_BF16 || dtype == RF_F16, "rf_gn_silu_conv3x3_small: dtype %d (RF_BF16 or RF_F16)", dtype);
    RF_CHECK(out_dtype == RF_F32 || out_dtype == dtype, "rf_gn_silu_conv3x3_small: out_dtype %d (fp32 or the input's type)", out_dtype);
    const long long M = (long long)B * H * W;
    RF_CHECK(workspace_bytes >= M * SC_YP * 4, "rf_gn_silu_conv3x3_small: workspace of %lld bytes needed", M * SC_YP * 4);
    SmallConvParams p;
    p.x = (const bf16_t*)x; p.ldx = ldx; p.B = B; p.H = H; p.W = W; p.C = C; p.No = No; p.nchunks = nchunks; p.partial = partial;
    p.gamma = gamma; p.beta = beta; p.eps = eps; p.silu = silu; p.w = (const bf16_t*)w; p.y = workspace;
    hipStream_t st = (hipStream_t)stream;
    const int nb = B * ((H * W + 127) / 128);
#define RF_SC(C_) { const int smem = 36 * (C_ * 2 + 16) + 2 * C_ * 4;                                                                      \
        if (dtype == RF_F16) hipLaunchKernelGGL((gn_silu_taps_kernel<C_, f16_t>), dim3(nb), dim3(256), smem, st, p);                          \
        else hipLaunchKernelGGL((gn_silu_taps_kernel<C_, bf16_t>), dim3(nb), dim3(256), smem, st, p); }
    if (C == 320) RF_SC(320) else if (C == 128) RF_SC(128) else RF_SC(64)
#undef RF_SC
    const int gb = (int)((M + 255) / 256);
    if (out_dtype == RF_F32) hipLaunchKernelGGL(gather_taps_kernel<float>, dim3(gb), dim3(256), 0, st, workspace, B, H, W, No, bias, (float*)out, ldo);
    else if (out_dtype == RF_F16) hipLaunchKernelGGL(gather_taps_kernel<f16_t>, dim3(gb), dim3(256), 0, st, workspace, B, H, W, No, bias, (f16_t*)out, ldo);
    else hipLaunchKernelGGL(gather_taps_kernel<bf16_t>, dim3(gb), dim3(256), 0, st, workspace, B, H, W, No, bias, (bf16_t*)out, ldo);
    RF_LAUNCH_CHECK("rf_gn_silu_conv3x3_small");
    return 0;
}

extern "C" int rf_conv3x3_stem(const rf_stem_desc* d, void* stream) {
    using namespace rf;
    RF_CHECK(d && d->x && d->w && d->out, "rf_conv3x3_stem: null argument");
    RF_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && (d->H * d->W) % 128 == 0, "rf_conv3x3_stem: H*W = %d must be a multiple of the 128-pixel block", d->H * d->W);
    RF_CHECK(d->C == 320 || d->C == 128 || d->C == 64, "rf_conv3x3_stem: built for C = 320 (REFace), 128, 64 (reduced-width tests), got %d", d->C);
    const int dtype = d->dtype == 0 ? RF_BF16 : d->dtype;
    RF_CHECK(dtype == RF_BF16 || dtype == RF_F16, "rf_conv3x3_stem: dtype %d (RF_BF16 or RF_F16)", d->dtype);
    RF_CHECK(d->ldx >= 16 && d->ldx % 8 == 0 && d->ldo >= d->C && d->ldo % 8 == 0 && d->dup_off % 8 == 0, "rf_conv3x3_stem: ldx=%d (>= 16 stored channels) / ldo=%d / dup_off must be multiples of 8", d->ldx, d->ldo);
    RF_CHECK(((uintptr_t)d->x | (uintptr_t)d->w | (uintptr_t)d->out) % 16 == 0, "rf_conv3x3_stem: operands must be 16-byte aligned");
    StemParams p;
    p.x = (const bf16_t*)d->x; p.ldx = d->ldx; p.B = d->B; p.H = d->H; p.W = d->W; p.w = (const bf16_t*)d->w; p.bias = d->bias;
    p.out = (bf16_t*)d->out; p.ldo = d->ldo; p.dup_off = d->dup_off;
    double* const parts[3] = {d->gn_part0, d->gn_part1, d->gn_part2};
    const int cpg[3] = {d->gn_cpg0, d->gn_cpg1, d->gn_cpg2}, coff[3] = {d->gn_coff0, d->gn_coff1, d->gn_coff2};
    const int slot[3] = {d->gn_slot0, d->gn_slot1, d->gn_slot2}, nch[3] = {d->gn_nchunks0, d->gn_nchunks1, d->gn_nchunks2};
    for (int c = 0; c < 3; ++c) {
        RF_CHECK(!parts[c] || (cpg[c] > 0 && slot[c] >= 0 && slot[c] + d->H * d->W / 128 <= nch[c]),
                 "rf_conv3x3_stem: GroupNorm consumer %d: cpg=%d slot=%d needs %d slots of %d", c, cpg[c], slot[c], d->H * d->W / 128, nch[c]);
        p.gn_part[c] = parts[c]; p.gn_cpg[c] = cpg[c]; p.gn_coff[c] = coff[c]; p.gn_slot[c] = slot[c]; p.gn_nch[c] = nch[c];
    }
    const int nb = d->B * (d->H * d->W / 128);
#define RF_STEM_(NB_, T_) { constexpr int smem = 32 * NB_ * (144 * 2 + 16) + 32 * NB_ * 4 * 9; auto k = stem_conv3x3_kernel<NB_, T_>; \
        RF_RAISE_LDS(k, smem, "rf_conv3x3_stem"); \
        hipLaunchKernelGGL(k, dim3(nb), dim3(256), smem, (hipStream_t)stream, p); }
#define RF_STEM(NB_) { if (dtype == RF_F16) RF_STEM_(NB_, f16_t) else RF_STEM_(NB_, bf16_t) }
    if (d->C == 320) RF_STEM(10) else if (d->C == 128) RF_STEM(4) else RF_STEM(2)
#undef RF_STEM_
#undef RF_STEM
    RF_LAUNCH_CHECK("rf_conv3x3_stem");
    return 0;
}
