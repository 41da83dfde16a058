// Side kernels of the conditioning encoders (ArcFace IR-SE50, CLIP ViT-L/14): per-channel affine (+PReLU),
// squeeze-excite pooling / rescale, adaptive average pooling with crop, bilinear resize, CLIP token assembly,
// row L2-normalisation.  All HBM-bound and tiny next to the GEMMs.
#include "common.h"

namespace rf {

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stv(T* p, float v);
template <> __device__ __forceinline__ void stv<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stv<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// y[m, c] = prelu(x[m, c] * a[c] + b[c])   (eval-mode BatchNorm = affine; helpers.py:103-118)
template <typename TI, typename TO>
__global__ void channel_affine_kernel(const TI* __restrict__ x, int ldx, const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ slope, TO* __restrict__ y, int ldy, long long M, int Cc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Cc) return;
    const long long m = i / Cc;
    const int c = (int)(i - m * Cc);
    float v = ld<TI>(x + m * ldx + c) * a[c] + b[c];
    if (slope) v = v >= 0.f ? v : v * slope[c];
    stv<TO>(y + m * ldy + c, v);
}

// mean over HW of channels-last x [B, HW, C] -> fp32 [B, C]   (AdaptiveAvgPool2d(1), helpers.py:59)
template <typename T>
__global__ __launch_bounds__(256) void spatial_mean_kernel(const T* __restrict__ x, int HW, int Cc, int ldx, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;            // 4 pixel lanes
    float s = 0.f;
    if (c < Cc)
        for (int p = pl; p < HW; p += 4) s += ld<T>(x + ((long long)b * HW + p) * ldx + c);
    __shared__ float red[4][64];
    red[pl][threadIdx.x & 63] = s;
    __syncthreads();
    if (pl == 0 && c < Cc) out[(long long)b * Cc + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW;
}

// out[b, oy, ox, c] = r[b, oy, ox, c] * s[b, c] + sc[b, oy*st, ox*st, c]   (SE rescale + shortcut; MaxPool2d(1, st) = subsample)
template <typename T>
__global__ void se_scale_add_kernel(const T* __restrict__ r, const float* __restrict__ s, const T* __restrict__ sc, int ldsc, int Hs, int Ws,
                                    int st, T* __restrict__ out, int B, int Ho, int Wo, int Cc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Ho * Wo * Cc) return;
    const int c = (int)(i % Cc);
    const long long pix = i / Cc;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long long)Wo * Ho));
    const float shortcut = ld<T>(sc + (((long long)b * Hs + oy * st) * Ws + ox * st) * ldsc + c);
    stv<T>(out + i, ld<T>(r + i) * s[(long long)b * Cc + c] + shortcut);
}

// AdaptiveAvgPool2d over a crop window of an NCHW fp32 image, with a per-channel affine applied to the input:
// out = mean_{bin}( x * a[c] + b[c] ).  Bins follow PyTorch: [floor(i*in/out), ceil((i+1)*in/out)).
// out_nhwc = 0: NCHW fp32 [B, C, Ho, Wo]; 1: channels-last TO [B, Ho, Wo, Cpad] (pad channels zero).
template <typename TO>
__global__ void adaptive_pool_kernel(const float* __restrict__ x, int B, int Cc, int Hf, int Wf, int y0, int x0, int hc, int wc,
                                     const float* __restrict__ a, const float* __restrict__ bsh, int Ho, int Wo, int out_nhwc, int Cpad,
                                     TO* __restrict__ out) {
    const int CO = out_nhwc ? Cpad : Cc;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Ho * Wo * CO) return;
    int b, c, oy, ox;
    if (out_nhwc) {
        c = (int)(i % CO);
        const long long pix = i / CO;
        ox = (int)(pix % Wo); oy = (int)((pix / Wo) % Ho); b = (int)(pix / ((long long)Wo * Ho));
    } else {
        ox = (int)(i % Wo); oy = (int)((i / Wo) % Ho);
        c = (int)((i / ((long long)Wo * Ho)) % Cc); b = (int)(i / ((long long)Wo * Ho * Cc));
    }
    if (c >= Cc) { stv<TO>(out + i, 0.f); return; }
    const int ys = (oy * hc) / Ho, ye = ((oy + 1) * hc + Ho - 1) / Ho;
    const int xs = (ox * wc) / Wo, xe = ((ox + 1) * wc + Wo - 1) / Wo;
    const float sa = a ? a[c] : 1.f, sb = bsh ? bsh[c] : 0.f;
    float s = 0.f;
    for (int yy = ys; yy < ye; ++yy)
        for (int xx = xs; xx < xe; ++xx) s += x[(((long long)b * Cc + c) * Hf + y0 + yy) * Wf + x0 + xx] * sa + sb;
    stv<TO>(out + i, s / (float)((ye - ys) * (xe - xs)));
}

// bilinear resize (align_corners=False, no antialias) of NCHW fp32 with the input affine x*a[c]+b[c] applied first
// (ddpm.py:907-912: (tar+1)/2 -> CLIP normalise -> TF.resize).  out NCHW fp32.
__global__ void bilinear_resize_kernel(const float* __restrict__ x, int B, int Cc, int Hi, int Wi, const float* __restrict__ a,
                                       const float* __restrict__ bsh, int Ho, int Wo, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * Cc * Ho * Wo) return;
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
    const int c = (int)((i / ((long long)Wo * Ho)) % Cc);
    const long long bc = i / ((long long)Wo * Ho);
    const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
    float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float sa = a ? a[c] : 1.f, sb = bsh ? bsh[c] : 0.f;
    const float* p = x + bc * Hi * Wi;
    const float v00 = p[(long long)y0 * Wi + x0] * sa + sb, v01 = p[(long long)y0 * Wi + x1] * sa + sb;
    const float v10 = p[(long long)y1 * Wi + x0] * sa + sb, v11 = p[(long long)y1 * Wi + x1] * sa + sb;
    out[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
}

// CLIP tokens: out[b, 0, :] = cls + pos[0]; out[b, 1+p, :] = patch[b, p, :] + pos[1+p]   (HF CLIPVisionEmbeddings)
template <typename T>
__global__ void clip_tokens_kernel(const T* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                   T* __restrict__ out, int B, int NP, int Cc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * (NP + 1) * Cc) return;
    const int c = (int)(i % Cc);
    const int t = (int)((i / Cc) % (NP + 1));
    const int b = (int)(i / ((long long)Cc * (NP + 1)));
    const float v = (t == 0 ? cls[c] : ld<T>(patch + ((long long)b * NP + t - 1) * Cc + c)) + pos[(long long)t * Cc + c];
    stv<T>(out + i, v);
}

// y[r, :] = x[r, :] / ||x[r, :]||_2   (helpers.py:15-18), fp32, one wave per row
__global__ __launch_bounds__(64) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int cols) {
    const float* r = x + (long long)blockIdx.x * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 64) s += r[c] * r[c];
    s = sqrtf(wave_sum(s));
    for (int c = threadIdx.x; c < cols; c += 64) y[(long long)blockIdx.x * cols + c] = r[c] / s;
}

// ddpm.py:1038-1039: (a*wa + b*wb + c*wc) / den, evaluated left to right in fp32 (b / c optional)
__global__ void combine3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c, float wa, float wb,
                                float wc, float den, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = a[i] * wa;
    if (b) v = v + b[i] * wb;
    if (c) v = v + c[i] * wc;
    out[i] = den != 0.f ? v / den : v;
}

static inline dim3 g1(long long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace rf

using namespace rf;

extern "C" int rf_channel_affine(int dtype, const void* x, int ldx, const float* a, const float* b, const float* slope, int out_dtype,
                                 void* y, int ldy, int64_t M, int C, void* stream) {
    RF_CHECK(x && a && b && y && M > 0 && C > 0, "rf_channel_affine: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)M * C;
#define CA(TI, TO) hipLaunchKernelGGL((channel_affine_kernel<TI, TO>), g1(n), dim3(256), 0, st, (const TI*)x, ldx, a, b, slope, (TO*)y, ldy, (long long)M, C)
    if (dtype == RF_F32 && out_dtype == RF_F32) CA(float, float);
    else if (dtype == RF_F32 && out_dtype == RF_BF16) CA(float, bf16_t);
    else if (dtype == RF_BF16 && out_dtype == RF_F32) CA(bf16_t, float);
    else if (dtype == RF_BF16 && out_dtype == RF_BF16) CA(bf16_t, bf16_t);
    else RF_CHECK(false, "rf_channel_affine: bad dtypes");
#undef CA
    RF_LAUNCH_CHECK("rf_channel_affine");
    return 0;
}

extern "C" int rf_spatial_mean(int dtype, const void* x, int B, int HW, int C, int ldx, float* out, void* stream) {
    RF_CHECK(x && out && B > 0 && HW > 0 && C > 0, "rf_spatial_mean: bad arguments");
    dim3 grid((C + 63) / 64, B);
    if (dtype == RF_F32) hipLaunchKernelGGL(spatial_mean_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, HW, C, ldx, out);
    else if (dtype == RF_BF16) hipLaunchKernelGGL(spatial_mean_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, HW, C, ldx, out);
    else RF_CHECK(false, "rf_spatial_mean: bad dtype");
    RF_LAUNCH_CHECK("rf_spatial_mean");
    return 0;
}

extern "C" int rf_se_scale_add(int dtype, const void* r, const float* s, const void* sc, int ldsc, int Hs, int Ws, int stride, void* out,
                               int B, int Ho, int Wo, int C, void* stream) {
    RF_CHECK(r && s && sc && out && B > 0 && Ho > 0 && Wo > 0 && C > 0 && stride >= 1, "rf_se_scale_add: bad arguments");
    RF_CHECK((Ho - 1) * stride < Hs && (Wo - 1) * stride < Ws, "rf_se_scale_add: shortcut too small");
    const long long n = (long long)B * Ho * Wo * C;
    if (dtype == RF_F32) hipLaunchKernelGGL(se_scale_add_kernel<float>, g1(n), dim3(256), 0, (hipStream_t)stream, (const float*)r, s, (const float*)sc, ldsc, Hs, Ws, stride, (float*)out, B, Ho, Wo, C);
    else if (dtype == RF_BF16) hipLaunchKernelGGL(se_scale_add_kernel<bf16_t>, g1(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)r, s, (const bf16_t*)sc, ldsc, Hs, Ws, stride, (bf16_t*)out, B, Ho, Wo, C);
    else RF_CHECK(false, "rf_se_scale_add: bad dtype");
    RF_LAUNCH_CHECK("rf_se_scale_add");
    return 0;
}

extern "C" int rf_adaptive_avgpool(const float* x, int B, int C, int Hf, int Wf, int y0, int x0, int hc, int wc, const float* a, const float* b,
                                   int Ho, int Wo, int out_nhwc, int out_dtype, int Cpad, void* out, void* stream) {
    RF_CHECK(x && out && B > 0 && C > 0 && y0 >= 0 && x0 >= 0 && y0 + hc <= Hf && x0 + wc <= Wf && Ho > 0 && Wo > 0, "rf_adaptive_avgpool: bad arguments");
    RF_CHECK(out_nhwc ? Cpad >= C : out_dtype == RF_F32, "rf_adaptive_avgpool: NCHW output is fp32; NHWC needs Cpad >= C");
    const long long n = (long long)B * Ho * Wo * (out_nhwc ? Cpad : C);
    if (out_dtype == RF_F32) hipLaunchKernelGGL(adaptive_pool_kernel<float>, g1(n), dim3(256), 0, (hipStream_t)stream, x, B, C, Hf, Wf, y0, x0, hc, wc, a, b, Ho, Wo, out_nhwc, Cpad, (float*)out);
    else if (out_dtype == RF_BF16) hipLaunchKernelGGL(adaptive_pool_kernel<bf16_t>, g1(n), dim3(256), 0, (hipStream_t)stream, x, B, C, Hf, Wf, y0, x0, hc, wc, a, b, Ho, Wo, out_nhwc, Cpad, (bf16_t*)out);
    else RF_CHECK(false, "rf_adaptive_avgpool: bad out_dtype");
    RF_LAUNCH_CHECK("rf_adaptive_avgpool");
    return 0;
}

extern "C" int rf_bilinear_resize(const float* x, int B, int C, int Hi, int Wi, const float* a, const float* b, int Ho, int Wo, float* out, void* stream) {
    RF_CHECK(x && out && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "rf_bilinear_resize: bad arguments");
    hipLaunchKernelGGL(bilinear_resize_kernel, g1((long long)B * C * Ho * Wo), dim3(256), 0, (hipStream_t)stream, x, B, C, Hi, Wi, a, b, Ho, Wo, out);
    RF_LAUNCH_CHECK("rf_bilinear_resize");
    return 0;
}

extern "C" int rf_clip_tokens(int dtype, const void* patch, const float* cls, const float* pos, void* out, int B, int NP, int C, void* stream) {
    RF_CHECK(patch && cls && pos && out && B > 0 && NP > 0 && C > 0, "rf_clip_tokens: bad arguments");
    const long long n = (long long)B * (NP + 1) * C;
    if (dtype == RF_F32) hipLaunchKernelGGL(clip_tokens_kernel<float>, g1(n), dim3(256), 0, (hipStream_t)stream, (const float*)patch, cls, pos, (float*)out, B, NP, C);
    else if (dtype == RF_BF16) hipLaunchKernelGGL(clip_tokens_kernel<bf16_t>, g1(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)patch, cls, pos, (bf16_t*)out, B, NP, C);
    else RF_CHECK(false, "rf_clip_tokens: bad dtype");
    RF_LAUNCH_CHECK("rf_clip_tokens");
    return 0;
}

extern "C" int rf_l2norm_rows(const float* x, float* y, int rows, int cols, void* stream) {
    RF_CHECK(x && y && rows > 0 && cols > 0, "rf_l2norm_rows: bad arguments");
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, y, cols);
    RF_LAUNCH_CHECK("rf_l2norm_rows");
    return 0;
}

extern "C" int rf_combine3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float den, float* out, int64_t n, void* stream) {
    RF_CHECK(a && out && n > 0, "rf_combine3: bad arguments");
    hipLaunchKernelGGL(combine3_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, a, b, c, wa, wb, wc, den, out, (long long)n);
    RF_LAUNCH_CHECK("rf_combine3");
    return 0;
}
