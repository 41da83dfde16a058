// Small HBM-bound glue kernels: DDIM step arithmetic, layout changes, casts, timestep embedding.
#include "common.h"

namespace rf {

template <typename TO> __device__ __forceinline__ void st(TO* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void st<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }

// x_in[(dup*B), hw, Cpad] = [img(4) | z_inpaint(4) | mask(1) | 0...]   (ddim.py:330, 338)
template <typename TO>
__global__ void ddim_pack_kernel(const float* __restrict__ img, const float* __restrict__ z, const float* __restrict__ mask,
                                 int B, int hw, int dup, TO* __restrict__ out, int Cpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*hw pixels
    if (i >= (long long)B * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long long)b * hw);
    float v[9];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v[c] = img[((long long)b * 4 + c) * hw + p];
        v[4 + c] = z[((long long)b * 4 + c) * hw + p];
    }
    v[8] = mask[(long long)b * hw + p];
    for (int r = 0; r < dup; ++r) {
        TO* o = out + ((long long)(r * B + b) * hw + p) * Cpad;
        for (int c = 0; c < Cpad; ++c) st<TO>(o + c, c < 9 ? v[c] : 0.f);
    }
}

// ddim.py:346, 364-374 on NCHW fp32 latents; eps is channels-last [(2B|B), hw, ld]
// coefs (device, fp32): {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev-sigma^2), sigma}
__global__ void ddim_update_kernel(const float* __restrict__ eps, int ld, int cfg, float scale, float* __restrict__ img,
                                   float* __restrict__ pred_x0, const float* __restrict__ noise, int B, int hw,
                                   const float* __restrict__ coefs) {
    const float sqrt_at = coefs[0], sqrt_1m_at = coefs[1], sqrt_aprev = coefs[2], dir_coef = coefs[3], sigma = coefs[4];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*4*hw
    if (i >= (long long)B * 4 * hw) return;
    const int p = (int)(i % hw);
    const int c = (int)((i / hw) % 4);
    const int b = (int)(i / ((long long)4 * hw));
    float e;
    if (cfg) {
        const float eu = eps[((long long)b * hw + p) * ld + c];
        const float ec = eps[((long long)(B + b) * hw + p) * ld + c];
        e = eu + scale * (ec - eu);
    } else {
        e = eps[((long long)b * hw + p) * ld + c];
    }
    const float x = img[i];
    const float px0 = (x - sqrt_1m_at * e) / sqrt_at;
    float xp = sqrt_aprev * px0 + dir_coef * e;
    if (noise) xp += sigma * noise[i];
    img[i] = xp;
    if (pred_x0) pred_x0[i] = px0;
}

template <typename TO>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int B, int C, int HW, TO* __restrict__ out, int Cpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*HW*Cpad (output order)
    if (i >= (long long)B * HW * Cpad) return;
    const int c = (int)(i % Cpad);
    const long long bp = i / Cpad;
    const int p = (int)(bp % HW), b = (int)(bp / HW);
    st<TO>(out + i, c < C ? x[((long long)b * C + c) * HW + p] : 0.f);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int B, int C, int HW, int ldx, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*HW (output order)
    if (i >= (long long)B * C * HW) return;
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((long long)C * HW));
    out[i] = elem<T>::load(x + ((long long)b * HW + p) * ldx + c);
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st<TO>(y + i, elem<TI>::load(x + i));
}

// fp32 [M, C] (row pitch ldx) -> split-bf16 pairs [M][C hi | C lo] (row pitch ldo): the RF_BF16X3 operand form of a tensor that no
// normalisation pass rewrites on its way into a convolution (residual stream -> Upsample conv / nin_shortcut, model.py:53-57,117-121)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, long long M, int C, int ldx, bf16_t* __restrict__ out, int ldo) {
    const int vpr = C >> 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * vpr) return;
    const long long r = i / vpr;
    const int c = (int)(i - r * vpr) * 4;
    const f32x4_t v = *(const f32x4_t*)(x + r * ldx + c);
    const float f[4] = {v[0], v[1], v[2], v[3]};
    u32x2_t h, l;
    split4_bf16(f, h, l);
    *(u32x2_t*)(out + r * ldo + c) = h;
    *(u32x2_t*)(out + r * ldo + C + c) = l;
}

// util.py:151-166; `freqs` [dim/2] = exp(-ln(max_period) * k / half) is a host-built fp32 table
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, const float* __restrict__ freqs, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= n * half) return;
    const int r = i / half, k = i - r * half;
    const float a = t[r] * freqs[k];
    out[(long long)r * dim + k] = cosf(a);
    out[(long long)r * dim + half + k] = sinf(a);
    if ((dim & 1) && k == 0) out[(long long)r * dim + dim - 1] = 0.f;
}

__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = silu_exact(x[i]);
}

// distributions.py:24-37 + ddpm.py:857: out = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps), NCHW
__global__ void gaussian_sample_kernel(const float* __restrict__ moments, const float* __restrict__ eps, float scale,
                                       float* __restrict__ out, int B, int Cc, int HW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*HW
    if (i >= (long long)B * Cc * HW) return;
    const int p = (int)(i % HW);
    const int c = (int)((i / HW) % Cc);
    const int b = (int)(i / ((long long)Cc * HW));
    const float mean = moments[((long long)b * 2 * Cc + c) * HW + p];
    float lv = moments[((long long)b * 2 * Cc + Cc + c) * HW + p];
    lv = fminf(fmaxf(lv, -30.0f), 20.0f);
    const float x = mean + expf(0.5f * lv) * (eps ? eps[i] : 0.0f);
    out[i] = scale * x;
}

// inference_test_bench.py:494: clamp((x + 1) / 2, 0, 1)
__global__ void to_image_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fminf(fmaxf((x[i] + 1.0f) / 2.0f, 0.0f), 1.0f);
}

static inline dim3 grid1d(long long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }


// ---- input preparation on the device (test_bench_dataset.py:283-355 does this per image on the host with PIL / torch-CPU ops)
// uint8 HWC image -> fp32 NCHW (x / 255 - mean[c]) / std[c]   (torchvision ToTensor + Normalize, same operation order)
__global__ void u8_to_norm_kernel(const uint8_t* __restrict__ x, int B, int HW, const float* __restrict__ mean,
                                  const float* __restrict__ stdv, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over B*HW pixels
    if (i >= (long long)B * HW) return;
    const int b = (int)(i / HW), p = (int)(i - (long long)b * HW);
    const uint8_t* px = x + i * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((long long)b * 3 + c) * HW + p] = ((float)px[c] / 255.0f - mean[c]) / stdv[c];
}
// label map -> {0, 1} mask: out = invert ? 1 - lut[label] : lut[label]   (np.isin(label, keep) -> 255 -> ToTensor; `1 -` for targets)
__global__ void label_mask_kernel(const uint8_t* __restrict__ lab, long long n, const uint8_t* __restrict__ lut, int invert,
                                  float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float m = lut[lab[i]] ? 1.0f : 0.0f;
    out[i] = invert ? 1.0f - m : m;
}
// out[b, c, p] = x[b, c, p] * mask[b, 0, p]
__global__ void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask, int B, int C, int HW, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * C * HW) return;
    const int p = (int)(i % HW);
    const int b = (int)(i / ((long long)C * HW));
    out[i] = x[i] * mask[(long long)b * HW + p];
}

// cv2.resize(img, (Wo, Ho), interpolation=cv2.INTER_LINEAR) of uint8 HWC images (A.Resize(224, 224) on the source face,
// ldm/data/test_bench_dataset.py:141-148, 324), in OpenCV's integer arithmetic and operation order (cv2 itself is absent here: last bit unpinned) (resize.cpp: HResizeLinear<uchar, int, short,
// 2048> + the u8 VResizeLinear): two taps per axis at half-pixel centres, 11-bit weights rounded to nearest-even, columns clamped
// with their weight ((s, f) = (0, 0) left of the image, (W - 1, 0) from the last column on), rows clipped with the weights kept;
//   out = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2,   Hk = S[yk][sx] * a0 + S[yk][sx + 1] * a1.
// An exact 2:1 reduction on both axes is OpenCV's fast-area case: (a + b + c + d + 2) >> 2.  One thread per output pixel.
__device__ __forceinline__ void cv_linear_tap(int d, double scale, int n_src, bool clamp, int& s, int& w0, int& w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (clamp) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    }
    w1 = __float2int_rn(f * 2048.0f);
    w0 = __float2int_rn((1.0f - f) * 2048.0f);
}
__global__ void resize_u8_linear_kernel(const uint8_t* __restrict__ x, int B, int H, int W, int C, long long sb, int Ho, int Wo,
                                        uint8_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // over B * Ho * Wo
    if (i >= (long long)B * Ho * Wo) return;
    const int dx = (int)(i % Wo), dy = (int)((i / Wo) % Ho), b = (int)(i / ((long long)Wo * Ho));
    const uint8_t* img = x + (long long)b * sb;
    uint8_t* o = out + i * C;
    if (H == 2 * Ho && W == 2 * Wo) {
        const uint8_t* p0 = img + ((long long)(2 * dy) * W + 2 * dx) * C;
        const uint8_t* p1 = p0 + (long long)W * C;
        for (int c = 0; c < C; ++c) o[c] = (uint8_t)(((int)p0[c] + (int)p0[C + c] + (int)p1[c] + (int)p1[C + c] + 2) >> 2);
        return;
    }
    int sx, a0, a1, sy, b0, b1;
    // (OpenCV's order: inv_scale = dsize / ssize, scale = 1. / inv_scale -- two roundings in double, as reface_amd/data.py:_linear_taps)
    cv_linear_tap(dx, 1.0 / ((double)Wo / (double)W), W, true, sx, a0, a1);
    cv_linear_tap(dy, 1.0 / ((double)Ho / (double)H), H, false, sy, b0, b1);
    const int sx1 = min(sx + 1, W - 1), y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
    const uint8_t* r0 = img + (long long)y0 * W * C;
    const uint8_t* r1 = img + (long long)y1 * W * C;
    for (int c = 0; c < C; ++c) {
        const int h0 = ((int)r0[sx * C + c] * a0 + (int)r0[sx1 * C + c] * a1) >> 4;
        const int h1 = ((int)r1[sx * C + c] * a0 + (int)r1[sx1 * C + c] * a1) >> 4;
        o[c] = (uint8_t)((((b0 * h0) >> 16) + ((b1 * h1) >> 16) + 2) >> 2);
    }
}

// The output files of the test-bench CLI as uint8 HWC panels, composed on the device (scripts/inference_test_bench.py:500-553 does it per
// image on the host with torch-CPU / numpy float ops): per image ONE packed record
//   [result | mask | GT | inpaint | ref]  5 x [H][W][3]      then      grid [H + 4][4 W + 10][3]  (make_grid: 4 panels, padding 2, pad value 0)
// with every float -> uint8 conversion the reference's `(255. * x).astype(np.uint8)`: fp32 product, truncation toward zero, low byte
// (out-of-range values of the un-clamped reference panel wrap modulo 256 exactly as numpy's cast does on x86-64).  Panels:
//   result  255 * r                                  (r = clamp((x + 1) / 2, 0, 1) already, rf_to_image)
//   GT / inpaint / mask   255 * ((v + 1) / 2)        (mask: the one channel replicated, cv2.COLOR_GRAY2RGB)
//   ref     255 * (v * clip_std[c] + clip_mean[c])   (the CLIP-normalised reference resized to H x W beforehand)
// The grid's pad bytes are never written (the buffer is zero-initialised once by the host).  One thread per pixel.
__device__ __forceinline__ uint8_t u8_trunc(float v) { return (uint8_t)((int)v & 0xff); }
__global__ void compose_outputs_u8_kernel(const float* __restrict__ res, const float* __restrict__ tgt, const float* __restrict__ inp,
                                          const float* __restrict__ msk, const float* __restrict__ ref, int B, int H, int W, int with_grid,
                                          uint8_t* __restrict__ out, long long rec_bytes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long HW = (long long)H * W;
    if (i >= B * HW) return;
    const int b = (int)(i / HW);
    const long long p = i - b * HW;
    const int y = (int)(p / W), x = (int)(p - (long long)y * W);
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f}, mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    uint8_t* rec = out + b * rec_bytes;
    const long long panel = HW * 3;
    const int GW = 4 * W + 10;
    uint8_t* grid = rec + 5 * panel;
    const float m = (msk[b * HW + p] + 1.0f) / 2.0f;
    const uint8_t mu8 = u8_trunc(255.0f * m);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long long src = ((long long)b * 3 + c) * HW + p;
        const float r = res[src], g = (tgt[src] + 1.0f) / 2.0f, n = (inp[src] + 1.0f) / 2.0f;
        const float rf = ref[src] * stdv[c] + mean[c];
        const uint8_t r8 = u8_trunc(255.0f * r), g8 = u8_trunc(255.0f * g), n8 = u8_trunc(255.0f * n), f8 = u8_trunc(255.0f * rf);
        const long long o = p * 3 + c;
        rec[o] = r8;
        rec[panel + o] = mu8;
        rec[2 * panel + o] = g8;
        rec[3 * panel + o] = n8;
        rec[4 * panel + o] = f8;
        if (with_grid) {          // make_grid([GT, inpaint, ref, result]): panel k at columns 2 + k (W + 2), rows 2 ..
            uint8_t* gp = grid + ((long long)(y + 2) * GW + (x + 2)) * 3 + c;
            gp[0] = g8;
            gp[(long long)(W + 2) * 3] = n8;
            gp[(long long)(W + 2) * 6] = f8;
            gp[(long long)(W + 2) * 9] = r8;
        }
    }
}

}  // namespace rf

using namespace rf;

extern "C" int rf_ddim_pack_input(const float* img, const float* z_inpaint, const float* mask, int B, int hw, int dup,
                                  int out_dtype, void* x_in, int Cpad, void* stream) {
    RF_CHECK(img && z_inpaint && mask && x_in && B > 0 && hw > 0 && (dup == 1 || dup == 2) && Cpad >= 9, "rf_ddim_pack_input: bad arguments");
    hipStream_t st_ = (hipStream_t)stream;
    if (out_dtype == RF_F32) hipLaunchKernelGGL(ddim_pack_kernel<float>, grid1d((long long)B * hw), dim3(256), 0, st_, img, z_inpaint, mask, B, hw, dup, (float*)x_in, Cpad);
    else if (out_dtype == RF_BF16) hipLaunchKernelGGL(ddim_pack_kernel<bf16_t>, grid1d((long long)B * hw), dim3(256), 0, st_, img, z_inpaint, mask, B, hw, dup, (bf16_t*)x_in, Cpad);
    else if (out_dtype == RF_F16) hipLaunchKernelGGL(ddim_pack_kernel<f16_t>, grid1d((long long)B * hw), dim3(256), 0, st_, img, z_inpaint, mask, B, hw, dup, (f16_t*)x_in, Cpad);
    else RF_CHECK(false, "rf_ddim_pack_input: bad out_dtype %d", out_dtype);
    RF_LAUNCH_CHECK("rf_ddim_pack_input");
    return 0;
}

extern "C" int rf_ddim_update(const float* eps, int ld_eps, int cfg, float scale, float* img, float* pred_x0, const float* noise,
                              int B, int hw, const float* coefs, void* stream) {
    RF_CHECK(eps && img && coefs && B > 0 && hw > 0 && ld_eps >= 4, "rf_ddim_update: bad arguments");
    hipLaunchKernelGGL(ddim_update_kernel, grid1d((long long)B * 4 * hw), dim3(256), 0, (hipStream_t)stream, eps, ld_eps, cfg, scale, img,
                       pred_x0, noise, B, hw, coefs);
    RF_LAUNCH_CHECK("rf_ddim_update");
    return 0;
}

extern "C" int rf_nchw_to_nhwc(const float* x, int B, int C, int HW, int out_dtype, void* out, int Cpad, void* stream) {
    RF_CHECK(x && out && B > 0 && C > 0 && HW > 0 && Cpad >= C, "rf_nchw_to_nhwc: bad arguments");
    const long long n = (long long)B * HW * Cpad;
    if (out_dtype == RF_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, B, C, HW, (float*)out, Cpad);
    else if (out_dtype == RF_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, B, C, HW, (bf16_t*)out, Cpad);
    else if (out_dtype == RF_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, B, C, HW, (f16_t*)out, Cpad);
    else RF_CHECK(false, "rf_nchw_to_nhwc: bad out_dtype %d", out_dtype);
    RF_LAUNCH_CHECK("rf_nchw_to_nhwc");
    return 0;
}

extern "C" int rf_nhwc_to_nchw(int dtype, const void* x, int B, int C, int HW, int ldx, float* out, void* stream) {
    RF_CHECK(x && out && B > 0 && C > 0 && HW > 0 && ldx >= C, "rf_nhwc_to_nchw: bad arguments");
    const long long n = (long long)B * C * HW;
    if (dtype == RF_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const float*)x, B, C, HW, ldx, out);
    else if (dtype == RF_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, B, C, HW, ldx, out);
    else if (dtype == RF_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const f16_t*)x, B, C, HW, ldx, out);
    else RF_CHECK(false, "rf_nhwc_to_nchw: bad dtype %d", dtype);
    RF_LAUNCH_CHECK("rf_nhwc_to_nchw");
    return 0;
}

extern "C" int rf_cast(int in_dtype, const void* x, int out_dtype, void* out, int64_t n, void* stream) {
    RF_CHECK(x && out && n > 0, "rf_cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == RF_F32 && out_dtype == RF_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid1d(n), dim3(256), 0, s, (const float*)x, (bf16_t*)out, (long long)n);
    else if (in_dtype == RF_BF16 && out_dtype == RF_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid1d(n), dim3(256), 0, s, (const bf16_t*)x, (float*)out, (long long)n);
    else if (in_dtype == RF_F32 && out_dtype == RF_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid1d(n), dim3(256), 0, s, (const float*)x, (float*)out, (long long)n);
    else if (in_dtype == RF_BF16 && out_dtype == RF_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid1d(n), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, (long long)n);
    else if (in_dtype == RF_F32 && out_dtype == RF_F16) hipLaunchKernelGGL((cast_kernel<float, f16_t>), grid1d(n), dim3(256), 0, s, (const float*)x, (f16_t*)out, (long long)n);
    else if (in_dtype == RF_F16 && out_dtype == RF_F32) hipLaunchKernelGGL((cast_kernel<f16_t, float>), grid1d(n), dim3(256), 0, s, (const f16_t*)x, (float*)out, (long long)n);
    else RF_CHECK(false, "rf_cast: bad dtypes %d -> %d", in_dtype, out_dtype);
    RF_LAUNCH_CHECK("rf_cast");
    return 0;
}

extern "C" int rf_split_bf16(const float* x, int64_t M, int C, int ldx, void* out, int ldo, void* stream) {
    RF_CHECK(x && out && M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= 2 * C, "rf_split_bf16: bad arguments M=%lld C=%d ldx=%d ldo=%d",
             (long long)M, C, ldx, ldo);
    RF_CHECK((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "rf_split_bf16: operands must be 16-byte aligned");
    hipLaunchKernelGGL(split_bf16_kernel, grid1d((long long)M * (C / 4)), dim3(256), 0, (hipStream_t)stream, x, (long long)M, C, ldx, (bf16_t*)out, ldo);
    RF_LAUNCH_CHECK("rf_split_bf16");
    return 0;
}

extern "C" int rf_timestep_embedding(const float* t, int n, int dim, const float* freqs, float* out, void* stream) {
    RF_CHECK(t && out && freqs && n > 0 && dim >= 2, "rf_timestep_embedding: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, grid1d((long long)n * (dim / 2)), dim3(256), 0, (hipStream_t)stream, t, n, dim, freqs, out);
    RF_LAUNCH_CHECK("rf_timestep_embedding");
    return 0;
}

extern "C" int rf_silu_f32(const float* x, float* y, int64_t n, void* stream) {
    RF_CHECK(x && y && n > 0, "rf_silu_f32: bad arguments");
    hipLaunchKernelGGL(silu_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n);
    RF_LAUNCH_CHECK("rf_silu_f32");
    return 0;
}

extern "C" int rf_gaussian_sample(const float* moments, const float* eps, float scale, float* out, int B, int C, int HW, void* stream) {
    RF_CHECK(moments && out && B > 0 && C > 0 && HW > 0, "rf_gaussian_sample: bad arguments");
    hipLaunchKernelGGL(gaussian_sample_kernel, grid1d((long long)B * C * HW), dim3(256), 0, (hipStream_t)stream, moments, eps, scale, out, B, C, HW);
    RF_LAUNCH_CHECK("rf_gaussian_sample");
    return 0;
}

extern "C" int rf_to_image(const float* x, float* y, int64_t n, void* stream) {
    RF_CHECK(x && y && n > 0, "rf_to_image: bad arguments");
    hipLaunchKernelGGL(to_image_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n);
    RF_LAUNCH_CHECK("rf_to_image");
    return 0;
}

extern "C" int rf_u8_to_norm(const void* x, int B, int HW, const float* mean, const float* stdv, float* out, void* stream) {
    using namespace rf;
    RF_CHECK(x && mean && stdv && out && B > 0 && HW > 0, "rf_u8_to_norm: bad arguments");
    const long long n = (long long)B * HW;
    hipLaunchKernelGGL(u8_to_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)x, B, HW, mean, stdv, out);
    RF_LAUNCH_CHECK("rf_u8_to_norm");
    return 0;
}

extern "C" int rf_label_mask(const void* labels, int64_t n, const void* lut256, int invert, float* out, void* stream) {
    using namespace rf;
    RF_CHECK(labels && lut256 && out && n > 0, "rf_label_mask: bad arguments");
    hipLaunchKernelGGL(label_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)labels, (long long)n,
                       (const uint8_t*)lut256, invert, out);
    RF_LAUNCH_CHECK("rf_label_mask");
    return 0;
}

extern "C" int rf_mul_mask(const float* x, const float* mask, int B, int C, int HW, float* out, void* stream) {
    using namespace rf;
    RF_CHECK(x && mask && out && B > 0 && C > 0 && HW > 0, "rf_mul_mask: bad arguments");
    const long long n = (long long)B * C * HW;
    hipLaunchKernelGGL(mul_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, mask, B, C, HW, out);
    RF_LAUNCH_CHECK("rf_mul_mask");
    return 0;
}

extern "C" int rf_resize_u8_linear(const void* x, int B, int H, int W, int C, int64_t image_stride, int Ho, int Wo, void* out, void* stream) {
    using namespace rf;
    RF_CHECK(x && out && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4 && Ho > 0 && Wo > 0 && image_stride >= (int64_t)H * W * C,
             "rf_resize_u8_linear: bad arguments (B=%d H=%d W=%d C=%d Ho=%d Wo=%d)", B, H, W, C, Ho, Wo);
    const long long n = (long long)B * Ho * Wo;
    hipLaunchKernelGGL(resize_u8_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)x, B, H, W, C,
                       (long long)image_stride, Ho, Wo, (uint8_t*)out);
    RF_LAUNCH_CHECK("rf_resize_u8_linear");
    return 0;
}

extern "C" int rf_compose_outputs_u8(const float* result01, const float* target, const float* inpaint, const float* mask, const float* ref,
                                     int B, int H, int W, int with_grid, void* out_u8, int64_t record_bytes, void* stream) {
    using namespace rf;
    const long long need = 5LL * H * W * 3 + (with_grid ? (long long)(H + 4) * (4 * W + 10) * 3 : 0);
    RF_CHECK(result01 && target && inpaint && mask && ref && out_u8 && B > 0 && H > 0 && W > 0 && record_bytes >= need,
             "rf_compose_outputs_u8: bad arguments (B=%d H=%d W=%d record %lld < %lld)", B, H, W, (long long)record_bytes, need);
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(compose_outputs_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, result01, target, inpaint, mask, ref,
                       B, H, W, with_grid, (uint8_t*)out_u8, (long long)record_bytes);
    RF_LAUNCH_CHECK("rf_compose_outputs_u8");
    return 0;
}
