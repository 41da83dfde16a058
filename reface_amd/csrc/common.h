// Shared device/host helpers for the gfx950 REFace kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/reface_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

typedef uint16_t bf16_t;   // storage type of bf16 values
typedef _Float16 f16_t;    // storage type of fp16 (IEEE binary16) values -- a type of its own, so that every kernel template instantiates per 16-bit format
                           // (RF_F16, the "fp16" throughput mode: same MFMA rate as bf16 -- v_mfma_f32_32x32x16_f16 -- with 11 significant bits instead of 8)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef uint8_t fp8_t;     // storage type of OCP e4m3fn values (fp8 activations / weights of the MX-scaled MFMA path)

namespace rf {

void set_error(const char* fmt, ...);

// Tuning / experiment switches (tile overrides, epilogue forms, attention variants, the RF_GEMM_DBG timing decompositions) exist only in
// builds made with -DRF_EXPERIMENT (tools/build_variant.sh, loaded through REFACE_HIP_LIB for same-box A/B runs).  The release library
// reads NO environment variable: a stray variable in a user's shell cannot change -- or, for the timing decompositions, corrupt -- results.
#ifdef RF_EXPERIMENT
static inline int tune_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define RF_DBG(p, bits) ((p).dbg & (bits))
#else
static inline int tune_env(const char*, int dflt) { return dflt; }
#define RF_DBG(p, bits) 0
#endif

#define RF_CHECK(cond, ...)                                     \
    do {                                                        \
        if (!(cond)) {                                          \
            rf::set_error(__VA_ARGS__);                         \
            return 1;                                           \
        }                                                       \
    } while (0)

#define RF_LAUNCH_CHECK(name)                                               \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            rf::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                       \
        }                                                                   \
    } while (0)

// The dynamic-LDS limit of a kernel above the 64 KB default is a PER-DEVICE function attribute: raised once per (call site = kernel instantiation,
// device) -- the call site keeps its own bit mask of the devices it has served -- and the status is checked instead of surfacing later as a
// failed launch.
#define RF_RAISE_LDS(kernel, bytes, name)                                                                                                   \
    do {                                                                                                                                    \
        static unsigned long long done_ = 0ull;                                                                                             \
        int dev_ = 0;                                                                                                                       \
        (void)hipGetDevice(&dev_);                                                                                                          \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                                                \
        if (!(done_ & bit_)) {                                                                                                              \
            const hipError_t e_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));          \
            if (e_ != hipSuccess) {                                                                                                         \
                rf::set_error("%s: cannot raise the dynamic LDS limit to %d bytes on device %d: %s", name, (int)(bytes), dev_, hipGetErrorString(e_)); \
                return 2;                                                                                                                   \
            }                                                                                                                               \
            done_ |= bit_;                                                                                                                  \
        }                                                                                                                                   \
    } while (0)

// Bit casts BY VALUE.  hipcc (ROCm 7.2) miscompiles __builtin_bit_cast applied directly to an element
// lvalue of an ext_vector_type (v[q] in an unrolled loop collapses to v[0]); copying the element into a
// by-value parameter first is safe.
__device__ __forceinline__ float as_f32(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t as_u32(float x) { return __builtin_bit_cast(uint32_t, x); }

// fp32 -> bf16, round-to-nearest-even (same rule as torch .to(bfloat16)): native conversions, which hipcc lowers to
// v_cvt_pk_bf16_f32 on gfx950 (one instruction per pair instead of ~5 integer ops per value).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bf2f(bf16_t h) { return as_f32(((uint32_t)h) << 16); }

// fp32 -> fp16, round-to-nearest-even (torch .to(float16)): v_cvt_pk_f16_f32 on gfx950; fp16 -> fp32: v_cvt_f32_f16 (low half) / its SDWA form (high half)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const f16x2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
// The two 16-bit storage formats behind one set of names: T = bf16_t or f16_t.
//   pack2<T>(lo, hi): two fp32 values -> one packed dword (round-to-nearest-even);  lo16<T> / hi16<T>: the halves of a packed dword as fp32;
//   round16<T>(v): the value as it is stored;  one16<T>(): the bit pattern of 1.0
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_h2(lo, hi); }
template <typename T> __device__ __forceinline__ float lo16(uint32_t w);
template <typename T> __device__ __forceinline__ float hi16(uint32_t w);
template <> __device__ __forceinline__ float lo16<bf16_t>(uint32_t w) { return as_f32(w << 16); }
template <> __device__ __forceinline__ float hi16<bf16_t>(uint32_t w) { return as_f32(w & 0xffff0000u); }
template <> __device__ __forceinline__ float lo16<f16_t>(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
template <> __device__ __forceinline__ float hi16<f16_t>(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
template <typename T> __device__ __forceinline__ float round16(float v);
template <> __device__ __forceinline__ float round16<bf16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float round16<f16_t>(float v) { return (float)(_Float16)v; }
template <typename T> __device__ __forceinline__ constexpr uint16_t one16();
template <> __device__ __forceinline__ constexpr uint16_t one16<bf16_t>() { return 0x3f80; }
template <> __device__ __forceinline__ constexpr uint16_t one16<f16_t>() { return 0x3c00; }

// split-bf16 pair of an fp32 value (RF_BF16X3 operands): hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even; x - hi is exact
// in fp32, so hi + lo carries 16 significant bits of x
__device__ __forceinline__ void split4_bf16(const float* f, u32x2_t& hi, u32x2_t& lo) {
    hi[0] = pack_bf2(f[0], f[1]);
    hi[1] = pack_bf2(f[2], f[3]);
    const float r0 = f[0] - as_f32(hi[0] << 16), r1 = f[1] - as_f32(hi[0] & 0xffff0000u);
    const float r2 = f[2] - as_f32(hi[1] << 16), r3 = f[3] - as_f32(hi[1] & 0xffff0000u);
    lo[0] = pack_bf2(r0, r1);
    lo[1] = pack_bf2(r2, r3);
}

// ---- fp8 (OCP e4m3fn) activation quantisation with one E8M0 scale per 32-channel block (the A operand of the MX-scaled MFMA path)
// max over the 4 lanes of a quad (a 32-channel block = 4 consecutive lanes holding 8 channels each)
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0xB1, 0xF, 0xF, true)));      // quad_perm [1,0,3,2]
    v = fmaxf(v, as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x4E, 0xF, 0xF, true)));      // quad_perm [2,3,0,1]
    return v;
}
// E8M0 code c of the smallest power of two 2^(c - 127) with amax / 2^(c - 127) <= 448 = 1.75 * 2^8 (e4m3fn's largest finite value)
__device__ __forceinline__ int e8m0_for_amax(float amax) {
    const uint32_t b = as_u32(amax);
    int c = (int)((b >> 23) & 0xffu) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0);
    return c < 0 ? 0 : (c > 253 ? 253 : c);
}
// 8 floats -> 8 e4m3fn bytes of x * 2^(127 - code) (exact scaling, round-to-nearest-even, saturating)
__device__ __forceinline__ u32x2_t quant8_fp8(const float* f, int code) {
    const float inv = as_f32((uint32_t)(254 - code) << 23);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
    return u32x2_t{(uint32_t)lo, (uint32_t)hi};
}
// one lane's 8 consecutive channels of a 32-channel block (4 consecutive lanes): block amax -> scale code -> 8 bytes; lane 0 of the quad
// also stores the code
__device__ __forceinline__ void quant_block8(const float* f, fp8_t* qdst, fp8_t* sdst, bool quad_leader) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(f[e]));
    const int code = e8m0_for_amax(quad_max(m));
    *(u32x2_t*)qdst = quant8_fp8(f, code);
    if (quad_leader) *sdst = (fp8_t)code;
}

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int VEC = 4;   // elements per 16-byte vector
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct elem<fp8_t> {
    static constexpr int VEC = 16;
};
template <> struct elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct elem<f16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void store(f16_t* p, float v) { *p = (f16_t)v; }
};

// unpack a 16-byte vector of T into floats
template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = as_f32( v[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = as_f32( v[i] << 16);
        f[2 * i + 1] = as_f32( v[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = v[i];          // (by value: see the note on bit casts of vector elements above)
        f[2 * i] = lo16<f16_t>(w);
        f[2 * i + 1] = hi16<f16_t>(w);
    }
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* f);
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = as_u32( f[i]);
    return v;
}
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return v;
}

template <> __device__ __forceinline__ u32x4_t pack16<f16_t>(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_h2(f[2 * i], f[2 * i + 1]);
    return v;
}

// acc += A B^T over one pair of 16-byte fragments of 16-bit elements (8 k per lane half): the bf16 / fp16 MFMA of the same shape and rate
template <typename T> __device__ __forceinline__ void mma16(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<f16_t>(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// x * sigmoid(x) on the hardware transcendentals: v_exp_f32 (2^x, ~1 ulp) + v_rcp_f32 (~1 ulp): 4 VALU ops instead of the
// ~30 of libm expf + IEEE division (the GroupNorm+SiLU apply pass was VALU-bound on them).  Relative error ~1e-6.
__device__ __forceinline__ float silu_exact(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// erf(x) as a clamped rational approximation x * P(x^2) / Q(x^2) (the float kernel used by Eigen / XLA, error of a
// few ulp): branch-free, ~13 FMAs + 1 division -- the libm erff costs several times more VALU work, which made the
// GEGLU epilogue (84 M evaluations per 64x64-level FF layer) VALU-bound.
__device__ __forceinline__ float erf_rational(float x) {
    x = fminf(fmaxf(x, -4.0f), 4.0f);
    const float x2 = x * x;
    float a = -2.72614225801306e-10f;
    a = fmaf(a, x2, 2.77068142495902e-08f);
    a = fmaf(a, x2, -2.10102402082508e-06f);
    a = fmaf(a, x2, -5.69250639462346e-05f);
    a = fmaf(a, x2, -7.34990630326855e-04f);
    a = fmaf(a, x2, -2.95459980854025e-03f);
    a = fmaf(a, x2, -1.60960333262415e-02f);
    float b = -1.45660718464996e-05f;
    b = fmaf(b, x2, -2.13374055278905e-04f);
    b = fmaf(b, x2, -1.68282697438203e-03f);
    b = fmaf(b, x2, -7.37332916720468e-03f);
    b = fmaf(b, x2, -1.42647390514189e-02f);
    return x * a * __builtin_amdgcn_rcpf(b);      // b in [-0.0143, -4.4e-2 * ...]: bounded away from 0; v_rcp_f32 is ~1 ulp
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_rational(x * 0.70710678118654752440f)); }
// GELU for the 16-bit compute modes (bf16 / fp16): the sigmoid form  x * sigmoid(g(x))  on v_exp_f32 / v_rcp_f32 with an odd DEGREE-5 argument
//   g(x) = x (1.5950158 + 0.0740113 x^2 - 0.00070303 x^4)          (exactly: g = logit(Phi(x)); coefficients fitted for the max |error| of the product)
// -- the familiar tanh form is the degree-3 truncation (1.5957691 + 0.0713548 x^2) and is off by up to 4.7e-4 ABSOLUTE near |x| = 2.7, i.e. one fp16
// ulp of a hidden value of ~1 and a systematic (not random) deviation from the reference's erf GELU (attention.py:42-44).  One more FMA brings that
// to <= 2.6e-5 over the reals (round 6; tools-free check: tests/test_ops_gpu.py::test_geglu_negative_gates pins the bound), a twentieth of an fp16 ulp.
// The polynomial's x^4 term is negative: its argument is clamped to |x| <= 7, beyond which sigmoid(g) is 0 or 1 to eleven digits.  9 VALU operations
// (two of them transcendental) against ~25 for the erf form, which the exact-fp32 mode keeps.
__device__ __forceinline__ float gelu_sigmoid5(float x) {
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -7.0f), 7.0f);
    const float x2 = xc * xc;
    const float u = xc * __builtin_fmaf(__builtin_fmaf(-0.0010142630f, x2, 0.10677572f), x2, 2.3011212f);          // g(x) * log2(e)
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) {
    if constexpr (sizeof(T) == 2) return gelu_sigmoid5(x);
    else return gelu_erf(x);
}
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + expf(-1.702f * x)); }

// Reduce-scatter over the 32 lanes of a half-wave: every lane brings 32 values (index k), lane `lrow` leaves with the total of value index
// lrow in gx[0].  After the stage with mask m a lane keeps the half of its values selected by its own bit m: 16 + 8 + 4 + 2 + 1 exchanges.
// (bit select, not ?: -- the compiler turns a select between two array elements into a lane-indexed array access, i.e. a 32-way compare
//  chain per value)
__device__ __forceinline__ void halfwave_reduce_scatter32(float (&gx)[32], int lrow) {
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int m = 16 >> st, n = 16 >> st;           // lane mask, values kept after this stage
        const unsigned up = (lrow & m) ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const unsigned a = __float_as_uint(gx[k]), b = __float_as_uint(gx[k + n]);
            const float send = __uint_as_float((a & up) | (b & ~up));
            const float keep = __uint_as_float((b & up) | (a & ~up));
            gx[k] = keep + __shfl_xor(send, m, 64);
        }
    }
}

// Sum over the 64 lanes, result in every lane.  DPP / permlane-swap steps (a few cycles of latency each) instead of six dependent
// ds_bpermute round trips (__shfl_xor: ~120 cycles each -- the two reductions of a LayerNorm row were 1.5k cycles of pure latency).
// Within a row of 16 lanes: xor 1, xor 2, half-mirror, mirror; then v_permlane16_swap / v_permlane32_swap (gfx950) for rows / halves.
__device__ __forceinline__ float wave_sum(float v) {
    v += as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x140, 0xF, 0xF, true));     // row_mirror
    {
        const auto r = __builtin_amdgcn_permlane16_swap(as_u32(v), as_u32(v), false, false);
        v = as_f32(r[0]) + as_f32(r[1]);
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(as_u32(v), as_u32(v), false, false);
        v = as_f32(r[0]) + as_f32(r[1]);
    }
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(v), 0x140, 0xF, 0xF, true)));
    {
        const auto r = __builtin_amdgcn_permlane16_swap(as_u32(v), as_u32(v), false, false);
        v = fmaxf(as_f32(r[0]), as_f32(r[1]));
    }
    {
        const auto r = __builtin_amdgcn_permlane32_swap(as_u32(v), as_u32(v), false, false);
        v = fmaxf(as_f32(r[0]), as_f32(r[1]));
    }
    return v;
}

}  // namespace rf
