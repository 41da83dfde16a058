// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
//   out[m, n] = act(alpha * sum_k A[m, k] W[n, k] + bias[n] + rowvec[sample(m), n]) + residual[m, n]
//
// One kernel template serves both storage types through a common *byte* geometry: a K-tile is
// 128 bytes of K per row (64 bf16 or 32 fp32), LDS rows are 128 B with a 16-B-slot XOR swizzle,
// every lane feeds the MFMA from one 16-B fragment (lane l: row l&31, slot 2*kk + (l>>5)):
//   bf16 : 1 x v_mfma_f32_32x32x16_bf16 per fragment pair (8 k per lane-half)
//   fp16 : 1 x v_mfma_f32_32x32x16_f16  -- the same kernels instantiated for f16_t (RF_F16, the "fp16" throughput mode): identical geometry and
//          matrix-core rate, 11 significant bits per operand instead of 8; no fp8-weight (W8) and no split-operand (x3) variants
//   fp32 : 4 x v_mfma_f32_32x32x2_f32   per fragment pair (exact fp32 FMA chain, 157 TF peak)
// The A operand is gathered on the fly from channels-last source tensors (3x3 / 1x1, stride, asymmetric padding, nearest x2
// upsample folded into the addressing, 2-source channel concat).  Hot path (GLDS): both operands go global -> LDS directly
// (buffer_load_dwordx4 ... lds, no staging registers; the XOR swizzle is applied on the SOURCE side), two LDS stages, the pieces
// of tile kt+2 issued right behind the barrier that retires tile kt, a register-level fragment pipeline under the MFMAs.  Shapes the
// DMA cannot express (two sources, K tiles straddling a filter tap, fp32 operands with ragged K) take the register-staged loop
// (global -> registers -> LDS, next tile's loads issued before the current tile's MFMAs).  Epilogues: see the EPI parameter.
#include <cstdlib>
#include <type_traits>

#include "common.h"

// How the LDS-DMA pieces of tile kt+2 are issued behind the barrier that frees their stage (same-box A/B on the whole benchmark, r03):
//   0: one burst (both waves of every SIMD stall on ~9 x 60-180 issue cycles at the same time)        935.9 / 933.6 ms per batch
//   1: between the MFMA columns of k-step 3 (the issue slots hide under the matrix pipe)              921.4 / 920.0      <- default
//   2: as 1 with the barrier ahead of all of k-step 3's columns                                      920.4 / 920.5 (vs 921.5 / 921.6 for 1 on that box)
//   3: in thirds over k-step 3 and k-steps 0 / 1 of the next tile                                    965.3 / 964.1 (vs 939.9 / 937.0 for 1 on that box)
#ifndef RF_SPREAD_DMA
#define RF_SPREAD_DMA 1
#endif

// Experiment (RF_STORE_SC1 = 1): the direct epilogue's output stores as write-through `sc1` stores -- the bytes leave the XCD's L2 while the
// kernel runs instead of in the write-back burst at the kernel boundary (every dirty line must reach the memory side before the next kernel
// starts: the 8 L2s are not coherent), at the price of dropping the line from this XCD's L2.
#ifndef RF_STORE_SC1
#define RF_STORE_SC1 0
#endif

namespace rf {
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void st16_out(void* ptr, const u32x4_t& w) {
#if RF_STORE_SC1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(w) : "memory");
#else
    *(u32x4_t*)ptr = w;
#endif
}

struct GemmParams {
    int M, N, K;
    const void* src0;
    const void* src1;
    int C0, Ctot, ld0, ld1;
    int Hin, Win, Hout, Wout, KH, KW, stride, pad_t, pad_l, ups;
    const void* W;
    int ldw;
    const float* bias;
    const float* rowvec;
    int rows_per_sample, ldv;
    const void* residual;
    int ldr, act;
    const float* act_vec;
    void* out;
    int ldo;
    float alpha;
    long long sA, sW, sO, sR;
    int tiles_m, tiles_n;
    unsigned a_bytes, w_bytes;   // extents of src0 / W (per batch element) for the buffer descriptors
    int glds;        // use the direct-to-LDS main loop
    int korder;      // 1: K is ordered (channel chunk of BK, tap, channel-in-chunk) instead of (tap, channel)
    int vec_ok;      // epilogue may use 16-byte (fp32) / 8-byte (bf16) vector accesses
    int splitk;      // > 1: blockIdx.z owns a K range and writes raw fp32 partial sums to `ws` (epilogue in splitk_reduce_kernel)
    float* ws;       // [splitk][M][N] fp32
    // fused GroupNorm statistics of `out` (rf_conv_gemm_desc.gn_*): up to two consumers with their own channel grouping
    int pm, pn;            // tile order inside an XCD's run: patches of pm x pn tiles (M-fastest inside a patch), patches N-fastest inside a super-row
                           // of pm tile rows.  (1, tiles_n) = N-fastest strips, (tiles_m, 1) = M-fastest strips; chosen per launch by estimated
                           // L2-miss traffic (launch_cfg)
    int gn_rows;
    double* gn_part[2];
    int gn_cpg[2], gn_coff[2], gn_slot[2], gn_nch[2];
    int* plan;       // host only: rf_conv_gemm_plan / rf_conv_gemm_plan2 (no launch): {stat rows, stat cols, splitk, BM, BN, 32 TN, epilogue form, frag slabs}
    int epi2_ok;     // host only: operand alignment / feature set allow the direct (register -> global) epilogue
    int dbg;         // RF_GEMM_DBG (timing experiments only): bit 0 = skip the epilogue, bit 1 = skip the main loop
    const float* wscale;   // W8 kernels: per-output-channel power-of-two scale of the fp8 (e4m3fn) weights
    void* oscale;          // A8 kernels with GEGLU: `out` receives e4m3fn bytes (pitch ldo BYTES) and oscale one E8M0 code per (row, 32 output columns)
    int os_ld;
    const void* ascale;    // A8 kernels (fp8 activations): E8M0 scale bytes, one per (pixel, 32-channel block), pixel pitch as_ld bytes (a multiple of 4)
    int as_ld;
    unsigned as_bytes;
    // LayerNorm folded around two GEMMs (rf_conv_gemm_desc.ln_*).  Producer: per row and per 32 TN-column stripe of the wave tile, (mean, M2)
    // of the values as stored -> ln_out [M][ln_out_parts][2].  Consumer: out = rstd[m] (alpha acc - mean[m] ln_u[n]) + bias[n] with the row
    // statistics combined from ln_in's parts (Chan's formula); gamma rides in W, beta W^T in the bias (host).
    float* ln_out;
    int ln_out_parts;
    const float* ln_in;
    int ln_in_parts, ln_in_cols;
    float ln_eps;
    const float* ln_u;
    int x3;          // split-bf16 operands (RF_BF16X3): K counts VIRTUAL tiles, three per real 64-element K tile -- (A hi, W hi), (A hi, W lo),
                     // (A lo, W hi); W rows hold them in that order, the A lo plane lies lo_off bytes behind the hi plane of the same pixel
    int lo_off;
    long long w_ps;  // per-sample weights (rf_conv_gemm_desc.w_sample_stride, elements): the rows of sample s multiply W + s * w_ps; 0 = one W
};

// 8 fp8 (e4m3fn) weights -> 8 bf16, times the row's power-of-two scale: v_cvt_scalef32_pk_bf16_fp8, one instruction per pair (exact:
// 3 mantissa bits fit in bf16's 7, a power-of-two scale only moves the exponent)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
__device__ __forceinline__ u32x4_t fp8x8_to_bf16x8(const u32x2_t& r, float scale) {
    u32x4_t o;
    o[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], scale, false));
    o[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[0], scale, true));
    o[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], scale, false));
    o[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r[1], scale, true));
    return o;
}

template <typename T> struct MmaFrag;
template <> struct MmaFrag<bf16_t> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
#if RF_PIN_MFMA
        // experiment: MFMAs as volatile asm statements -- their order against each other and against the loop's waits / barrier is then the
        // source order (what fixed the fp8 x fp8 loop, where the builtin form was sunk below every fragment load)
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
#endif
    }
};
template <> struct MmaFrag<f16_t> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) { mma16<f16_t>(acc, a, b); }
};
template <> struct MmaFrag<fp8_t> {      // (the fp8 x fp8 path has its own main loop with block scales; this keeps the shared lambdas well-formed)
    __device__ static __forceinline__ void mma(f32x16_t&, const u32x4_t&, const u32x4_t&) {}
};
typedef int v8i_t __attribute__((ext_vector_type(8)));
// acc += (A row block, 64 K) x (B row block, 64 K)^T on the MX-scaled fp8 MFMA: each lane supplies 32 bytes per operand -- 16-byte slots h and
// 2 + h of the 64-byte K slice (h = lane >> 5) -- and ONE E8M0 scale per operand in byte 0 of sa / sb, which the hardware applies to the
// 32 consecutive K elements [32 h, 32 h + 32) of that lane's row (tools/mx_probe2.py pins this mapping on the chip)
__device__ __forceinline__ void mma_mx8(f32x16_t& acc, const u32x4_t& a0, const u32x4_t& a1, int sa, const u32x4_t& b0, const u32x4_t& b1, int sb) {
    const v8i_t a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const v8i_t b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    // inline asm, volatile: as a builtin (a pure call) hipcc sinks all MFMAs of a K tile to the end of the loop body, past every fragment
    // load and the barrier -- three fragment sets live (256 VGPRs + scratch) and no LDS latency hidden under the matrix pipe.  Volatile
    // statements keep their order against each other and against the loop's wait / barrier statements.  (s_nop 1: wait states between a
    // VALU write of a scale / operand register and the MFMA that reads it, which the compiler does not pad inside asm.)
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
template <> struct MmaFrag<float> {
    __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as_f32( a[q]), as_f32( b[q]), acc, 0, 0, 0);
    }
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for every outstanding
// global store to be acknowledged (~1-2 us each time in the chunked epilogue)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

template <bool W8> __device__ __forceinline__ int w_stage(int t, int buf) { return W8 ? ((t >> 1) & 1) : buf; }
template <bool W8> __device__ __forceinline__ int w_soff(int t) { return (W8 ? (t >> 1) : t) * 128; }

template <typename TO> __device__ __forceinline__ void store_out(TO* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <> __device__ __forceinline__ void store_out<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }
template <typename TO> __device__ __forceinline__ float load_out(const TO* p);
template <> __device__ __forceinline__ float load_out<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_out<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <> __device__ __forceinline__ float load_out<f16_t>(const f16_t* p) { return (float)*p; }
// 16-bit storage element (bf16 or fp16): the forms that differ only in their conversions share one code path
template <typename X> struct is16 { static constexpr bool value = std::is_same<X, bf16_t>::value || std::is_same<X, f16_t>::value; };

// EPI = 1: "direct" epilogue.  The MFMA operands are swapped (W fragment as the row operand, A fragment as the column operand)
// and the W rows of each 32-row block are read in the order  row(i') = 16*((i'>>2)&1) + 4*(i'>>3) + (i'&3), so that lane
// (m = l & 31, h = l >> 5) ends up with accumulator register r = output column 16*h + r of its row: 16 CONTIGUOUS columns per
// 32x32 block.  Bias / timestep vector / residual / GEGLU then happen in registers and every lane writes its row segments with 16-byte
// stores -- no staging pass of the tile through LDS (only the BN column constants are parked there, one barrier), and the waves
// retire independently.  EPI = 2: the whole tile staged ONCE as bf16 by all waves (opt-in, measured neutral).
// W8 = true: the weights are fp8 (e4m3fn) with one power-of-two scale per output channel.  A W tile in LDS keeps the 128-byte row
// geometry and therefore holds 128 K elements = the W operand of TWO consecutive A tiles: the W pieces are issued every other
// tile (half the weight bytes through L2 / LDS), and a W fragment is an 8-byte LDS read turned into 8 bf16 by 4 conversions
// (issued between the MFMAs of the previous fragment column).  The MFMA itself is the bf16 one: same matrix-core ceiling.
// LNF: LayerNorm role of the launch (rf_conv_gemm_desc.ln_*), a compile-time variant of the direct epilogue so that ordinary launches carry none
// of its registers: 0 none, 1 producer (row statistics of the stored output), 2 consumer (row affine; such launches have no residual)
// HX (round 4): 3x3 stride-1 convolutions with the K order (filter row dy, channel chunk, filter column dx) -- rf_conv_gemm_desc.korder = 2.  The three
// horizontal taps of one (dy, chunk) read the SAME image rows shifted by one pixel, so ONE row-extended A tile -- every image row of the output tile
// with one halo pixel on each side, (BM / Wout) * (Wout + 2) rows of 128 bytes -- is staged per group of three K tiles instead of one BM-row tile per
// K tile: a third of the A-operand fill (the per-CU operand fill is what caps these kernels).  Fragment reads address row  erow(pixel) + dx  of that
// tile (the XOR swizzle follows the row), the W side and everything behind the main loop are unchanged.
template <typename T, typename TO, int WM, int WN, int TM, int TN, bool CONV, bool GLDS, int NST, int EPI = 0, bool W8 = false, int LNF = 0, bool HX = false>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(const GemmParams p) {
    static_assert(LNF == 0 || (EPI == 1 && !CONV && !W8 && sizeof(T) == 2 && sizeof(TO) == 2), "LayerNorm folding: bf16 linear layers on the direct epilogue");
    static_assert(!HX || (CONV && GLDS && NST == 2 && sizeof(T) == 2 && !W8 && EPI != 2), "row-extended A tiles: bf16 3x3 convolutions on the two-stage direct-to-LDS loop");
    static_assert(EPI == 0 || GLDS, "the direct / packed epilogues are built on the direct-to-LDS main loop");
    static_assert(EPI != 2 || std::is_same<TO, bf16_t>::value, "the packed staged epilogue writes bf16");
    static_assert(!W8 || (GLDS && std::is_same<T, bf16_t>::value), "fp8 weights: bf16 activations on the direct-to-LDS main loop");
    // A8: fp8 (e4m3fn) activations with one E8M0 scale per 32 channels x fp8 weights with one power-of-two scale per output channel on
    // v_mfma_scale_f32_32x32x64_f8f6f4: a K tile (128-byte rows) holds 128 K elements, two k-steps of 64
    constexpr bool A8 = std::is_same<T, fp8_t>::value;
    static_assert(!A8 || (GLDS && !W8 && NST == 2), "fp8 activations: direct-to-LDS main loop, two stages");
    constexpr int NT = WM * WN * 64;
    constexpr int BM = 32 * TM * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int VEC = elem<T>::VEC;
    constexpr int BK = 8 * VEC;          // elements of K per tile (128 bytes)
    constexpr int RPP = NT / 8;          // rows covered per staging pass
    constexpr int AXR = HX ? ((BM * 5 / 4 + RPP - 1) / RPP) * RPP : BM;      // rows of an A stage (HX: up to BM + 2 BM / Wout rows, Wout >= 8)
    constexpr int AV = AXR / RPP;        // A vectors per thread
    constexpr int BV = BN / RPP;         // B vectors per thread
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/threads mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;                       // [2][AXR][128]
    char* const ldsB = smem + NST * AXR * 128;     // [NST][BN][128]
    char* const ldsS = smem + NST * (AXR + BN) * 128;     // A8: [2][8 waves][64 rows] scale dwords (4 E8M0 bytes = the 4 blocks of a K tile)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: XCD x (= block id mod 8 as dispatched) owns a contiguous run of tiles,
    // N-fastest inside the run, so the N-tiles that share an A panel hit the same L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // ... in 2-D patches: the ~32 tiles an XCD works on at any time (one block per CU) form a pm x pn patch, so that they share pm A panels and
    // pn W panels (a strip of 32 tiles along N touches every W panel of a wide layer: more than the 4 MB of L2 hold, re-streamed from the
    // Infinity Cache at ~7 TB/s -- tools/fill_probe.py: 11 B/clk/CU against 50 for L2 hits)
    int tile_m, tile_n;
    {
        const int srow = p.pm * p.tiles_n;                           // tiles per super-row of pm tile rows
        const int sr = bid / srow, rem = bid - sr * srow;
        const int h = min(p.pm, p.tiles_m - sr * p.pm);              // (the last super-row may be shorter)
        const int pw = h * p.pn;
        const int pc = rem / pw, r2 = rem - pc * pw;
        const int c = r2 / h;
        tile_m = sr * p.pm + (r2 - c * h);
        tile_n = pc * p.pn + c;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const long long zb = blockIdx.y;
    const T* src0 = (const T*)p.src0 + zb * p.sA;
    const T* src1 = (const T*)p.src1;
    const T* Wp = W8 ? (const T*)((const char*)p.W + zb * p.sW) : (const T*)p.W + zb * p.sW;
    if (!W8 && p.w_ps) Wp += (long long)(m0 / p.rows_per_sample) * p.w_ps;          // (a tile lies inside one sample: host)

    // EPI 1: the per-column constants of this tile (bias + the tile's timestep / context vector) are fetched NOW, one column per
    // thread, and parked in LDS after the main loop: fetched in the epilogue they cost one exposed L2 / HBM round trip per
    // 32-column block (5 per tile: 25 of the 60 us of the 65536x960x320 qkv GEMM, RF_GEMM_DBG=8 experiment r02g)
    float colc = 0.f;
    if constexpr (EPI == 1) {
        if (tid < BN && n0 + tid < p.N && p.splitk == 1) {
            if (p.bias) colc = p.bias[n0 + tid];
            if (p.rowvec) colc += p.rowvec[(long long)(m0 / p.rows_per_sample) * p.ldv + n0 + tid];
        }
    }
    float colu = 0.f;          // LayerNorm consumer: sum_k W[n, k] of this thread's column (the mean's coefficient)
    if constexpr (LNF == 2) {
        if (tid < BN && n0 + tid < p.N) colu = p.ln_u[n0 + tid];
    }
#if RF_RES_TOUCH
    // experiment: pull the residual rows of this tile into L2 now (one dword per 128-byte line, data discarded), so that the epilogue's residual
    // segments do not wait out HBM latency three times per wave tile
    uint32_t touch_[4] = {0u, 0u, 0u, 0u};
    if constexpr (EPI == 1) {
        if (p.residual && p.splitk == 1) {
            constexpr int EPL = 128 / (int)sizeof(TO), LPR = (BN + EPL - 1) / EPL;
            const TO* const rbase = (const TO*)p.residual + zb * p.sR;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + k * NT;
                const int row = m0 + i / LPR, c = n0 + (i % LPR) * EPL;
                if (i < BM * LPR && row < p.M && c < p.N) {
                    const TO* ptr = rbase + (long long)row * p.ldr + c;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(touch_[k]) : "v"(ptr) : "memory");
                }
            }
        }
    }
#endif
    const int r0 = tid >> 3;
    // GLDS: the LDS image of a direct-to-LDS load is lane-linear, so the XOR swizzle moves to the SOURCE: the lane that
    // lands on 16-byte position p of row r fetches k-slot p ^ ((r >> 1) & 7)  (r0 + 32*i keeps (r >> 1) & 7 for every i)
    const int slot = GLDS ? ((tid & 7) ^ ((r0 >> 1) & 7)) : (tid & 7);

    // ---- per-thread A row descriptors
    int a_pix[AV];     // CONV: b*Hin*Win ; plain: row index (or -1 invalid)
    int a_iy[AV], a_ix[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int m = m0 + r0 + i * RPP;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_pix[i] = (m < p.M) ? b * p.Hin * p.Win : -1;
            a_iy[i] = oy * p.stride - p.pad_t;
            a_ix[i] = ox * p.stride - p.pad_l;
        } else {
            a_pix[i] = (m < p.M) ? m : -1;
            a_iy[i] = a_ix[i] = 0;
        }
    }
    // running K position of this thread's vector slot
    int kvec = slot * VEC;       // element index along K
    int cv = kvec, ky = 0, kx = 0;   // CONV: channel within tap, tap coords
    if (CONV) {
        while (cv >= p.Ctot) {
            cv -= p.Ctot;
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    }
    const int Hv = p.ups ? p.Hin * 2 : p.Hin, Wv = p.ups ? p.Win * 2 : p.Win;

    // advance this thread's K position to the next K tile
    auto advance_k = [&]() {
        kvec += BK;
        if (CONV) {
            if (p.korder) {       // channel-chunk-major K: the KH*KW taps of one BK-channel chunk are consecutive tiles
                if (++kx == p.KW) { kx = 0; if (++ky == p.KH) { ky = 0; cv += BK; } }
            } else {
                cv += BK;
                while (cv >= p.Ctot) {
                    cv -= p.Ctot;
                    if (++kx == p.KW) { kx = 0; ++ky; }
                }
            }
        }
    };

    u32x4_t ra[AV], rb[BV];

    auto load_tiles = [&]() {
        const bool kval = kvec < p.K;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (CONV) {
                int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                const bool ok = kval && a_pix[i] >= 0 && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                if (ok) {
                    if (p.ups) { iy >>= 1; ix >>= 1; }
                    const long long pix = (long long)a_pix[i] + iy * p.Win + ix;
                    const T* ptr = (cv < p.C0) ? src0 + pix * p.ld0 + cv : src1 + pix * p.ld1 + (cv - p.C0);
                    v = *(const u32x4_t*)ptr;
                }
            } else {
                if (kval && a_pix[i] >= 0) v = *(const u32x4_t*)(src0 + (long long)a_pix[i] * p.ld0 + kvec);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int n = n0 + r0 + j * RPP;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (kval && n < p.N) v = *(const u32x4_t*)(Wp + (long long)n * p.ldw + kvec);
            rb[j] = v;
        }
        // advance to the next K tile
        advance_k();
    };
    auto store_tiles = [&](int buf) {
        char* a = ldsA + buf * BM * 128;
        char* b = ldsB + buf * BN * 128;
#pragma unroll
        for (int i = 0; i < AV; ++i) *(u32x4_t*)(a + lds_off(r0 + i * RPP, slot)) = ra[i];
#pragma unroll
        for (int j = 0; j < BV; ++j) *(u32x4_t*)(b + lds_off(r0 + j * RPP, slot)) = rb[j];
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    int nk = (RF_DBG(p, 2)) ? 0 : (p.K + BK - 1) / BK;
    int kb0_tiles = 0;       // first K tile of this block (split-K)
    if (p.splitk > 1) {          // this block's K-tile range [kb0, kb0 + nk)
        int per = (nk + p.splitk - 1) / p.splitk;
        if (p.x3 || HX) per = (per + 2) / 3 * 3;          // whole groups of three tiles (split-bf16 passes / the three dx taps of an HX group)
        const int kb0 = blockIdx.z * per;
        nk = max(0, min(nk, kb0 + per) - kb0);
        kb0_tiles = kb0;
        if (!GLDS)
            for (int i = 0; i < kb0; ++i) advance_k();
    }
    // K-tile counts are wave-uniform, but hipcc's divergence analysis does not see it through the split-K arithmetic: everything derived from
    // them (the issue state, the SGPR offsets of the LDS-DMA pieces, the `more` conditions) would live in VGPRs, every DMA piece would sit in a
    // waterfall loop (v_readfirstlane / v_cmp / s_and_saveexec / branch: ~10 instructions and a serialisation point per piece, nine pieces per
    // tile right behind the barrier) and every `if (more)` would be an exec-masked region
    nk = __builtin_amdgcn_readfirstlane(nk);
    kb0_tiles = __builtin_amdgcn_readfirstlane(kb0_tiles);
    const int lrow = lane & 31, lhalf = lane >> 5;

    auto compute_tile = [&](int buf) {
        const char* a = ldsA + buf * BM * 128;
        const char* b = ldsB + buf * BN * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int s = kk * 2 + lhalf;
            u32x4_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4_t*)(a + lds_off((wm * TM + i) * 32 + lrow, s));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const u32x4_t*)(b + lds_off((wn * TN + j) * 32 + lrow, s));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) MmaFrag<T>::mma(acc[i][j], fa[i], fb[j]);
        }
    };

    if constexpr (GLDS) {
        // Direct global -> LDS staging (buffer_load_dwordx4 ... lds) with a register-level fragment pipeline.
        // Host guarantees (launch_typed): one source, K % BK == 0 and, for convs, Ctot % BK == 0 with tap-major K -- so a K tile
        // lies inside ONE filter tap and all of its addresses are  per-lane offset (fixed within a tap) + uniform K offset
        // (SGPR soffset).  Out-of-range rows / padding taps carry an out-of-range voffset: the buffer range check returns zeros.
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)src0, 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, p.w_bytes, 0x00020000);
        constexpr int OOB = 0x7fffffff;
        constexpr int NP = AV + BV;
        static_assert(NP <= 20, "piece table too small");
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const int lane_k = slot * VEC * (int)sizeof(T);       // byte offset of this lane's 16-byte K slot inside a K tile
        int offs[20];        // [0, AV): A pieces, [AV, NP): B pieces (fixed size: see the note on hipcc at `rowd`)
        unsigned rowd[8];    // CONV: packed (sample << 20 | oy << 10 | ox) of this thread's A rows, ~0u = row beyond M
        static_assert(AV <= 8, "row table too small");
        // A8: lane l of wave w also fetches the scale dword of tile row 64 w + l (one 4-byte piece per wave and K tile)
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(A8 ? p.ascale : (const void*)src0), 0, A8 ? p.as_bytes : 0u, 0x00020000);
        unsigned srowd = ~0u;
        int soffs = OOB;
        if constexpr (A8) {
            static_assert(BM <= NT, "one scale row per thread");
            const int srow = wave_u * 64 + lane, m = m0 + srow;
            const bool sv = srow < BM && m < p.M;
            if (CONV) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                srowd = sv ? ((unsigned)b << 20 | (unsigned)oy << 10 | (unsigned)ox) : ~0u;
            } else {
                soffs = sv ? m * p.as_ld : OOB;
            }
        }
#pragma unroll
        for (int j = 0; j < BV; ++j) {
            const int n = n0 + r0 + j * RPP;
            offs[AV + j] = n < p.N ? n * p.ldw * (W8 ? 1 : (int)sizeof(T)) + lane_k : OOB;
        }
        // W8: W tile of A tile `t` = 128-byte tile (t >> 1) of the weight rows, kept in W stage ((t >> 1) & 1)
        // the W pieces of the first tile go out before the A row descriptors (two integer divisions per row) are computed
        if (nk > 0) {
            // (values computed OUTSIDE the builtin's argument list: a call expression inside it makes hipcc's host pass drop the
            // kernel's stub without a diagnostic)
            char* const b0 = ldsB + w_stage<W8>(kb0_tiles, 0) * BN * 128 + wave_u * 1024;
            const int so0 = __builtin_amdgcn_readfirstlane(w_soff<W8>(kb0_tiles));
#pragma unroll
            for (int j = 0; j < BV; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(b0 + j * (RPP * 128)), 16, offs[AV + j], so0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int m = m0 + r0 + i * RPP;
            if constexpr (HX) {
                // row j of the row-extended tile = image row (m0 / Wout + j / (Wout + 2)) over all samples, column code xc = j % (Wout + 2)
                // (input column xc - 1: the halo pixels are xc = 0 and xc = Wout + 1)
                const int j = r0 + i * RPP, we = p.Wout + 2;
                const int ir = j / we, xc = j - ir * we;
                const int R = m0 / p.Wout + ir;                       // (host: BM % Wout == 0, so m0 is the first pixel of an image row)
                const int b = R / p.Hout, oy = R - b * p.Hout;
                rowd[i] = (ir < BM / p.Wout && R * p.Wout < p.M) ? ((unsigned)b << 20 | (unsigned)oy << 10 | (unsigned)xc) : ~0u;
            } else if (CONV) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                rowd[i] = m < p.M ? ((unsigned)b << 20 | (unsigned)oy << 10 | (unsigned)ox) : ~0u;
            } else {
                offs[i] = m < p.M ? m * p.ld0 * (int)sizeof(T) + lane_k : OOB;
            }
        }
        // HX: A-piece offsets of filter row dy (3x3, stride 1, pad 1: input row oy + dy - 1, input column xc - 1), once per dy
        auto set_grp = [&](int dy) {
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const unsigned d = rowd[i];
                const int iy = (int)((d >> 10) & 1023u) + dy - 1, ix = (int)(d & 1023u) - 1;
                const bool ok = d != ~0u && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                offs[i] = ok ? (((int)(d >> 20) * p.Hin + iy) * p.Win + ix) * p.ld0 * (int)sizeof(T) + lane_k : OOB;
            }
        };
        // A-piece offsets of filter tap (ty, tx): padding / upsampling / stride live here, once per tap
        auto set_tap = [&](int ty, int tx) {
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const unsigned d = rowd[i];
                int iy = (int)((d >> 10) & 1023u) * p.stride - p.pad_t + ty;
                int ix = (int)(d & 1023u) * p.stride - p.pad_l + tx;
                const bool ok = d != ~0u && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                offs[i] = ok ? (((int)(d >> 20) * p.Hin + iy) * p.Win + ix) * p.ld0 * (int)sizeof(T) + lane_k : OOB;
            }
            if constexpr (A8) {
                const unsigned d = srowd;
                int iy = (int)((d >> 10) & 1023u) * p.stride - p.pad_t + ty;
                int ix = (int)(d & 1023u) * p.stride - p.pad_l + tx;
                const bool ok = d != ~0u && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                if (p.ups) { iy >>= 1; ix >>= 1; }
                soffs = ok ? (((int)(d >> 20) * p.Hin + iy) * p.Win + ix) * p.as_ld : OOB;
            }
        };
        // uniform K state of the NEXT tile to issue: absolute tile index, and for convs (tap, channel chunk inside the tap)
        const int tpt = CONV ? p.Ctot / BK : 1;     // K tiles per tap
        // korder 1 (channel-chunk-major K): the KH*KW taps of one BK-channel chunk are consecutive tiles, so the ~BM x 128 B of A
        // that a block touches per chunk stay L2-resident across the taps (tap-major K sweeps all channels between reuses)
        // X3OK: the split-bf16 mode (bf16 operand pairs, fp32 output) -- `it` counts virtual tiles (the W side), `ia` the real A tile,
        // `ph` the pass (0: hi x hi, 1: hi x lo, 2: lo x hi) of the tile being issued
        constexpr bool X3OK = sizeof(T) == 2 && sizeof(TO) == 4 && !W8 && NST == 2;
        const bool x3 = X3OK && p.x3;
        int it = kb0_tiles, ity = 0, itx = 0, ic = 0, ph = 0;
        int ia = x3 ? kb0_tiles / 3 : kb0_tiles;          // (split-K ranges of the split mode start on a multiple of 3)
        int hx_dx = 0, hx_ist = 0;                         // HX issue state: filter column of tile `it`, A stage of its group
        if constexpr (HX) {
            const int grp = kb0_tiles / 3;                 // K order (dy, chunk, dx): group = dy * tpt + chunk
            ity = grp / tpt;
            ic = grp - ity * tpt;
            set_grp(ity);
        } else
        if (CONV) {
            const int ntap = p.KH * p.KW;
            const int tap = p.korder ? ia % ntap : ia / tpt;
            ic = p.korder ? ia / ntap : ia - tap * tpt;
            ity = tap / p.KW;
            itx = tap - ity * p.KW;
            set_tap(ity, itx);
        }
        auto next_tile = [&]() {     // advance the issue state by one K tile
            ++it;
            if constexpr (HX) {
                if (++hx_dx == 3) {
                    hx_dx = 0;
                    hx_ist ^= 1;
                    if (++ic == tpt) { ic = 0; ++ity; set_grp(ity); }
                }
                return;
            }
            if (X3OK && x3) {
                if (++ph != 3) return;
                ph = 0;
            }
            ++ia;
            if (CONV) {
                if (p.korder) {
                    if (++itx == p.KW) { itx = 0; if (++ity == p.KH) { ity = 0; ++ic; } }
                    set_tap(ity, itx);
                } else if (++ic == tpt) {
                    ic = 0;
                    if (++itx == p.KW) { itx = 0; ++ity; }
                    set_tap(ity, itx);
                }
            }
        };
        // issue pieces [q0, q1) of the issue-state tile into LDS stage `buf`
        auto issue_pieces = [&](int buf, int q0, int q1) {
            char* a = ldsA + (HX ? hx_ist : buf) * AXR * 128 + wave_u * 1024;          // (HX: the A stage belongs to the group of three tiles)
            char* b = ldsB + w_stage<W8>(it, buf) * BN * 128 + wave_u * 1024;
            const int soA = __builtin_amdgcn_readfirstlane((CONV ? ic : ia) * 128 + ((X3OK && ph == 2) ? p.lo_off : 0));
            const int soB = __builtin_amdgcn_readfirstlane(w_soff<W8>(it));
            if (W8 && (it & 1) && q1 > AV) q1 = AV;          // odd tile: its W half arrived with the even tile before it
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (q >= q0 && q < q1) {
                    if (q < AV) {
                        // (RF_GEMM_DBG bit 8, timing only: the A pieces of two K tiles out of three are not issued -- the fill a row-extended A
                        //  tile shared by the three horizontal taps of a 3x3 window would need; stale operands, wrong results)
                        if (!(RF_DBG(p, 256) && CONV && p.KW == 3 && p.stride == 1 && !p.ups && (it % 3) != 0) && !(HX && hx_dx != 0))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(a + q * (RPP * 128)), 16, offs[q], soA, 0, 0);
                    } else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(b + (q - AV) * (RPP * 128)), 16, offs[q], soB, 0, 0);
                }
            }
            if (A8 && q0 == 0) {          // the call that starts a tile also issues its scale piece: NP + 1 pieces per wave and tile
                char* const sdst = ldsS + buf * 2048 + wave_u * 256;
                const int soS = __builtin_amdgcn_readfirstlane((CONV ? ic : ia) * 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) void*)sdst, 4, soffs, soS, 0, 0);
            }
        };
        if constexpr (A8) {
            // ---- fp8 x fp8 main loop: two k-steps of 64 per K tile, A fragments double-buffered, W fragments rotated in place (column j of
            // the next k-step is fetched right behind column j's MFMAs), one barrier per tile -- the structure of the bf16 loop below at twice
            // the K per tile and per MFMA (64 matrix-pipe cycles each)
            using std::integral_constant;
            constexpr int JS8 = TN > 2 ? 2 : 1;
            u32x4_t fa8[2][TM][2], fb8[TN][2];
            int sa8[2][TM], wsc[TN];
            const int lrow_ = lane & 31, lhalf_ = lane >> 5;
            const int frag_sw8 = (lrow_ >> 1) & 7;
            const int brow8 = EPI == 1 ? (16 * ((lrow_ >> 2) & 1) + 4 * (lrow_ >> 3) + (lrow_ & 3)) : lrow_;
            const int frag_swb8 = (brow8 >> 1) & 7;
            int fk8[2], fkb8[2], ssh[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                fk8[kk] = ((4 * kk + lhalf_) ^ frag_sw8) << 4;           // 16-byte slot 4 kk + h; the lane's second slot (4 kk + 2 + h) is this ^ 32
                fkb8[kk] = ((4 * kk + lhalf_) ^ frag_swb8) << 4;
                ssh[kk] = (2 * kk + lhalf_) * 8;                        // the lane's 32-channel block inside the K tile: byte 2 kk + h of the scale dword
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + brow8;
                wsc[j] = n < p.N ? (int)((as_u32(p.wscale[n]) >> 23) & 0xffu) : 127;          // power-of-two fp32 scale -> its E8M0 code
            }
            const char* curA = ldsA + (wm * TM) * 4096 + lrow_ * 128;
            const char* curB = ldsB + (wn * TN) * 4096 + brow8 * 128;
            const char* curS = ldsS + ((wm * TM) * 32 + lrow_) * 4;
            const char* othA = curA + BM * 128;
            const char* othB = curB + BN * 128;
            const char* othS = curS + 2048;
            auto ld_a = [&](int buf, const char* bA, const char* bS, int kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    fa8[buf][i][0] = *(const u32x4_t*)(bA + i * 4096 + fk8[kk]);
                    fa8[buf][i][1] = *(const u32x4_t*)(bA + i * 4096 + (fk8[kk] ^ 32));
                    sa8[buf][i] = (int)((*(const uint32_t*)(bS + i * 128) >> ssh[kk]) & 0xffu);
                }
            };
            auto ld_b = [&](int j, const char* bB, int kk) {
                fb8[j][0] = *(const u32x4_t*)(bB + j * 4096 + fkb8[kk]);
                fb8[j][1] = *(const u32x4_t*)(bB + j * 4096 + (fkb8[kk] ^ 32));
            };
            auto mma_col = [&](int buf, int j) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (EPI == 1) mma_mx8(acc[i][j], fb8[j][0], fb8[j][1], wsc[j], fa8[buf][i][0], fa8[buf][i][1], sa8[buf][i]);
                    else mma_mx8(acc[i][j], fa8[buf][i][0], fa8[buf][i][1], sa8[buf][i], fb8[j][0], fb8[j][1], wsc[j]);
                }
            };
            auto phase8 = [&](auto KK, int stage, bool more) {
                constexpr int kk = decltype(KK)::value;
                constexpr int cur = kk, nx = kk ^ 1, nkk = kk ^ 1;
                const char* const nA = kk == 0 ? curA : othA;
                const char* const nB = kk == 0 ? curB : othB;
                const char* const nS = kk == 0 ? curS : othS;
#pragma unroll
                for (int j = 0; j < JS8; ++j) mma_col(cur, j);
                if (kk == 1) {
                    // own pieces of the next tile have landed and every fragment of this tile is in registers; past the barrier that holds
                    // for all waves: this stage may be overwritten (tile kt+2) and the other stage may be read
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                ld_a(nx, nA, nS, nkk);
#pragma unroll
                for (int j = 0; j < JS8; ++j) ld_b(j, nB, nkk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = JS8; j < TN; ++j) {
                    mma_col(cur, j);
                    ld_b(j, nB, nkk);
                    if (kk == 1) {          // the pieces of tile kt+2 go out between the MFMA columns behind the barrier (RF_SPREAD_DMA 1)
                        constexpr int COLS = TN - JS8, PPC = (NP + COLS - 1) / COLS;
                        if (more) issue_pieces(stage, (j - JS8) * PPC, (j - JS8 + 1) * PPC < NP ? (j - JS8 + 1) * PPC : NP);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (nk > 0) {
                issue_pieces(0, 0, AV);               // (its W pieces are already in flight; + the scale piece)
                if (nk > 1) {
                    next_tile();
                    issue_pieces(1, 0, NP);
                    if (nk > 2) next_tile();
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + 1) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                ld_a(0, curA, curS, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) ld_b(j, curB, 0);
            }
#if RF_RES_TOUCH
            // (the touch loads are older than every piece: the prologue's wait covered them; their registers are free from here on)
            if (nk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::"v"(touch_[0]), "v"(touch_[1]), "v"(touch_[2]), "v"(touch_[3]));
#endif
            for (int kt = 0; kt < nk; ++kt) {
                const int stage = kt & 1;
                const bool more = kt + 2 < nk;
                phase8(integral_constant<int, 0>{}, stage, more);
                phase8(integral_constant<int, 1>{}, stage, more);
                if (kt + 3 < nk) next_tile();
                const char* t = curA; curA = othA; othA = t;
                t = curB; curB = othB; othB = t;
                t = curS; curS = othS; othS = t;
            }
            // the MFMAs are asm statements: hipcc's hazard recognizer pads nothing between the last of them (16 passes) and the epilogue's first
            // VALU read of an accumulator -- 2 x 16 wait states here cover the required 18 whatever the barrier below costs
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            __syncthreads();
        } else {
        // Fragment pipeline: the ds_reads of k-step kk+1 are issued while the MFMAs of k-step kk run, so the matrix pipe does not
        // wait on LDS latency (all waves of a block are phase-locked by the per-tile barrier, nobody else would cover it).
        // Small wave tiles keep two full fragment sets; the 2x5 wave tile has no registers for that and rotates its B fragments
        // in place: column j of the next k-step is fetched right after column j's MFMAs have been issued.
        constexpr bool ROT = TM * TN > 8;
        constexpr int JS_ = ROT ? 2 : 0;      // B columns whose MFMAs go ahead of the first next-fragment reads
        u32x4_t fa[2][TM], fb[W8 ? 1 : 2][W8 ? 1 : TN];
        u32x2_t fbr[W8 ? 2 : 1][W8 ? TN : 1];          // W8: raw 8-byte fp8 fragments
        float ws[W8 ? TN : 1];                         // W8: scale of this lane's W row in each 32-row block
        const int frag_sw = (lrow >> 1) & 7;
        // W-tile row this lane reads for the row operand of its 32-row blocks (EPI = 1: permuted, see the kernel comment)
        const int brow = EPI == 1 ? (16 * ((lrow >> 2) & 1) + 4 * (lrow >> 3) + (lrow & 3)) : lrow;
        const int frag_swb = (brow >> 1) & 7;
        int fk[4], fkb[4];   // swizzled byte position of k-step kk inside a fragment row (A side, W side)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fk[kk] = ((kk * 2 + lhalf) ^ frag_sw) << 4;
            // W8: one 16-byte slot per k-step (16 fp8), the lane half picks its 8 bytes; the A-tile parity flips address bit 6
            fkb[kk] = W8 ? (((kk ^ frag_swb) << 4) + lhalf * 8) : (((kk * 2 + lhalf) ^ frag_swb) << 4);
        }
        if constexpr (W8) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wn * TN + j) * 32 + brow;
                ws[j] = n < p.N ? p.wscale[n] : 1.0f;
            }
        }
        // fragment row bases of the stage being multiplied (cur) and of the other stage (oth); swapped after every tile.
        // The loop body is ONE straight-line block (no per-tile variants): branches around MFMAs make the register allocator
        // keep two copies of the accumulators.
        const char* curA = ldsA + (wm * TM) * 4096 + lrow * 128;
        // HX: this lane's row of block i in the row-extended tile for filter column 0 (erow), the A row pointer / swizzle term of the tile being
        // multiplied (hcb / hcs) and of the next one (hnb / hns), the filter column and A stage of the tile being multiplied
        int erow[TM];
        const char* hcb[TM];
        const char* hnb[TM];
        int hcs[TM], hns[TM], m_dx = 0, m_st = 0;
        const int ks_[4] = {(0 + lhalf) << 4, (2 + lhalf) << 4, (4 + lhalf) << 4, (6 + lhalf) << 4};          // k-step kk -> 16-byte slot 2 kk + half
        auto hx_set = [&](const char** hb, int* hs, int st, int dx) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int e = erow[i] + dx;
                hb[i] = ldsA + st * (AXR * 128) + e * 128;
                hs[i] = ((e >> 1) & 7) << 4;
            }
        };
        if constexpr (HX) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pm = (wm * TM + i) * 32 + lrow, ir = pm / p.Wout;
                erow[i] = ir * (p.Wout + 2) + (pm - ir * p.Wout);
            }
            hx_set(hcb, hcs, 0, 0);
            hx_set(hnb, hns, 0, 1);
        }
        const char* const wbase = ldsB + (wn * TN) * 4096 + brow * 128;
        const char* curB = wbase + w_stage<W8>(kb0_tiles, 0) * BN * 128;
        const char* othA = curA + BM * 128;
        const char* othB = wbase + w_stage<W8>(kb0_tiles + 1, 1) * BN * 128;       // W8: the NEXT tile's W stage (may equal the current one)
        int curPar = W8 ? (kb0_tiles & 1) * 64 : 0, othPar = W8 ? ((kb0_tiles + 1) & 1) * 64 : 0;
        int t_abs = kb0_tiles;                                                   // absolute index of the tile being multiplied
        auto load_b = [&](int set, int j, const char* ptr) {
            if constexpr (W8) fbr[set][j] = *(const u32x2_t*)ptr;
            else fb[set][j] = *(const u32x4_t*)ptr;
        };
        auto bfrag = [&](int set, int j) -> u32x4_t {
            if constexpr (W8) return fp8x8_to_bf16x8(fbr[set][j], ws[j]);
            else return fb[set][j];
        };
        // one k-step: its MFMAs, the fetch of the next step's fragments (k-step 3 fetches from the other stage, after the
        // barrier; on the last tile that fetch reads stale bytes that are never used) and a third of the next tile's pieces
        const bool late = RF_SPREAD_DMA == 4 && NT == 512 && wave_u >= 4;
        auto phase = [&](auto KK, int stage, bool more, bool deep, bool cont = false) {
            constexpr int kk = decltype(KK)::value;
            constexpr int cur = kk & 1, nx = cur ^ 1;
            constexpr int fbc = ROT ? 0 : cur, fbn = ROT ? 0 : nx;
            constexpr int nkk = (kk + 1) & 3;
            const char* const nA = (kk < 3 ? curA : othA) + fk[nkk];
            const char* const nB = (kk < 3 ? curB : othB) + (fkb[nkk] ^ (kk < 3 ? curPar : othPar));
#if RF_SPREAD_DMA == 2
            constexpr int JS = (kk == 3 && ROT) ? 0 : JS_;          // k-step 3: barrier first, every MFMA column behind it takes its share of the pieces
#else
            constexpr int JS = JS_;
#endif
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                const u32x4_t bj = bfrag(fbc, j);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (EPI == 1) MmaFrag<T>::mma(acc[i][j], bj, fa[cur][i]);
                    else MmaFrag<T>::mma(acc[i][j], fa[cur][i], bj);
                }
            }
            if (kk == 3) {
                // own pieces of the next tile have landed and every fragment of this tile is in registers; past the barrier
                // that holds for all waves: this stage may be overwritten (tile kt+2) and the other stage may be read
                // (NST > 2: tiles kt+2 .. kt+NST-1 stay in flight behind the one that must have landed -- `deep` says all of them exist)
                if (NST > 2 && deep) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * NP) : "memory");
                else if (RF_DBG(p, 64)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // timing decomposition: the pieces are never waited for
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (HX) fa[nx][i] = *(const u32x4_t*)((kk < 3 ? hcb[i] : hnb[i]) + (ks_[nkk] ^ (kk < 3 ? hcs[i] : hns[i])));
                else fa[nx][i] = *(const u32x4_t*)(nA + i * 4096);
            }
#pragma unroll
            for (int j = 0; j < (ROT ? JS : TN); ++j) load_b(fbn, j, nB + j * 4096);
#if !RF_SPREAD_DMA
            if (kk == 3 && more && !RF_DBG(p, 32)) issue_pieces(stage, 0, NP);      // 'more' here: tile kt+2 exists  (RF_GEMM_DBG bit 5: no DMA in
                                                                                    // the main loop -- stale operands, timing decomposition only)
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = JS; j < TN; ++j) {
                const u32x4_t bj = bfrag(fbc, j);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (EPI == 1) MmaFrag<T>::mma(acc[i][j], bj, fa[cur][i]);
                    else MmaFrag<T>::mma(acc[i][j], fa[cur][i], bj);
                }
                if (ROT) {
                    load_b(0, j, nB + j * 4096);
#if !RF_SPREAD_DMA
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
#if RF_SPREAD_DMA
                // the pieces of tile kt+2 go out between the MFMA columns of k-step 3 (their issue slots hide under the matrix pipe) instead of
                // in one burst behind the barrier, where both waves of a SIMD stall on ~9 x 60-180 issue cycles at the same time
#if RF_SPREAD_DMA == 3
                // ... in thirds: behind the barrier of k-step 3, then at k-steps 0 and 1 of the next tile (stage ^ 1 there: the stage that tile's
                // predecessor freed); the issue state advances after k-step 1
                if (kk != 2) {
                    constexpr int G = kk == 3 ? 0 : (kk == 0 ? 1 : 2), gs = G * NP / 3, gn = (G + 1) * NP / 3 - gs, COLS = TN - JS;
                    const int q0 = gs + (j - JS) * gn / COLS, q1 = gs + (j - JS + 1) * gn / COLS;          // (constants after unrolling)
                    if (q1 > q0 && !RF_DBG(p, 32)) {
                        if (kk == 3) { if (more) issue_pieces(stage, q0, q1); }
                        else if (cont) issue_pieces(stage ^ 1, q0, q1);
                    }
                }
                if (ROT || kk != 2) __builtin_amdgcn_sched_barrier(0);
#elif RF_SPREAD_DMA == 4
                // 8-wave blocks: the waves that share a SIMD issue at different times -- waves 0-3 behind the barrier of k-step 3 (tile kt+2 into
                // the stage just freed), waves 4-7 at k-step 0 of the next tile (`cont`: into stage ^ 1) -- so that one wave of every SIMD feeds
                // the matrix pipe while the other is stalled on the address path
                if (kk == 3 || kk == 0) {
                    constexpr int COLS = TN - JS, PPC = (NP + COLS - 1) / COLS;
                    const int q0 = (j - JS) * PPC, q1 = (j - JS + 1) * PPC < NP ? (j - JS + 1) * PPC : NP;
                    if (kk == 3) { if (more && !late) issue_pieces(stage, q0, q1); }
                    else if (cont && late) issue_pieces(stage ^ 1, q0, q1);
                }
                if (ROT || kk == 3 || kk == 0) __builtin_amdgcn_sched_barrier(0);
#else
                if (kk == 3) {
                    constexpr int COLS = TN - JS, PPC = (NP + COLS - 1) / COLS;
                    if (more && !RF_DBG(p, 32)) issue_pieces(stage, (j - JS) * PPC, (j - JS + 1) * PPC < NP ? (j - JS + 1) * PPC : NP);
                }
                if (ROT || kk == 3) __builtin_amdgcn_sched_barrier(0);
#endif
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using std::integral_constant;
        if (nk > 0) {
            // prologue: tile 0 lands, its first fragments are fetched, the issue state points at the next tile to issue (tile NST)
            issue_pieces(0, 0, AV);               // (its W pieces are already in flight)
            if constexpr (NST > 2) {
                static_assert(!W8, "the deep ring is for bf16 / fp32 weights");
#pragma unroll
                for (int s = 1; s < NST; ++s)
                    if (s < nk) {
                        next_tile();
                        issue_pieces(s, 0, NP);
                    }
                if (nk > NST) next_tile();
                if (nk >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (nk > 1) {
                next_tile();
                issue_pieces(1, 0, NP);
                if (nk > 2) next_tile();
                if constexpr (W8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile 1 may have had no W pieces: no fixed count
                else if (RF_DBG(p, 128)) {}          // timing decomposition: the first tile is not waited for (upper bound of what a tile loop that
                                                      // issues the next tile's first pieces ahead of the epilogue could hide)
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HX ? BV : NP) : "memory");          // (HX: tile 1 -- filter column 1 -- has W pieces only)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (HX) fa[0][i] = *(const u32x4_t*)(hcb[i] + (ks_[0] ^ hcs[i]));
                else fa[0][i] = *(const u32x4_t*)(curA + i * 4096 + fk[0]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) load_b(0, j, curB + j * 4096 + (fkb[0] ^ curPar));
        }
#if RF_RES_TOUCH
        // (the touch loads are older than every piece: the prologue's wait covered them; their registers are free from here on)
        if (nk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(touch_[0]), "v"(touch_[1]), "v"(touch_[2]), "v"(touch_[3]));
#endif
        if constexpr (NST > 2) {
            // ring of NST stages: tile kt is multiplied out of stage kt % NST while tiles kt+1 .. kt+NST-1 are in LDS or in flight; the
            // stage it frees takes tile kt+NST.  For launches of at most one block per CU, where no second block covers the latency
            // of the single tile of look-ahead that two stages give.
            const char* const baseA = curA;
            int sc = 0, so = 1;                   // stages of the tile being multiplied / of the next one
            for (int kt = 0; kt < nk; ++kt) {
                const bool more = kt + NST < nk, deep = kt + NST - 1 < nk;
                phase(integral_constant<int, 0>{}, sc, more, deep);
                phase(integral_constant<int, 1>{}, sc, more, deep);
                phase(integral_constant<int, 2>{}, sc, more, deep);
                phase(integral_constant<int, 3>{}, sc, more, deep);
                if (kt + NST + 1 < nk) next_tile();
                sc = so;
                so = so + 1 == NST ? 0 : so + 1;
                curA = othA; othA = baseA + so * (BM * 128);
                curB = othB; othB = wbase + so * (BN * 128);
            }
        } else {
        for (int kt = 0; kt < nk; ++kt) {
            const int stage = kt & 1;
            const bool more = kt + 2 < nk;
#if RF_SPREAD_DMA == 4
            const bool cont = kt >= 1 && kt + 1 < nk;          // late waves: tile kt+1 is still to be issued (k-step 0)
            phase(integral_constant<int, 0>{}, stage, more, false, cont);
            if (late && kt >= 1 && kt + 2 < nk) next_tile();
            phase(integral_constant<int, 1>{}, stage, more, false);
            phase(integral_constant<int, 2>{}, stage, more, false);
            phase(integral_constant<int, 3>{}, stage, more, false);
            if (!late && kt + 3 < nk) next_tile();
#elif RF_SPREAD_DMA == 3
            const bool cont = kt >= 1 && kt + 1 < nk;          // tile kt+1's second and third thirds are still to be issued
            phase(integral_constant<int, 0>{}, stage, more, false, cont);
            phase(integral_constant<int, 1>{}, stage, more, false, cont);
            if (kt >= 1 && kt + 2 < nk) next_tile();
            phase(integral_constant<int, 2>{}, stage, more, false);
            phase(integral_constant<int, 3>{}, stage, more, false);
#else
            phase(integral_constant<int, 0>{}, stage, more, false);
            phase(integral_constant<int, 1>{}, stage, more, false);
            phase(integral_constant<int, 2>{}, stage, more, false);
            phase(integral_constant<int, 3>{}, stage, more, false);
            if (kt + 3 < nk) next_tile();
#endif
            const char* t = curA; curA = othA; othA = t;
            if constexpr (HX) {          // the tile multiplied next becomes current; its successor: next filter column, or column 0 of the next group's stage
                if (++m_dx == 3) { m_dx = 0; m_st ^= 1; }
#pragma unroll
                for (int i = 0; i < TM; ++i) { hcb[i] = hnb[i]; hcs[i] = hns[i]; }
                hx_set(hnb, hns, m_dx == 2 ? (m_st ^ 1) : m_st, m_dx == 2 ? 0 : m_dx + 1);
            }
            if constexpr (W8) {
                ++t_abs;
                curB = othB; curPar = othPar;
                othB = wbase + (((t_abs + 1) >> 1) & 1) * BN * 128;
                othPar = ((t_abs + 1) & 1) * 64;
            } else {
                t = curB; curB = othB; othB = t;
            }
        }
        }
        __syncthreads();
        }       // (!A8)
    } else {
        load_tiles();
        store_tiles(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tiles();
            compute_tile(buf);
            if (kt + 1 < nk) store_tiles(buf ^ 1);
            __syncthreads();
        }
    }

    if (RF_DBG(p, 1)) {          // timing experiment: no epilogue (keep the accumulators observable)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) t += acc[i][j][0];
        if (t == 123.456f) ((float*)p.out)[0] = t;
        return;
    }
    if constexpr (EPI == 2) {
        // ---- packed staged epilogue (bf16 output).  The chunked fp32 staging of the general path keeps 6 of 8 waves idle while two
        // write 80 KB of LDS with 4-byte stores, four times per tile (15 us of a 256x320 tile's 27 us at K = 320, RF_GEMM_DBG
        // decomposition r02c).  Here bias / timestep vector / activation / GEGLU happen in registers (a lane owns ONE column per
        // 32-column block), neighbouring lanes exchange one value by DPP so that each holds a (col 2c, col 2c+1) bf16 pair of
        // alternating rows, and ALL waves write the whole tile as bf16 [BM][BNo] in one pass (half the LDS bytes: a 256x320 tile is
        // exactly the 160 KB of LDS).  One barrier later every thread streams 16-byte row segments: residual add (fp32),
        // GroupNorm partial sums of the values as stored, 16-byte global stores.  Host guarantees: N % 16 == 0, 16-byte aligned
        // rows, no split-K, no PReLU, per-tile uniform timestep vector.
        bool geglu = false;
        if constexpr (TN % 2 == 0) geglu = p.act == RF_ACT_GEGLU;
        const int BNo = geglu ? BN / 2 : BN;                 // columns of the staged tile
        char* const tile = smem;
        TO* const outp = (TO*)p.out + zb * p.sO;
        const TO* const resp = p.residual ? (const TO*)p.residual + zb * p.sR : nullptr;
        const float* const rvu = p.rowvec ? p.rowvec + (long long)(m0 / p.rows_per_sample) * p.ldv : nullptr;
        const int odd = lane & 1;
        // phase 1: registers -> bf16 tile
        auto stage_block = [&](const float* y, int rowb, int colb) {
            // y[16]: this lane's column of a 32x32 block (rows (r&3) + 8*(r>>2) + 4*lhalf); rowb / colb: block origin in the tile
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float send = odd ? y[r] : y[r + 1];            // what the neighbour lane (col ^ 1) needs
                const float keep = odd ? y[r + 1] : y[r];
                const float got = as_f32((uint32_t)__builtin_amdgcn_mov_dpp((int)as_u32(send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                const uint32_t w = odd ? pack_bf2(got, keep) : pack_bf2(keep, got);       // (col 2c, col 2c+1)
                const int row = rowb + (r & 3) + 8 * (r >> 2) + 4 * lhalf + odd;        // even lanes: row of r, odd lanes: row of r+1
                *(uint32_t*)(tile + (row * BNo + colb + (lrow & ~1)) * 2) = w;
            }
        };
        if (geglu) {
            if constexpr (TN % 2 == 0) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int cv = n0 + (wn * TN + j) * 32 + lrow;
                    const float bv = (p.bias && cv < p.N) ? p.bias[cv] : 0.f, bg = (p.bias && cv + 32 < p.N) ? p.bias[cv + 32] : 0.f;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float y[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) y[r] = (acc[i][j][r] * p.alpha + bv) * gelu_for<T>(acc[i][j + 1][r] * p.alpha + bg);
                        stage_block(y, (wm * TM + i) * 32, ((wn * TN + j) >> 1) * 32);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + lrow;
                float cadd = 0.f;
                if (col < p.N) {
                    if (p.bias) cadd = p.bias[col];
                    if (rvu) cadd += rvu[col];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float y[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = acc[i][j][r] * p.alpha + cadd;       // (host: act is NONE or GEGLU on this path --
                                                                                            // an unrolled 5-way activation chain per value
                                                                                            // is 24 k instructions, beyond the I-cache)
                    stage_block(y, (wm * TM + i) * 32, (wn * TN + j) * 32);
                }
            }
        }
        lds_barrier();
        // phase 2: thread -> one fixed 8-column segment, every EROWS-th row
        const int VPR8 = BNo / 8;                        // 16-byte segments per row
        const int EROWS = NT / VPR8;
        const int cs = tid % VPR8, er = tid / VPR8;
        const int ncol0 = geglu ? (n0 >> 1) : n0, Nout = geglu ? (p.N >> 1) : p.N;
        const int cl = cs * 8, col = ncol0 + cl;
        const bool t_on = er < EROWS && col < Nout;
        const bool gn_on = p.gn_rows > 0;
        float gsum[8], gsq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gsum[e] = gsq[e] = 0.f;
        constexpr int U = 4;                              // rows of residual loads in flight
        for (int rl0 = er; rl0 < BM; rl0 += U * EROWS) {
            u32x4_t rq[U], sv[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = rl0 + u * EROWS, row = m0 + rl;
                ok[u] = t_on && rl < BM && row < p.M;
                if (ok[u]) {
                    if (resp) rq[u] = *(const u32x4_t*)(resp + (long long)row * p.ldr + col);
                    sv[u] = *(const u32x4_t*)(tile + (rl * BNo + cl) * 2);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int row = m0 + rl0 + u * EROWS;
                u32x4_t w = sv[u];
                if (resp) {
                    float a[8], b[8];
                    unpack16<bf16_t>(sv[u], a);
                    unpack16<bf16_t>(rq[u], b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += b[e];
                    w = pack16<bf16_t>(a);
                }
                if (gn_on) {
                    float a[8];
                    unpack16<bf16_t>(w, a);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gsum[e] += a[e]; gsq[e] += a[e] * a[e]; }
                }
                if (RF_DBG(p, 4)) {          // experiment: two 8-byte stores instead of one 16-byte store
                    u32x2_t* d2 = (u32x2_t*)(outp + (long long)row * p.ldo + col);
                    d2[0] = u32x2_t{w[0], w[1]};
                    d2[1] = u32x2_t{w[2], w[3]};
                } else if (!(RF_DBG(p, 8))) {
                    *(u32x4_t*)(outp + (long long)row * p.ldo + col) = w;
                }
            }
        }
        if (gn_on) {
            // column sums of the tile: [EROWS][BNo] per-thread partials -> 32 group sums per consumer -> one chunk slot (fp64)
            lds_barrier();                                 // every thread is done with the staged tile
            float* const csum = (float*)smem;              // [2][EROWS][BNo]
            if (er < EROWS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    csum[er * BNo + cl + e] = gsum[e];
                    csum[(EROWS + er) * BNo + cl + e] = gsq[e];
                }
            }
            lds_barrier();
            if (tid < 64) {
                const int c = tid >> 5, g = tid & 31;
                if (p.gn_part[c]) {
                    const int cpg = p.gn_cpg[c], base = p.gn_coff[c] + n0;            // consumer channel of local column 0
                    const int lo = max(0, g * cpg - base), hi = min(min(BN, p.N - n0), (g + 1) * cpg - base);
                    double sa = 0.0, sq = 0.0;
                    for (int k = lo; k < hi; ++k)
                        for (int r = 0; r < EROWS; ++r) { sa += (double)csum[r * BNo + k]; sq += (double)csum[(EROWS + r) * BNo + k]; }
                    const int b = m0 / p.gn_rows, mt = (m0 - b * p.gn_rows) / BM;
                    double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + mt * p.tiles_n + tile_n) * 32 + g) * 2;
                    o[0] = sa;
                    o[1] = sq;
                }
            }
        }
        return;
    }
    if constexpr (EPI == 1) {
        // ---- direct epilogue: lane (row lrow of its 32-row block, half lhalf) holds columns 16*lhalf .. 16*lhalf+15 of every
        // 32-column block of its wave tile.  Host guarantees (launch_typed): N % 16 == 0, 16-byte aligned rows of out / residual /
        // bias / rowvec, no PReLU, rowvec uniform per tile, fused GroupNorm statistics only with act NONE.
        if (p.splitk > 1) {
            // split-K partial sums in FRAGMENT order: slab [z][tile] = [wave][i][j][q][lane] x 16 bytes, so that every store instruction of a
            // wave writes one contiguous KiB (direct fp32 ROW segments -- 64 scattered 16-byte pieces per instruction -- measured slower than
            // the staged rows, 61.7 vs 56.3 us; this form needs neither LDS nor a barrier).  splitk_reduce_frag_kernel reads them back the same way.
            f32x4_t* const slab = (f32x4_t*)(p.ws + ((long long)blockIdx.z * (p.tiles_m * p.tiles_n) + (tile_m * p.tiles_n + tile_n)) * (long long)(BM * BN)) + lane;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4_t* const dst = slab + ((wave * TM + i) * TN + j) * 256;
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q * 64] = f32x4_t{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
            return;
        }
        TO* const outp = (TO*)p.out + zb * p.sO;
        const TO* const resp = p.residual ? (const TO*)p.residual + zb * p.sR : nullptr;
        constexpr int OV = sizeof(TO) == 2 ? 2 : 4;      // 16-byte vectors per 16 output values
        auto store16 = [&](TO* dst, const float* v) {
            if (RF_DBG(p, 8)) return;                                                          // experiment: no global stores
            if (RF_DBG(p, 16)) dst = (TO*)p.out + (((dst - (TO*)p.out) * (long long)sizeof(TO)) & 0xFFFFF) / (long long)sizeof(TO);   // experiment: all stores into 1 MB
            if constexpr (sizeof(TO) == 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4_t w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = pack2<TO>(v[8 * h + 2 * e], v[8 * h + 2 * e + 1]);
                    st16_out((u32x4_t*)dst + h, w);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 4; ++h) st16_out((u32x4_t*)dst + h, u32x4_t{as_u32(v[4 * h]), as_u32(v[4 * h + 1]), as_u32(v[4 * h + 2]), as_u32(v[4 * h + 3])});
            }
        };
        // column constants -> LDS (the main loop ended on a barrier: the operand stages are dead)
        float* const colc_lds = (float*)smem;
        float* const colu_lds = (float*)(smem + 12288);            // (behind the GroupNorm column sums at +2048 .. +12288)
        constexpr bool ln_c = LNF == 2;                             // this launch consumes a LayerNorm-ed operand: row affine in the epilogue
        if (tid < BN) {
            colc_lds[tid] = colc;
            if (ln_c) colu_lds[tid] = colu;
        }
        // row statistics of this lane's TM rows from the producer's per-stripe (mean, M2) records: equal counts per part, Chan's combination
        float ln_rs[TM], ln_rm[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { ln_rs[i] = 1.f; ln_rm[i] = 0.f; }
        if constexpr (ln_c) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + (wm * TM + i) * 32 + lrow;
                if (row < p.M) {
                    const f32x2_t* const sp = (const f32x2_t*)p.ln_in + (long long)row * p.ln_in_parts;
                    float mean = 0.f;
                    for (int q = 0; q < p.ln_in_parts; ++q) mean += sp[q][0];
                    mean /= (float)p.ln_in_parts;
                    float m2 = 0.f;
                    for (int q = 0; q < p.ln_in_parts; ++q) { const float dq = sp[q][0] - mean; m2 += sp[q][1] + (float)p.ln_in_cols * dq * dq; }
                    const float rs = 1.0f / sqrtf(m2 / (float)(p.ln_in_cols * p.ln_in_parts) + p.ln_eps);
                    ln_rs[i] = rs;
                    ln_rm[i] = rs * mean;
                }
            }
        }
        lds_barrier();
        bool geglu = false;
        if constexpr (TN % 2 == 0) geglu = p.act == RF_ACT_GEGLU;
        if (geglu) {
            if constexpr (TN % 2 == 0) {
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int cv = n0 + (wn * TN + j) * 32 + lhalf * 16;           // value columns (packed order); gate = +32
                    const int ocol = (n0 >> 1) + ((wn * TN + j) >> 1) * 32 + lhalf * 16;
                    const bool cok = cv + 32 < p.N;
                    float bv[16], bg[16], uv[16], ug[16];          // (uv / ug: LayerNorm consumers only -- untouched and eliminated otherwise)
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const f32x4_t a = ((const f32x4_t*)(colc_lds + (cv - n0)))[h], g = ((const f32x4_t*)(colc_lds + (cv - n0) + 32))[h];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { bv[4 * h + e] = a[e]; bg[4 * h + e] = g[e]; }
                        if constexpr (ln_c) {
                            const f32x4_t ua = ((const f32x4_t*)(colu_lds + (cv - n0)))[h], ugg = ((const f32x4_t*)(colu_lds + (cv - n0) + 32))[h];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { uv[4 * h + e] = ua[e]; ug[4 * h + e] = ugg[e]; }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = m0 + (wm * TM + i) * 32 + lrow;
                        if (row < p.M && cok) {
                            float v[16];
                            if constexpr (ln_c) {          // LayerNorm folded in: rstd (alpha acc - mean u) + b'  (gamma is in W, W beta in the bias)
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float a_ = __builtin_fmaf(acc[i][j][r] * p.alpha, ln_rs[i], __builtin_fmaf(-ln_rm[i], uv[r], bv[r]));
                                    const float g_ = __builtin_fmaf(acc[i][j + 1][r] * p.alpha, ln_rs[i], __builtin_fmaf(-ln_rm[i], ug[r], bg[r]));
                                    v[r] = a_ * gelu_for<T>(g_);
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    const float a_ = acc[i][j][r] * p.alpha + bv[r], g_ = acc[i][j + 1][r] * p.alpha + bg[r];
                                    v[r] = a_ * gelu_for<T>(g_);
                                }
                            }
                            if (!ln_c && resp) {
                                const u32x4_t* rp = (const u32x4_t*)(resp + (long long)row * p.ldr + ocol);
#pragma unroll
                                for (int h = 0; h < OV; ++h) {
                                    float f[16 / OV];
                                    unpack16<TO>(rp[h], f);
#pragma unroll
                                    for (int e = 0; e < 16 / OV; ++e) v[h * (16 / OV) + e] += f[e];
                                }
                            }
                            if constexpr (A8) {
                                if (p.oscale) {
                                    // fp8 output for the next fp8 x fp8 GEMM (ff.net.2): this lane holds columns [16 h, 16 h + 16) of a 32-column
                                    // block of its row, lane ^ 32 the other half: block amax across the two, one E8M0 code, 16 bytes per lane
                                    float m = 0.f;
#pragma unroll
                                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(v[r]));
                                    const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(m), as_u32(m), false, false);
                                    const int code = e8m0_for_amax(fmaxf(as_f32(sw[0]), as_f32(sw[1])));
                                    const u32x2_t q0 = quant8_fp8(v, code), q1 = quant8_fp8(v + 8, code);
                                    fp8_t* const qrow = (fp8_t*)p.out + (long long)row * p.ldo;
                                    *(u32x4_t*)(qrow + ocol) = u32x4_t{q0[0], q0[1], q1[0], q1[1]};
                                    if (lhalf == 0) ((fp8_t*)p.oscale)[(long long)row * p.os_ld + (ocol >> 5)] = (fp8_t)code;
                                    continue;
                                }
                            }
                            store16(outp + (long long)row * p.ldo + ocol, v);
                        }
                    }
                }
            }
        } else {
            // One 32x32 block at a time; the column constants come from LDS, the residual segments through a ring of PFD blocks in
            // flight.  sched_barrier(0) after every block keeps hipcc from hoisting every load of the unrolled nest to the top (the 2x5
            // wave tile then spills ~170 registers per lane).
            constexpr int NBLK = TM * TN;
#ifndef RF_EPI_PFD
#define RF_EPI_PFD 4
#endif
            constexpr int PFD = sizeof(TO) == 2 ? (NBLK < RF_EPI_PFD ? NBLK : RF_EPI_PFD) : 2;      // residual blocks in flight (8 / 16 registers each): the
                                                                                  // fragment registers of the main loop are free now
            u32x4_t rq[PFD][OV];
            auto load_res = [&](int blk, u32x4_t* r) {
                const int i = blk % TM, j = blk / TM;
                const int col = n0 + (wn * TN + j) * 32 + lhalf * 16;
                const int row = m0 + (wm * TM + i) * 32 + lrow;
                if (!ln_c && resp && row < p.M && col < p.N) {
                    const u32x4_t* rp = (const u32x4_t*)(resp + (long long)row * p.ldr + col);
#pragma unroll
                    for (int h = 0; h < OV; ++h) r[h] = rp[h];
                }
            };
#pragma unroll
            for (int b = 0; b < PFD; ++b) load_res(b, rq[b]);
            // Fused GroupNorm statistics (gn_rows > 0): per 32-column block the lane sums its 16 columns over its TM rows (values as stored),
            // a 5-stage butterfly over the 32 lanes of the half-wave leaves ONE column total per lane (lanes 0-15: sums, 16-31: sums of
            // squares), the wave rows meet in LDS and 64 threads form the 32 group sums per consumer in fp64 -- same slots, same
            // consumers as the staged epilogue.
            const bool gn_on = p.gn_rows > 0;
            constexpr bool ln_p = LNF == 1;                      // this launch also emits the LayerNorm row statistics of its output
            const bool keep_on = gn_on || ln_p;                  // the values as stored replace the (dead) accumulators
            float* const gcs = (float*)(smem + 2048);            // [2][WM][BN] column sums / sums of squares per wave row
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + lhalf * 16;
                const bool cok = col < p.N;
                f32x4_t cb[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) cb[h] = ((const f32x4_t*)(colc_lds + (col - n0)))[h];
                // (LayerNorm consumer: rstd (alpha acc - mean u) + b' -- the row factors differ per i, the two column vectors are combined per element)
                f32x4_t cu[4];          // (LayerNorm consumers only)
                if constexpr (ln_c) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) cu[h] = ((const f32x4_t*)(colu_lds + (col - n0)))[h];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int blk = j * TM + i;
                    const int row = m0 + (wm * TM + i) * 32 + lrow;
                    const u32x4_t* const myr = rq[blk % PFD];
                    const bool live = row < p.M && cok;
                    if (live) {
                        TO* dst = outp + (long long)row * p.ldo + col;
#pragma unroll
                        for (int h = 0; h < OV; ++h) {
                            constexpr int E = 16 / OV;
                            float f[E], v[E];
                            if (resp) unpack16<TO>(myr[h], f);
                            if constexpr (ln_c) {          // (no residual on a LayerNorm consumer: host)
#pragma unroll
                                for (int e = 0; e < E; ++e)
                                    v[e] = __builtin_fmaf(acc[i][j][h * E + e] * p.alpha, ln_rs[i],
                                                          __builtin_fmaf(-ln_rm[i], cu[(h * E + e) >> 2][(h * E + e) & 3], cb[(h * E + e) >> 2][(h * E + e) & 3]));
                            } else {
#pragma unroll
                                for (int e = 0; e < E; ++e) v[e] = acc[i][j][h * E + e] * p.alpha + cb[(h * E + e) >> 2][(h * E + e) & 3] + (resp ? f[e] : 0.0f);
                            }
                            const u32x4_t w = pack16<TO>(v);
                            if (!(RF_DBG(p, 8))) st16_out((u32x4_t*)dst + h, w);
                            if (keep_on) {
                                float y[E];
                                unpack16<TO>(w, y);              // the values as stored replace the (dead) accumulators: no extra registers
#pragma unroll
                                for (int e = 0; e < E; ++e) acc[i][j][h * E + e] = y[e];
                            }
                        }
                    } else if (keep_on) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                    }
                    if (blk + PFD < NBLK) load_res(blk + PFD, rq[blk % PFD]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float gx[32];                                    // [0, 16): column sums, [16, 32): sums of squares (this lane's rows)
                if (gn_on) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        float a = 0.f, q = 0.f;
#pragma unroll
                        for (int i = 0; i < TM; ++i) { a += acc[i][j][k]; q += acc[i][j][k] * acc[i][j][k]; }
                        gx[k] = a;
                        gx[16 + k] = q;
                    }
                }
                if (gn_on) {
                    halfwave_reduce_scatter32(gx, lrow);
                    // lane lrow now holds the total of value index lrow: column lrow (sums) / column lrow - 16 (squares)
                    gcs[((lrow >> 4) * WM + wm) * BN + (wn * TN + j) * 32 + lhalf * 16 + (lrow & 15)] = gx[0];
                }
            }
            if (gn_on) {
                lds_barrier();
                if (tid < 64) {
                    const int c = tid >> 5, g = tid & 31;
                    if (p.gn_part[c]) {
                        const int cpg = p.gn_cpg[c], base = p.gn_coff[c] + n0;            // consumer channel of local column 0
                        const int lo = max(0, g * cpg - base), hi = min(min(BN, p.N - n0), (g + 1) * cpg - base);
                        double sa = 0.0, sq = 0.0;
                        for (int k = lo; k < hi; ++k)
#pragma unroll
                            for (int r = 0; r < WM; ++r) { sa += (double)gcs[r * BN + k]; sq += (double)gcs[(WM + r) * BN + k]; }
                        const int b = m0 / p.gn_rows, mt = (m0 - b * p.gn_rows) / BM;
                        double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + mt * p.tiles_n + tile_n) * 32 + g) * 2;
                        o[0] = sa;
                        o[1] = sq;
                    }
                }
            }
            if constexpr (ln_p) {
                // LayerNorm statistics of the rows this wave wrote, over its 32 TN columns (values as stored): the lane pair (row, half 0 / 1)
                // holds them all -- two-pass (mean, then squared deviations) in registers, the halves meet through v_permlane32_swap;
                // record = (mean, M2) per row and column stripe, combined by the consumer GEMM's epilogue
                const int part = (n0 + wn * TN * 32) / (TN * 32);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float sm = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) sm += acc[i][j][e];
                    {
                        const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(sm), as_u32(sm), false, false);
                        sm = as_f32(sw[0]) + as_f32(sw[1]);
                    }
                    const float mean = sm / (float)(TN * 32);
                    float sq = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) { const float dq = acc[i][j][e] - mean; sq += dq * dq; }
                    {
                        const auto sw = __builtin_amdgcn_permlane32_swap(as_u32(sq), as_u32(sq), false, false);
                        sq = as_f32(sw[0]) + as_f32(sw[1]);
                    }
                    const int row = m0 + (wm * TM + i) * 32 + lrow;
                    // (a WN = 2 tile whose BN does not divide N has a wave-tile stripe beyond N: no record -- part would be >= ln_out_parts)
                    if (lhalf == 0 && row < p.M && part < p.ln_out_parts) *(f32x2_t*)(p.ln_out + ((long long)row * p.ln_out_parts + part) * 2) = f32x2_t{mean, sq};
                }
            }
        }
        return;
    }

    // ---- epilogue: accumulators -> LDS (fp32 [ER][BN]) -> coalesced 16-byte row segments, in NCH row chunks
    // (the main loop ended on a barrier, so the operand tiles are dead and the LDS can be reused)
    constexpr int NCH = (BM * BN * 4 > 96 * 1024) ? WM : 1;      // big tiles: one chunk per wave row
    constexpr int ER = BM / NCH;
    float* const stage = (float*)smem;
    TO* outp = (TO*)p.out + zb * p.sO;
    const TO* resp = p.residual ? (const TO*)p.residual + zb * p.sR : nullptr;
    const bool partial_out = p.splitk > 1;
    const bool gn_on = p.gn_rows > 0;          // fused GroupNorm statistics of the values this block writes
    float gsum[4] = {0.f, 0.f, 0.f, 0.f}, gsq[4] = {0.f, 0.f, 0.f, 0.f};      // this thread's 4 columns, all of its rows
    for (int ch = 0; ch < NCH; ++ch) {
    if (NCH == 1 || wm == ch) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = ((NCH == 1 ? wm * TM : 0) + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    stage[rl * BN + (wn * TN + j) * 32 + lrow] = acc[i][j][r];
                }
    }
    lds_barrier();           // LDS-only barrier: the previous chunk's global stores may stay in flight
    const int mch = m0 + ch * ER;
    if (partial_out) {            // split-K: raw fp32 partial sums, coalesced; bias / activation / residual happen in the reduce pass
        constexpr int VPRS = BN / 4;
        float* wsz = p.ws + (long long)blockIdx.z * p.M * p.N;
        for (int idx = tid; idx < ER * VPRS; idx += NT) {
            const int rl = idx / VPRS, cl = (idx - rl * VPRS) * 4;
            const int row = mch + rl, col = n0 + cl;
            if (row >= p.M || col >= p.N) continue;
            const f32x4_t a4 = *(const f32x4_t*)(stage + rl * BN + cl);
            float* dst = wsz + (long long)row * p.N + col;
            if ((p.N & 3) == 0) *(f32x4_t*)dst = a4;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (col + e < p.N) dst[e] = a4[e];
            }
        }
    } else if (p.act == RF_ACT_GEGLU) {
        // same thread -> fixed column mapping as the general path below: the value / gate bias vectors are loaded once
        constexpr int OV = BN / 8;                      // output 4-column vectors per row (N/2 columns)
        constexpr int EROWS = NT / OV;
        constexpr int PASSES = (ER + EROWS - 1) / EROWS;
        const int oc = (tid % OV) * 4, er = tid / OV;
        const int lv = (oc >> 5) * 64 + (oc & 31), lg = lv + 32;      // local value / gate columns
        const int ocol = (n0 >> 1) + oc;
        const bool t_on = er < EROWS && n0 + lg < p.N;
        f32x4_t ba = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
        if (t_on && p.bias) {
            if (p.vec_ok) { ba = *(const f32x4_t*)(p.bias + n0 + lv); bg = *(const f32x4_t*)(p.bias + n0 + lg); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ba[e] = p.bias[n0 + lv + e]; bg[e] = p.bias[n0 + lg + e]; }
            }
        }
#pragma unroll 2
        for (int k = 0; k < PASSES; ++k) {
            const int rl = er + k * EROWS, row = mch + rl;
            if (!t_on || rl >= ER || row >= p.M) continue;
            const f32x4_t a4 = *(const f32x4_t*)(stage + rl * BN + lv);
            const f32x4_t g4 = *(const f32x4_t*)(stage + rl * BN + lg);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a_ = a4[e] * p.alpha + ba[e], g_ = g4[e] * p.alpha + bg[e];
                v[e] = a_ * gelu_for<T>(g_);
            }
            TO* dst = outp + (long long)row * p.ldo + ocol;
            if (p.vec_ok) {
                if (resp) {
                    float rr[4];
                    if constexpr (sizeof(TO) == 2) {
                        const u32x2_t q = *(const u32x2_t*)(resp + (long long)row * p.ldr + ocol);
                        const uint32_t q0 = q[0], q1 = q[1];
                        rr[0] = lo16<TO>(q0); rr[1] = hi16<TO>(q0); rr[2] = lo16<TO>(q1); rr[3] = hi16<TO>(q1);
                    } else {
                        const f32x4_t q = *(const f32x4_t*)(resp + (long long)row * p.ldr + ocol);
                        rr[0] = q[0]; rr[1] = q[1]; rr[2] = q[2]; rr[3] = q[3];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rr[e];
                }
                if constexpr (sizeof(TO) == 2) {
                    u32x2_t w; w[0] = pack2<TO>(v[0], v[1]); w[1] = pack2<TO>(v[2], v[3]);
                    *(u32x2_t*)dst = w;
                } else {
                    *(f32x4_t*)dst = f32x4_t{v[0], v[1], v[2], v[3]};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = v[e];
                    if (resp) y += load_out<TO>(resp + (long long)row * p.ldr + ocol + e);
                    store_out<TO>(dst + e, y);
                }
            }
        }
    } else {
        // thread -> one fixed 4-column segment and every EROWS-th row of the chunk: bias / timestep vector are loaded once, and
        // the residual loads of U rows are issued together BEFORE the first one is consumed.  (One load -> wait -> store per
        // row serialises the whole HBM / L2 latency per row: that alone made the K = C projection layers 2x slower in situ.)
        constexpr int VPR = BN / 4;
        constexpr int EROWS = NT / VPR;
        constexpr int PASSES = (ER + EROWS - 1) / EROWS;
        constexpr int U = sizeof(TO) == 2 ? 4 : 2;      // rows of residual loads in flight (fp32 rows cost 4 registers each)
        const int cs = tid % VPR, er = tid / VPR;
        const int cl = cs * 4, col = n0 + cl;
        const bool t_on = er < EROWS && col < p.N;
        const bool full = p.vec_ok && col + 3 < p.N;
        float cadd[4] = {0.f, 0.f, 0.f, 0.f};           // bias (+ per-sample vector when the chunk lies inside one sample)
        const bool rv_uniform = p.rowvec && (p.rows_per_sample % ER == 0);
        if (t_on) {
            const float* rvu = rv_uniform ? p.rowvec + (long long)(mch / p.rows_per_sample) * p.ldv : nullptr;
            if (full) {
                if (p.bias) { const f32x4_t b4 = *(const f32x4_t*)(p.bias + col); cadd[0] = b4[0]; cadd[1] = b4[1]; cadd[2] = b4[2]; cadd[3] = b4[3]; }
                if (rvu) { const f32x4_t r4 = *(const f32x4_t*)(rvu + col); cadd[0] += r4[0]; cadd[1] += r4[1]; cadd[2] += r4[2]; cadd[3] += r4[3]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < p.N) { if (p.bias) cadd[e] = p.bias[col + e]; if (rvu) cadd[e] += rvu[col + e]; }
            }
        }
        typedef typename std::conditional<sizeof(TO) == 2, u32x2_t, f32x4_t>::type resv_t;
#pragma unroll 1
        for (int k0 = 0; k0 < PASSES; k0 += U) {
            resv_t rq[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rl = er + (k0 + u) * EROWS, row = mch + rl;
                ok[u] = t_on && (k0 + u) < PASSES && rl < ER && row < p.M;
                if (ok[u] && resp && full) rq[u] = *(const resv_t*)(resp + (long long)row * p.ldr + col);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int rl = er + (k0 + u) * EROWS, row = mch + rl;
                const f32x4_t a4 = *(const f32x4_t*)(stage + rl * BN + cl);
                float v[4] = {a4[0] * p.alpha + cadd[0], a4[1] * p.alpha + cadd[1], a4[2] * p.alpha + cadd[2], a4[3] * p.alpha + cadd[3]};
                if (p.rowvec && !rv_uniform) {
                    const float* rv = p.rowvec + (long long)(row / p.rows_per_sample) * p.ldv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (col + e < p.N) v[e] += rv[col + e];
                }
                if (p.act != RF_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = v[e];
                        if (p.act == RF_ACT_SILU) y = silu_exact(y);
                        else if (p.act == RF_ACT_QUICK_GELU) y = quick_gelu(y);
                        else if (p.act == RF_ACT_GELU) y = gelu_erf(y);
                        else if (p.act == RF_ACT_RELU) y = fmaxf(y, 0.0f);
                        else if (p.act == RF_ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
                        else if (p.act == RF_ACT_PRELU) y = y >= 0.0f ? y : y * p.act_vec[min(col + e, p.N - 1)];
                        v[e] = y;
                    }
                }
                TO* dst = outp + (long long)row * p.ldo + col;
                if (full) {
                    if (resp) {
                        if constexpr (sizeof(TO) == 2) {
                            const uint32_t q0 = rq[u][0], q1 = rq[u][1];
                            v[0] += lo16<TO>(q0); v[1] += hi16<TO>(q0); v[2] += lo16<TO>(q1); v[3] += hi16<TO>(q1);
                        } else {
                            v[0] += rq[u][0]; v[1] += rq[u][1]; v[2] += rq[u][2]; v[3] += rq[u][3];
                        }
                    }
                    if constexpr (sizeof(TO) == 2) {
                        u32x2_t w; w[0] = pack2<TO>(v[0], v[1]); w[1] = pack2<TO>(v[2], v[3]);
                        *(u32x2_t*)dst = w;
                        if (gn_on) {      // statistics of the values AS STORED (what the apply pass and the unfused statistics pass read)
                            const uint32_t w0 = w[0], w1 = w[1];
                            v[0] = lo16<TO>(w0); v[1] = hi16<TO>(w0); v[2] = lo16<TO>(w1); v[3] = hi16<TO>(w1);
                        }
                    }
                    if (gn_on) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { gsum[e] += v[e]; gsq[e] += v[e] * v[e]; }
                    }
                    if constexpr (sizeof(TO) == 2) {
                    } else {
                        *(f32x4_t*)dst = f32x4_t{v[0], v[1], v[2], v[3]};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < p.N) {
                            float y = v[e];
                            if (resp) y += load_out<TO>(resp + (long long)row * p.ldr + col + e);
                            store_out<TO>(dst + e, y);
                            if (gn_on) {
                                float ys = y;
                                if constexpr (sizeof(TO) == 2) ys = round16<TO>(y);
                                gsum[e] += ys; gsq[e] += ys * ys;
                            }
                        }
                }
            }
        }
    }
    if (NCH > 1) lds_barrier();
    }
    if (gn_on) {
        // column sums of the tile: [EROWS][BN] per-thread partials -> 32 group sums per consumer -> one chunk slot (fp64)
        constexpr int VPR = BN / 4, EROWS = NT / VPR;
        float* const cs = stage;                       // [2][EROWS][BN]
        if (NCH == 1) lds_barrier();                   // every thread is done with the staged tile
        const int cs_ = tid % VPR, er = tid / VPR;
        if (er < EROWS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cs[er * BN + cs_ * 4 + e] = gsum[e];
                cs[(EROWS + er) * BN + cs_ * 4 + e] = gsq[e];
            }
        }
        lds_barrier();
        if (tid < 64) {
            const int c = tid >> 5, g = tid & 31;
            if (p.gn_part[c]) {
                const int cpg = p.gn_cpg[c], base = p.gn_coff[c] + n0;            // consumer channel of local column 0
                const int lo = max(0, g * cpg - base), hi = min(min(BN, p.N - n0), (g + 1) * cpg - base);
                double sa = 0.0, sq = 0.0;
                for (int k = lo; k < hi; ++k)
#pragma unroll
                    for (int r = 0; r < EROWS; ++r) { sa += (double)cs[r * BN + k]; sq += (double)cs[(EROWS + r) * BN + k]; }
                const int b = m0 / p.gn_rows, mt = (m0 - b * p.gn_rows) / BM;
                double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + mt * p.tiles_n + tile_n) * 32 + g) * 2;
                o[0] = sa;
                o[1] = sq;
            }
        }
    }
}

// split-K second pass: out = act(alpha * sum_z ws[z] + bias + rowvec) + residual.  One block = SK_ROWS x SK_COLS of the output,
// thread -> one fixed 4-column segment and every 4th row, so the fused GroupNorm statistics reduce exactly as in the main epilogue
// (chunk slot = gn_slot + (row tile within the sample) * column tiles + column tile).
constexpr int SK_ROWS = 32, SK_COLS = 256;
// Rows per reduce block: 32, or 8 when 32-row blocks would leave most of the chip idle (M = 1024 x N = 1280 is 160 blocks of 32 rows:
// 21 us for 39 MB of partial sums = 1.9 TB/s, 34 launches per step).
static inline int sk_rows_for(int M, int N) { return ((long long)((M + 31) / 32) * ((N + SK_COLS - 1) / SK_COLS) >= 512) ? 32 : 8; }
template <typename TO, int SKR>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const int tid = threadIdx.x;
    const int cl = (tid & 63) * 4, rp = tid >> 6;
    const int n0 = blockIdx.x * SK_COLS, m0 = blockIdx.y * SKR;
    const int col = n0 + cl;
    const bool vec = (p.N & 3) == 0 && p.vec_ok;
    const bool gn_on = p.gn_rows > 0;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f}, gsq[4] = {0.f, 0.f, 0.f, 0.f};
    TO* outp = (TO*)p.out;
    const TO* resp = (const TO*)p.residual;
    if (col < p.N) {
        float cadd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (p.bias && col + e < p.N) cadd[e] = p.bias[col + e];
        constexpr int U = SKR / 4 < 4 ? SKR / 4 : 4;          // rows in flight: the partial-sum loads of U rows are issued together (one row at a time
                                      // serialises the L2 / HBM latency SK_ROWS / 4 times per thread)
#pragma unroll 1
        for (int k0 = 0; k0 < SKR / 4; k0 += U) {
            float v[U][4];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = m0 + rp + 4 * (k0 + u) < p.M;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = 0.f;
            }
            // residual and per-sample vector of the U rows: issued with (not after) the partial-sum loads
            typedef typename std::conditional<sizeof(TO) == 2, u32x2_t, f32x4_t>::type resv_t;
            const bool full = vec && col + 3 < p.N;
            resv_t rq[U];
            f32x4_t rv4[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = m0 + rp + 4 * (k0 + u);
                rv4[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                if (ok[u] && full) {
                    if (resp) rq[u] = *(const resv_t*)(resp + (long long)row * p.ldr + col);
                    if (p.rowvec) rv4[u] = *(const f32x4_t*)(p.rowvec + (long long)(row / p.rows_per_sample) * p.ldv + col);
                }
            }
            for (int z = 0; z < p.splitk; ++z) {
                f32x4_t a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float* src = p.ws + ((long long)z * p.M + (m0 + rp + 4 * (k0 + u))) * p.N + col;
                    a[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (ok[u]) {
                        if (vec) a[u] = *(const f32x4_t*)src;
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (col + e < p.N) a[u][e] = src[e];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { v[u][0] += a[u][0]; v[u][1] += a[u][1]; v[u][2] += a[u][2]; v[u][3] += a[u][3]; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                const int row = m0 + rp + 4 * (k0 + u);
                const float* rv = (p.rowvec && !full) ? p.rowvec + (long long)(row / p.rows_per_sample) * p.ldv : nullptr;
                float y4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {0.f, 0.f, 0.f, 0.f};
                if (resp && full) {
                    if constexpr (sizeof(TO) == 2) {
                        const uint32_t q0 = rq[u][0], q1 = rq[u][1];
                        r4[0] = lo16<TO>(q0); r4[1] = hi16<TO>(q0); r4[2] = lo16<TO>(q1); r4[3] = hi16<TO>(q1);
                    } else {
                        r4[0] = rq[u][0]; r4[1] = rq[u][1]; r4[2] = rq[u][2]; r4[3] = rq[u][3];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = col + e;
                    if (c >= p.N) break;
                    float y = v[u][e] * p.alpha + cadd[e] + rv4[u][e];
                    if (rv) y += rv[c];
                    if (p.act == RF_ACT_SILU) y = silu_exact(y);
                    else if (p.act == RF_ACT_QUICK_GELU) y = quick_gelu(y);
                    else if (p.act == RF_ACT_GELU) y = gelu_erf(y);
                    else if (p.act == RF_ACT_RELU) y = fmaxf(y, 0.0f);
                    else if (p.act == RF_ACT_SIGMOID) y = 1.0f / (1.0f + expf(-y));
                    else if (p.act == RF_ACT_PRELU) y = y >= 0.0f ? y : y * p.act_vec[c];
                    if (resp) y += full ? r4[e] : load_out<TO>(resp + (long long)row * p.ldr + c);
                    if (gn_on) {          // statistics of the value as stored
                        float ys = y;
                        if constexpr (sizeof(TO) == 2) ys = round16<TO>(y);
                        gsum[e] += ys; gsq[e] += ys * ys;
                    }
                    y4[e] = y;
                    if (!full) store_out<TO>(outp + (long long)row * p.ldo + c, y);
                }
                if (full) {
                    TO* dst = outp + (long long)row * p.ldo + col;
                    if constexpr (sizeof(TO) == 2) {
                        u32x2_t w; w[0] = pack2<TO>(y4[0], y4[1]); w[1] = pack2<TO>(y4[2], y4[3]);
                        *(u32x2_t*)dst = w;
                    } else {
                        *(f32x4_t*)dst = f32x4_t{y4[0], y4[1], y4[2], y4[3]};
                    }
                }
            }
        }
    }
    if (gn_on) {
        __shared__ float cs[2][4][SK_COLS];
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[0][rp][cl + e] = gsum[e]; cs[1][rp][cl + e] = gsq[e]; }
        __syncthreads();
        if (tid < 64) {
            const int c = tid >> 5, g = tid & 31;
            if (p.gn_part[c]) {
                const int cpg = p.gn_cpg[c], base = p.gn_coff[c] + n0;
                const int lo = max(0, g * cpg - base), hi = min(min(SK_COLS, p.N - n0), (g + 1) * cpg - base);
                double sa = 0.0, sq = 0.0;
                for (int k = lo; k < hi; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sa += (double)cs[0][r][k]; sq += (double)cs[1][r][k]; }
                const int b = m0 / p.gn_rows, mt = (m0 - b * p.gn_rows) / SKR;
                double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + mt * (int)gridDim.x + (int)blockIdx.x) * 32 + g) * 2;
                o[0] = sa;
                o[1] = sq;
            }
        }
    }
}

// Split-K second pass over FRAGMENT-ordered slabs (written by the direct-epilogue kernels, see there): one block = one 32-row stripe of one
// wave tile = TN waves, wave j owns the 32x32 block j of the stripe; lane (row lrow, half lhalf) sums its 16 columns over the z slices with
// contiguous-KiB loads, then does what the direct epilogue does: alpha, bias + per-sample vector, residual, 16-byte stores, and the fused
// GroupNorm statistics of the values as stored (slot = 32 rows x 32 TN columns).  act = NONE only (host).
template <typename TO, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * TN) void splitk_reduce_frag_kernel(const GemmParams p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int tile = blockIdx.x, tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int w = blockIdx.y / TM, i = blockIdx.y - w * TM, wm = w / WN, wn = w - wm * WN;
    const int row0 = tile_m * BM + (wm * TM + i) * 32, row = row0 + lrow;
    const int coln0 = tile_n * BN + wn * TN * 32;                    // first column of the stripe
    const int col = coln0 + j * 32 + lhalf * 16;
    const long long zstride = (long long)p.tiles_m * p.tiles_n * (BM * BN / 4);          // (16-byte units)
    const f32x4_t* src = (const f32x4_t*)p.ws + (long long)tile * (BM * BN / 4) + ((w * TM + i) * TN + j) * 256 + lane;
    f32x4_t a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = src[q * 64];
    for (int z = 1; z < p.splitk; ++z) {
        src += zstride;
        f32x4_t b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = src[q * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] += b[q];
    }
    const bool live = row < p.M && col < p.N;
    const bool gn_on = p.gn_rows > 0;
    constexpr int OV = sizeof(TO) == 2 ? 2 : 4, E = 16 / OV;
    float y[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = 0.f;
    if (live) {
        TO* const dst = (TO*)p.out + (long long)row * p.ldo + col;
        const TO* const resp = p.residual ? (const TO*)p.residual + (long long)row * p.ldr + col : nullptr;
        const float* const rv = p.rowvec ? p.rowvec + (long long)(row / p.rows_per_sample) * p.ldv + col : nullptr;
        u32x4_t rq[OV];
        if (resp) {
#pragma unroll
            for (int h = 0; h < OV; ++h) rq[h] = ((const u32x4_t*)resp)[h];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4_t c = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) c = ((const f32x4_t*)(p.bias + col))[q];
            if (rv) c += ((const f32x4_t*)rv)[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * q + e] = a[q][e] * p.alpha + c[e];
        }
#pragma unroll
        for (int h = 0; h < OV; ++h) {
            float f[E], v[E];
            if (resp) unpack16<TO>(rq[h], f);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = y[h * E + e] + (resp ? f[e] : 0.0f);
            const u32x4_t wv = pack16<TO>(v);
            ((u32x4_t*)dst)[h] = wv;
            if (gn_on) {
                unpack16<TO>(wv, f);              // statistics of the values as stored
#pragma unroll
                for (int e = 0; e < E; ++e) y[h * E + e] = f[e];
            }
        }
    }
    if (gn_on) {
        __shared__ float gcs[2][32 * TN];             // column sums / sums of squares of the stripe
        float gx[32];
#pragma unroll
        for (int k = 0; k < 16; ++k) { gx[k] = y[k]; gx[16 + k] = y[k] * y[k]; }
        halfwave_reduce_scatter32(gx, lrow);          // (as in the direct epilogue)
        gcs[lrow >> 4][j * 32 + lhalf * 16 + (lrow & 15)] = gx[0];
        __syncthreads();
        if (tid < 64) {
            const int c = tid >> 5, g = tid & 31;
            // (a stripe of a ragged tile that lies beyond M or N has no slot: M < BM with split-K -- CFG off at the 8x8 level -- would otherwise
            //  write past the end of `partial`; slots are numbered by the GLOBAL 32 TN-column stripe, which is what the host sizes them by)
            if (p.gn_part[c] && row0 < p.M && coln0 < p.N) {
                const int cpg = p.gn_cpg[c], base = p.gn_coff[c] + coln0;            // consumer channel of the stripe's first column
                const int lo = max(0, g * cpg - base), hi = min(min(32 * TN, p.N - coln0), (g + 1) * cpg - base);
                double sa = 0.0, sq = 0.0;
                for (int k = lo; k < hi; ++k) { sa += (double)gcs[0][k]; sq += (double)gcs[1][k]; }
                const int b = row0 / p.gn_rows, mt = (row0 - b * p.gn_rows) / 32;
                const int nstripe = (p.N + 32 * TN - 1) / (32 * TN);
                double* o = p.gn_part[c] + (((long long)b * p.gn_nch[c] + p.gn_slot[c] + mt * nstripe + coln0 / (32 * TN)) * 32 + g) * 2;
                o[0] = sa;
                o[1] = sq;
            }
        }
    }
}

#ifdef RF_KERNEL_ONLY          // (diagnostics: tools/kernel_regs.sh compiles single instantiations of the kernels above)
}  // namespace rf
#else
// Split-K factor for a launch of `tiles` output tiles, by a two-term cost model: GEMM time at ~600 TFLOP/s stretched by the
// fraction of the 256 CUs left idle, plus the fp32 partial-sum traffic (write + re-read of sk * M * N floats at ~4 TB/s).
static int pick_splitk(const rf_conv_gemm_desc* d, const GemmParams& p, long long tiles, int bk) {
    const int nk = (p.K + bk - 1) / bk;
    if (!d->workspace || d->batch != 1 || d->act == RF_ACT_GEGLU || tiles >= 200 || nk < 16) return 1;
    int maxsk = nk / 8;
    if (maxsk > 16) maxsk = 16;
    const double t_gemm = 2.0 * p.M * p.N * (double)p.K / 6e14;
    int best = 1;
    double best_cost = 1e30;
    for (int sk = 1; sk <= maxsk; ++sk) {
        if ((long long)sk * p.M * p.N * 4 > d->workspace_bytes) break;
        // (more blocks than CUs: whole rounds -- 288 blocks of one block per CU are two rounds with the second 12 % full)
        const long long blocks = tiles * sk;
        const double fill = (double)blocks / (256.0 * (double)((blocks + 255) / 256));
        const double cost = t_gemm / fill + (sk > 1 ? (double)sk * p.M * p.N * 8.0 / 4e12 + 3e-6 : 0.0);
        if (cost < best_cost) { best_cost = cost; best = sk; }
    }
    return best;
}

template <typename T, typename TO, int WM, int WN, int TM, int TN, bool W8 = false>
static int launch_cfg(const rf_conv_gemm_desc* d, GemmParams& p, bool conv, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int NST = 2;            // LDS stages of the direct-to-LDS main loop (3 stages of the big tiles at 1 block/CU measured slower)
    // The 4-wave 128x160 tile also exists with a ring of 4 stages (147 KB: one block per CU), for grids of at most one block per CU
    constexpr bool DEEP_OK = WM * WN == 4 && TM * TN == 5 && !W8 && !std::is_same<T, fp8_t>::value;
    constexpr int NSTD = DEEP_OK ? 4 : 2;
    constexpr int NCH = (BM * BN * 4 > 96 * 1024) ? WM : 1;
    constexpr bool A8 = std::is_same<T, fp8_t>::value;
    constexpr int smem_ml = NST * (BM + BN) * 128 + (A8 ? 4096 : 0), smem_ep = (BM / NCH) * BN * 4;
    constexpr int smem = smem_ml > smem_ep ? smem_ml : smem_ep;
    // HX (row-extended A tiles for 3x3 stride-1 convolutions, korder 2): bf16 -> bf16 only; the stage of BM + BM / 4 rows must fit the 160 KB of LDS
    constexpr int RPP_ = WM * WN * 8, AXR_ = ((BM * 5 / 4 + RPP_ - 1) / RPP_) * RPP_;
    constexpr int smem_hx_ml = NST * (AXR_ + BN) * 128, smem_hx = smem_hx_ml > smem_ep ? smem_hx_ml : smem_ep;
    constexpr bool HX_OK = is16<T>::value && std::is_same<TO, T>::value && !W8 && !A8 && smem_hx <= 160 * 1024 && AXR_ / RPP_ <= 8;
    const bool hx = p.korder == 2;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    RF_CHECK(!p.w_ps || (p.rows_per_sample > 0 && p.rows_per_sample % BM == 0 && p.M % p.rows_per_sample == 0),
             "rf_conv_gemm: per-sample weights need rows_per_sample (%d) to be a multiple of the %d-row tile", p.rows_per_sample, BM);
    // split-K for launches that cannot fill the chip: each z-slice owns a K range, partial sums go through the caller's workspace
    p.splitk = pick_splitk(d, p, (long long)p.tiles_m * p.tiles_n, 128 / (int)sizeof(T));
    if (p.splitk > 1) p.ws = (float*)d->workspace;
    {
        // Tile order inside each XCD's contiguous run of tiles (8 XCDs, one 4 MB L2 each, 32 CUs): at any time the XCD works on ~`conc` consecutive
        // tiles of its run (x the K slices of split-K, which share nothing).  Estimated L2-miss bytes of a run ordered in pm x pn patches:
        // per super-row of pm tile rows the patches sweep along N -- the pm A panels stay resident if they fit beside the patch's pn W panels, the
        // W panels are streamed once per super-row unless all of W fits beside the A panels.  Strips along N (pm = 1) keep an A panel in one L2
        // (large images); strips along M (pn = 1) keep a W panel there -- at the 8x8 / 16x16 levels W is the big operand and N-fastest makes every
        // XCD stream ALL of it (1024 x 1280 x 11520: 239 MB instead of 69 MB through the fabric); wide layers at the 32x32 / 16x16 levels (GEGLU
        // 16384 x 5120 x 640: 20 W panels of 327 KB) want both bounded.
        static const int pm_env = tune_env("RF_GEMM_PM", 0), pn_env = tune_env("RF_GEMM_PN", 0), patch_on = tune_env("RF_GEMM_PATCH", 1);
        static const int mf_env = tune_env("RF_GEMM_MFAST", -1);
        const long long tiles = (long long)p.tiles_m * p.tiles_n, run = (tiles + 7) / 8;
        const double a_panel = (double)BM * (conv ? p.Ctot : p.K) * sizeof(T), w_panel = (double)BN * p.K * (W8 ? 1 : (int)sizeof(T));
        // strips: the panels a run touches (the rule of rounds 2-4, kept as it was: every split-K and two-blocks-per-CU launch was tuned on it)
        const double n_mx = (double)((run + p.tiles_n - 1) / p.tiles_n + (run % p.tiles_n ? 1 : 0)), n_nx = (double)(run < p.tiles_n ? run : p.tiles_n);
        const double m_mx = (double)(run < p.tiles_m ? run : p.tiles_m), m_nx = (double)((run + p.tiles_m - 1) / p.tiles_m + (run % p.tiles_m ? 1 : 0));
        const double cost_n = n_mx * a_panel + n_nx * w_panel, cost_m = m_mx * a_panel + m_nx * w_panel;
        const bool mfast = mf_env >= 0 ? mf_env != 0 : cost_m < 0.8 * cost_n;
        int pm = mfast ? p.tiles_m : 1, pn = mfast ? 1 : p.tiles_n;
        // patches: only launches of several rounds of one 8-wave block per CU without split-K (measured, profiles/r05c_tile_order_fetch.txt: GEGLU
        // 16384 x 5120 x 640 296 -> 169 MB of fabric reads per launch, 4096 x 10240 x 1280 284 -> 243 MB and 129 -> 125 us, 4096 x 1280 x 5760
        // 346 -> 51 MB; the two-blocks-per-CU 128 x 160 launches and the split-K levels LOSE -- their whole run is in flight at once and the
        // dispatch-order neighbours that share an A panel stream it in step).  Estimated L2-miss bytes of a run: per super-row the patches sweep
        // along N -- the pm A panels stay resident if they fit beside the patch's pn W panels, the W panels are streamed once per super-row.
        // (bf16 / fp32 operands only: the fp8 x fp8 and fp8-weight kernels of configs[4] LOSE 3 % per batch on the same rule -- half the operand bytes per tile, other
        //  tile shapes; profiles/r05r_r04_vs_r05_same_box.txt)
#ifndef RF_PATCH_FP8
#define RF_PATCH_FP8 0          // (experiment builds: 1 applies the patch order to the fp8 / fp8-weight kernels too -- the A/B of profiles/r06f)
#endif
        constexpr bool PATCH_T = RF_PATCH_FP8 || (!W8 && !std::is_same<T, fp8_t>::value);
        if (PATCH_T && patch_on && p.splitk == 1 && WM * WN == 8 && tiles >= 2 * 256 && mf_env < 0) {
            const double conc = 32.0, cap = 3.0 * 1024 * 1024;          // tiles in flight per XCD; L2 bytes the operands may take (4 MB minus output lines)
            auto cost_of = [&](int h, int w) {
                const double nsr = (double)run / ((double)h * p.tiles_n) < 1.0 ? 1.0 : (double)run / ((double)h * p.tiles_n);          // super-rows per run
                const double npc = (double)p.tiles_n / w;                                                                            // patches per super-row
                const double a_t = (h * a_panel + w * w_panel <= cap) ? h * a_panel : npc * h * a_panel;
                return nsr * (a_t + p.tiles_n * w_panel);
            };
            double best = mfast ? cost_m : cost_n;
            const double strip_model = mfast ? cost_of(p.tiles_m, 1) : cost_of(1, p.tiles_n);
            if (strip_model > best) best = strip_model;
            for (int h = 2; h < p.tiles_m && h <= 32; h *= 2) {
                int w = (int)((conc + h - 1) / h);
                if (w < 1) w = 1;
                if (w >= p.tiles_n) continue;                        // (a full-width patch is the strip order)
                const double c = cost_of(h, w);
                if (c < 0.6 * best) { pm = h; pn = w; best = c / 0.6; }
            }
        }
        if (pm_env > 0 && pn_env > 0) { pm = pm_env < p.tiles_m ? pm_env : p.tiles_m; pn = pn_env < p.tiles_n ? pn_env : p.tiles_n; }
        p.pm = pm;
        p.pn = pn;
    }
    // Split-K through fragment-ordered slabs (direct-epilogue kernels + splitk_reduce_frag_kernel): wherever the reduce pass has nothing to do
    // but alpha / bias / per-sample vector / residual (+ statistics) on 16-byte aligned rows
    static const int frag_env = tune_env("RF_SK_FRAG", 1);
    constexpr bool FRAG_OK = (WM * WN == 8) || (TM * TN == 5) || (TM * TN >= 16) || std::is_same<T, fp8_t>::value;      // (= DIRECT_OK below)
    const bool frag = FRAG_OK && frag_env && p.splitk > 1 && p.glds && p.epi2_ok && d->act == RF_ACT_NONE && !p.oscale && d->batch == 1 &&
                      (long long)p.splitk * p.tiles_m * p.tiles_n * BM * BN * 4 <= d->workspace_bytes;
    // statistics tiling: the GEMM tile, or the reduce pass's tile when split-K moves the epilogue there
    const int skr = sk_rows_for(p.M, p.N);
    const int st_rows = p.splitk > 1 ? (frag ? 32 : skr) : BM, st_cols = p.splitk > 1 ? (frag ? 32 * TN : SK_COLS) : BN;
    if (W8 || A8) RF_CHECK(p.glds, "rf_conv_gemm: fp8 operands need the direct-to-LDS main loop (one source, K and channel count multiples of the K tile)");
    if (p.oscale) {            // fp8 GEGLU output exists only in the direct epilogue: decided here so that rf_conv_gemm_plan reports it
        constexpr bool DOK = (WM * WN == 8) || (TM * TN == 5) || (TM * TN >= 16) || A8;
        RF_CHECK(DOK && TN % 2 == 0 && p.glds && p.epi2_ok && p.splitk == 1,
                 "rf_conv_gemm: fp8 output needs the direct epilogue (even TN, aligned rows, no split-K) -- this launch got a %d x %d tile", BM, BN);
    }
    // epilogue form (needed by the plan query too): see the selection notes below
    static const int epi_env = tune_env("RF_EPI", -1);
    constexpr bool PACKED_OK = std::is_same<TO, bf16_t>::value && !A8;
    constexpr bool DIRECT_OK = (WM * WN == 8) || (TM * TN == 5) || (TM * TN >= 16) || std::is_same<T, fp8_t>::value;      // (fp8 x fp8: every tile --
                                                                                          // the GEGLU epilogue with fp8 output exists only in this form)
    const bool ep_common = p.glds && p.epi2_ok && p.splitk == 1 && (!p.rowvec || p.rows_per_sample % BM == 0) &&
                           (d->act == RF_ACT_NONE || (d->act == RF_ACT_GEGLU && TN % 2 == 0));
    static const int gn_direct = tune_env("RF_EPI_GN", 1);      // 0: fused statistics keep EPI 0
    const bool direct = DIRECT_OK && (epi_env < 0 || epi_env == 1) && ep_common && (p.gn_rows == 0 || (gn_direct && d->act == RF_ACT_NONE));
    RF_CHECK(!hx || (HX_OK && conv && p.glds && p.KH == 3 && p.KW == 3 && p.stride == 1 && !p.ups && p.pad_t == 1 && p.pad_l == 1 && p.Hin == p.Hout &&
                     p.Win == p.Wout && p.Wout >= 16 && BM % p.Wout == 0 && (BM / p.Wout) * (p.Wout + 2) <= AXR_ && epi_env != 2 &&
                     !(DEEP_OK && (long long)p.tiles_m * p.tiles_n * p.splitk <= 256)),
             "rf_conv_gemm: korder 2 (row-extended A tiles) needs a bf16 3x3 stride-1 pad-1 convolution, Wout >= 16, whose %d-row tile holds whole image rows (Wout = %d) "
             "and that does not take the 4-stage ring", BM, p.Wout);
    // LayerNorm folding lives in the direct epilogue only (bf16 linear layers: the LNF variants of the kernel)
    constexpr bool LN_OK = DIRECT_OK && sizeof(T) == 2 && sizeof(TO) == 2 && !W8 && !A8;
    RF_CHECK(!(p.ln_out || p.ln_in) || (LN_OK && !conv),
             "rf_conv_gemm: LayerNorm folding needs a bf16 linear layer on a direct-epilogue tile (this launch: %d x %d, conv %d)", BM, BN, (int)conv);
    RF_CHECK(!p.ln_in || !d->residual, "rf_conv_gemm: a LayerNorm consumer takes no residual");
    RF_CHECK(!p.ln_out || (direct && d->act == RF_ACT_NONE && p.N % (32 * TN) == 0 && p.ln_out_parts == p.N / (32 * TN)),
             "rf_conv_gemm: ln_stats_out needs the direct epilogue (no split-K, act NONE) and ln_out_parts = N / %d (got %d; direct %d)", 32 * TN, p.ln_out_parts, (int)direct);
    RF_CHECK(!p.ln_in || (direct && p.ln_u && p.ln_in_parts >= 1 && p.ln_in_cols >= 1 && p.ln_eps > 0.f && BN <= 320),
             "rf_conv_gemm: ln_stats_in needs the direct epilogue (no split-K), ln_u, ln_in_parts / ln_in_cols >= 1 and ln_eps > 0 (direct %d)", (int)direct);
    if (p.plan) {
        p.plan[0] = st_rows; p.plan[1] = st_cols; p.plan[2] = p.splitk; p.plan[3] = BM; p.plan[4] = BN; p.plan[5] = 32 * TN;
        p.plan[6] = direct ? 1 : 0; p.plan[7] = frag ? 1 : 0;
        return 0;
    }
    if (p.gn_rows > 0) {
        RF_CHECK(p.gn_rows % st_rows == 0 && p.M % p.gn_rows == 0 && d->batch == 1 && d->act != RF_ACT_GEGLU,
                 "rf_conv_gemm: fused GroupNorm statistics need gn_rows %% %d == 0 (%d), batch 1, no GEGLU -- ask rf_conv_gemm_plan", st_rows, p.gn_rows);
        const int need = (p.gn_rows / st_rows) * ((p.N + st_cols - 1) / st_cols);
        for (int c = 0; c < 2; ++c)
            RF_CHECK(!p.gn_part[c] || (p.gn_cpg[c] > 0 && p.gn_slot[c] >= 0 && p.gn_slot[c] + need <= p.gn_nch[c]),
                     "rf_conv_gemm: GroupNorm consumer %d: cpg=%d slot=%d needs %d slots of %d", c, p.gn_cpg[c], p.gn_slot[c], need, p.gn_nch[c]);
    }
    dim3 grid(p.tiles_m * p.tiles_n, d->batch, p.splitk), block(WM * WN * 64);
    // Epilogue selection (RF_GEMM_DBG decomposition + tools/bench_gemm.py A/B, r02c: the chunked fp32 staging costs 47 us of the 84 us
    // of the 65536x960x320 qkv GEMM).
    //   EPI 1, direct register -> global row segments: no LDS pass, no barriers; fastest wherever nothing needs the tile as a whole
    //          (qkv 86 -> 59 us, proj 33 -> 24, ff2 62 -> 52, GEGLU 196 -> 164).  8-wave tiles and the 128x160 tile.
    //   EPI 2, bf16 tile staged in ONE pass by all waves (packed pairs): carries the fused GroupNorm statistics (bf16 output).
    //   EPI 0, fp32 tile staged in row chunks: everything else (split-K partials, per-row timestep vectors, activations other than
    //          GEGLU, fp32 output with statistics, unaligned shapes).
    // RF_EPI=0 forces EPI 0, RF_EPI=1 / 2 allow only that fast form (A/B runs).
    // (packed: measured neutral-to-negative in situ -- proj_out 4096x1280x1280 with fused statistics 37 -> 63 us, the rest within
    //  noise, r02f -- so it is opt-in: RF_EPI=2)
    const bool packed = !direct && PACKED_OK && epi_env == 2 && ep_common;
    RF_CHECK(!p.oscale || direct, "rf_conv_gemm: fp8 output needs the direct epilogue (8-wave tile, aligned rows, no split-K)");
    constexpr int smem_pk = BM * BN * 2;
    const int smem_l = (packed && smem_pk > smem) ? smem_pk : smem;
#define RF_LAUNCH_VARIANT_X(CONV_, GLDS_, EPI_, LNF_, HX_)                                                                      \
    {                                                                                                                            \
        constexpr int E_ = (EPI_ == 2 && PACKED_OK) ? 2 : ((EPI_ == 1 && DIRECT_OK) ? 1 : 0);                                   \
        if (GLDS_ && deep) {                                                                                                     \
            auto k = conv_gemm_kernel<T, TO, WM, WN, TM, TN, CONV_, GLDS_, (GLDS_ ? NSTD : 2), E_, (W8 && GLDS_), LNF_, false>; \
            RF_RAISE_LDS(k, smem_deep, "rf_conv_gemm");                                                                          \
            hipLaunchKernelGGL(k, grid, block, smem_deep, st, p);                                                                \
        } else {                                                                                                                 \
            auto k = conv_gemm_kernel<T, TO, WM, WN, TM, TN, CONV_, GLDS_, (GLDS_ ? NST : 2), E_, (W8 && GLDS_), LNF_, HX_>;    \
            constexpr int SM_ = HX_ ? smem_hx : (smem_pk > smem ? smem_pk : smem);                                               \
            RF_RAISE_LDS(k, SM_, "rf_conv_gemm");                                                                                \
            hipLaunchKernelGGL(k, grid, block, HX_ ? smem_hx : smem_l, st, p);                                                   \
        }                                                                                                                        \
    }
#define RF_LAUNCH_VARIANT_LN(CONV_, GLDS_, EPI_, LNF_) RF_LAUNCH_VARIANT_X(CONV_, GLDS_, EPI_, LNF_, false)
#define RF_LAUNCH_VARIANT(CONV_, GLDS_, EPI_) RF_LAUNCH_VARIANT_LN(CONV_, GLDS_, EPI_, 0)
    const int esel = packed ? 2 : ((direct || frag) ? 1 : 0);
    // The ring of four stages for 128x160 launches of at most one block per CU (4096 x 1280 x K <= 6000: the projections, ff.net.2 and 1x1 skips
    // of the 16x16 level, 25 launches per step): nothing else covers the single tile of look-ahead there.  Alone (warm weights) it is neutral
    // (4096x1280x5120 68.3 -> 66.7 us); in situ, where every launch streams weights the previous ones pushed out of the caches, -0.9 % per batch
    // (tools/archive/exp_r03_10.sh: 921.7 -> 913.5 ms, same box, two runs each).  Forcing the 8x8 level (M = 1024) onto this tile + ring: neutral for the
    // 3x3 convs, +0.9 % for its small projections.
    static const int deep_env = tune_env("RF_GEMM_DEEP", 256);        // largest grid (blocks) that takes the ring
    constexpr int smem_deep = NSTD * (BM + BN) * 128 > smem ? NSTD * (BM + BN) * 128 : smem;
    const bool deep = DEEP_OK && p.glds && !packed && !p.x3 && !hx && (long long)grid.x * grid.y * grid.z <= deep_env;
    if constexpr (A8) {            // fp8 activations: direct-to-LDS kernels only, staged or direct epilogue
        if (conv) { if (esel == 1) RF_LAUNCH_VARIANT(true, true, 1) else RF_LAUNCH_VARIANT(true, true, 0) }
        else { if (esel == 1) RF_LAUNCH_VARIANT(false, true, 1) else RF_LAUNCH_VARIANT(false, true, 0) }
    } else
    if (conv && p.glds) {
        if (hx) {
            if constexpr (HX_OK) {          // (checked above: korder 2 only reaches tiles that can take it)
                if (esel == 1) RF_LAUNCH_VARIANT_X(true, true, 1, 0, true)
                else RF_LAUNCH_VARIANT_X(true, true, 0, 0, true)
            }
        }
        else if (esel == 2) RF_LAUNCH_VARIANT(true, true, 2)
        else if (esel == 1) RF_LAUNCH_VARIANT(true, true, 1)
        else RF_LAUNCH_VARIANT(true, true, 0)
    } else if (conv) {
        RF_LAUNCH_VARIANT(true, false, 0)
    } else if (p.glds) {
        if (esel == 2) RF_LAUNCH_VARIANT(false, true, 2)
        else if (esel == 1) {
            // (LayerNorm roles: compile-time variants of the direct epilogue, bf16 linear layers only)
            if constexpr (LN_OK) {
                if (p.ln_in) RF_LAUNCH_VARIANT_LN(false, true, 1, 2)
                else if (p.ln_out) RF_LAUNCH_VARIANT_LN(false, true, 1, 1)
                else RF_LAUNCH_VARIANT(false, true, 1)
            } else RF_LAUNCH_VARIANT(false, true, 1)
        }
        else RF_LAUNCH_VARIANT(false, true, 0)
    } else {
        RF_LAUNCH_VARIANT(false, false, 0)
    }
#undef RF_LAUNCH_VARIANT
#undef RF_LAUNCH_VARIANT_LN
#undef RF_LAUNCH_VARIANT_X
    if (p.splitk > 1 && frag) {
        if constexpr (FRAG_OK)
            hipLaunchKernelGGL((splitk_reduce_frag_kernel<TO, WM, WN, TM, TN>), dim3(p.tiles_m * p.tiles_n, WM * WN * TM), dim3(64 * TN), 0, st, p);
    } else if (p.splitk > 1)
    {
        if (skr == 8) hipLaunchKernelGGL((splitk_reduce_kernel<TO, 8>), dim3((p.N + SK_COLS - 1) / SK_COLS, (p.M + 7) / 8), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((splitk_reduce_kernel<TO, 32>), dim3((p.N + SK_COLS - 1) / SK_COLS, (p.M + 31) / 32), dim3(256), 0, st, p);
    }
    RF_LAUNCH_CHECK("rf_conv_gemm");
    return 0;
}

#ifdef RF_EXPERIMENT
template <typename T, typename TO, int TN_, bool W8>
static int launch_wide(const rf_conv_gemm_desc* d, GemmParams& p, bool conv, hipStream_t st) {
    if constexpr (!W8 && sizeof(T) == 2 && sizeof(TO) == 2) return launch_cfg<T, TO, 2, 2, 4, TN_, false>(d, p, conv, st);
    else { RF_CHECK(false, "rf_conv_gemm: the one-wave-per-SIMD tiles are bf16 -> bf16 experiments"); }
}
#endif

template <typename T, typename TO, bool W8 = false>
static int launch_typed(const rf_conv_gemm_desc* d, GemmParams& p, bool conv, hipStream_t st) {
    if constexpr (std::is_same<T, fp8_t>::value) {
        // fp8 x fp8: 32-byte fragments (8 registers per 32-row block and k-step) leave no room for the 64-row wave tiles at two waves per
        // SIMD (they spill 260-940 bytes per lane): 128 x 320 / 128 x 256 blocks of 8 waves (wave tile 32 x 160 / 32 x 128), else the 4-wave tiles
        const int N_ = p.N;
        const bool g = d->act == RF_ACT_GEGLU;
        const bool n320 = !g && N_ % 320 == 0, n256 = N_ % 256 == 0 && !n320;
        const long long mt128 = (p.M + 127) / 128;
        const long long nt = n320 ? N_ / 320 : (n256 ? N_ / 256 : 0);
        if (n256 && ((p.M + 255) / 256) * nt >= 192) return launch_cfg<T, TO, 4, 2, 2, 4, false>(d, p, conv, st);
        if (nt > 0 && mt128 * nt * pick_splitk(d, p, mt128 * nt, 128) >= 160)
            return n320 ? launch_cfg<T, TO, 4, 2, 1, 5, false>(d, p, conv, st) : launch_cfg<T, TO, 4, 2, 1, 4, false>(d, p, conv, st);
        if (g) return launch_cfg<T, TO, 2, 2, 2, 2, false>(d, p, conv, st);
        if (N_ <= 64) return launch_cfg<T, TO, 4, 1, 1, 2, false>(d, p, conv, st);
        const int pad128 = ((N_ + 127) / 128) * 128, pad160 = ((N_ + 159) / 160) * 160;
        if (pad160 < pad128) return launch_cfg<T, TO, 4, 1, 1, 5, false>(d, p, conv, st);
        return launch_cfg<T, TO, 2, 2, 2, 2, false>(d, p, conv, st);
    } else {
    const int N = p.N;
    {   // experiments: RF_GEMM_CFG=<0..6> forces one tile configuration (GEGLU still needs an even TN)
        static const int forced_all = tune_env("RF_GEMM_CFG", -1);
        static const int small_k = tune_env("RF_SMALLK_K", 0);          // K <= this ...
        static const int small_cfg = tune_env("RF_SMALLK_CFG", -1);     // ... uses this config
        static const int m_exact = tune_env("RF_MCFG_M", 0);               // M == this ...
        static const int m_cfg = tune_env("RF_MCFG_CFG", -1);              // ... uses this config
        const int forced = forced_all >= 0 ? forced_all : (p.M == m_exact ? m_cfg : (p.K <= small_k ? small_cfg : -1));
        const bool g = d->act == RF_ACT_GEGLU;
        switch (forced) {
            case 0: if (!g && p.glds && d->batch == 1) return launch_cfg<T, TO, 4, 2, 2, 5, W8>(d, p, conv, st); break;
            case 1: if (p.glds && d->batch == 1) return launch_cfg<T, TO, 4, 2, 2, 4, W8>(d, p, conv, st); break;
            case 2: if (!g && p.glds && d->batch == 1) return launch_cfg<T, TO, 4, 2, 1, 5, W8>(d, p, conv, st); break;
            case 3: if (p.glds && d->batch == 1) return launch_cfg<T, TO, 4, 2, 1, 4, W8>(d, p, conv, st); break;
            case 4: return launch_cfg<T, TO, 2, 2, 2, 2, W8>(d, p, conv, st);
            case 5: return launch_cfg<T, TO, 4, 1, 1, 2, W8>(d, p, conv, st);
            case 6: if (!g) return launch_cfg<T, TO, 4, 1, 1, 5, W8>(d, p, conv, st); break;
#ifdef RF_EXPERIMENT
            // one wave per SIMD, 512 registers per lane: 4 waves as 2 x 2 with 128x160 / 128x128 wave tiles (a third fewer fragment reads per FLOP)
            case 7: if (!g && p.glds && d->batch == 1) return launch_wide<T, TO, 5, W8>(d, p, conv, st); break;
            case 8: if (p.glds && d->batch == 1) return launch_wide<T, TO, 4, W8>(d, p, conv, st); break;
#endif
            default: break;
        }
    }
    // 8-wave blocks with 320- / 256-wide tiles: half the LDS and L2 traffic per FLOP of the 4-wave configs.  Take the tallest
    // tile (256 rows, wave tile 64 x 160 / 64 x 128) that still gives ~one block per CU, else the 128-row variant.
    if (p.glds && d->batch == 1) {
        const int bk = 128 / (int)sizeof(T);
        const long long mt256 = (p.M + 255) / 256, mt128 = (p.M + 127) / 128;
        bool n320 = d->act != RF_ACT_GEGLU && N % 320 == 0;
        if (n320 && N % 256 == 0 && mt256 * (N / 320) >= 192) {
            // both widths divide N: whole rounds of 256 blocks decide -- qkv of the 16x16 level (4096 x 3840) is 192 tiles of 256x320 (one
            // round, a quarter of the CUs idle) or 240 of 256x256
            const long long t320 = mt256 * (N / 320), t256 = mt256 * (N / 256);
            if (((t256 + 255) / 256) * 256 < ((t320 + 255) / 256) * 320) n320 = false;
        }
        const bool n256 = N % 256 == 0 && !n320;
        const long long nt = n320 ? N / 320 : (n256 ? N / 256 : 0);
        if (nt > 0) {
            if (mt256 * nt >= 192) {
                // wave quantisation: 288 tiles of 256 rows (M = 73728, the 96x96 level at B = 4) are two rounds with the second one
                // 12 % full; quarter-size tiles at two blocks per CU fill the tail (768^2 bench: GEMM family 17.32 -> 16.77 ms per step)
                // (also 1.5 rounds: qkv of the 32x32 level, 384 tiles -> 1536 quarter tiles = three full rounds at two blocks per CU; and the
                //  K = C projections with a residual at the 64x64 level -- 107 FLOP per byte of compulsory traffic, bandwidth-bound: two
                //  co-resident blocks keep loads, residual reads and stores of different tiles in flight together, 61 -> 48 us with cold
                //  operands; together -0.7 % per batch, tools/archive/exp_r03_14.sh)
                const double rounds = (double)(mt256 * nt) / 256.0, rfill = rounds / (double)((mt256 * nt + 255) / 256);
                // 256-wide tiles (GEGLU, N not a multiple of 320) whose last round is 20-60 % full -- 4096 x 10240 x 1280, the GEGLU projection of
                // the 16x16 level: 640 tiles = 2.5 rounds, three round times -- are split along N: the whole rounds on 256-row tiles, the columns of
                // the partial round as a second launch, which this dispatcher gives 128-row tiles (one full round of half-size tiles).  Launches
                // that carry nothing tile-numbered (no fused GroupNorm statistics, no LayerNorm producer role, no fp8 output, one W).
                static const int tail_on = tune_env("RF_TAIL_SPLIT", 1);
                const long long tiles_ = mt256 * nt, full_ = tiles_ / 256, rem_ = tiles_ - full_ * 256;
                // (... and whose tail still fills the chip with 8-wave 128-row tiles: smaller tails fall to 4-wave tiles without the direct epilogue a
                //  LayerNorm consumer needs)
                const long long tail_nt_ = full_ >= 1 && mt256 > 0 ? nt - full_ * 256 / mt256 : 0;
                if (tail_on && !n320 && full_ >= 1 && rem_ * 5 >= 256 && rem_ * 5 <= 3 * 256 && (full_ * 256) % mt256 == 0 && mt128 * tail_nt_ >= 192 && p.gn_rows == 0 && !p.ln_out && !p.oscale &&
                    !p.w_ps && !p.x3 && d->batch == 1 && (d->act == RF_ACT_NONE || d->act == RF_ACT_GEGLU)) {
                    const int n1 = (int)(full_ * 256 / mt256) * 256;                       // columns of the whole rounds
                    auto part = [&](int n_off, int n_len) {
                        GemmParams q = p;
                        const int esw = W8 ? 1 : (int)sizeof(T);
                        q.N = n_len;
                        q.W = (const char*)p.W + (long long)n_off * p.ldw * esw;
                        q.w_bytes = (unsigned)(W8 ? (long long)n_len * p.ldw : ((long long)(n_len - 1) * p.ldw + p.K) * esw);
                        if (p.bias) q.bias = p.bias + n_off;
                        if (p.rowvec) q.rowvec = p.rowvec + n_off;
                        if (p.ln_u) q.ln_u = p.ln_u + n_off;
                        if (p.wscale) q.wscale = p.wscale + n_off;
                        if (p.act_vec) q.act_vec = p.act_vec + n_off;
                        const int o_off = d->act == RF_ACT_GEGLU ? n_off / 2 : n_off;       // GEGLU: 32 value | 32 gate column blocks -> N / 2 output columns
                        q.out = (char*)p.out + (long long)o_off * sizeof(TO);
                        if (p.residual) q.residual = (const char*)p.residual + (long long)o_off * sizeof(TO);
                        return q;
                    };
                    GemmParams q1 = part(0, n1);
                    const int rc = launch_cfg<T, TO, 4, 2, 2, 4, W8>(d, q1, conv, st);
                    if (rc != 0) return rc;
                    GemmParams q2 = part(n1, N - n1);
                    if (p.plan) {
                        // a plan query reports the whole-round tiling of the first part -- but BOTH parts must be launchable (a LayerNorm consumer
                        // needs the direct epilogue in the tail too): the tail's plan is validated here, not at the first replay.  Bit 1 of the
                        // eighth plan word says that the call runs as two GEMM kernels.
                        int tail_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                        q2.plan = tail_plan;
                        const int rc2 = launch_typed<T, TO, W8>(d, q2, conv, st);
                        if (rc2 == 0) p.plan[7] |= 2;
                        return rc2;
                    }
                    return launch_typed<T, TO, W8>(d, q2, conv, st);
                }
                if (n320 && (rfill < 0.6 || (rfill < 0.8 && rounds > 1.0) || (d->residual && p.K <= 320))) return launch_cfg<T, TO, 4, 1, 1, 5, W8>(d, p, conv, st);
                return n320 ? launch_cfg<T, TO, 4, 2, 2, 5, W8>(d, p, conv, st) : launch_cfg<T, TO, 4, 2, 2, 4, W8>(d, p, conv, st);
            }
            // K up to ~90 tiles: two co-resident 4-wave 128x160 blocks per CU (each other's prologue / epilogue cover) beat one
            // 8-wave 128x320 block in situ (sweep: -1.2 % per batch at 6000, worse again from 11520); RF_SHORTK overrides.  Also when K is
            // too short for split-K to bring the 128x320 grid to size (4096 x 1280 x 1280: 23 us, against 26 us on 128x128 tiles).
            static const int shortk = tune_env("RF_SHORTK", 6000);
            if (n320 && p.K <= shortk && mt128 * (N / 160) >= 256) return launch_cfg<T, TO, 4, 1, 1, 5, W8>(d, p, conv, st);
            // 64-128 tiles of 256 rows (the 3x3 convs of the 16x16 level, the long-K N = 640 convs of the 32x32 level): split-K 2-4 over the
            // 256-row tiles rather than 128-row tiles -- the 64x160 wave tile's main loop runs 1.0-1.25 PF where the 32x160 one stays below
            // 0.95 (6 KB of fragment reads per 5 MFMAs), and the fragment-ordered slabs keep the partial sums cheap
            static const int sk256_on = tune_env("RF_SK256", 1);
            if (sk256_on && d->act == RF_ACT_NONE) {
                const long long blocks = mt256 * nt * pick_splitk(d, p, mt256 * nt, bk);
                if (blocks > mt256 * nt && blocks <= 4 * mt256 * nt && blocks >= 192 && blocks <= 256)
                    return n320 ? launch_cfg<T, TO, 4, 2, 2, 5, W8>(d, p, conv, st) : launch_cfg<T, TO, 4, 2, 2, 4, W8>(d, p, conv, st);
            }
            if (mt128 * nt * pick_splitk(d, p, mt128 * nt, bk) >= 192) {
                return n320 ? launch_cfg<T, TO, 4, 2, 1, 5, W8>(d, p, conv, st) : launch_cfg<T, TO, 4, 2, 1, 4, W8>(d, p, conv, st);
            }
        }
    }
    if (d->act == RF_ACT_GEGLU) return launch_cfg<T, TO, 2, 2, 2, 2, W8>(d, p, conv, st);
    if (N <= 64) return launch_cfg<T, TO, 4, 1, 1, 2, W8>(d, p, conv, st);
    const int pad128 = ((N + 127) / 128) * 128, pad160 = ((N + 159) / 160) * 160;
    if (pad160 < pad128) return launch_cfg<T, TO, 4, 1, 1, 5, W8>(d, p, conv, st);
    return launch_cfg<T, TO, 2, 2, 2, 2, W8>(d, p, conv, st);
    }
}

#ifdef RF_GEMM_F16_UNIT
// gemm_f16.hip: the fp16 (RF_F16) instantiations of the templates above as a compile unit of their own -- built beside gemm.hip instead of
// lengthening its four minutes of compile time by another two.  rf_conv_gemm's argument checks and GemmParams set-up stay in gemm.hip.
int launch_f16(const rf_conv_gemm_desc* d, GemmParams& p, bool conv, hipStream_t st) {
    if (d->out_dtype == RF_F32) return launch_typed<f16_t, float>(d, p, conv, st);
    return launch_typed<f16_t, f16_t>(d, p, conv, st);
}
}  // namespace rf
#else
int launch_f16(const rf_conv_gemm_desc* d, GemmParams& p, bool conv, hipStream_t st);          // gemm_f16.hip
}  // namespace rf

static int conv_gemm_impl(const rf_conv_gemm_desc* d, void* stream, int* plan) {
    using namespace rf;
    RF_CHECK(d != nullptr, "rf_conv_gemm: null descriptor");
    const bool oq = d->out_dtype == RF_FP8_E4M3;       // fp8 output + block scales (GEGLU of the fp8 x fp8 path)
    const bool x3 = d->dtype == RF_BF16X3;
    const bool h16 = d->dtype == RF_F16;             // fp16 operands: the bf16 kernels' geometry on v_mfma_f32_32x32x16_f16
    RF_CHECK(d->dtype == RF_F32 || d->dtype == RF_BF16 || h16 || x3 || d->dtype == RF_FP8_E4M3, "rf_conv_gemm: bad dtype %d", d->dtype);
    RF_CHECK(!h16 || ((d->out_dtype == RF_F16 || d->out_dtype == RF_F32) && d->w_dtype == 0),
             "rf_conv_gemm: fp16 operands write fp16 or fp32 and take fp16 weights (out_dtype %d, w_dtype %d)", d->out_dtype, d->w_dtype);
    RF_CHECK(d->out_dtype != RF_F16 || h16, "rf_conv_gemm: fp16 output needs fp16 operands (dtype %d)", d->dtype);
    RF_CHECK(d->dtype != RF_FP8_E4M3 || (d->w_dtype == RF_FP8_E4M3 && d->wscale && d->ascale && d->as_ld > 0 && d->as_ld % 4 == 0 && (d->out_dtype == RF_BF16 || oq) &&
                                         d->C1 == 0 && d->batch == 1 && d->korder == 0 && d->K % 128 == 0 && (d->C0 % 128 == 0 || (d->KH == 1 && d->KW == 1))),
             "rf_conv_gemm: fp8 activations need fp8 weights + wscale, ascale with a pitch that is a multiple of 4, bf16 output, one source, batch 1, "
             "K a multiple of 128 (zero-padded weights) and, for k x k windows, C0 a multiple of 128");
    RF_CHECK(!x3 || (d->out_dtype == RF_F32 && d->w_dtype == 0 && d->korder == 0 && d->C1 == 0 && d->batch == 1 && d->ld0 >= 2 * d->C0),
             "rf_conv_gemm: split-bf16 operands need fp32 output, one source with pixel pitch >= 2*C0, batch 1, tap-major K");
    RF_CHECK(d->out_dtype == RF_F32 || d->out_dtype == RF_BF16 || d->out_dtype == RF_F16 || oq, "rf_conv_gemm: bad out_dtype %d", d->out_dtype);
    RF_CHECK(!oq || (d->dtype == RF_FP8_E4M3 && d->act == RF_ACT_GEGLU && d->oscale && d->os_ld >= d->N / 64 && !d->residual && d->N % 64 == 0),
             "rf_conv_gemm: fp8 output is the GEGLU epilogue of the fp8 x fp8 path (oscale, no residual, N a multiple of 64)");
    const bool a8 = d->dtype == RF_FP8_E4M3;           // fp8 activations + E8M0 block scales x fp8 weights (MX-scaled MFMA)
    const int vec = d->dtype == RF_F32 ? 4 : (a8 ? 16 : 8);
    const int ctot = d->C0 + d->C1;
    RF_CHECK(d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1, "rf_conv_gemm: bad sizes M=%d N=%d K=%d batch=%d", d->M, d->N, d->K, d->batch);
    RF_CHECK(d->src0 && d->W && d->out, "rf_conv_gemm: null operand");
    RF_CHECK(d->K % vec == 0 && ctot % vec == 0 && d->C0 % vec == 0 && d->ld0 % vec == 0,
             "rf_conv_gemm: K=%d C0=%d C1=%d ld0=%d must be multiples of %d", d->K, d->C0, d->C1, d->ld0, vec);
    RF_CHECK(d->C1 == 0 || (d->src1 && d->ld1 % vec == 0), "rf_conv_gemm: bad second source");
    const bool w8 = d->w_dtype == RF_FP8_E4M3 && d->dtype != RF_FP8_E4M3;          // W8A16: fp8 weights dequantised to bf16 in the fragment path
    RF_CHECK(d->w_dtype == 0 || d->w_dtype == RF_FP8_E4M3, "rf_conv_gemm: bad w_dtype %d", d->w_dtype);
    RF_CHECK(!w8 || (d->dtype == RF_BF16 && d->wscale && d->ldw % 128 == 0 && d->ldw >= d->K && d->batch == 1 && d->korder == 0),
             "rf_conv_gemm: fp8 weights need bf16 activations, wscale, batch 1 and ldw (bytes) a multiple of 128 >= K (ldw=%d K=%d)", d->ldw, d->K);
    RF_CHECK(w8 || d->ldw == 0 || (d->ldw >= (x3 ? 3 : 1) * d->K && d->ldw % vec == 0), "rf_conv_gemm: bad ldw=%d", d->ldw);
    RF_CHECK(!a8 || (d->ldw % 128 == 0 && d->ldw >= d->K), "rf_conv_gemm: fp8 x fp8 needs ldw (bytes) a multiple of 128 >= K");
    RF_CHECK(d->KH >= 1 && d->KW >= 1 && d->stride >= 1, "rf_conv_gemm: bad window");
    RF_CHECK(d->KH * d->KW * ctot <= d->K && d->K < d->KH * d->KW * ctot + 8 * vec,
             "rf_conv_gemm: K=%d inconsistent with KH*KW*(C0+C1)=%d", d->K, d->KH * d->KW * ctot);
    RF_CHECK(((uintptr_t)d->src0 | (uintptr_t)d->src1 | (uintptr_t)d->W) % 16 == 0, "rf_conv_gemm: operands must be 16-byte aligned");
    RF_CHECK(d->act != RF_ACT_GEGLU || (d->N % 64 == 0 && !d->rowvec), "rf_conv_gemm: GEGLU needs N %% 64 == 0 and no rowvec");
    RF_CHECK(!d->rowvec || d->rows_per_sample > 0, "rf_conv_gemm: rowvec needs rows_per_sample");
    RF_CHECK(d->korder == 0 || ((d->korder == 1 || d->korder == 2) && ctot % (8 * vec) == 0 && d->K == d->KH * d->KW * ctot),
             "rf_conv_gemm: korder=%d needs (C0+C1) to be a multiple of %d", d->korder, 8 * vec);
    RF_CHECK(d->korder != 2 || ((d->dtype == RF_BF16 || h16) && d->out_dtype == d->dtype && d->w_dtype == 0 && d->C1 == 0 && d->batch == 1),
             "rf_conv_gemm: korder=2 (row-extended A tiles) is built for bf16 -> bf16 / fp16 -> fp16 convolutions of one source");
    RF_CHECK(d->act != RF_ACT_PRELU || d->act_vec, "rf_conv_gemm: PReLU needs act_vec (per-column slopes)");
    const bool conv = !(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_t == 0 && d->pad_l == 0 && d->ups == 0 &&
                        d->C1 == 0 && d->Hin == d->Hout && d->Win == d->Wout);
    RF_CHECK(!conv || d->batch == 1, "rf_conv_gemm: batch > 1 only for plain GEMM");
    RF_CHECK(!conv || (d->Hout > 0 && d->Wout > 0 && d->M % (d->Hout * d->Wout) == 0), "rf_conv_gemm: M not a multiple of Hout*Wout");
    GemmParams p;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.src0 = d->src0; p.src1 = d->src1;
    p.C0 = d->C0; p.Ctot = ctot; p.ld0 = d->ld0; p.ld1 = d->ld1;
    p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ups = d->ups;
    p.W = d->W; p.ldw = d->ldw > 0 ? d->ldw : (x3 ? 3 : 1) * d->K; p.bias = d->bias; p.rowvec = d->rowvec; p.rows_per_sample = d->rows_per_sample; p.ldv = d->ldv;
    p.residual = d->residual; p.ldr = d->ldr; p.act = d->act; p.act_vec = d->act_vec; p.out = d->out; p.ldo = d->ldo; p.alpha = d->alpha;
    p.sA = d->sA; p.sW = d->sW; p.sO = d->sO; p.sR = d->sR;
    p.gn_rows = (d->gn_part0 || d->gn_part1) ? d->gn_rows : 0;
    p.gn_part[0] = d->gn_part0; p.gn_cpg[0] = d->gn_cpg0; p.gn_coff[0] = d->gn_coff0; p.gn_slot[0] = d->gn_slot0; p.gn_nch[0] = d->gn_nchunks0;
    p.gn_part[1] = d->gn_part1; p.gn_cpg[1] = d->gn_cpg1; p.gn_coff[1] = d->gn_coff1; p.gn_slot[1] = d->gn_slot1; p.gn_nch[1] = d->gn_nchunks1;
    p.plan = plan;
    p.ln_out = (float*)d->ln_stats_out; p.ln_out_parts = d->ln_out_parts;
    p.ln_in = (const float*)d->ln_stats_in; p.ln_in_parts = d->ln_in_parts; p.ln_in_cols = d->ln_in_cols; p.ln_eps = d->ln_eps; p.ln_u = d->ln_u;
    RF_CHECK(!(p.ln_out || p.ln_in) || ((d->dtype == RF_BF16 || h16) && d->out_dtype == d->dtype && d->batch == 1 && !(p.ln_out && p.ln_in)),
             "rf_conv_gemm: LayerNorm folding is built for bf16 / fp16 GEMMs (batch 1, one role per launch)");
    {
        static const int dbg = tune_env("RF_GEMM_DBG", 0);
        p.dbg = dbg;
    }
    RF_CHECK(!(d->gn_part0 || d->gn_part1) || d->gn_rows > 0, "rf_conv_gemm: gn_part set but gn_rows = %d", d->gn_rows);
    {
        const uintptr_t oa = d->out_dtype == RF_F32 ? 16 : 8;
        p.oscale = oq ? d->oscale : nullptr;
        p.os_ld = d->os_ld;
        const int nout = d->act == RF_ACT_GEGLU ? d->N / 2 : d->N;
        bool ok = (d->N % 4 == 0) && (nout % 4 == 0) && (d->ldo % 4 == 0) && ((uintptr_t)d->out % oa == 0) && (d->sO % 4 == 0);
        if (d->residual) ok = ok && (d->ldr % 4 == 0) && ((uintptr_t)d->residual % oa == 0) && (d->sR % 4 == 0);
        if (d->bias) ok = ok && ((uintptr_t)d->bias % 16 == 0);
        if (d->rowvec) ok = ok && (d->ldv % 4 == 0) && ((uintptr_t)d->rowvec % 16 == 0);
        p.vec_ok = ok ? 1 : 0;
        // direct epilogue: 16 contiguous output columns per lane, written / read as 16-byte vectors
        const int es_o = d->out_dtype == RF_F32 ? 4 : (oq ? 1 : 2);
        bool e2 = (d->N % 16 == 0) && (nout % 16 == 0) && ((long long)d->ldo * es_o % 16 == 0) && ((uintptr_t)d->out % 16 == 0) &&
                  ((long long)d->sO * es_o % 16 == 0) && d->act != RF_ACT_PRELU;
        if (d->residual) e2 = e2 && ((long long)d->ldr * es_o % 16 == 0) && ((uintptr_t)d->residual % 16 == 0) && ((long long)d->sR * es_o % 16 == 0);
        if (d->bias) e2 = e2 && ((uintptr_t)d->bias % 16 == 0);
        if (d->rowvec) e2 = e2 && (d->ldv % 4 == 0) && ((uintptr_t)d->rowvec % 16 == 0);
        p.epi2_ok = e2 ? 1 : 0;
    }
    {
        // direct-to-LDS main loop needs one source, 31-bit byte offsets ...
        const long long es = d->dtype == RF_F32 ? 4 : (a8 ? 1 : 2);
        const long long rows_a = conv ? (long long)(d->M / (d->Hout * d->Wout)) * d->Hin * d->Win : d->M;
        const long long ab = ((rows_a - 1) * d->ld0 + (conv ? d->C0 : d->K) + (x3 ? d->C0 : 0)) * es;
        const long long wb = (w8 || a8) ? (long long)d->N * p.ldw : ((long long)(d->N - 1) * p.ldw + (x3 ? 3 : 1) * d->K) * es;
        // ... and K tiles that never straddle a filter tap (uniform K offset per tile; rows packed as sample:12 | oy:10 | ox:10)
        const int bk = (int)(128 / es);
        const bool uniform = d->K % bk == 0 && (!conv || (ctot % bk == 0 && d->K == d->KH * d->KW * ctot &&
                                                         d->Hout <= 1024 && d->Wout <= 1024 && d->M / (d->Hout * d->Wout) < 4095));
        p.glds = (d->C1 == 0 && uniform && ab < 0x7fff0000LL && wb < 0x7fff0000LL) ? 1 : 0;
        p.korder = d->korder;
        p.a_bytes = (unsigned)(p.glds ? ab : 0);
        p.w_bytes = (unsigned)(p.glds ? wb : 0);
        p.ascale = d->ascale; p.as_ld = d->as_ld;
        p.as_bytes = (unsigned)(a8 ? rows_a * d->as_ld : 0);
        p.x3 = x3 ? 1 : 0;
        p.lo_off = x3 ? d->C0 * 2 : 0;
        p.w_ps = d->w_sample_stride;
        RF_CHECK(d->w_sample_stride == 0 || (!conv && !w8 && !a8 && !x3 && d->batch == 1 && d->rows_per_sample > 0 && p.glds &&
                                             d->w_sample_stride >= (long long)(d->N - 1) * p.ldw + d->K),
                 "rf_conv_gemm: w_sample_stride needs a plain bf16 / fp32 GEMM on the direct-to-LDS loop, batch 1, rows_per_sample > 0 and a stride of at least one W");
        if (x3) {
            RF_CHECK(p.glds && (conv || d->C0 == d->K), "rf_conv_gemm: split-bf16 operands need the direct-to-LDS main loop (K and channel count multiples of 64) and C0 == K for plain GEMMs");
            p.K = 3 * d->K;          // virtual K: three passes per real K tile
        }
    }
    hipStream_t st = (hipStream_t)stream;
    p.wscale = d->wscale;
    if (w8) {
        if (d->out_dtype == RF_F32) return launch_typed<bf16_t, float, true>(d, p, conv, st);
        return launch_typed<bf16_t, bf16_t, true>(d, p, conv, st);
    }
    if (x3) return launch_typed<bf16_t, float>(d, p, conv, st);
    if (h16) return launch_f16(d, p, conv, st);
    if (a8) return launch_typed<fp8_t, bf16_t>(d, p, conv, st);
    RF_CHECK(d->out_dtype != RF_F16, "rf_conv_gemm: fp16 output needs fp16 operands");
    if (d->dtype == RF_F32) {
        if (d->out_dtype == RF_F32) return launch_typed<float, float>(d, p, conv, st);
        return launch_typed<float, bf16_t>(d, p, conv, st);
    }
    if (d->out_dtype == RF_F32) return launch_typed<bf16_t, float>(d, p, conv, st);
    return launch_typed<bf16_t, bf16_t>(d, p, conv, st);
}

extern "C" int rf_conv_gemm(const rf_conv_gemm_desc* d, void* stream) { return conv_gemm_impl(d, stream, nullptr); }

namespace rf {
// one block per weight row: amax -> power-of-two scale -> e4m3fn bytes (v_cvt_pk_fp8_f32: round-to-nearest-even, saturating)
__global__ __launch_bounds__(256) void quantize_fp8_rows_kernel(const float* __restrict__ w, int K, int ldq, uint8_t* __restrict__ q,
                                                                float* __restrict__ scale) {
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* row = w + (long long)n * K;
    float m = 0.f;
    for (int k = tid; k < K; k += 256) m = fmaxf(m, fabsf(row[k]));
    m = wave_max(m);
    __shared__ float sm[4];
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    // smallest power of two s with m / s <= 448
    float s = 1.0f;
    if (m > 0.f) {
        int e;
        const float f = frexpf(m / 448.0f, &e);          // m / 448 = f * 2^e, f in [0.5, 1)
        s = ldexpf(1.0f, f == 0.5f ? e - 1 : e);
    }
    if (tid == 0) scale[n] = s;
    const float inv = 1.0f / s;                            // exact (power of two)
    uint8_t* qrow = q + (long long)n * ldq;
    for (int k4 = tid * 4; k4 < ldq; k4 += 1024) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k4 + e < K) ? row[k4 + e] * inv : 0.f;
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        *(int*)(qrow + k4) = pk;
    }
}
}  // namespace rf

extern "C" int rf_quantize_fp8_rows(const float* w, int N, int K, int ldq, void* q, float* scale, void* stream) {
    using namespace rf;
    RF_CHECK(w && q && scale && N > 0 && K > 0, "rf_quantize_fp8_rows: bad arguments");
    RF_CHECK(ldq >= K && ldq % 128 == 0 && (uintptr_t)q % 16 == 0, "rf_quantize_fp8_rows: ldq=%d must be a multiple of 128 >= K=%d", ldq, K);
    hipLaunchKernelGGL(quantize_fp8_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, w, K, ldq, (uint8_t*)q, scale);
    RF_LAUNCH_CHECK("rf_quantize_fp8_rows");
    return 0;
}

extern "C" int rf_conv_gemm_plan(const rf_conv_gemm_desc* d, int32_t* bm, int32_t* bn, int32_t* splitk) {
    using namespace rf;
    RF_CHECK(bm && bn && splitk, "rf_conv_gemm_plan: null output");
    int plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int rc = conv_gemm_impl(d, nullptr, plan);
    *bm = plan[0]; *bn = plan[1]; *splitk = plan[2];
    return rc;
}

extern "C" int rf_conv_gemm_plan2(const rf_conv_gemm_desc* d, int32_t* info8) {
    using namespace rf;
    RF_CHECK(info8, "rf_conv_gemm_plan2: null output");
    int plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int rc = conv_gemm_impl(d, nullptr, plan);
    for (int i = 0; i < 8; ++i) info8[i] = plan[i];
    return rc;
}

#endif          // RF_GEMM_F16_UNIT
#endif          // RF_KERNEL_ONLY
