// The fp16 (RF_F16, "fp16" throughput mode) instantiations of rf_conv_gemm's kernel templates: gemm.hip compiled as a second unit that keeps only
// rf::launch_f16 (conv_gemm_kernel<f16_t, f16_t | float, ...> through the same dispatcher, tile rules and epilogues as the bf16 mode).
#define RF_GEMM_F16_UNIT 1
#include "gemm.hip"
