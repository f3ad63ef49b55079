// conv_internal.h -- C++-internal interface between conv.hip (entry points, VALU kernels)
// and conv_mfma.hip (v_mfma_f32_32x32x2_f32 implicit-GEMM kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int tem_conv_fwd_mfma(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w_packed,
                      const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H,
                      int W, int Cin, int Cout, int kd, int kh, int kw, int act, hipStream_t s);

int64_t tem_conv_wgrad_mfma_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int tem_conv_wgrad_mfma(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                        int64_t g_ld, float* dw_tap_ci_co, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                        int Cin, int Cout, int kd, int kh, int kw, hipStream_t s);
