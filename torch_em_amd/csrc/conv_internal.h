// conv_internal.h -- C++-internal interface between conv.hip (entry points, VALU kernels)
// and conv_mfma.hip (v_mfma_f32_32x32x2_f32 implicit-GEMM kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int64_t tem_conv_fwd_mfma_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int tem_conv_fwd_mfma(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w_packed,
                      const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                      int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                      hipStream_t s);

int64_t tem_conv_wgrad_mfma_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int tem_conv_wgrad_mfma(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                        int64_t g_ld, float* dw_tap_ci_co, float* db, void* ws, int64_t ws_bytes, int N, int D, int H,
                        int W, int Cin, int Cout, int kd, int kh, int kw, int sd_layout, hipStream_t s);

// conv_small.hip: HBM-bound special cases (return false when the shape is not covered)
bool tem_conv_fwd_cin1(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w,
                       const float* bias, float* y, int64_t y_ld, const float* ref, int N, int D, int H, int W,
                       int Cin, int Cout, int kd, int kh, int kw, int act, float* stat, hipStream_t s);
int64_t tem_conv_fwd_cin1_stat_blocks(int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int64_t tem_conv_wgrad_cin1_ws(int Cout, int ntaps);
bool tem_conv_wgrad_cin1(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                         int64_t g_ld, float* dw, float* db, void* ws, int N, int D, int H, int W, int Cin, int Cout,
                         int kd, int kh, int kw, int sd_layout, const float* gnx, int64_t gnx_ld, const float* gcoef,
                         hipStream_t s);
bool tem_conv_fwd_cout1(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w,
                        const float* bias, float* y, int64_t y_ld, const float* ref, int N, int D, int H, int W,
                        int Cin, int Cout, int kd, int kh, int kw, int act, hipStream_t s);
bool tem_conv1x1_proj(const float* x, int64_t x_ld, const float* scale, const float* w, const float* bias, float* y,
                      int64_t y_ld, const float* ref, int64_t NV, int Cin, int Cout, int act, hipStream_t s);
int64_t tem_conv1x1_proj_wgrad_ws(int Cin, int Cout);
bool tem_conv1x1_proj_wgrad(const float* x, int64_t x_ld, const float* scale, const float* g, int64_t g_ld, float* dw,
                            float* db, void* ws, int64_t NV, int Cin, int Cout, int sd_layout, hipStream_t s);
bool tem_conv1x1_out_bwd(const float* x, int64_t x_ld, const float* g, int64_t g_ld, const float* w, float* gx, int64_t gx_ld,
                         float* dw, float* db, void* ws, int64_t NV, int Cin, int Cout, int sd_layout, hipStream_t s);
bool tem_conv1x1_expand(const float* x, int64_t x_ld, const float* scale, const float* w, const float* bias, float* y,
                        int64_t y_ld, const float* ref, int64_t ref_ld, int64_t NV, int Cin, int Cout, int act,
                        hipStream_t s);
void tem_reduce_slabs_w(const float* part, int nchunks, int ntaps, int Cin, int Cout, int64_t chunk_stride, float* dw,
                        int sd_layout, hipStream_t s);
void tem_reduce_slabs_w_db(const float* part, int nchunks, int ntaps, int Cin, int Cout, int64_t chunk_stride, float* dw,
                           int sd_layout, const float* dbpart, int db_chunks, float* db, hipStream_t s, int64_t db_stride = 0);
void tem_reduce_slabs(const float* part, int nchunks, int64_t n, int64_t chunk_stride, float* out, hipStream_t s);

// conv_bf16x3.hip: split-bf16 ("bf16x3") MFMA path
int tem_pack_weights_bf16x3(const float* w, float* dst, int Cout, int Cin, int kd, int kh, int kw, int transpose,
                            int nsplit, hipStream_t s);
int tem_conv_fwd_bf16x3(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                        const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                        int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                        int nsplit, float* stat, hipStream_t s);
int64_t tem_conv_fwd_bf16x3_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit);
// conv_pp.hip: ping-pong team kernel for the levels with many patches (1 launched, 0 shape not taken, -1 error set)
int tem_conv_fwd_pp(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                     const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H,
                     int W, int Cin, int Cout, int kd, int kh, int kw, int act, int nsplit, float* stat, hipStream_t s);
int64_t tem_conv_pp_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit);
int tem_conv_pp_tiles(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit);
// conv_zr.hip: z-reuse ping-pong kernel, 3x3x3 only (same return convention as tem_conv_fwd_pp)
int tem_conv_fwd_zr(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp, const float* bias,
                    float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout,
                    int kd, int kh, int kw, int act, int nsplit, float* stat, hipStream_t s);
int tem_conv_fwd_zr_splitk(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                           const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                           int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                           int nsplit, float* stat, hipStream_t s);
int tem_conv_zr_splitk_ks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit);
int64_t tem_conv_zr_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit);
// conv1x1_stream.hip: 1x1x1 convolution / data gradient as a streaming GEMM (false: not taken)
bool tem_conv1x1_stream(const float* x, int64_t x_ld, const float* scale, const float* wp, const float* bias, float* y,
                        int64_t y_ld, const float* ref, int64_t ref_ld, int64_t NV, int Cin, int Cout, int act, int nsplit,
                        const float* stat, hipStream_t s);
// shared with conv_mfma.hip
int tem_fwd_ksplit(int64_t nblk, int nchunks);
void tem_splitk_epilogue(const float* part, int ksplit, int64_t NV, int Cout, const float* bias, int act,
                         const float* ref, int64_t ref_ld, float* y, int64_t y_ld, hipStream_t s);
// TEM_BP_NORM_SUMS of the call in flight, as the split-K data gradient whose epilogue can deliver the rows reads it
struct TemDgradSumsReq {
    const void* x;       // input of the norm the gradient lands behind: [N*V][x_ld], element type of the gradient
    int64_t x_ld;
    const float* mean;
    const float* rstd;
    int G;
    float* part;         // [N][nblk][C][2]
    int64_t nblk;
};
void tem_splitk_epilogue_bwd_sums(const float* part, int ksplit, int N, int64_t V, int Cout, const float* bias, int act,
                                  const float* ref, int64_t ref_ld, float* y, int64_t y_ld, const TemDgradSumsReq& rq,
                                  hipStream_t s);
int64_t tem_splitk_stat_blocks(int64_t V, int Cout);
int64_t tem_conv_zr_splitk_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit);
void tem_splitk_epilogue_stats(const float* part, int ksplit, int N, int64_t V, int Cout, const float* bias, int act,
                               const float* ref, int64_t ref_ld, float* y, int64_t y_ld, float* stat, hipStream_t s);
int64_t tem_conv_wgrad_bf16x3_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int tem_conv_wgrad_bf16x3(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                          int64_t g_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                          int Cin, int Cout, int kd, int kh, int kw, int sd_layout, int h16, const float* w_sd,
                          const float* gamma, const float* beta, float* norm_sums, hipStream_t s);
// largest |g| as a by-product of the z-sliding weight gradient (tem_conv3d_wgrad_gmax)
extern thread_local unsigned* tem_wgrad_gmax_target;
int tem_conv_wgrad_gmax_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
// h16 == 3 of tem_conv_wgrad_bf16x3 ("fp16 2x1": x^ two fp16 terms, g one fp16 term prescaled from this device word)
extern thread_local const unsigned* tem_wgrad_gscale_source;
int tem_conv_wgrad_gscaled_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int tem_conv_wgrad_cs_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int st, int64_t x_cs);
int tem_conv_wgrad_tr_fp32_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);   // h16 == 4
// prescale of the z-reuse kernel's input by a power of two derived from a device-side |max| (tem_conv3d_fwd_gscaled)
extern thread_local const unsigned* tem_zr_in_amax;
extern thread_local const float* tem_zr_ref_coef;
// conv_wgrad_tr.hip: z-sliding weight gradient with a staging team and transposing LDS reads (option wgrad_zs = 3)
void tem_conv_wgrad_tr_launch(int h16, unsigned nblk, const float* x, int64_t x_ld, const float* scale, const float* shift,
                              const float* g, int64_t g_ld, float* zpart, float* zdb, int N, int D, int H, int W, int Cin,
                              int Cout, int T, int nY, int nX, int zsegs, int Ss, int ncz, unsigned* gmax,
                              const unsigned* g_amax, hipStream_t s);
// TEM_BP_NORM_COEF of the call in flight, as tem_wgrad_sums_launch reads it (delivered when the layer's group layout allows it)
struct TemWgradCoefReq {
    int G;
    const float* mean;
    const float* rstd;
    float* coef;
};
// wgrad_sums.hip: norm-backward sums from the weight gradient
int tem_conv_wgrad_sums_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
int64_t tem_wgrad_sums_ws_floats(int N, int D, int H, int Cin, int Cout);
void tem_wgrad_sums_launch(const float* zpart, int Ss, int ks2, const float* zdb, const float* g, int64_t g_ld,
                           const float* w, const float* gamma, const float* beta, float* dw, float* extra, int N, int D,
                           int H, int W, int Cin, int Cout, float* sums, int db_chunks, float* db, hipStream_t s);
