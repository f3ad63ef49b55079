/*
 * tem_hip.h -- C-ABI of libtem_hip.so: the MI355X (gfx950) kernels behind the
 * torch-em 3D U-Net training path.
 *
 * The reference (constantinpape/torch-em) has NO native code and no FFI: its hot
 * path is Python nn.Modules over ATen ops.  This header
 * is therefore the boundary the build ADDS: every entry point names the
 * reference symbol (file:line under /root/reference) whose arithmetic it
 * replaces.  INTEGRATION.md shows the ctypes binding a torch-em maintainer
 * would add.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every buffer (inputs, outputs,
 *     workspaces) is owned by the caller and is DEVICE memory on the current
 *     HIP device; the library allocates nothing persistent;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no
 *     implicit device synchronisation, re-entrant; the only global state is the
 *     thread-local last-error string and the dispatch options of
 *     tem_set_option() (the library never reads the environment);
 *   - return 0 on success, negative TEM_E* on failure (never throws);
 *     tem_last_error() gives the message for the calling thread;
 *   - activations are channels-last "NDHWC": element (n,z,y,x,c) of a
 *     tensor with leading dimension `ld` (ELEMENTS between consecutive voxels,
 *     ld >= C, so a tensor can be a channel slice of a wider concat buffer)
 *     lives at ((((n*D+z)*H+y)*W+x)*ld + c);  2-D data uses D == 1.
 *     Element type: fp32 by default (the `const float*` signatures); the two
 *     mixed modes store fp16 / bf16 tensors between the kernels of a step
 *     (round 5) -- such a call names the storage of its tensors explicitly
 *     (TEM_ST_F32 / TEM_ST_F16 / TEM_ST_BF16: the TEM_MFMA_STX / STY bits of
 *     `use_mfma`, or the `int st` of the *_st entry points) and passes the
 *     16-bit buffers through the same pointer arguments;
 *   - V = D*H*W voxels per sample.
 */
#ifndef TEM_HIP_H
#define TEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEM_OK 0
#define TEM_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define TEM_ELAUNCH (-2)  /* HIP launch / runtime error */
#define TEM_EWS (-3)      /* workspace too small */

typedef void* tem_stream_t; /* hipStream_t */

/* ---- activation storage types (round 5) --------------------------------------
 * The reference trains under torch.autocast(float16 | bfloat16) whenever mixed_precision=True on a GPU
 * (trainer/default_trainer.py:134-142, 781-794): every activation between two ops is a 16-bit tensor.  The `_st` / `_ex`
 * entry points take the element type of their activation tensors explicitly; the plain entry points are the TEM_ST_F32
 * case.  With a 16-bit type the tensor pointers are `const void*`, leading dimensions stay in ELEMENTS, a vector of 4
 * elements is 8 bytes (alignment requirements scale accordingly), all arithmetic stays fp32 and a value is rounded once
 * (to nearest even) when it is stored.  Statistics, coefficients, parameters, gradients of parameters, split-K
 * workspaces and the network output are always fp32. */
#define TEM_ST_F32 0
#define TEM_ST_F16 1
#define TEM_ST_BF16 2
/* The convolution entry points (and their query functions) carry the storage types in the high bits of `use_mfma`:
 *   use_mfma | TEM_MFMA_STX(st of x) | TEM_MFMA_STY(st of y and ref; weight gradients: of g)
 * Supported pairs: both fp32; one side fp32 and the other 16-bit (first layer: fp32 network input -> 16-bit activations;
 * out_conv: 16-bit activations -> fp32 prediction; VALU kernels, use_mfma 0); both the SAME 16-bit type (MFMA kernels: fp16
 * storage with use_mfma 5, bf16 storage with use_mfma 7 -- the stored values ARE the MFMA operands; the exact-fp32 and the
 * split-precision modes take fp32 tensors only).  Workspaces, statistics partials and split-K partial sums stay fp32. */
#define TEM_MFMA_STX(st) ((st) << 8)
#define TEM_MFMA_STY(st) ((st) << 12)

/* ---- library ---------------------------------------------------------- */
const char* tem_last_error(void);
int tem_version(void);
/* number of CUs of the current device (used by callers to size split-K). */
int tem_device_cus(void);
/* Dispatch options: process-wide switches between kernel variants that compute the same result (used by
 * profiling scripts and A/B tests; the defaults are the measured-fastest choices).  No reference counterpart
 * (the reference selects nothing: it calls ATen).  Names:
 *   "conv_fwd_variant"   -1 auto | 0 one-patch-per-workgroup kernel | 1 ping-pong team kernel (conv_pp.hip) for every shape
 *                        it can take | 2 z-reuse team kernel (conv_zr.hip, 3x3x3) for every shape it can take
 *   "wgrad_zs"            3 | 2 | 1 | 0   z-sliding weight gradient (3x3x3, D >= 8): 3 (default) = staging team + voxel-major
 *                        LDS records read with ds_read_b64_tr_b16 (k_conv_wgrad_tr), 2 = staging team with channel-major
 *                        planes (k_conv_wgrad_zt), 1 = the round-2 kernel (k_conv_wgrad_zs), 0 = patch kernel
 *   "wgrad_zs_persist"    1 | 0   persistent column segments of that kernel
 *   "wgrad_sums"          1 | 0   norm-backward sums taken from the weight gradient
 *   "wgrad_sums_min_mb"   128     ... for layers whose norm input has at least this many MiB
 *   "fwd_persistent"     -1 | 0 | 1   exact-fp32 forward: persistent variant (-1: 64-column tiles only)
 *   "conv1x1_stream"      1 | 0   1x1x1 convolutions / data gradients as a streaming GEMM instead of the patch kernel
 *   "dice_vox"            1 | 0   Dice sums / gradient kernels with one voxel per thread (C <= 16; 0: the (channel, voxel) kernels)
 *   "fwd_ksplit_chunks"   0       split-K forward: at most this many 16-channel chunks per partial (0: heuristic)
 *   "wgrad_cus"           256     workgroups the z-sliding weight gradient asks for
 *   "upsample_generic"    0 | 1   1: the any-factor gather kernels also for factors (1|2, 2, 2)
 *   "team_min_units"      0       units a launch needs for the team kernels (0 = two per CU; lower values measured slower)
 *   "zr_splitk"           1 | 0   z-reuse kernel with split input channels for launches with too few tiles (16^3 / 32^3 levels)
 *   "zr_wide"             1 | 0   one-term modes (5, 7) of the z-reuse kernel stage 32 channels = whole 128-byte lines per phase
 *   "zr_tile_blocks"      1 | 0   z-reuse kernel walks its tiles in 4 x 4 x 4 blocks (one compact block per XCD at a time)
 *   "fp32_zr"             1 | 2 | 0   use_mfma 1 (exact fp32): 3x3x3 forward / data gradient on the z-reuse team kernel (2: its
 *                        one-team-per-workgroup variant, staging from inside the tap loop: measured 1-4 % slower, same results)
 *                        (k_conv_zr<..., X32>: fused statistics, ReLU mask, norm backward, split-K as in the split modes;
 *                        0: the one-patch-per-workgroup kernels of rounds 1-5, which deliver none of those)
 * Unknown names return TEM_EINVAL. */
int tem_set_option(const char* name, int64_t value);
int tem_get_option(const char* name, int64_t* value);

/* ---- convolution -------------------------------------------------------
 * Replaces nn.Conv3d / nn.Conv2d as used by ConvBlock (model/unet.py:417-438),
 * Upsampler.conv (model/unet.py:453) and out_conv (model/unet.py:638), plus the
 * autograd convolution_backward of those modules.  Stride 1, zero padding
 * (k-1)/2 per axis ("same"), kernel sizes 1 or 3 per axis.
 *
 * Weight layouts (tem_conv_pack_weights converts from the reference's
 * state_dict layout [Cout][Cin][kd][kh][kw]):
 *   TEM_WL_GENERIC  [tap][ci][co]                       any Cin, Cout
 *   TEM_WL_MFMA     [co/32][tap][ci/8][2][32][4]        Cin%16==0, Cout%32==0
 *                   (the B-fragment order of v_mfma_f32_32x32x2_f32)
 *   TEM_WL_BF16X3   [co/32][tap][ci/16][hi|lo][64][8 bf16]  Cin%16==0, Cout%32==0
 *                   (each fp32 weight split into two bf16 terms, B-fragment order of
 *                   v_mfma_f32_32x32x16_bf16; same byte count as fp32)
 *   TEM_WL_BF16X6   same with three bf16 terms per weight (all 24 mantissa bits; 1.5x the bytes --
 *                   tem_conv_packed_size() returns the size of the largest layout)
 * transpose==1 packs the data-gradient operator: taps flipped, Cin<->Cout
 * swapped, so that dgrad is again a tem_conv3d_fwd call.
 */
#define TEM_WL_GENERIC 0
#define TEM_WL_MFMA 1
#define TEM_WL_BF16X3 2
#define TEM_WL_BF16X6 3
#define TEM_WL_F16X3 4  /* like BF16X3 with two fp16 terms per weight (22 mantissa bits), lo plane stored x 2^12 */
#define TEM_WL_F16 5    /* ONE fp16 term per weight (half the bytes): the mixed-precision mode, use_mfma 5 */
#define TEM_WL_BF16 7   /* ONE bf16 term per weight: mixed precision with mixed_precision_dtype="bfloat16", use_mfma 7 */
#define TEM_WL_F16X3S 6 /* two fp16 terms of the weight x 2^7 (both terms carry the prescale; activations are staged x 2^5,
                           the kernel's epilogue multiplies by 2^-12): one accumulator for all three products, use_mfma 6 */
#define TEM_ACT_NONE 0
#define TEM_ACT_RELU 1
#define TEM_ACT_SIGMOID 2

int64_t tem_conv_packed_size(int Cout, int Cin, int kd, int kh, int kw); /* floats */
int tem_conv_pack_weights(const float* w, float* dst, int Cout, int Cin, int kd, int kh, int kw,
                          int transpose, int layout, tem_stream_t stream);
/* All split-layout (TEM_WL_BF16X3 / BF16X6 / F16X3) packs of a model in ONE launch.  descs_dev: device array of n
 * records { const float* w; void* dst; int32 Cout, Cin, kd, kh, kw, transpose, nsplit(1|2|3), fp16(0 bf16 | 1 fp16 | 2 fp16 with the lo plane scaled by 2^12 = TEM_WL_F16X3); int64 begin }
 * (56 bytes each, `begin` = running offset in units of 8 weights, ascending); total = sum of Cout*Cin*taps/8.  Same result as n calls
 * of tem_conv_pack_weights. */
int tem_conv_pack_weights_batch(const void* descs_dev, int n, int64_t total, tem_stream_t stream);
/* Same records and result, coalesced reads: one workgroup per [32 out][32 in][taps] tile of a tensor staged in LDS
 * (taps <= 27, Cout and Cin multiples of 16).  `begin` of a record = its first TILE (tiles per tensor:
 * ceil(Cout / 32) * ceil(Cin / 32)); total_tiles = their sum = the grid. */
int tem_conv_pack_weights_tiles(const void* descs_dev, int n, int64_t total_tiles, tem_stream_t stream);
/* inverse of the GENERIC pack for weight gradients: [tap][ci][co] -> [Cout][Cin][kd][kh][kw] */
int tem_conv_unpack_wgrad(const float* dw_tap_ci_co, float* dw, int Cout, int Cin, int kd, int kh, int kw,
                          tem_stream_t stream);

/* y = act(conv(x * scale + shift) + bias) [* (ref > 0)]
 *   scale/shift: optional [N][Cin] per-sample per-channel affine applied to the
 *                input BEFORE zero padding (the fused pre-norm of ConvBlock,
 *                model/unet.py:429-438); NULL = identity.
 *   bias:        optional [Cout].
 *   ref:         optional tensor (ld ref_ld) of y's shape; when given the result
 *                is zeroed where ref <= 0 (ReLU backward, threshold_backward).
 *   use_mfma:    1 = v_mfma_f32_32x32x2_f32 implicit-GEMM kernels, exact fp32: every output value is ONE fp32 fmaf
 *                chain (tests/test_gpu_ops.py::test_conv_exact_fp32_zreuse_is_an_fmaf_chain), the arithmetic of the
 *                reference's CPU path (needs the TEM_WL_MFMA pack); 2 = split-bf16 kernel: every operand x = hi + lo in bf16,
 *                products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32
 *                accumulation, ~1e-5 relative per product (needs the TEM_WL_BF16X3 pack);
 *                3 = the same with three bf16 terms per operand and the six products of order
 *                <= 2^-16 ("bf16x6"): per-product error ~2^-23, the fp32 class (TEM_WL_BF16X6 pack);
 *                4 = "fp16x3": x = hi + lo in fp16 (22 mantissa bits), hi*hi + hi*lo + lo*hi on
 *                v_mfma_f32_32x32x16_f16: ~2^-22 per product at half the MFMAs of mode 3; operands
 *                must stay far inside the fp16 range, i.e. pre-normalised activations (scale/shift
 *                given) and weights -- not gradients (TEM_WL_F16X3 pack);
 *                5 = mixed precision: operands rounded to fp16 (round-to-nearest-even, overflow -> inf), ONE
 *                v_mfma_f32_32x32x16_f16 per product, fp32 accumulation and fp32 output -- the arithmetic of
 *                the reference's default GPU mode, torch.autocast(float16) around nn.Conv3d
 *                (trainer/default_trainer.py:134-142,789-794); NOT parity-grade (2^-11 per operand), used only
 *                when the trainer is created with mixed_precision=True (TEM_WL_F16 pack);
 *                7 = the same with operands rounded to bf16 (ONE v_mfma_f32_32x32x16_bf16 per product): the
 *                arithmetic of torch.autocast(bfloat16), mixed_precision_dtype="bfloat16" (:134-142: no GradScaler
 *                for this dtype); 2^-8 per operand (TEM_WL_BF16 pack);
 *                0 = VALU kernel (TEM_WL_GENERIC pack).
 *   ws:          optional workspace of tem_conv3d_fwd_ws() bytes.  Spatially small, channel-rich
 *                layers (the 8^3/16^3 levels) cannot fill 256 CUs with (patch x Cout-tile)
 *                workgroups; with a workspace the MFMA kernel also splits the input channels
 *                ("split-K") and a second tiny kernel sums the slices and applies the epilogue.
 */
int64_t tem_conv3d_fwd_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
int tem_conv3d_fwd(const float* x, int64_t x_ld, const float* scale, const float* shift,
                   const float* w_packed, const float* bias, float* y, int64_t y_ld,
                   const float* ref, int64_t ref_ld, void* ws, int64_t ws_bytes,
                   int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                   int act, int use_mfma, tem_stream_t stream);

/* tem_conv3d_fwd that ALSO emits the first stage of the next layer's normalisation statistics: per (sample, output
 * patch, channel) partial sums (sum y, sum y^2) of the stored output, taken from the accumulators in the epilogue --
 * the 2 x 128^3 x 32 output of a level-0 conv is not read back for `nn.InstanceNorm3d` / `GroupNorm` / `BatchNorm3d`
 * (reference ConvBlock, model/unet.py:429-438: norm(conv(x))).  stat_part: [N][stat_blocks][Cout][2] floats with
 * stat_blocks = tem_conv3d_fwd_stat_blocks(...), which returns 0 for launches that cannot provide them (the generic VALU
 * kernels, the exact-fp32 PATCH kernels (use_mfma 1 on shapes the z-reuse kernel does not take, or option fp32_zr = 0),
 * the patch kernel's split-K launches; the split-K launches of the z-reuse kernel DO:
 * their epilogue writes the rows): use tem_norm_stats there.  tem_norm_finalize_partials (below) merges them. */
int64_t tem_conv3d_fwd_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
/* Which kernel family a tem_conv3d_fwd launch of this shape selects under the current options: 3 = the z-reuse team
 * kernel (csrc/conv_zr.hip: 3x3x3, 4x16x8 patches), 4 = the same kernel with the input channels split over several units
 * and a summing epilogue (16^3 / 32^3 levels: too few tiles otherwise; needs the tem_conv3d_fwd_ws() workspace, no fused
 * statistics), 1 or 2 = the ping-pong team kernel (csrc/conv_pp.hip: 3x3x3 / 1x3x3,
 * two-plane layouts, enough patches to fill the chip) with that many 32-column output tiles per team, 0 = everything else.  Profiling / test aid (kernel tables of bench.py, the per-
 * instantiation parity tests); alignment fall-backs of an individual launch are not reflected. */
int tem_conv3d_fwd_kernel(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
int tem_conv3d_fwd_stats(const float* x, int64_t x_ld, const float* scale, const float* shift,
                         const float* w_packed, const float* bias, float* y, int64_t y_ld,
                         const float* ref, int64_t ref_ld, void* ws, int64_t ws_bytes,
                         int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                         int act, int use_mfma, float* stat_part, int64_t stat_blocks, tem_stream_t stream);

/* dw = sum_v xhat[v+tap][ci] * g[v][co]  (xhat = x*scale+shift, zero padded)
 * db[co] = sum_v g[v][co] (optional).  ws: workspace of tem_conv3d_wgrad_ws() bytes.
 * sd_layout != 0: dw is written in the reference's state_dict order [Cout][Cin][kd][kh][kw]
 * (what param.grad needs); 0: tap-major [tap][ci][co] (tem_conv_unpack_wgrad converts).
 * use_mfma: 0 VALU, 1 exact-fp32 MFMA, 2 split-bf16 MFMA (tem_conv3d_fwd), 5 mixed precision (x and g rounded to
 * fp16, one MFMA per product, in the z-sliding 3x3x3 kernel; other shapes run mode 2); 8 (workspace query only) the
 * fp16 2x1 arithmetic of tem_conv3d_wgrad_gscaled. */
int64_t tem_conv3d_wgrad_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
int tem_conv3d_wgrad(const float* x, int64_t x_ld, const float* scale, const float* shift,
                     const float* g, int64_t g_ld, float* dw_tap_ci_co, float* db,
                     void* ws, int64_t ws_bytes,
                     int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                     int use_mfma, int sd_layout, tem_stream_t stream);

/* Backward of the output projection out_conv = nn.Conv3d(features, out_channels, 1) (/root/reference/torch_em/model/unet.py:638:
 * the wgrad + dgrad halves of its convolution_backward, and the threshold_backward of the ReLU in front of it) in ONE pass over
 * x [NV][Cin] (the ReLU output the projection read): dw [Cout][Cin] (state_dict order), db [Cout] (optional) from
 * g [NV][Cout], and gx[v][ci] = x[v][ci] > 0 ? sum_co g[v][co] w[co][ci] : 0.  w in state_dict layout.  Shapes:
 * tem_conv1x1_out_bwd_ok (Cin 32 or 64, Cout <= 4); workspace tem_conv1x1_out_bwd_ws() bytes.  */
int tem_conv1x1_out_bwd_ok(int Cin, int Cout);
int64_t tem_conv1x1_out_bwd_ws(int Cin, int Cout);
int tem_conv1x1_out_bwd(const float* x, int64_t x_ld, const float* g, int64_t g_ld, const float* w, float* gx, int64_t gx_ld,
                        float* dw, float* db, void* ws, int64_t ws_bytes, int64_t NV, int Cin, int Cout, tem_stream_t stream);
/* ... for x / gx of storage type st_x and g of storage type st_g; out_amax (optional): device word that receives max |gx|
 * as an integer atomicMax of the bit patterns */
int tem_conv1x1_out_bwd_st(const void* x, int64_t x_ld, const void* g, int64_t g_ld, const float* w, void* gx, int64_t gx_ld,
                           float* dw, float* db, void* ws, int64_t ws_bytes, int64_t NV, int Cin, int Cout,
                           unsigned* out_amax, int st_x, int st_g, tem_stream_t stream);

/* tem_conv3d_wgrad (w == NULL, norm_sums == NULL) or tem_conv3d_wgrad_sums that ALSO reports the largest |g|: g_amax
 * (device, one 32-bit word the caller cleared) receives the bit pattern of max |g| by an integer atomicMax -- exact and
 * order-independent.  dw comes in state_dict order.  Only the z-sliding 3x3x3 kernel stages all of g
 * (tem_conv3d_wgrad_gmax_ok: 3x3x3, D >= 8, Cin, Cout % 32 == 0, use_mfma == 2).  Consumer: tem_conv3d_fwd_gscaled. */
int tem_conv3d_wgrad_gmax_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
int tem_conv3d_wgrad_gmax(const float* x, int64_t x_ld, const float* scale, const float* shift,
                          const float* g, int64_t g_ld, const float* w, const float* gamma, const float* beta,
                          float* dw, float* db, float* norm_sums, unsigned* g_amax, void* ws, int64_t ws_bytes,
                          int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma,
                          tem_stream_t stream);

/* Weight gradient (the wgrad half of convolution_backward behind nn.Conv3d, /root/reference/torch_em/model/unet.py:
 * 429-438) in the "fp16 2x1" arithmetic: xhat = hi + lo in two fp16 terms (16-bit class: xhat is a normalised activation),
 * g rounded to ONE fp16 term after the power-of-two prescale that puts *g_amax (device word: bit pattern of max |g|, from
 * tem_absmax or a producer of g) into [2^14, 2^15) -- TWO v_mfma_f32_32x32x16_f16 per product instead of the three of
 * the split-bf16 mode, fp32 accumulation, the result multiplied by the inverse power (exact).  Every dw entry carries
 * the random rounding of an 11-bit g: ~2e-4 relative (scripts/backward_arith_sim.py), unbiased.  dw in state_dict
 * order; w / gamma / beta / norm_sums as in tem_conv3d_wgrad_sums (NULL: plain weight gradient).  Only the z-sliding
 * 3x3x3 kernel (tem_conv3d_wgrad_gscaled_ok: 3x3x3, D >= 8, Cin, Cout % 32 == 0); workspace tem_conv3d_wgrad_ws(.., 8). */
int tem_conv3d_wgrad_gscaled_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw);
/* Does tem_conv3d_wgrad_ex honour a CHUNK STRIDE x_cs (the planar halves of a 16-bit concat buffer: 32-channel chunk k of a
 * voxel at base + k * x_cs + voxel * x_ld) for this layer, storage type `st` (TEM_ST_*) and stride?  Exactly the conditions
 * the launch checks: the transposing z-sliding kernel (3x3x3, D >= 8, Cin, Cout % 32 == 0, option wgrad_zs = 3), 16-bit
 * storage, x_cs % 8 == 0.  A caller lays a tensor out in planes only when this (and the forward / data-gradient queries)
 * say yes -- a launch that cannot honour a stride returns TEM_EINVAL before anything is enqueued. */
int tem_conv3d_wgrad_cs_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int st, int64_t x_cs);
int tem_conv3d_wgrad_gscaled(const float* x, int64_t x_ld, const float* scale, const float* shift,
                             const float* g, int64_t g_ld, const float* w, const float* gamma, const float* beta,
                             float* dw, float* db, float* norm_sums, const unsigned* g_amax, void* ws, int64_t ws_bytes,
                             int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, tem_stream_t stream);
/* ---- BY-PRODUCTS of a call, as explicit arguments (round 5; rounds 3-4 attached them to "the calling thread's next
 * launch" through tem_arm_* / tem_disarm_* -- those entry points are gone) ------------------------------------------------
 * A caller that wants one fills the matching fields of a TemByproducts (the others zero), passes it to the *_ex entry point
 * and reads `delivered` afterwards: the call sets the bit of every by-product it wrote and leaves the others untouched (the
 * caller then runs the separate stage).  Never affects the values of the call's main outputs.
 *
 *  TEM_BP_OUT_AMAX   out_amax (device word, cleared by the caller) receives the bit pattern of max |y| of the tensor the call
 *                    writes (integer atomicMax; saves tem_absmax's pass) -- tem_conv3d_fwd_ex on the 1x1x1 streaming /
 *                    expanding kernels and on the z-reuse kernel with a ReLU mask or ref_coef.  tem_maxpool3d_bwd_st,
 *                    tem_norm_bwd_st and tem_conv1x1_out_bwd_st take the word as a plain argument and always deliver.
 *  TEM_BP_NORM_COEF  tem_conv3d_wgrad_ex with norm_sums: when C / coef_G is a power of two <= 32 the call also writes
 *                    coef [N][C][4] -- bit for bit what tem_norm_bwd_coef(sums = norm_sums, dgamma = dbeta = NULL) would
 *                    (one launch less per layer).  dgamma / dbeta are not part of it.
 *  TEM_BP_NORM_SUMS  tem_conv3d_fwd_ex as a data gradient on the z-reuse kernel with split input channels
 *                    (tem_conv3d_fwd_kernel() == 4: the 16^3 / 8^3 levels): its split-K epilogue also writes the FIRST stage
 *                    of the backward of the norm the gradient lands behind -- sums_part [N][sums_nblk][C][2] rows of
 *                    (sum gy, sum gy * xn), sums_nblk = tem_conv3d_fwd_stat_blocks() of the launch, sums_x = that norm's
 *                    input [N*V][sums_x_ld] (element type of y), its mean / rstd [N][sums_G].  Feed the rows to
 *                    tem_norm_bwd_from_partials / tem_norm_bwd_st. */
#define TEM_BP_OUT_AMAX 1u
#define TEM_BP_NORM_COEF 2u
#define TEM_BP_NORM_SUMS 4u
typedef struct TemByproducts {
    unsigned* out_amax;
    int coef_G;
    const float* coef_mean;
    const float* coef_rstd;
    float* coef;
    const void* sums_x;
    int64_t sums_x_ld;
    const float* sums_mean;
    const float* sums_rstd;
    int sums_G;
    float* sums_part;
    int64_t sums_nblk;
    unsigned delivered; /* out: TEM_BP_* bits */
} TemByproducts;
/* tem_conv3d_fwd / _fwd_gscaled / _fwd_refnorm in one entry point with every variant as an argument:
 *   in_amax  != NULL: tem_conv3d_fwd_gscaled (use_mfma must be 4; scale / shift / bias NULL, act none);
 *   ref_coef != NULL: tem_conv3d_fwd_refnorm (ref required; scale / shift / bias NULL, act none);
 *   stat_part != NULL: tem_conv3d_fwd_stats (stat_blocks = tem_conv3d_fwd_stat_blocks() of the launch);
 *   bp       != NULL: by-products, see above (NULL: none);
 *   x_cs / y_cs != 0: CHUNK strides in elements (16-bit tensors, use_mfma 5 / 7, launches tem_conv3d_fwd_kernel() reports as
 *                     3): the 32-channel chunk k of a voxel lives at x + k * x_cs + voxel * x_ld (y likewise) instead of
 *                     x + k * 32 -- the two halves of a 2 x 32-channel concat as two DENSE planes, so that the kernels which
 *                     read one half move whole 128-byte lines (DESIGN.md 6.R5 "half lines").  A launch that cannot honour
 *                     them returns TEM_EINVAL before anything is enqueued. */
int tem_conv3d_fwd_ex(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w_packed,
                      const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws, int64_t ws_bytes,
                      int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act, int use_mfma,
                      const unsigned* in_amax, const float* ref_coef, float* stat_part, int64_t stat_blocks, int64_t x_cs,
                      int64_t y_cs, TemByproducts* bp, tem_stream_t stream);
/* tem_conv3d_wgrad (sd_layout = 1) / _wgrad_sums / _wgrad_gmax / _wgrad_gscaled in one entry point:
 *   norm_sums  != NULL: tem_conv3d_wgrad_sums (w and db required; gamma / beta of the norm or NULL);
 *   g_amax_out != NULL: tem_conv3d_wgrad_gmax;   g_amax_in != NULL: tem_conv3d_wgrad_gscaled (use_mfma must be 8);
 *   bp         != NULL: TEM_BP_NORM_COEF (needs norm_sums);
 *   x_cs       != 0:    chunk stride of x as in tem_conv3d_fwd_ex (16-bit tensors on the z-sliding 3x3x3 kernel only). */
int tem_conv3d_wgrad_ex(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g, int64_t g_ld,
                        const float* w, const float* gamma, const float* beta, float* dw, float* db, float* norm_sums,
                        const unsigned* g_amax_in, unsigned* g_amax_out, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                        int Cin, int Cout, int kd, int kh, int kw, int use_mfma, int64_t x_cs, TemByproducts* bp, tem_stream_t stream);
int tem_norm_bwd_from_partials(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld, int N, int64_t V, int C, int G,
                               const float* gamma, const float* mean, const float* rstd, int relu_mask, float* gx, int64_t gx_ld,
                               float* dgamma, float* dbeta, const float* part, int64_t nblk, float* coef, void* ws,
                               int64_t ws_bytes, tem_stream_t stream);
/* *amax = max(*amax, bit pattern of max |x|) over nvox rows of C floats (row stride ld): integer atomicMax, exact and
 * order-independent; the caller clears the word.  The prescale source of tem_conv3d_wgrad_gscaled / tem_conv3d_fwd_gscaled
 * when no producer of the tensor delivered it (no reference counterpart: torch.autocast has no per-tensor scale). */
int tem_absmax(const float* x, int64_t ld, int C, int64_t nvox, unsigned* amax, tem_stream_t stream);

/* Data gradient of nn.Conv3d (the dgrad half of convolution_backward behind model/unet.py:417-438) with fp32-class
 * products on an UNNORMALISED input: tem_conv3d_fwd with use_mfma = 4 (two fp16 terms per operand, 22 mantissa bits)
 * on w_packed = tem_conv_pack_weights(transpose = 1, use_mfma = 4), no bias / norm / activation, where the kernel first
 * multiplies x by the power of two that puts *in_amax (bit pattern of max |x|, e.g. from tem_conv3d_wgrad_gmax) into
 * [2^14, 2^15) and divides the result by it: exact, and nothing can leave fp16's range.  Only for launches that
 * tem_conv3d_fwd_kernel(..., 4) reports as 3. */
int tem_conv3d_fwd_gscaled(const float* x, int64_t x_ld, const float* w_packed, float* y, int64_t y_ld,
                           const float* ref, int64_t ref_ld, const unsigned* in_amax, void* ws, int64_t ws_bytes,
                           int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, tem_stream_t stream);
/* tem_conv3d_fwd of a data gradient that lands behind a ReLU + norm (the first conv of a block, seen from the second's
 * backward): y = ref > 0 ? a*conv(x) - m1 - (ref - mean)*m2r : 0 with coef[N][Cout][4] = (a, m1, m2r, mean) from
 * tem_norm_bwd_coef.  The epilogue of the z-reuse kernel replaces the elementwise pass of tem_norm_bwd_from_sums (the
 * reference: autograd's native_layer_norm / relu backward kernels).  Only for launches tem_conv3d_fwd_kernel() == 3. */
int tem_conv3d_fwd_refnorm(const float* x, int64_t x_ld, const float* w_packed, float* y, int64_t y_ld,
                           const float* ref, int64_t ref_ld, const float* coef, void* ws, int64_t ws_bytes,
                           int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma,
                           tem_stream_t stream);

/* tem_conv3d_wgrad of a first layer (Cin <= 4, VALU kernel) whose g is still the RAW data gradient behind the norm that
 * follows this conv's ReLU (the second norm of the first ConvBlock, model/unet.py:429-438): the norm backward
 * g := (y > 0) ? a*g - m1 - (y - mean)*m2r : 0 (gcoef[N][Cout][4] from tem_norm_bwd_coef, y = this conv's output) is
 * applied while g is loaded, instead of a pass that rewrites g.  Only where tem_conv3d_wgrad_gnorm_ok() != 0. */
int tem_conv3d_wgrad_gnorm_ok(int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
int tem_conv3d_wgrad_gnorm(const float* x, int64_t x_ld, const float* scale, const float* shift,
                           const float* g, int64_t g_ld, const float* y, int64_t y_ld, const float* gcoef,
                           float* dw, float* db, void* ws, int64_t ws_bytes,
                           int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                           int sd_layout, tem_stream_t stream);
/* ... for x of storage type st_x and g / y of storage type st_g (TEM_ST_*) */
int tem_conv3d_wgrad_gnorm_st(const void* x, int64_t x_ld, const float* scale, const float* shift, const void* g, int64_t g_ld,
                              const void* y, int64_t y_ld, const float* gcoef, float* dw, float* db, void* ws, int64_t ws_bytes,
                              int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int sd_layout, int st_x,
                              int st_g, tem_stream_t stream);

/* tem_conv3d_wgrad that ALSO delivers the first stage of the backward of the norm in front of this conv -- per (sample,
 * input channel) sums[n][ci] = (sum_v gz, sum_v gz * xn), gz = the data gradient of this conv (tem_conv3d_fwd with the
 * transposed pack), xn = the normalised input -- WITHOUT reading gz or x: sum_v gz*z = sum_{tap,co} w * dw_n (the
 * per-sample weight gradient, which the slab merge has anyway) and sum_v gz = sum_{tap,co} w * T_n[tap][co] with T_n the
 * per-sample bias gradient minus boundary faces (csrc/wgrad_sums.hip).  Replaces the reduction pass of
 * aten::native_group_norm_backward / native_batch_norm_backward for nn.InstanceNorm3d / nn.GroupNorm (reference
 * model/unet.py:391-406, 429-438).  w: the weights in state_dict layout; gamma/beta: the norm's affine parameters or
 * NULL; dw is written in state_dict layout; db is required.  Only where tem_conv3d_wgrad_sums_ok() != 0 (z-sliding
 * 3x3x3 kernel, N <= 4, Cout <= 128); workspace: tem_conv3d_wgrad_ws().  Feed `norm_sums` to tem_norm_bwd_from_sums. */
int tem_conv3d_wgrad_sums_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma);
int tem_conv3d_wgrad_sums(const float* x, int64_t x_ld, const float* scale, const float* shift,
                          const float* g, int64_t g_ld, const float* w, const float* gamma, const float* beta,
                          float* dw, float* db, float* norm_sums, void* ws, int64_t ws_bytes,
                          int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                          int use_mfma, tem_stream_t stream);

/* ---- normalisation ------------------------------------------------------
 * Replaces nn.InstanceNorm3d / nn.GroupNorm from get_norm_layer
 * (model/unet.py:391-406): statistics per (sample, group) over V*(C/G) values,
 * biased variance, eps inside the sqrt.  InstanceNorm == G = C, no affine.
 * Outputs: mean[N][G], rstd[N][G] and the fused per-channel affine
 *   scale[N][C] = rstd*gamma, shift[N][C] = beta - mean*rstd*gamma
 * that tem_conv3d_fwd / tem_conv3d_wgrad apply while loading x.
 */
int64_t tem_norm_ws(int N, int64_t V, int C);
int tem_norm_stats(const float* x, int64_t x_ld, int N, int64_t V, int C, int G,
                   const float* gamma, const float* beta, float eps,
                   float* mean, float* rstd, float* scale, float* shift,
                   void* ws, int64_t ws_bytes, tem_stream_t stream);
/* Second stage alone, for statistics whose first stage was fused into the producer (tem_conv3d_fwd_stats):
 * part = [N][nblk][C][2] partial sums (sum x, sum x^2) over disjoint voxel blocks; same outputs as tem_norm_stats.
 * BatchNorm: call with N = 1, nblk = N*blocks, V = N*V (the partial layout is contiguous over samples). */
int tem_norm_finalize_partials(const float* part, int64_t nblk, int N, int64_t V, int C, int G,
                               const float* gamma, const float* beta, float eps,
                               float* mean, float* rstd, float* scale, float* shift, tem_stream_t stream);
/* ... for a tensor concatenated from two producers along the channels (Decoder._concat, model/unet.py:363-373): channels
 * [0, CA) are summarised by partA [N][nblkA][CA][2] (tem_upsample_stats), channels [CA, C) by partB [N][nblkB][C-CA][2]
 * (tem_conv3d_fwd_stats of the skip tensor); a group must lie inside one half. */
int tem_norm_finalize_partials2(const float* partA, int64_t nblkA, int CA, const float* partB, int64_t nblkB,
                                int N, int64_t V, int C, int G, const float* gamma, const float* beta, float eps,
                                float* mean, float* rstd, float* scale, float* shift, tem_stream_t stream);
/* Backward of y = norm(x)*gamma+beta given gy:
 *   gx = rstd*(gy*gamma - mean_grp(gy*gamma) - xn*mean_grp(gy*gamma*xn)),  xn=(x-mean)*rstd
 *   [gx *= (x > 0) when relu_mask != 0: x is itself a ReLU output]
 *   dgamma[c] = sum_{n,v} gy*xn, dbeta[c] = sum_{n,v} gy   (optional, accumulate==0 overwrites)
 */
int tem_norm_bwd(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld,
                 int N, int64_t V, int C, int G, const float* gamma,
                 const float* mean, const float* rstd, int relu_mask,
                 float* gx, int64_t gx_ld, float* dgamma, float* dbeta,
                 void* ws, int64_t ws_bytes, tem_stream_t stream);

/* tem_norm_bwd with the reduction stage replaced by sums[N][C][2] from tem_conv3d_wgrad_sums */
int tem_norm_bwd_from_sums(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld,
                           int N, int64_t V, int C, int G, const float* gamma,
                           const float* mean, const float* rstd, int relu_mask,
                           float* gx, int64_t gx_ld, float* dgamma, float* dbeta,
                           const float* sums, void* ws, int64_t ws_bytes, tem_stream_t stream);

/* Reduction stage of tem_norm_bwd only: coef[n][c] = {a, m1, m2r, mean}, gx = a*gy - m1 - (x - mean)*m2r (plus dgamma /
 * dbeta).  For the norm in front of a decoder block (its input is the concat of an upsampled tensor and a skip tensor,
 * reference Decoder._concat model/unet.py:363-373) the elementwise stage is then applied by the two kernels that read
 * the gradient next -- tem_upsample_bwd_norm and tem_maxpool3d_bwd_norm -- and the 3-tensor pass over the concat
 * buffer disappears.  sums: optional first stage from tem_conv3d_wgrad_sums. */
int tem_norm_bwd_coef(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld,
                      int N, int64_t V, int C, int G, const float* gamma,
                      const float* mean, const float* rstd, float* dgamma, float* dbeta,
                      const float* sums, float* coef, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* tem_norm_stats for a tensor of storage type st (TEM_ST_*) */
int tem_norm_stats_st(const void* x, int64_t x_ld, int N, int64_t V, int C, int G, const float* gamma, const float* beta,
                      float eps, float* mean, float* rstd, float* scale, float* shift, void* ws, int64_t ws_bytes, int st,
                      tem_stream_t stream);
/* tem_norm_bwd / _from_sums / _from_partials / _coef in one entry point, for tensors of storage type st:
 *   part, part_nblk : first stage rows [N][part_nblk][C][2] (sum gy, sum gy * xn) delivered by a producer -- the weight
 *                     gradient (tem_conv3d_wgrad_ex, 1 row) or the data gradient (tem_conv3d_fwd_ex, its nblk rows);
 *                     NULL = the reduction pass over gy and x runs here
 *   coef_out        : non-NULL = reduction stage only, writes coef[N][C][4] = {a, m1, m2r, mean} (gx may be NULL)
 *   out_amax        : optional device word that receives max |gx| (see tem_maxpool3d_bwd_st) */
int tem_norm_bwd_st(const void* gy, int64_t gy_ld, const void* x, int64_t x_ld, int N, int64_t V, int C, int G,
                    const float* gamma, const float* mean, const float* rstd, int relu_mask, void* gx, int64_t gx_ld,
                    float* dgamma, float* dbeta, const float* part, int64_t part_nblk, float* coef_out,
                    unsigned* out_amax, void* ws, int64_t ws_bytes, int st, tem_stream_t stream);


/* ---- pooling / upsampling ------------------------------------------------
 * nn.MaxPool3d(factor) (model/unet.py:300-302,645): kernel == stride == factor.
 * Backward routes the gradient to the first maximum in (z,y,x) scan order (ATen's
 * rule), optionally adds a skip-connection gradient and applies the ReLU mask of x:
 *   gx = [gskip] + scatter(gy) ; gx *= (x > 0) if relu_mask
 */
int tem_maxpool3d_fwd(const float* x, int64_t x_ld, float* y, int64_t y_ld,
                      int N, int D, int H, int W, int C, int fz, int fy, int fx, tem_stream_t stream);
/* tem_maxpool3d_fwd that also writes the first stage of the statistics of y -- stat_part [N][stat_blocks][C][2], one row of
 * per-channel (sum, sum of squares) per output row (zo, yo), stat_blocks = tem_maxpool3d_fwd_stat_blocks() (0: this channel
 * count cannot) -- for the norm in front of the next encoder block's first conv (model/unet.py:311-321, 429-438): feed them to
 * tem_norm_finalize_partials instead of a tem_norm_stats pass over the pooled tensor. */
int64_t tem_maxpool3d_fwd_stat_blocks(int D, int H, int C, int fz, int fy);
int tem_maxpool3d_fwd_stats(const float* x, int64_t x_ld, float* y, int64_t y_ld, int N, int D, int H, int W, int C,
                            int fz, int fy, int fx, float* stat_part, int64_t stat_blocks, tem_stream_t stream);
int tem_maxpool3d_bwd(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld,
                      const float* gskip, int64_t gskip_ld, int relu_mask,
                      float* gx, int64_t gx_ld,
                      int N, int D, int H, int W, int C, int fz, int fy, int fx, tem_stream_t stream);
/* ... where gskip is still the RAW data gradient behind a norm whose input is x: gskip := a*gskip - m1 - (x - mean)*m2r
 * with gcoef = tem_norm_bwd_coef()'s rows for these C channels (gcoef_ld floats per sample) -- see tem_norm_bwd_coef.
 * ycoef (optional, dense [N][C][4]; gcoef may then be NULL): gy is raw as well -- the gradient behind the norm whose
 * input is the POOLED tensor (first norm of the next level's block); x there is the maximum the kernel recomputes. */
int tem_maxpool3d_bwd_norm(const float* gy, int64_t gy_ld, const float* x, int64_t x_ld,
                           const float* gskip, int64_t gskip_ld, int relu_mask,
                           float* gx, int64_t gx_ld,
                           int N, int D, int H, int W, int C, int fz, int fy, int fx,
                           const float* gcoef, int64_t gcoef_ld, const float* ycoef, tem_stream_t stream);
/* F.interpolate(mode="trilinear"/"bilinear", align_corners=False, integer scale
 * factors) (model/unet.py:456).  (D,H,W) are the INPUT dims; output is (D*fz,H*fy,W*fx).
 * bwd is the exact adjoint (upsample_trilinear3d_backward). */
int tem_upsample_fwd(const float* x, int64_t x_ld, float* y, int64_t y_ld,
                     int N, int D, int H, int W, int C, int fz, int fy, int fx, tem_stream_t stream);
int tem_upsample_bwd(const float* gy, int64_t gy_ld, float* gx, int64_t gx_ld,
                     int N, int D, int H, int W, int C, int fz, int fy, int fx, tem_stream_t stream);
/* First stage of the statistics of y = upsample(u) from the LOW-RESOLUTION u alone (the interpolation U is linear:
 * sum y = sum_i (U^T 1)[i] u[i], sum y^2 = sum_i u[i] (U^T U u)[i]): part [N][D*H][C][2], one block per input row.
 * (D,H,W) are u's dims; C = 4 * 2^k <= 256.  Merged by tem_norm_finalize_partials2 with V = D*fz*H*fy*W*fx. */
int tem_upsample_stats(const float* u, int64_t u_ld, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                       float* part, tem_stream_t stream);
/* tem_upsample_fwd for factors (1|2, 2, 2) that also returns the first stage of y's statistics, part [N][D*H][C][2]
 * (sum y, sum y^2 per coarse row: the layout of tem_upsample_stats, whose launch it saves -- the outputs are in
 * registers).  tem_upsample_fwd_stats_ok: 1 when (C, factors) are taken (C = 4 * 2^k <= 256), else 0. */
int tem_upsample_fwd_stats_ok(int C, int fz, int fy, int fx);
int tem_upsample_fwd_stats(const float* x, int64_t x_ld, float* y, int64_t y_ld, int N, int D, int H, int W, int C,
                           int fz, int fy, int fx, float* part, tem_stream_t stream);
/* ... of the RAW data gradient gy behind a norm whose input was upsample(u) (u: the low-resolution tensor, same shape
 * as gx): gx = a*U^T gy - m1*U^T 1 - m2r*(U^T U u - mean*U^T 1), ncoef as for tem_maxpool3d_bwd_norm */
int tem_upsample_bwd_norm(const float* gy, int64_t gy_ld, float* gx, int64_t gx_ld,
                          int N, int D, int H, int W, int C, int fz, int fy, int fx,
                          const float* u, int64_t u_ld, const float* ncoef, int64_t ncoef_ld, tem_stream_t stream);

/* The same operations for tensors of storage type st (TEM_ST_*): one entry point per operation, the optional by-products as
 * explicit arguments (NULL = not wanted).
 *   tem_maxpool3d_fwd_st   = tem_maxpool3d_fwd / _fwd_stats (stat_part != NULL)
 *   tem_maxpool3d_bwd_st   = tem_maxpool3d_bwd / _bwd_norm (gcoef / ycoef != NULL); out_amax (optional): device word that
 *                            receives max |gx| as an integer atomicMax of the bit patterns (exact, order-independent) -- the
 *                            weight gradient that reads gx next takes its fp16 prescale from it (tem_conv3d_wgrad_ex)
 *   tem_upsample_fwd_st    = tem_upsample_fwd / _fwd_stats (part != NULL)
 *   tem_upsample_bwd_st    = tem_upsample_bwd / _bwd_norm (u, ncoef != NULL)
 *   tem_upsample_stats_st  = tem_upsample_stats */
int tem_maxpool3d_fwd_st(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int D, int H, int W, int C,
                         int fz, int fy, int fx, float* stat_part, int64_t stat_blocks, int st, tem_stream_t stream);
int tem_maxpool3d_bwd_st(const void* gy, int64_t gy_ld, const void* x, int64_t x_ld, const void* gskip, int64_t gskip_ld,
                         int relu_mask, void* gx, int64_t gx_ld, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                         const float* gcoef, int64_t gcoef_ld, const float* ycoef, unsigned* out_amax, int st,
                         tem_stream_t stream);
int tem_upsample_fwd_st(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int D, int H, int W, int C,
                        int fz, int fy, int fx, float* part, int st, tem_stream_t stream);
int tem_upsample_bwd_st(const void* gy, int64_t gy_ld, void* gx, int64_t gx_ld, int N, int D, int H, int W, int C,
                        int fz, int fy, int fx, const void* u, int64_t u_ld, const float* ncoef, int64_t ncoef_ld, int st,
                        tem_stream_t stream);
int tem_upsample_stats_st(const void* u, int64_t u_ld, int N, int D, int H, int W, int C, int fz, int fy, int fx,
                          float* part, int st, tem_stream_t stream);

/* ---- Dice ------------------------------------------------------------------
 * dice_score / DiceLoss (loss/dice.py:34-133) and the masked variant
 * LossWrapper + ApplyAndRemoveMask("multiply") (loss/wrapper.py:84-87,129-152).
 * Generic strides (in floats): element (n,c,v) at n*sn + c*sc + v*sv, so NCDHW
 * targets and NDHWC predictions are read in place (no flatten_samples copy).
 * sums[c] = { sum p*t, sum p*p, sum t*t } over (n,v) with p,t already multiplied by
 * mask when mask != NULL.  (mask uses the target strides.)
 */
int64_t tem_dice_ws(int N, int64_t V, int C);
int tem_dice_sums(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv,
                  const float* t, int64_t t_sn, int64_t t_sc, int64_t t_sv,
                  const float* mask, int N, int C, int64_t V,
                  double* sums /*[C][3]*/, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* sums -> score (loss/dice.py:65-84) on device, no host sync:
 *   score_c = 2*num_c/max(den_c,eps); out_c = invert ? 1-score_c : score_c;
 *   channelwise==0 pools all channels first; reduce: 0 none (out[C]), 1 sum, 2 mean, 3 max, 4 min.
 * Also emits ca[C], cb[C] with  d out / d p[n,c,v] = ca[c]*t + cb[c]*p  (clamp gradient included). */
int tem_dice_finalize(const double* sums, int C, double eps, int channelwise, int invert, int reduce,
                      float* out, float* ca, float* cb, tem_stream_t stream);
/* gp[n,c,v] = gout * (ca[c]*t + cb[c]*p) * mask, with p,t masked as above; gout: device scalar
 * (or [C] when gout_per_channel), NULL = 1. */
int tem_dice_grad(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv,
                  const float* t, int64_t t_sn, int64_t t_sc, int64_t t_sv,
                  const float* mask, const float* ca, const float* cb,
                  const float* gout, int gout_per_channel,
                  float* gp, int64_t g_sn, int64_t g_sc, int64_t g_sv,
                  int N, int C, int64_t V, tem_stream_t stream);

/* The logits / BCE members of the Dice family (reference loss/dice.py:136-253: DiceLossWithLogits, BCEDiceLoss,
 * BCEDiceLossWithLogits) on the same kernels.  flags: TEM_DICE_LOGITS -- p holds logits, the Dice terms use sigmoid(p);
 * TEM_DICE_BCE -- sums gets a 4th column per channel with the summed binary cross entropy (F.binary_cross_entropy's log
 * clamp at -100; with TEM_DICE_LOGITS F.binary_cross_entropy_with_logits).  tem_dice_finalize2 reads sums with ncol
 * columns; tem_dice_grad2 returns gout * (w_dice * dDice/dp + w_bce * dBCEsum/dp) (w_bce = beta / element count). */
#define TEM_DICE_LOGITS 1
#define TEM_DICE_BCE 2
int tem_dice_sums2(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn, int64_t t_sc,
                   int64_t t_sv, int N, int C, int64_t V, double* sums, void* ws, int64_t ws_bytes, int flags,
                   tem_stream_t stream);
int tem_dice_finalize2(const double* sums, int ncol, int C, double eps, int channelwise, int invert, int reduce,
                       float* out, float* ca, float* cb, tem_stream_t stream);
int tem_dice_grad2(const float* p, int64_t p_sn, int64_t p_sc, int64_t p_sv, const float* t, int64_t t_sn, int64_t t_sc,
                   int64_t t_sv, const float* ca, const float* cb, const float* gout, int gout_per_channel, float* gp,
                   int64_t g_sn, int64_t g_sc, int64_t g_sv, int N, int C, int64_t V, int flags, float w_dice,
                   float w_bce, tem_stream_t stream);

/* ---- optimizer -------------------------------------------------------------
 * torch.optim.AdamW step as configured by default_segmentation_trainer
 * (segmentation.py:543): decoupled weight decay, bias correction, eps outside
 * the sqrt; one launch over a flat parameter arena.  `step` is the 1-based step
 * count AFTER the increment.  grad_scale multiplies the gradient first (1/world
 * for data-parallel SUM all-reduce). */
int tem_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                   float grad_scale, tem_stream_t stream);
/* The same step for HIP-graph capture: a captured launch has its kernel arguments frozen, but the step count (bias
 * corrections) and the learning rate change every step.  tem_adamw_hyper fills a 12-float HOST buffer
 * [lr, beta1, beta2, eps, weight_decay, lr/bc1, 1/sqrt(bc2), grad_scale, skip, 0, 0, 0] exactly as tem_adamw_step
 * derives those values; the caller copies it to the device before each replay and tem_adamw_step_dev reads it there
 * (skip != 0: no update -- an overflowed mixed-precision step).  Bit-identical to tem_adamw_step. */
int tem_adamw_hyper(float* hyper_host, float lr, float beta1, float beta2, float eps, float weight_decay,
                    int64_t step, float grad_scale);
int tem_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* hyper, tem_stream_t stream);
/* ... under dynamic loss scaling (torch.amp.GradScaler; reference trainer/default_trainer.py:789-794) the device may skip
 * a step (overflow) before the host has enqueued the next one.  The scaler state lives on the device,
 * sstate = [scale, growth_tracker, found_inf, applied_steps]; tem_amp_unscale_dev / tem_amp_update_dev are
 * GradScaler.unscale_ / update on it; tem_adamw_step_tab takes its scalars from row (applied_steps + 1 - lo) of
 * table = [lo, J, -, -] + J x 12 floats (row j = tem_adamw_hyper for step lo + j) and does nothing when found_inf != 0.
 * A row outside the window (the caller's step count lagged by more than J - 1) is NOT clamped: nothing is updated and
 * found_inf is set, so the step counts as skipped instead of running with another step's bias corrections. */
int tem_adamw_step_tab(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const float* table, float* sstate, tem_stream_t stream);
int tem_amp_unscale_dev(float* grad, int64_t n, float* sstate, tem_stream_t stream);
int tem_amp_update_dev(float* sstate, float growth, float backoff, int interval, tem_stream_t stream);
/* theta_k = m*theta_k + (1-m)*theta_q : SPOCOTrainer._momentum_update (trainer/spoco_trainer.py:45-47) */
int tem_ema_update(float* theta_k, const float* theta_q, int64_t n, float momentum, tem_stream_t stream);
/* Mixed-precision training (reference trainer/default_trainer.py:134-142,789-794: torch.amp.GradScaler): the unscale_
 * step -- grad *= inv_scale in place, *found_inf (device float, zeroed by the caller) is set to 1 if any element is
 * inf/NaN.  Replaces torch._amp_foreach_non_finite_check_and_unscale_ over the flat gradient arena. */
int tem_amp_unscale(float* grad, int64_t n, float inv_scale, float* found_inf, tem_stream_t stream);

/* ---- label targets (integer, bit-exact) -----------------------------------------
 * BoundaryTransform (transform/label.py:100-129; skimage find_boundaries mode="thick":
 * a voxel is boundary iff any face neighbour inside the volume has a different label)
 * and AffinityTransform (transform/label.py:248-327; semantics pinned by
 * test/transform/test_label_transforms.py:5-55).  labels: int64 [D][H][W] on device.
 * out: float32 channels [C][D][H][W] (the reference's output layout).
 */
int tem_boundary_target(const int64_t* labels, float* out, int D, int H, int W,
                        int add_binary_target, tem_stream_t stream);
/* find_boundaries' other same-shape modes: mode 0 "thick", 1 "inner" (thick & label != 0), 2 "outer" (thick &
 * (background | two objects touch in the full 3^ndim window)); BoundaryTransform(mode=...) transform/label.py:108,123.
 * "subpixel" changes the output shape to 2n-1 and cannot be a training target: not provided. */
int tem_boundary_target_mode(const int64_t* labels, float* out, int D, int H, int W,
                             int add_binary_target, int mode, tem_stream_t stream);
/* offsets: HOST array [n_off][3] (z,y,x); 2-D data: D==1 and z offset 0.
 * out channels: [binary?] + n_off affinities (1 = different/invalid) [+ (binary mask?) + n_off mask].
 * has_ignore==0: no ignore label (mask = in-bounds only). */
int tem_affinity_target(const int64_t* labels, float* out, int D, int H, int W,
                        const int* offsets, int n_off, int has_ignore, int64_t ignore_label,
                        int add_binary_target, int add_mask, int include_ignore_transitions,
                        tem_stream_t stream);

/* ---- SPOCO / contrastive embedding losses --------------------
 * Embeddings: float [E][V] planes of ONE sample, voxel-fastest, channel stride cs (>= V); labels int64 [V],
 * consecutive ids 0..C-1.  E <= 32.  Per-slice Dice terms see the volume as [nz][V/nz] (nz = first spatial
 * axis, which is DiceLoss()'s channel axis for the reference's [1|A, *spatial] pmaps).  Every *_out / means /
 * counts pointer is DEVICE memory; scalars come back as device floats (no host sync inside the library).
 * Segment sums use 64-bit fixed point (2^-28) so results are bit-reproducible.  ws: tem_spoco_ws bytes. */
int64_t tem_spoco_ws(int C, int E, int64_t V, int nz, int n_anchors, int n_offsets);
/* out_minmax[2] (device) = {min, max} label: C = max+1; reference asserts min == 0 (loss/spoco_loss.py:30-31) */
int tem_label_range(const int64_t* labels, int64_t V, int64_t* out_minmax, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* compute_cluster_means (loss/spoco_loss.py:16-33; torch_scatter.scatter_mean, contrastive_impl.py:14-25):
 * means [C][E], counts [C] (as float; empty label -> mean 0, count 0). */
int tem_spoco_cluster_means(const float* emb, int64_t cs, const int64_t* labels, int64_t V, int E, int C,
                            float* means, float* counts, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* variance (pull) term, contrastive_impl.py:86-129: value_out[0] = sum_v (|e_v-mu_l|-delta_var)+^2 / count_l
 * (caller divides by the instance count); S_out [C][E] = sum_{v in l} h_v * unit(e_v - mu_l) for the backward. */
int tem_spoco_pull(const float* emb, int64_t cs, const int64_t* labels, int64_t V, int E, int C, const float* means,
                   const float* counts, float delta_var, float* value_out, float* S_out, void* ws, int64_t ws_bytes,
                   tem_stream_t stream);
/* distance (push) term contrastive_impl.py:28-80 and regulariser spoco_loss.py:205-213 on the [C][E] means:
 * values_out2 = {distance term, regulariser}; ddist/dreg [C][E] = their gradients wrt the means. */
int tem_spoco_means_terms(const float* means, int C, int E, float delta_dist, int ignore_zero, float* values_out2,
                          float* ddist, float* dreg, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* instance Dice term, spoco_loss.py:386-430 + GaussianKernel :85-95 (value only: the reference detaches it at :422):
 * mean over instances i>=1 of sum_z (1 - dice(exp(-|e-mu_i|^2/two_sigma), label==i)). */
int tem_spoco_instance_dice(const float* emb, int64_t cs, const int64_t* labels, int64_t V, int nz, int E, int C,
                            const float* means, float two_sigma, float eps, float* value_out, void* ws, int64_t ws_bytes,
                            tem_stream_t stream);
/* unlabeled push, spoco_loss.py:162-190: value_out[0]; grad (nullable) += grad_scale * d push/d emb;
 * dpush [C][E] = d push / d means.  Needs C > 1; counts[0] = number of background voxels. */
int tem_spoco_push(const float* emb, int64_t cs, const int64_t* labels, int64_t V, int E, int C, const float* means,
                   const float* counts, float delta_dist, float* value_out, float grad_scale, float* grad, int64_t gcs,
                   float* dpush, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* d/d emb of  w_var*variance + w_dist*distance + w_reg*regulariser + w_push*push(means part), chained through the
 * means: grad (=|+=) w_var*2h/(n_inst*count_l)*unit + dmu_l/count_l.  dpush may be NULL. */
int tem_spoco_embed_grad(const float* emb, int64_t cs, const int64_t* labels, int64_t V, int E, int C,
                         const float* means, const float* counts, const float* S, const float* ddist, const float* dreg,
                         const float* dpush, float delta_var, float n_inst, float w_var, float w_dist, float w_reg,
                         float w_push, float* grad, int64_t gcs, int accumulate, void* ws, int64_t ws_bytes,
                         tem_stream_t stream);
/* unlabeled-voxel bookkeeping for the consistency anchors (spoco_loss.py:509-514): chunk_counts has
 * ceil(V/1024) ints; total_out[0] (device) = number of label==0 voxels; tem_zero_select maps ranks (device int64
 * [A], each < total) to flat voxel indices in row-major order (the order of torch.nonzero). */
int tem_zero_count(const int64_t* labels, int64_t V, int* chunk_counts, int64_t* total_out, tem_stream_t stream);
int tem_zero_select(const int64_t* labels, int64_t V, const int* chunk_counts, const int64_t* ranks, int A,
                    int64_t* idx_out, tem_stream_t stream);
/* embedding consistency, spoco_loss.py:503-527: Dice between the q and k Gaussian pmaps of A anchors (A <= 64);
 * value_out[0]; grad_q (nullable) += grad_scale * d/d emb_q (including the anchor voxels). */
int tem_spoco_consistency(const float* emb_q, const float* emb_k, int64_t cs, int64_t V, int nz, int E,
                          const int64_t* anchor_idx, int A, float two_sigma, float eps, float* value_out,
                          float grad_scale, float* grad_q, int64_t gcs, void* ws, int64_t ws_bytes, tem_stream_t stream);
/* AffinitySideLoss, loss/affinity_side_loss.py:92-172 for already-drawn offsets (HOST int [K][3] z,y,x; K <= 32;
 * 2-D data: D == 1, z offset 0): value_out[0] = sum_k (1 - dice_k); grad (nullable) += grad_scale * d/d emb. */
int tem_affinity_side(const float* emb, int64_t cs, const int64_t* labels, int D, int H, int W, int E,
                      const int* offsets_zyx, int K, float delta, float eps, float* value_out, float grad_scale,
                      float* grad, int64_t gcs, void* ws, int64_t ws_bytes, tem_stream_t stream);

/* ---- on-device augmentations (transform/augmentation.py) ----------
 * tem_flip3d: H/V/D flips of the default 3-D pipeline (:254-258) in ONE pass; 4-byte elements, src/dst
 * [N][planes][D][H][W], flags_dev device int [N][3] (z,y,x).  2-D data: D == 1.
 * tem_elastic_field / tem_elastic_warp2d: RandomElasticDeformation[Stacked] (:11-151) = kornia
 * elastic_transform2d restated (kornia is not in this image: parity unpinned, see csrc/augment.hip):
 * noise [2][H][W], gauss1d device [2][ksize] (row 0: sigma[0], row 1: sigma[1]), disp [2][H][W] in normalised
 * grid units; the warp applies the same field to `planes` [H][W] planes (bilinear, or nearest for labels). */
int tem_flip3d(const void* src, void* dst, const int* flags_dev, int N, int planes, int D, int H, int W,
               tem_stream_t stream);
/* tem_affine_warp3d: the resampling pass of kornia RandomAffine3D / RandomRotation3D (transform/augmentation.py:235,240;
 * kornia absent: parity unpinned).  src/dst [N][planes][D][H][W] float; mat_dev device float [N][12]: the 3x4 row-major
 * map from an OUTPUT voxel (x, y, z, 1) to its SOURCE position (sx, sy, sz), composed on the host from the drawn
 * angles / scales; trilinear (nearest != 0: nearest neighbour, for labels), zeros outside the volume. */
int tem_affine_warp3d(const float* src, const float* mat_dev, float* dst, int N, int planes, int D, int H, int W,
                      int nearest, tem_stream_t stream);
int tem_elastic_field(const float* noise, const float* gauss1d, int ksize, int H, int W, float alpha0, float alpha1,
                      float* disp, tem_stream_t stream);
int tem_elastic_warp2d(const float* src, const float* disp, float* dst, int64_t planes, int H, int W, int nearest,
                       tem_stream_t stream);

/* ---- tiled inference with a halo (util/prediction.py:98-330) ----------
 * The volume stays in HBM.  tem_block_load_reflect = `_load_block` (:98-142): the (block + halo) box clipped to the
 * volume is the segment [seg_start, seg_start + seg_len); dst [C][out_shape] is that segment padded by numpy
 * "reflect" (pad_left on the left, the rest on the right).  tem_block_store_inner = the write-back (:270-302):
 * out[c][out_start + i] = pred[c][inner_start + i] for i < size, zero where mask (uint8 [D][H][W], nullable) is 0.
 * src / out: [C][D][H][W] float32; all int arrays are HOST int[3] (z, y, x); 2-D data: D == 1. */
int tem_block_load_reflect(const float* src, float* dst, int C, int D, int H, int W, const int* seg_start,
                           const int* seg_len, const int* pad_left, const int* out_shape, tem_stream_t stream);
int tem_block_store_inner(const float* pred, const int* pred_shape, float* out, int C, int D, int H, int W,
                          const unsigned char* mask, const int* inner_start, const int* out_start, const int* size,
                          tem_stream_t stream);

/* AccumulateChannels (model/unet.py:15-44, export-time post-processing): out [N][(i1-i0)+1][V] contiguous =
 * cat(x[:, i0:i1], reduce(x[:, c0:c1])), mode 0 mean / 1 min / 2 max; x element (n,c,v) at n*sn + c*sc + v*sv. */
int tem_accumulate_channels(const float* x, int64_t sn, int64_t sc, int64_t sv, float* out, int N, int C, int64_t V,
                            int i0, int i1, int c0, int c1, int mode, tem_stream_t stream);

/* ---- small utilities ---------------------------------------------------------- */
/* NCDHW (contiguous) <-> NDHWC(ld) layout change at the module boundary. */
int tem_nchw_to_nhwc(const float* src, float* dst, int64_t dst_ld, int N, int C, int64_t V, tem_stream_t stream);
int tem_nhwc_to_nchw(const float* src, int64_t src_ld, float* dst, int N, int C, int64_t V, tem_stream_t stream);
/* backward of the final activation on a contiguous array: sigmoid gx = gy*y*(1-y); relu gx = gy*(y>0)
 * (UNetBase._get_activation, model/unet.py:162-172) */
int tem_act_bwd(const float* gy, const float* y, float* gx, int64_t n, int act, tem_stream_t stream);
/* per-sample standardize (transform/raw.py:40-65): (x-mean)/(std+eps) over each of N rows of length L */
int tem_standardize(const float* x, float* y, int N, int64_t L, float eps, void* ws, int64_t ws_bytes, tem_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TEM_HIP_H */
