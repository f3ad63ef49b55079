// conv.hip -- convolution entry points (reference nn.Conv3d/nn.Conv2d inside ConvBlock,
// Upsampler.conv and out_conv: model/unet.py:417-438,453,638, and their autograd backward),
// weight (un)packing, and the VALU kernels used where the MFMA path does not apply:
// Cin==1 first layer (HBM-bound, 13 flop/B), the 32->Cout<=16 output projection
// (HBM-bound, 0.9 flop/B) and the small test networks.
#include "tem_common.h"
#include "conv_internal.h"
#include "tem_act.h"

// `use_mfma` of the conv entry points carries the storage types of the call in its high bits (tem_hip.h: TEM_MFMA_STX / _STY);
// every entry point strips them first and keeps them in tem_call_st for the duration of the call
// (a call that arrives WITHOUT storage bits from inside another entry point inherits that call's types)
#define TEM_MODE_SCOPE(use_mfma)                                                                        \
    TemStScope st_scope__(((use_mfma) >> 8) ? (((use_mfma) >> 8) & 15) : tem_call_st.x,                 \
                          ((use_mfma) >> 8) ? (((use_mfma) >> 12) & 15) : tem_call_st.y);               \
    use_mfma &= 0xff

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
extern "C" int64_t tem_conv_packed_size(int Cout, int Cin, int kd, int kh, int kw) {
    // floats; the bf16x6 layout stores 3 bf16 planes = 1.5 floats per weight
    return ((int64_t)Cout * Cin * kd * kh * kw * 3 + 1) / 2;
}

__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, float* __restrict__ dst, int Cout,
                                                      int Cin, int KD, int KH, int KW, int transpose, int layout) {
    const int ntaps = KD * KH * KW;
    const int64_t total = (int64_t)Cout * Cin * ntaps;
    // logical operator: CoutL x CinL
    const int CoutL = transpose ? Cin : Cout, CinL = transpose ? Cout : Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // enumerate logical (tap, ci, co), co fastest
        int co = (int)(i % CoutL);
        int64_t r = i / CoutL;
        int ci = (int)(r % CinL);
        int tap = (int)(r / CinL);
        int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
        float val;
        if (!transpose) {
            val = w[(((int64_t)co * Cin + ci) * KD + tz) * KH * KW + ty * KW + tx];
        } else {
            // Wl[co][ci][tap] = w[ci][co][flip(tap)]  (source is [Cout=CinL][Cin=CoutL])
            val = w[(((int64_t)ci * Cin + co) * KD + (KD - 1 - tz)) * KH * KW + (KH - 1 - ty) * KW + (KW - 1 - tx)];
        }
        int64_t o;
        if (layout == TEM_WL_GENERIC) {
            o = ((int64_t)tap * CinL + ci) * CoutL + co;
        } else {
            int nt = co >> 5, col = co & 31, c8 = ci >> 3, kh = (ci >> 2) & 1, j = ci & 3;
            o = (((((int64_t)nt * ntaps + tap) * (CinL >> 3) + c8) * 2 + kh) * 32 + col) * 4 + j;
        }
        dst[o] = val;
    }
}

extern "C" int tem_conv_pack_weights(const float* w, float* dst, int Cout, int Cin, int kd, int kh, int kw,
                                     int transpose, int layout, tem_stream_t stream) {
    TEM_REQUIRE(w && dst && Cout > 0 && Cin > 0, "tem_conv_pack_weights: bad arguments");
    TEM_REQUIRE((kd == 1 || kd == 3) && (kh == 1 || kh == 3) && (kw == 1 || kw == 3),
                "tem_conv_pack_weights: kernel size (%d,%d,%d) not supported (1 or 3 per axis)", kd, kh, kw);
    if (layout == TEM_WL_BF16X3 || layout == TEM_WL_BF16X6 || layout == TEM_WL_F16X3 || layout == TEM_WL_F16 || layout == TEM_WL_F16X3S ||
        layout == TEM_WL_BF16) {
        int rc = tem_pack_weights_bf16x3(w, dst, Cout, Cin, kd, kh, kw, transpose,
                                         layout == TEM_WL_BF16X6 ? 3 : (layout == TEM_WL_F16X3 ? 4 : (layout == TEM_WL_F16 ? 5 : (layout == TEM_WL_F16X3S ? 6 : (layout == TEM_WL_BF16 ? 7 : 2)))),
                                         (hipStream_t)stream);
        if (rc != TEM_OK) return rc;
        TEM_CHECK_LAUNCH("tem_conv_pack_weights(bf16x3)");
        return TEM_OK;
    }
    if (layout == TEM_WL_MFMA) {
        int CoutL = transpose ? Cin : Cout, CinL = transpose ? Cout : Cin;
        TEM_REQUIRE(CinL % 16 == 0 && CoutL % 32 == 0, "tem_conv_pack_weights: MFMA layout needs Cin%%16==0, Cout%%32==0");
    } else {
        TEM_REQUIRE(layout == TEM_WL_GENERIC, "tem_conv_pack_weights: unknown layout %d", layout);
    }
    int64_t total = (int64_t)Cout * Cin * kd * kh * kw;
    hipLaunchKernelGGL(k_pack_weights, dim3(tem_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout,
                       Cin, kd, kh, kw, transpose, layout);
    TEM_CHECK_LAUNCH("tem_conv_pack_weights");
    return TEM_OK;
}

__global__ __launch_bounds__(256) void k_unpack_wgrad(const float* __restrict__ src, float* __restrict__ dw, int Cout,
                                                      int Cin, int ntaps) {
    const int64_t total = (int64_t)Cout * Cin * ntaps;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates the destination [co][ci][tap]
        int tap = (int)(i % ntaps);
        int64_t r = i / ntaps;
        int ci = (int)(r % Cin);
        int co = (int)(r / Cin);
        dw[i] = src[((int64_t)tap * Cin + ci) * Cout + co];
    }
}

extern "C" int tem_conv_unpack_wgrad(const float* src, float* dw, int Cout, int Cin, int kd, int kh, int kw,
                                     tem_stream_t stream) {
    TEM_REQUIRE(src && dw && Cout > 0 && Cin > 0, "tem_conv_unpack_wgrad: bad arguments");
    int64_t total = (int64_t)Cout * Cin * kd * kh * kw;
    hipLaunchKernelGGL(k_unpack_wgrad, dim3(tem_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dw, Cout,
                       Cin, kd * kh * kw);
    TEM_CHECK_LAUNCH("tem_conv_unpack_wgrad");
    return TEM_OK;
}

// ---------------------------------------------------------------------------
// generic VALU forward: thread <-> (voxel, co), co fastest.  Lanes that share a
// voxel broadcast the x reads; weight reads and the output store are coalesced.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == TEM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TEM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

template <int KD, int KH, int KW, typename TX, typename TY>
__global__ __launch_bounds__(256) void k_conv_fwd_generic(const TX* __restrict__ x, int64_t x_ld,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ w,
                                                          const float* __restrict__ bias, TY* __restrict__ y,
                                                          int64_t y_ld, const TY* __restrict__ ref, int64_t ref_ld,
                                                          int N, int D, int H, int W, int Cin, int Cout, int act) {
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    const int64_t NV = (int64_t)N * D * H * W;
    const int64_t items = NV * Cout;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t v = i / Cout;
    int co = (int)(i % Cout);
    const int64_t dv = stride / Cout;
    const int dco = (int)(stride % Cout);
    for (; i < items; i += stride, v += dv, co += dco) {
        if (co >= Cout) {
            co -= Cout;
            ++v;
        }
        float acc = bias ? bias[co] : 0.f;
        if constexpr (KD * KH * KW == 1) {
            const int n1 = scale ? (int)(v / ((int64_t)D * H * W)) : 0;
            const TX* xp = x + v * x_ld;
            for (int ci = 0; ci < Cin; ++ci) {
                float xv = act_ld1(xp + ci);
                if (scale) xv = fmaf(xv, scale[(int64_t)n1 * Cin + ci], shift[(int64_t)n1 * Cin + ci]);
                acc = fmaf(xv, w[(int64_t)ci * Cout + co], acc);
            }
            acc = apply_act(acc, act);
            if (ref && !(act_ld1(ref + v * ref_ld + co) > 0.f)) acc = 0.f;
            act_st1(y + v * y_ld + co, acc);
            continue;
        }
        int xx = (int)(v % W);
        int64_t r = v / W;
        int yy = (int)(r % H);
        r /= H;
        int zz = (int)(r % D);
        int n = (int)(r / D);
        const float* sc = scale ? scale + (int64_t)n * Cin : nullptr;
        const float* sf = shift ? shift + (int64_t)n * Cin : nullptr;
#pragma unroll
        for (int tz = 0; tz < KD; ++tz) {
            int z2 = zz + tz - PZ;
            if (z2 < 0 || z2 >= D) continue;
#pragma unroll
            for (int ty = 0; ty < KH; ++ty) {
                int y2 = yy + ty - PY;
                if (y2 < 0 || y2 >= H) continue;
#pragma unroll
                for (int tx = 0; tx < KW; ++tx) {
                    int x2 = xx + tx - PX;
                    if (x2 < 0 || x2 >= W) continue;
                    const int tap = (tz * KH + ty) * KW + tx;
                    const TX* xp = x + ((((int64_t)n * D + z2) * H + y2) * W + x2) * x_ld;
                    const float* wp = w + (int64_t)tap * Cin * Cout + co;
                    for (int ci = 0; ci < Cin; ++ci) {
                        float xv = act_ld1(xp + ci);
                        if (sc) xv = fmaf(xv, sc[ci], sf[ci]);
                        acc = fmaf(xv, wp[(int64_t)ci * Cout], acc);
                    }
                }
            }
        }
        acc = apply_act(acc, act);
        if (ref && !(act_ld1(ref + v * ref_ld + co) > 0.f)) acc = 0.f;
        act_st1(y + v * y_ld + co, acc);
    }
}

// 1x1x1 projection to a few channels (out_conv 32->2/12): thread <-> voxel, reads its
// whole channel row with 16-byte loads, keeps Cout accumulators; weights via L1.
template <int COUT, typename TX, typename TY>
__global__ __launch_bounds__(256) void k_conv1x1_smallcout(const TX* __restrict__ x, int64_t x_ld,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ w /*[ci][co]*/,
                                                           const float* __restrict__ bias, TY* __restrict__ y,
                                                           int64_t y_ld, const TY* __restrict__ ref, int64_t ref_ld,
                                                           int64_t V, int64_t NV, int Cin, int act) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < NV; v += (int64_t)gridDim.x * 256) {
        const int n = (int)(v / V);
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
        const TX* xp = x + v * x_ld;
        for (int ci = 0; ci < Cin; ci += 4) {
            float4 t = act_ld4(xp + ci);
            float xv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (scale) xv[j] = fmaf(xv[j], scale[(int64_t)n * Cin + ci + j], shift[(int64_t)n * Cin + ci + j]);
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv[j], w[(ci + j) * COUT + co], acc[co]);
            }
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float a = apply_act(acc[co], act);
            if (ref && !(act_ld1(ref + v * ref_ld + co) > 0.f)) a = 0.f;
            act_st1(y + v * y_ld + co, a);
        }
    }
}

template <int KD, int KH, int KW>
static void launch_fwd_generic(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w,
                               const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D,
                               int H, int W, int Cin, int Cout, int act, hipStream_t s) {
    int64_t items = (int64_t)N * D * H * W * Cout;
    TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, (void)0,
                   hipLaunchKernelGGL((k_conv_fwd_generic<KD, KH, KW, TX, TY>), dim3(tem_grid_1d(items, 256, 256 * 16)), dim3(256), 0, s,
                                      (const TX*)x, x_ld, scale, shift, w, bias, (TY*)y, y_ld, (const TY*)ref, ref_ld, N, D, H, W, Cin,
                                      Cout, act));
}

#define DISPATCH_K(KD, KH, KW, CALL)                                  \
    do {                                                              \
        int key__ = ((KD) == 3) * 4 + ((KH) == 3) * 2 + ((KW) == 3);  \
        switch (key__) {                                              \
            case 0: { CALL(1, 1, 1); } break;                         \
            case 1: { CALL(1, 1, 3); } break;                         \
            case 2: { CALL(1, 3, 1); } break;                         \
            case 3: { CALL(1, 3, 3); } break;                         \
            case 4: { CALL(3, 1, 1); } break;                         \
            case 5: { CALL(3, 1, 3); } break;                         \
            case 6: { CALL(3, 3, 1); } break;                         \
            default: { CALL(3, 3, 3); } break;                        \
        }                                                             \
    } while (0)

extern "C" int64_t tem_conv3d_fwd_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                     int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    if (!use_mfma || Cin % 16 || Cout % 32) return 0;
    int64_t ws = tem_conv_fwd_mfma_ws(N, D, H, W, Cin, Cout, kd, kh, kw);
    if (use_mfma >= 1 && use_mfma <= 7) {   // the z-reuse kernel's split-K launch may want more slices than the patch kernel's
        const int64_t zk = (int64_t)tem_conv_zr_splitk_ks(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma) * N * D * H * W * Cout * 4;
        if (zk > ws) ws = zk;
    }
    return ws;
}

static int conv3d_fwd_impl(const float* x, int64_t x_ld, const float* scale, const float* shift,
                           const float* w_packed, const float* bias, float* y, int64_t y_ld, const float* ref,
                           int64_t ref_ld, void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout,
                           int kd, int kh, int kw, int act, int use_mfma, float* stat, tem_stream_t stream) {
    TEM_REQUIRE(x && w_packed && y, "tem_conv3d_fwd: null pointer");
    TEM_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && x_ld >= (tem_call_cs.x ? 32 : Cin) &&
                    y_ld >= (tem_call_cs.y ? 32 : Cout),
                "tem_conv3d_fwd: bad shape");
    TEM_REQUIRE((kd == 1 || kd == 3) && (kh == 1 || kh == 3) && (kw == 1 || kw == 3),
                "tem_conv3d_fwd: kernel size (%d,%d,%d) not supported (1 or 3 per axis)", kd, kh, kw);
    TEM_REQUIRE((scale == nullptr) == (shift == nullptr), "tem_conv3d_fwd: scale and shift must both be given");
    TEM_REQUIRE(act >= 0 && act <= 2, "tem_conv3d_fwd: Invalid activation: %d", act);
    TEM_REQUIRE(!ref || ref_ld >= Cout, "tem_conv3d_fwd: bad ref_ld");
    const int stx = tem_call_st.x, sty = tem_call_st.y;
    TEM_REQUIRE(stx >= 0 && stx <= 2 && sty >= 0 && sty <= 2 && (stx == 0 || sty == 0 || stx == sty),
                "tem_conv3d_fwd: unsupported storage types (x %d, y %d)", stx, sty);
    TEM_REQUIRE(!(stx || sty) || use_mfma != 1, "tem_conv3d_fwd: the exact-fp32 MFMA kernels take fp32 tensors only");
    hipStream_t s = (hipStream_t)stream;
    TEM_REQUIRE(!(tem_call_cs.x || tem_call_cs.y) || use_mfma == 5 || use_mfma == 7, "tem_conv3d_fwd_ex: chunk strides need use_mfma 5 / 7");
    if (use_mfma >= 2 && use_mfma <= 7) {
        int rc = tem_conv_fwd_bf16x3(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H,
                                     W, Cin, Cout, kd, kh, kw, act, use_mfma, stat, s);
        if (rc != TEM_OK) return rc;
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(bf16x3)");
        return TEM_OK;
    }
    if (use_mfma == 1 && Cin % 16 == 0 && Cout % 32 == 0 && x_ld % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w_packed % 16 == 0) &&
        (!scale || (((uintptr_t)scale % 16 == 0) && ((uintptr_t)shift % 16 == 0)))) {
        // exact fp32 on the z-reuse team kernel (k_conv_zr<..., X32>, round 6): the levels with enough units directly, the
        // 16^3 / 8^3 levels with split input channels; other shapes stay with k_conv_fwd_mfma[_p] below
        const int zr = tem_conv_fwd_zr(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, kd, kh, kw,
                                       act, 1, stat, s);
        if (zr < 0) return TEM_EINVAL;
        if (!zr && tem_conv_fwd_zr_splitk(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H, W, Cin,
                                          Cout, kd, kh, kw, act, 1, stat, s)) {
            TEM_CHECK_LAUNCH("tem_conv3d_fwd(fp32, z-reuse split-K)");
            return TEM_OK;
        }
        if (zr) {
            TEM_CHECK_LAUNCH("tem_conv3d_fwd(fp32, z-reuse)");
            return TEM_OK;
        }
    }
    TEM_REQUIRE(!stat || !use_mfma, "tem_conv3d_fwd_stats: the exact-fp32 patch kernel writes no statistics (tem_conv3d_fwd_stat_blocks() == 0)");
    if (use_mfma) {
        int rc = tem_conv_fwd_mfma(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H,
                                   W, Cin, Cout, kd, kh, kw, act, s);
        if (rc != TEM_OK) return rc;
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(mfma)");
        return TEM_OK;
    }
    const int64_t NV = (int64_t)N * D * H * W;
    if (tem_conv_fwd_cin1(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, N, D, H, W, Cin, Cout, kd, kh, kw, act,
                          stat, s)) {
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(cin1)");
        return TEM_OK;
    }
    TEM_REQUIRE(!stat, "tem_conv3d_fwd_stats: this launch cannot write statistics (tem_conv3d_fwd_stat_blocks() == 0)");
    if (tem_conv_fwd_cout1(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, N, D, H, W, Cin, Cout, kd, kh, kw, act,
                           s)) {
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(cout1)");
        return TEM_OK;
    }
    if (kd == 1 && kh == 1 && kw == 1 &&
        tem_conv1x1_proj(x, x_ld, scale, w_packed, bias, y, y_ld, ref, NV, Cin, Cout, act, s)) {
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(proj)");
        return TEM_OK;
    }
    if (kd == 1 && kh == 1 && kw == 1 &&
        tem_conv1x1_expand(x, x_ld, scale, w_packed, bias, y, y_ld, ref, ref_ld, NV, Cin, Cout, act, s)) {
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(expand)");
        return TEM_OK;
    }
    if (kd == 1 && kh == 1 && kw == 1 && Cin % 4 == 0 && x_ld % 4 == 0 && ((uintptr_t)x % 16 == 0) &&
        (Cout == 1 || Cout == 2 || Cout == 3 || Cout == 4 || Cout == 8 || Cout == 12 || Cout == 16)) {
        dim3 grid(tem_grid_1d(NV, 256, 256 * 16));
#define SC(CO)                                                                                                   \
    case CO:                                                                                                     \
        TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, (void)0,                                            \
                       hipLaunchKernelGGL((k_conv1x1_smallcout<CO, TX, TY>), grid, dim3(256), 0, s, (const TX*)x, x_ld, scale, shift, \
                                          w_packed, bias, (TY*)y, y_ld, (const TY*)ref, ref_ld, (int64_t)D * H * W, NV, Cin, act)); \
        break;
        switch (Cout) {
            SC(1) SC(2) SC(3) SC(4) SC(8) SC(12) SC(16)
        }
#undef SC
        TEM_CHECK_LAUNCH("tem_conv3d_fwd(1x1)");
        return TEM_OK;
    }
#define CALL(A, B, C) \
    launch_fwd_generic<A, B, C>(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, s)
    DISPATCH_K(kd, kh, kw, CALL);
#undef CALL
    TEM_CHECK_LAUNCH("tem_conv3d_fwd(generic)");
    return TEM_OK;
}

extern "C" int tem_conv3d_fwd(const float* x, int64_t x_ld, const float* scale, const float* shift,
                              const float* w_packed, const float* bias, float* y, int64_t y_ld, const float* ref,
                              int64_t ref_ld, void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout,
                              int kd, int kh, int kw, int act, int use_mfma, tem_stream_t stream) {
    TEM_MODE_SCOPE(use_mfma);
    return conv3d_fwd_impl(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H, W, Cin,
                           Cout, kd, kh, kw, act, use_mfma, nullptr, stream);
}

static inline bool ref_free_cin1_ok(int Cout) { return Cout % 4 == 0; }

extern "C" int64_t tem_conv3d_fwd_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                              int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    if (use_mfma == 0)  // VALU kernels: only the small-Cin first-layer kernel (conv_small.hip) provides them
        return (ref_free_cin1_ok(Cout)) ? tem_conv_fwd_cin1_stat_blocks(D, H, W, Cin, Cout, kd, kh, kw) : 0;
    if (use_mfma == 1) {   // exact fp32: only the z-reuse kernel (direct or split-K) writes statistics
        if (Cin % 16 || Cout % 32) return 0;
        const int64_t zrb = tem_conv_zr_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, 1);
        if (zrb >= 0) return zrb;
        const int64_t skb = tem_conv_zr_splitk_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, 1);
        return skb > 0 ? skb : 0;
    }
    if (use_mfma < 2 || use_mfma > 7) return 0;
    return tem_conv_fwd_bf16x3_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma);
}

extern "C" int tem_conv3d_fwd_kernel(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    if (use_mfma >= 2 && use_mfma <= 7 && Cin % 16 == 0 && Cout % 32 == 0) {
        if (tem_conv_zr_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma) >= 0) return 3;
        if (tem_conv_zr_splitk_ks(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma)) return 4;   // z-reuse kernel, split input channels
        return tem_conv_pp_tiles(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma);
    }
    if (use_mfma == 1 && Cin % 16 == 0 && Cout % 32 == 0) {
        if (tem_conv_zr_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, 1) >= 0) return 3;
        if (tem_conv_zr_splitk_ks(N, D, H, W, Cin, Cout, kd, kh, kw, 1)) return 4;
    }
    return 0;
}

extern "C" int tem_conv3d_fwd_stats(const float* x, int64_t x_ld, const float* scale, const float* shift,
                                    const float* w_packed, const float* bias, float* y, int64_t y_ld, const float* ref,
                                    int64_t ref_ld, void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Cin,
                                    int Cout, int kd, int kh, int kw, int act, int use_mfma, float* stat_part,
                                    int64_t stat_blocks, tem_stream_t stream) {
    TEM_MODE_SCOPE(use_mfma);
    TEM_REQUIRE(stat_part, "tem_conv3d_fwd_stats: null statistics buffer");
    TEM_REQUIRE(stat_blocks > 0 && stat_blocks == tem_conv3d_fwd_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma),
                "tem_conv3d_fwd_stats: stat_blocks must be tem_conv3d_fwd_stat_blocks() of this launch (and > 0)");
    return conv3d_fwd_impl(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H, W, Cin,
                           Cout, kd, kh, kw, act, use_mfma, stat_part, stream);
}

// ---------------------------------------------------------------------------
// generic VALU weight gradient.  thread <-> (pair=(ci,co) of a <=256-pair block, row r);
// all taps accumulate in registers so g is read once.  Two-stage deterministic:
// partial[chunk][tap][ci][co] -> fp64 merge.
// ---------------------------------------------------------------------------
template <int KD, int KH, int KW, typename TX, typename TG>
__global__ __launch_bounds__(256) void k_conv_wgrad_generic(const TX* __restrict__ x, int64_t x_ld,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const TG* __restrict__ g, int64_t g_ld, int N, int D,
                                                            int H, int W, int Cin, int Cout, int npairs_blk, int rows,
                                                            int64_t vper, float* __restrict__ part) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    extern __shared__ float sh[];  // [NT][rows][npairs_blk]
    const int64_t NV = (int64_t)N * D * H * W;
    const int chunk = blockIdx.x, pb = blockIdx.y;
    const int pl = threadIdx.x % npairs_blk, r = threadIdx.x / npairs_blk;
    const int pair = pb * npairs_blk + pl;
    const bool active = pair < Cin * Cout && r < rows;
    const int ci = active ? pair / Cout : 0, co = active ? pair % Cout : 0;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    int64_t v0 = (int64_t)chunk * vper, v1 = v0 + vper;
    if (v1 > NV) v1 = NV;
    if (active) {
        for (int64_t v = v0 + r; v < v1; v += rows) {
            int xx = (int)(v % W);
            int64_t q = v / W;
            int yy = (int)(q % H);
            q /= H;
            int zz = (int)(q % D);
            int n = (int)(q / D);
            const float gv = act_ld1(g + v * g_ld + co);
            float sc = 1.f, sf = 0.f;
            if (scale) {
                sc = scale[(int64_t)n * Cin + ci];
                sf = shift[(int64_t)n * Cin + ci];
            }
#pragma unroll
            for (int tz = 0; tz < KD; ++tz) {
                int z2 = zz + tz - PZ;
#pragma unroll
                for (int ty = 0; ty < KH; ++ty) {
                    int y2 = yy + ty - PY;
#pragma unroll
                    for (int tx = 0; tx < KW; ++tx) {
                        int x2 = xx + tx - PX;
                        if (z2 < 0 || z2 >= D || y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
                        float xv = act_ld1(x + ((((int64_t)n * D + z2) * H + y2) * W + x2) * x_ld + ci);
                        xv = fmaf(xv, sc, sf);
                        acc[(tz * KH + ty) * KW + tx] = fmaf(xv, gv, acc[(tz * KH + ty) * KW + tx]);
                    }
                }
            }
        }
    }
    if (r < rows) {
#pragma unroll
        for (int t = 0; t < NT; ++t) sh[((int64_t)t * rows + r) * npairs_blk + pl] = acc[t];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NT * npairs_blk; idx += 256) {
        int t = idx / npairs_blk, p = idx % npairs_blk;
        int gp = pb * npairs_blk + p;
        if (gp >= Cin * Cout) continue;
        float s = 0.f;
        for (int rr = 0; rr < rows; ++rr) s += sh[((int64_t)t * rows + rr) * npairs_blk + p];
        // partial[chunk][tap][ci][co]; pair index == ci*Cout+co
        part[((int64_t)chunk * NT + t) * Cin * Cout + gp] = s;
    }
}

// column sums: db[co] = sum_v g[v][co]
template <typename TG>
__global__ __launch_bounds__(256) void k_colsum_partial(const TG* __restrict__ g, int64_t g_ld, int64_t NV, int C,
                                                        int rows, int64_t vper, float* __restrict__ part) {
    extern __shared__ float sh[];  // [rows][Cb]
    const int Cb = C < 256 ? C : 256;
    const int cl = threadIdx.x % Cb, r = threadIdx.x / Cb;
    int64_t v0 = (int64_t)blockIdx.x * vper, v1 = v0 + vper;
    if (v1 > NV) v1 = NV;
    for (int c0 = 0; c0 < C; c0 += Cb) {
        int c = c0 + cl;
        float s = 0.f;
        if (c < C && r < rows)
            for (int64_t v = v0 + r; v < v1; v += rows) s += act_ld1(g + v * g_ld + c);
        if (r < rows) sh[r * Cb + cl] = s;
        __syncthreads();
        if (r == 0 && c < C) {
            float a = 0.f;
            for (int rr = 0; rr < rows; ++rr) a += sh[rr * Cb + cl];
            part[(int64_t)blockIdx.x * C + c] = a;
        }
        __syncthreads();
    }
}

struct WgradGenericPlan {
    int npairs_blk, rows, npb, nchunks;
    int64_t vper;
    int64_t part_floats;   // wgrad partials
    int db_chunks;
    int64_t db_floats;
};

static WgradGenericPlan wgrad_generic_plan(int64_t NV, int Cin, int Cout, int ntaps) {
    WgradGenericPlan p;
    int pairs = Cin * Cout;
    p.npairs_blk = pairs < 256 ? pairs : 256;
    p.rows = 256 / p.npairs_blk;
    if (p.rows < 1) p.rows = 1;
    p.npb = (int)tem_cdiv(pairs, p.npairs_blk);
    int64_t nch = tem_cdiv(NV, (int64_t)p.rows * 64);
    int64_t cap = (64ll << 20) / ((int64_t)ntaps * pairs * 4);
    if (cap < 1) cap = 1;
    if (nch > cap) nch = cap;
    if (nch > 2048) nch = 2048;
    if (nch < 1) nch = 1;
    p.nchunks = (int)nch;
    p.vper = tem_cdiv(NV, nch);
    p.part_floats = (int64_t)p.nchunks * ntaps * pairs;
    int64_t dbc = tem_cdiv(NV, 2048);
    if (dbc > 512) dbc = 512;
    if (dbc < 1) dbc = 1;
    p.db_chunks = (int)dbc;
    p.db_floats = (int64_t)p.db_chunks * Cout;
    return p;
}

extern "C" int64_t tem_conv3d_wgrad_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                       int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    int64_t NV = (int64_t)N * D * H * W;
    int ntaps = kd * kh * kw;
    WgradGenericPlan p = wgrad_generic_plan(NV, Cin, Cout, ntaps);
    int64_t bytes = tem_align_up(p.db_floats, 64) * 4;
    if (use_mfma == 2 || use_mfma == 5 || use_mfma == 7 || use_mfma == 8) {
        bytes += tem_conv_wgrad_bf16x3_ws(N, D, H, W, Cin, Cout, kd, kh, kw);
    } else if (use_mfma) {
        int64_t b = tem_conv_wgrad_mfma_ws(N, D, H, W, Cin, Cout, kd, kh, kw);
        if (tem_conv_wgrad_tr_fp32_ok(N, D, H, W, Cin, Cout, kd, kh, kw)) {
            const int64_t b2 = tem_conv_wgrad_bf16x3_ws(N, D, H, W, Cin, Cout, kd, kh, kw);
            if (b2 > b) b = b2;
        }
        bytes += b;
    } else {
        int64_t b = p.part_floats * 4;
        int64_t c1 = tem_conv_wgrad_cin1_ws(Cout, ntaps), pj = tem_conv1x1_proj_wgrad_ws(Cin, Cout);
        if (Cin <= 4 && c1 > b) b = c1;
        if (ntaps == 1 && pj > b) b = pj;
        bytes += b;
    }
    return bytes + 256;
}

template <int KD, int KH, int KW>
static void launch_wgrad_generic(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                                 int64_t g_ld, int N, int D, int H, int W, int Cin, int Cout,
                                 const WgradGenericPlan& p, float* part, hipStream_t s) {
    constexpr int NT = KD * KH * KW;
    size_t lds = (size_t)NT * p.rows * p.npairs_blk * sizeof(float);
    TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TG, (void)0,
                   hipLaunchKernelGGL((k_conv_wgrad_generic<KD, KH, KW, TX, TG>), dim3(p.nchunks, p.npb), dim3(256), lds, s, (const TX*)x,
                                      x_ld, scale, shift, (const TG*)g, g_ld, N, D, H, W, Cin, Cout, p.npairs_blk, p.rows, p.vper, part));
}

static int conv3d_wgrad_impl(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                             int64_t g_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int N, int D, int H,
                             int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma, int sd_layout,
                             const float* w_sd, const float* gamma, const float* beta, float* norm_sums,
                             const float* gnx, int64_t gnx_ld, const float* gcoef, tem_stream_t stream) {
    TEM_REQUIRE(x && g && dw && ws, "tem_conv3d_wgrad: null pointer");
    TEM_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && x_ld >= (tem_call_cs.x ? 32 : Cin) && g_ld >= Cout,
                "tem_conv3d_wgrad: bad shape");
    TEM_REQUIRE((kd == 1 || kd == 3) && (kh == 1 || kh == 3) && (kw == 1 || kw == 3),
                "tem_conv3d_wgrad: kernel size (%d,%d,%d) not supported (1 or 3 per axis)", kd, kh, kw);
    TEM_REQUIRE((scale == nullptr) == (shift == nullptr), "tem_conv3d_wgrad: scale and shift must both be given");
    const int stx = tem_call_st.x, sty = tem_call_st.y;
    TEM_REQUIRE(stx >= 0 && stx <= 2 && sty >= 0 && sty <= 2 && (stx == 0 || sty == 0 || stx == sty),
                "tem_conv3d_wgrad: unsupported storage types (x %d, g %d)", stx, sty);
    TEM_REQUIRE(!(stx || sty) || !(use_mfma == 1 || use_mfma == 3 || use_mfma == 4 || use_mfma == 6),
                "tem_conv3d_wgrad: the exact-fp32 MFMA kernels take fp32 tensors only");
    if (ws_bytes < tem_conv3d_wgrad_ws(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma)) {
        tem_set_error("tem_conv3d_wgrad: workspace too small");
        return TEM_EWS;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t NV = (int64_t)N * D * H * W;
    const int ntaps = kd * kh * kw;
    WgradGenericPlan p = wgrad_generic_plan(NV, Cin, Cout, ntaps);
    float* dbpart = (float*)ws;
    float* rest = dbpart + tem_align_up(p.db_floats, 64);
    TEM_REQUIRE(!gcoef || !use_mfma, "tem_conv3d_wgrad_gnorm: use_mfma must be 0");
    TEM_REQUIRE(!tem_call_cs.x || use_mfma == 5 || use_mfma == 7, "tem_conv3d_wgrad_ex: a chunk stride needs use_mfma 5 / 7");
    if (use_mfma == 2 || use_mfma == 5 || use_mfma == 7 || use_mfma == 8) {
        // 5: single fp16 product in the z-sliding kernel (autocast-equivalent); the other shapes keep bf16x3
        // 8: fp16 2x1 (tem_conv3d_wgrad_gscaled; z-sliding kernel only)
        int rc = tem_conv_wgrad_bf16x3(x, x_ld, scale, shift, g, g_ld, dw, db, rest,
                                       ws_bytes - (int64_t)((char*)rest - (char*)ws), N, D, H, W, Cin, Cout, kd, kh, kw,
                                       sd_layout, use_mfma == 5 ? 1 : (use_mfma == 7 ? 2 : (use_mfma == 8 ? 3 : 0)), w_sd, gamma,
                                       beta, norm_sums, s);
        if (rc != TEM_OK) return rc;
        TEM_CHECK_LAUNCH("tem_conv3d_wgrad(bf16x3)");
        return TEM_OK;
    }
    if (use_mfma && !(stx || sty) && tem_conv_wgrad_tr_fp32_ok(N, D, H, W, Cin, Cout, kd, kh, kw) && x_ld % 4 == 0 && g_ld % 4 == 0 &&
        ((uintptr_t)x % 16 == 0) && ((uintptr_t)g % 16 == 0)) {
        // exact fp32 on the z-sliding staging-team kernel (round 4): fp32 records in LDS, v_mfma_f32_32x32x2_f32; the slab
        // merge delivers the norm sums as in the split modes (round 6: the merge does not care which arithmetic filled the slabs)
        int rc = tem_conv_wgrad_bf16x3(x, x_ld, scale, shift, g, g_ld, dw, db, rest,
                                       ws_bytes - (int64_t)((char*)rest - (char*)ws), N, D, H, W, Cin, Cout, kd, kh, kw,
                                       sd_layout, 4, w_sd, gamma, beta, norm_sums, s);
        if (rc != TEM_OK) return rc;
        TEM_CHECK_LAUNCH("tem_conv3d_wgrad(fp32, z-sliding)");
        return TEM_OK;
    }
    TEM_REQUIRE(!norm_sums, "tem_conv3d_wgrad_sums: only the z-sliding kernels deliver the norm sums (tem_conv3d_wgrad_sums_ok() == 0)");
    if (use_mfma) {
        int rc = tem_conv_wgrad_mfma(x, x_ld, scale, shift, g, g_ld, dw, db, rest,
                                     ws_bytes - (int64_t)((char*)rest - (char*)ws), N, D, H, W, Cin, Cout, kd, kh, kw,
                                     sd_layout, s);
        if (rc != TEM_OK) return rc;
        TEM_CHECK_LAUNCH("tem_conv3d_wgrad(mfma)");
        return TEM_OK;
    }
    if (tem_conv_wgrad_cin1(x, x_ld, scale, shift, g, g_ld, dw, db, rest, N, D, H, W, Cin, Cout, kd, kh, kw, sd_layout,
                            gnx, gnx_ld, gcoef, s)) {
        TEM_CHECK_LAUNCH("tem_conv3d_wgrad(cin1)");
        return TEM_OK;
    }
    TEM_REQUIRE(!gcoef, "tem_conv3d_wgrad_gnorm: only the small-Cin first-layer kernel applies a norm backward to g");
    if (ntaps == 1 && tem_conv1x1_proj_wgrad(x, x_ld, scale, g, g_ld, dw, db, rest, NV, Cin, Cout, sd_layout, s)) {
        TEM_CHECK_LAUNCH("tem_conv3d_wgrad(proj)");
        return TEM_OK;
    }
    if (db) {
        int Cb = Cout < 256 ? Cout : 256;
        int rows = 256 / Cb;
        int64_t vper = tem_cdiv(NV, p.db_chunks);
        TEM_ST_SWITCH(tem_call_st.y, TG,
                      hipLaunchKernelGGL(k_colsum_partial<TG>, dim3(p.db_chunks), dim3(256), (size_t)rows * Cb * sizeof(float), s,
                                         (const TG*)g, g_ld, NV, Cout, rows, vper, dbpart));
        tem_reduce_slabs(dbpart, p.db_chunks, (int64_t)Cout, (int64_t)Cout, db, s);
    }
#define CALL(A, B, C) launch_wgrad_generic<A, B, C>(x, x_ld, scale, shift, g, g_ld, N, D, H, W, Cin, Cout, p, rest, s)
    DISPATCH_K(kd, kh, kw, CALL);
#undef CALL
    int64_t n = (int64_t)ntaps * Cin * Cout;
    tem_reduce_slabs_w(rest, p.nchunks, ntaps, Cin, Cout, n, dw, sd_layout, s);
    TEM_CHECK_LAUNCH("tem_conv3d_wgrad(generic)");
    return TEM_OK;
}

extern "C" int tem_conv3d_wgrad(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                                int64_t g_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int N, int D, int H,
                                int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma, int sd_layout,
                                tem_stream_t stream) {
    TEM_MODE_SCOPE(use_mfma);
    return conv3d_wgrad_impl(x, x_ld, scale, shift, g, g_ld, dw, db, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh, kw,
                             use_mfma, sd_layout, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, stream);
}

// Backward of the network's output projection (out_conv: nn.Conv3d(features, out_channels, 1), reference model/unet.py:638)
// in ONE pass over its input x (a ReLU output): dw, db as tem_conv3d_wgrad AND the masked data gradient
// gx = (x > 0) * (g . w) as tem_conv3d_fwd(transposed pack, ref = x) -- two kernels that each read the 512 MB tensor before.
extern "C" int tem_conv1x1_out_bwd_ok(int Cin, int Cout) {
    return Cin % 32 == 0 && Cin <= 64 && Cout >= 1 && Cout <= 4 && (Cin / 32) * Cout <= 8;
}
extern "C" int tem_conv1x1_out_bwd(const float* x, int64_t x_ld, const float* g, int64_t g_ld, const float* w, float* gx,
                                   int64_t gx_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int64_t NV, int Cin,
                                   int Cout, tem_stream_t stream) {
    TEM_REQUIRE(x && g && w && gx && dw && ws && NV > 0 && x_ld >= Cin && gx_ld >= Cin && g_ld >= Cout,
                "tem_conv1x1_out_bwd: bad arguments");
    TEM_REQUIRE(tem_conv1x1_out_bwd_ok(Cin, Cout), "tem_conv1x1_out_bwd: tem_conv1x1_out_bwd_ok() == 0 for %d -> %d", Cin, Cout);
    TEM_REQUIRE(ws_bytes >= tem_conv1x1_proj_wgrad_ws(Cin, Cout), "tem_conv1x1_out_bwd: workspace too small");
    if (!tem_conv1x1_out_bwd(x, x_ld, g, g_ld, w, gx, gx_ld, dw, db, ws, NV, Cin, Cout, 1, (hipStream_t)stream)) {
        tem_set_error("tem_conv1x1_out_bwd: x / gx / w need 16-byte alignment and ld %% 4 == 0");
        return TEM_EINVAL;
    }
    TEM_CHECK_LAUNCH("tem_conv1x1_out_bwd");
    return TEM_OK;
}
extern "C" int64_t tem_conv1x1_out_bwd_ws(int Cin, int Cout) { return tem_conv1x1_proj_wgrad_ws(Cin, Cout); }

// tem_conv1x1_out_bwd for x / gx of storage type st_x and g of storage type st_g (g is the gradient of the network output:
// normally fp32); out_amax (optional): device word that receives max |gx| (see tem_maxpool3d_bwd_st)
extern "C" int tem_conv1x1_out_bwd_st(const void* x, int64_t x_ld, const void* g, int64_t g_ld, const float* w, void* gx,
                                      int64_t gx_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int64_t NV, int Cin,
                                      int Cout, unsigned* out_amax, int st_x, int st_g, tem_stream_t stream) {
    TEM_REQUIRE(st_x >= 0 && st_x <= 2 && st_g >= 0 && st_g <= 2, "tem_conv1x1_out_bwd_st: unknown storage type");
    TemStScope sc(st_x, st_g);
    TemByproducts bp = {};
    bp.out_amax = out_amax;
    TemBpScope bsc(&bp);
    return tem_conv1x1_out_bwd((const float*)x, x_ld, (const float*)g, g_ld, w, (float*)gx, gx_ld, dw, db, ws, ws_bytes, NV, Cin,
                               Cout, stream);
}

extern "C" int tem_conv3d_wgrad_gmax_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                        int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    return use_mfma == 2 && tem_conv_wgrad_gmax_ok(N, D, H, W, Cin, Cout, kd, kh, kw);
}

extern "C" int tem_conv3d_wgrad_gmax(const float* x, int64_t x_ld, const float* scale, const float* shift,
                                     const float* g, int64_t g_ld, const float* w, const float* gamma,
                                     const float* beta, float* dw, float* db, float* norm_sums, unsigned* g_amax,
                                     void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd,
                                     int kh, int kw, int use_mfma, tem_stream_t stream) {
    TEM_MODE_SCOPE(use_mfma);
    TEM_REQUIRE(g_amax, "tem_conv3d_wgrad_gmax: null g_amax");
    TEM_REQUIRE(tem_conv3d_wgrad_gmax_ok(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma),
                "tem_conv3d_wgrad_gmax: tem_conv3d_wgrad_gmax_ok() == 0 for this layer");
    TEM_REQUIRE(!norm_sums || (w && db && tem_conv3d_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma)),
                "tem_conv3d_wgrad_gmax: norm_sums needs weights, a bias gradient and tem_conv3d_wgrad_sums_ok() != 0");
    tem_wgrad_gmax_target = g_amax;
    const int rc = conv3d_wgrad_impl(x, x_ld, scale, shift, g, g_ld, dw, db, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh,
                                     kw, use_mfma, 1, norm_sums ? w : nullptr, gamma, beta, norm_sums, nullptr, 0, nullptr,
                                     stream);
    tem_wgrad_gmax_target = nullptr;
    return rc;
}

// ---- weight gradient with 16-bit-class x^ and an 11-bit g (two MFMAs per product instead of three) -----------------------
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, int64_t ld, int cq, int64_t nq, unsigned* __restrict__ amax) {
    // nq = voxels * cq float4 items; max |x| as an integer max of the bit patterns (exact, order-independent; inf wins, NaNs are
    // skipped by v_max_f32 -- the consumers multiply the NaN itself through, so nothing is hidden)
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += (int64_t)gridDim.x * 256) {
        const int64_t v = i / cq;
        const int q = (int)(i - v * cq);
        typedef float f4n __attribute__((ext_vector_type(4)));
        const f4n t = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(x + v * ld + q * 4));
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(t.x), __builtin_fabsf(t.y)),
                                               __builtin_fmaxf(__builtin_fabsf(t.z), __builtin_fabsf(t.w))));
    }
    unsigned u = __builtin_bit_cast(unsigned, m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o, 64));
    __shared__ unsigned red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, max(max(red[0], red[1]), max(red[2], red[3])));
}

extern "C" int tem_absmax(const float* x, int64_t ld, int C, int64_t nvox, unsigned* amax, tem_stream_t stream) {
    TEM_REQUIRE(x && amax && C > 0 && C % 4 == 0 && ld >= C && ld % 4 == 0 && ((uintptr_t)x % 16 == 0) && nvox >= 0,
                "tem_absmax: needs 16-byte aligned rows of C %% 4 == 0 floats");
    if (nvox == 0) return TEM_OK;
    const int64_t nq = nvox * (C / 4);
    int64_t nb = tem_cdiv(nq, 256 * 8);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, ld, C / 4, nq, amax);
    TEM_CHECK_LAUNCH("tem_absmax");
    return TEM_OK;
}

extern "C" int tem_conv3d_wgrad_gscaled_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    return tem_conv_wgrad_gscaled_ok(N, D, H, W, Cin, Cout, kd, kh, kw);
}

extern "C" int tem_conv3d_wgrad_cs_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int st, int64_t x_cs) {
    return tem_conv_wgrad_cs_ok(N, D, H, W, Cin, Cout, kd, kh, kw, st, x_cs);
}

extern "C" int tem_conv3d_wgrad_gscaled(const float* x, int64_t x_ld, const float* scale, const float* shift,
                                        const float* g, int64_t g_ld, const float* w, const float* gamma,
                                        const float* beta, float* dw, float* db, float* norm_sums,
                                        const unsigned* g_amax, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                                        int Cin, int Cout, int kd, int kh, int kw, tem_stream_t stream) {
    TEM_REQUIRE(g_amax, "tem_conv3d_wgrad_gscaled: null g_amax");
    TEM_REQUIRE(tem_conv3d_wgrad_gscaled_ok(N, D, H, W, Cin, Cout, kd, kh, kw),
                "tem_conv3d_wgrad_gscaled: tem_conv3d_wgrad_gscaled_ok() == 0 for this layer");
    TEM_REQUIRE(!norm_sums || (w && db && tem_conv3d_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw, 8)),
                "tem_conv3d_wgrad_gscaled: norm_sums needs weights, a bias gradient and tem_conv3d_wgrad_sums_ok() != 0");
    tem_wgrad_gscale_source = g_amax;
    const int rc = conv3d_wgrad_impl(x, x_ld, scale, shift, g, g_ld, dw, db, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh,
                                     kw, 8, 1, norm_sums ? w : nullptr, gamma, beta, norm_sums, nullptr, 0, nullptr,
                                     stream);
    tem_wgrad_gscale_source = nullptr;
    return rc;
}

extern "C" int tem_conv3d_fwd_gscaled(const float* x, int64_t x_ld, const float* w_packed, float* y, int64_t y_ld,
                                      const float* ref, int64_t ref_ld, const unsigned* in_amax, void* ws,
                                      int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh,
                                      int kw, tem_stream_t stream) {
    TEM_REQUIRE(in_amax, "tem_conv3d_fwd_gscaled: null in_amax");
    TEM_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0 && tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, kd, kh, kw, 4) == 3,
                "tem_conv3d_fwd_gscaled: only launches that tem_conv3d_fwd_kernel() reports as 3 (z-reuse kernel) take a "
                "device-side prescale");
    tem_zr_in_amax = in_amax;
    const int rc = conv3d_fwd_impl(x, x_ld, nullptr, nullptr, w_packed, nullptr, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H,
                                   W, Cin, Cout, kd, kh, kw, TEM_ACT_NONE, 4, nullptr, stream);
    const bool consumed = tem_zr_in_amax == nullptr;
    tem_zr_in_amax = nullptr;
    if (rc == TEM_OK && !consumed) {
        tem_set_error("tem_conv3d_fwd_gscaled: the launch did not take the z-reuse kernel (alignment of y / ref?)");
        return TEM_EINVAL;
    }
    return rc;
}

// Data gradient that lands behind a ReLU + norm: y = ref > 0 ? a*(conv) - m1 - (ref - mean)*m2r : 0 with coef[N][Cout][4] =
// (a, m1, m2r, mean) from tem_norm_bwd_coef -- the epilogue of the z-reuse kernel replaces tem_norm_bwd_from_sums' pass over
// g and ref.  Only for launches tem_conv3d_fwd_kernel() reports as 3.
extern "C" int tem_conv3d_fwd_refnorm(const float* x, int64_t x_ld, const float* w_packed, float* y, int64_t y_ld,
                                      const float* ref, int64_t ref_ld, const float* coef, void* ws, int64_t ws_bytes,
                                      int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma,
                                      tem_stream_t stream) {
    TEM_MODE_SCOPE(use_mfma);
    TEM_REQUIRE(ref && coef, "tem_conv3d_fwd_refnorm: null ref / coef");
    TEM_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0 && tem_conv3d_fwd_kernel(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma) == 3,
                "tem_conv3d_fwd_refnorm: only launches that tem_conv3d_fwd_kernel() reports as 3 (z-reuse kernel) apply a "
                "norm backward in their epilogue");
    tem_zr_ref_coef = coef;
    const int rc = conv3d_fwd_impl(x, x_ld, nullptr, nullptr, w_packed, nullptr, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H,
                                   W, Cin, Cout, kd, kh, kw, TEM_ACT_NONE, use_mfma, nullptr, stream);
    const bool consumed = tem_zr_ref_coef == nullptr;
    tem_zr_ref_coef = nullptr;
    if (rc == TEM_OK && !consumed) {
        tem_set_error("tem_conv3d_fwd_refnorm: the launch did not take the z-reuse kernel (alignment of y / ref?)");
        return TEM_EINVAL;
    }
    return rc;
}

extern "C" int tem_conv3d_wgrad_sums_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                        int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    if (use_mfma == 1)   // exact fp32: the layers k_conv_wgrad_tr<4> takes (option fp32_zr: the data gradient that consumes the sums)
        return tem_option(TEM_OPT_FP32_ZR) && !tem_call_st.x && !tem_call_st.y && tem_conv_wgrad_tr_fp32_ok(N, D, H, W, Cin, Cout, kd, kh, kw) &&
               tem_conv_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw);
    if (use_mfma != 2 && use_mfma != 5 && use_mfma != 7 && use_mfma != 8) return 0;
    return tem_conv_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw);
}

extern "C" int tem_conv3d_wgrad_sums(const float* x, int64_t x_ld, const float* scale, const float* shift,
                                     const float* g, int64_t g_ld, const float* w, const float* gamma,
                                     const float* beta, float* dw, float* db, float* norm_sums, void* ws,
                                     int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh,
                                     int kw, int use_mfma, tem_stream_t stream) {
    TEM_MODE_SCOPE(use_mfma);
    TEM_REQUIRE(w && norm_sums && db, "tem_conv3d_wgrad_sums: null pointer (weights, sums and bias gradient are required)");
    TEM_REQUIRE(tem_conv3d_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma),
                "tem_conv3d_wgrad_sums: tem_conv3d_wgrad_sums_ok() == 0 for this layer");
    return conv3d_wgrad_impl(x, x_ld, scale, shift, g, g_ld, dw, db, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh, kw,
                             use_mfma, 1, w, gamma, beta, norm_sums, nullptr, 0, nullptr, stream);
}

// ---- every variant and by-product of the forward / data-gradient convolution as explicit arguments (tem_hip.h) -----------------
extern "C" int tem_conv3d_fwd_ex(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w_packed,
                                 const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                                 int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                                 int use_mfma, const unsigned* in_amax, const float* ref_coef, float* stat_part,
                                 int64_t stat_blocks, int64_t x_cs, int64_t y_cs, TemByproducts* bp, tem_stream_t stream) {
    TEM_REQUIRE(!stat_part || (!in_amax && !ref_coef && !bp), "tem_conv3d_fwd_ex: stat_part excludes in_amax / ref_coef / by-products");
    TEM_REQUIRE(!(in_amax && ref_coef), "tem_conv3d_fwd_ex: in_amax and ref_coef exclude each other");
    TEM_REQUIRE(x_cs >= 0 && y_cs >= 0 && (!(x_cs || y_cs) || (!in_amax && !ref_coef && ((use_mfma & 0xff) == 5 || (use_mfma & 0xff) == 7))),
                "tem_conv3d_fwd_ex: chunk strides go with the one-term modes on 16-bit tensors (use_mfma 5 / 7), no in_amax / ref_coef");
    TemCsScope csc(x_cs, y_cs);
    TEM_REQUIRE(!bp || (!bp->coef && (!bp->sums_part || (bp->sums_x && bp->sums_mean && bp->sums_rstd && bp->sums_G > 0 && bp->sums_nblk > 0))),
                "tem_conv3d_fwd_ex: bad by-product request (TEM_BP_NORM_COEF belongs to tem_conv3d_wgrad_ex; TEM_BP_NORM_SUMS needs "
                "sums_x / sums_mean / sums_rstd / sums_G / sums_nblk)");
    TemBpScope bsc(bp);
    if (in_amax) {
        TEM_REQUIRE(!scale && !shift && !bias && act == TEM_ACT_NONE && (use_mfma & 0xff) == 4,
                    "tem_conv3d_fwd_ex: in_amax (the fp16 two-term data gradient) takes use_mfma 4 and no scale / shift / bias / act");
        return tem_conv3d_fwd_gscaled(x, x_ld, w_packed, y, y_ld, ref, ref_ld, in_amax, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh,
                                      kw, stream);
    }
    if (ref_coef) {
        TEM_REQUIRE(!scale && !shift && !bias && act == TEM_ACT_NONE, "tem_conv3d_fwd_ex: ref_coef takes no scale / shift / bias / act");
        return tem_conv3d_fwd_refnorm(x, x_ld, w_packed, y, y_ld, ref, ref_ld, ref_coef, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh,
                                      kw, use_mfma, stream);
    }
    if (stat_part)
        return tem_conv3d_fwd_stats(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H, W, Cin, Cout,
                                    kd, kh, kw, act, use_mfma, stat_part, stat_blocks, stream);
    return tem_conv3d_fwd(x, x_ld, scale, shift, w_packed, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh,
                          kw, act, use_mfma, stream);
}

extern "C" int tem_conv3d_wgrad_ex(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                                   int64_t g_ld, const float* w, const float* gamma, const float* beta, float* dw, float* db,
                                   float* norm_sums, const unsigned* g_amax_in, unsigned* g_amax_out, void* ws, int64_t ws_bytes,
                                   int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int use_mfma,
                                   int64_t x_cs, TemByproducts* bp, tem_stream_t stream) {
    TEM_REQUIRE(!(g_amax_in && g_amax_out), "tem_conv3d_wgrad_ex: g_amax_in and g_amax_out exclude each other");
    TEM_REQUIRE(x_cs >= 0 && (!x_cs || (!g_amax_in && !g_amax_out && ((use_mfma & 0xff) == 5 || (use_mfma & 0xff) == 7))),
                "tem_conv3d_wgrad_ex: a chunk stride goes with the one-term modes on 16-bit tensors (use_mfma 5 / 7)");
    TemCsScope csc(x_cs, 0);
    TEM_REQUIRE(!bp || (!bp->out_amax && !bp->sums_part && (!bp->coef || (norm_sums && bp->coef_mean && bp->coef_rstd && bp->coef_G > 0))),
                "tem_conv3d_wgrad_ex: bad by-product request (only TEM_BP_NORM_COEF, which needs norm_sums, coef_mean, coef_rstd, coef_G)");
    TemBpScope bsc(bp);
    if (g_amax_in) {
        TEM_REQUIRE((use_mfma & 0xff) == 8, "tem_conv3d_wgrad_ex: g_amax_in (the fp16 2x1 arithmetic) takes use_mfma 8");
        return tem_conv3d_wgrad_gscaled(x, x_ld, scale, shift, g, g_ld, w, gamma, beta, dw, db, norm_sums, g_amax_in, ws, ws_bytes, N, D,
                                        H, W, Cin, Cout, kd, kh, kw, stream);
    }
    if (g_amax_out)
        return tem_conv3d_wgrad_gmax(x, x_ld, scale, shift, g, g_ld, w, gamma, beta, dw, db, norm_sums, g_amax_out, ws, ws_bytes, N, D,
                                     H, W, Cin, Cout, kd, kh, kw, use_mfma, stream);
    if (norm_sums)
        return tem_conv3d_wgrad_sums(x, x_ld, scale, shift, g, g_ld, w, gamma, beta, dw, db, norm_sums, ws, ws_bytes, N, D, H, W, Cin,
                                     Cout, kd, kh, kw, use_mfma, stream);
    return tem_conv3d_wgrad(x, x_ld, scale, shift, g, g_ld, dw, db, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh, kw, use_mfma, 1,
                            stream);
}

// tem_conv3d_wgrad of a FIRST layer (small Cin, VALU kernel) whose output gradient g is still the raw data gradient
// behind the norm that follows this conv's ReLU: the norm backward (coefficients from tem_norm_bwd_coef) and the ReLU
// mask are applied while g is loaded -- y (this conv's output, the norm's input) is read instead of a rewritten g.
extern "C" int tem_conv3d_wgrad_gnorm_ok(int Cin, int Cout, int kd, int kh, int kw, int use_mfma) {
    TEM_MODE_SCOPE(use_mfma);
    const int cq = Cout / 4, key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    return use_mfma == 0 && Cin >= 1 && Cin <= 4 && Cout % 4 == 0 && cq <= 16 && (cq & (cq - 1)) == 0 && (key == 7 || key == 3);
}

extern "C" int tem_conv3d_wgrad_gnorm(const float* x, int64_t x_ld, const float* scale, const float* shift,
                                      const float* g, int64_t g_ld, const float* y, int64_t y_ld, const float* gcoef,
                                      float* dw, float* db, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                                      int Cin, int Cout, int kd, int kh, int kw, int sd_layout, tem_stream_t stream) {
    TEM_REQUIRE(y && gcoef && y_ld >= Cout, "tem_conv3d_wgrad_gnorm: null pointer");
    TEM_REQUIRE(tem_conv3d_wgrad_gnorm_ok(Cin, Cout, kd, kh, kw, 0), "tem_conv3d_wgrad_gnorm: tem_conv3d_wgrad_gnorm_ok() == 0");
    return conv3d_wgrad_impl(x, x_ld, scale, shift, g, g_ld, dw, db, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh, kw, 0,
                             sd_layout, nullptr, nullptr, nullptr, nullptr, y, y_ld, gcoef, stream);
}

// tem_conv3d_wgrad_gnorm for x of storage type st_x and g / y of storage type st_g (TEM_ST_*)
extern "C" int tem_conv3d_wgrad_gnorm_st(const void* x, int64_t x_ld, const float* scale, const float* shift, const void* g,
                                         int64_t g_ld, const void* y, int64_t y_ld, const float* gcoef, float* dw, float* db,
                                         void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh,
                                         int kw, int sd_layout, int st_x, int st_g, tem_stream_t stream) {
    TEM_REQUIRE(st_x >= 0 && st_x <= 2 && st_g >= 0 && st_g <= 2, "tem_conv3d_wgrad_gnorm_st: unknown storage type");
    TemStScope sc(st_x, st_g);
    return tem_conv3d_wgrad_gnorm((const float*)x, x_ld, scale, shift, (const float*)g, g_ld, (const float*)y, y_ld, gcoef, dw, db,
                                  ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh, kw, sd_layout, stream);
}
