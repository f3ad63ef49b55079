// conv1x1_stream.hip -- 1x1x1 convolution (torch_em/model/unet.py:455-458, the Conv3d(kernel_size=1) of every Upsampler,
// run by the engine at the LOW resolution) and its data gradient as a streaming split-bf16 GEMM: no halo, so none of the
// 3x3x3 patch machinery (LDS tile, barriers, 256-voxel patches) that ran these layers at 1.3-1.8 TB/s.
// A wave owns 32 voxels at a time: lane (voxel r, k-half kh) loads its 8 input channels of a 16-channel k-step straight
// from HBM (two 16-byte loads; a voxel's 128-byte line is used up by two k-steps of the same wave), splits them into the
// bf16 terms in registers and multiplies with the weight fragments of the existing TEM_WL_BF16X3 / BF16X6 / F16 packs
// (L1-resident: 12 ... 786 KB per layer).  No LDS, ~100 registers: 8+ waves per SIMD hide the load latency.
// HBM-bound: algorithmic bytes = 4 (Cin + Cout) per voxel (+ 4 Cout with a ReLU mask).
#include "tem_common.h"
#include "tem_act.h"
#include "conv_split.h"
#include "conv_internal.h"

// T: element type of x, y and ref.  A 16-bit T is also the operand type of the one-term modes (fp16 storage <-> fp16 operands,
// bf16 <-> bf16; the launcher checks): a lane's 8 channels of a k-step are ONE 16-byte load and go to the MFMA as loaded.
template <int NS, bool F16, int CT, typename T>
__global__ __launch_bounds__(256) void k_conv1x1_stream(const T* __restrict__ x, int64_t x_ld,
                                                        const unsigned short* __restrict__ wp,
                                                        const float* __restrict__ bias, T* __restrict__ y,
                                                        int64_t y_ld, const T* __restrict__ ref, int64_t ref_ld,
                                                        int64_t NV, int Cin, int Cout, int act, int64_t nmt,
                                                        unsigned* __restrict__ amax) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float amx = 0.f;
    const int r = lane & 31, kh = lane >> 5;
    const int ctg = blockIdx.y;                 // group of CT Cout tiles
    const int nks = Cin >> 4;
    const uint4* wfrag = reinterpret_cast<const uint4*>(wp) + lane;   // fragment f: wfrag[f * 64]
    for (int64_t mt = (int64_t)blockIdx.x * 4 + wv; mt < nmt; mt += (int64_t)gridDim.x * 4) {
        const int64_t v = mt * 32 + r;
        const bool vok = v < NV;
        const T* xr = x + (vok ? v : 0) * x_ld + kh * 8;
        floatx16 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[ct][k] = 0.f;
        for (int ks = 0; ks < nks; ++ks) {
            // operand terms of the 8 channels: NS bf16 terms (or one fp16 term)
            uint4 t[NS];
            if constexpr (sizeof(T) == 2) {
                static_assert(sizeof(T) == 4 || NS == 1, "16-bit storage: one-term modes only");
                t[0] = make_uint4(0u, 0u, 0u, 0u);
                if (vok) t[0] = *reinterpret_cast<const uint4*>(xr + ks * 16);
            } else {
                float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
                if (vok) {
                    a0 = act_ld4(xr + ks * 16);
                    a1 = act_ld4(xr + ks * 16 + 4);
                }
                float rem[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    unsigned wds[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned h = pk16<F16>(rem[2 * e], rem[2 * e + 1]);
                        wds[e] = h;
                        if (p + 1 < NS) {
                            rem[2 * e] -= lo16<F16>(h);
                            rem[2 * e + 1] -= hi16<F16>(h);
                        }
                    }
                    t[p] = make_uint4(wds[0], wds[1], wds[2], wds[3]);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int nt = ctg * CT + ct;
                if (nt * 32 < Cout) {
                    const uint4* f = wfrag + ((int64_t)(nt * nks + ks) * NS) * 64;
                    uint4 b[NS];
#pragma unroll
                    for (int p = 0; p < NS; ++p) b[p] = f[p * 64];
                    // all products of order < NS, smallest first: x_i * w_j with i + j < NS
#pragma unroll
                    for (int o = NS - 1; o >= 0; --o)
#pragma unroll
                        for (int i = 0; i <= o; ++i) acc[ct] = mfma16<F16>(b[o - i], t[i], acc[ct]);   // D = W X^T
                }
            }
        }
        // epilogue: D[row = channel][col = voxel] -- lane = voxel r, registers 4q .. 4q+3 = channels 8q + 4kh .. +3 of the
        // tile: one 16-byte store per (tile, q)
        if (vok) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int co = (ctg * CT + ct) * 32 + 8 * q4 + 4 * kh;
                    if (co < Cout) {
                        float4 o = make_float4(acc[ct][4 * q4], acc[ct][4 * q4 + 1], acc[ct][4 * q4 + 2], acc[ct][4 * q4 + 3]);
                        if (bias) {
                            const float4 b4 = *reinterpret_cast<const float4*>(bias + co);
                            o.x += b4.x;
                            o.y += b4.y;
                            o.z += b4.z;
                            o.w += b4.w;
                        }
                        o.x = act_apply_b(o.x, act);
                        o.y = act_apply_b(o.y, act);
                        o.z = act_apply_b(o.z, act);
                        o.w = act_apply_b(o.w, act);
                        if (ref) {
                            const float4 rv = act_ld4(ref + v * ref_ld + co);
                            o.x = rv.x > 0.f ? o.x : 0.f;
                            o.y = rv.y > 0.f ? o.y : 0.f;
                            o.z = rv.z > 0.f ? o.z : 0.f;
                            o.w = rv.w > 0.f ? o.w : 0.f;
                        }
                        act_st4(y + v * y_ld + co, o);   // (a non-temporal store here cost 30-80 %: the consumer reads y from L2 / MALL)
                        amx = tem_amax4(amx, o.x, o.y, o.z, o.w);
                    }
                }
            }
        }
    }
    if (amax) tem_amax_commit(amax, amx);
}

template <int NS, bool F16, typename T>
static void stream_launch(const T* x, int64_t x_ld, const float* wp, const float* bias, T* y, int64_t y_ld,
                          const T* ref, int64_t ref_ld, int64_t NV, int Cin, int Cout, int act, hipStream_t s) {
    const int64_t nmt = (NV + 31) / 32;
    const int ntile = Cout / 32;
    const int CT = ntile >= 2 ? 2 : 1;
    const int ngroups = (ntile + CT - 1) / CT;
    int64_t gx = (nmt + 3) / 4;
    if (gx > 8192) gx = 8192;   // grid-stride: 8 workgroups per CU keep the loads in flight
    const dim3 grid((unsigned)gx, (unsigned)ngroups);
    unsigned* const amax = tem_take_output_amax();
    if (CT == 2)
        hipLaunchKernelGGL((k_conv1x1_stream<NS, F16, 2, T>), grid, dim3(256), 0, s, x, x_ld, (const unsigned short*)wp, bias, y, y_ld,
                           ref, ref_ld, NV, Cin, Cout, act, nmt, amax);
    else
        hipLaunchKernelGGL((k_conv1x1_stream<NS, F16, 1, T>), grid, dim3(256), 0, s, x, x_ld, (const unsigned short*)wp, bias, y, y_ld,
                           ref, ref_ld, NV, Cin, Cout, act, nmt, amax);
}

// nsplit as in tem_conv_fwd_bf16x3: 2 = bf16x3, 3 = bf16x6, 5 = one fp16 term, 7 = one bf16 term.  false: not taken (pre-norm, statistics, the
// scaled fp16x3 layouts, sigmoid -- the patch kernel handles those).  Storage types (tem_call_st): fp32, or the 16-bit type
// that IS the operand type of the mode (fp16 with nsplit 5, bf16 with nsplit 7).
bool tem_conv1x1_stream(const float* x, int64_t x_ld, const float* scale, const float* wp, const float* bias, float* y,
                        int64_t y_ld, const float* ref, int64_t ref_ld, int64_t NV, int Cin, int Cout, int act, int nsplit,
                        const float* stat, hipStream_t s) {
    const int st = tem_call_st.x;
    if (tem_call_st.y != st) return false;
    if (st && !((st == 1 && nsplit == 5) || (st == 2 && nsplit == 7))) return false;
    const uintptr_t a8 = st ? 15 : 15, a4 = tem_st_align4(st) - 1;   // x: 16-byte loads either way; y / ref: vectors of 4 elements
    if (scale || stat || act == TEM_ACT_SIGMOID) return false;
    if (NV < 16384) return false;   // too few 32-voxel tiles to hide the k-loop's load latency: the split-K patch kernel wins
    if (Cin % 16 || Cout % 32 || (x_ld & (st ? 7 : 3)) || (reinterpret_cast<uintptr_t>(x) & a8)) return false;
    if ((y_ld & 3) || (reinterpret_cast<uintptr_t>(y) & a4) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) ||
        (ref && ((ref_ld & 3) || (reinterpret_cast<uintptr_t>(ref) & a4))))
        return false;
    if (st == 1)
        stream_launch<1, true, tem_f16>((const tem_f16*)x, x_ld, wp, bias, (tem_f16*)y, y_ld, (const tem_f16*)ref, ref_ld, NV, Cin, Cout, act, s);
    else if (st == 2)
        stream_launch<1, false, tem_bf16>((const tem_bf16*)x, x_ld, wp, bias, (tem_bf16*)y, y_ld, (const tem_bf16*)ref, ref_ld, NV, Cin, Cout, act, s);
    else if (nsplit == 2)
        stream_launch<2, false, float>(x, x_ld, wp, bias, y, y_ld, ref, ref_ld, NV, Cin, Cout, act, s);
    else if (nsplit == 3)
        stream_launch<3, false, float>(x, x_ld, wp, bias, y, y_ld, ref, ref_ld, NV, Cin, Cout, act, s);
    else if (nsplit == 5)
        stream_launch<1, true, float>(x, x_ld, wp, bias, y, y_ld, ref, ref_ld, NV, Cin, Cout, act, s);
    else if (nsplit == 7)
        stream_launch<1, false, float>(x, x_ld, wp, bias, y, y_ld, ref, ref_ld, NV, Cin, Cout, act, s);
    else
        return false;
    return true;
}
