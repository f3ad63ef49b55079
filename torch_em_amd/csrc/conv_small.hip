// conv_small.hip -- HBM-bound convolution kernels for the two ends of the U-Net where the
// matrix cores have nothing to chew on (first conv 13 flop/B, out_conv 0.9 flop/B):
//   * Cin == 1 first layer (forward + weight gradient; its data gradient is never needed),
//   * 1x1x1 projection to <= 16 channels (out_conv: forward + weight gradient).
// Each reads/writes its big tensor exactly once with 16-byte, fully coalesced accesses.
#include "tem_common.h"
#include "conv_internal.h"
#include "tem_act.h"

#ifndef TEM_SMALL_NT
#define TEM_SMALL_NT 0
#endif
template <typename T>
__device__ __forceinline__ void ST4(T* p, float4 v) {
#if TEM_SMALL_NT
    act_st4_nt(p, v);
#else
    act_st4(p, v);
#endif
}
// v rounded to the storage type T (the statistics by-products describe the tensor AS STORED)
template <typename T>
__device__ __forceinline__ float4 act_round4(float4 v) {
    if constexpr (sizeof(T) == 2) {
        const unsigned p0 = act_pk<T>(v.x, v.y), p1 = act_pk<T>(v.z, v.w);
        return make_float4(act_lo<T>(p0), act_hi<T>(p0), act_lo<T>(p1), act_hi<T>(p1));
    } else
        return v;
}

typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef TEM_C1_ABL
#define TEM_C1_ABL 0   // profiling ablations of the Cin = 1 forward row path: 1 = no output stores, 2 = 3 taps instead of 27
#endif

__device__ __forceinline__ float act_apply_s(float v, int act) {
    if (act == TEM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TEM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// ---------------------------------------------------------------------------
// Cin == 1 forward.  Workgroup = 4x8x8 output patch; the (tiny) halo tile and the weights
// live in LDS; thread <-> (voxel, 4 output channels): 27 LDS broadcasts + 27 float4 weight
// reads + 108 FMA, one 16-byte store.
// ---------------------------------------------------------------------------
template <int KD, int KH, int KW, int CIN, typename EX, typename EY>
__global__ __launch_bounds__(256) void k_conv_fwd_cin1(const EX* __restrict__ x, int64_t x_ld,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ w /*[tap][ci][co]*/,
                                                       const float* __restrict__ bias, EY* __restrict__ y,
                                                       int64_t y_ld, int N, int D, int H, int W, int Cout, int act,
                                                       int nZ, int nY, int nX, float* __restrict__ stat) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int TZ = 4, TY = 8, TX = 8;
    constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1, HV = HZ * HY * HX;
    constexpr int HXP = ((HX + 3) / 4) * 4;   // x rows start on 16 bytes: the row path reads them with ds_read_b128
    constexpr int HVP = HZ * HY * HXP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lx = lds;              // [CIN][HZ][HY][HXP]
    float* lw = lds + CIN * HVP;  // [NT][CIN][Cout]
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int ptx = bid % nX;
    bid /= nX;
    const int pty = bid % nY;
    bid /= nY;
    const int ptz = bid % nZ;
    const int n = bid / nZ;
    const int z0 = ptz * TZ, y0 = pty * TY, x0 = ptx * TX;
    for (int i = tid; i < NT * CIN * Cout; i += 256) lw[i] = w[i];
    for (int item = tid; item < HV * CIN; item += 256) {
        const int hv = item / CIN, ci = item % CIN;
        const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
        const int gz = z0 + hz - PZ, gy = y0 + hy - PY, gx = x0 + hx - PX;
        float v = 0.f;
        if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            v = act_ld1(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * x_ld + ci);
            if (scale) v = fmaf(v, scale[n * CIN + ci], shift[n * CIN + ci]);
        }
        lx[ci * HVP + (hz * HY + hy) * HXP + hx] = v;
    }
    __syncthreads();
    const int cq = Cout >> 2;
    float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f), ssq = ssum;  // fused statistics (cq divides 64: a thread keeps its quad)
    for (int item = tid; item < 256 * cq; item += 256) {
        const int p = item / cq, q = item % cq;
        const int pz = p / (TY * TX), py = (p / TX) % TY, px = p % TX;
        const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
        if (gz >= D || gy >= H || gx >= W) continue;
        float4 acc = bias ? *reinterpret_cast<const float4*>(bias + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xb = lx + (pz * HY + py) * HXP + px;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float xv = xb[ci * HVP + (tz * HY + ty) * HXP + tx];
                const float4 wv = *reinterpret_cast<const float4*>(lw + (tap * CIN + ci) * Cout + q * 4);
                acc.x = fmaf(xv, wv.x, acc.x);
                acc.y = fmaf(xv, wv.y, acc.y);
                acc.z = fmaf(xv, wv.z, acc.z);
                acc.w = fmaf(xv, wv.w, acc.w);
            }
        }
        acc.x = act_apply_s(acc.x, act);
        acc.y = act_apply_s(acc.y, act);
        acc.z = act_apply_s(acc.z, act);
        acc.w = act_apply_s(acc.w, act);
        acc = act_round4<EY>(acc);
        const int64_t v = (((int64_t)n * D + gz) * H + gy) * W + gx;
        ST4(y + v * y_ld + q * 4, acc);
        ssum.x += acc.x; ssum.y += acc.y; ssum.z += acc.z; ssum.w += acc.w;
        ssq.x = fmaf(acc.x, acc.x, ssq.x); ssq.y = fmaf(acc.y, acc.y, ssq.y);
        ssq.z = fmaf(acc.z, acc.z, ssq.z); ssq.w = fmaf(acc.w, acc.w, ssq.w);
    }
    if (stat) {  // per (sample, patch, channel) partial sums for the next norm (see tem_conv3d_fwd_stats)
        float vals[8] = {ssum.x, ssum.y, ssum.z, ssum.w, ssq.x, ssq.y, ssq.z, ssq.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = cq; o < 64; o <<= 1) vals[j] += __shfl_xor(vals[j], o, 64);  // lanes sharing q sit cq apart
        __syncthreads();
        float* red = lds;  // [4 waves][cq][8]
        const int wv = tid >> 6, lane = tid & 63;
        if (lane < cq)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(wv * cq + lane) * 8 + j] = vals[j];
        __syncthreads();
        if (tid < Cout) {
            const int q = tid >> 2, j = tid & 3;
            float a = 0.f, b = 0.f;
            for (int w4 = 0; w4 < 4; ++w4) {
                a += red[(w4 * cq + q) * 8 + j];
                b += red[(w4 * cq + q) * 8 + 4 + j];
            }
            const int64_t patch = ((int64_t)ptz * nY + pty) * nX + ptx;
            float* dst = stat + ((((int64_t)n * nZ * nY * nX) + patch) * Cout + tid) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

// ---------------------------------------------------------------------------
// Cin == 1, 3x3 in-plane taps, Cout = 2 * 2^k <= 128: the first layer of every U-Net here.  Workgroup = 4x8x32 output
// tile (4 x-segments of 8 share one halo staging, one weight load and one statistics reduction: at 4x8x8 those fixed
// costs were half the kernel).  thread <-> (x-row, channel PAIR): the pair's 27 weight pairs and a segment's 8
// accumulator pairs live in registers (v_pk_fma_f32: two FMAs per lane and issue slot); per (tz, ty) the 10 input values
// of the segment come from LDS in three wide reads and feed 3 taps x 8 voxels: 27 packed FMAs and 3.4 LDS reads per
// voxel-pair where the voxel-per-thread kernel had 54 scalar FMAs and 27 reads.
// ---------------------------------------------------------------------------
#define C1R_TXW 32
template <int KD, typename EX, typename EY>
__global__ __launch_bounds__(256) void k_conv_fwd_c1rows(const EX* __restrict__ x, int64_t x_ld,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ w /*[tap][co]*/,
                                                         const float* __restrict__ bias, EY* __restrict__ y,
                                                         int64_t y_ld, int N, int D, int H, int W, int Cout, int act,
                                                         int nZ, int nY, int nX, float* __restrict__ stat) {
    constexpr int NT = KD * 9, PZ = KD / 2;
    constexpr int TZ = 4, TY = 8, TX = 8, HZ = TZ + KD - 1, HY = TY + 2, HXW = C1R_TXW + 2, HXP = C1R_TXW + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lx = lds;                  // [HZ][HY][HXP]
    float* lw = lds + HZ * HY * HXP;  // [NT][Cout]
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int ptx = bid % nX;
    bid /= nX;
    const int pty = bid % nY;
    bid /= nY;
    const int ptz = bid % nZ;
    const int n = bid / nZ;
    const int z0 = ptz * TZ, y0 = pty * TY, x0 = ptx * C1R_TXW;
    for (int i = tid; i < NT * Cout; i += 256) lw[i] = w[i];
    {
        float sc = 1.f, sf = 0.f;
        if (scale) {
            sc = scale[n];
            sf = shift[n];
        }
        for (int hv = tid; hv < HZ * HY * HXW; hv += 256) {
            const int hr = hv / HXW, hx = hv % HXW, hz = hr / HY, hy = hr % HY;
            const int gz = z0 + hz - PZ, gy = y0 + hy - 1, gx = x0 + hx - 1;
            float v = 0.f;
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = fmaf(act_ld1(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * x_ld), sc, sf);
            lx[hr * HXP + hx] = v;
        }
    }
    __syncthreads();
    const int cp = Cout >> 1;
    const int q = tid % cp;
    f2 wr[NT];
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) wr[tap] = *reinterpret_cast<const f2*>(lw + tap * Cout + q * 2);
    const f2 b2 = bias ? *reinterpret_cast<const f2*>(bias + q * 2) : f2{0.f, 0.f};
    f2 ss = {0.f, 0.f}, sq = ss;
    // (Storing a segment's outputs one by one behind the next segment's FMA groups instead of back to back at its end was
    // measured equal -- 182 vs 176 us at 2x128^3: FMA time and store time add up either way, the issuing waves wait on
    // the write path.)
#pragma unroll 1
    for (int row = tid / cp; row < TZ * TY; row += 256 / cp) {
        const int pz = row / TY, py = row % TY;
        const int gz = z0 + pz, gy = y0 + py;
        if (gz >= D || gy >= H) continue;
#pragma unroll 1
        for (int xs = 0; xs < C1R_TXW && x0 + xs < W; xs += TX) {
            f2 acc[TX];
#pragma unroll
            for (int px = 0; px < TX; ++px) acc[px] = b2;
#pragma unroll
            for (int tz = 0; tz < ((TEM_C1_ABL & 2) ? 1 : KD); ++tz)
#pragma unroll
                for (int ty = 0; ty < ((TEM_C1_ABL & 2) ? 1 : 3); ++ty) {
                    const float* xr = lx + ((pz + tz) * HY + py + ty) * HXP + xs;
                    const float4 a = *reinterpret_cast<const float4*>(xr);
                    const float4 b = *reinterpret_cast<const float4*>(xr + 4);
                    const float2 c = *reinterpret_cast<const float2*>(xr + 8);
                    const float xv[10] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y};
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int px = 0; px < TX; ++px)
                            acc[px] = __builtin_elementwise_fma(wr[(tz * 3 + ty) * 3 + tx], f2{xv[px + tx], xv[px + tx]}, acc[px]);
                }
            const int64_t v0 = (((int64_t)n * D + gz) * H + gy) * W + x0 + xs;
            // pin the accumulators here: otherwise the 27 FMAs of a voxel sink into its store branch below and all
            // 90 row values stay live across them (183 VGPRs)
#pragma unroll
            for (int px = 0; px < TX; ++px) asm volatile("" : "+v"(acc[px]));
#pragma unroll
            for (int px = 0; px < TX; ++px) {
                if (x0 + xs + px >= W) break;
                f2 o = {act_apply_s(acc[px].x, act), act_apply_s(acc[px].y, act)};
                if constexpr (sizeof(EY) == 2) {   // the statistics describe the tensor as stored
                    const unsigned pk = act_pk<EY>(o.x, o.y);
                    o = f2{act_lo<EY>(pk), act_hi<EY>(pk)};
                }
#if !(TEM_C1_ABL & 1)
                act_st2_nt(y + (v0 + px) * y_ld + q * 2, o);
#endif
                ss += o;
                sq = __builtin_elementwise_fma(o, o, sq);
            }
        }
    }
    if (stat) {  // per (sample, tile, channel) partial sums for the next norm (see tem_conv3d_fwd_stats)
        float vals[4] = {ss.x, ss.y, sq.x, sq.y};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int o = cp; o < 64; o <<= 1) vals[j] += __shfl_xor(vals[j], o, 64);  // lanes sharing q sit cp apart
        __syncthreads();
        float* red = lds;  // [4 waves][cp][4]
        const int wv = tid >> 6, lane = tid & 63;
        if (lane < cp)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wv * cp + lane) * 4 + j] = vals[j];
        __syncthreads();
        if (tid < Cout) {
            const int pr = tid >> 1, j = tid & 1;
            float sa = 0.f, sb = 0.f;
            for (int w4 = 0; w4 < 4; ++w4) {
                sa += red[(w4 * cp + pr) * 4 + j];
                sb += red[(w4 * cp + pr) * 4 + 2 + j];
            }
            const int64_t patch = ((int64_t)ptz * nY + pty) * nX + ptx;
            float* dst = stat + ((((int64_t)n * nZ * nY * nX) + patch) * Cout + tid) * 2;
            dst[0] = sa;
            dst[1] = sb;
        }
    }
}

static inline bool c1rows_ok(int Cin, int Cout, int kh, int kw) {
    const int cp = Cout / 2;
    return Cin == 1 && kh == 3 && kw == 3 && Cout % 2 == 0 && cp >= 1 && cp <= 64 && (cp & (cp - 1)) == 0;
}

// blocks per sample of the fused statistics, 0 when the cin1 kernel does not take this shape / cannot keep a channel
// quad per thread
int64_t tem_conv_fwd_cin1_stat_blocks(int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    const int cq = Cout / 4;
    if (Cin > 4 || Cout % 4 || Cin * Cout > 128 || cq > 64 || (cq & (cq - 1))) return 0;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    if (key != 7 && key != 3) return 0;
    const int tx = c1rows_ok(Cin, Cout, kh, kw) ? C1R_TXW : 8;
    return (int64_t)((D + 3) / 4) * ((H + 7) / 8) * ((W + tx - 1) / tx);
}

bool tem_conv_fwd_cin1(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w,
                       const float* bias, float* y, int64_t y_ld, const float* ref, int N, int D, int H, int W,
                       int Cin, int Cout, int kd, int kh, int kw, int act, float* stat, hipStream_t s) {
    // Cin 2..4 (RGB / multi-channel raw data) share the kernel; weights stay in LDS, so Cin * Cout is bounded
    if (Cin > 4 || Cout % 4 || Cin * Cout > 128 || ref || y_ld % 4 || ((uintptr_t)y % tem_st_align4(tem_call_st.y)) ||
        (bias && ((uintptr_t)bias % 16)))
        return false;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    if (key != 7 && key != 3) return false;
    if (c1rows_ok(Cin, Cout, kh, kw) && y_ld % 2 == 0) {
        const int nZ = (D + 3) / 4, nY = (H + 7) / 8, nX = (W + C1R_TXW - 1) / C1R_TXW;
        const int64_t nblk = (int64_t)N * nZ * nY * nX;
        const int kdv = key == 7 ? 3 : 1;
        const size_t ldsb = (size_t)((kdv + 3) * 10 * (C1R_TXW + 4) + kdv * 9 * Cout) * sizeof(float);
        TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, return false, {
            if (key == 7)
                hipLaunchKernelGGL((k_conv_fwd_c1rows<3, TX, TY>), dim3((unsigned)nblk), dim3(256), ldsb, s, (const TX*)x, x_ld, scale,
                                   shift, w, bias, (TY*)y, y_ld, N, D, H, W, Cout, act, nZ, nY, nX, stat);
            else
                hipLaunchKernelGGL((k_conv_fwd_c1rows<1, TX, TY>), dim3((unsigned)nblk), dim3(256), ldsb, s, (const TX*)x, x_ld, scale,
                                   shift, w, bias, (TY*)y, y_ld, N, D, H, W, Cout, act, nZ, nY, nX, stat);
        });
        return true;
    }
    const int nZ = (D + 3) / 4, nY = (H + 7) / 8, nX = (W + 7) / 8;
    const int64_t nblk = (int64_t)N * nZ * nY * nX;
#define C1(KD_, CI)                                                                                                   \
    case CI: {                                                                                                        \
        size_t ldsb = (size_t)(CI * ((KD_ + 3) * 10 * 12) + KD_ * 9 * CI * Cout) * sizeof(float);                    \
        TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, return false,                                            \
                       hipLaunchKernelGGL((k_conv_fwd_cin1<KD_, 3, 3, CI, TX, TY>), dim3((unsigned)nblk), dim3(256), ldsb, s, \
                                          (const TX*)x, x_ld, scale, shift, w, bias, (TY*)y, y_ld, N, D, H, W, Cout, act, nZ, nY, nX, stat)); \
    } break;
    if (key == 7) {
        switch (Cin) { C1(3, 1) C1(3, 2) C1(3, 3) C1(3, 4) }
    } else {
        switch (Cin) { C1(1, 1) C1(1, 2) C1(1, 3) C1(1, 4) }
    }
#undef C1
    return true;
}

// ---------------------------------------------------------------------------
// Cout == 1: the data gradient of a first layer (Cin = 1) w.r.t. its input -- needed when the norm in front of it
// has affine parameters (GroupNorm, BatchNorm; reference model/unet.py:391-406).  y[v] = sum_tap sum_ci w[tap][ci]
// x[v+tap][ci]: 864 FMAs per output voxel over a halo tile staged in LDS per 16-channel chunk (80 B per voxel = 5
// 16-byte slots: conflict-free float4 reads).  Was the generic VALU kernel: 17.5 ms for 2x128^3x32 -> 1; now HBM/VALU
// balanced (one read of x).
// ---------------------------------------------------------------------------
template <int KD, int KH, int KW, typename EX, typename EY>
__global__ __launch_bounds__(256) void k_conv_fwd_cout1(const EX* __restrict__ x, int64_t x_ld,
                                                        const float* __restrict__ w /*[tap][ci]*/,
                                                        const float* __restrict__ bias, EY* __restrict__ y,
                                                        int64_t y_ld, int N, int D, int H, int W, int Cin, int act,
                                                        int nZ, int nY, int nX) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int TZ = 4, TY = 8, TX = 8;
    constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1, HV = HZ * HY * HX;
    constexpr int LS = 20;  // floats per halo voxel: 16 channels + 4 pad
    __shared__ __attribute__((aligned(16))) float lx[HV * LS];
    __shared__ __attribute__((aligned(16))) float lw[NT * 16];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int ptx = bid % nX;
    bid /= nX;
    const int pty = bid % nY;
    bid /= nY;
    const int ptz = bid % nZ;
    const int n = bid / nZ;
    const int z0 = ptz * TZ, y0 = pty * TY, x0 = ptx * TX;
    const int pz = tid / (TY * TX), py = (tid / TX) % TY, px = tid % TX;
    float acc = 0.f;
    for (int c16 = 0; c16 < Cin; c16 += 16) {
        __syncthreads();
        for (int i = tid; i < NT * 16; i += 256) lw[i] = w[(int64_t)(i / 16) * Cin + c16 + (i % 16)];
        for (int item = tid; item < HV * 4; item += 256) {
            const int hv = item >> 2, c4 = item & 3;
            const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
            const int gz = z0 + hz - PZ, gy = y0 + hy - PY, gx = x0 + hx - PX;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = act_ld4(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * x_ld + c16 + c4 * 4);
            *reinterpret_cast<float4*>(lx + hv * LS + c4 * 4) = v;
        }
        __syncthreads();
        const float* xb = lx + ((pz * HY + py) * HX + px) * LS;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
            const float* xp = xb + ((tz * HY + ty) * HX + tx) * LS;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 xv = *reinterpret_cast<const float4*>(xp + c4 * 4);
                const float4 wv = *reinterpret_cast<const float4*>(lw + tap * 16 + c4 * 4);
                acc = fmaf(xv.x, wv.x, acc);
                acc = fmaf(xv.y, wv.y, acc);
                acc = fmaf(xv.z, wv.z, acc);
                acc = fmaf(xv.w, wv.w, acc);
            }
        }
    }
    const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
    if (gz < D && gy < H && gx < W) {
        if (bias) acc += bias[0];
        act_st1(y + ((((int64_t)n * D + gz) * H + gy) * W + gx) * y_ld, act_apply_s(acc, act));
    }
}

bool tem_conv_fwd_cout1(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* w,
                        const float* bias, float* y, int64_t y_ld, const float* ref, int N, int D, int H, int W,
                        int Cin, int Cout, int kd, int kh, int kw, int act, hipStream_t s) {
    (void)shift;
    if (Cout != 1 || Cin % 16 || scale || ref || x_ld % 4 || ((uintptr_t)x % 16)) return false;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    if (key != 7 && key != 3) return false;
    const int nZ = (D + 3) / 4, nY = (H + 7) / 8, nX = (W + 7) / 8;
    const int64_t nblk = (int64_t)N * nZ * nY * nX;
    TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, return false, {
        if (key == 7)
            hipLaunchKernelGGL((k_conv_fwd_cout1<3, 3, 3, TX, TY>), dim3((unsigned)nblk), dim3(256), 0, s, (const TX*)x, x_ld, w, bias,
                               (TY*)y, y_ld, N, D, H, W, Cin, act, nZ, nY, nX);
        else
            hipLaunchKernelGGL((k_conv_fwd_cout1<1, 3, 3, TX, TY>), dim3((unsigned)nblk), dim3(256), 0, s, (const TX*)x, x_ld, w, bias,
                               (TY*)y, y_ld, N, D, H, W, Cin, act, nZ, nY, nX);
    });
    return true;
}

// ---------------------------------------------------------------------------
// Cin == 1 weight gradient: dw[tap][co] = sum_v xhat[v+tap] * g[v][co]  (+ db[co] = sum_v g).
// Persistent workgroups walk 4x8x8 patches; thread <-> (x-row of 8 voxels, co PAIR) keeps NT + 1 accumulator pairs in
// registers (v_pk_fma_f32), reads its 8 g pairs once with 8-byte loads (16 lanes = one 128-byte line per voxel) and the
// row's 10 input values per (tz, ty) from LDS in three wide reads; one reduction at the end.
// ---------------------------------------------------------------------------
template <int KD, int KH, int KW, typename EX, typename EG>
__global__ __launch_bounds__(256) void k_conv_wgrad_cin1(const EX* __restrict__ x, int64_t x_ld,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const EG* __restrict__ g, int64_t g_ld,
                                                         float* __restrict__ part /*[grid][NT+1][Cout]*/, int N,
                                                         int D, int H, int W, int Cout, int P, int nZ, int nY,
                                                         int nX, int sstride /*Cin: scale[n*Cin] of this channel*/,
                                                         const EG* __restrict__ gnx, int64_t gnx_ld,
                                                         const float* __restrict__ gcoef) {
    static_assert(KH == 3 && KW == 3, "row layout: 3x3 in-plane taps");
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = 1, PX = 1;
    constexpr int TZ = 4, TY = 8, TX = 8;
    constexpr int HZ = TZ + KD - 1, HY = TY + 2, HX = TX + 2, HXP = 12, HV = HZ * HY * HX;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // max(HZ*HY*HXP, 4*(NT+1)*Cout)
    const int tid = threadIdx.x;
    const int cp = Cout >> 1;       // power of two, <= 32
    const int q = tid % cp, rl = tid / cp;
    const int nrl = 256 / cp;       // rows per pass
    f2 acc[NT + 1];
#pragma unroll
    for (int t = 0; t <= NT; ++t) acc[t] = f2{0.f, 0.f};
    for (int pidx = blockIdx.x; pidx < P; pidx += gridDim.x) {
        int b = pidx;
        const int ptx = b % nX;
        b /= nX;
        const int pty = b % nY;
        b /= nY;
        const int ptz = b % nZ;
        const int n = b / nZ;
        const int z0 = ptz * TZ, y0 = pty * TY, x0 = ptx * TX;
        float sc = 1.f, sf = 0.f;
        if (scale) {
            sc = scale[n * sstride];
            sf = shift[n * sstride];
        }
        // g may still be the RAW data gradient behind the norm that follows this conv's ReLU (tem_conv3d_wgrad_gnorm):
        // g := (y > 0) ? a*g - m1 - (y - mean)*m2r : 0 with y = this conv's output (gnx), applied while loading
        float4 kc[2];
        if (gcoef) {
#pragma unroll
            for (int j = 0; j < 2; ++j) kc[j] = *reinterpret_cast<const float4*>(gcoef + ((int64_t)n * Cout + q * 2 + j) * 4);
        }
        __syncthreads();
        for (int hv = tid; hv < HV; hv += 256) {
            const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
            const int gz = z0 + hz - PZ, gy = y0 + hy - PY, gx = x0 + hx - PX;
            float v = 0.f;
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = fmaf(act_ld1(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * x_ld), sc, sf);
            lds[(hz * HY + hy) * HXP + hx] = v;
        }
        __syncthreads();
        // 16-bit g: the loads of TWO row passes are issued before the first one is consumed (same registers as one pass of
        // fp32 pairs: the raw words are widened on use) -- the kernel waits for memory once per patch instead of twice
        constexpr int NPASS = sizeof(EG) == 2 ? 2 : 1;
        for (int row0 = rl; row0 < TZ * TY; row0 += NPASS * nrl) {
            // (16-bit: the raw 4-byte words wait in registers, widened when their pass is computed)
            using Raw = std::conditional_t<sizeof(EG) == 2, unsigned, f2>;
            Raw gvp[NPASS][TX], yvp[NPASS][TX];
            bool live[NPASS];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int row = row0 + ps * nrl;
                const int pz = row / TY, py = row % TY;
                const int gz = z0 + pz, gy = y0 + py;
                live[ps] = row < TZ * TY && gz < D && gy < H;
                const int64_t v0 = live[ps] ? (((int64_t)n * D + gz) * H + gy) * W + x0 : (((int64_t)n * D + z0) * H + y0) * W + x0;
#pragma unroll
                for (int px = 0; px < TX; ++px) {     // branch-free: a voxel beyond W reads the row's first one and is zeroed
                    const int pc = x0 + px < W ? px : 0;
                    if constexpr (sizeof(EG) == 2) {
                        gvp[ps][px] = *reinterpret_cast<const unsigned*>(g + (v0 + pc) * g_ld + q * 2);
                        yvp[ps][px] = gcoef ? *reinterpret_cast<const unsigned*>(gnx + (v0 + pc) * gnx_ld + q * 2) : 0u;
                    } else {
                        gvp[ps][px] = act_ld2(g + (v0 + pc) * g_ld + q * 2);
                        yvp[ps][px] = gcoef ? act_ld2(gnx + (v0 + pc) * gnx_ld + q * 2) : f2{0.f, 0.f};
                    }
                }
            }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
            if (!live[ps]) continue;
            const int row = row0 + ps * nrl;
            const int pz = row / TY, py = row % TY;
            f2 gv[TX], yv[TX];
#pragma unroll
            for (int px = 0; px < TX; ++px) {
                if constexpr (sizeof(EG) == 2) {
                    gv[px] = f2{act_lo<EG>(gvp[ps][px]), act_hi<EG>(gvp[ps][px])};
                    yv[px] = f2{act_lo<EG>(yvp[ps][px]), act_hi<EG>(yvp[ps][px])};
                } else {
                    gv[px] = gvp[ps][px];
                    yv[px] = yvp[ps][px];
                }
            }
#pragma unroll
            for (int px = 0; px < TX; ++px)
                if (x0 + px >= W) gv[px] = f2{0.f, 0.f};
            if (gcoef) {
#pragma unroll
                for (int px = 0; px < TX; ++px) {
                    gv[px].x = yv[px].x > 0.f ? kc[0].x * gv[px].x - kc[0].y - (yv[px].x - kc[0].w) * kc[0].z : 0.f;
                    gv[px].y = yv[px].y > 0.f ? kc[1].x * gv[px].y - kc[1].y - (yv[px].y - kc[1].w) * kc[1].z : 0.f;
                    if (x0 + px >= W) gv[px] = f2{0.f, 0.f};
                }
            }
#pragma unroll
            for (int tz = 0; tz < KD; ++tz)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    const float* xr = lds + ((pz + tz) * HY + py + ty) * HXP;
                    const float4 a = *reinterpret_cast<const float4*>(xr);
                    const float4 bq = *reinterpret_cast<const float4*>(xr + 4);
                    const float2 c = *reinterpret_cast<const float2*>(xr + 8);
                    const float xv[10] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, c.x, c.y};
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int px = 0; px < TX; ++px)
                            acc[(tz * 3 + ty) * 3 + tx] = __builtin_elementwise_fma(gv[px], f2{xv[px + tx], xv[px + tx]},
                                                                                    acc[(tz * 3 + ty) * 3 + tx]);
                }
#pragma unroll
            for (int px = 0; px < TX; ++px) acc[NT] += gv[px];
            }
        }
    }
    // reduce over the row lanes: lanes sharing q inside a wave sit cp apart
#pragma unroll
    for (int t = 0; t <= NT; ++t) {
        for (int o = cp; o < 64; o <<= 1) {
            acc[t].x += __shfl_xor(acc[t].x, o, 64);
            acc[t].y += __shfl_xor(acc[t].y, o, 64);
        }
    }
    __syncthreads();
    const int wv = tid >> 6, lane = tid & 63;
    if (lane < cp) {
#pragma unroll
        for (int t = 0; t <= NT; ++t) *reinterpret_cast<f2*>(lds + ((wv * (NT + 1) + t) * Cout) + lane * 2) = acc[t];
    }
    __syncthreads();
    for (int i = tid; i < (NT + 1) * Cout; i += 256) {
        float s = lds[i] + lds[(NT + 1) * Cout + i] + lds[2 * (NT + 1) * Cout + i] + lds[3 * (NT + 1) * Cout + i];
        part[(int64_t)blockIdx.x * (NT + 1) * Cout + i] = s;
    }
}

// out[i] = sum_c part[c][i]  (fp64 accumulation; chunk-parallel so that small n is not latency-bound)
__global__ __launch_bounds__(512) void k_reduce_slabs(const float* __restrict__ part, int nchunks, int64_t n,
                                                      int64_t chunk_stride, float* __restrict__ out) {
    __shared__ double sh[8][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < n; i0 += (int64_t)gridDim.x * 64) {
        const int64_t i = i0 + tx;
        double s = 0.0;
        if (i < n) {
            // four independent loads in flight per thread (fixed order: still deterministic)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int c = ty;
            for (; c + 24 < nchunks; c += 32) {
                s += (double)part[(int64_t)c * chunk_stride + i];
                s1 += (double)part[(int64_t)(c + 8) * chunk_stride + i];
                s2 += (double)part[(int64_t)(c + 16) * chunk_stride + i];
                s3 += (double)part[(int64_t)(c + 24) * chunk_stride + i];
            }
            for (; c < nchunks; c += 8) s += (double)part[(int64_t)c * chunk_stride + i];
            s = (s + s1) + (s2 + s3);
        }
        sh[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && i < n) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[k][tx];
            out[i] = (float)a;
        }
        __syncthreads();
    }
}

void tem_reduce_slabs(const float* part, int nchunks, int64_t n, int64_t chunk_stride, float* out, hipStream_t s) {
    int64_t nb = tem_cdiv(n, 64);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)nb), dim3(512), 0, s, part, nchunks, n, chunk_stride, out);
}

// weight-gradient merge that also converts the kernels' [tap][ci][co] order into the reference's
// state_dict order [co][ci][tap] (fuses the former k_unpack_wgrad pass)
__global__ __launch_bounds__(512) void k_reduce_slabs_sd(const float* __restrict__ part, int nchunks, int ntaps, int Cin,
                                                         int Cout, int64_t chunk_stride, float* __restrict__ out,
                                                         int nb_w, const float* __restrict__ dbpart, int db_chunks,
                                                         float* __restrict__ db, int64_t db_stride) {
    __shared__ double sh[8][64];
    const int64_t n = (int64_t)ntaps * Cin * Cout;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if ((int)blockIdx.x >= nb_w) {  // the bias gradient rides along: db[co] = sum_c dbpart[c][co] (one launch less)
        const int co = ((int)blockIdx.x - nb_w) * 64 + tx;
        double s = 0.0;
        if (co < Cout)
            for (int c = ty; c < db_chunks; c += 8) s += (double)dbpart[(int64_t)c * db_stride + co];
        sh[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && co < Cout) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[k][tx];
            db[co] = (float)a;
        }
        return;
    }
    for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < n; i0 += (int64_t)nb_w * 64) {
        const int64_t i = i0 + tx;
        double s = 0.0;
        if (i < n) {
            // four independent loads in flight per thread (fixed order: still deterministic)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int c = ty;
            for (; c + 24 < nchunks; c += 32) {
                s += (double)part[(int64_t)c * chunk_stride + i];
                s1 += (double)part[(int64_t)(c + 8) * chunk_stride + i];
                s2 += (double)part[(int64_t)(c + 16) * chunk_stride + i];
                s3 += (double)part[(int64_t)(c + 24) * chunk_stride + i];
            }
            for (; c < nchunks; c += 8) s += (double)part[(int64_t)c * chunk_stride + i];
            s = (s + s1) + (s2 + s3);
        }
        sh[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && i < n) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[k][tx];
            const int co = (int)(i % Cout);
            const int64_t r = i / Cout;
            const int ci = (int)(r % Cin), tap = (int)(r / Cin);
            out[((int64_t)co * Cin + ci) * ntaps + tap] = (float)a;
        }
        __syncthreads();
    }
}

// one input channel (ci) of a [tap][CinT][co] (or state_dict [co][CinT][tap]) gradient from [chunk][tap][co] slabs
__global__ __launch_bounds__(512) void k_reduce_slabs_ci(const float* __restrict__ part, int nchunks, int ntaps, int CinT,
                                                         int ci, int Cout, int64_t chunk_stride, float* __restrict__ out,
                                                         int sd_layout) {
    __shared__ double sh[8][64];
    const int64_t n = (int64_t)ntaps * Cout;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < n; i0 += (int64_t)gridDim.x * 64) {
        const int64_t i = i0 + tx;
        double s = 0.0;
        if (i < n) {
            // four independent loads in flight per thread (fixed order: still deterministic)
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int c = ty;
            for (; c + 24 < nchunks; c += 32) {
                s += (double)part[(int64_t)c * chunk_stride + i];
                s1 += (double)part[(int64_t)(c + 8) * chunk_stride + i];
                s2 += (double)part[(int64_t)(c + 16) * chunk_stride + i];
                s3 += (double)part[(int64_t)(c + 24) * chunk_stride + i];
            }
            for (; c < nchunks; c += 8) s += (double)part[(int64_t)c * chunk_stride + i];
            s = (s + s1) + (s2 + s3);
        }
        sh[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && i < n) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[k][tx];
            const int co = (int)(i % Cout), tap = (int)(i / Cout);
            if (sd_layout)
                out[((int64_t)co * CinT + ci) * ntaps + tap] = (float)a;
            else
                out[((int64_t)tap * CinT + ci) * Cout + co] = (float)a;
        }
        __syncthreads();
    }
}

// the same with 16-byte loads: a thread owns 4 consecutive outputs (the 4-byte version ran at 2.4 TB/s)
__global__ __launch_bounds__(512) void k_reduce_slabs_sd4(const float* __restrict__ part, int nchunks, int ntaps, int Cin,
                                                          int Cout, int64_t chunk_stride, float* __restrict__ out,
                                                          int nb_w, const float* __restrict__ dbpart, int db_chunks,
                                                          float* __restrict__ db, int64_t db_stride) {
    __shared__ double sh[8][64][4];
    const int64_t n = (int64_t)ntaps * Cin * Cout;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if ((int)blockIdx.x >= nb_w) {
        const int co = ((int)blockIdx.x - nb_w) * 64 + tx;
        double s = 0.0;
        if (co < Cout)
            for (int c = ty; c < db_chunks; c += 8) s += (double)dbpart[(int64_t)c * db_stride + co];
        sh[ty][tx][0] = s;
        __syncthreads();
        if (ty == 0 && co < Cout) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[k][tx][0];
            db[co] = (float)a;
        }
        return;
    }
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += (int64_t)nb_w * 256) {
        const int64_t i = i0 + tx * 4;
        double s[4] = {0.0, 0.0, 0.0, 0.0}, r[4] = {0.0, 0.0, 0.0, 0.0};
        if (i < n) {
            int c = ty;
            for (; c + 56 < nchunks; c += 64) {  // eight 16-byte loads in flight: the loop is latency-bound
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(part + (int64_t)(c + 8 * k) * chunk_stride + i);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    s[0] += (double)v[k].x; s[1] += (double)v[k].y; s[2] += (double)v[k].z; s[3] += (double)v[k].w;
                    r[0] += (double)v[k + 1].x; r[1] += (double)v[k + 1].y; r[2] += (double)v[k + 1].z; r[3] += (double)v[k + 1].w;
                }
            }
            for (; c + 8 < nchunks; c += 16) {
                const float4 a = *reinterpret_cast<const float4*>(part + (int64_t)c * chunk_stride + i);
                const float4 b = *reinterpret_cast<const float4*>(part + (int64_t)(c + 8) * chunk_stride + i);
                s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
                r[0] += (double)b.x; r[1] += (double)b.y; r[2] += (double)b.z; r[3] += (double)b.w;
            }
            if (c < nchunks) {
                const float4 a = *reinterpret_cast<const float4*>(part + (int64_t)c * chunk_stride + i);
                s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[ty][tx][j] = s[j] + r[j];
        __syncthreads();
        if (ty < 4 && i < n) {  // wave ty finishes output j = ty of every thread's quad
            const int j = ty;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[k][tx][j];
            const int64_t ii = i + j;
            const int co = (int)(ii % Cout);
            const int64_t rr = ii / Cout;
            const int ci = (int)(rr % Cin), tap = (int)(rr / Cin);
            out[((int64_t)co * Cin + ci) * ntaps + tap] = (float)a;
        }
        __syncthreads();
    }
}

// Large weight tensors (>= 1M entries): the [tap][ci][co] -> [co][ci][tap] transposition goes through LDS so that both
// sides are coalesced (128-byte reads of 32 consecutive co, 108-byte writes of 27 consecutive taps); the element-wise
// version scattered 4-byte writes 55 KB apart (94 us for the 512 -> 512 layer).  Block <-> (ci, 32-co group).
__global__ __launch_bounds__(256) void k_reduce_slabs_sd_t(const float* __restrict__ part, int nchunks, int Cin, int Cout,
                                                           int64_t chunk_stride, float* __restrict__ out, int nb_w,
                                                           const float* __restrict__ dbpart, int db_chunks,
                                                           float* __restrict__ db, int64_t db_stride) {
    __shared__ float tile[27][33];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nb_w) {  // bias gradient (see k_reduce_slabs_sd)
        const int co = ((int)blockIdx.x - nb_w) * 256 + tid;
        if (co < Cout) {
            double a = 0.0;
            for (int c0 = 0; c0 < db_chunks; c0 += 8) {   // eight rows per trip, unconditional loads (clamped row, dropped below)
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = dbpart[(int64_t)(c0 + k < db_chunks ? c0 + k : db_chunks - 1) * db_stride + co];
#pragma unroll
                for (int k = 0; k < 8; ++k) a += c0 + k < db_chunks ? (double)v[k] : 0.0;
            }
            db[co] = (float)a;
        }
        return;
    }
    const int cg = Cout >> 5;
    const int ci = (int)blockIdx.x / cg, co0 = ((int)blockIdx.x % cg) * 32;
    for (int e = tid; e < 27 * 32; e += 256) {
        const int tap = e >> 5, col = e & 31;
        const int64_t i = ((int64_t)tap * Cin + ci) * Cout + co0 + col;
        double a = 0.0;
        for (int c = 0; c < nchunks; ++c) a += (double)part[(int64_t)c * chunk_stride + i];
        tile[tap][col] = (float)a;
    }
    __syncthreads();
    for (int e = tid; e < 27 * 32; e += 256) {
        const int col = e / 27, tap = e % 27;
        out[((int64_t)(co0 + col) * Cin + ci) * 27 + tap] = tile[tap][col];
    }
}

void tem_reduce_slabs_w_db(const float* part, int nchunks, int ntaps, int Cin, int Cout, int64_t chunk_stride, float* dw,
                           int sd_layout, const float* dbpart, int db_chunks, float* db, hipStream_t s, int64_t db_stride) {
    // db_stride: floats between the bias-gradient rows (0: dense rows of Cout; kernels that keep db inside their slabs pass
    // the slab size)
    if (!db_stride) db_stride = Cout;
    const int64_t n = (int64_t)ntaps * Cin * Cout;
    if (!sd_layout) {
        tem_reduce_slabs(part, nchunks, n, chunk_stride, dw, s);
        if (db) tem_reduce_slabs(dbpart, db_chunks, Cout, db_stride, db, s);
        return;
    }
    int64_t nb = tem_cdiv(n, 64);
    if (nb > 4096) nb = 4096;
    const int64_t nbd = db ? tem_cdiv((int64_t)Cout, 64) : 0;
    if (ntaps == 27 && Cout % 32 == 0 && n >= (1 << 20) && nchunks <= 64) {
        const int64_t nbt = (int64_t)Cin * (Cout / 32), nbdt = db ? tem_cdiv((int64_t)Cout, 256) : 0;
        hipLaunchKernelGGL(k_reduce_slabs_sd_t, dim3((unsigned)(nbt + nbdt)), dim3(256), 0, s, part, nchunks, Cin, Cout,
                           chunk_stride, dw, (int)nbt, dbpart, db_chunks, db, db_stride);
        return;
    }
    if (n % 4 == 0 && chunk_stride % 4 == 0 && ((uintptr_t)part % 16 == 0)) {
        int64_t nb4 = tem_cdiv(n, 256);
        if (nb4 > 4096) nb4 = 4096;
        hipLaunchKernelGGL(k_reduce_slabs_sd4, dim3((unsigned)(nb4 + nbd)), dim3(512), 0, s, part, nchunks, ntaps, Cin, Cout,
                           chunk_stride, dw, (int)nb4, dbpart, db_chunks, db, db_stride);
        return;
    }
    hipLaunchKernelGGL(k_reduce_slabs_sd, dim3((unsigned)(nb + nbd)), dim3(512), 0, s, part, nchunks, ntaps, Cin, Cout,
                       chunk_stride, dw, (int)nb, dbpart, db_chunks, db, db_stride);
}

void tem_reduce_slabs_w(const float* part, int nchunks, int ntaps, int Cin, int Cout, int64_t chunk_stride, float* dw,
                        int sd_layout, hipStream_t s) {
    tem_reduce_slabs_w_db(part, nchunks, ntaps, Cin, Cout, chunk_stride, dw, sd_layout, nullptr, 0, nullptr, s, 0);
}

#define CIN1_GRID 1024

int64_t tem_conv_wgrad_cin1_ws(int Cout, int ntaps) { return (int64_t)CIN1_GRID * (ntaps + 1) * Cout * 4; }

bool tem_conv_wgrad_cin1(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                         int64_t g_ld, float* dw, float* db, void* ws, int N, int D, int H, int W, int Cin, int Cout,
                         int kd, int kh, int kw, int sd_layout, const float* gnx, int64_t gnx_ld, const float* gcoef,
                         hipStream_t s) {
    const int cq = Cout / 4;
    if (gcoef && (gnx_ld % 4 || ((uintptr_t)gnx % 16) || ((uintptr_t)gcoef % 16))) return false;
    // Cin 2..4: one pass per input channel (g is re-read Cin times: still ~6x faster than the generic kernel)
    if (Cin > 4 || Cout % 4 || cq > 16 || (cq & (cq - 1)) || g_ld % 4 || ((uintptr_t)g % 16)) return false;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    if (key != 7 && key != 3) return false;
    const int nZ = (D + 3) / 4, nY = (H + 7) / 8, nX = (W + 7) / 8;
    const int64_t P64 = (int64_t)N * nZ * nY * nX;
    const int P = (int)P64;
    const int grid = P < CIN1_GRID ? P : CIN1_GRID;
    const int NT = kd * kh * kw;
    float* part = (float*)ws;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* sc = scale ? scale + ci : nullptr;
        const float* sf = scale ? shift + ci : nullptr;
        TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TG, return false, {
            if (key == 7) {
                size_t ldsf = 6 * 10 * 12 > 4 * (NT + 1) * Cout ? 6 * 10 * 12 : 4 * (NT + 1) * Cout;
                hipLaunchKernelGGL((k_conv_wgrad_cin1<3, 3, 3, TX, TG>), dim3(grid), dim3(256), ldsf * sizeof(float), s, (const TX*)x + ci,
                                   x_ld, sc, sf, (const TG*)g, g_ld, part, N, D, H, W, Cout, P, nZ, nY, nX, Cin, (const TG*)gnx, gnx_ld, gcoef);
            } else {
                size_t ldsf = 4 * 10 * 12 > 4 * (NT + 1) * Cout ? 4 * 10 * 12 : 4 * (NT + 1) * Cout;
                hipLaunchKernelGGL((k_conv_wgrad_cin1<1, 3, 3, TX, TG>), dim3(grid), dim3(256), ldsf * sizeof(float), s, (const TX*)x + ci,
                                   x_ld, sc, sf, (const TG*)g, g_ld, part, N, D, H, W, Cout, P, nZ, nY, nX, Cin, (const TG*)gnx, gnx_ld, gcoef);
            }
        });
        // the first NT*Cout entries of a slab are dw[tap][ci][co]; the last Cout are db
        if (Cin == 1) {   // the usual first layer: weight and bias gradient in one merge launch
            tem_reduce_slabs_w_db(part, grid, NT, 1, Cout, (int64_t)(NT + 1) * Cout, dw, sd_layout, part + (int64_t)NT * Cout, grid, db, s,
                                  (int64_t)(NT + 1) * Cout);
        } else {
            int64_t nb = tem_cdiv((int64_t)NT * Cout, 64);
            hipLaunchKernelGGL(k_reduce_slabs_ci, dim3((unsigned)nb), dim3(512), 0, s, part, grid, NT, Cin, ci, Cout,
                               (int64_t)(NT + 1) * Cout, dw, sd_layout);
        }
        if (db && ci == 0 && Cin != 1) tem_reduce_slabs(part + (int64_t)NT * Cout, grid, Cout, (int64_t)(NT + 1) * Cout, db, s);
    }
    return true;
}

// ---------------------------------------------------------------------------
// 1x1x1 projection to COUT <= 16 channels, Cin % 32 == 0: 8 lanes share a voxel, each reads
// 16 bytes of its channel row (one fully coalesced 128-byte line per voxel per pass), partial
// dot products are combined with 3 xor-shuffles.
// ---------------------------------------------------------------------------
template <int COUT, typename EX, typename EY>
__global__ __launch_bounds__(256) void k_conv1x1_proj(const EX* __restrict__ x, int64_t x_ld,
                                                      const float* __restrict__ w /*[ci][co]*/,
                                                      const float* __restrict__ bias, EY* __restrict__ y,
                                                      int64_t y_ld, int64_t NV, int Cin, int act) {
    const int l8 = threadIdx.x & 7;
    const int64_t vstride = (int64_t)gridDim.x * 32;
    for (int64_t v = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); v < NV; v += vstride) {
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        const EX* xp = x + v * x_ld;
        for (int c0 = l8 * 4; c0 < Cin; c0 += 32) {
            const float4 t = act_ld4(xp + c0);
            const float xv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv[j], w[(c0 + j) * COUT + co], acc[co]);
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            acc[co] += __shfl_xor(acc[co], 1, 64);
            acc[co] += __shfl_xor(acc[co], 2, 64);
            acc[co] += __shfl_xor(acc[co], 4, 64);
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co)
            if ((co & 7) == l8) act_st1(y + v * y_ld + co, act_apply_s(acc[co] + (bias ? bias[co] : 0.f), act));
    }
}

// The same projection with the lane's weights in registers (NJ = Cin / 32 16-byte pieces per lane, NJ * COUT <= 32) and,
// for COUT = 8 / 16, a TRANSPOSING reduction over the 8 lanes of a voxel: after the exchange with lane ^ 4 a lane keeps the
// half of its partial sums selected by its own bit 2, then the quarter selected by bit 1, then one of the remaining pair --
// 7 (15) shuffles instead of 24 (48), and lane l8 ends up holding output channel l8 (and l8 + 8): exactly what it stores.
// The generic kernel above re-reads its 4 * COUT weights from the L1 for every voxel: 0.49 ms for 32 -> 8 at 96 x 192 x 192
// (the SPOCO embedding head, 0.9 TB/s) against 0.14 ms for 32 -> 2 at 2 x 128^3.
template <int COUT, int NJ, typename EX, typename EY>
__global__ __launch_bounds__(256) void k_conv1x1_proj_r(const EX* __restrict__ x, int64_t x_ld,
                                                        const float* __restrict__ w /*[ci][co]*/,
                                                        const float* __restrict__ bias, EY* __restrict__ y,
                                                        int64_t y_ld, int64_t NV, int act) {
    const int l8 = threadIdx.x & 7;
    float wr[NJ][4][COUT];
#pragma unroll
    for (int a = 0; a < NJ; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int co = 0; co < COUT; ++co) wr[a][j][co] = w[(a * 32 + l8 * 4 + j) * COUT + co];
    const int64_t vstride = (int64_t)gridDim.x * 32;
    for (int64_t v = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); v < NV; v += vstride) {
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        const EX* xp = x + v * x_ld + l8 * 4;
#pragma unroll
        for (int a = 0; a < NJ; ++a) {
            const float4 t = act_ld4(xp + a * 32);
            const float xv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv[j], wr[a][j][co], acc[co]);
        }
        if constexpr (COUT == 8 || COUT == 16) {
            constexpr int R = COUT / 8;   // outputs a lane ends up with: channels l8 + 8 r
            float h4[4 * R], h2[2 * R], h1[R];
            const bool b4 = (l8 & 4) != 0, b2 = (l8 & 2) != 0, b1 = (l8 & 1) != 0;
            // channel co = 8 r + c, c = 0..7: bit 2 of c picks the half, bit 1 the quarter, bit 0 the element
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float keep = b4 ? acc[8 * r + 4 + i] : acc[8 * r + i], send = b4 ? acc[8 * r + i] : acc[8 * r + 4 + i];
                    h4[4 * r + i] = keep + __shfl_xor(send, 4, 64);
                }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float keep = b2 ? h4[4 * r + 2 + i] : h4[4 * r + i], send = b2 ? h4[4 * r + i] : h4[4 * r + 2 + i];
                    h2[2 * r + i] = keep + __shfl_xor(send, 2, 64);
                }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float keep = b1 ? h2[2 * r + 1] : h2[2 * r], send = b1 ? h2[2 * r] : h2[2 * r + 1];
                h1[r] = keep + __shfl_xor(send, 1, 64);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int co = 8 * r + l8;
                act_st1(y + v * y_ld + co, act_apply_s(h1[r] + (bias ? bias[co] : 0.f), act));
            }
        } else {
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                acc[co] += __shfl_xor(acc[co], 1, 64);
                acc[co] += __shfl_xor(acc[co], 2, 64);
                acc[co] += __shfl_xor(acc[co], 4, 64);
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co)
                if ((co & 7) == l8) act_st1(y + v * y_ld + co, act_apply_s(acc[co] + (bias ? bias[co] : 0.f), act));
        }
    }
}

bool tem_conv1x1_proj(const float* x, int64_t x_ld, const float* scale, const float* w, const float* bias, float* y,
                      int64_t y_ld, const float* ref, int64_t NV, int Cin, int Cout, int act, hipStream_t s) {
    if (scale || ref || Cin % 32 || x_ld % 4 || ((uintptr_t)x % 16)) return false;
    dim3 grid(tem_grid_1d(NV, 32, 256 * 16));
    // weights in registers where they fit (NJ * COUT <= 32 values per 16-byte piece and lane)
#define PJR(CO, NJ_)                                                                                                 \
    if (Cout == CO && Cin == 32 * NJ_) {                                                                             \
        TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, return false,                                           \
                       hipLaunchKernelGGL((k_conv1x1_proj_r<CO, NJ_, TX, TY>), grid, dim3(256), 0, s, (const TX*)x, x_ld, w, bias, \
                                          (TY*)y, y_ld, NV, act));                                                   \
        return true;                                                                                                 \
    }
    PJR(1, 1) PJR(2, 1) PJR(3, 1) PJR(4, 1) PJR(6, 1) PJR(8, 1) PJR(12, 1) PJR(16, 1)
    PJR(1, 2) PJR(2, 2) PJR(3, 2) PJR(4, 2) PJR(6, 2) PJR(8, 2) PJR(12, 2) PJR(16, 2)
    PJR(1, 4) PJR(2, 4) PJR(3, 4) PJR(4, 4) PJR(6, 4) PJR(8, 4)
#undef PJR
#define PJ(CO)                                                                                                     \
    case CO:                                                                                                       \
        TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, return false,                                         \
                       hipLaunchKernelGGL((k_conv1x1_proj<CO, TX, TY>), grid, dim3(256), 0, s, (const TX*)x, x_ld, w, bias, (TY*)y, \
                                          y_ld, NV, Cin, act));                                                    \
        return true;
    switch (Cout) {
        PJ(1) PJ(2) PJ(3) PJ(4) PJ(6) PJ(8) PJ(12) PJ(16)
        default: return false;
    }
#undef PJ
}

// weight gradient of the projection: dw[ci][co] = sum_v x[v][ci] * g[v][co], db[co] = sum_v g[v][co]
// EXP (tem_conv1x1_out_bwd, round 4): the same pass also writes the DATA gradient of the projection,
//   gx[v][ci] = x[v][ci] > 0 ? sum_co g[v][co] * w[co][ci] : 0   (x is the ReLU output the projection read),
// which used to be k_conv1x1_expand's own pass over x (a 512 MB tensor at the out_conv of a 128^3 net): the thread that
// holds 4 channels of x for the weight gradient holds what the mask and the 16-byte store need.
template <int COUT, int NJ, bool EXP, typename EX, typename EG>
__global__ __launch_bounds__(256) void k_conv1x1_proj_wgrad(const EX* __restrict__ x, int64_t x_ld,
                                                            const EG* __restrict__ g, int64_t g_ld,
                                                            float* __restrict__ part /*[grid][Cin+1][COUT]*/,
                                                            int64_t NV, int Cin, const float* __restrict__ w = nullptr,
                                                            EX* __restrict__ gx = nullptr, int64_t gx_ld = 0,
                                                            unsigned* __restrict__ amax = nullptr) {
    extern __shared__ float lds[];  // [4][Cin+1][COUT]
    const int l8 = threadIdx.x & 7;
    const int64_t vstride = (int64_t)gridDim.x * 32;
    constexpr int nj = NJ;          // = Cin / 32: 16-byte pieces per lane
    float acc[NJ][4][COUT];
    float gacc[COUT];
#pragma unroll
    for (int a = 0; a < NJ; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[a][j][co] = 0.f;
#pragma unroll
    for (int co = 0; co < COUT; ++co) gacc[co] = 0.f;
    float wq[EXP ? NJ : 1][EXP ? COUT : 1][4];   // EXP: w[co][a * 32 + l8 * 4 .. + 3]
    float amx = 0.f;
    if (EXP) {
#pragma unroll
        for (int a = 0; a < NJ; ++a)
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float4 t = *reinterpret_cast<const float4*>(w + (int64_t)co * Cin + a * 32 + l8 * 4);
                wq[a][co][0] = t.x;
                wq[a][co][1] = t.y;
                wq[a][co][2] = t.z;
                wq[a][co][3] = t.w;
            }
    }
    for (int64_t v = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); v < NV; v += vstride) {
        float gv[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            gv[co] = act_ld1(g + v * g_ld + co);
            gacc[co] += gv[co];
        }
        const EX* xp = x + v * x_ld;
#pragma unroll
        for (int a = 0; a < NJ; ++a) {
            if (a < nj) {
                const float4 t = act_ld4(xp + a * 32 + l8 * 4);
                const float xv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[a][j][co] = fmaf(xv[j], gv[co], acc[a][j][co]);
                if (EXP) {
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t2 = 0.f;
#pragma unroll
                        for (int co = 0; co < COUT; ++co) t2 = fmaf(gv[co], wq[a][co][j], t2);   // same order as k_conv1x1_expand
                        o[j] = xv[j] > 0.f ? t2 : 0.f;
                    }
                    ST4(gx + v * gx_ld + a * 32 + l8 * 4, make_float4(o[0], o[1], o[2], o[3]));
                    amx = tem_amax4(amx, o[0], o[1], o[2], o[3]);
                }
            }
        }
    }
    if (EXP && amax) tem_amax_commit(amax, amx);
    // lanes with equal l8 inside a wave: xor 8, 16, 32
#pragma unroll
    for (int a = 0; a < NJ; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                float t = acc[a][j][co];
                t += __shfl_xor(t, 8, 64);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                acc[a][j][co] = t;
            }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float t = gacc[co];
        t += __shfl_xor(t, 8, 64);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        gacc[co] = t;
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slab = (Cin + 1) * COUT;
    if (lane < 8) {
#pragma unroll
        for (int a = 0; a < NJ; ++a)
            if (a < nj)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
                        lds[wv * slab + (a * 32 + lane * 4 + j) * COUT + co] = acc[a][j][co];
        if (lane == 0)
#pragma unroll
            for (int co = 0; co < COUT; ++co) lds[wv * slab + Cin * COUT + co] = gacc[co];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < slab; i += 256)
        part[(int64_t)blockIdx.x * slab + i] = lds[i] + lds[slab + i] + lds[2 * slab + i] + lds[3 * slab + i];
}

#define PROJ_GRID 1024
int64_t tem_conv1x1_proj_wgrad_ws(int Cin, int Cout) { return (int64_t)PROJ_GRID * (Cin + 1) * Cout * 4; }

bool tem_conv1x1_proj_wgrad(const float* x, int64_t x_ld, const float* scale, const float* g, int64_t g_ld, float* dw,
                            float* db, void* ws, int64_t NV, int Cin, int Cout, int sd_layout, hipStream_t s) {
    if (scale || Cin % 32 || Cin > 128 || x_ld % 4 || ((uintptr_t)x % 16)) return false;
    const int njr = Cin / 32;
    if (njr == 3 || njr * Cout > 32) return false;  // accumulators: NJ * 4 * COUT registers per lane
    int64_t nb = tem_cdiv(NV, 32);
    const int grid = (int)(nb < PROJ_GRID ? nb : PROJ_GRID);
    float* part = (float*)ws;
    size_t ldsb = (size_t)4 * (Cin + 1) * Cout * sizeof(float);
#define PWJ(CO, J)                                                                                                \
    TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TG, return false,                                            \
                   hipLaunchKernelGGL((k_conv1x1_proj_wgrad<CO, J, false, TX, TG>), dim3(grid), dim3(256), ldsb, s, (const TX*)x, x_ld, \
                                      (const TG*)g, g_ld, part, NV, Cin))
#define PW(CO)                                                                                                    \
    case CO:                                                                                                      \
        if (njr == 1) PWJ(CO, 1);                                                                                 \
        else if (njr == 2) PWJ(CO, 2);                                                                            \
        else PWJ(CO, 4);                                                                                          \
        break;
#define PW2(CO)                                                                                                   \
    case CO:                                                                                                      \
        if (njr == 1) PWJ(CO, 1);                                                                                 \
        else PWJ(CO, 2);                                                                                          \
        break;
#define PW1(CO)                                                                                                   \
    case CO: PWJ(CO, 1); break;
    switch (Cout) {
        PW(1) PW(2) PW(3) PW(4) PW(6) PW(8) PW2(12) PW2(16) PW1(24) PW1(32)
        default: return false;
    }
#undef PW
#undef PW2
#undef PW1
#undef PWJ
    const int64_t slab = (int64_t)(Cin + 1) * Cout;
    tem_reduce_slabs_w_db(part, grid, 1, Cin, Cout, slab, dw, sd_layout, part + (int64_t)Cin * Cout, grid, db, s, slab);  // [tap=0][ci][co], db behind it
    return true;
}

// out_conv backward in ONE pass over x: weight / bias gradient of the projection AND its masked data gradient
// (w: state_dict layout [Cout][Cin]).  false: shape not covered (the caller runs the two separate kernels).
bool tem_conv1x1_out_bwd(const float* x, int64_t x_ld, const float* g, int64_t g_ld, const float* w, float* gx, int64_t gx_ld,
                         float* dw, float* db, void* ws, int64_t NV, int Cin, int Cout, int sd_layout, hipStream_t s) {
    if (Cin % 32 || Cin > 64 || x_ld % 4 || ((uintptr_t)x % 16) || gx_ld % 4 || ((uintptr_t)gx % 16) || ((uintptr_t)w % 16)) return false;
    const int njr = Cin / 32;
    if (Cout > 4 || njr * Cout > 8) return false;   // wq registers: NJ * COUT * 4 per lane beside the accumulators
    int64_t nb = tem_cdiv(NV, 32);
    const int grid = (int)(nb < PROJ_GRID ? nb : PROJ_GRID);
    float* part = (float*)ws;
    size_t ldsb = (size_t)4 * (Cin + 1) * Cout * sizeof(float);
    if (!tem_st2_ok(tem_call_st.x, tem_call_st.y)) return false;
    unsigned* const amax = tem_take_output_amax();
#define OBJ(CO, J)                                                                                                     \
    TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TG, return false,                                                 \
                   hipLaunchKernelGGL((k_conv1x1_proj_wgrad<CO, J, true, TX, TG>), dim3(grid), dim3(256), ldsb, s, (const TX*)x, x_ld, \
                                      (const TG*)g, g_ld, part, NV, Cin, w, (TX*)gx, gx_ld, amax))
#define OB(CO)                \
    case CO:                  \
        if (njr == 1) OBJ(CO, 1); \
        else OBJ(CO, 2);      \
        break;
    switch (Cout) {
        OB(1) OB(2) OB(3) OB(4)
        default: return false;
    }
#undef OB
#undef OBJ
    const int64_t slab = (int64_t)(Cin + 1) * Cout;
    tem_reduce_slabs_w_db(part, grid, 1, Cin, Cout, slab, dw, sd_layout, part + (int64_t)Cin * Cout, grid, db, s, slab);
    return true;
}

// ---------------------------------------------------------------------------
// 1x1x1 expansion from a few channels (data gradient of out_conv: Cout_net <= 16 -> 32k channels):
// thread <-> (voxel, 4 output channels); the 512 MB result is written once with 16-byte stores,
// ReLU-backward mask fused.
// ---------------------------------------------------------------------------
template <typename EX, typename EY>
__global__ __launch_bounds__(256) void k_conv1x1_expand(const EX* __restrict__ x, int64_t x_ld,
                                                        const float* __restrict__ w /*[ci][co]*/,
                                                        const float* __restrict__ bias, EY* __restrict__ y,
                                                        int64_t y_ld, const EY* __restrict__ ref, int64_t ref_ld,
                                                        int64_t NV, int Cin, int Cout, int act,
                                                        unsigned* __restrict__ amax) {
    const int cq = Cout >> 2;
    const int64_t items = NV * cq, stride = (int64_t)gridDim.x * 256;
    float amx = 0.f;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t v = i / cq;
    int q = (int)(i % cq);
    const int64_t dv = stride / cq;
    const int dq = (int)(stride % cq);
    for (; i < items; i += stride, v += dv, q += dq) {
        if (q >= cq) {
            q -= cq;
            ++v;
        }
        float4 a = bias ? *reinterpret_cast<const float4*>(bias + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = act_ld1(x + v * x_ld + ci);
            const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)ci * Cout + q * 4);
            a.x = fmaf(xv, wv.x, a.x);
            a.y = fmaf(xv, wv.y, a.y);
            a.z = fmaf(xv, wv.z, a.z);
            a.w = fmaf(xv, wv.w, a.w);
        }
        a.x = act_apply_s(a.x, act);
        a.y = act_apply_s(a.y, act);
        a.z = act_apply_s(a.z, act);
        a.w = act_apply_s(a.w, act);
        if (ref) {
            const float4 rr = act_ld4(ref + v * ref_ld + q * 4);
            if (!(rr.x > 0.f)) a.x = 0.f;
            if (!(rr.y > 0.f)) a.y = 0.f;
            if (!(rr.z > 0.f)) a.z = 0.f;
            if (!(rr.w > 0.f)) a.w = 0.f;
        }
        ST4(y + v * y_ld + q * 4, a);
        amx = tem_amax4(amx, a.x, a.y, a.z, a.w);
    }
    if (amax) tem_amax_commit(amax, amx);
}

bool tem_conv1x1_expand(const float* x, int64_t x_ld, const float* scale, const float* w, const float* bias, float* y,
                        int64_t y_ld, const float* ref, int64_t ref_ld, int64_t NV, int Cin, int Cout, int act,
                        hipStream_t s) {
    if (scale || Cin > 16 || Cout % 4 || y_ld % 4 || ((uintptr_t)y % 16) || ((uintptr_t)w % 16) ||
        (bias && (uintptr_t)bias % 16) || (ref && (ref_ld % 4 || (uintptr_t)ref % 16)) || !tem_st2_ok(tem_call_st.x, tem_call_st.y))
        return false;
    unsigned* const amax = tem_take_output_amax();
    TEM_ST2_SWITCH(tem_call_st.x, tem_call_st.y, TX, TY, return false,
                   hipLaunchKernelGGL((k_conv1x1_expand<TX, TY>), dim3(tem_grid_1d(NV * (Cout / 4), 256, 256 * 16)), dim3(256), 0, s,
                                      (const TX*)x, x_ld, w, bias, (TY*)y, y_ld, (const TY*)ref, ref_ld, NV, Cin, Cout, act, amax));
    return true;
}
