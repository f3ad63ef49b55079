// conv_wino.hip -- 3x3x3 convolution as F(2x2, 3x3) Winograd in the (y, x) plane with the z taps kept direct:
// 16 frequency products x 3 z taps per 2x2 output tile = 12 multiplies per output instead of 27, i.e. 2.25x fewer
// split-precision MFMAs than the direct kernels (conv_pp.hip) for the same result to fp32-class rounding.
// Same operation as conv_pp.hip (torch_em/model/unet.py:429-438 ConvBlock: Conv3d(3, padding=1) [+ the fused pre-norm
// affine, bias, ReLU / ReLU-mask of the neighbouring layers]); forward and data gradient (flipped, transposed pack).
//
// Work decomposition (one workgroup per CU, 4 waves, one per SIMD, up to 512 registers each):
//   * a workgroup owns an 8 x 16 (y, x) output region = 4 x 8 Winograd tiles = the 32 rows of one 32x32x16 MFMA, and
//     walks a z-column (or a z-segment of it) plane by plane -- no halo re-reads along z, 1.41x in the plane;
//   * wave w owns frequency ROW fy = w (4 of the 16 frequencies).  B^T has two non-zeros per row, so the wave reads just
//     two halo rows per tile, forms t = d[ra] +- d[rb] and the four x-frequencies, splits them (hi, lo) and has its MFMA
//     A operands in registers: the transformed activations never touch LDS or HBM;
//   * input plane zi feeds output planes zi-1, zi, zi+1 through the three z taps; the three live output planes x 4
//     frequencies are 12 accumulators (192 registers; twice that with the second set of the fp16x3 cross terms or with two
//     Cout tiles) which rotate through three code phases (plane loop unrolled by 3, all register indices static);
//   * when an output plane is complete every wave applies A^T along x in registers (4 -> 2), the four frequency rows meet
//     in LDS and are combined along y (A^T again), + bias, activation / ReLU mask, 16-byte stores.
// Frequency row 2 is carried NEGATED on both sides (t = d[1] - d[2] instead of d[2] - d[1]; the pack negates U) so that
// every wave computes t = d[ra] + sb * d[rb] with one scalar sign.
// LDS: 2 halo planes of one 32-channel chunk [row 10][voxel 18][32 ch fp32], 16-byte slots XOR-swizzled by the tile
// column so that the 8 tile columns of a row hit 8 different bank groups, rows padded by 64 B so that tile rows alternate
// between the two halves of the banks; 2 exchange buffers for the output transform.
#include <type_traits>
#include "../../torch_em_amd/csrc/tem_common.h"
#include "../../torch_em_amd/csrc/conv_split.h"
#include "../../torch_em_amd/csrc/conv_internal.h"

#define WN_RS 2368                 // bytes per halo row: 18 voxels x 128 B + 64
#define WN_PLB (10 * WN_RS)        // bytes per staged plane chunk
#define WN_PFL 8192                // floats per exchange buffer and Cout tile: [fy 4][tile 32][xo 2][co 32]
#define WN_NIT 6                   // staging items per thread and chunk: 10 x 18 voxels x 8 slots = 1440 <= 6 x 256

typedef __amdgpu_buffer_rsrc_t wn_rsrc_t;
typedef float wn_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int wn_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ wn_rsrc_t wn_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ wn_f4 wn_load4(wn_rsrc_t r, unsigned voff, unsigned soff) {
    const wn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(wn_f4, v);   // whole-vector cast (element-wise casts get the load narrowed, see conv_pp.hip)
}
__device__ __forceinline__ uint4 wn_load4u(wn_rsrc_t r, unsigned voff, unsigned soff) {
    const wn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// (hi, lo) words of two transformed activations
template <bool F16>
__device__ __forceinline__ void wn_split(float a, float b, unsigned& hi, unsigned& lo) {
    if constexpr (F16) {
        hi = pk16<true>(a, b);
        lo = pk16<true>((a - lo16<true>(hi)) * F16_LO_SCALE, (b - hi16<true>(hi)) * F16_LO_SCALE);
    } else {
        split2(a, b, hi, lo);
    }
}

// `TEM_WN_R` groups of weight fragments are in flight ahead of the MFMAs that use them (ring of TEM_WN_R + 1 slots).
#ifndef WN_R
#define WN_R 2
#endif
#ifndef WN_FIRST
#define WN_FIRST 0
#endif
#ifndef WN_ABL
#define WN_ABL 0   // harness ablations: 1 no staging in the loop, 2 no output combine / stores, 4 no weight loads in the loop
#endif
#ifndef WN_PIN
#define WN_PIN 1
#endif
#ifdef WN_TRACE   // developer build (scripts/wino_harness.cpp): shader-clock stamps of the steps of workgroup 0
__device__ unsigned long long tem_wn_trace_buf[4][64][8];
#define WN_STAMP(i)                                                                          \
    do {                                                                                     \
        if (blockIdx.x == 0 && lane == 0 && step < 64)                                       \
            tem_wn_trace_buf[fy][step][i] = __builtin_amdgcn_s_memtime();                    \
    } while (0)
void tem_wn_trace_read(unsigned long long* dst) {
    (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tem_wn_trace_buf), sizeof(unsigned long long) * 4 * 64 * 8);
}
#else
#define WN_STAMP(i)
#endif
#define WN_NBUF 3                  // staged plane chunks in LDS: steps s, s+1 (being read), s+2 (being written)

typedef float wn_f2 __attribute__((ext_vector_type(2)));
// (hi, lo) bf16 words of two values in 5 VALU ops: the conversion is opaque (asm) so that the hi floats come from one shift
// and one mask of the packed word instead of a second conversion
__device__ __forceinline__ void wn_split2(float a, float b, unsigned& hi, unsigned& lo) {
    unsigned h;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    hi = h;
    lo = pk_bf16(ra, rb);
}
__device__ __forceinline__ void wn_store4(wn_rsrc_t r, unsigned voff, unsigned soff, wn_f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wn_u4, v), r, voff, soff, 0);
}

template <bool ONE>   // ONE: a single 32-channel chunk (Cin == 32)
__global__ __launch_bounds__(256, 1) void k_conv_wino(const float* __restrict__ x, int64_t x_ld,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const unsigned short* __restrict__ wp,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int64_t y_ld, const float* __restrict__ ref, int64_t ref_ld, int N,
                                                      int D, int H, int W, int Cin, int Cout, int act, int nY, int nX,
                                                      int zsegs, int nct) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* raw = lds;                                              // [WN_NBUF][WN_PLB]
    float* P = reinterpret_cast<float*>(lds + WN_NBUF * WN_PLB);           // [2][WN_PFL]

    const int tid = threadIdx.x, lane = tid & 63;
    const int fy = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, r = lane & 31, ty = r >> 3, tx = r & 7;
    int q = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int ct = q % nct;   // Cout tile: workgroups that share an input region are neighbours (same XCD, same L2)
    q /= nct;
    const int ptx = q % nX;
    q /= nX;
    const int pty = q % nY;
    q /= nY;
    const int zseg = q % zsegs;
    const int n = q / zsegs;
    const int y0 = pty * 8, x0 = ptx * 16;
    const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
    const int zlo = za > 0 ? za - 1 : 0;
    const int zhi = zb < D ? zb : D - 1;          // last real input plane
    const int nch = ONE ? 1 : (Cin >> 5);
    const int nsteps = (zhi - zlo + 1) * nch;     // step s = (plane zlo + s / nch, chunk s % nch)

    // ---- staging items: (voxel of the 10 x 18 halo, 16-byte slot); the slot is the same for all items of a thread.
    // No branches: an item outside the volume reads past the buffer range (returns 0) and gets no shift, an item past
    // the 1440 of a chunk is written into the row padding.  Item k of step s+3 is LOADED in group k of the first half of
    // step s and STORED (pre-norm applied) in group k of step s+1: a whole step for the HBM latency, and the loads of the
    // weight stream issued in between never have to wait for a younger HBM load (vmcnt is in-order).
    const int sslot = tid & 7;
    unsigned goff[WN_NIT], loff[WN_NIT];
    unsigned okmask = 0;
#pragma unroll
    for (int k = 0; k < WN_NIT; ++k) {
        const int i = tid + 256 * k;
        const int v = i >> 3, row = v / 18, vx = v % 18;
        const int gy = y0 - 1 + row, gx = x0 - 1 + vx;
        const bool ok = i < 1440 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        okmask |= (ok ? 1u : 0u) << k;
        goff[k] = ok ? (unsigned)((gy * W + gx) * (int)x_ld + sslot * 4) * 4u : 0x80000000u;
        loff[k] = i < 1440 ? (unsigned)(row * WN_RS + vx * 128 + ((sslot ^ ((vx >> 1) & 7)) << 4))
                           : (unsigned)(18 * 128 + (tid & 3) * 16);
    }
    const wn_rsrc_t rsx = wn_rsrc(x + (int64_t)n * D * H * W * x_ld);
    const unsigned xplane = (unsigned)(H * W) * (unsigned)x_ld * 4u;
    wn_f4 sd[WN_NIT];
    wn_f4 sc4 = {1.f, 1.f, 1.f, 1.f}, sf4 = {0.f, 0.f, 0.f, 0.f};

    auto stage_so = [&](int s) {
        s = s < nsteps ? s : nsteps - 1;   // past the end: the last chunk again (into a free buffer, never read)
        const int zi = zlo + s / nch, c = s % nch;
        return (unsigned)zi * xplane + (unsigned)c * 128u;
    };
    auto stage_norm = [&](int s) {   // pre-norm coefficients of the chunk of step s
        if (scale) {
            s = s < nsteps ? s : nsteps - 1;
            const int c = s % nch;
            sc4 = *reinterpret_cast<const wn_f4*>(scale + (int64_t)n * Cin + c * 32 + sslot * 4);
            sf4 = *reinterpret_cast<const wn_f4*>(shift + (int64_t)n * Cin + c * 32 + sslot * 4);
        }
    };
    auto stage_store_item = [&](unsigned char* dst, const int k) {
        const bool ok = (okmask >> k) & 1u;
        const wn_f4 f = {ok ? sf4.x : 0.f, ok ? sf4.y : 0.f, ok ? sf4.z : 0.f, ok ? sf4.w : 0.f};
        const wn_f4 o = {fmaf(sd[k].x, sc4.x, f.x), fmaf(sd[k].y, sc4.y, f.y), fmaf(sd[k].z, sc4.z, f.z), fmaf(sd[k].w, sc4.w, f.w)};
        *reinterpret_cast<wn_f4*>(dst + loff[k]) = o;   // padding stays zero AFTER the pre-norm
    };

    // ---- A-side read addresses: two halo rows of this wave's frequency row, 4 voxels of the tile, 8 channels ----
    const int ra = fy == 0 ? 0 : 1;
    const int rdelta = (fy == 0 || fy == 3) ? 2 * WN_RS : WN_RS;    // rb - ra rows
    const float sb = (fy == 1) ? 1.f : -1.f;                        // t = d[ra] + sb * d[rb]   (row 2 negated, see above)
    unsigned aoff[2][2][2];                                         // [x >> 1][kstep][16-byte half]
#pragma unroll
    for (int sv = 0; sv < 2; ++sv)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                aoff[sv][ks][j] = (unsigned)((2 * ty + ra) * WN_RS + 2 * tx * 128 + (((ks * 4 + kh * 2 + j) ^ ((tx + sv) & 7)) << 4));

    // ---- weights: this wave's fragment stream [fy][ct][chunk][kstep][dz][fx][hi|lo][64 lanes x 16 B] ----
    const wn_rsrc_t rsw = wn_rsrc(wp);
    const unsigned wvoff = (unsigned)lane * 16u;
    const unsigned wstream = (unsigned)((fy * nct + ct) * nch) * (2u * 24u * 1024u);
    auto wsoff = [&](int c, int ks) { return wstream + (unsigned)(c * 2 + ks) * (24u * 1024u); };
    const unsigned wzero = (unsigned)(4 * nct * nch) * (2u * 24u * 1024u);   // 4 KB of zeros behind the stream: taps of planes
                                                                           // outside this workgroup's z range multiply by 0
    // group g of a half-step: z tap 2 - g / 2 (the oldest output plane first), frequencies 2 (g & 1), 2 (g & 1) + 1
    auto goffs = [](int g) { return (unsigned)(((2 - g / 2) * 4 + (g & 1) * 2) * 2048); };

    floatx16 acc[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[s][f][k] = 0.f;
    uint4 ah[2][4], al[2][4];          // A operands of the current and the next half-step
    wn_f4 da[4], db[4];                // raw rows of the half-build in flight
    uint4 bfr[WN_R + 1][4];            // weight fragment ring: per group (f0 hi, f0 lo, f1 hi, f1 lo)

    auto a_reads = [&](const unsigned char* bufp, auto ks_tag, auto j_tag) {
        constexpr int KS = decltype(ks_tag)::value, J = decltype(j_tag)::value;
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
            const unsigned char* p = bufp + aoff[xx >> 1][KS][J] + xx * 128;
            da[xx] = *reinterpret_cast<const wn_f4*>(p);
            db[xx] = *reinterpret_cast<const wn_f4*>(p + rdelta);
        }
    };
    wn_f4 vq[4];   // the four x-frequencies (4 channels) of the half-build in flight
    auto a_tv = [&]() {   // scalar f32 ops on purpose: packed f32 VALU beside MFMAs costs more than the two ops it replaces
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float t0 = fmaf(db[0][cc], sb, da[0][cc]), t1 = fmaf(db[1][cc], sb, da[1][cc]);
            const float t2 = fmaf(db[2][cc], sb, da[2][cc]), t3 = fmaf(db[3][cc], sb, da[3][cc]);
            vq[0][cc] = t0 - t2;
            vq[1][cc] = t1 + t2;
            vq[2][cc] = t2 - t1;
            vq[3][cc] = t1 - t3;
        }
    };
    auto a_split = [&](auto set_tag, auto j_tag, auto f_tag) {
        constexpr int SET = decltype(set_tag)::value, J = decltype(j_tag)::value, f = decltype(f_tag)::value;
        unsigned h0, l0, h1, l1;
        wn_split2(vq[f].x, vq[f].y, h0, l0);
        wn_split2(vq[f].z, vq[f].w, h1, l1);
        if constexpr (J == 0) {
            ah[SET][f].x = h0;
            ah[SET][f].y = h1;
            al[SET][f].x = l0;
            al[SET][f].y = l1;
        } else {
            ah[SET][f].z = h0;
            ah[SET][f].w = h1;
            al[SET][f].z = l0;
            al[SET][f].w = l1;
        }
    };
    auto a_compute = [&](auto set_tag, auto j_tag) {
        a_tv();
        a_split(set_tag, j_tag, std::integral_constant<int, 0>{});
        a_split(set_tag, j_tag, std::integral_constant<int, 1>{});
        a_split(set_tag, j_tag, std::integral_constant<int, 2>{});
        a_split(set_tag, j_tag, std::integral_constant<int, 3>{});
    };

    // ---- output side: per-thread constants of its two combine items (tile, x output) x 2 y outputs ----
    const int ocq = tid & 7;   // channel quad
    wn_f4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && ct * 32 + ocq * 4 < Cout) bias4 = *reinterpret_cast<const wn_f4*>(bias + ct * 32 + ocq * 4);
    const float act_floor = act == TEM_ACT_RELU ? 0.f : -__builtin_inff();
    const wn_rsrc_t rsy = wn_rsrc(y + (int64_t)n * D * H * W * y_ld);
    const bool has_ref = ref != nullptr;
    const wn_rsrc_t rsr = wn_rsrc(has_ref ? ref + (int64_t)n * D * H * W * ref_ld : y);
    const unsigned yplane = (unsigned)(H * W) * (unsigned)y_ld * 4u, rplane = (unsigned)(H * W) * (unsigned)ref_ld * 4u;
    unsigned yoff[2][2], roff[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int id = tid + 256 * m;
        const int xo = (id >> 3) & 1, tile = (id >> 4) & 31;
        const int ox = x0 + 2 * (tile & 7) + xo, ch = ct * 32 + ocq * 4;
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const int oy = y0 + 2 * (tile >> 3) + yy;
            const bool ok = oy < H && ox < W && ch < Cout;
            yoff[m][yy] = ok ? (unsigned)((oy * W + ox) * (int)y_ld + ch) * 4u : 0x80000000u;
            roff[m][yy] = ok ? (unsigned)((oy * W + ox) * (int)ref_ld + ch) * 4u : 0x80000000u;
        }
    }
    int pb = 0;

    auto epi_combine = [&](const int zo) {
        // ---- A^T along y over the four frequency rows, bias, activation / ReLU mask, 16-byte stores ----
        const float* Pr = P + pb * WN_PFL;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int id = tid + 256 * m;
            const int xo = (id >> 3) & 1, tile = (id >> 4) & 31;
            const float* s0 = Pr + ((tile * 2 + xo) * 32) + ocq * 4;
            const wn_f4 p0 = *reinterpret_cast<const wn_f4*>(s0);
            const wn_f4 p1 = *reinterpret_cast<const wn_f4*>(s0 + 2048);
            const wn_f4 p2 = *reinterpret_cast<const wn_f4*>(s0 + 4096);
            const wn_f4 p3 = *reinterpret_cast<const wn_f4*>(s0 + 6144);
            wn_f4 o[2];
            o[0] = p0 + p1 + p2 + bias4;
            o[1] = p1 - p2 - p3 + bias4;
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                wn_f4 v = o[yy];
                v.x = fmaxf(v.x, act_floor);
                v.y = fmaxf(v.y, act_floor);
                v.z = fmaxf(v.z, act_floor);
                v.w = fmaxf(v.w, act_floor);
                if (has_ref) {
                    const wn_f4 rv = wn_load4(rsr, roff[m][yy], (unsigned)zo * rplane);
                    v.x = rv.x > 0.f ? v.x : 0.f;
                    v.y = rv.y > 0.f ? v.y : 0.f;
                    v.z = rv.z > 0.f ? v.z : 0.f;
                    v.w = rv.w > 0.f ? v.w : 0.f;
                }
                wn_store4(rsy, yoff[m][yy], (unsigned)zo * yplane, v);
            }
        }
        pb ^= 1;
    };

    // ---- prologue: chunks of steps 0 and 1 into LDS, the chunk of step 2 into the staging registers, A operands of the
    // first half-step, first weight groups ----
#pragma unroll
    for (int s0 = 0; s0 < 3; ++s0) {
        const unsigned so = stage_so(s0);
#pragma unroll
        for (int k = 0; k < WN_NIT; ++k) sd[k] = wn_load4(rsx, goff[k], so);
        stage_norm(s0);
        if (s0 < 2) {
#pragma unroll
            for (int k = 0; k < WN_NIT; ++k) stage_store_item(raw + s0 * WN_PLB, k);
        }
    }
    __syncthreads();
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        a_reads(raw, I0{}, I0{});
        a_compute(I0{}, I0{});
        a_reads(raw, I0{}, I1{});
        a_compute(I0{}, I1{});
        // tap 2 of the first plane (groups 0, 1) feeds output plane zlo - 1: never one of ours
#pragma unroll
        for (int sl = 0; sl < WN_R; ++sl)
#pragma unroll
            for (int u = 0; u < 4; ++u) bfr[sl][u] = wn_load4u(rsw, wvoff, wzero + (unsigned)u * 1024u);
    }

    int step = 0;
    auto plane = [&](auto ph_tag, const int zi) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int S_PREV = (PH + 2) % 3, S_CUR = PH, S_NEXT = (PH + 1) % 3;
        // tap dz feeds output plane zi + 1 - dz: slots S_NEXT, S_CUR, S_PREV
        // piece q of the x transform of the finished output plane zi-1: accumulator registers 2q, 2q+1
        auto epi_write_piece = [&](const int q4) {
            float* Pw = P + pb * WN_PFL + fy * (32 * 2 * 32);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = q4 * 2 + ii;
                const float a0 = acc[S_PREV][0][i], a1 = acc[S_PREV][1][i], a2 = acc[S_PREV][2][i], a3 = acc[S_PREV][3][i];
                const int tile = (i & 3) + 8 * (i >> 2) + 4 * kh;
                float* d = Pw + (tile * 2) * 32 + r;
                d[0] = a0 + a1 + a2;
                d[32] = a1 - a2 - a3;
                if (!ONE) {   // ONE: the first tap of the next user starts from zero (see FIRST)
                    acc[S_PREV][0][i] = 0.f;
                    acc[S_PREV][1][i] = 0.f;
                    acc[S_PREV][2][i] = 0.f;
                    acc[S_PREV][3][i] = 0.f;
                }
            }
        };
        if (zi >= D) {   // the plane past the volume: only the last output plane is left to finish
#pragma unroll
            for (int q4 = 0; q4 < 8; ++q4) epi_write_piece(q4);
            __syncthreads();
            epi_combine(zi - 1);
            return;
        }
        // which taps of this plane / the next plane feed output planes of this workgroup (tap dz -> plane zi + 1 - dz)
        bool vc[3], vn[3];
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
            vc[dz] = (zi + 1 - dz) >= za && (zi + 1 - dz) < zb;
            vn[dz] = (zi + 2 - dz) >= za && (zi + 2 - dz) < zb;
        }
        const bool out_done = zi - 1 >= za;

        // one half-step: 6 groups of 6 MFMAs on A set KS.  Woven between the MFMA pairs: the 4 weight-fragment loads of
        // the group WN_R ahead, the build of the next half-step's A operands (set KS ^ 1) in four pieces, one staging item
        // (first half: store the item loaded a step ago, load the one of three steps ahead) and, when an output plane
        // completes, a quarter of its x transform.  No branches: a tap that feeds no output plane of this workgroup
        // multiplies by the zero fragments.
        auto half = [&](auto ks_tag, const unsigned char* nbuf, const unsigned so_cur, const unsigned so_nxt,
                        const bool* v_cur, const bool* v_nxt, const bool epi, unsigned char* sdst, const unsigned sso) {
            constexpr int KS = decltype(ks_tag)::value;
            using NKS = std::integral_constant<int, KS ^ 1>;
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
#pragma unroll
            for (int sl_ = 0; sl_ < 18; ++sl_) {
                const int g = sl_ / 3, part = sl_ % 3;
                const int dz = 2 - g / 2, fp = g & 1;
                const int gn = (g + WN_R) % 6;
                const bool nx = g + WN_R >= 6;
                const bool ok = nx ? v_nxt[2 - gn / 2] : v_cur[2 - gn / 2];
                const unsigned so = ok ? (nx ? so_nxt : so_cur) + goffs(gn) : wzero;
                const int sl = g % (WN_R + 1), sn = (g + WN_R) % (WN_R + 1);
                const int f0 = fp * 2, f1 = fp * 2 + 1;
                floatx16& c0 = dz == 0 ? acc[S_NEXT][f0] : (dz == 1 ? acc[S_CUR][f0] : acc[S_PREV][f0]);
                floatx16& c1 = dz == 0 ? acc[S_NEXT][f1] : (dz == 1 ? acc[S_CUR][f1] : acc[S_PREV][f1]);
                // weight fragments of the group WN_R ahead: 1 + 1 + 2 loads over the three slices of a group
                if (!(WN_ABL & 4)) {
                    if (part == 0) bfr[sn][0] = wn_load4u(rsw, wvoff, so);
                    if (part == 1) bfr[sn][1] = wn_load4u(rsw, wvoff, so + 1024u);
                    if (part == 2) {
                        bfr[sn][2] = wn_load4u(rsw, wvoff, so + 2048u);
                        bfr[sn][3] = wn_load4u(rsw, wvoff, so + 3072u);
                    }
                }
                // staging item g (first half-step only): store what was loaded a step ago, load three steps ahead
                if (KS == 0 && !(WN_ABL & 1)) {
                    if (part == 1) stage_store_item(sdst, g);
                    if (part == 2) sd[g] = wn_load4(rsx, goff[g], sso);
                }
                // A operands of the next half-step
                if (sl_ == 0) a_reads(nbuf, NKS{}, I0{});
                if (sl_ == 2) a_tv();
                if (sl_ == 3) {
                    a_split(NKS{}, I0{}, I0{});
                    a_reads(nbuf, NKS{}, I1{});
                }
                if (sl_ == 4) a_split(NKS{}, I0{}, I1{});
                if (sl_ == 5) a_split(NKS{}, I0{}, I2{});
                if (sl_ == 6) a_split(NKS{}, I0{}, I3{});
                if (sl_ == 7) a_tv();
                if (sl_ == 8) a_split(NKS{}, I1{}, I0{});
                if (sl_ == 9) a_split(NKS{}, I1{}, I1{});
                if (sl_ == 10) a_split(NKS{}, I1{}, I2{});
                if (sl_ == 11) a_split(NKS{}, I1{}, I3{});
                // x transform of the finished output plane: its accumulators are final after group 1
                if (KS == 1 && epi && sl_ >= 10) epi_write_piece(sl_ - 10);
                if (part == 0) {
                    if (ONE && KS == 0 && dz == 0) {   // FIRST: chunk 0, kstep 0, tap 0 STARTS output plane zi + 1
                        floatx16 z;
#pragma unroll
                        for (int k = 0; k < 16; ++k) z[k] = 0.f;
                        c0 = mfma16<false>(al[KS][f0], bfr[sl][0], z);
                        c1 = mfma16<false>(al[KS][f1], bfr[sl][2], z);
                    } else {
                        c0 = mfma16<false>(al[KS][f0], bfr[sl][0], c0);
                        c1 = mfma16<false>(al[KS][f1], bfr[sl][2], c1);
                    }
                } else if (part == 1) {
                    c0 = mfma16<false>(ah[KS][f0], bfr[sl][1], c0);
                    c1 = mfma16<false>(ah[KS][f1], bfr[sl][3], c1);
                } else {
                    c0 = mfma16<false>(ah[KS][f0], bfr[sl][0], c0);
                    c1 = mfma16<false>(ah[KS][f1], bfr[sl][2], c1);
                }
#if WN_PIN
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        };

        for (int c = 0; c < nch; ++c, ++step) {
            WN_STAMP(0);
            const unsigned char* bcur = raw + (step % WN_NBUF) * WN_PLB;
            const unsigned char* bnxt = raw + ((step + 1) % WN_NBUF) * WN_PLB;
            unsigned char* bst = raw + ((step + 2) % WN_NBUF) * WN_PLB;   // the staging registers hold the chunk of step + 2
            const bool last_c = ONE || c + 1 == nch;
            const int cn = last_c ? 0 : c + 1;
            const unsigned sso = stage_so(step + 3);
            WN_STAMP(1);
            half(std::integral_constant<int, 0>{}, bcur, wsoff(c, 0), wsoff(c, 1), vc, vc, false, bst, sso);
            if (!ONE) stage_norm(step + 3);
            WN_STAMP(2);
            // the x transform of output plane zi-1 runs unconditionally in the last chunk (before the first output plane it
            // writes values nobody combines)
            half(std::integral_constant<int, 1>{}, bnxt, wsoff(c, 1), wsoff(cn, 0), vc, last_c ? vn : vc, last_c, bst, sso);
            WN_STAMP(3);
            WN_STAMP(4);
            __syncthreads();
            WN_STAMP(5);
#if !(WN_ABL & 2)
            if (last_c && out_done) epi_combine(zi - 1);
#endif
            WN_STAMP(6);
        }
    };
    int zi = zlo;
    while (true) {   // plane zb is a real input plane when zb < D, else only the flush of output plane D-1
        plane(std::integral_constant<int, 0>{}, zi);
        if (++zi > zb) break;
        plane(std::integral_constant<int, 1>{}, zi);
        if (++zi > zb) break;
        plane(std::integral_constant<int, 2>{}, zi);
        if (++zi > zb) break;
    }
}

// ---------------------------------------------------------------------------
// Weight pack: U[dz][fy][fx] = G w[dz] G^T per (ci, co) in fp32, split (hi, lo), laid out as the fragment stream the
// kernel walks: [fy][ct][chunk][kstep][dz][fx][hi|lo][lane 64][8].  transpose = data gradient (channels swapped, taps
// flipped).  Frequency row 2 negated (see the kernel).
// ---------------------------------------------------------------------------
__global__ void k_pack_wino(const float* __restrict__ w, unsigned short* __restrict__ dst, int Cw_out, int Cw_in,
                            int transpose, int f16, int CT, int nch, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), fx = (int)((idx >> 9) & 3);
    int64_t rest = idx >> 11;
    const int dz = (int)(rest % 3);
    rest /= 3;
    const int ks = (int)(rest & 1);
    rest >>= 1;
    const int c = (int)(rest % nch);
    rest /= nch;
    const int ct = (int)(rest % CT);
    const int fy = (int)(rest / CT);
    const int Cin_e = transpose ? Cw_out : Cw_in, Cout_e = transpose ? Cw_in : Cw_out;
    const int i = c * 32 + ks * 16 + (lane >> 5) * 8 + e, o = ct * 32 + (lane & 31);
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    float u = 0.f;
    if (i < Cin_e && o < Cout_e) {
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const float wv = transpose ? w[((int64_t)i * Cw_in + o) * 27 + (2 - dz) * 9 + (2 - ky) * 3 + (2 - kx)]
                                           : w[((int64_t)o * Cw_in + i) * 27 + dz * 9 + ky * 3 + kx];
                u = fmaf(G[fy][ky] * G[fx][kx], wv, u);
            }
    }
    if (fy == 2) u = -u;
    unsigned short hi, lo;
    if (f16) {
        const _Float16 h = (_Float16)u;
        const _Float16 l = (_Float16)((u - (float)h) * F16_LO_SCALE);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    } else {
        const __bf16 h = (__bf16)u;
        const __bf16 l = (__bf16)(u - (float)h);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    }
    const int64_t F = (((((int64_t)(fy * CT + ct) * nch + c) * 2 + ks) * 3 + dz) * 4 + fx);
    dst[(F * 2 + 0) * 512 + lane * 8 + e] = hi;
    dst[(F * 2 + 1) * 512 + lane * 8 + e] = lo;
}

int64_t tem_conv_wino_pack_bytes(int Cin, int Cout) {
    const int CT = (Cout + 31) / 32, nch = Cin / 32;
    return (int64_t)4 * CT * nch * 2 * 3 * 4 * 2 * 1024 + 4096;   // + the zero fragments
}

void tem_pack_weights_wino(const float* w, void* dst, int Cw_out, int Cw_in, int transpose, int f16, hipStream_t s) {
    const int Cin_e = transpose ? Cw_out : Cw_in, Cout_e = transpose ? Cw_in : Cw_out;
    const int CT = (Cout_e + 31) / 32, nch = Cin_e / 32;
    const int64_t total = (int64_t)4 * CT * nch * 2 * 3 * 4 * 512;
    hipLaunchKernelGGL(k_pack_wino, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, (unsigned short*)dst, Cw_out,
                       Cw_in, transpose, f16, CT, nch, total);
    (void)hipMemsetAsync((char*)dst + total * 4, 0, 4096, s);
}

struct WinoGeom {
    bool ok;
    int nY, nX, zsegs, nct, nwg;
};
static WinoGeom wino_geometry(int N, int D, int H, int W, int Cin, int Cout, int64_t max_ld) {
    WinoGeom g;
    g.nY = (H + 7) / 8;
    g.nX = (W + 15) / 16;
    g.nct = (Cout + 31) / 32;
    int zs = 1;
    while ((int64_t)N * g.nY * g.nX * g.nct * zs < 256 && D / (zs * 2) >= 8) zs *= 2;
    g.zsegs = zs;
    g.nwg = N * g.nY * g.nX * zs * g.nct;
    // buffer addressing: 32-bit byte offsets within one sample
    g.ok = Cin % 32 == 0 && Cout % 32 == 0 && D >= 2 && (int64_t)D * H * W * max_ld * 4 < (1ll << 31);
    return g;
}

static void wino_launch(const WinoGeom& g, const float* x, int64_t x_ld, const float* scale, const float* shift,
                        const void* wp, const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N,
                        int D, int H, int W, int Cin, int Cout, int act, hipStream_t s) {
    const int lb = WN_NBUF * WN_PLB + 2 * WN_PFL * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wino<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wino<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lb);
        attr = true;
    }
    if (Cin == 32)
        hipLaunchKernelGGL(k_conv_wino<true>, dim3(g.nwg), dim3(256), lb, s, x, x_ld, scale, shift, (const unsigned short*)wp, bias,
                           y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, g.nY, g.nX, g.zsegs, g.nct);
    else
        hipLaunchKernelGGL(k_conv_wino<false>, dim3(g.nwg), dim3(256), lb, s, x, x_ld, scale, shift, (const unsigned short*)wp, bias,
                           y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, g.nY, g.nX, g.zsegs, g.nct);
}

bool tem_conv_fwd_wino(const float* x, int64_t x_ld, const float* scale, const float* shift, const void* wp,
                       const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H,
                       int W, int Cin, int Cout, int act, int f16, hipStream_t s) {
    const int64_t mld = x_ld > y_ld ? (x_ld > ref_ld ? x_ld : ref_ld) : (y_ld > ref_ld ? y_ld : ref_ld);
    const WinoGeom g = wino_geometry(N, D, H, W, Cin, Cout, mld);
    if (!g.ok || act == TEM_ACT_SIGMOID) return false;
    if ((reinterpret_cast<uintptr_t>(y) & 15) || (y_ld & 3) || (ref && ((reinterpret_cast<uintptr_t>(ref) & 15) || (ref_ld & 3))) ||
        (reinterpret_cast<uintptr_t>(x) & 15) || (x_ld & 3))
        return false;
    if (f16) return false;   // bf16x3 only
    wino_launch(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, s);
    return true;
}
