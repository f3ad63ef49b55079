// conv_bf16x3.hip -- implicit-GEMM 3-D convolution on the bf16 matrix cores with fp32-class accuracy:
// every fp32 operand is split into two bf16 terms (x = hi + lo, residual <= 2^-18 |x|) and each
// product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (the dropped lo*lo term is <= 2^-18 of the product).  Per product that is ~1e-5 relative error,
// random in sign, i.e. ~1e-6..1e-5 on a K >= 432 dot product -- two orders inside the 1e-3 parity
// tolerance of the path -- at 3/16 of the exact-fp32 MFMA cost (2.5 PFLOP/s bf16 peak / 3 = 833
// TFLOP/s effective vs 157 TFLOP/s).  Same data flow as conv_mfma.hip (fp32 NDHWC activations in HBM,
// fused pre-norm while staging, bias/ReLU/mask epilogue, split-K for the small levels); the split into
// hi/lo happens once per staged element on the way into LDS, and once per weight at pack time.
#include "tem_common.h"
#include <set>
#include "conv_internal.h"

#ifndef TEM_SPLIT_N
#define TEM_SPLIT_N 1  // NR == 2 workgroups: waves tiled 2 (voxel halves) x 2 (column tiles) instead of 4 x (64 voxels, 64 columns)
#endif
#ifndef TEM_SC_RD
#define TEM_SC_RD 1   // weight ring depth of that kernel (must divide the tap count: 1, 3 or 9); 3 spills 11 VGPRs: +1.5 % step time
#endif
#ifndef TEM_X3_RD1
#define TEM_X3_RD1 3   // bf16x3 kernel (dgrad / no-grad forward), 32-column tiles
#endif
#ifndef TEM_X3_RD2
#define TEM_X3_RD2 3   // bf16x3 kernel, 64-column tiles
#endif
#ifndef TEM_SC2_RD
#define TEM_SC2_RD 3
#endif
#ifndef TEM_SETPRIO
#define TEM_SETPRIO 0
#endif
#ifndef TEM_ZS_NT
#define TEM_ZS_NT 0
#endif
typedef float floatx4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
    floatx4n v = __builtin_nontemporal_load(reinterpret_cast<const floatx4n*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#if TEM_ZS_NT
#define ZS_GLOAD(p) ld4_nt(p)
#else
#define ZS_GLOAD(p) (*reinterpret_cast<const float4*>(p))
#endif
#ifndef TEM_NT_STORE
#define TEM_NT_STORE 1   // epilogue stores bypass the write-allocate path: the output is not re-read by this kernel (-0.3 ms/step)
#endif
#ifndef TEM_SC_WPC
#define TEM_SC_WPC 3   // resident workgroups per CU of the fp16x3 forward kernel with 32-column tiles
#endif
#ifndef TEM_SC_CLAMP
#define TEM_SC_CLAMP 1
#endif
#ifndef TEM_NS1_WPC
#define TEM_NS1_WPC 4  // resident workgroups per CU of the single-product (mixed precision) forward kernel
#endif

#include "conv_split.h"
#include "tem_act.h"

// ---------------------------------------------------------------------------
// weight packing: [Cout][Cin][kd][kh][kw] fp32 -> [co/32][tap][ci/16][NS planes][64 lanes][8 bf16]
// lane = kh*32 + col, slot j <-> input channel (ci/16)*16 + kh*8 + j  (the B fragment of 32x32x16)
// NS = 2: (hi, lo);  NS = 3: three bf16 terms = all 24 mantissa bits of the fp32 weight.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned short bf16_bits(float v) {
    return (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
}

__global__ __launch_bounds__(256) void k_pack_weights_bfsplit(const float* __restrict__ w, unsigned short* __restrict__ dst,
                                                              int Cout, int Cin, int KD, int KH, int KW, int transpose,
                                                              int NS, int fp16) {
    const int ntaps = KD * KH * KW;
    const int64_t total = (int64_t)Cout * Cin * ntaps;
    const int CoutL = transpose ? Cin : Cout, CinL = transpose ? Cout : Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int co = (int)(i % CoutL);
        int64_t r = i / CoutL;
        int ci = (int)(r % CinL);
        int tap = (int)(r / CinL);
        int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
        float val;
        if (!transpose)
            val = w[(((int64_t)co * Cin + ci) * KD + tz) * KH * KW + ty * KW + tx];
        else
            val = w[(((int64_t)ci * Cin + co) * KD + (KD - 1 - tz)) * KH * KW + (KH - 1 - ty) * KW + (KW - 1 - tx)];
        const int nt = co >> 5, col = co & 31, c16 = ci >> 4, kh = (ci >> 3) & 1, j = ci & 7;
        const int64_t base = ((((int64_t)nt * ntaps + tap) * (CinL >> 4) + c16) * NS) * 512;  // 512 bf16 per plane
        float rem = val;
        if (fp16 == 3) rem = __builtin_amdgcn_fmed3f(val * F16_W_PRESCALE, -64000.f, 64000.f);
        for (int p = 0; p < NS; ++p) {
            if (fp16) {
                const _Float16 hv = (_Float16)rem;
                dst[base + p * 512 + (kh * 32 + col) * 8 + j] = __builtin_bit_cast(unsigned short, hv);
                rem -= (float)hv;
                if (fp16 == 2) rem *= F16_LO_SCALE;
            } else {
                const unsigned short hb = bf16_bits(rem);
                dst[base + p * 512 + (kh * 32 + col) * 8 + j] = hb;
                rem -= __builtin_bit_cast(float, (unsigned)hb << 16);
            }
        }
    }
}

int tem_pack_weights_bf16x3(const float* w, float* dst, int Cout, int Cin, int kd, int kh, int kw, int transpose,
                            int nsplit, hipStream_t s) {
    // nsplit 4 = fp16x3: two fp16 planes, the lo plane scaled by 2^12 (fp16 = 2); 5 = fp16: one plane (fp16 = 1);
    // 6 = fp16x3 with the whole weight prescaled by 2^7 (fp16 = 3, conv_split.h)
    // 7 = one bf16 term (the counterpart of torch.autocast(bfloat16)): one plane, bf16 bits
    const int fp16 = nsplit == 4 ? 2 : (nsplit == 5 ? 1 : (nsplit == 6 ? 3 : 0));
    if (fp16) nsplit = nsplit == 5 ? 1 : 2;
    if (nsplit == 7) nsplit = 1;
    int CoutL = transpose ? Cin : Cout, CinL = transpose ? Cout : Cin;
    TEM_REQUIRE(CinL % 16 == 0 && CoutL % 32 == 0, "tem_conv_pack_weights: split-bf16 layout needs Cin%%16==0, Cout%%32==0");
    int64_t total = (int64_t)Cout * Cin * kd * kh * kw;
    hipLaunchKernelGGL(k_pack_weights_bfsplit, dim3(tem_grid_1d(total, 256)), dim3(256), 0, s, w, (unsigned short*)dst,
                       Cout, Cin, kd, kh, kw, transpose, nsplit, fp16);
    return TEM_OK;
}

// One launch for every split-layout weight tensor of a model (after an optimizer step all of them are stale:
// 42 separate pack launches x 15 us were 0.63 ms of a 29 ms training step).
struct PackDesc {
    const float* w;
    unsigned short* dst;
    int Cout, Cin, KD, KH, KW, transpose, NS, fp16;
    long long begin;  // first global work item (8-channel lane slot) of this tensor: running sum of Cout*Cin*taps/8
};
__global__ __launch_bounds__(256) void k_pack_weights_batch(const PackDesc* __restrict__ descs, int n, long long total) {
    // one work item = one MFMA lane slot: 8 consecutive input channels of one (column tile, tap, 16-channel chunk, lane);
    // it gathers 8 weights (L2-resident) and writes NS aligned 16-byte vectors -- a wave writes 1 KB contiguous per plane
    for (long long gi = (long long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long long)gridDim.x * 256) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {  // last descriptor with begin <= gi
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].begin <= gi) lo = mid; else hi = mid - 1;
        }
        const PackDesc d = descs[lo];
        long long i = gi - d.begin;
        const int ntaps = d.KD * d.KH * d.KW;
        if (d.NS == 0) {
            // TEM_WL_GENERIC fp32 layout [tap][ci][co] (k_pack_weights): the first conv (Cin = 1) and out_conv ride along in
            // the batched launch instead of one launch per tensor and direction; one work item = 8 consecutive outputs
            const int CoutL = d.transpose ? d.Cin : d.Cout, CinG = d.transpose ? d.Cout : d.Cin;
            float* dstf = reinterpret_cast<float*>(d.dst);
            for (int j = 0; j < 8; ++j) {
                const long long e = i * 8 + j;
                const int co = (int)(e % CoutL);
                const long long r = e / CoutL;
                const int ci = (int)(r % CinG), tap = (int)(r / CinG);
                const int tz = tap / (d.KH * d.KW), ty = (tap / d.KW) % d.KH, tx = tap % d.KW;
                const int ftap = ((d.KD - 1 - tz) * d.KH + (d.KH - 1 - ty)) * d.KW + (d.KW - 1 - tx);
                dstf[e] = d.transpose ? d.w[((long long)ci * d.Cin + co) * ntaps + ftap] : d.w[((long long)co * d.Cin + ci) * ntaps + tap];
            }
            continue;
        }
        const int CinL = d.transpose ? d.Cout : d.Cin;
        const int c16n = CinL >> 4;
        const int lane = (int)(i & 63);
        i >>= 6;
        const int c16 = (int)(i % c16n);
        i /= c16n;
        const int tap = (int)(i % ntaps);
        const int nt = (int)(i / ntaps);
        const int kh = lane >> 5, col = lane & 31;
        const int co = nt * 32 + col, ci0 = c16 * 16 + kh * 8;
        const int tz = tap / (d.KH * d.KW), ty = (tap / d.KW) % d.KH, tx = tap % d.KW;
        const int ftap = ((d.KD - 1 - tz) * d.KH + (d.KH - 1 - ty)) * d.KW + (d.KW - 1 - tx);
        float rem[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = ci0 + j;
            rem[j] = d.transpose ? d.w[((long long)ci * d.Cin + co) * ntaps + ftap] : d.w[((long long)co * d.Cin + ci) * ntaps + tap];
            if (d.fp16 == 3) rem[j] = __builtin_amdgcn_fmed3f(rem[j] * F16_W_PRESCALE, -64000.f, 64000.f);
        }
        uint4* out = reinterpret_cast<uint4*>(d.dst) + ((((long long)nt * ntaps + tap) * c16n + c16) * d.NS) * 64 + lane;
        for (int p = 0; p < d.NS; ++p) {
            unsigned pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (d.fp16) {
                    const _Float16 a = (_Float16)rem[2 * q], b = (_Float16)rem[2 * q + 1];
                    pk[q] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
                    rem[2 * q] -= (float)a;
                    rem[2 * q + 1] -= (float)b;
                    if (d.fp16 == 2) {
                        rem[2 * q] *= F16_LO_SCALE;
                        rem[2 * q + 1] *= F16_LO_SCALE;
                    }
                } else {
                    const unsigned short a = bf16_bits(rem[2 * q]), b = bf16_bits(rem[2 * q + 1]);
                    pk[q] = (unsigned)a | ((unsigned)b << 16);
                    rem[2 * q] -= __builtin_bit_cast(float, (unsigned)a << 16);
                    rem[2 * q + 1] -= __builtin_bit_cast(float, (unsigned)b << 16);
                }
            }
            out[(long long)p * 64] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
}
// The same re-pack with COALESCED reads: a workgroup stages one [32 out][32 in][taps] tile of a weight tensor (rows of
// 32 * taps consecutive floats, 16-byte loads) in LDS and emits the fragments of that tile from there.  The gather kernel
// above reads 8 floats `taps` apart per work item: 454 MB fetched per launch for 85 MB of weights (cfg 2), 0.21 ms.
// `tbegin` of a record = its first tile in the grid (tiles per tensor: ceil(Cout / 32) * ceil(Cin / 32)).
struct PackTileDesc {
    const float* w;
    unsigned short* dst;
    int Cout, Cin, KD, KH, KW, transpose, NS, fp16;
    long long tbegin;
};
#define PT_MAXF 28672   // LDS floats of a tile: 32 rows x (32 x 27 + 4) = 27776
__global__ __launch_bounds__(1024) void k_pack_weights_tiles(const PackTileDesc* __restrict__ descs, int n) {
    __shared__ float tile[PT_MAXF];
    __shared__ int sh_desc;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {  // last descriptor with tbegin <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (descs[mid].tbegin <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sh_desc = lo;
    }
    __syncthreads();
    const PackTileDesc d = descs[sh_desc];
    const int ntaps = d.KD * d.KH * d.KW;
    const int nci = (d.Cin + 31) >> 5;
    const int t = (int)((long long)blockIdx.x - d.tbegin);
    const int cob = t / nci, cib = t % nci;                  // 32-blocks of the weight's own (out, in) channels
    const int R = min(32, d.Cout - cob * 32), C = min(32, d.Cin - cib * 32);   // 16 or 32
    const int rowf = C * ntaps, pitch = rowf + 4;
    // ---- load: R rows of C * taps consecutive floats (row start and length are multiples of 4 floats) ----
    for (int i = threadIdx.x; i < R * (rowf >> 2); i += 1024) {
        const int rr = i / (rowf >> 2), q = i % (rowf >> 2);
        const float4 v = *reinterpret_cast<const float4*>(d.w + ((long long)(cob * 32 + rr) * d.Cin + cib * 32) * ntaps + q * 4);
        *reinterpret_cast<float4*>(tile + rr * pitch + q * 4) = v;
    }
    __syncthreads();
    // ---- emit: fragments (32 out-channels of the EXECUTED conv x 16 of its in-channels) of every tap ----
    // forward: out = co (R must be 32), in = ci (C / 16 fragments per tap); data gradient: out = ci (C must be 32), in = co
    const int nfr = d.transpose ? (R >> 4) : (C >> 4);
    const int CinL = d.transpose ? d.Cout : d.Cin;
    const int c16n = CinL >> 4;
    const int ntL = d.transpose ? cib : cob;
    for (int it = threadIdx.x; it < ntaps * nfr * 64; it += 1024) {
        const int lane = it & 63, fr = (it >> 6) % nfr, tap = (it >> 6) / nfr;
        const int kh = lane >> 5, col = lane & 31;
        const int tz = tap / (d.KH * d.KW), ty = (tap / d.KW) % d.KH, tx = tap % d.KW;
        const int ftap = ((d.KD - 1 - tz) * d.KH + (d.KH - 1 - ty)) * d.KW + (d.KW - 1 - tx);
        float rem[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = fr * 16 + kh * 8 + j;   // in-channel of the executed conv inside the tile
            rem[j] = d.transpose ? tile[k * pitch + col * ntaps + ftap] : tile[col * pitch + k * ntaps + tap];
            if (d.fp16 == 3) rem[j] = __builtin_amdgcn_fmed3f(rem[j] * F16_W_PRESCALE, -64000.f, 64000.f);
        }
        const int c16 = (d.transpose ? cob * 2 : cib * 2) + fr;
        uint4* out = reinterpret_cast<uint4*>(d.dst) + ((((long long)ntL * ntaps + tap) * c16n + c16) * d.NS) * 64 + lane;
        if (d.fp16 == 4) {
            // exact fp32 (TEM_WL_MFMA, round 6: the exact mode re-packed 43 tensors with one launch each): the two 64-lane groups
            // of a 16-channel chunk hold channels 8 p + 4 kh + (0..3) as they are -- rem[] has 8 kh + j: gather the other four
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float f[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = fr * 16 + 8 * p + 4 * kh + c;
                    f[c] = d.transpose ? tile[k * pitch + col * ntaps + ftap] : tile[col * pitch + k * ntaps + tap];
                }
                out[(long long)p * 64] = make_uint4(__builtin_bit_cast(unsigned, f[0]), __builtin_bit_cast(unsigned, f[1]),
                                                    __builtin_bit_cast(unsigned, f[2]), __builtin_bit_cast(unsigned, f[3]));
            }
            continue;
        }
        for (int p = 0; p < d.NS; ++p) {
            unsigned pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (d.fp16) {
                    const _Float16 a = (_Float16)rem[2 * q], b = (_Float16)rem[2 * q + 1];
                    pk[q] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
                    rem[2 * q] -= (float)a;
                    rem[2 * q + 1] -= (float)b;
                    if (d.fp16 == 2) {
                        rem[2 * q] *= F16_LO_SCALE;
                        rem[2 * q + 1] *= F16_LO_SCALE;
                    }
                } else {
                    const unsigned short a = bf16_bits(rem[2 * q]), b = bf16_bits(rem[2 * q + 1]);
                    pk[q] = (unsigned)a | ((unsigned)b << 16);
                    rem[2 * q] -= __builtin_bit_cast(float, (unsigned)a << 16);
                    rem[2 * q + 1] -= __builtin_bit_cast(float, (unsigned)b << 16);
                }
            }
            out[(long long)p * 64] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
}
extern "C" int tem_conv_pack_weights_tiles(const void* descs_dev, int n, int64_t total_tiles, tem_stream_t stream) {
    TEM_REQUIRE(descs_dev && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "tem_conv_pack_weights_tiles: bad arguments");
    hipLaunchKernelGGL(k_pack_weights_tiles, dim3((unsigned)total_tiles), dim3(1024), 0, (hipStream_t)stream,
                       (const PackTileDesc*)descs_dev, n);
    TEM_CHECK_LAUNCH("tem_conv_pack_weights_tiles");
    return TEM_OK;
}

extern "C" int tem_conv_pack_weights_batch(const void* descs_dev, int n, int64_t total, tem_stream_t stream) {
    TEM_REQUIRE(descs_dev && n > 0 && total > 0, "tem_conv_pack_weights_batch: bad arguments");
    hipLaunchKernelGGL(k_pack_weights_batch, dim3(tem_grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const PackDesc*)descs_dev, n, (long long)total);
    TEM_CHECK_LAUNCH("tem_conv_pack_weights_batch");
    return TEM_OK;
}

// ---------------------------------------------------------------------------
// forward / dgrad.  NS = 2: "bf16x3" (3 MFMAs per product, ~1e-5 relative), used for the gradient
// side.  NS = 3: "bf16x6": x = a1+a2+a3 carries all 24 mantissa bits, products with i+j <= 4
// (6 MFMAs) -- per-product error ~2^-23, i.e. the fp32 class, at 16/6 of the exact-fp32 MFMA rate;
// used for the forward pass, whose rounding noise the gradient amplifies (engine.py, PRECISION).
// ---------------------------------------------------------------------------
template <int KD, int KH, int KW, int TZ, int TY, int TX, int NR, int NS, bool F16 = false, bool PS = false, typename T = float>
__global__ __launch_bounds__(256, NS == 1 ? TEM_NS1_WPC : (F16 && NR == 1) ? TEM_SC_WPC : (NR == 2 || NS == 3) ? 2 : 3) void k_conv_fwd_bfsplit(
    const T* __restrict__ x, int64_t x_ld, const float* __restrict__ scale, const float* __restrict__ shift,
    const uint4* __restrict__ wp, const float* __restrict__ bias, T* __restrict__ y, int64_t y_ld,
    const T* __restrict__ ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout, int act, int nZ,
    int nY, int nX, int ksplit, float* __restrict__ part, float* __restrict__ stat) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1;
    constexpr int HV = HZ * HY * HX;
    constexpr int NIT = (HV * 4 + 255) / 256;
    constexpr int RD = (NT % 3 == 0) ? ((F16 && NS == 2) ? (NR == 1 ? TEM_SC_RD : TEM_SC2_RD) : (NS == 2 ? (NR == 1 ? TEM_X3_RD1 : TEM_X3_RD2) : 3)) : 1;  // weight-fragment ring depth over taps
    static_assert(NT % RD == 0, "the ring slot of a tap must not depend on the chunk");
    constexpr int LSV = NS * 8 + 4;            // LDS floats per halo voxel: NS planes of 16 bf16 (32 B) + 16 B pad
    // Wave tiling.  The vector-memory pipe (TA) is as loaded as the matrix pipe: per 16-channel chunk a workgroup pulls
    // 38 KB of halo but 27 taps x NS KB of weight fragments PER WAVE through it.  With two 32-column tiles per workgroup
    // (NR == 2) the waves are arranged 2 (voxel halves) x 2 (column tiles): a wave owns 4 M-tiles x 1 column tile, so it
    // loads HALF the weight fragments per MFMA and reads twice the A fragments from LDS instead (LDS reads are not the
    // limiter: serving the three tx taps from one read changed nothing).  NR == 1: 4 waves x (2 M-tiles x 1 tile).
    constexpr bool SN = (NR == 2) && TEM_SPLIT_N;
    constexpr int MT = SN ? 4 : 2;             // M-tiles (32 voxels) per wave
    constexpr int NW = SN ? 1 : NR;            // column tiles per wave
    static_assert(TZ * TY * TX == 256, "patch must hold 256 voxels");
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [HV][LSV]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;

    int bid = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int ncot = Cout / (32 * NR);
    const int cot = bid % ncot;
    bid /= ncot;
    const int ptx = bid % nX;
    bid /= nX;
    const int pty = bid % nY;
    bid /= nY;
    const int ptz = bid % nZ;
    bid /= nZ;
    const int n = bid % N;
    const int ks = bid / N;
    const int z0 = ptz * TZ, y0 = pty * TY, x0 = ptx * TX;

    const int pw0 = SN ? (wv & 1) * 128 : wv * 64;  // first patch voxel of this wave
    const int nw0 = SN ? (wv >> 1) : 0;             // first column tile of this wave
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = pw0 + m * 32 + r;
        const int pz = p / (TY * TX), py = (p / TX) % TY, px = p % TX;
        abase[m] = ((pz * HY + py) * HX + px) * LSV + kh * 4;  // + 8 floats (32 B) per further plane
    }
    constexpr bool SC = F16 && NS == 2 && !PS; // fp16x3: scaled lo planes, cross products in their own accumulators
                                               // (PS: whole operands prescaled instead, one accumulator, conv_split.h)
    floatx16 acc[MT][NW];
    floatx16 accl[SC ? MT : 1][SC ? NW : 1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < NW; ++nn)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[m][nn][i] = 0.f;
                if (SC) accl[SC ? m : 0][SC ? nn : 0][i] = 0.f;
            }

    const int cin16 = Cin >> 4;
    const int c4 = tid & 3;
    const int cpk = cin16 / ksplit;
    const int chunk_begin = ks * cpk, chunk_end = (ks + 1) * cpk;
    // uint4 index of this lane's slot: ((((nt*NT + tap)*cin16 + c16)*NS + plane)*64 + lane)
    constexpr int FR = NS * 64;                // uint4s per (tap, c16) fragment group
    const int tapstride = cin16 * FR;
    const uint4* wq[NW];
#pragma unroll
    for (int nn = 0; nn < NW; ++nn) wq[nn] = wp + (int64_t)(cot * NR + nw0 + nn) * NT * cin16 * FR + lane;
    uint4 bq[RD][NW][NS];
    if (RD > 1) {
#pragma unroll
        for (int gp = 0; gp < RD - 1; ++gp)
#pragma unroll
            for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                for (int p = 0; p < NS; ++p) bq[gp][nn][p] = wq[nn][(int64_t)chunk_begin * FR + gp * tapstride + p * 64];
    }
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        // ---- stage: global fp32 -> fused pre-norm -> NS bf16 planes -> LDS ----
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) {
            sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + chunk * BCK + c4 * 4);
            sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + chunk * BCK + c4 * 4);
        }
        float4 tmp[NIT];
        // every halo load of this thread is issued from straight-line code (clamped address, result masked afterwards):
        // a load + its norm FMA inside a per-element bounds branch made the ten loads wait for one another
        bool inb[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hv = min((tid + it * 256) >> 2, HV - 1);
            const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
            const int gz = z0 + hz - PZ, gy = y0 + hy - PY, gx = x0 + hx - PX;
            inb[it] = (gz >= 0) & (gz < D) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W) & (((tid + it * 256) >> 2) < HV);
            const int cz = min(max(gz, 0), D - 1), cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
            tmp[it] = act_ld4(x + ((((int64_t)n * D + cz) * H + cy) * W + cx) * x_ld + chunk * BCK + c4 * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            float4 v = tmp[it];
            v.x = inb[it] ? fmaf(v.x, sc4.x, sf4.x) : 0.f;
            v.y = inb[it] ? fmaf(v.y, sc4.y, sf4.y) : 0.f;
            v.z = inb[it] ? fmaf(v.z, sc4.z, sf4.z) : 0.f;
            v.w = inb[it] ? fmaf(v.w, sc4.w, sf4.w) : 0.f;
            tmp[it] = v;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hv = (tid + it * 256) >> 2;
            if (hv < HV) {
                float e[4] = {tmp[it].x, tmp[it].y, tmp[it].z, tmp[it].w};
                if (PS) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) e[c] *= F16_A_PRESCALE;
                }
                if (F16 && TEM_SC_CLAMP) {
                    // an activation beyond the fp16 range (|x^| > 6e4 after the norm: not a training state) saturates
                    // instead of turning the whole receptive field into NaN
#pragma unroll
                    for (int c = 0; c < 4; ++c) e[c] = __builtin_amdgcn_fmed3f(e[c], -60000.f, 60000.f);
                }
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    unsigned h0 = pk16<F16>(e[0], e[1]), h1 = pk16<F16>(e[2], e[3]);
                    *reinterpret_cast<uint2*>(lds + hv * LSV + p * 8 + c4 * 2) = make_uint2(h0, h1);
                    if (p + 1 < NS) {
                        e[0] -= lo16<F16>(h0);
                        e[1] -= hi16<F16>(h0);
                        e[2] -= lo16<F16>(h1);
                        e[3] -= hi16<F16>(h1);
                        if (SC) {
                            e[0] *= F16_LO_SCALE;
                            e[1] *= F16_LO_SCALE;
                            e[2] *= F16_LO_SCALE;
                            e[3] *= F16_LO_SCALE;
                        }
                    }
                }
            }
        }
        __syncthreads();

        int ts = tapstride;
        asm volatile("" : "+s"(ts));
        if (TEM_SETPRIO) __builtin_amdgcn_s_setprio(TEM_SETPRIO);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
            const int toff = ((tz * HY + ty) * HX + tx) * LSV;
            if (RD > 1) {
                const int gp = tap + RD - 1;
                if (gp < NT) {
#pragma unroll
                    for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                        for (int p = 0; p < NS; ++p)
                            bq[gp % RD][nn][p] = wq[nn][(int64_t)chunk * FR + (int64_t)gp * ts + p * 64];
                } else if (chunk + 1 < chunk_end) {
#pragma unroll
                    for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                        for (int p = 0; p < NS; ++p)
                            bq[gp % RD][nn][p] = wq[nn][(int64_t)(chunk + 1) * FR + (int64_t)(gp - NT) * ts + p * 64];
                }
                __builtin_amdgcn_sched_barrier(0x38F);
            } else {
#pragma unroll
                for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                    for (int p = 0; p < NS; ++p) bq[0][nn][p] = wq[nn][(int64_t)chunk * FR + (int64_t)tap * ts + p * 64];
            }
            uint4 af[MT][NS];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    af[m][p] = *reinterpret_cast<const uint4*>(lds + abase[m] + toff + p * 8);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NW; ++nn) {
                    // smallest terms first: all plane pairs (i, j) with i + j <= NS - 1 (0-based).  (Product-major
                    // issue, which helps the wgrad kernel, is 10 % SLOWER here: accumulator-major lets the MFMAs of
                    // m = 0 start while the A fragments of m = 1 are still in flight.)
#pragma unroll
                    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                        for (int i = 0; i <= sum; ++i) {
                            const int j = sum - i;
                            if (SC && sum == 1)
                                accl[SC ? m : 0][SC ? nn : 0] = mfma16<F16>(af[m][i], bq[tap % RD][nn][j], accl[SC ? m : 0][SC ? nn : 0]);
                            else
                                acc[m][nn] = mfma16<F16>(af[m][i], bq[tap % RD][nn][j], acc[m][nn]);
                        }
                }
        }
        if (TEM_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }

    if (SC) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[m][nn][i] = fmaf(accl[SC ? m : 0][SC ? nn : 0][i], 1.f / F16_LO_SCALE, acc[m][nn][i]);
    }
    if (PS) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[m][nn][i] *= F16_PRESCALE_INV;
    }
    // optional fused forward statistics of the STORED output (the next layer's InstanceNorm / GroupNorm / BatchNorm):
    // per (sample, patch, channel) partial sums (sum y, sum y^2); merged in fp64 by tem_norm_finalize_partials
    float ssum[NW], ssq[NW];
#pragma unroll
    for (int nn = 0; nn < NW; ++nn) ssum[nn] = ssq[nn] = 0.f;
#pragma unroll
    for (int nn = 0; nn < NW; ++nn) {
        const int co = (cot * NR + nw0 + nn) * 32 + r;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                const int p = pw0 + m * 32 + row;
                const int pz = p / (TY * TX), py = (p / TX) % TY, px = p % TX;
                const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
                if (gz < D && gy < H && gx < W) {
                    const int64_t v = (((int64_t)n * D + gz) * H + gy) * W + gx;
                    if (ksplit > 1) {
                        part[((int64_t)ks * N * D * H * W + v) * Cout + co] = acc[m][nn][reg];
                        continue;
                    }
                    float o = act_apply_b(acc[m][nn][reg] + bv, act);
                    if (ref && !(act_ld1(ref + v * ref_ld + co) > 0.f)) o = 0.f;
                    if constexpr (sizeof(T) == 2) {   // the statistics describe the tensor as stored
                        const T ot = (T)o;
                        o = (float)ot;
                        y[v * y_ld + co] = ot;
                    } else if (TEM_NT_STORE)
                        __builtin_nontemporal_store(o, y + v * y_ld + co);
                    else
                        y[v * y_ld + co] = o;
                    ssum[nn] += o;
                    ssq[nn] = fmaf(o, o, ssq[nn]);
                }
            }
        }
    }
    if (stat) {  // grid-uniform
#pragma unroll
        for (int nn = 0; nn < NW; ++nn) {
            ssum[nn] += __shfl_xor(ssum[nn], 32, 64);  // the two lane halves hold different rows of the same column
            ssq[nn] += __shfl_xor(ssq[nn], 32, 64);
        }
        __syncthreads();  // every wave is done with the halo tile
        float* red = lds;  // [wave 4][NW][32][2]
        if (kh == 0) {
#pragma unroll
            for (int nn = 0; nn < NW; ++nn) {
                red[((wv * NW + nn) * 32 + r) * 2 + 0] = ssum[nn];
                red[((wv * NW + nn) * 32 + r) * 2 + 1] = ssq[nn];
            }
        }
        __syncthreads();
        if (tid < NR * 32) {
            const int j = tid >> 5, rr = tid & 31;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                if (SN && (w4 >> 1) != j) continue;
                const int nn = SN ? 0 : j;
                a += red[((w4 * NW + nn) * 32 + rr) * 2 + 0];
                b += red[((w4 * NW + nn) * 32 + rr) * 2 + 1];
            }
            const int64_t patch = ((int64_t)ptz * nY + pty) * nX + ptx;
            const int64_t nblk = (int64_t)nZ * nY * nX;
            float* dst = stat + (((int64_t)n * nblk + patch) * Cout + (cot * NR + j) * 32 + rr) * 2;
            dst[0] = a;
            dst[1] = b;
        }
    }
}

template <int KD, int KH, int KW, int TZ, int TY, int TX, int NR, int NS, bool F16 = false, bool PS = false, typename T = float>
static void launch_b(const float* x_, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                     const float* bias, float* y_, int64_t y_ld, const float* ref_, int64_t ref_ld, int N, int D, int H,
                     int W, int Cin, int Cout, int act, int ksplit, float* part, float* stat, hipStream_t s) {
    const T* x = reinterpret_cast<const T*>(x_);
    const T* ref = reinterpret_cast<const T*>(ref_);
    T* y = reinterpret_cast<T*>(y_);
    constexpr int HV = (TZ + KD - 1) * (TY + KH - 1) * (TX + KW - 1);
    const int nZ = (D + TZ - 1) / TZ, nY = (H + TY - 1) / TY, nX = (W + TX - 1) / TX;
    const int64_t nblk = (int64_t)N * nZ * nY * nX * (Cout / (32 * NR)) * ksplit;
    constexpr size_t ldsb = (size_t)HV * (NS * 8 + 4) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done && ldsb > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_fwd_bfsplit<KD, KH, KW, TZ, TY, TX, NR, NS, F16, PS, T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_fwd_bfsplit<KD, KH, KW, TZ, TY, TX, NR, NS, F16, PS, T>), dim3((unsigned)nblk), dim3(256), ldsb, s, x,
                       x_ld, scale, shift, reinterpret_cast<const uint4*>(wp), bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin,
                       Cout, act, nZ, nY, nX, ksplit, part, ksplit > 1 ? nullptr : stat);
    if (ksplit > 1) {
        const int64_t NV = (int64_t)N * D * H * W;
        tem_splitk_epilogue(part, ksplit, NV, Cout, bias, act, ref_, ref_ld, y_, y_ld, s);
    }
}

int tem_conv_fwd_bf16x3(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                        const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                        int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                        int nsplit, float* stat, hipStream_t s) {
    TEM_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0, "tem_conv3d_fwd(split-bf16): needs Cin%%16==0 and Cout%%32==0 (got %d,%d)",
                Cin, Cout);
    const int st = tem_call_st.x;
    TEM_REQUIRE(st == tem_call_st.y, "tem_conv3d_fwd(split-bf16): x and y must have the same storage type");
    TEM_REQUIRE(st == 0 || (st == 1 && nsplit == 5) || (st == 2 && nsplit == 7),
                "tem_conv3d_fwd(split-bf16): 16-bit storage goes with the one-term mode of the same type (fp16: use_mfma 5, "
                "bf16: use_mfma 7), got storage %d with use_mfma %d", st, nsplit);
    TEM_REQUIRE(x_ld % (st ? 8 : 4) == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)wp % 16 == 0),
                "tem_conv3d_fwd(split-bf16): x / packed weights must be 16-byte aligned with ld%%4==0 (16-bit storage: ld%%8==0)");
    TEM_REQUIRE(!scale || (((uintptr_t)scale % 16 == 0) && ((uintptr_t)shift % 16 == 0)),
                "tem_conv3d_fwd(split-bf16): scale/shift must be 16-byte aligned");
    const int zr = tem_conv_fwd_zr(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, kd, kh, kw, act,
                                   nsplit, stat, s);
    if (zr < 0) return TEM_EINVAL;
    if (zr) return TEM_OK;
    const int pp = tem_conv_fwd_pp(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, kd, kh, kw, act,
                                   nsplit, stat, s);
    if (pp < 0) return TEM_EINVAL;
    if (pp) return TEM_OK;
    if (tem_conv_fwd_zr_splitk(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, ws, ws_bytes, N, D, H, W, Cin, Cout, kd, kh,
                               kw, act, nsplit, stat, s))
        return TEM_OK;
    TEM_REQUIRE(!stat || tem_conv_zr_splitk_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit) < 0,
                "tem_conv3d_fwd_stats: the split-K launch that writes the statistics needs its workspace "
                "(tem_conv3d_fwd_ws) and 16-byte aligned y / ref / bias");
    if (kd == 1 && kh == 1 && kw == 1 && tem_option(TEM_OPT_CONV1X1_STREAM) &&
        tem_conv1x1_stream(x, x_ld, scale, wp, bias, y, y_ld, ref, ref_ld, (int64_t)N * D * H * W, Cin, Cout, act, nsplit, stat, s))
        return TEM_OK;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    const bool flat = (D == 1 && kd == 1);
    const int TZ = flat ? 1 : 4, TY = flat ? 16 : 8, TX = flat ? 16 : 8;
    const bool nr2 = (Cout % 64 == 0);
    const int64_t nblk0 = (int64_t)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * (Cout / (nr2 ? 64 : 32));
    int ks = tem_fwd_ksplit(nblk0, Cin / BCK);
    const bool vec_ok = (y_ld % 4 == 0) && ((uintptr_t)y % 16 == 0) &&
                        (!ref || (ref_ld % 4 == 0 && (uintptr_t)ref % 16 == 0)) && (!bias || (uintptr_t)bias % 16 == 0);
    if (ks > 1 && (!ws || !vec_ok || ws_bytes < (int64_t)ks * N * D * H * W * Cout * 4)) ks = 1;
    TEM_REQUIRE(!stat || ks == 1, "tem_conv3d_fwd_stats: this shape runs split-K (tem_conv3d_fwd_stat_blocks() == 0)");
    float* part = (float*)ws;
#define GO2(KD, KH, KW, TZ, TY, TX, NS)                                                                             \
    do {                                                                                                            \
        if (nr2)                                                                                                    \
            launch_b<KD, KH, KW, TZ, TY, TX, 2, NS>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, \
                                                    Cin, Cout, act, ks, part, stat, s);                                   \
        else                                                                                                        \
            launch_b<KD, KH, KW, TZ, TY, TX, 1, NS>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, \
                                                    Cin, Cout, act, ks, part, stat, s);                                   \
    } while (0)
#define GO(KD, KH, KW, TZ, TY, TX)                                                                                    \
    do {                                                                                                              \
        if (nsplit == 3)                                                                                              \
            GO2(KD, KH, KW, TZ, TY, TX, 3);                                                                           \
        else if (nsplit == 4) {                                                                                       \
            if (nr2)                                                                                                  \
                launch_b<KD, KH, KW, TZ, TY, TX, 2, 2, true>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                             W, Cin, Cout, act, ks, part, stat, s);                         \
            else                                                                                                      \
                launch_b<KD, KH, KW, TZ, TY, TX, 1, 2, true>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                             W, Cin, Cout, act, ks, part, stat, s);                         \
        } else if (nsplit == 6) {                                                                                     \
            if (nr2)                                                                                                  \
                launch_b<KD, KH, KW, TZ, TY, TX, 2, 2, true, true>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, \
                                                                   H, W, Cin, Cout, act, ks, part, stat, s);               \
            else                                                                                                      \
                launch_b<KD, KH, KW, TZ, TY, TX, 1, 2, true, true>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, \
                                                                   H, W, Cin, Cout, act, ks, part, stat, s);               \
        } else if (nsplit == 5 && st == 1) {                                                                          \
            if (nr2)                                                                                                  \
                launch_b<KD, KH, KW, TZ, TY, TX, 2, 1, true, false, tem_f16>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                             W, Cin, Cout, act, ks, part, stat, s);                         \
            else                                                                                                      \
                launch_b<KD, KH, KW, TZ, TY, TX, 1, 1, true, false, tem_f16>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                             W, Cin, Cout, act, ks, part, stat, s);                         \
        } else if (nsplit == 7 && st == 2) {                                                                          \
            if (nr2)                                                                                                  \
                launch_b<KD, KH, KW, TZ, TY, TX, 2, 1, false, false, tem_bf16>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                              W, Cin, Cout, act, ks, part, stat, s);                        \
            else                                                                                                      \
                launch_b<KD, KH, KW, TZ, TY, TX, 1, 1, false, false, tem_bf16>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                              W, Cin, Cout, act, ks, part, stat, s);                        \
        } else if (nsplit == 5) {                                                                                     \
            if (nr2)                                                                                                  \
                launch_b<KD, KH, KW, TZ, TY, TX, 2, 1, true>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                             W, Cin, Cout, act, ks, part, stat, s);                         \
            else                                                                                                      \
                launch_b<KD, KH, KW, TZ, TY, TX, 1, 1, true>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                             W, Cin, Cout, act, ks, part, stat, s);                         \
        } else if (nsplit == 7) {                                                                                     \
            if (nr2)                                                                                                  \
                launch_b<KD, KH, KW, TZ, TY, TX, 2, 1, false>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                              W, Cin, Cout, act, ks, part, stat, s);                        \
            else                                                                                                      \
                launch_b<KD, KH, KW, TZ, TY, TX, 1, 1, false>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, \
                                                              W, Cin, Cout, act, ks, part, stat, s);                        \
        } else                                                                                                        \
            GO2(KD, KH, KW, TZ, TY, TX, 2);                                                                           \
    } while (0)
    if (key == 7) {
        GO(3, 3, 3, 4, 8, 8);
    } else if (key == 3) {
        if (flat)
            GO(1, 3, 3, 1, 16, 16);
        else
            GO(1, 3, 3, 4, 8, 8);
    } else if (key == 0) {
        if (flat)
            GO(1, 1, 1, 1, 16, 16);
        else
            GO(1, 1, 1, 4, 8, 8);
    } else {
        tem_set_error("tem_conv3d_fwd(split-bf16): kernel (%d,%d,%d) has no MFMA instantiation", kd, kh, kw);
        return TEM_EINVAL;
    }
#undef GO
#undef GO2
    return TEM_OK;
}

// ---------------------------------------------------------------------------

// number of per-sample statistic blocks (= patches) the fused-statistics forward writes, 0 when this shape cannot
// produce them (split-K over the input channels, or no MFMA instantiation)
int64_t tem_conv_fwd_bf16x3_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit) {
    if (Cin % 16 || Cout % 32) return 0;
    const int64_t zrb = tem_conv_zr_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit);
    if (zrb >= 0) return zrb;
    const int64_t ppb = tem_conv_pp_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit);
    if (ppb >= 0) return ppb;
    if (tem_conv_zr_splitk_ks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit)) {   // z-reuse kernel with split input channels: its epilogue
        const int64_t skb = tem_conv_zr_splitk_stat_blocks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit);
        return skb > 0 ? skb : 0;
    }
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    if (key != 7 && key != 3 && key != 0) return 0;
    const bool flat = (D == 1 && kd == 1);
    const int TZ = flat ? 1 : 4, TY = flat ? 16 : 8, TX = flat ? 16 : 8;
    const bool nr2 = (Cout % 64 == 0);
    const int64_t per = (int64_t)((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX);
    if (tem_fwd_ksplit((int64_t)N * per * (Cout / (nr2 ? 64 : 32)), Cin / BCK) > 1) return 0;
    return per;
}

// ---------------------------------------------------------------------------
// weight gradient on the bf16 matrix cores (split-bf16, fp32 accumulate)
//   dw[tap][ci][co] = sum_v xhat[v+tap][ci] * g[v][co]
// GEMM view per tap: M = ci (32), N = co (32), K = voxels; v_mfma_f32_32x32x16_bf16 wants 8
// consecutive k per lane, i.e. 8 voxels of ONE channel -- the transpose of the NDHWC layout.  The
// transpose is paid once per staged element: x halo tile and g tile are written to LDS channel-major
// ([ci][halo row][16 slots] / [co][128 voxels], bf16 hi and lo planes).  A k-slab is two x-rows of
// 8 voxels (lane half kh takes one row each), so the B fragment is one aligned ds_read_b128 and the
// A fragment of tap (tz,ty,tx) is the 8-element window starting at slot tx of halo row
// (z+tz, y+ty): one row read (b128 + b32) serves the three tx taps, tx=1 via v_alignbyte.
// The three tx accumulators of a (tz,ty) "row group" stay in one wave; row groups x Cout tiles are
// dealt to the 4 waves.  One workgroup per CU (about 100 KB LDS, up to 512 VGPRs/wave): the next
// patch's global loads are issued before the MFMAs of the current one.  Split over patch ranges,
// partial slabs merged by the deterministic fp64 reduction (tem_reduce_slabs).
// ---------------------------------------------------------------------------
#define WB_TZ 2
#define WB_TY 8
#define WB_TX 8
#define WB_PV (WB_TZ * WB_TY * WB_TX)   // 128 patch voxels = 8 k-slabs
#define WB_GS 272                        // bytes per co row of Gt: 128 bf16 + 16 pad

// PTZ x PTY x 8 patch of 128 voxels: 2 x 8 x 8 for volumes, 1 x 16 x 8 for 2-D data (D == 1: a 2-plane patch would be
// half padding)
template <int KD, int KH, int KW, int NCO, int KS2, int PTZ = 2, typename TS = float>
__global__ __launch_bounds__(256, 1) void k_conv_wgrad_bf16x3(const TS* __restrict__ x, int64_t x_ld,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const TS* __restrict__ g, int64_t g_ld,
                                                              float* __restrict__ part, float* __restrict__ dbpart,
                                                              int N, int D, int H, int W, int Cin, int Cout, int T,
                                                              int S, int P, int nZ, int nY, int nX) {
    constexpr int NT = KD * KH * KW;
    constexpr int NRG = KD * KH;  // row groups (tz, ty)
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int PTY = 16 / PTZ;
    constexpr int HZ = PTZ + KD - 1, HY = PTY + KH - 1, HX = WB_TX + KW - 1;
    constexpr int ROWS = HZ * HY;
    constexpr int CIS = ROWS * 32 + 16;          // bytes per ci plane of Xt (padded: conflict-free b128 reads)
    constexpr int XPL = 32 * CIS;                // bytes per (hi|lo) plane of Xt
    constexpr int GC = 32 * NCO;
    constexpr int GPL = GC * WB_GS;              // bytes per (hi|lo) plane of Gt
    constexpr int XPAIRS = HX / 2;               // x-adjacent voxel pairs per halo row
    constexpr int XITEMS = ROWS * XPAIRS * 8;    // (row, pair, channel quad)
    constexpr int XIT = (XITEMS + 255) / 256;
    constexpr int GITEMS = 16 * 4 * (GC / 4);    // (patch row, pair, channel quad)
    constexpr int GIT = GITEMS / 256;
    // work units dealt round-robin to the 4 waves: (row group, co tile, k-half), KW accumulators each (the three
    // tx taps share one row window).  KS2 = 2 splits the 8 k-slabs of a patch between two units that write
    // separate partial slabs: with one co tile that turns 9 units (3,2,2,2 per wave) into 18 (5,5,4,4).
    constexpr int MAXU = (NRG * NCO * KS2 + 3) / 4;
    constexpr int ACW = KW;
    constexpr int SPU = 8 / KS2;                 // k-slabs per unit and patch
    static_assert(GITEMS % 256 == 0, "g staging items");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* Xh = ldsb;
    unsigned char* Xl = ldsb + XPL;
    unsigned char* Gh = ldsb + 2 * XPL;
    unsigned char* Gl = Gh + GPL;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const int bid = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid % T, sp = bid / T;
    const int ncit = Cin >> 5;
    const int cit = tile % ncit, cog = tile / ncit;
    int nco_here = (Cout >> 5) - cog * NCO;
    if (nco_here > NCO) nco_here = NCO;
    const int U = NRG * nco_here * KS2;

    floatx16 acc[MAXU][ACW];
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
#pragma unroll
        for (int t = 0; t < ACW; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][t][k] = 0.f;

    // per-unit LDS offsets (a non-existent last unit aliases unit 0; its result is dropped)
    int urow[MAXU], uct[MAXU], uhalf[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        int u = wv + 4 * i;
        if (u >= U) u = 0;
        const int rg = u % NRG;
        urow[i] = ((rg / KH) * HY + (rg % KH)) * 32;
        uct[i] = (u / NRG) % nco_here;
        uhalf[i] = u / (NRG * nco_here);
    }
    const int p_lo = (int)(((int64_t)sp * P) / S), p_hi = (int)(((int64_t)(sp + 1) * P) / S);
    const bool do_db = (dbpart != nullptr) && (cit == 0);
    float dbacc[GIT][4];
#pragma unroll
    for (int it = 0; it < GIT; ++it)
#pragma unroll
        for (int c = 0; c < 4; ++c) dbacc[it][c] = 0.f;

    float4 xa[XIT], xb[XIT], ga[GIT], gb[GIT];
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned inbA = 0, inbB = 0;
    const int xcq = tid & 7;                 // channel quad of this thread's X items (256 % 8 == 0)

#define WB_LOAD(PIDX)                                                                                              \
    do {                                                                                                           \
        int q_ = (PIDX);                                                                                           \
        const int ptx_ = q_ % nX; q_ /= nX;                                                                        \
        const int pty_ = q_ % nY; q_ /= nY;                                                                        \
        const int ptz_ = q_ % nZ;                                                                                  \
        const int n_ = q_ / nZ;                                                                                    \
        const int z0_ = ptz_ * PTZ, y0_ = pty_ * PTY, x0_ = ptx_ * WB_TX;                                              \
        if (scale) {                                                                                               \
            sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n_ * Cin + cit * 32 + xcq * 4);                \
            sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n_ * Cin + cit * 32 + xcq * 4);                \
        }                                                                                                          \
        inbA = 0; inbB = 0;                                                                                        \
        _Pragma("unroll") for (int it = 0; it < XIT; ++it) {                                                       \
            const int item = tid + it * 256;                                                                       \
            const int rp = item >> 3;                                                                              \
            xa[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            xb[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            if (rp < ROWS * XPAIRS) {                                                                              \
                const int row = rp / XPAIRS, pr = rp % XPAIRS;                                                     \
                const int gz = z0_ + row / HY - PZ, gy = y0_ + row % HY - PY, gx = x0_ + 2 * pr - PX;              \
                if (gz >= 0 && gz < D && gy >= 0 && gy < H) {                                \
                    const TS* rowp = x + (((int64_t)n_ * D + gz) * H + gy) * W * x_ld + cit * 32 + xcq * 4;    \
                    if (gx >= 0 && gx < W) { xa[it] = act_ld4(rowp + (int64_t)gx * x_ld); inbA |= 1u << it; } \
                    if (gx + 1 >= 0 && gx + 1 < W) { xb[it] = act_ld4(rowp + (int64_t)(gx + 1) * x_ld); inbB |= 1u << it; } \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int it = 0; it < GIT; ++it) {                                                       \
            const int item = tid + it * 256;                                                                       \
            const int cq = item % (GC / 4), rp = item / (GC / 4);                                                  \
            const int prow = rp >> 2, pr = rp & 3;                                                                 \
            const int gz = z0_ + prow / PTY, gy = y0_ + prow % PTY, gx = x0_ + 2 * pr;                             \
            ga[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            gb[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                              \
            if (gz < D && gy < H && cq < nco_here * 8) {                                     \
                const TS* rowp = g + (((int64_t)n_ * D + gz) * H + gy) * W * g_ld + cog * NCO * 32 + cq * 4;   \
                if (gx < W) ga[it] = act_ld4(rowp + (int64_t)gx * g_ld);                  \
                if (gx + 1 < W) gb[it] = act_ld4(rowp + (int64_t)(gx + 1) * g_ld);        \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)

    if (p_lo < p_hi) WB_LOAD(p_lo);
    for (int pidx = p_lo; pidx < p_hi; ++pidx) {
        __syncthreads();  // previous patch's fragment reads are done
        // ---- registers -> LDS: fused pre-norm, hi/lo split, transpose (two x-neighbours per 32-bit store) ----
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int item = tid + it * 256;
            const int rp = item >> 3;
            if (rp < ROWS * XPAIRS) {
                const int row = rp / XPAIRS, pr = rp % XPAIRS;
                float a[4] = {xa[it].x, xa[it].y, xa[it].z, xa[it].w}, b[4] = {xb[it].x, xb[it].y, xb[it].z, xb[it].w};
                const float s4[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, f4[4] = {sf4.x, sf4.y, sf4.z, sf4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float va = (inbA & (1u << it)) ? fmaf(a[c], s4[c], f4[c]) : 0.f;
                    const float vb = (inbB & (1u << it)) ? fmaf(b[c], s4[c], f4[c]) : 0.f;
                    unsigned hi, lo;
                    split2(va, vb, hi, lo);
                    const int off = (xcq * 4 + c) * CIS + row * 32 + pr * 4;
                    *reinterpret_cast<unsigned*>(Xh + off) = hi;
                    *reinterpret_cast<unsigned*>(Xl + off) = lo;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int item = tid + it * 256;
            const int cq = item % (GC / 4), rp = item / (GC / 4);
            const int prow = rp >> 2, pr = rp & 3;
            const float a[4] = {ga[it].x, ga[it].y, ga[it].z, ga[it].w}, b[4] = {gb[it].x, gb[it].y, gb[it].z, gb[it].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hi, lo;
                split2(a[c], b[c], hi, lo);
                const int off = (cq * 4 + c) * WB_GS + (prow * 8 + pr * 2) * 2;
                *reinterpret_cast<unsigned*>(Gh + off) = hi;
                *reinterpret_cast<unsigned*>(Gl + off) = lo;
                dbacc[it][c] += a[c] + b[c];
            }
        }
        __syncthreads();
        if (pidx + 1 < p_hi) WB_LOAD(pidx + 1);  // in flight during the MFMAs below

        // ---- MFMA.  Branch-free: every wave runs MAXU units (a wave whose last unit does not exist recomputes
        // unit 0 and drops it in the epilogue), so hipcc can hoist the ds_reads of the next (unit, slab) above the
        // MFMAs of the current one.  Per (unit, slab): 6 LDS reads + 8 v_alignbyte feed 9 MFMAs. ----
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
#pragma unroll 2
            for (int sl = 0; sl < SPU; ++sl) {
                const int prow = 2 * (uhalf[i] * SPU + sl) + kh;  // this lane half's patch row
                const int pz = prow / PTY, py = prow % PTY;
                const int goff = (uct[i] * 32 + r) * WB_GS + prow * 16;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Gh + goff));
                const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Gl + goff));
                const int xoff = r * CIS + (pz * HY + py) * 32 + urow[i];
                const uint4 wh = *reinterpret_cast<const uint4*>(Xh + xoff);
                const uint4 wl = *reinterpret_cast<const uint4*>(Xl + xoff);
                unsigned wh4 = 0, wl4 = 0;
                if (KW == 3) {
                    wh4 = *reinterpret_cast<const unsigned*>(Xh + xoff + 16);
                    wl4 = *reinterpret_cast<const unsigned*>(Xl + xoff + 16);
                }
#pragma unroll
                for (int tx = 0; tx < KW; ++tx) {
                    uint4 fh, fl;
                    if (tx == 0) {
                        fh = wh;
                        fl = wl;
                    } else if (tx == 1) {
                        fh = make_uint4(__builtin_amdgcn_alignbyte(wh.y, wh.x, 2), __builtin_amdgcn_alignbyte(wh.z, wh.y, 2),
                                        __builtin_amdgcn_alignbyte(wh.w, wh.z, 2), __builtin_amdgcn_alignbyte(wh4, wh.w, 2));
                        fl = make_uint4(__builtin_amdgcn_alignbyte(wl.y, wl.x, 2), __builtin_amdgcn_alignbyte(wl.z, wl.y, 2),
                                        __builtin_amdgcn_alignbyte(wl.w, wl.z, 2), __builtin_amdgcn_alignbyte(wl4, wl.w, 2));
                    } else {
                        fh = make_uint4(wh.y, wh.z, wh.w, wh4);
                        fl = make_uint4(wl.y, wl.z, wl.w, wl4);
                    }
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, fh), al = __builtin_bit_cast(bf16x8, fl);
                    acc[i][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][tx], 0, 0, 0);
                    acc[i][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][tx], 0, 0, 0);
                    acc[i][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][tx], 0, 0, 0);
                }
            }
        }
    }
#undef WB_LOAD

    // ---- bias-gradient partial of this workgroup ----
    if (do_db) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(ldsb);  // [256 / (GC/4)][GC]
#pragma unroll
        for (int it = 0; it < GIT; ++it) {
            const int item = tid + it * 256;
            const int cq = item % (GC / 4), rp = item / (GC / 4);
            // contributions of the GIT items of one thread share cq; fold them into slot rp % (256/(GC/4))
            if (it == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < GIT; ++j) a += dbacc[j][c];
                    red[(rp % (256 / (GC / 4))) * GC + cq * 4 + c] = a;
                }
            }
        }
        __syncthreads();
        if (tid < nco_here * 32) {
            float a = 0.f;
            for (int rr = 0; rr < 256 / (GC / 4); ++rr) a += red[rr * GC + tid];
            dbpart[(int64_t)sp * Cout + cog * NCO * 32 + tid] = a;
        }
    }
    // ---- partial slab: D[row = ci][col = co] ----
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = wv + 4 * i;
        if (u < U) {
            const int rg = u % NRG, ct = (u / NRG) % nco_here, half = u / (NRG * nco_here);
#pragma unroll
            for (int tx = 0; tx < ACW; ++tx) {
                const int tap = rg * KW + tx;
                float* dst = part + ((((int64_t)sp * KS2 + half) * NT + tap) * Cin + cit * 32) * Cout +
                             (cog * NCO + ct) * 32 + r;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                    dst[(int64_t)row * Cout] = acc[i][tx][reg];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 3x3x3 weight gradient, z-sliding variant (the default for D >= 16).  Ablations of k_conv_wgrad_bf16x3 (2x128^3,
// 32->32: 1.46 ms): MFMA phase alone 0.74 ms, staging alone (global loads, pre-norm, hi/lo split, transposed LDS
// stores) 0.96 ms, and the two do not overlap (one workgroup per CU, every wave does both).  A 2x8x8 patch re-stages
// 4x10x10 halo voxels per 128 outputs (3.1x); here a workgroup walks a z-COLUMN of 8x8 planes and keeps a ring of 4
// halo planes in LDS, so each step stages ONE new 10x10 plane (1.56x) plus the 8x8 g plane, and
//   * 8 waves, 18 work units (row group x {k-half | Cout tile}) dealt 3/3/2/2/2/2/2/2: a wave that finishes its
//     MFMAs converts + stores its share of the NEXT planes into the free ring slot / g buffer right away and
//     issues the loads after that -- staging of step t+1 overlaps the MFMAs of slower waves of step t;
//   * ONE barrier per plane; a column accumulates over up to D planes, so far fewer partial slabs are written.
// Ring slot of plane z is (z + 4) & 3; planes -1 and D are stored as zeros (padding applies after the pre-norm).
// ---------------------------------------------------------------------------
#ifndef TEM_ZS_BALANCE
#define TEM_ZS_BALANCE 1   // z-sliding wgrad: units 16, 17 cut in halves over waves 0..3 (4.5 units per SIMD instead of 5/5/4/4)
#endif
#define ZS_NPL 4
#ifndef TEM_ZS_LD_AUX
#define TEM_ZS_LD_AUX 0   // cache policy of the x / g plane loads (2 = nt: measured +0.03 ms / step, not used)
#endif
#ifndef TEM_ZS_L2HIT
#define TEM_ZS_L2HIT 0   // harness only: loads of all planes hit the first four (L2-resident) planes; wrong results
#endif
#define ZS_PLB 320                       // bytes per ci per plane: 10 halo rows x 32 B (16 bf16 slots, 10 used)
#define ZS_CIS (ZS_NPL * ZS_PLB + 16)    // bytes per ci (padded like the patch kernel: conflict-free b128 reads)
#define ZS_GS 144                        // bytes per co row of one g plane: 64 bf16 + 16 pad

typedef __amdgpu_buffer_rsrc_t zs_rsrc_t;
typedef float zs_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int zs_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ zs_rsrc_t zs_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 zs_load4(zs_rsrc_t r, unsigned voff, unsigned soff) {
    const zs_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, TEM_ZS_LD_AUX);
    const zs_f4 f = __builtin_bit_cast(zs_f4, v);   // whole-vector cast (element-wise bit casts get the load narrowed, conv_pp.hip)
    return make_float4(f.x, f.y, f.z, f.w);
}

// ONE: 0 bf16x3 (hi + lo planes of both operands, 3 MFMAs per product), 1 one fp16 term, 2 one bf16 term (the mixed
// precision modes), 3 "fp16 2x1": x^ = hi + lo in two fp16 terms (x^ is a normalised activation: |lo| <= 2^-12 |x^| stays
// accurate to 2^-25 absolute without a scaled lo plane), g ONE fp16 term after the power-of-two prescale that puts
// *g_amax into [2^14, 2^15) -- two MFMAs per product (hi * g, lo * g), the result multiplied by the inverse power.
template <int NCO, int ONE = 0>
__global__ __launch_bounds__(512, 1) void k_conv_wgrad_zs(const float* __restrict__ x, int64_t x_ld,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ g,
                                                          int64_t g_ld, float* __restrict__ part,
                                                          float* __restrict__ dbpart, int N, int D, int H, int W,
                                                          int Cin, int Cout, int T, int nY, int nX, int zsegs,
                                                          int S, int ncz, unsigned* __restrict__ gmax,
                                                          const unsigned* __restrict__ g_amax) {
    constexpr int NT = 27, NRG = 9, KW = 3;
    constexpr int KS2 = (NCO == 1) ? 2 : 1;
    constexpr int GC = 32 * NCO;
    constexpr bool H21 = ONE == 3;        // fp16 2x1
    float psc = 1.f, pinv = 1.f;
    if (H21) {
        // (no MODE.FP16_OVFL: see k_conv_wgrad_tr -- a NaN / inf in g or x^ must reach dw)
        const int e = (int)((*g_amax >> 23) & 0xffu);                       // biased exponent of max |g| (0: all zeros)
        const int k = e == 0 ? 0 : min(max(141 - e, -100), 100);            // max |g| * 2^k in [2^14, 2^15)
        psc = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
        pinv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    }
    constexpr int XPL = 32 * ZS_CIS;       // bytes per (hi|lo) plane set of Xt
    constexpr int GPL = GC * ZS_GS;        // bytes per (hi|lo) g plane
    constexpr int MAXU = 3;                // 18 units over 8 waves
    constexpr int SPU = (NCO == 1) ? 2 : 4;  // k-slabs per unit and plane
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* Xh = ldsb;
    unsigned char* Xl = ldsb + XPL;
    unsigned char* Gb = ldsb + 2 * XPL;    // [buffer 2][hi|lo][GC][ZS_GS]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const int bid0 = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid0 % T;
    const int sp = bid0 / T;               // partial-slab index: this workgroup walks column segments sp, sp+S, ...
    const int ncit = Cin >> 5;
    const int cit = tile % ncit, cog = tile / ncit;

    floatx16 acc[MAXU][KW];
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
#pragma unroll
        for (int t = 0; t < KW; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][t][k] = 0.f;
    // unit u = wv + 8 i: row group rg = u % 9, v = u / 9 (NCO == 1: k-half, NCO == 2: Cout tile)
    int urg[MAXU], uv[MAXU];
    bool uok[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = wv + 8 * i;
        uok[i] = u < 18;
        urg[i] = (uok[i] ? u : 0) % NRG;
        uv[i] = (uok[i] ? u : 0) / NRG;
    }
    // 18 units over 8 waves would put 5 / 5 / 4 / 4 on the four SIMDs (waves w and w+4 share one).  The two units left
    // after two rounds (16, 17) are therefore cut in halves along their k-slabs: waves 0 and 2 each take half the slabs of
    // unit 16, waves 1 and 3 of unit 17 -- 4.5 units per SIMD -- and the partial accumulators of waves 2, 3 are added to
    // those of waves 0, 1 through LDS once, before the slabs are written.
    const int sh2 = TEM_ZS_BALANCE ? (wv >> 1) & 1 : 0;   // which half of the third unit's slabs
    if (TEM_ZS_BALANCE) {
        const int u = 16 + (wv & 1);
        uok[MAXU - 1] = wv < 4;
        urg[MAXU - 1] = u % NRG;
        uv[MAXU - 1] = u / NRG;
    }

    // staging items of this thread
    const bool xit = tid < 400;                    // (halo row 10, x pair 5, channel quad 8)
    const int xcq = tid & 7, xrp = tid >> 3, xrow = xrp / 5, xpr = xrp % 5;
    const bool git = tid < 256 * NCO;              // (patch row 8, x pair 4, channel quad 8 NCO)
    const int gcq = tid % (8 * NCO), grp = tid / (8 * NCO), gprow = grp >> 2, gpr = grp & 3;
    float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa, ga = xa, gb = xa;
    bool inA = false, inB = false;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_db = (dbpart != nullptr) && (cit == 0);
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    float gmx = 0.f;   // largest |g| this thread staged (gmax != nullptr: tem_conv3d_wgrad_gmax)

    // The accumulators live across column segments: a workgroup writes ONE partial slab however many columns it
    // walks (a slab is 8 % of a column's own traffic, and the merge kernel reads every slab back).  The pipeline
    // re-primes per segment; the barrier that ends a segment's last plane orders its LDS reads before the next
    // segment's first stores.
    // slabs are grouped per SAMPLE (S = Ss slabs for each of the N samples; `ncz` = column segments of one sample), so
    // that the merge kernel can also form per-sample weight gradients (tem_conv3d_wgrad_sums)
    const int n = sp / S;
    for (int cz = sp % S; cz < ncz; cz += S) {
    const int zseg = cz % zsegs;
    const int col = cz / zsegs;
    const int ptx = col % nX;
    const int pty = col / nX;
    const int y0 = pty * 8, x0 = ptx * 8;
    const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
    if (scale && xit) {
        sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + cit * 32 + xcq * 4);
        sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + cit * 32 + xcq * 4);
    }
    // Everything about a thread's two x voxels and two g voxels except the plane is fixed for the whole column: in-range
    // flags and BYTE offsets inside a z-plane are computed here once; a load in the plane loop is then
    // "buffer_load_dwordx4 v, voff, s[rsrc], s_plane" (the 64-bit address products and the bounds tests per load made the
    // four loads of a plane cost ~1600 cycles of issue, half an MFMA phase; needs H*W*ld*4 < 2^31, checked by the host side)
    bool okxa = false, okxb = false, okga = false, okgb = false;
    unsigned offx = 0, offg = 0;
    {
        const int gy = y0 + xrow - 1, gx = x0 + 2 * xpr - 1;
        const bool rowok = xit && gy >= 0 && gy < H;
        okxa = rowok && gx >= 0 && gx < W;
        okxb = rowok && gx + 1 >= 0 && gx + 1 < W;
        offx = (unsigned)(((gy < 0 ? 0 : gy) * W + (gx < 0 ? 0 : gx)) * (int)x_ld + cit * 32 + xcq * 4) * 4u;  // of voxel gx (or 0)
        const int hy = y0 + gprow, hx = x0 + 2 * gpr;
        const bool grow = git && hy < H && gcq * 4 < Cout - cog * GC;
        okga = grow && hx < W;
        okgb = grow && hx + 1 < W;
        offg = (unsigned)(((hy < H ? hy : 0) * W + (hx < W ? hx : 0)) * (int)g_ld + cog * GC + gcq * 4) * 4u;
    }
    // first voxel of the pair out of range on the left (gx == -1): the second one sits at offset 0 of the row
    const unsigned offxb = (x0 + 2 * xpr - 1 < 0) ? offx : offx + (unsigned)x_ld * 4u;
    const unsigned offgb = offg + (unsigned)g_ld * 4u;
    const float* const xn = x + (int64_t)n * D * H * W * x_ld;   // the resource of a load starts at its z-plane (scalar
    const float* const gn = g + (int64_t)n * D * H * W * g_ld;   // 64-bit add per plane): offsets stay inside one plane
    const int64_t xplane = (int64_t)H * W * x_ld, gplane = (int64_t)H * W * g_ld;
    xa = make_float4(0.f, 0.f, 0.f, 0.f);
    xb = xa;
    ga = xa;
    gb = xa;
    inA = inB = false;
    // iteration t: MFMA over plane t (if t >= za); store the pending registers (x plane t+2, g plane t+1); load the
    // next pending set (x plane t+3, g plane t+2); barrier.
    for (int t = za - 4; t < zb; ++t) {
        if (t >= za) {
            const unsigned char* Gh = Gb + (t & 1) * 2 * GPL;
            const unsigned char* Gl = Gh + GPL;
            int slot[3];
#pragma unroll
            for (int tz = 0; tz < 3; ++tz) slot[tz] = ((t + tz - 1 + 4) & 3) * ZS_PLB;
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                if (i == MAXU - 1 && !uok[i]) break;  // wave-uniform: waves 2..7 have two units
                const int tz = urg[i] / 3, ty = urg[i] % 3;
                const int ct = (NCO == 2) ? uv[i] : 0;
                const int sl0 = (NCO == 1) ? uv[i] * SPU : 0;
                const int xbase = r * ZS_CIS + (tz == 0 ? slot[0] : (tz == 1 ? slot[1] : slot[2])) + ty * 32;
#pragma unroll
                for (int sl = 0; sl < SPU; ++sl) {
                    if (TEM_ZS_BALANCE && i == MAXU - 1 && (sl / (SPU / 2)) != sh2) continue;   // the other wave's slabs
                    const int prow = 2 * (sl0 + sl) + kh;  // this lane half's patch row (0..7)
                    const int goff = (ct * 32 + r) * ZS_GS + prow * 16;
                    const int xoff = xbase + prow * 32;
                    if constexpr (H21) {
                        // x^ two fp16 terms, g one: lo * g first (small products enter the accumulator before the large ones)
                        const uint4 bh = *reinterpret_cast<const uint4*>(Gh + goff);
                        const uint4 wh = *reinterpret_cast<const uint4*>(Xh + xoff);
                        const uint4 wl = *reinterpret_cast<const uint4*>(Xl + xoff);
                        const unsigned wh4 = *reinterpret_cast<const unsigned*>(Xh + xoff + 16);
                        const unsigned wl4 = *reinterpret_cast<const unsigned*>(Xl + xoff + 16);
                        uint4 fh[KW], fl[KW];
                        fh[0] = wh;
                        fl[0] = wl;
                        fh[1] = make_uint4(__builtin_amdgcn_alignbyte(wh.y, wh.x, 2), __builtin_amdgcn_alignbyte(wh.z, wh.y, 2),
                                           __builtin_amdgcn_alignbyte(wh.w, wh.z, 2), __builtin_amdgcn_alignbyte(wh4, wh.w, 2));
                        fl[1] = make_uint4(__builtin_amdgcn_alignbyte(wl.y, wl.x, 2), __builtin_amdgcn_alignbyte(wl.z, wl.y, 2),
                                           __builtin_amdgcn_alignbyte(wl.w, wl.z, 2), __builtin_amdgcn_alignbyte(wl4, wl.w, 2));
                        fh[2] = make_uint4(wh.y, wh.z, wh.w, wh4);
                        fl[2] = make_uint4(wl.y, wl.z, wl.w, wl4);
#pragma unroll
                        for (int tx = 0; tx < KW; ++tx) acc[i][tx] = mfma16<true>(fl[tx], bh, acc[i][tx]);
#pragma unroll
                        for (int tx = 0; tx < KW; ++tx) acc[i][tx] = mfma16<true>(fh[tx], bh, acc[i][tx]);
                        continue;
                    } else if constexpr (ONE != 0) {
                        // single fp16 product (the autocast-equivalent mode): hi planes only
                        const uint4 bh = *reinterpret_cast<const uint4*>(Gh + goff);
                        const uint4 wh = *reinterpret_cast<const uint4*>(Xh + xoff);
                        const unsigned wh4 = *reinterpret_cast<const unsigned*>(Xh + xoff + 16);
                        const uint4 f1 = make_uint4(__builtin_amdgcn_alignbyte(wh.y, wh.x, 2), __builtin_amdgcn_alignbyte(wh.z, wh.y, 2),
                                                    __builtin_amdgcn_alignbyte(wh.w, wh.z, 2), __builtin_amdgcn_alignbyte(wh4, wh.w, 2));
                        const uint4 f2 = make_uint4(wh.y, wh.z, wh.w, wh4);
                        acc[i][0] = mfma16<ONE == 1>(wh, bh, acc[i][0]);
                        acc[i][1] = mfma16<ONE == 1>(f1, bh, acc[i][1]);
                        acc[i][2] = mfma16<ONE == 1>(f2, bh, acc[i][2]);
                        continue;
                    }
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Gh + goff));
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Gl + goff));
                    const uint4 wh = *reinterpret_cast<const uint4*>(Xh + xoff);
                    const uint4 wl = *reinterpret_cast<const uint4*>(Xl + xoff);
                    const unsigned wh4 = *reinterpret_cast<const unsigned*>(Xh + xoff + 16);
                    const unsigned wl4 = *reinterpret_cast<const unsigned*>(Xl + xoff + 16);
                    // the three tx windows first, then product-major issue: consecutive MFMAs hit different accumulators
                    uint4 fh[KW], fl[KW];
                    fh[0] = wh;
                    fl[0] = wl;
                    fh[1] = make_uint4(__builtin_amdgcn_alignbyte(wh.y, wh.x, 2), __builtin_amdgcn_alignbyte(wh.z, wh.y, 2),
                                       __builtin_amdgcn_alignbyte(wh.w, wh.z, 2), __builtin_amdgcn_alignbyte(wh4, wh.w, 2));
                    fl[1] = make_uint4(__builtin_amdgcn_alignbyte(wl.y, wl.x, 2), __builtin_amdgcn_alignbyte(wl.z, wl.y, 2),
                                       __builtin_amdgcn_alignbyte(wl.w, wl.z, 2), __builtin_amdgcn_alignbyte(wl4, wl.w, 2));
                    fh[2] = make_uint4(wh.y, wh.z, wh.w, wh4);
                    fl[2] = make_uint4(wl.y, wl.z, wl.w, wl4);
#pragma unroll
                    for (int tx = 0; tx < KW; ++tx)
                        acc[i][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fl[tx]), bh, acc[i][tx], 0, 0, 0);
#pragma unroll
                    for (int tx = 0; tx < KW; ++tx)
                        acc[i][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh[tx]), bl, acc[i][tx], 0, 0, 0);
#pragma unroll
                    for (int tx = 0; tx < KW; ++tx)
                        acc[i][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh[tx]), bh, acc[i][tx], 0, 0, 0);
                }
            }
        }
        // ---- pending registers -> LDS: x plane t+2 into its ring slot, g plane t+1 into buffer (t+1)&1 ----
        if (t >= za - 3 && xit) {
            const int sl = ((t + 2 + 4) & 3) * ZS_PLB;
            const float a[4] = {xa.x, xa.y, xa.z, xa.w}, b[4] = {xb.x, xb.y, xb.z, xb.w};
            const float s4[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, f4[4] = {sf4.x, sf4.y, sf4.z, sf4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float va = inA ? fmaf(a[c], s4[c], f4[c]) : 0.f;
                const float vb = inB ? fmaf(b[c], s4[c], f4[c]) : 0.f;
                const int off = (xcq * 4 + c) * ZS_CIS + sl + xrow * 32 + xpr * 4;
                if constexpr (H21) {
                    const unsigned hi = pk16<true>(va, vb);
                    *reinterpret_cast<unsigned*>(Xh + off) = hi;
                    *reinterpret_cast<unsigned*>(Xl + off) = pk16<true>(va - lo16<true>(hi), vb - hi16<true>(hi));
                } else if constexpr (ONE != 0) {
                    *reinterpret_cast<unsigned*>(Xh + off) = pk16<ONE == 1>(va, vb);
                } else {
                    unsigned hi, lo;
                    split2(va, vb, hi, lo);
                    *reinterpret_cast<unsigned*>(Xh + off) = hi;
                    *reinterpret_cast<unsigned*>(Xl + off) = lo;
                }
            }
        }
        if (t + 1 >= za && t + 1 < zb && git) {
            unsigned char* Gh = Gb + ((t + 1) & 1) * 2 * GPL;
            unsigned char* Gl = Gh + GPL;
            const float a[4] = {ga.x, ga.y, ga.z, ga.w}, b[4] = {gb.x, gb.y, gb.z, gb.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int off = (gcq * 4 + c) * ZS_GS + (gprow * 8 + gpr * 2) * 2;
                if constexpr (H21) {
                    *reinterpret_cast<unsigned*>(Gh + off) = pk16<true>(a[c] * psc, b[c] * psc);
                } else if constexpr (ONE != 0) {
                    *reinterpret_cast<unsigned*>(Gh + off) = pk16<ONE == 1>(a[c], b[c]);
                } else {
                    unsigned hi, lo;
                    split2(a[c], b[c], hi, lo);
                    *reinterpret_cast<unsigned*>(Gh + off) = hi;
                    *reinterpret_cast<unsigned*>(Gl + off) = lo;
                }
                dbacc[c] += a[c] + b[c];
            }
            if (gmax) {   // grid-uniform; two v_max3 per plane
                gmx = __builtin_fmaxf(gmx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1])),
                                                           __builtin_fmaxf(__builtin_fabsf(a[2]), __builtin_fabsf(a[3]))));
                gmx = __builtin_fmaxf(gmx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b[0]), __builtin_fabsf(b[1])),
                                                           __builtin_fmaxf(__builtin_fabsf(b[2]), __builtin_fabsf(b[3]))));
            }
        }
        // ---- loads for the next pending set: x plane t+3 (planes za-1 .. zb), g plane t+2 (za .. zb-1) ----
        {
            const int zx = t + 3;
            const bool zxok = zx >= za - 1 && zx <= zb && zx >= 0 && zx < D;   // wave-uniform
            inA = zxok && okxa;
            inB = zxok && okxb;
            xa = make_float4(0.f, 0.f, 0.f, 0.f);
            xb = xa;
            if (zxok) {
                const zs_rsrc_t rsx = zs_rsrc(xn + (TEM_ZS_L2HIT ? (zx & 3) : zx) * xplane);   // (harness experiment: every column reads the first planes)
                if (okxa) xa = zs_load4(rsx, offx, 0);
                if (okxb) xb = zs_load4(rsx, offxb, 0);
            }
            const int zg = t + 2;
            ga = make_float4(0.f, 0.f, 0.f, 0.f);
            gb = ga;
            if (zg >= za && zg < zb) {
                const zs_rsrc_t rsg = zs_rsrc(gn + (TEM_ZS_L2HIT ? (zg & 3) : zg) * gplane);
                if (okga) ga = zs_load4(rsg, offg, 0);
                if (okgb) gb = zs_load4(rsg, offgb, 0);
            }
        }
        __syncthreads();
    }
    }  // column segments

    // ---- largest |g|: the bit patterns of non-negative floats order like unsigned integers, and an integer max is
    // exact and order-independent (deterministic); NaN / inf propagate as the largest patterns ----
    if (gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmx = __builtin_fmaxf(gmx, __shfl_xor(gmx, o, 64));
        if (lane == 0) atomicMax(gmax, __builtin_bit_cast(unsigned, gmx));
    }
    // ---- bias-gradient partial of this workgroup ----
    if (do_db) {
        float* red = reinterpret_cast<float*>(ldsb);  // [32][GC]
        if (git) {
#pragma unroll
            for (int c = 0; c < 4; ++c) red[grp * GC + gcq * 4 + c] = dbacc[c];
        }
        __syncthreads();
        if (tid < GC && cog * GC + tid < Cout) {
            float a = 0.f;
            for (int rr = 0; rr < 32; ++rr) a += red[rr * GC + tid];
            dbpart[(int64_t)sp * Cout + cog * GC + tid] = a;
        }
    }
    // ---- the halves of units 16 / 17: waves 2, 3 hand their partial sums to waves 0, 1 ----
    if (TEM_ZS_BALANCE) {
        __syncthreads();
        float* xch = reinterpret_cast<float*>(ldsb);   // [2][KW][16][64]
        if (wv == 2 || wv == 3) {
#pragma unroll
            for (int tx = 0; tx < KW; ++tx)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) xch[(((wv - 2) * KW + tx) * 16 + reg) * 64 + lane] = acc[MAXU - 1][tx][reg];
        }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int tx = 0; tx < KW; ++tx)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) acc[MAXU - 1][tx][reg] += xch[((wv * KW + tx) * 16 + reg) * 64 + lane];
        } else {
            uok[MAXU - 1] = false;
        }
    }
    // ---- partial slabs: D[row = ci][col = co] ----
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        if (uok[i]) {
            const int half = (NCO == 1) ? uv[i] : 0, ct = (NCO == 2) ? uv[i] : 0;
            if ((cog * NCO + ct) * 32 < Cout) {
#pragma unroll
                for (int tx = 0; tx < KW; ++tx) {
                    const int tap = urg[i] * KW + tx;
                    float* dst = part + ((((int64_t)sp * KS2 + half) * NT + tap) * Cin + cit * 32) * Cout +
                                 (cog * NCO + ct) * 32 + r;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                        dst[(int64_t)row * Cout] = H21 ? acc[i][tx][reg] * pinv : acc[i][tx][reg];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 3x3x3 weight gradient, z-sliding with a STAGING TEAM (round 3; the default for D >= 16).
// k_conv_wgrad_zs runs "MFMA phase, staging, barrier" per plane on all eight waves: the matrix pipe idles while the
// planes of the next step are converted (0.54 busy).  What the forward / data-gradient kernels showed (conv_zr.hip) is
// that ONE wave per SIMD keeps the pipe full when it does nothing else, and that the wave beside it has ~1500 issue
// slots per 10 k cycles for everything else.  So here the roles are fixed:
//   * waves 0..3 (one per SIMD) only multiply: they own the 27 x (32 x 32) accumulators of ONE (Cin tile, Cout tile)
//     pair -- row groups (tz, ty) w and w + 4 with their three tx taps each, plus tap tx = w of row group 8 for w < 3:
//     7 / 7 / 7 / 6 taps = 112 accumulator registers, 84 / 84 / 84 / 72 MFMAs per plane; no k-halves, nothing to merge;
//   * waves 4..7 only stage: the next x plane into the free ring slot, the next g plane into the free buffer (pre-norm,
//     hi/lo split, transposed LDS stores), the loads after that, the bias-gradient sums and max |g|;
//   * one barrier per plane, the ring / buffer discipline of k_conv_wgrad_zs unchanged.
// A layer with two or more Cout tiles becomes that many workgroup columns: each stages x again, which costs HBM/L2 reads
// but no time (the staging team has the slack).  Same LDS layout as k_conv_wgrad_zs<1>, same partial-slab format
// with KS2 = 1.
// ---------------------------------------------------------------------------
template <int ONE>
__global__ __launch_bounds__(512, 2) void k_conv_wgrad_zt(const float* __restrict__ x, int64_t x_ld,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ g,
                                                          int64_t g_ld, float* __restrict__ part,
                                                          float* __restrict__ dbpart, int N, int D, int H, int W,
                                                          int Cin, int Cout, int T, int nY, int nX, int zsegs,
                                                          int S, int ncz, unsigned* __restrict__ gmax,
                                                          const unsigned* __restrict__ /* g_amax: k_conv_wgrad_zs<., 3> only */) {
    constexpr int NT = 27, KW = 3, NA = 7;   // accumulators per multiplying wave
    constexpr int GC = 32;
    constexpr int XPL = 32 * ZS_CIS;       // bytes per (hi|lo) plane set of Xt
    constexpr int GPL = GC * ZS_GS;        // bytes per (hi|lo) g plane
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* Xh = ldsb;
    unsigned char* Xl = ldsb + XPL;
    unsigned char* Gb = ldsb + 2 * XPL;    // [buffer 2][hi|lo][GC][ZS_GS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mteam = wv < 4;             // multiplying team / staging team
    const int kh = lane >> 5, r = lane & 31;
    const int bid0 = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid0 % T;
    const int sp = bid0 / T;               // partial-slab index: this workgroup walks column segments sp, sp+S, ...
    const int ncit = Cin >> 5;
    const int cit = tile % ncit, cog = tile / ncit;

    // multiplying waves: accumulator j < 6 is tap (row group wv + 4 (j / 3), tx = j % 3); j == 6 is (row group 8, tx = wv)
    floatx16 acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;

    const int tl = tid & 255;
    const bool git = !mteam;
    const int gcq = tl & 7, grp = tl >> 3, gprow = grp >> 2, gpr = grp & 3;
    const bool do_db = (dbpart != nullptr) && (cit == 0);
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    float gmx = 0.f;
    const int n = sp / S;
    if (mteam) {
        // ---------------- multiplying team ----------------
        for (int cz = sp % S; cz < ncz; cz += S) {
        const int zseg = cz % zsegs;
        const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
        // the staging team primes the ring for four barriers; then one plane per barrier (no `if (t >= za)` inside the
        // loop: with it hipcc kept two register tuples per accumulator across the back edge)
        for (int t = za - 4; t < za; ++t) __syncthreads();
#pragma unroll 1
        for (int t = za; t < zb; ++t) {
            int wvl = wv;
            asm volatile("" : "+s"(wvl));   // keeps the wave-id tests inside ONE copy of the loop body
        {
            const unsigned char* Gh = Gb + (t & 1) * 2 * GPL;
            const unsigned char* Gl = Gh + GPL;
            const int gbase = r * ZS_GS + kh * 16;   // g fragment of k-slab sl: + 32 sl (re-read per row group: LDS has room,
                                                     // 32 registers for all four slabs do not)
            auto rowgroup = [&](int rg, auto ntx_tag, int txs, int j0) {
                // NTX == 3: all three tx windows of row group rg into acc[j0 .. j0+2]; NTX == 1: window txs into acc[j0]
                constexpr int NTX = decltype(ntx_tag)::value;
                const int tz = rg / 3, ty = rg % 3;
                const int xbase = r * ZS_CIS + ((t + tz - 1 + 4) & 3) * ZS_PLB + ty * 32;   // ring slot of plane t + tz - 1 (scalar)
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const int xoff = xbase + (2 * sl + kh) * 32;
                    const uint4 bhs = *reinterpret_cast<const uint4*>(Gh + gbase + 32 * sl);
                    uint4 bls = bhs;
                    if (ONE == 0) bls = *reinterpret_cast<const uint4*>(Gl + gbase + 32 * sl);
                    const uint4 wh = *reinterpret_cast<const uint4*>(Xh + xoff);
                    const unsigned wh4 = *reinterpret_cast<const unsigned*>(Xh + xoff + 16);
                    uint4 wl = wh;
                    unsigned wl4 = wh4;
                    if (ONE == 0) {
                        wl = *reinterpret_cast<const uint4*>(Xl + xoff);
                        wl4 = *reinterpret_cast<const unsigned*>(Xl + xoff + 16);
                    }
                    uint4 fh[NTX], fl[NTX];
                    if (NTX == 3) {
                        fh[0] = wh;
                        fl[0] = wl;
                        fh[NTX > 1 ? 1 : 0] = make_uint4(__builtin_amdgcn_alignbyte(wh.y, wh.x, 2), __builtin_amdgcn_alignbyte(wh.z, wh.y, 2),
                                                         __builtin_amdgcn_alignbyte(wh.w, wh.z, 2), __builtin_amdgcn_alignbyte(wh4, wh.w, 2));
                        fl[NTX > 1 ? 1 : 0] = make_uint4(__builtin_amdgcn_alignbyte(wl.y, wl.x, 2), __builtin_amdgcn_alignbyte(wl.z, wl.y, 2),
                                                         __builtin_amdgcn_alignbyte(wl.w, wl.z, 2), __builtin_amdgcn_alignbyte(wl4, wl.w, 2));
                        fh[NTX > 2 ? 2 : 0] = make_uint4(wh.y, wh.z, wh.w, wh4);
                        fl[NTX > 2 ? 2 : 0] = make_uint4(wl.y, wl.z, wl.w, wl4);
                    } else {   // one window, txs wave-uniform: v_alignbyte shifts by 0..3 bytes only, window 2 starts one dword up
                        const unsigned sh = txs == 1 ? 2u : 0u;
                        const bool up = txs == 2;
                        const unsigned h0 = up ? wh.y : wh.x, h1 = up ? wh.z : wh.y, h2 = up ? wh.w : wh.z, h3 = up ? wh4 : wh.w;
                        const unsigned l0 = up ? wl.y : wl.x, l1 = up ? wl.z : wl.y, l2 = up ? wl.w : wl.z, l3 = up ? wl4 : wl.w;
                        fh[0] = make_uint4(__builtin_amdgcn_alignbyte(h1, h0, sh), __builtin_amdgcn_alignbyte(h2, h1, sh),
                                           __builtin_amdgcn_alignbyte(h3, h2, sh), __builtin_amdgcn_alignbyte(wh4, h3, sh));
                        fl[0] = make_uint4(__builtin_amdgcn_alignbyte(l1, l0, sh), __builtin_amdgcn_alignbyte(l2, l1, sh),
                                           __builtin_amdgcn_alignbyte(l3, l2, sh), __builtin_amdgcn_alignbyte(wl4, l3, sh));
                    }
                    if constexpr (ONE != 0) {
#pragma unroll
                        for (int tx = 0; tx < NTX; ++tx) acc[j0 + tx] = mfma16<ONE == 1>(fh[tx], bhs, acc[j0 + tx]);
                    } else {
                        const bf16x8 gh8 = __builtin_bit_cast(bf16x8, bhs), gl8 = __builtin_bit_cast(bf16x8, bls);
#pragma unroll
                        for (int tx = 0; tx < NTX; ++tx)
                            acc[j0 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fl[tx]), gh8, acc[j0 + tx], 0, 0, 0);
#pragma unroll
                        for (int tx = 0; tx < NTX; ++tx)
                            acc[j0 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh[tx]), gl8, acc[j0 + tx], 0, 0, 0);
#pragma unroll
                        for (int tx = 0; tx < NTX; ++tx)
                            acc[j0 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fh[tx]), gh8, acc[j0 + tx], 0, 0, 0);
                    }
                }
            };
            rowgroup(wvl, std::integral_constant<int, 3>{}, 0, 0);
            rowgroup(wvl + 4, std::integral_constant<int, 3>{}, 0, 3);
            // wave 3 has no seventh tap: it multiplies window 0 of row group 8 into an accumulator that is never stored (the
            // other waves need these 12 MFMA slots anyway; an `if` here made hipcc copy accumulators around the loop)
            rowgroup(8, std::integral_constant<int, 1>{}, wvl < 3 ? wvl : 0, 6);
        }
            __syncthreads();
        }
        }  // column segments
        // (written here, inside the multiplying team's branch: behind the join the staging team would have to keep 112
        // registers of zeros alive for it)
        // ---- partial slabs: D[row = ci][col = co] ----
        if (cog * 32 < Cout) {
    #pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (j == 6 && wv >= 3) break;
                const int tap = (j < 6) ? (wv + 4 * (j / 3)) * KW + (j % 3) : 8 * KW + wv;
                float* dst = part + (((int64_t)sp * NT + tap) * Cin + cit * 32) * Cout + cog * 32 + r;
    #pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                    dst[(int64_t)row * Cout] = acc[j][reg];
                }
            }
        }

    } else {
        // ---------------- staging team ----------------
    // staging items of a staging-team thread: x items tl and tl + 256 (400 of them: halo row 10, x pair 5, channel quad 8),
    // g item tl (256: patch row 8, x pair 4, channel quad 8)
    float4 xa[2], xb[2], ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
    bool inA[2] = {false, false}, inB[2] = {false, false};
    float4 sc4[2], sf4[2];
    int xcq[2], xrow[2], xpr[2];
    bool xit[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int it = tl + 256 * q;
        xit[q] = !mteam && it < 400;
        xcq[q] = it & 7;
        const int xrp = (it >> 3) % 50;
        xrow[q] = xrp / 5;
        xpr[q] = xrp % 5;
        xa[q] = xb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        sc4[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        sf4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    for (int cz = sp % S; cz < ncz; cz += S) {
    const int zseg = cz % zsegs;
    const int col = cz / zsegs;
    const int ptx = col % nX;
    const int pty = col / nX;
    const int y0 = pty * 8, x0 = ptx * 8;
    const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
    bool okxa[2] = {false, false}, okxb[2] = {false, false}, okga = false, okgb = false;
    unsigned offx[2] = {0u, 0u}, offxb[2] = {0u, 0u}, offg = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (scale && xit[q]) {
            sc4[q] = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + cit * 32 + xcq[q] * 4);
            sf4[q] = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + cit * 32 + xcq[q] * 4);
        }
        const int gy = y0 + xrow[q] - 1, gx = x0 + 2 * xpr[q] - 1;
        const bool rowok = xit[q] && gy >= 0 && gy < H;
        okxa[q] = rowok && gx >= 0 && gx < W;
        okxb[q] = rowok && gx + 1 >= 0 && gx + 1 < W;
        offx[q] = (unsigned)(((gy < 0 ? 0 : gy) * W + (gx < 0 ? 0 : gx)) * (int)x_ld + cit * 32 + xcq[q] * 4) * 4u;
        offxb[q] = (gx < 0) ? offx[q] : offx[q] + (unsigned)x_ld * 4u;
        xa[q] = xb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        inA[q] = inB[q] = false;
    }
    {
        const int hy = y0 + gprow, hx = x0 + 2 * gpr;
        const bool grow = git && hy < H && gcq * 4 < Cout - cog * GC;
        okga = grow && hx < W;
        okgb = grow && hx + 1 < W;
        offg = (unsigned)(((hy < H ? hy : 0) * W + (hx < W ? hx : 0)) * (int)g_ld + cog * GC + gcq * 4) * 4u;
    }
    const unsigned offgb = offg + (unsigned)g_ld * 4u;
    const float* const xn = x + (int64_t)n * D * H * W * x_ld;
    const float* const gn = g + (int64_t)n * D * H * W * g_ld;
    const int64_t xplane = (int64_t)H * W * x_ld, gplane = (int64_t)H * W * g_ld;
    ga = make_float4(0.f, 0.f, 0.f, 0.f);
    gb = ga;
    // iteration t: the multiplying team works on plane t (if t >= za); the staging team stores its pending registers
    // (x plane t+2, g plane t+1) and loads the next pending set (x plane t+3, g plane t+2); barrier.  Two loops with the
    // same trip count, one per role (one loop with the role test inside made hipcc unswitch and peel it into 30 copies).
#pragma unroll 1
        for (int t = za - 4; t < zb; ++t) {
        // ---- pending registers -> LDS: x plane t+2 into its ring slot, g plane t+1 into buffer (t+1)&1 ----
        if (t >= za - 3) {
            const int sl = ((t + 2 + 4) & 3) * ZS_PLB;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!xit[q]) continue;
                const float a[4] = {xa[q].x, xa[q].y, xa[q].z, xa[q].w}, b[4] = {xb[q].x, xb[q].y, xb[q].z, xb[q].w};
                const float s4[4] = {sc4[q].x, sc4[q].y, sc4[q].z, sc4[q].w}, f4[4] = {sf4[q].x, sf4[q].y, sf4[q].z, sf4[q].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float va = inA[q] ? fmaf(a[c], s4[c], f4[c]) : 0.f;
                    const float vb = inB[q] ? fmaf(b[c], s4[c], f4[c]) : 0.f;
                    const int off = (xcq[q] * 4 + c) * ZS_CIS + sl + xrow[q] * 32 + xpr[q] * 4;
                    if constexpr (ONE != 0) {
                        *reinterpret_cast<unsigned*>(Xh + off) = pk16<ONE == 1>(va, vb);
                    } else {
                        unsigned hi, lo;
                        split2(va, vb, hi, lo);
                        *reinterpret_cast<unsigned*>(Xh + off) = hi;
                        *reinterpret_cast<unsigned*>(Xl + off) = lo;
                    }
                }
            }
        }
        if (t + 1 >= za && t + 1 < zb) {
            unsigned char* Gh = Gb + ((t + 1) & 1) * 2 * GPL;
            unsigned char* Gl = Gh + GPL;
            const float a[4] = {ga.x, ga.y, ga.z, ga.w}, b[4] = {gb.x, gb.y, gb.z, gb.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int off = (gcq * 4 + c) * ZS_GS + (gprow * 8 + gpr * 2) * 2;
                if constexpr (ONE != 0) {
                    *reinterpret_cast<unsigned*>(Gh + off) = pk16<ONE == 1>(a[c], b[c]);
                } else {
                    unsigned hi, lo;
                    split2(a[c], b[c], hi, lo);
                    *reinterpret_cast<unsigned*>(Gh + off) = hi;
                    *reinterpret_cast<unsigned*>(Gl + off) = lo;
                }
                dbacc[c] += a[c] + b[c];
            }
            if (gmax) {
                gmx = __builtin_fmaxf(gmx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a[0]), __builtin_fabsf(a[1])),
                                                           __builtin_fmaxf(__builtin_fabsf(a[2]), __builtin_fabsf(a[3]))));
                gmx = __builtin_fmaxf(gmx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b[0]), __builtin_fabsf(b[1])),
                                                           __builtin_fmaxf(__builtin_fabsf(b[2]), __builtin_fabsf(b[3]))));
            }
        }
        // ---- loads for the next pending set: x plane t+3 (planes za-1 .. zb), g plane t+2 (za .. zb-1) ----
        {
            const int zx = t + 3;
            const bool zxok = zx >= za - 1 && zx <= zb && zx >= 0 && zx < D;   // wave-uniform
            const zs_rsrc_t rsx = zs_rsrc(xn + (zxok ? zx : 0) * xplane);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                inA[q] = zxok && okxa[q];
                inB[q] = zxok && okxb[q];
                xa[q] = xb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (zxok) {
                    if (okxa[q]) xa[q] = zs_load4(rsx, offx[q], 0);
                    if (okxb[q]) xb[q] = zs_load4(rsx, offxb[q], 0);
                }
            }
            const int zg = t + 2;
            ga = make_float4(0.f, 0.f, 0.f, 0.f);
            gb = ga;
            if (zg >= za && zg < zb) {
                const zs_rsrc_t rsg = zs_rsrc(gn + zg * gplane);
                if (okga) ga = zs_load4(rsg, offg, 0);
                if (okgb) gb = zs_load4(rsg, offgb, 0);
            }
        }
            __syncthreads();
        }
    }  // column segments
    }


    if (gmax && !mteam) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmx = __builtin_fmaxf(gmx, __shfl_xor(gmx, o, 64));
        if (lane == 0) atomicMax(gmax, __builtin_bit_cast(unsigned, gmx));
    }
    // ---- bias-gradient partial of this workgroup (staging team: 256 threads = 32 (row, x pair) groups x 8 quads) ----
    if (do_db) {
        float* red = reinterpret_cast<float*>(ldsb);  // [32][GC]
        if (git) {
#pragma unroll
            for (int c = 0; c < 4; ++c) red[grp * GC + gcq * 4 + c] = dbacc[c];
        }
        __syncthreads();
        if (tid < GC && cog * GC + tid < Cout) {
            float a = 0.f;
            for (int rr = 0; rr < 32; ++rr) a += red[rr * GC + tid];
            dbpart[(int64_t)sp * Cout + cog * GC + tid] = a;
        }
    }
}

#ifndef TEM_ZS_MIN_D
#define TEM_ZS_MIN_D 8   // shortest z column of the z-sliding kernels (16 until round 4: a column of 8 .. 15 planes pays six priming
#endif                   // iterations for its planes and still beats the patch kernel; k_conv_wgrad_zs is wrong below 8)
struct ZsPlan {
    bool use;
    bool teams;   // k_conv_wgrad_zt (staging team) instead of k_conv_wgrad_zs
    bool tr;      // k_conv_wgrad_tr (conv_wgrad_tr.hip: staging team + transposing LDS reads); same plan as `teams`
    int nco, ks2, T, nY, nX, zsegs, S, Ss, ncz;
};
static ZsPlan zs_plan(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    ZsPlan p;
    const int enable = (int)tem_option(TEM_OPT_WGRAD_ZS);
    p.use = enable && kd == 3 && kh == 3 && kw == 3 && D >= TEM_ZS_MIN_D;
    const int ncot = Cout / 32;
    p.teams = enable >= 2;
    p.tr = enable >= 3;
    p.nco = (ncot >= 2 && !p.teams) ? 2 : 1;
    p.ks2 = (p.nco == 1 && !p.teams) ? 2 : 1;
    p.T = (Cin / 32) * ((ncot + p.nco - 1) / p.nco);
    p.nY = (H + 7) / 8;
    p.nX = (W + 7) / 8;
    int64_t cols = (int64_t)N * p.nY * p.nX;
    int zs = 1;
    // "wgrad_cus": workgroups (= CUs, one workgroup per CU) this kernel asks for; fewer than the chip has leaves CUs to
    // HBM-bound kernels of another stream (TEM_OVERLAP_WGRAD=2)
    long long ncu = tem_option(TEM_OPT_WGRAD_CUS);
    if (ncu < 8 || ncu > 256) ncu = 256;
    while (p.T * cols * zs < ncu && D / (zs * 2) >= 8) zs *= 2;  // one workgroup per CU: fill the chip
    p.zsegs = zs;
    p.ncz = p.nY * p.nX * zs;  // column segments per sample
    // persistent over column segments: q segments per workgroup, T * S workgroups ~ one per CU; a workgroup stays
    // inside one sample (Ss slabs per sample, S = N * Ss)
    const int persist = (int)tem_option(TEM_OPT_WGRAD_ZS_PERSIST);
    const int64_t q = persist ? ((int64_t)N * p.ncz * p.T + ncu - 1) / ncu : 1;
    p.Ss = (int)((p.ncz + q - 1) / q);
    p.S = N * p.Ss;
    return p;
}

struct WbPlan {
    int nco, ks2, T, S, P, nZ, nY, nX, ptz;
};

static WbPlan wb_plan(int N, int D, int H, int W, int Cin, int Cout, int ntaps) {
    WbPlan p;
    const int ncot = Cout / 32;
    p.nco = (ntaps == 1) ? (ncot >= 4 ? 4 : (ncot >= 2 ? 2 : 1)) : (ncot >= 2 ? 2 : 1);
    p.ks2 = (ntaps > 1 && p.nco == 1) ? 2 : 1;  // single co tile: split the k-slabs between two units
    int ngroups = (ncot + p.nco - 1) / p.nco;
    p.T = (Cin / 32) * ngroups;
    p.ptz = (D == 1) ? 1 : 2;
    p.nZ = (D + p.ptz - 1) / p.ptz;
    p.nY = (H + 16 / p.ptz - 1) / (16 / p.ptz);
    p.nX = (W + WB_TX - 1) / WB_TX;
    int64_t P = (int64_t)N * p.nZ * p.nY * p.nX;
    p.P = (int)P;
    int64_t S = (768 + p.T - 1) / p.T;  // one workgroup per CU: ~3 waves of workgroups
    int64_t slab = (int64_t)ntaps * Cin * Cout * 4 * p.ks2;
    // channel-rich, spatially tiny layers (the 8^3 level: 28 MB per slab for 512 -> 512): every slab is written and read
    // back by the merge, which then dominates -- one workgroup per CU (a single wave of workgroups) instead of three
    if (slab >= (8ll << 20)) S = (256 + p.T - 1) / p.T;
    int64_t cap = (256ll << 20) / slab;
    if (cap < 1) cap = 1;
    if (S > cap) S = cap;
    if (S > P) S = P;
    if (S < 1) S = 1;
    p.S = (int)S;
    return p;
}

// Can this weight gradient also deliver the norm-backward sums (wgrad_sums.hip)?  z-sliding kernel, a few samples,
// widths whose [27][Cout] tables fit a block's LDS
int tem_conv_wgrad_sums_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    const int enable = (int)tem_option(TEM_OPT_WGRAD_SUMS);
    if (!enable || Cin % 32 || Cout % 32) return 0;
    const ZsPlan z = zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw);
    TEM_REQUIRE(!tem_call_cs.x || z.use, "tem_conv3d_wgrad_ex: a chunk stride (x_cs) needs the z-sliding kernel (3x3x3, D >= 8)");
    const int cq = Cout / 4;
    // four small launches replace one pass over gz and x: only worth it where that pass is long (>= 128 MB tensors by default)
    const int64_t min_bytes = (int64_t)tem_option(TEM_OPT_WGRAD_SUMS_MIN_MB) << 20;
    const int64_t bytes = (int64_t)N * D * H * W * Cin * 4;
    return z.use && N <= 4 && Cout <= 128 && Cin <= 256 && (cq & (cq - 1)) == 0 && H >= 3 && W >= 3 && D >= 3 &&
           bytes >= min_bytes;
}

int64_t tem_conv_wgrad_bf16x3_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    const ZsPlan z = zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw);
    if (z.use)
        return tem_align_up((int64_t)z.S * z.ks2 * 27 * Cin * Cout, 64) * 4 + tem_align_up((int64_t)z.S * Cout, 64) * 4 +
               (tem_conv_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw) ? tem_wgrad_sums_ws_floats(N, D, H, Cin, Cout) * 4 : 0) +
               256;
    WbPlan p = wb_plan(N, D, H, W, Cin, Cout, kd * kh * kw);
    return tem_align_up((int64_t)p.S * p.ks2 * kd * kh * kw * Cin * Cout, 64) * 4 + (int64_t)p.S * Cout * 4 + 256;
}

template <int KD, int KH, int KW, int NCO, int KS2, int PTZ, typename T>
static void launch_wb_tt(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g, int64_t g_ld,
                         float* part, float* dbpart, int N, int D, int H, int W, int Cin, int Cout, const WbPlan& p,
                         hipStream_t s) {
    constexpr int ROWS = (PTZ + KD - 1) * (16 / PTZ + KH - 1);
    constexpr size_t ldsbytes = 2 * (size_t)32 * (ROWS * 32 + 16) + 2 * (size_t)32 * NCO * WB_GS;
    static_assert(ldsbytes <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad_bf16x3<KD, KH, KW, NCO, KS2, PTZ, T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsbytes);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_wgrad_bf16x3<KD, KH, KW, NCO, KS2, PTZ, T>), dim3((unsigned)(p.T * p.S)), dim3(256), ldsbytes, s,
                       reinterpret_cast<const T*>(x), x_ld, scale, shift, reinterpret_cast<const T*>(g), g_ld, part, dbpart, N, D, H, W,
                       Cin, Cout, p.T, p.S, p.P, p.nZ, p.nY, p.nX);
}
template <int KD, int KH, int KW, int NCO, int KS2, int PTZ>
static void launch_wb_t(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g, int64_t g_ld,
                        float* part, float* dbpart, int N, int D, int H, int W, int Cin, int Cout, const WbPlan& p,
                        hipStream_t s) {
    TEM_ST_SWITCH(tem_call_st.x, T,
                  (launch_wb_tt<KD, KH, KW, NCO, KS2, PTZ, T>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s)));
}
template <int KD, int KH, int KW, int NCO, int KS2 = 1>
static void launch_wb(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g, int64_t g_ld,
                      float* part, float* dbpart, int N, int D, int H, int W, int Cin, int Cout, const WbPlan& p,
                      hipStream_t s) {
    if constexpr (KD == 1) {
        if (p.ptz == 1) {
            launch_wb_t<KD, KH, KW, NCO, KS2, 1>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
            return;
        }
    }
    launch_wb_t<KD, KH, KW, NCO, KS2, 2>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
}

// tem_conv3d_wgrad_gmax (conv.hip) parks its output pointer here around its call into tem_conv_wgrad_bf16x3: one more
// positional argument would have to thread through four internal signatures for the one kernel that honours it
thread_local unsigned* tem_wgrad_gmax_target = nullptr;
// tem_conv3d_wgrad_gscaled (conv.hip) parks the device word with max |g| here the same way (h16 == 3 reads it)
thread_local const unsigned* tem_wgrad_gscale_source = nullptr;
int tem_conv_wgrad_gscaled_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    const ZsPlan z = zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw);
    return Cin % 32 == 0 && Cout % 32 == 0 && z.use && (!z.teams || z.tr);
}
// exact-fp32 weight gradient on k_conv_wgrad_tr<4> (TEM_PRECISION=fp32): the layers the z-sliding plan takes, option wgrad_zs >= 3
int tem_conv_wgrad_tr_fp32_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    const ZsPlan z = zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw);
    return Cin % 32 == 0 && Cout % 32 == 0 && z.use && z.tr;
}
// the chunk-stride conditions of the launch below (x_cs != 0), as a query
int tem_conv_wgrad_cs_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int st, int64_t x_cs) {
    if (Cin % 32 || Cout % 32) return 0;
    const ZsPlan z = zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw);
    return z.use && z.tr && st != 0 && x_cs % 8 == 0;
}
int tem_conv_wgrad_gmax_ok(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    return Cin % 32 == 0 && Cout % 32 == 0 && zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw).use;
}

int tem_conv_wgrad_bf16x3(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                          int64_t g_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                          int Cin, int Cout, int kd, int kh, int kw, int sd_layout, int h16, const float* w_sd,
                          const float* gamma, const float* beta, float* norm_sums, hipStream_t s) {
    TEM_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "tem_conv3d_wgrad(bf16x3): needs Cin%%32==0 and Cout%%32==0 (got %d,%d)",
                Cin, Cout);
    const int st = tem_call_st.x;
    TEM_REQUIRE(st == tem_call_st.y, "tem_conv3d_wgrad(split-bf16): x and g must have the same storage type");
    TEM_REQUIRE(st == 0 || (st == 1 && h16 == 1) || (st == 2 && h16 == 2),
                "tem_conv3d_wgrad(split-bf16): 16-bit storage goes with the one-term mode of the same type (fp16: use_mfma 5, "
                "bf16: use_mfma 7), got storage %d", st);
    TEM_REQUIRE(x_ld % (st ? 8 : 4) == 0 && g_ld % (st ? 8 : 4) == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)g % 16 == 0),
                "tem_conv3d_wgrad(bf16x3): x / g must be 16-byte aligned with ld%%4==0 (16-bit storage: ld%%8==0)");
    TEM_REQUIRE(!scale || (((uintptr_t)scale % 16 == 0) && ((uintptr_t)shift % 16 == 0)),
                "tem_conv3d_wgrad(bf16x3): scale/shift must be 16-byte aligned");
    const int ntaps = kd * kh * kw;
    WbPlan p = wb_plan(N, D, H, W, Cin, Cout, ntaps);
    if (ws_bytes < tem_conv_wgrad_bf16x3_ws(N, D, H, W, Cin, Cout, kd, kh, kw)) {
        tem_set_error("tem_conv3d_wgrad(bf16x3): workspace too small");
        return TEM_EWS;
    }
    const ZsPlan z = zs_plan(N, D, H, W, Cin, Cout, kd, kh, kw);
    unsigned* const gmax = tem_wgrad_gmax_target;   // set by tem_conv3d_wgrad_gmax for the duration of this call
    TEM_REQUIRE(!gmax || (z.use && !h16), "tem_conv3d_wgrad_gmax: tem_conv3d_wgrad_gmax_ok() == 0 for this layer");
    const unsigned* const g_amax = tem_wgrad_gscale_source;   // set by tem_conv3d_wgrad_gscaled for the duration of this call
    TEM_REQUIRE(h16 != 4 || (z.use && z.tr), "tem_conv3d_wgrad(fp32 on k_conv_wgrad_tr): tem_conv_wgrad_tr_fp32_ok() == 0 for this layer");
    TEM_REQUIRE(h16 != 3 || (g_amax && z.use && (!z.teams || z.tr)),
                "tem_conv3d_wgrad_gscaled: tem_conv3d_wgrad_gscaled_ok() == 0 for this layer (or no g_amax)");
    if (z.use) {
        TEM_REQUIRE((int64_t)H * W * (x_ld > g_ld ? x_ld : g_ld) * 4 < (1ll << 31),
                    "tem_conv3d_wgrad(bf16x3): one z-plane of x / g must stay below 2 GiB (32-bit offsets inside a plane)");
        float* zpart = (float*)ws;
        float* zdb = db ? zpart + tem_align_up((int64_t)z.S * z.ks2 * 27 * Cin * Cout, 64) : nullptr;
        TEM_REQUIRE(!norm_sums || (db && w_sd && sd_layout && tem_conv_wgrad_sums_ok(N, D, H, W, Cin, Cout, kd, kh, kw)),
                    "tem_conv3d_wgrad_sums: this layer cannot deliver the norm sums (tem_conv3d_wgrad_sums_ok() == 0)");
        const unsigned nblk = (unsigned)((int64_t)z.T * z.S);
        // h16: 0 bf16x3 (hi + lo planes, 3 MFMAs per product), 1 one fp16 term, 2 one bf16 term (the mixed-precision modes)
        auto launch = [&](auto kern, size_t lb) {
            static std::set<const void*> sized;   // kernels whose dynamic-LDS limit was raised already
            const void* key = reinterpret_cast<const void*>(kern);
            if (!sized.count(key)) {
                (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                sized.insert(key);
            }
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lb, s, x, x_ld, scale, shift, g, g_ld, zpart, zdb, N, D, H, W, Cin,
                               Cout, z.T, z.nY, z.nX, z.zsegs, z.Ss, z.ncz, gmax, g_amax);
        };
        auto go = [&](auto k0, auto k1, auto k2, auto k3, size_t lb) {
            if (h16 == 1) launch(k1, lb);
            else if (h16 == 2) launch(k2, lb);
            else if (h16 == 3) launch(k3, lb);
            else launch(k0, lb);
        };
        TEM_REQUIRE(!st || z.tr, "tem_conv3d_wgrad: 16-bit storage needs the transposing z-sliding kernel (option wgrad_zs = 3)");
        TEM_REQUIRE(!tem_call_cs.x || (st && z.tr && tem_call_cs.x % 8 == 0),
                    "tem_conv3d_wgrad_ex: a chunk stride (x_cs) needs 16-bit tensors on the transposing z-sliding kernel, x_cs %% 8 == 0");
        if (z.tr)
            tem_conv_wgrad_tr_launch(h16, nblk, x, x_ld, scale, shift, g, g_ld, zpart, zdb, N, D, H, W, Cin, Cout, z.T, z.nY, z.nX,
                                     z.zsegs, z.Ss, z.ncz, gmax, g_amax, s);
        else if (z.teams)
            go(&k_conv_wgrad_zt<0>, &k_conv_wgrad_zt<1>, &k_conv_wgrad_zt<2>, &k_conv_wgrad_zt<0>,
               2 * (size_t)32 * ZS_CIS + 4 * (size_t)32 * ZS_GS);
        else if (z.nco == 2)
            go(&k_conv_wgrad_zs<2, 0>, &k_conv_wgrad_zs<2, 1>, &k_conv_wgrad_zs<2, 2>, &k_conv_wgrad_zs<2, 3>,
               2 * (size_t)32 * ZS_CIS + 4 * (size_t)64 * ZS_GS);
        else
            go(&k_conv_wgrad_zs<1, 0>, &k_conv_wgrad_zs<1, 1>, &k_conv_wgrad_zs<1, 2>, &k_conv_wgrad_zs<1, 3>,
               2 * (size_t)32 * ZS_CIS + 4 * (size_t)32 * ZS_GS);
        if (norm_sums) {
            float* extra = zdb + tem_align_up((int64_t)z.S * Cout, 64);
            tem_wgrad_sums_launch(zpart, z.Ss, z.ks2, zdb, g, g_ld, w_sd, gamma, beta, dw, extra, N, D, H, W, Cin, Cout,
                                  norm_sums, z.S, db, s);
        } else {
            tem_reduce_slabs_w_db(zpart, z.S * z.ks2, 27, Cin, Cout, (int64_t)27 * Cin * Cout, dw, sd_layout, zdb, z.S, db, s);
        }
        return TEM_OK;
    }
    TEM_REQUIRE(!norm_sums, "tem_conv3d_wgrad_sums: this layer cannot deliver the norm sums (tem_conv3d_wgrad_sums_ok() == 0)");
    float* part = (float*)ws;
    float* dbpart = db ? part + tem_align_up((int64_t)p.S * p.ks2 * ntaps * Cin * Cout, 64) : nullptr;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
#define WGO(KD, KH, KW)                                                                                              \
    do {                                                                                                             \
        if (p.nco == 4)                                                                                              \
            launch_wb<KD, KH, KW, 4>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);     \
        else if (p.nco == 2)                                                                                         \
            launch_wb<KD, KH, KW, 2>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);     \
        else                                                                                                         \
            launch_wb<KD, KH, KW, 1>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);     \
    } while (0)
    if (key == 7) {
        if (p.nco == 2)
            launch_wb<3, 3, 3, 2>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
        else
            launch_wb<3, 3, 3, 1, 2>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
    } else if (key == 3) {
        if (p.nco == 2)
            launch_wb<1, 3, 3, 2>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
        else
            launch_wb<1, 3, 3, 1, 2>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
    } else if (key == 0) {
        WGO(1, 1, 1);
    } else {
        tem_set_error("tem_conv3d_wgrad(bf16x3): kernel (%d,%d,%d) has no MFMA instantiation", kd, kh, kw);
        return TEM_EINVAL;
    }
#undef WGO
    const int64_t n = (int64_t)ntaps * Cin * Cout;
    tem_reduce_slabs_w_db(part, p.S * p.ks2, ntaps, Cin, Cout, n, dw, sd_layout, dbpart, p.S, db, s);
    return TEM_OK;
}
