// conv_mfma.hip -- implicit-GEMM 3-D convolution on the gfx950 matrix cores in EXACT fp32
// (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain, 157 TFLOP/s peak = the fp32 roofline of
// this path).  Replaces aten::convolution / convolution_backward behind
// nn.Conv3d in ConvBlock (reference model/unet.py:417-438).
//
// Forward / data-gradient (one kernel; dgrad = forward with the transposed, tap-flipped pack):
//   GEMM view  M = voxels, N = Cout, K = taps*Cin.
//   workgroup  = 4 waves, output patch TZxTYxTX = 256 voxels x (32*NR) output channels;
//                wave w owns 64 consecutive patch voxels (two 32-row MFMA tiles) x NR column tiles.
//   per 16-channel chunk of Cin the (TZ+2)(TY+2)(TX+2) halo tile is staged ONCE into LDS
//   (with the fused pre-norm x*scale+shift applied, zero padding after the norm) and reused
//   by all taps: A fragments are ds_read_b128 (4 channels -> 4 MFMAs) at compile-time tap offsets,
//   B fragments (weights, pre-packed in fragment order, L2-resident) are 16-byte global loads.
//   Epilogue: bias + ReLU/Sigmoid + optional ReLU-backward mask, 128-byte row stores.
//
// Weight gradient:
//   GEMM view  M = Cin (32-row tile), N = Cout (32-col tiles), K = voxels, one accumulator per tap.
//   workgroup  = 4 waves over a (ci-tile, co-tile group, patch range); the taps x co-tiles
//   "units" are dealt round-robin to the waves (<= 7 accumulators = 112 VGPRs each); X halo
//   tile and G tile staged in LDS per 2x8x8 patch; split over patch ranges, partial slabs are
//   merged by a deterministic fp64 reduction (no atomics).
#include "tem_common.h"
#include "conv_internal.h"
#include "tem_act.h"
#include <type_traits>

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifndef TEM_MF_RD
#define TEM_MF_RD 0      // weight ring depth of the forward kernels (0: 3 for 32-column tiles, 2 for 64-column tiles)
#endif
#ifndef TEM_MF_KOUTER
#define TEM_MF_KOUTER 1  // MFMA order inside a k-group: k-step outermost (1) or the four k-steps of one accumulator back to back (0, rounds 1-5)
#endif
#ifndef TEM_MF_APF
#define TEM_MF_APF 1     // A fragments of k-group g + 1 are requested before the MFMAs of k-group g
#endif
#ifndef TEM_MF_EARLY_HALO
#define TEM_MF_EARLY_HALO 0   // persistent forward kernel: the next step's halo is requested in front of the first k-group (1) or behind the last
                              // ring load of the step (0).  Measured (profiles/r06_fp32_variants.txt): early is SLOWER (0.646 against 0.631 ms per
                              // launch) -- vmcnt counts in order, so every wait for a ring load issued after the prefetch waits for the ten
                              // HBM loads in front of it; deeper rings (6 / 9 k-groups of cover) do not buy it back
#endif
#ifndef TEM_MF_RD1
#define TEM_MF_RD1 0     // persistent kernel, 32-column tiles: weight ring depth (0: as TEM_MF_RD)
#endif
#ifndef TEM_MF_RD2
#define TEM_MF_RD2 0     // persistent kernel, 64-column tiles
#endif
#ifndef TEM_MF_OCC2
#define TEM_MF_OCC2 2    // workgroups per CU of the 64-column instantiations
#endif
#define CK 16      // input channels per staged chunk (fwd)
#define LSF 20     // LDS floats per halo voxel (16 + 4 pad -> 80 B, keeps b128 alignment)

// Row of a 32-row MFMA tile -> voxel of the wave's 64-voxel share (index into the patch: wv * 64 + value).
// ds_read_b128 is served in four groups of 16 lanes that are not contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the
// same + 32; MI355X_MICROARCH.md, LDS).  With 80-byte halo records (LSF = 20 floats) the eight voxels of an x-row fall into bank
// quads {0, 5, 10, 15, 4, 9, 14, 3} and a y-row moves them by 2: in lane order the 16 reads of a group collided 2-4 ways
// (SQ_LDS_BANK_CONFLICT = 65 % of SQ_LDS_IDX_ACTIVE, profiles/r06_pmc_stalls_fp32.txt).  Rows y and y + 4 are 8 quads apart --
// exactly the complement -- so the lanes are numbered such that every service group holds the two complete x-rows y, y + 4
// (conv_zr.hip uses the same lane numbering for its footprint).  8 x 8 (y, x) shares only; other tiles keep lane order.
#ifndef TEM_MF_LANEMAP
#define TEM_MF_LANEMAP 1
#endif
template <int TY, int TX>
__device__ __forceinline__ int mf_row_voxel(int m, int row) {
    if constexpr (TY == 8 && TX == 8 && TEM_MF_LANEMAP) {
        const int vu = (int)((0x73261540u >> (4 * (row >> 2))) & 7u) * 4 + (row & 3);   // service group -> two complete rows
        const int t = vu >> 3;                                                          // 0..3 -> y rows 0, 4, 1, 5 (+ 2 m)
        return ((t >> 1) + 4 * (t & 1) + 2 * m) * 8 + (vu & 7);
    } else {
        return m * 32 + row;
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == TEM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TEM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// ---------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// Epilogue of the exact-fp32 forward / data-gradient kernels (round 6).  The accumulator tile is D[row = voxel][col = co]:
// register reg of lane (kh, r) holds row (reg & 3) + 8 (reg >> 2) + 4 kh of M-tile m, column r.  Rounds 1-5 computed, PER ELEMENT,
// the global voxel index (((n D + gz) H + gy) W + gx) and two 64-bit addresses from it and ran `act_apply` with its sigmoid
// branch inline: ~31 instructions and a handful of branches per element, 4000+ per unit and wave -- 10-20 k cycles behind every
// unit of 55-220 k MFMA cycles, which is where most of the 22 % the matrix pipe idled in this mode went
// (profiles/r06_mfma_busy_fp32.txt: 0.77-0.78 busy at full clock).  Here a wave's 64 voxels lie in ONE z-plane of the patch, so
// the plane base is a scalar (buffer resource per wave), the in-plane offset of an element is 32-bit arithmetic on compile-time
// row constants, the bounds test disappears for patches inside the volume and the activation is chosen once per unit.
// Needs one z-plane of y / ref / part below 2 GiB (checked by the launcher).
// ---------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mf_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <int TZ, int TY, int TX, int NR>
__device__ __forceinline__ void mf_store_tile(floatx16 (&acc)[2][NR], int wv, int kh, int r, int n, int z0, int y0, int x0,
                                              int cot, int ks, int ksplit, int N, int D, int H, int W, int Cout,
                                              const float* __restrict__ bias, int act, float* __restrict__ y, int64_t y_ld,
                                              const float* __restrict__ ref, int64_t ref_ld, float* __restrict__ part) {
    static_assert((TY * TX) % 64 == 0, "a wave's 64 voxels lie in one z-plane of the patch");
    const int pz = (wv * 64) / (TY * TX);            // wave-uniform
    const int gz = z0 + pz;
    if (gz >= D) return;
    const int wbase = (wv * 64) % (TY * TX);         // first in-plane voxel of this wave's share
    const bool split = ksplit > 1;
    const int64_t plane_v = (((int64_t)n * D + gz) * H + y0) * W + x0;   // voxel index of the patch corner in this z-plane
    const int64_t o_ld = split ? (int64_t)Cout : y_ld;
    float* const obase = split ? part + ((int64_t)ks * N * D * H * W + plane_v) * Cout : y + plane_v * y_ld;
    const __amdgpu_buffer_rsrc_t ro = mf_rsrc(obase);
    const __amdgpu_buffer_rsrc_t rr = mf_rsrc(ref ? ref + plane_v * ref_ld : obase);
    const bool full = (y0 + TY <= H) & (x0 + TX <= W);
    const unsigned old4 = (unsigned)o_ld * 4u, rld4 = (unsigned)ref_ld * 4u;
    auto body = [&](auto full_tag, auto mode_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;   // 0 raw partial sums, 1 bias + max(., floor), 2 bias + sigmoid; +4: ReLU mask from ref
        const bool relu = act == TEM_ACT_RELU;   // (a select, not v_max: NaN and -0 behave exactly as in act_apply)
#pragma unroll
        for (int nn = 0; nn < NR; ++nn) {
            const int co = (cot * NR + nn) * 32 + r;
            const float bv = ((MODE & 3) && bias) ? bias[co] : 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                    const int pw = wbase + mf_row_voxel<TY, TX>(m, row);
                    const int py = pw / TX, px = pw % TX;
                    const unsigned vo = (unsigned)(py * W + px);
                    bool ok = true;
                    if (!FULL) ok = (y0 + py < H) & (x0 + px < W);
                    float o = acc[m][nn][reg];
                    if ((MODE & 3) == 1) {
                        const float t = o + bv;
                        o = (relu && !(t > 0.f)) ? 0.f : t;
                    }
                    if ((MODE & 3) == 2) o = 1.f / (1.f + __expf(-(o + bv)));
                    if (MODE & 4) {
                        const float q = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rr, (FULL || ok) ? vo * rld4 + (unsigned)co * 4u : 0u, 0, 0));
                        o = q > 0.f ? o : 0.f;
                    }
                    if (FULL || ok)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ro, vo * old4 + (unsigned)co * 4u, 0, 0);
                }
            }
        }
    };
    auto go = [&](auto mode_tag) {
        if (full) body(std::true_type{}, mode_tag);
        else body(std::false_type{}, mode_tag);
    };
    if (split) go(std::integral_constant<int, 0>{});
    else if (act == TEM_ACT_SIGMOID) { if (ref) go(std::integral_constant<int, 6>{}); else go(std::integral_constant<int, 2>{}); }
    else if (ref) go(std::integral_constant<int, 5>{});
    else go(std::integral_constant<int, 1>{});
}

template <int KD, int KH, int KW, int TZ, int TY, int TX, int NR, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : (NR == 2 ? TEM_MF_OCC2 : 3)) void k_conv_fwd_mfma(
    const float* __restrict__ x, int64_t x_ld, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int64_t y_ld,
    const float* __restrict__ ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout, int act, int nZ,
    int nY, int nX, int ksplit, float* __restrict__ part) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1;
    constexpr int HV = HZ * HY * HX;
    constexpr int NTH = NW * 64;  // NW waves; wave w owns patch voxels [64w, 64w+64)
    constexpr int NIT = (HV * 4 + NTH - 1) / NTH;
    static_assert(TZ * TY * TX == NTH, "one patch voxel per thread");
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [HV][LSF]

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
    const int ks = bid / N;  // split-K slice of the input channels (deep, spatially small layers)
    const int z0 = ptz * TZ, y0 = pty * TY, x0 = ptx * TX;

    int abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int p = wv * 64 + mf_row_voxel<TY, TX>(m, r);
        const int pz = p / (TY * TX), py = (p / TX) % TY, px = p % TX;
        abase[m] = ((pz * HY + py) * HX + px) * LSF + kh * 4;
    }

    floatx16 acc[2][NR];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < NR; ++nn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][nn][i] = 0.f;

    const int cin8 = Cin >> 3;
    const int c4 = tid & 3;  // channel quad of this thread's staging items (256 % 4 == 0)

    const int cpk = (Cin / CK) / ksplit;
    const int chunk_begin = ks * cpk, chunk_end = (ks + 1) * cpk;
    constexpr int NG = 2 * NT;                       // k-groups (8 input channels each) per chunk
    // ring depth (NG % RD == 0 keeps the phase across chunks); one k-group is 8*NR MFMAs = 512*NR cycles
    constexpr int RD = (TEM_MF_RD && NG % TEM_MF_RD == 0) ? TEM_MF_RD : (NR == 1 && NG % 3 == 0) ? 3 : 2;
    const int tapstride = cin8 * 64;
    const float4* wq[NR];  // this lane's slot in the first fragment of each column tile
#pragma unroll
    for (int nn = 0; nn < NR; ++nn)
        wq[nn] = reinterpret_cast<const float4*>(wp) + (int64_t)(cot * NR + nn) * NT * cin8 * 64 + lane;
    float4 bq[RD][NR];
#pragma unroll
    for (int gp = 0; gp < RD - 1; ++gp)
#pragma unroll
        for (int nn = 0; nn < NR; ++nn)
            bq[gp][nn] = wq[nn][(int64_t)chunk_begin * 128 + (gp >> 1) * tapstride + (gp & 1) * 64];
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        // ---- stage the halo tile of this 16-channel chunk (global -> regs -> LDS) ----
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) {
            sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + chunk * CK + c4 * 4);
            sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + chunk * CK + c4 * 4);
        }
        float4 tmp[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * NTH;
            const int hv = item >> 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hv < HV) {
                const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
                const int gz = z0 + hz - PZ, gy = y0 + hy - PY, gx = x0 + hx - PX;
                if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    v = *reinterpret_cast<const float4*>(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * x_ld +
                                                         chunk * CK + c4 * 4);
                    v.x = fmaf(v.x, sc4.x, sf4.x);
                    v.y = fmaf(v.y, sc4.y, sf4.y);
                    v.z = fmaf(v.z, sc4.z, sf4.z);
                    v.w = fmaf(v.w, sc4.w, sf4.w);
                }
            }
            tmp[it] = v;
        }
        __syncthreads();  // previous chunk's MFMA reads are done
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * NTH;
            const int hv = item >> 2;
            if (hv < HV) *reinterpret_cast<float4*>(lds + hv * LSF + c4 * 4) = tmp[it];
        }
        __syncthreads();

        // ---- MFMA over all taps of this chunk ----
        // B fragments (weights) come straight from L2 with a register ring RD deep: the load for
        // k-group g+RD-1 is issued before the MFMAs of k-group g, so ~RD-1 k-groups (>= 512 MFMA
        // cycles each) cover the L2 latency.  (Left to itself hipcc issues each load two MFMAs before
        // its use: PMC showed the matrix pipe 28% idle.)
        int ts = tapstride;
        asm volatile("" : "+s"(ts));  // opaque per chunk: keeps LICM from hoisting 2*NT address pairs out of the loop
        float4 apf[2];
        if (TEM_MF_APF) {
#pragma unroll
            for (int m = 0; m < 2; ++m) apf[m] = *reinterpret_cast<const float4*>(lds + abase[m]);   // tap 0, k-group 0
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int tap = g >> 1, kg = g & 1;
            const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
            const int toff = ((tz * HY + ty) * HX + tx) * LSF;
            {   // prefetch
                const int gp = g + RD - 1;
                if (gp < NG) {
#pragma unroll
                    for (int nn = 0; nn < NR; ++nn)
                        bq[gp % RD][nn] = wq[nn][(int64_t)chunk * 128 + (gp >> 1) * ts + (gp & 1) * 64];
                } else if (chunk + 1 < chunk_end) {
#pragma unroll
                    for (int nn = 0; nn < NR; ++nn)
                        bq[gp % RD][nn] = wq[nn][(int64_t)(chunk + 1) * 128 + ((gp - NG) >> 1) * ts + ((gp - NG) & 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0x38F);  // everything but VMEM may cross: keep the loads early
            }
            // A fragments one k-group AHEAD (TEM_MF_APF, round 6): left to itself hipcc issues the two ds_read_b128 of a k-group one MFMA
            // before the MFMAs that consume them -- an LDS round trip in front of every 8 NR MFMAs of the wave
            float4 a[2];
            if (TEM_MF_APF) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = apf[m];
                if (g + 1 < NG) {
                    const int tap1 = (g + 1) >> 1, kg1 = (g + 1) & 1;
                    const int toff1 = (((tap1 / (KH * KW)) * HY + (tap1 / KW) % KH) * HX + tap1 % KW) * LSF;
#pragma unroll
                    for (int m = 0; m < 2; ++m) apf[m] = *reinterpret_cast<const float4*>(lds + abase[m] + toff1 + kg1 * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const float4*>(lds + abase[m] + toff + kg * 8);
            }
            // k-step outermost (TEM_MF_KOUTER): consecutive MFMAs go to DIFFERENT accumulators -- a dependent v_mfma_f32_32x32x2_f32
            // cannot start before its predecessor's 16 passes have written back
#pragma unroll
            for (int kq = 0; kq < (TEM_MF_KOUTER ? 4 : 1); ++kq)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < NR; ++nn) {
                    const float4 bb = bq[g % RD][nn];
                    if (!TEM_MF_KOUTER || kq == 0) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bb.x, acc[m][nn], 0, 0, 0);
                    if (!TEM_MF_KOUTER || kq == 1) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bb.y, acc[m][nn], 0, 0, 0);
                    if (!TEM_MF_KOUTER || kq == 2) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bb.z, acc[m][nn], 0, 0, 0);
                    if (!TEM_MF_KOUTER || kq == 3) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bb.w, acc[m][nn], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: D[row = voxel][col = co];  row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31 ----
    mf_store_tile<TZ, TY, TX, NR>(acc, wv, kh, r, n, z0, y0, x0, cot, ks, ksplit, N, D, H, W, Cout, bias, act, y, y_ld, ref, ref_ld, part);
}

// ---------------------------------------------------------------------------
// forward / dgrad, persistent + software-pipelined variant.
// Ablation of the kernel above (32->32 @ 2x128^3: 107 TFLOP/s; without its staging loads 130,
// without its stores 116, without both and without operand fetches 142) showed that what
// keeps the matrix pipe idle is the per-workgroup serial part: halo loads -> wait -> LDS write ->
// barrier in front of every 16-channel chunk, and the store tail.  Here a fixed grid of workgroups
// walks (patch, Cout-tile, chunk) steps; while the MFMAs of step s run, the halo tile of step s+1
// (possibly of the next patch) is already in flight into registers and the epilogue stores of a
// finished patch overlap the next patch's arithmetic.  The halo loads are issued right after the
// last weight-ring load of the step so that no in-order vmcnt wait of the ring sits behind them.
// ---------------------------------------------------------------------------
template <int KD, int KH, int KW, int TZ, int TY, int TX, int NR>
__global__ __launch_bounds__(256, NR == 1 ? 3 : TEM_MF_OCC2) void k_conv_fwd_mfma_p(
    const float* __restrict__ x, int64_t x_ld, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int64_t y_ld,
    const float* __restrict__ ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout, int act, int nZ,
    int nY, int nX, int ksplit, float* __restrict__ part, int total_units) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1;
    constexpr int HV = HZ * HY * HX;
    constexpr int NIT = (HV * 4 + 255) / 256;
    constexpr int NG = 2 * NT;
    // ring depth of the persistent kernel: the waits for ring loads issued BEFORE the halo prefetch are the ones that do not wait
    // for it (vmcnt counts in order), i.e. RD k-groups of cover for the prefetch
    constexpr int RDP = NR == 1 ? TEM_MF_RD1 : TEM_MF_RD2;
    constexpr int RD = (RDP && NG % RDP == 0) ? RDP : (TEM_MF_RD && NG % TEM_MF_RD == 0) ? TEM_MF_RD : (NR == 1 && NG % 3 == 0) ? 3 : 2;
    constexpr int GH = TEM_MF_EARLY_HALO ? 0 : (NG - RD + 1 > 0 ? NG - RD + 1 : 0);  // k-group after whose ring issue the halo prefetch goes out
    static_assert(TZ * TY * TX == 256, "patch must hold 256 voxels");
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [HV][LSF]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;
    const int ncot = Cout / (32 * NR);
    const int cin8 = Cin >> 3;
    const int c4 = tid & 3;
    const int cpk = (Cin / CK) / ksplit;
    const int tapstride = cin8 * 64;

    int abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int p = wv * 64 + mf_row_voxel<TY, TX>(m, r);
        const int pz = p / (TY * TX), py = (p / TX) % TY, px = p % TX;
        abase[m] = ((pz * HY + py) * HX + px) * LSF + kh * 4;
    }

    // ---- step descriptors (all wave-uniform) ----
    int unit = blockIdx.x;
    if (unit >= total_units) return;
    int c_n, c_z0, c_y0, c_x0, c_cot, c_ks, c_chunk;
    {
        int b = unit;
        c_cot = b % ncot; b /= ncot;
        c_x0 = (b % nX) * TX; b /= nX;
        c_y0 = (b % nY) * TY; b /= nY;
        c_z0 = (b % nZ) * TZ; b /= nZ;
        c_n = b % N;
        c_ks = b / N;
        c_chunk = c_ks * cpk;
    }
    const float4* wbase = reinterpret_cast<const float4*>(wp) + lane;

    float4 tmp[NIT];
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned inb = 0;  // bit `it`: staging item is inside the volume
    // Halo loads of one step (raw values; the fused norm is applied at LDS-write time).  Round 6: UNCONDITIONAL buffer loads with
    // per-thread constant 32-bit offsets from the halo origin (an item outside the volume, or beyond the tile, reads with an offset
    // beyond the buffer: zeros) -- rounds 1-5 kept every load inside `if (voxel in range)`, so the compiler could not count what was
    // in flight at the join points.  The prefetch is still issued behind the LAST weight-ring load of a step (TEM_MF_EARLY_HALO
    // above says why not earlier); the straight-line loads alone are worth 1.7 % of the launch.  Needs the halo of a patch inside
    // 2 GiB (launcher).
    constexpr unsigned OOB = 0x80000000u;
    unsigned hoff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int hv = (tid + it * 256) >> 2;
        const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
        hoff[it] = hv < HV ? ((unsigned)((hz * H + hy) * W + hx) * (unsigned)x_ld + (unsigned)(c4 * 4)) * 4u : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsc = mf_rsrc(scale ? (const void*)scale : (const void*)x);
    const __amdgpu_buffer_rsrc_t rsf = mf_rsrc(scale ? (const void*)shift : (const void*)x);
    const bool has_scale = scale != nullptr;
    typedef unsigned int mf_u4 __attribute__((ext_vector_type(4)));
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned off) -> float4 {
        const mf_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        typedef float mf_f4 __attribute__((ext_vector_type(4)));
        const mf_f4 f = __builtin_bit_cast(mf_f4, v);
        return make_float4(f.x, f.y, f.z, f.w);
    };
#define LOAD_HALO(NN, Z0, Y0, X0, CHUNK)                                                                         \
    do {                                                                                                         \
        inb = 0;                                                                                                 \
        {                                                                                                        \
            const unsigned so = has_scale ? (unsigned)(((int64_t)(NN) * Cin + (CHUNK) * CK + c4 * 4) * 4) : OOB;  \
            sc4 = ld4(rsc, so);                                                                                  \
            sf4 = ld4(rsf, so);                                                                                  \
        }                                                                                                        \
        const __amdgpu_buffer_rsrc_t rx_ = mf_rsrc(x + ((((int64_t)(NN) * D + ((Z0) - PZ)) * H + ((Y0) - PY)) * W + ((X0) - PX)) * x_ld + \
                                                   (CHUNK) * CK);                                                \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                     \
            const int hv = (tid + it * 256) >> 2;                                                                \
            const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;                   \
            const unsigned gz = (unsigned)((Z0) + hz - PZ), gy = (unsigned)((Y0) + hy - PY), gx = (unsigned)((X0) + hx - PX); \
            const bool ok = (hv < HV) & (gz < (unsigned)D) & (gy < (unsigned)H) & (gx < (unsigned)W);            \
            inb |= ok ? (1u << it) : 0u;                                                                         \
            tmp[it] = ld4(rx_, ok ? hoff[it] : OOB);                                                             \
        }                                                                                                        \
    } while (0)

    LOAD_HALO(c_n, c_z0, c_y0, c_x0, c_chunk);

    float4 bq[RD][NR];
    {
        const int64_t wo = (int64_t)(c_cot * NR) * NT * cin8 * 64 + (int64_t)c_chunk * 128;
#pragma unroll
        for (int gp = 0; gp < RD - 1; ++gp)
#pragma unroll
            for (int nn = 0; nn < NR; ++nn)
                bq[gp][nn] = wbase[wo + (int64_t)nn * NT * cin8 * 64 + (gp >> 1) * tapstride + (gp & 1) * 64];
    }

    floatx16 acc[2][NR];
    bool first = true;
    while (true) {
        // ---- next step ----
        const bool last_chunk = (c_chunk + 1 == (c_ks + 1) * cpk);
        int n_unit = unit, n_n = c_n, n_z0 = c_z0, n_y0 = c_y0, n_x0 = c_x0, n_cot = c_cot, n_ks = c_ks,
            n_chunk = c_chunk + 1;
        bool has_next = true;
        if (last_chunk) {
            n_unit = unit + gridDim.x;
            has_next = n_unit < total_units;
            if (has_next) {
                int b = n_unit;
                n_cot = b % ncot; b /= ncot;
                n_x0 = (b % nX) * TX; b /= nX;
                n_y0 = (b % nY) * TY; b /= nY;
                n_z0 = (b % nZ) * TZ; b /= nZ;
                n_n = b % N;
                n_ks = b / N;
                n_chunk = n_ks * cpk;
            } else {
                n_chunk = c_chunk;   // nothing follows: the (unconditional) prefetch below re-reads this step's own tile
            }
        }
        // ---- halo tile of the current step: registers -> LDS (fused pre-norm, zero padding after it) ----
        __syncthreads();  // every wave is done reading the previous tile
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int hv = (tid + it * 256) >> 2;
            if (hv < HV) {
                float4 v = tmp[it];
                if (has_scale && (inb & (1u << it))) {
                    v.x = fmaf(v.x, sc4.x, sf4.x);
                    v.y = fmaf(v.y, sc4.y, sf4.y);
                    v.z = fmaf(v.z, sc4.z, sf4.z);
                    v.w = fmaf(v.w, sc4.w, sf4.w);
                }
                *reinterpret_cast<float4*>(lds + hv * LSF + c4 * 4) = v;
            }
        }
        __syncthreads();
        if (first) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < NR; ++nn)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][nn][i] = 0.f;
        }
        // ---- MFMA over all taps of this chunk ----
        int ts = tapstride;
        asm volatile("" : "+s"(ts));
        const int64_t wo_c = (int64_t)(c_cot * NR) * NT * cin8 * 64 + (int64_t)c_chunk * 128;
        const int64_t wo_n = (int64_t)(n_cot * NR) * NT * cin8 * 64 + (int64_t)n_chunk * 128;
        const int64_t wnn = (int64_t)NT * cin8 * 64;
        float4 apf[2];
        if (TEM_MF_APF) {
#pragma unroll
            for (int m = 0; m < 2; ++m) apf[m] = *reinterpret_cast<const float4*>(lds + abase[m]);   // tap 0, k-group 0
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int tap = g >> 1, kg = g & 1;
            const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
            const int toff = ((tz * HY + ty) * HX + tx) * LSF;
            {
                const int gp = g + RD - 1;
                if (gp < NG) {
#pragma unroll
                    for (int nn = 0; nn < NR; ++nn)
                        bq[gp % RD][nn] = wbase[wo_c + nn * wnn + (gp >> 1) * ts + (gp & 1) * 64];
                } else if (has_next) {
#pragma unroll
                    for (int nn = 0; nn < NR; ++nn)
                        bq[gp % RD][nn] = wbase[wo_n + nn * wnn + ((gp - NG) >> 1) * ts + ((gp - NG) & 1) * 64];
                }
                __builtin_amdgcn_sched_barrier(0x38F);
            }
            if (g == GH) {   // (no `if (has_next)`: a branch around loads would hide them from the wait counting; the last step of a
                             //  workgroup reloads its own tile)
                LOAD_HALO(n_n, n_z0, n_y0, n_x0, n_chunk);
                __builtin_amdgcn_sched_barrier(0x38F);
            }
            // A fragments one k-group AHEAD (TEM_MF_APF, round 6): left to itself hipcc issues the two ds_read_b128 of a k-group one MFMA
            // before the MFMAs that consume them -- an LDS round trip in front of every 8 NR MFMAs of the wave
            float4 a[2];
            if (TEM_MF_APF) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = apf[m];
                if (g + 1 < NG) {
                    const int tap1 = (g + 1) >> 1, kg1 = (g + 1) & 1;
                    const int toff1 = (((tap1 / (KH * KW)) * HY + (tap1 / KW) % KH) * HX + tap1 % KW) * LSF;
#pragma unroll
                    for (int m = 0; m < 2; ++m) apf[m] = *reinterpret_cast<const float4*>(lds + abase[m] + toff1 + kg1 * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const float4*>(lds + abase[m] + toff + kg * 8);
            }
            // k-step outermost (TEM_MF_KOUTER): consecutive MFMAs go to DIFFERENT accumulators -- a dependent v_mfma_f32_32x32x2_f32
            // cannot start before its predecessor's 16 passes have written back
#pragma unroll
            for (int kq = 0; kq < (TEM_MF_KOUTER ? 4 : 1); ++kq)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < NR; ++nn) {
                    const float4 bb = bq[g % RD][nn];
                    if (!TEM_MF_KOUTER || kq == 0) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bb.x, acc[m][nn], 0, 0, 0);
                    if (!TEM_MF_KOUTER || kq == 1) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bb.y, acc[m][nn], 0, 0, 0);
                    if (!TEM_MF_KOUTER || kq == 2) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bb.z, acc[m][nn], 0, 0, 0);
                    if (!TEM_MF_KOUTER || kq == 3) acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bb.w, acc[m][nn], 0, 0, 0);
                }
        }
        first = false;
        // ---- epilogue of a finished (patch, Cout tile) ----
        if (last_chunk) {
            mf_store_tile<TZ, TY, TX, NR>(acc, wv, kh, r, c_n, c_z0, c_y0, c_x0, c_cot, c_ks, ksplit, N, D, H, W, Cout, bias, act, y,
                                          y_ld, ref, ref_ld, part);
            first = true;
        }
        if (!has_next) break;
        unit = n_unit;
        c_n = n_n; c_z0 = n_z0; c_y0 = n_y0; c_x0 = n_x0; c_cot = n_cot; c_ks = n_ks; c_chunk = n_chunk;
    }
#undef LOAD_HALO
}


// y = act(sum_ks part[ks] + bias) [* (ref > 0)]
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_epilogue(const float* __restrict__ part, int ksplit, int64_t NV,
                                                         int Cout, const float* __restrict__ bias, int act,
                                                         const T* __restrict__ ref, int64_t ref_ld,
                                                         T* __restrict__ y, int64_t y_ld) {
    const int cq = Cout >> 2;
    const int64_t items = NV * cq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int64_t v = i / cq;
        const int c0 = (int)(i % cq) * 4;
        float4 a = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r = make_float4(1.f, 1.f, 1.f, 1.f);
        if (ref) r = act_ld4(ref + v * ref_ld + c0);   // issued with the slices, used at the end
        // four slices per trip, loads unconditional (slices beyond ksplit re-read the last one and are dropped): the
        // one-load-per-trip loop was ksplit dependent round trips; the sum keeps its order k = 0, 1, 2, ...
        for (int k0 = 0; k0 < ksplit; k0 += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u < ksplit ? k0 + u : ksplit - 1;
                p[u] = *reinterpret_cast<const float4*>(part + ((int64_t)k * NV + v) * Cout + c0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = k0 + u < ksplit;
                a.x += in ? p[u].x : 0.f;
                a.y += in ? p[u].y : 0.f;
                a.z += in ? p[u].z : 0.f;
                a.w += in ? p[u].w : 0.f;
            }
        }
        a.x = act_apply(a.x, act);
        a.y = act_apply(a.y, act);
        a.z = act_apply(a.z, act);
        a.w = act_apply(a.w, act);
        if (ref) {
            if (!(r.x > 0.f)) a.x = 0.f;
            if (!(r.y > 0.f)) a.y = 0.f;
            if (!(r.z > 0.f)) a.z = 0.f;
            if (!(r.w > 0.f)) a.w = 0.f;
        }
        act_st4(y + v * y_ld + c0, a);
    }
}

// k_splitk_epilogue that also writes the first stage of the statistics of y (sum, sum of squares per channel) for the norm
// that reads y next -- the partial rows k_norm_partial<.,0> would produce in a pass of its own (one launch and one read
// of y less per layer of the 8^3 / 16^3 levels).  Block (b, n): voxels [b VB, (b+1) VB) of sample n, thread = (channel
// quad q, row r of 256 / cq rows); stat: [N][gridDim.x][Cout][2].
// BWD: the launch is a data gradient whose output lands behind a norm (the one in front of the conv): the rows are
// (sum g, sum g * xn) with xn = (xin - mean) * rstd of that norm's input xin -- the first stage of its backward
// (k_norm_partial<.,1>), TEM_BP_NORM_SUMS of tem_conv3d_fwd_ex.
struct SplitkNormIn {
    const void* xin;   // element type of the launch (T)
    int64_t xin_ld;
    const float* mean;
    const float* rstd;
    int G;
};
template <bool BWD, typename T>
__global__ __launch_bounds__(256) void k_splitk_epilogue_stats(const float* __restrict__ part, int ksplit, int64_t V,
                                                               int Cout, const float* __restrict__ bias, int act,
                                                               const T* __restrict__ ref, int64_t ref_ld,
                                                               T* __restrict__ y, int64_t y_ld, int VB,
                                                               float* __restrict__ stat, SplitkNormIn ni) {
    __shared__ float sh[2048];   // [rows][Cout][2], rows * Cout = 1024
    const int cq = Cout >> 2, rows = 256 / cq;
    const int q = threadIdx.x % cq, r = threadIdx.x / cq, c0 = q * 4;
    const int n = blockIdx.y, b = blockIdx.x;
    const int64_t NV = V * gridDim.y;
    const int64_t v0 = (int64_t)b * VB, v1 = v0 + VB < V ? v0 + VB : V;
    const float4 bz = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (BWD) {
        const int cg = Cout / ni.G;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mu[j] = ni.mean[n * ni.G + (c0 + j) / cg];
            rs[j] = ni.rstd[n * ni.G + (c0 + j) / cg];
        }
    }
    // VB = 4 rows of `rows` voxels: exactly four trips, unrolled so that the loads of all four voxels are in flight together
    // (voxels beyond the range read a clamped one and are dropped)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t vl_raw = v0 + r + (int64_t)u * rows;
        const bool vin = vl_raw < v1;
        const int64_t vl = vin ? vl_raw : v1 - 1;
        const int64_t v = (int64_t)n * V + vl;
        float4 a = bz;
        float4 rr = make_float4(1.f, 1.f, 1.f, 1.f), xi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ref) rr = act_ld4(ref + v * ref_ld + c0);
        if constexpr (BWD) xi = act_ld4(static_cast<const T*>(ni.xin) + v * ni.xin_ld + c0);
        for (int k0 = 0; k0 < ksplit; k0 += 4) {
            float4 p[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = k0 + s < ksplit ? k0 + s : ksplit - 1;
                p[s] = *reinterpret_cast<const float4*>(part + ((int64_t)k * NV + v) * Cout + c0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool in = k0 + s < ksplit;
                a.x += in ? p[s].x : 0.f;
                a.y += in ? p[s].y : 0.f;
                a.z += in ? p[s].z : 0.f;
                a.w += in ? p[s].w : 0.f;
            }
        }
        a.x = act_apply(a.x, act);
        a.y = act_apply(a.y, act);
        a.z = act_apply(a.z, act);
        a.w = act_apply(a.w, act);
        if (ref) {
            if (!(rr.x > 0.f)) a.x = 0.f;
            if (!(rr.y > 0.f)) a.y = 0.f;
            if (!(rr.z > 0.f)) a.z = 0.f;
            if (!(rr.w > 0.f)) a.w = 0.f;
        }
        if constexpr (sizeof(T) == 2) {   // the sums below describe the tensor AS STORED (what the next norm reads)
            const unsigned p0 = act_pk<T>(a.x, a.y), p1 = act_pk<T>(a.z, a.w);
            a = make_float4(act_lo<T>(p0), act_hi<T>(p0), act_lo<T>(p1), act_hi<T>(p1));
        }
        if (vin) act_st4(y + v * y_ld + c0, a);
        const float av[4] = {a.x, a.y, a.z, a.w}, xv[4] = {xi.x, xi.y, xi.z, xi.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float aj = vin ? av[j] : 0.f;
            s0[j] += aj;
            s1[j] = fmaf(aj, BWD ? (xv[j] - mu[j]) * rs[j] : aj, s1[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh[(r * Cout + c0 + j) * 2 + 0] = s0[j];
        sh[(r * Cout + c0 + j) * 2 + 1] = s1[j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Cout * 2; t += 256) {
        float a = 0.f;
        for (int k = 0; k < rows; ++k) a += sh[k * Cout * 2 + t];
        stat[((int64_t)n * gridDim.x + b) * Cout * 2 + t] = a;
    }
}

// partial rows per sample of tem_splitk_epilogue_stats (0: this Cout has no such epilogue)
int64_t tem_splitk_stat_blocks(int64_t V, int Cout) {
    const int cq = Cout >> 2;
    if (Cout % 4 || cq < 8 || cq > 256 || (cq & (cq - 1))) return 0;
    const int VB = 4 * (256 / cq);
    return (V + VB - 1) / VB;
}

void tem_splitk_epilogue_stats(const float* part, int ksplit, int N, int64_t V, int Cout, const float* bias, int act,
                               const float* ref, int64_t ref_ld, float* y, int64_t y_ld, float* stat, hipStream_t s) {
    const int VB = 4 * (256 / (Cout >> 2));
    TEM_ST_SWITCH(tem_call_st.y, T,
                  hipLaunchKernelGGL((k_splitk_epilogue_stats<false, T>), dim3((unsigned)tem_splitk_stat_blocks(V, Cout), (unsigned)N),
                                     dim3(256), 0, s, part, ksplit, V, Cout, bias, act, (const T*)ref, ref_ld, (T*)y, y_ld, VB, stat,
                                     SplitkNormIn{nullptr, 0, nullptr, nullptr, 1}));
}

// the epilogue of a DATA GRADIENT that also writes the first stage of the backward of the norm its output lands behind
void tem_splitk_epilogue_bwd_sums(const float* part, int ksplit, int N, int64_t V, int Cout, const float* bias, int act,
                                  const float* ref, int64_t ref_ld, float* y, int64_t y_ld, const TemDgradSumsReq& rq,
                                  hipStream_t s) {
    const int VB = 4 * (256 / (Cout >> 2));
    TEM_ST_SWITCH(tem_call_st.y, T,
                  hipLaunchKernelGGL((k_splitk_epilogue_stats<true, T>), dim3((unsigned)tem_splitk_stat_blocks(V, Cout), (unsigned)N),
                                     dim3(256), 0, s, part, ksplit, V, Cout, bias, act, (const T*)ref, ref_ld, (T*)y, y_ld, VB, rq.part,
                                     SplitkNormIn{rq.x, rq.x_ld, rq.mean, rq.rstd, rq.G}));
}

// Split the input channels over `ks` workgroups when the (patches x Cout tiles) grid cannot fill
// 256 CUs: 8^3 and 16^3 levels of the U-Net (1% of the FLOPs, 30% of the time without it).
int tem_fwd_ksplit(int64_t nblk, int nchunks) {
    if (nblk >= 384) return 1;
    int64_t target = (768 + nblk - 1) / nblk;
    int ks = 1;
    for (int d = 1; d <= nchunks; ++d)
        if (nchunks % d == 0 && d <= target) ks = d;
    // "fwd_ksplit_chunks" = c > 0: no partial accumulates more than c chunks (27 c MFMA steps per accumulator).  The matrix
    // core rounds each step's products into the fp32 accumulator; many steps in ONE accumulator cost accuracy that the
    // split-K epilogue's fp32 adds do not -- only where split-K runs anyway (few workgroups).
    const long long mc = tem_option(TEM_OPT_FWD_KSPLIT_CHUNKS);
    if (mc > 0)
        for (int d = ks; d <= nchunks; ++d)
            if (nchunks % d == 0 && nchunks / d <= mc) { ks = d; break; }
    return ks;
}

void tem_splitk_epilogue(const float* part, int ksplit, int64_t NV, int Cout, const float* bias, int act,
                         const float* ref, int64_t ref_ld, float* y, int64_t y_ld, hipStream_t s) {
    TEM_ST_SWITCH(tem_call_st.y, T,
                  hipLaunchKernelGGL(k_splitk_epilogue<T>, dim3(tem_grid_1d(NV * (Cout / 4), 256)), dim3(256), 0, s, part, ksplit, NV,
                                     Cout, bias, act, (const T*)ref, ref_ld, (T*)y, y_ld));
}

static void fwd_geometry(int N, int D, int H, int W, int Cout, int kd, bool& flat, int& TZ, int& TY, int& TX, int& NR,
                         int64_t& nblk) {
    flat = (D == 1 && kd == 1);
    TZ = flat ? 1 : 4;
    TY = flat ? 16 : 8;
    TX = flat ? 16 : 8;
    NR = (Cout % 64 == 0) ? 2 : 1;
    nblk = (int64_t)N * ((D + TZ - 1) / TZ) * ((H + TY - 1) / TY) * ((W + TX - 1) / TX) * (Cout / (32 * NR));
}

int64_t tem_conv_fwd_mfma_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    (void)kh;
    (void)kw;
    bool flat;
    int TZ, TY, TX, NR;
    int64_t nblk;
    fwd_geometry(N, D, H, W, Cout, kd, flat, TZ, TY, TX, NR, nblk);
    int ks = tem_fwd_ksplit(nblk, Cin / CK);
    return ks > 1 ? (int64_t)ks * N * D * H * W * Cout * 4 : 0;
}

template <int KD, int KH, int KW, int TZ, int TY, int TX, int NR, int NW = 4>
static void launch_fwd(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                       const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H,
                       int W, int Cin, int Cout, int act, int ksplit, float* part, hipStream_t s) {
    constexpr int HV = (TZ + KD - 1) * (TY + KH - 1) * (TX + KW - 1);
    const int nZ = (D + TZ - 1) / TZ, nY = (H + TY - 1) / TY, nX = (W + TX - 1) / TX;
    const int64_t nblk = (int64_t)N * nZ * nY * nX * (Cout / (32 * NR)) * ksplit;
    size_t ldsb = (size_t)HV * LSF * sizeof(float);
    // persistent pipelining pays for the 64-column tiles (measured +5%), not for the 32-column ones
    const int pmode = (int)tem_option(TEM_OPT_FWD_PERSISTENT);
    // (the persistent kernel addresses a patch's halo with 32-bit offsets from its origin)
    const bool halo32 = (int64_t)(TZ + KD - 1) * H * W * x_ld * 4 < (1ll << 31);
    const bool persistent = NW == 4 && halo32 && (pmode < 0 ? NR == 2 : pmode != 0);
    if constexpr (NW == 4) if (persistent) {
        static int ncu = 0;
        if (!ncu) {
            ncu = tem_device_cus();
            if (ncu <= 0) ncu = 256;
        }
        const int bpc = NR == 1 ? 3 : TEM_MF_OCC2;
        const int64_t grid = nblk < (int64_t)ncu * bpc ? nblk : (int64_t)ncu * bpc;
        hipLaunchKernelGGL((k_conv_fwd_mfma_p<KD, KH, KW, TZ, TY, TX, NR>), dim3((unsigned)grid), dim3(256), ldsb, s, x,
                           x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, nZ, nY, nX,
                           ksplit, part, (int)nblk);
    }
    if (!persistent) {
        hipLaunchKernelGGL((k_conv_fwd_mfma<KD, KH, KW, TZ, TY, TX, NR, NW>), dim3((unsigned)nblk), dim3(NW * 64), ldsb,
                           s, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, nZ, nY,
                           nX, ksplit, part);
    }
    if (ksplit > 1) {
        const int64_t NV = (int64_t)N * D * H * W;
        hipLaunchKernelGGL(k_splitk_epilogue<float>, dim3(tem_grid_1d(NV * (Cout / 4), 256)), dim3(256), 0, s, part, ksplit, NV,
                           Cout, bias, act, ref, ref_ld, y, y_ld);
    }
}

int tem_conv_fwd_mfma(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                      const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                      int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                      hipStream_t s) {
    TEM_REQUIRE(Cin % 16 == 0 && Cout % 32 == 0, "tem_conv3d_fwd(mfma): needs Cin%%16==0 and Cout%%32==0 (got %d,%d)",
                Cin, Cout);
    TEM_REQUIRE(x_ld % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)wp % 16 == 0),
                "tem_conv3d_fwd(mfma): x / packed weights must be 16-byte aligned with ld%%4==0");
    TEM_REQUIRE(!scale || (((uintptr_t)scale % 16 == 0) && ((uintptr_t)shift % 16 == 0)),
                "tem_conv3d_fwd(mfma): scale/shift must be 16-byte aligned");
    {
        int64_t mld = y_ld > Cout ? y_ld : Cout;
        if (ref && ref_ld > mld) mld = ref_ld;
        TEM_REQUIRE((int64_t)H * W * mld * 4 < (1ll << 31),
                    "tem_conv3d_fwd(mfma): one z-plane of y / ref must stay below 2 GiB (32-bit offsets inside a plane)");
    }
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    bool flat;
    int TZ, TY, TX, NR;
    int64_t nblk0;
    fwd_geometry(N, D, H, W, Cout, kd, flat, TZ, TY, TX, NR, nblk0);
    const bool nr2 = NR == 2;
    int ks = tem_fwd_ksplit(nblk0, Cin / CK);
    const bool vec_ok = (y_ld % 4 == 0) && ((uintptr_t)y % 16 == 0) && (!ref || (ref_ld % 4 == 0 && (uintptr_t)ref % 16 == 0)) &&
                        (!bias || (uintptr_t)bias % 16 == 0);
    if (ks > 1 && (!ws || !vec_ok || ws_bytes < (int64_t)ks * N * D * H * W * Cout * 4)) ks = 1;
    float* part = (float*)ws;
#define GO(KD, KH, KW, TZ, TY, TX)                                                                                 \
    do {                                                                                                           \
        if (nr2)                                                                                                   \
            launch_fwd<KD, KH, KW, TZ, TY, TX, 2>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, \
                                                  Cin, Cout, act, ks, part, s);                                    \
        else                                                                                                       \
            launch_fwd<KD, KH, KW, TZ, TY, TX, 1>(x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, \
                                                  Cin, Cout, act, ks, part, s);                                    \
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
        tem_set_error("tem_conv3d_fwd(mfma): kernel (%d,%d,%d) has no MFMA instantiation", kd, kh, kw);
        return TEM_EINVAL;
    }
#undef GO
    return TEM_OK;
}

// ---------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------
#define WG_TZ 2
#define WG_TY 8
#define WG_TX 8
#define WG_MAXU 7   // accumulators per wave

template <int KD, int KH, int KW, int NCO>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_mfma(const float* __restrict__ x, int64_t x_ld,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ g, int64_t g_ld,
                                                            float* __restrict__ part,
                                                            float* __restrict__ dbpart, int N, int D, int H, int W,
                                                            int Cin, int Cout, int T, int S, int P, int nZ, int nY,
                                                            int nX) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int HZ = WG_TZ + KD - 1, HY = WG_TY + KH - 1, HX = WG_TX + KW - 1;
    constexpr int HV = HZ * HY * HX;
    constexpr int PV = WG_TZ * WG_TY * WG_TX;  // 128 patch voxels
    constexpr int XIT = (HV * 8 + 255) / 256;
    constexpr int GC = 32 * NCO;               // staged g channels
    constexpr int GIT = (PV * (GC / 4) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ldsX = lds;                // [HV][32]
    float* ldsG = lds + HV * 32;      // [PV][GC]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kh = lane >> 5, r = lane & 31;

    const int bid = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid % T, sp = bid / T;
    const int ncit = Cin >> 5;
    const int cit = tile % ncit, cog = tile / ncit;
    const int ncot_total = Cout >> 5;
    int nco_here = ncot_total - cog * NCO;
    if (nco_here > NCO) nco_here = NCO;
    const int U = NT * nco_here;

    // units of this wave
    int xoff[WG_MAXU], goff[WG_MAXU];
#pragma unroll
    for (int i = 0; i < WG_MAXU; ++i) {
        const int u = wv + 4 * i;
        const int tap = u % NT, ct = u / NT;
        const int tz = tap / (KH * KW), ty = (tap / KW) % KH, tx = tap % KW;
        xoff[i] = ((tz * HY + ty) * HX + tx) * 32 + r;
        goff[i] = ct * 32 + r;
    }

    floatx16 acc[WG_MAXU];
#pragma unroll
    for (int i = 0; i < WG_MAXU; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;

    const int p_lo = (int)(((int64_t)sp * P) / S), p_hi = (int)(((int64_t)(sp + 1) * P) / S);
    const int c8x = tid & 7;  // channel quad (of 8) for X staging items
    // bias gradient db[co] = sum_v g[v][co]: the ci-tile-0 workgroups already hold every g tile in LDS
    constexpr int DR = 256 / GC;
    const bool do_db = (dbpart != nullptr) && (cit == 0) && (tid < DR * GC);
    const int dbc = tid % GC, dbr = tid / GC;
    float dbacc = 0.f;

    for (int pidx = p_lo; pidx < p_hi; ++pidx) {
        int q = pidx;
        const int ptx = q % nX;
        q /= nX;
        const int pty = q % nY;
        q /= nY;
        const int ptz = q % nZ;
        const int n = q / nZ;
        const int z0 = ptz * WG_TZ, y0 = pty * WG_TY, x0 = ptx * WG_TX;

        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) {
            sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + cit * 32 + c8x * 4);
            sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + cit * 32 + c8x * 4);
        }
        __syncthreads();  // previous patch's MFMA reads are done
        // X halo tile, batches of 4 x 16-byte loads per thread (bounded registers: the 7
        // accumulators own the register file; the co-resident workgroup hides this latency)
        constexpr int XB = 4;
#pragma unroll 1
        for (int it0 = 0; it0 < XIT; it0 += XB) {
            float4 t4[XB];
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int item = tid + (it0 + j) * 256;
                const int hv = item >> 3;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (hv < HV) {
                    const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
                    const int gz = z0 + hz - PZ, gy = y0 + hy - PY, gx = x0 + hx - PX;
                    if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                        v = *reinterpret_cast<const float4*>(x + ((((int64_t)n * D + gz) * H + gy) * W + gx) * x_ld +
                                                             cit * 32 + c8x * 4);
                        v.x = fmaf(v.x, sc4.x, sf4.x);
                        v.y = fmaf(v.y, sc4.y, sf4.y);
                        v.z = fmaf(v.z, sc4.z, sf4.z);
                        v.w = fmaf(v.w, sc4.w, sf4.w);
                    }
                }
                t4[j] = v;
            }
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int item = tid + (it0 + j) * 256;
                const int hv = item >> 3;
                if (hv < HV) *reinterpret_cast<float4*>(ldsX + hv * 32 + c8x * 4) = t4[j];
            }
        }
#pragma unroll 1
        for (int it0 = 0; it0 < GIT; it0 += XB) {
            float4 t4[XB];
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int item = tid + (it0 + j) * 256;
                const int pv = item / (GC / 4), cq = item % (GC / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pv < PV && cq < nco_here * 8) {
                    const int pz = pv / (WG_TY * WG_TX), py = (pv / WG_TX) % WG_TY, px = pv % WG_TX;
                    const int gz = z0 + pz, gy = y0 + py, gx = x0 + px;
                    if (gz < D && gy < H && gx < W)
                        v = *reinterpret_cast<const float4*>(g + ((((int64_t)n * D + gz) * H + gy) * W + gx) * g_ld +
                                                             cog * NCO * 32 + cq * 4);
                }
                t4[j] = v;
            }
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int item = tid + (it0 + j) * 256;
                const int pv = item / (GC / 4), cq = item % (GC / 4);
                if (pv < PV) *reinterpret_cast<float4*>(ldsG + pv * GC + cq * 4) = t4[j];
            }
        }
        __syncthreads();

        if (do_db)
            for (int pv = dbr; pv < PV; pv += DR) dbacc += ldsG[pv * GC + dbc];

        // K loop over voxel pairs: lanes 0-31 take voxel 2k, lanes 32-63 voxel 2k+1
        for (int zy = 0; zy < WG_TZ * WG_TY; ++zy) {
            const int pz = zy / WG_TY, py = zy % WG_TY;
            const int rowX = ((pz * HY + py) * HX) * 32;
            const int rowG = (zy * WG_TX) * GC;
#pragma unroll
            for (int xp = 0; xp < WG_TX / 2; ++xp) {
                const int px = 2 * xp + kh;
                const float* ax = ldsX + rowX + px * 32;
                const float* bg = ldsG + rowG + px * GC;
#pragma unroll
                for (int i = 0; i < WG_MAXU; ++i) {
                    if (wv + 4 * i < U) {
                        const float a = ax[xoff[i]];
                        const float b = bg[goff[i]];
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    if (dbpart != nullptr && cit == 0) {
        __syncthreads();
        if (do_db) lds[dbr * GC + dbc] = dbacc;
        __syncthreads();
        if (tid < nco_here * 32) {
            float a = 0.f;
            for (int rr = 0; rr < DR; ++rr) a += lds[rr * GC + tid];
            dbpart[(int64_t)sp * Cout + cog * NCO * 32 + tid] = a;
        }
    }
    // epilogue: D[row = ci][col = co]
#pragma unroll
    for (int i = 0; i < WG_MAXU; ++i) {
        const int u = wv + 4 * i;
        if (u < U) {
            const int tap = u % NT, ct = u / NT;
            float* dst = part + (((int64_t)sp * NT + tap) * Cin + cit * 32) * Cout + (cog * NCO + ct) * 32 + r;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                dst[(int64_t)row * Cout] = acc[i][reg];
            }
        }
    }
}

struct WgradPlan {
    int nco, T, S, P, nZ, nY, nX;
};

static WgradPlan wgrad_plan(int N, int D, int H, int W, int Cin, int Cout, int ntaps) {
    WgradPlan p;
    p.nco = ntaps == 27 ? 1 : (ntaps == 9 ? 3 : 4);
    int ncot = Cout / 32;
    int ngroups = (ncot + p.nco - 1) / p.nco;
    p.T = (Cin / 32) * ngroups;
    p.nZ = (D + WG_TZ - 1) / WG_TZ;
    p.nY = (H + WG_TY - 1) / WG_TY;
    p.nX = (W + WG_TX - 1) / WG_TX;
    int64_t P = (int64_t)N * p.nZ * p.nY * p.nX;
    p.P = (int)P;
    int64_t S = (1024 + p.T - 1) / p.T;
    // keep the partial slab below 256 MiB
    int64_t slab = (int64_t)ntaps * Cin * Cout * 4;
    int64_t cap = (256ll << 20) / slab;
    if (cap < 1) cap = 1;
    if (S > cap) S = cap;
    if (S > P) S = P;
    if (S < 1) S = 1;
    p.S = (int)S;
    return p;
}

int64_t tem_conv_wgrad_mfma_ws(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw) {
    WgradPlan p = wgrad_plan(N, D, H, W, Cin, Cout, kd * kh * kw);
    return (int64_t)p.S * kd * kh * kw * Cin * Cout * 4 + (int64_t)p.S * Cout * 4 + 256;
}

template <int KD, int KH, int KW, int NCO>
static void launch_wgrad(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                         int64_t g_ld, float* part, float* dbpart, int N, int D, int H, int W, int Cin, int Cout,
                         const WgradPlan& p, hipStream_t s) {
    constexpr int HV = (WG_TZ + KD - 1) * (WG_TY + KH - 1) * (WG_TX + KW - 1);
    constexpr int PV = WG_TZ * WG_TY * WG_TX;
    size_t ldsb = ((size_t)HV * 32 + (size_t)PV * 32 * NCO) * sizeof(float);
    hipLaunchKernelGGL((k_conv_wgrad_mfma<KD, KH, KW, NCO>), dim3((unsigned)(p.T * p.S)), dim3(256), ldsb, s, x, x_ld,
                       scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p.T, p.S, p.P, p.nZ, p.nY, p.nX);
}

int tem_conv_wgrad_mfma(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* g,
                        int64_t g_ld, float* dw, float* db, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                        int Cin, int Cout, int kd, int kh, int kw, int sd_layout, hipStream_t s) {
    TEM_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, "tem_conv3d_wgrad(mfma): needs Cin%%32==0 and Cout%%32==0 (got %d,%d)",
                Cin, Cout);
    TEM_REQUIRE(x_ld % 4 == 0 && g_ld % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)g % 16 == 0),
                "tem_conv3d_wgrad(mfma): x / g must be 16-byte aligned with ld%%4==0");
    TEM_REQUIRE(!scale || (((uintptr_t)scale % 16 == 0) && ((uintptr_t)shift % 16 == 0)),
                "tem_conv3d_wgrad(mfma): scale/shift must be 16-byte aligned");
    const int ntaps = kd * kh * kw;
    WgradPlan p = wgrad_plan(N, D, H, W, Cin, Cout, ntaps);
    if (ws_bytes < tem_conv_wgrad_mfma_ws(N, D, H, W, Cin, Cout, kd, kh, kw)) {
        tem_set_error("tem_conv3d_wgrad(mfma): workspace too small");
        return TEM_EWS;
    }
    float* part = (float*)ws;
    float* dbpart = db ? part + tem_align_up((int64_t)p.S * ntaps * Cin * Cout, 64) : nullptr;
    const int key = (kd == 3) * 4 + (kh == 3) * 2 + (kw == 3);
    if (key == 7)
        launch_wgrad<3, 3, 3, 1>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
    else if (key == 3)
        launch_wgrad<1, 3, 3, 3>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
    else if (key == 0)
        launch_wgrad<1, 1, 1, 4>(x, x_ld, scale, shift, g, g_ld, part, dbpart, N, D, H, W, Cin, Cout, p, s);
    else {
        tem_set_error("tem_conv3d_wgrad(mfma): kernel (%d,%d,%d) has no MFMA instantiation", kd, kh, kw);
        return TEM_EINVAL;
    }
    const int64_t n = (int64_t)ntaps * Cin * Cout;
    tem_reduce_slabs_w(part, p.S, ntaps, Cin, Cout, n, dw, sd_layout, s);
    if (db) tem_reduce_slabs(dbpart, p.S, Cout, Cout, db, s);
    return TEM_OK;
}
