// conv_pp.hip -- split-precision implicit-GEMM 3-D convolution (forward and data gradient), "ping-pong teams".
//
// Replaces aten::convolution / the dgrad half of convolution_backward behind ConvBlock (model/unet.py:417-438) for
// the layers with many patches (128^3 ... 32^3 levels).  Same arithmetic and operand layouts as k_conv_fwd_bfsplit
// (conv_bf16x3.hip): fp32 NDHWC activations, fused pre-norm while staging, operands split into two 16-bit terms,
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_{bf16,f16}, bias / ReLU / mask / statistics epilogue.
//
// What is different is the schedule.  In the one-patch-per-workgroup kernel every workgroup alternates between a
// staging phase (global loads -> norm -> split -> LDS, matrix pipe idle) and a 27-tap MFMA phase; identical
// workgroups started together stay in lockstep, so the co-resident workgroups of a CU stage at the same time and
// compete for the matrix pipe at the same time (measured: the phases add up, matrix pipe 41-45 % busy).  Here a
// workgroup is TWO teams of four waves -- one wave of each team per SIMD -- that are forced into opposite phases
// by the workgroup barrier: while team A runs the tap loop of its (patch, 16-channel chunk) step out of its LDS
// tile, team B writes back the previous patch, loads / normalises / splits the halo of its next step into its own
// tile and primes its weight-fragment ring; at the barrier they swap roles.  Each SIMD therefore always has exactly
// one wave issuing MFMAs, with the whole VALU / VMEM side of the partner hidden beside it, as long as a staging
// phase is shorter than a tap loop (it has 5-10 thousand cycles).  Workgroups are persistent (one per CU) and walk
// contiguous unit ranges per XCD.
#include "tem_common.h"
#include "conv_internal.h"
#include "conv_split.h"
#include "tem_act.h"
#include <type_traits>

#ifndef TEM_PP_RD
#define TEM_PP_RD 3      // weight-fragment ring depth over taps
#endif
#ifndef TEM_PP_PRIO
#define TEM_PP_PRIO 1    // s_setprio of the team in its MFMA phase
#endif
#ifndef TEM_PP_SCHED
#define TEM_PP_SCHED 1   // tap loop: 0 = loads of the next tap | MFMAs of this tap, pinned; 1 = one load between two MFMAs
#endif
#ifndef TEM_PP_LF
#define TEM_PP_LF -1     // staging order: 1 halo loads before the epilogue, 0 after it, -1 by tile size (see LOADS_FIRST)
#endif
#ifndef TEM_PP_ST_AUX
#define TEM_PP_ST_AUX 2  // cache policy of the epilogue stores: 2 = nt (the output is not re-read by this kernel), 0 = default
#endif
#ifndef TEM_PP_ABL
#define TEM_PP_ABL 0     // harness-only ablations (scripts/pp_harness.cpp): 1 no halo loads, 2 no stores, 4 no weight loads in
#endif                   // the tap loop, 8 no split / LDS writes, 16 no MFMAs

struct PpUnit {
    int cot, n, z0, y0, x0;
};
typedef float floatx4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer accesses: "buffer_load_dwordx4 v, voff, s[rsrc], soff offen" -- a wave-uniform 48-bit base in four SGPRs,
// ONE 32-bit VGPR byte offset per lane (a per-thread constant here) and a scalar byte offset.  Plain pointer arithmetic
// made hipcc build a 64-bit VGPR address per access (v_lshl_add_u64 + v_mov, two VGPRs each) and, short of registers,
// wait for the first loads of a step before it could form the addresses of the last ones (two memory latencies per step).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 pp_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    // whole-vector bit cast: __builtin_bit_cast(float, v.x) on the ELEMENTS makes hipcc (ROCm 7.2) narrow the access to one
    // buffer_load_dword and hand out element 0 four times
    const floatx4s f = __builtin_bit_cast(floatx4s, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ uint4 pp_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pp_store4_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float4 v) {
    const u32x4 d = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                     __builtin_bit_cast(unsigned, v.w)};
    // soffset stays an immediate: with an SGPR there LLVM skips the wait state a VALU write of the data registers needs
    // after a 16-byte store on gfx950 (see zr_store4 in conv_zr.hip)
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff + soff, 0, TEM_PP_ST_AUX);
}

#ifdef TEM_PP_TRACE   // developer build (scripts/pp_harness.cpp): shader-clock stamps of the phases of one workgroup
#ifndef TEM_PP_TRACE_BLOCK
#define TEM_PP_TRACE_BLOCK 0
#endif
__device__ unsigned long long tem_pp_trace_buf[2][64][8];
#define PP_STAMP(i)                                                                           \
    do {                                                                                      \
        if (blockIdx.x == TEM_PP_TRACE_BLOCK && tw == 0 && lane == 0 && s < 64)               \
            tem_pp_trace_buf[team][s][i] = __builtin_amdgcn_s_memtime();                      \
    } while (0)
void tem_pp_trace_read(unsigned long long* dst) {
    (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tem_pp_trace_buf), sizeof(unsigned long long) * 2 * 64 * 8);
}
#else
#define PP_STAMP(i)
#endif

// KD,KH,KW kernel; TZ,TY,TX voxel patch of a TEAM (TX == 8); CT 32-column tiles per team; WN waves side by side over
// the columns (WM = 4 / WN waves over the voxels); NS planes per operand; F16: fp16 terms, lo planes stored x 2^12 and the
// cross products hi*lo' + lo'*hi in their own accumulators (TEM_WL_F16X3, conv_split.h), else bf16 terms.
// (A single-accumulator variant with prescaled operands -- TEM_WL_F16X3S -- was built and measured: the matrix core
// truncates small addends against a large accumulator, a one-sided error of ~1e-5 per output that the sums of a
// GroupNorm backward turn into 4e-3; rejected, the layout stays available to the patch kernel for experiments.)
//
// Instruction budget.  A staging phase runs on ONE wave per SIMD (its partner issues MFMAs), so nothing hides its issue
// latency: the first version spent ~1000 VALU + ~1000 SALU instructions per phase on index arithmetic (halo coordinates,
// bounds, 64-bit addresses, one 4-byte store per accumulator register) and was bound by exactly that -- the kernel took
// the same time with the MFMAs removed.  Hence: halo offsets are per-thread constants computed once; a load is
// "global_load_dwordx4 v, voff, s[base]"; interior patches (72 % at 128^3) skip every bounds operation; the unit is decoded
// once per patch, not per step; and the epilogue transposes each 32 x 32 accumulator tile through a wave-private LDS
// scratch so that a lane stores 16 bytes (4 channels of one voxel): 4 coalesced stores per tile instead of 16 partial ones.
template <int KD, int KH, int KW, int TZ, int TY, int TX, int CT, int WN, int NS, bool F16>
__global__ __launch_bounds__(512, 2) void k_conv_pp(
    const float* __restrict__ x, int64_t x_ld, const float* __restrict__ scale, const float* __restrict__ shift,
    const uint4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int64_t y_ld,
    const float* __restrict__ ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout, int act, int nZ,
    int nY, int nX, float* __restrict__ stat, int nunits) {
    constexpr int NT = KD * KH * KW;
    constexpr int PZ = KD / 2, PY = KH / 2, PX = KW / 2;
    constexpr int HZ = TZ + KD - 1, HY = TY + KH - 1, HX = TX + KW - 1;
    constexpr int HV = HZ * HY * HX;
    constexpr int PV = TZ * TY * TX;
    constexpr int WM = 4 / WN;
    constexpr int MT = PV / (32 * WM);          // 32-voxel M-tiles per wave
    constexpr int NW = CT / WN;                 // 32-column tiles per wave
    constexpr int NIT = (HV * 4 + 255) / 256;   // float4 slots (4 per halo voxel and chunk) per thread of a team
    constexpr int LSV = NS * 8 + 4;             // LDS floats per halo voxel: NS planes of 16 x 16 bit + 16 B pad
    constexpr int RD = NT >= TEM_PP_RD ? TEM_PP_RD : 1;  // weight ring: the slot of a tap is tap % RD (the ring restarts every phase)
    constexpr int FR = NS * 64;                 // uint4s per (tap, 16-channel chunk) fragment group
    constexpr int SCP = 36;                     // floats per row of the transpose scratch (32 + 4: 16-byte aligned rows)
    static_assert(PV % (32 * WM) == 0 && CT % WN == 0 && WN * WM == 4 && TX == 8, "team tiling");
    extern __shared__ __attribute__((aligned(16))) float lds_all[];  // [2 teams][HV][LSV] tiles, [8 waves][32][SCP] scratch

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wv >> 2, tw = wv & 3, tl = tid & 255;
    const int kh = lane >> 5, r = lane & 31;
    const int wm = tw % WM, wn = tw / WM;
    float* lds = lds_all + team * (HV * LSV);
    float* scr = lds_all + 2 * HV * LSV + wv * (32 * SCP);

    // units: (column group, patch) pairs; a team takes every G-th unit starting at its logical slot.  Logical slots
    // are contiguous per XCD (dispatch places block b on XCD b % 8), so the 64 units an XCD works on at any time are
    // neighbouring patches that share halo lines in that XCD's L2.
    const int G = 2 * gridDim.x;
    const int slot = tem_xcd_remap(blockIdx.x, gridDim.x) * 2 + team;
    const int ncot = Cout / (32 * CT);
    const int nch = Cin >> 4;
    const int my_units = slot < nunits ? (nunits - slot + G - 1) / G : 0;
    const int P = ((nunits + G - 1) / G) * nch;  // steps of the busiest team: every wave runs 2P + 3 barriers

    auto decode = [&](int ui) {
        int u = slot + ui * G;
        PpUnit t;
        t.cot = u % ncot; u /= ncot;
        t.x0 = (u % nX) * TX; u /= nX;
        t.y0 = (u % nY) * TY; u /= nY;
        t.z0 = (u % nZ) * TZ; u /= nZ;
        t.n = u;
        return t;
    };

    // ---- per-thread constants ----
    // staging: slot it of this thread is halo voxel hv0 + 64 it, channels c4*4 .. c4*4+3 of the chunk
    const int c4 = tl & 3, hv0 = tl >> 2;
    unsigned hoff[NIT];             // BYTE offset from the halo origin (z0-PZ, y0-PY, x0-PX)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int hv = min(hv0 + 64 * it, HV - 1);
        const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
        hoff[it] = ((unsigned)((hz * H + hy) * W + hx) * (unsigned)x_ld + (unsigned)(c4 * 4)) * 4u;
    }
    const unsigned ctr_off = ((unsigned)((PZ * H + PY) * W + PX) * (unsigned)x_ld + (unsigned)(c4 * 4)) * 4u;  // always inside
    // tap loop: A-fragment rows of this lane
    int abase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int p = (wm * MT + m) * 32 + r;
        const int pz = p / (TY * TX), py = (p / TX) % TY, px = p % TX;
        abase[m] = ((pz * HY + py) * HX + px) * LSV + kh * 4;  // + 8 floats (32 B) per further plane
    }
    // epilogue: after the transpose a lane holds channels cq*4 .. cq*4+3 of the voxels (x = rr, y-row j) of a tile
    const int rr = lane >> 3, cq = lane & 7;
    float* scr_w = scr + (4 * kh) * SCP + r;     // accumulator register reg -> row (reg & 3) + 8 (reg >> 2) + 4 kh, column r
    const float* scr_r = scr + rr * SCP + cq * 4;
    const unsigned yoff_lane = ((unsigned)rr * (unsigned)y_ld + (unsigned)(cq * 4)) * 4u;     // bytes
    const unsigned roff_lane = ((unsigned)rr * (unsigned)ref_ld + (unsigned)(cq * 4)) * 4u;
    const __amdgpu_buffer_rsrc_t rw = pp_rsrc(wp);
    const unsigned woff_lane = (unsigned)lane * 16u;

    // ReLU / no activation as one v_max (the sigmoid of a final activation never sits behind these layers: the host side
    // sends such a launch to the patch kernel)
    const float act_floor = act == TEM_ACT_RELU ? 0.f : -__builtin_inff();
    constexpr bool SC = F16 && NS == 2;   // second accumulator set for the scaled cross products
    floatx16 acc[MT][NW];
    floatx16 accl[SC ? MT : 1][SC ? NW : 1];
    const int tapstride = nch * FR;
    uint4 bq[RD][NW][NS];
    unsigned wsoff[NW];
#pragma unroll
    for (int nn = 0; nn < NW; ++nn) wsoff[nn] = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < NW; ++nn)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[m][nn][i] = 0.f;
                if (SC) accl[SC ? m : 0][SC ? nn : 0][i] = 0.f;
            }

    // step cursor of this team: (unit index, chunk), advanced once per loop trip; `cu` is the unit being staged /
    // computed, `eu` the one whose accumulators wait for their epilogue
    int ui = 0, ci = 0;
    PpUnit cu = decode(0), eu = cu;
    bool epi_pending = false;
    float bvc[NW], bve[NW];   // bias of this lane's column in the unit being computed / awaiting its epilogue
#pragma unroll
    for (int nn = 0; nn < NW; ++nn) {
        bvc[nn] = (bias && my_units > 0) ? bias[((cu.cot * WN + wn) * NW + nn) * 32 + r] : 0.f;
        bve[nn] = 0.f;
    }

    // Both teams run the same straight-line sequence  stage(s) | barrier | taps(s) | barrier ;  team 1 passes one extra
    // barrier first (and team 0 one at the end), which shifts it by one phase: its staging runs beside team 0's tap loop.
    if (team) __syncthreads();
    for (int s = 0; s <= P; ++s) {
        const bool do_stage = ui < my_units;
        {
            // ================= staging phase (the partner team runs its tap loop) =================
            PP_STAMP(0);
            // ---- issue the halo loads of this step first: they fly during the epilogue of the previous unit ----
            float4 tmp[NIT];
            unsigned inb = 0xffffffffu;
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
            bool interior = true;
            // With 2 M-tiles per wave the loads go first and fly during the epilogue; with 4 the tap loop of the partner is
            // twice as long as a staging phase, so the epilogue runs first and the 40 load registers are not live beside it.
            constexpr bool LOADS_FIRST = TEM_PP_LF < 0 ? (MT <= 2) : (TEM_PP_LF != 0);
            auto issue_loads = [&]() {
                if (scale) {
                    sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)cu.n * Cin + ci * BCK + c4 * 4);
                    sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)cu.n * Cin + ci * BCK + c4 * 4);
                }
                // one uniform base pointer per step (the halo origin: may lie outside the tensor for border patches, where
                // only in-range voxels are dereferenced) + per-thread constant 32-bit offsets
                const float* xb = x + ((((int64_t)cu.n * D + (cu.z0 - PZ)) * H + (cu.y0 - PY)) * W + (cu.x0 - PX)) * x_ld + ci * BCK;
                const __amdgpu_buffer_rsrc_t rx = pp_rsrc(xb);
                interior = (cu.z0 >= PZ) & (cu.z0 + HZ - PZ <= D) & (cu.y0 >= PY) & (cu.y0 + HY - PY <= H) & (cu.x0 >= PX) &
                           (cu.x0 + HX - PX <= W);
                if (interior) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        if (TEM_PP_ABL & 1)
                            tmp[it] = make_float4(0.5f + it, 0.25f, -1.f, 2.f);
                        else
                            tmp[it] = pp_load4(rx, hoff[it], 0);
                    }
                } else {
                    inb = 0;
                    const unsigned zlim = (unsigned)D, ylim = (unsigned)H, xlim = (unsigned)W;
                    const int bz = cu.z0 - PZ, by = cu.y0 - PY, bx = cu.x0 - PX;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        // halo coordinates recomputed here (border patches only) rather than kept in registers
                        const int hv = min(hv0 + 64 * it, HV - 1);
                        const int hz = hv / (HY * HX), rem = hv % (HY * HX), hy = rem / HX, hx = rem % HX;
                        const unsigned gz = (unsigned)(bz + hz), gy = (unsigned)(by + hy), gx = (unsigned)(bx + hx);
                        const bool ok = (gz < zlim) & (gy < ylim) & (gx < xlim);   // unsigned: negative coordinates are huge
                        inb |= ok ? (1u << it) : 0u;
                        tmp[it] = pp_load4(rx, ok ? hoff[it] : ctr_off, 0);
                    }
                }
            };
            if (LOADS_FIRST && do_stage) issue_loads();
            PP_STAMP(1);
            // ---- epilogue of the unit whose last chunk this team computed in its previous phase ----
            if (epi_pending) {
                const __amdgpu_buffer_rsrc_t ry = pp_rsrc(y + ((((int64_t)eu.n * D + eu.z0) * H + eu.y0) * W + eu.x0) * y_ld);
                const bool has_ref = ref != nullptr;
                const __amdgpu_buffer_rsrc_t rr_ = pp_rsrc(has_ref ? ref + ((((int64_t)eu.n * D + eu.z0) * H + eu.y0) * W + eu.x0) * ref_ld : y);
                const bool full = (eu.z0 + TZ <= D) & (eu.y0 + TY <= H) & (eu.x0 + TX <= W);
                // FULL: the patch lies inside the volume (no per-element bounds); else its outside voxels count as zero in
                // the statistics and are not stored.  One uniform branch, two straight-line bodies.
                auto body = [&](auto full_tag, auto ref_tag) {
                    constexpr bool FULL = decltype(full_tag)::value, HASREF = decltype(ref_tag)::value;
#pragma unroll
                    for (int nn = 0; nn < NW; ++nn) {
                        const int cb = ((eu.cot * WN + wn) * NW + nn) * 32;   // first channel of this column tile
                        const float bv = bve[nn];   // loaded when the unit was decoded: a load issued HERE would sit behind the
                                                    // halo loads in vmcnt order and stall the epilogue for their whole latency
                        float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};     // two chains: the adds of a tile are independent
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const int yy0 = (wm * MT + m) * 4;   // combined (z, y) row index of the tile's first y-row
#pragma unroll
                            for (int reg = 0; reg < 16; ++reg) {
                                float a = acc[m][nn][reg];
                                if (SC) {
                                    a = fmaf(accl[SC ? m : 0][SC ? nn : 0][reg], 1.f / F16_LO_SCALE, a);
                                    accl[SC ? m : 0][SC ? nn : 0][reg] = 0.f;
                                }
                                float o = fmaxf(a + bv, act_floor);
                                if (!FULL) {
                                    const int yy = yy0 + (reg >> 2);
                                    const bool ok = (eu.z0 + yy / TY < D) & (eu.y0 + yy % TY < H) & (eu.x0 + (reg & 3) + 4 * kh < W);
                                    o = ok ? o : 0.f;
                                }
                                ssum[reg & 1] += o;
                                ssq[reg & 1] = fmaf(o, o, ssq[reg & 1]);
                                scr_w[((reg & 3) + 8 * (reg >> 2)) * SCP] = o;
                                acc[m][nn][reg] = 0.f;  // the next unit of this team starts from zero
                            }
                            // transposed read-back: 4 x (8 voxels x 128 B) per tile, one 16-byte store per lane each
                            float4 v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(scr_r + 8 * j * SCP);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int yy = yy0 + j, pz = yy / TY, py = yy % TY;
                                const unsigned srow = (unsigned)((pz * H + py) * W);   // scalar: voxel offset of this y-row
                                bool ok = true;
                                if (!FULL) ok = (eu.z0 + pz < D) & (eu.y0 + py < H) & (eu.x0 + rr < W);
                                if (HASREF) {
                                    const float4 q = pp_load4(rr_, (FULL || ok) ? roff_lane + (unsigned)cb * 4u : 0u,
                                                              (FULL || ok) ? srow * (unsigned)ref_ld * 4u : 0u);
                                    v[j].x = q.x > 0.f ? v[j].x : 0.f;
                                    v[j].y = q.y > 0.f ? v[j].y : 0.f;
                                    v[j].z = q.z > 0.f ? v[j].z : 0.f;
                                    v[j].w = q.w > 0.f ? v[j].w : 0.f;
                                }
                                if ((FULL || ok) && (!(TEM_PP_ABL & 2) || v[j].x == 12345.678f))
                                    pp_store4_nt(ry, yoff_lane + (unsigned)cb * 4u, srow * (unsigned)y_ld * 4u, v[j]);
                            }
                        }
                        if (stat) {  // grid-uniform: per (sample, patch, voxel-wave, channel) partial sums of the stored output
                            float sa = ssum[0] + ssum[1], sb = ssq[0] + ssq[1];
                            sa += __shfl_xor(sa, 32, 64);  // the lane halves hold different rows of one column
                            sb += __shfl_xor(sb, 32, 64);
                            if (kh == 0) {
                                const int64_t patch = ((int64_t)(eu.z0 / TZ) * nY + eu.y0 / TY) * nX + eu.x0 / TX;
                                const int64_t nblk = (int64_t)nZ * nY * nX * WM;
                                float* dst = stat + (((int64_t)eu.n * nblk + patch * WM + wm) * Cout + cb + r) * 2;
                                dst[0] = sa;
                                dst[1] = sb;
                            }
                        }
                    }
                };
                if (full) {
                    if (has_ref) body(std::true_type{}, std::true_type{});
                    else body(std::true_type{}, std::false_type{});
                } else {
                    if (has_ref) body(std::false_type{}, std::true_type{});
                    else body(std::false_type{}, std::false_type{});
                }
                epi_pending = false;
            }
            if (!LOADS_FIRST && do_stage) issue_loads();
            PP_STAMP(6);
            // ---- norm, split, LDS tile; then prime the weight-fragment ring of the coming tap loop ----
            if (do_stage) {
                float* lw = lds + hv0 * LSV + c4 * 2;
                auto convert = [&](auto interior_tag) {
                    constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        if (hv0 + 64 * it < HV) {
                            float e[4] = {fmaf(tmp[it].x, sc4.x, sf4.x), fmaf(tmp[it].y, sc4.y, sf4.y),
                                          fmaf(tmp[it].z, sc4.z, sf4.z), fmaf(tmp[it].w, sc4.w, sf4.w)};
                            if (!INTERIOR) {  // zero padding comes after the norm (model/unet.py:429-438)
                                const bool ok = (inb >> it) & 1u;
#pragma unroll
                                for (int c = 0; c < 4; ++c) e[c] = ok ? e[c] : 0.f;
                            }
                            if (F16 && NS == 2) {   // fp16x3: range clamp; the one-term mixed mode overflows to inf like autocast
#pragma unroll
                                for (int c = 0; c < 4; ++c) e[c] = __builtin_amdgcn_fmed3f(e[c], -60000.f, 60000.f);
                            }
                            if (F16 && NS == 1) {
                                // the mixed mode rounds the fp32 pre-norm result to fp16 (what autocast does to the fp32 output
                                // of a norm layer): keep hipcc from fusing the two into one v_fma_mix*_f16, which rounds the
                                // exact product-sum once and differs from the patch kernel by one fp16 ulp in 1 of 2^17 elements
#pragma unroll
                                for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(e[c]));
                            }
#pragma unroll
                            for (int p = 0; p < NS; ++p) {
                                const unsigned h0 = pk16<F16>(e[0], e[1]), h1 = pk16<F16>(e[2], e[3]);
                                if (!(TEM_PP_ABL & 8) || h0 == 0x12345678u)
                                    *reinterpret_cast<uint2*>(lw + 64 * it * LSV + p * 8) = make_uint2(h0, h1);
                                if (p + 1 < NS) {
                                    e[0] -= lo16<F16>(h0);
                                    e[1] -= hi16<F16>(h0);
                                    e[2] -= lo16<F16>(h1);
                                    e[3] -= hi16<F16>(h1);
                                    if (SC) {
#pragma unroll
                                        for (int c = 0; c < 4; ++c) e[c] *= F16_LO_SCALE;
                                    }
                                }
                            }
                        }
                    }
                };
                if (interior)
                    convert(std::true_type{});
                else
                    convert(std::false_type{});
                PP_STAMP(7);
#pragma unroll
                for (int nn = 0; nn < NW; ++nn)   // scalar byte offset of (column tile, tap 0, chunk ci) in the packed weights
                    wsoff[nn] = (unsigned)((((cu.cot * WN + wn) * NW + nn) * NT * nch + ci) * FR) * 16u;
                if (RD > 1) {
#pragma unroll
                    for (int gp = 0; gp < RD - 1; ++gp)
#pragma unroll
                        for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                            for (int p = 0; p < NS; ++p)
                                bq[gp][nn][p] = pp_load4u(rw, woff_lane, wsoff[nn] + (unsigned)(gp * tapstride + p * 64) * 16u);
                }
            }
            PP_STAMP(2);
        }
        __syncthreads();
        {
            PP_STAMP(3);
            // ================= MFMA phase: 27 taps of one 16-channel chunk out of this team's tile =================
            if (do_stage) {
                int ts = tapstride;
                asm volatile("" : "+s"(ts));
                if (TEM_PP_PRIO) __builtin_amdgcn_s_setprio(TEM_PP_PRIO);
                // Software pipeline, pinned with sched_barrier: this wave is the only one of its SIMD that issues MFMAs in
                // this phase, so nothing else hides its LDS / L2 latencies.  While the MFMAs of tap t run, the A fragments
                // of tap t+1 are already on their way from LDS (double-buffered registers) and the weight fragments of tap
                // t+RD-1 on their way from L2 (ring of RD slots; slots 0..RD-2 were primed by the staging phase).
                uint4 af[2][MT][NS];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int p = 0; p < NS; ++p) af[0][m][p] = *reinterpret_cast<const uint4*>(lds + abase[m] + p * 8);
#pragma unroll
                for (int tap = 0; tap < NT; ++tap) {
                    if (tap + 1 < NT) {
                        const int t1 = tap + 1;
                        const int tz = t1 / (KH * KW), ty = (t1 / KW) % KH, tx = t1 % KW;
                        const int toff = ((tz * HY + ty) * HX + tx) * LSV;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int p = 0; p < NS; ++p)
                                af[t1 & 1][m][p] = *reinterpret_cast<const uint4*>(lds + abase[m] + toff + p * 8);
                    }
                    if (tap + RD - 1 < NT && !(TEM_PP_ABL & 4)) {
                        const int gp = tap + RD - 1;
#pragma unroll
                        for (int nn = 0; nn < NW; ++nn)
#pragma unroll
                            for (int p = 0; p < NS; ++p)
                                bq[gp % RD][nn][p] = pp_load4u(rw, woff_lane, wsoff[nn] + (unsigned)(gp * ts + p * 64) * 16u);
                    }
                    if (TEM_PP_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nn = 0; nn < NW; ++nn) {
                            // smallest terms first: plane pairs (i, j) with i + j <= NS - 1
#pragma unroll
                            for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                                for (int i = 0; i <= sum; ++i) {
                                    if (TEM_PP_ABL & 16) {
                                        asm volatile("" ::"v"(af[tap & 1][m][i].x), "v"(af[tap & 1][m][i].w),
                                                     "v"(bq[tap % RD][nn][sum - i].x), "v"(bq[tap % RD][nn][sum - i].w));
                                        continue;
                                    }
                                    if (SC && sum == 1)
                                        accl[SC ? m : 0][SC ? nn : 0] = mfma16<F16>(af[tap & 1][m][i], bq[tap % RD][nn][sum - i],
                                                                                    accl[SC ? m : 0][SC ? nn : 0]);
                                    else
                                        acc[m][nn] = mfma16<F16>(af[tap & 1][m][i], bq[tap % RD][nn][sum - i], acc[m][nn]);
                                }
                        }
                    if (TEM_PP_SCHED == 1) {
                        // an in-order wave issues the next tap's loads in the shadow of this tap's MFMAs only if they sit
                        // BETWEEN them: a block of 6+ memory instructions after the last MFMA outlasts its 32 cycles
                        constexpr int NM = MT * NW * (NS * (NS + 1) / 2), NL = MT * NS, NV = NW * NS;
#pragma unroll
                        for (int k = 0; k < NM; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (k < NL)
                                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            else if (k - NL < NV)
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (TEM_PP_PRIO) __builtin_amdgcn_s_setprio(0);
            }
            PP_STAMP(4);
        }
        // advance the cursor: the unit is complete after its last chunk (its epilogue runs in the next staging phase)
        if (do_stage) {
            if (++ci == nch) {
                ci = 0;
                eu = cu;
                epi_pending = true;
#pragma unroll
                for (int nn = 0; nn < NW; ++nn) bve[nn] = bvc[nn];
                if (++ui < my_units) {
                    cu = decode(ui);
                    if (bias) {
#pragma unroll
                        for (int nn = 0; nn < NW; ++nn) bvc[nn] = bias[((cu.cot * WN + wn) * NW + nn) * 32 + r];
                    }
                }
            }
        }
        __syncthreads();
        PP_STAMP(5);
    }
    if (!team) __syncthreads();
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct PpGeom {
    int variant;  // 0: not handled here; 1: ping-pong teams, 4 x 8 x 8 voxel patch per team
    int TZ, TY, TX, CT, WM;
    int nZ, nY, nX;
    int64_t nunits;
};

// Which shapes run on the ping-pong kernel: 3x3x3 (and 1x3x3 with depth) kernels, two 16-bit planes per operand, and
// enough (patch, column group) units to give every team of every CU at least one.
static PpGeom pp_geometry(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit, int64_t max_ld) {
    PpGeom g = {};
    const long long opt = tem_option(TEM_OPT_CONV_FWD_VARIANT);
    if (opt == 0) return g;   // (2 = z-reuse kernel forced: shapes it does not take still come here)
    if (tem_call_st.x || tem_call_st.y) return g;   // 16-bit activation storage: the z-reuse / patch kernels carry the element type
    if (!(nsplit == 2 || nsplit == 4 || nsplit == 5 || nsplit == 7)) return g;   // bf16x3, fp16x3 (scaled lo), one fp16 / bf16 term (mixed modes)
    if (!(kh == 3 && kw == 3 && (kd == 3 || kd == 1))) return g;
    if (D < 4 || Cin % 16 || Cout % 32) return g;
    if ((int64_t)H * W * 8 * 4 * max_ld >= (1ll << 31)) return g;  // 32-bit byte offsets inside one halo / one patch
    static int ncu = 0;
    if (!ncu) {
        ncu = tem_device_cus();
        if (ncu <= 0) ncu = 256;
    }
    g.CT = (Cout % 64 == 0) ? 2 : 1;
    g.TZ = 4;
    g.TY = g.TX = 8;
    g.WM = (g.CT == 2) ? 2 : 4;
    g.nZ = (D + g.TZ - 1) / g.TZ;
    g.nY = (H + 7) / 8;
    g.nX = (W + 7) / 8;
    g.nunits = (int64_t)N * g.nZ * g.nY * g.nX * (Cout / (32 * g.CT));
    const long long minu = tem_option(TEM_OPT_TEAM_MIN_UNITS);
    if (g.nunits < (opt == 1 ? 1 : (minu > 0 ? minu : 2ll * ncu))) return g;
    g.variant = 1;
    return g;
}

int64_t tem_conv_pp_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit) {
    const PpGeom g = pp_geometry(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit, 1);
    if (!g.variant) return -1;
    return (int64_t)g.nZ * g.nY * g.nX * g.WM;
}

// 32-column tiles per team of the instantiation this shape selects (1 or 2), 0 when the shape is not handled here
int tem_conv_pp_tiles(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit) {
    const PpGeom g = pp_geometry(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit, 1);
    return g.variant ? g.CT : 0;
}

template <int KD, int KH, int KW, int TZ, int CT, bool F16, int NS = 2>
static void pp_launch(const PpGeom& g, const float* x, int64_t x_ld, const float* scale, const float* shift,
                      const float* wp, const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld,
                      int N, int D, int H, int W, int Cin, int Cout, int act, float* stat, hipStream_t s) {
    constexpr int WN = CT;  // 64-column teams: 2 x 2 waves; 32-column teams: 4 x 1
    constexpr int HV = (TZ + KD - 1) * 10 * 10;
    constexpr size_t ldsb = ((size_t)2 * HV * (NS * 8 + 4) + 8 * 32 * 36) * sizeof(float);  // two halo tiles + 8 wave scratches
    static_assert(ldsb <= 160 * 1024, "LDS budget");
    auto kern = &k_conv_pp<KD, KH, KW, TZ, 8, 8, CT, WN, NS, F16>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        attr = true;
    }
    static int ncu = 0;
    if (!ncu) {
        ncu = tem_device_cus();
        if (ncu <= 0) ncu = 256;
    }
    int64_t grid = (g.nunits + 1) / 2;
    if (grid > ncu) grid = ncu;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), ldsb, s, x, x_ld, scale, shift,
                       reinterpret_cast<const uint4*>(wp), bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, g.nZ,
                       g.nY, g.nX, stat, (int)g.nunits);
}

// -> 1 when the launch was taken, 0 when the shape belongs to another kernel, -1 when the caller sized a statistics
// buffer for this kernel (tem_conv_pp_stat_blocks) but an alignment condition of the launch fails: falling through to the
// patch kernel would write a differently shaped partials buffer (tem_last_error is set)
int tem_conv_fwd_pp(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                     const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H,
                     int W, int Cin, int Cout, int kd, int kh, int kw, int act, int nsplit, float* stat, hipStream_t s) {
    int64_t max_ld = x_ld > y_ld ? x_ld : y_ld;
    if (ref && ref_ld > max_ld) max_ld = ref_ld;
    const PpGeom g = pp_geometry(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit, max_ld);
    if (!g.variant) return 0;
    // 16-byte epilogue accesses; statistics of a masked output are the patch kernel's business (never asked for together)
    if ((y_ld % 4) || ((uintptr_t)y % 16) || (ref && ((ref_ld % 4) || ((uintptr_t)ref % 16))) || (stat && ref) ||
        act == TEM_ACT_SIGMOID) {
        if (stat) {
            tem_set_error("tem_conv3d_fwd_stats: statistics were sized for the ping-pong kernel but this launch cannot take it "
                          "(y / ref need 16-byte alignment and ld %% 4 == 0, no ref, no sigmoid)");
            return -1;
        }
        return 0;
    }
    const bool f16 = nsplit == 4;
#define PPGO(KD, CT)                                                                                                  \
    do {                                                                                                              \
        if (nsplit == 5)                                                                                              \
            pp_launch<KD, 3, 3, 4, CT, true, 1>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, \
                                                Cout, act, stat, s);                                                  \
        else if (nsplit == 7)                                                                                         \
            pp_launch<KD, 3, 3, 4, CT, false, 1>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, \
                                                 Cout, act, stat, s);                                                 \
        else if (f16)                                                                                                 \
            pp_launch<KD, 3, 3, 4, CT, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin,  \
                                             Cout, act, stat, s);                                                     \
        else                                                                                                          \
            pp_launch<KD, 3, 3, 4, CT, false>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, \
                                              Cout, act, stat, s);                                                    \
    } while (0)
    if (kd == 3) {
        if (g.CT == 1) PPGO(3, 1);
        else PPGO(3, 2);
    } else {
        if (g.CT == 1) PPGO(1, 1);
        else PPGO(1, 2);
    }
#undef PPGO
    return 1;
}
