// conv_wgrad_tr.hip -- 3x3x3 weight gradient (the wgrad half of aten::convolution_backward behind nn.Conv3d,
// /root/reference/torch_em/model/unet.py:429-438): z-sliding column, staging team, HARDWARE-TRANSPOSED operand reads.
//
// dW[tap][ci][co] = sum_v x^[v + tap][ci] * g[v][co] is a GEMM whose K dimension is the VOXEL index, so both MFMA
// operands want 8 consecutive voxels of ONE channel per lane -- the transpose of the channels-last tensors in HBM.  The
// round-2 / round-3 kernels (k_conv_wgrad_zs / _zt, conv_bf16x3.hip) paid for that transpose on the vector ALU: channel-major
// 16-bit planes in LDS written with one ds_write_b32 per (voxel pair, channel), x-shifted windows by v_alignbyte in the
// multiplying waves -- 5.5 VALU instructions per MFMA (profiles/r03_pmc_stalls.txt), and the power samples of round 4
// (profiles/r04_power_*.txt) show the kernel at the 1.4 kW socket limit while its matrix pipe is 0.53 busy: the watts go
// into that staging work.  gfx950 has the transpose in the LDS read path: ds_read_b64_tr_b16 hands lane i of a 16-lane
// group element (i & 3) of the 8-byte items of lanes 4 j + (i >> 2), j = 0..3 -- a 4 x 16 transpose of 16-bit values.  So:
//   * LDS holds VOXEL-major records, 32 channels x 16 bit = 64 B per voxel and term, exactly as the staging thread has
//     them after its 16-byte global load: 4 channels of a voxel -> norm -> convert -> ONE ds_write_b64 per term;
//   * a k-slab of the MFMA = two x-rows of 8 voxels (lanes 0..31: row 2 s, lanes 32..63: row 2 s + 1); a fragment read is two
//     ds_read_b64_tr_b16 (voxels x .. x+3 and x+4 .. x+7 of the row); the 32 lanes of a half address 4 consecutive
//     records = 256 contiguous bytes: conflict free for every tap shift (a tap is an address offset: no v_alignbyte, no
//     17th element);
//   * roles as in k_conv_wgrad_zt: waves 0..3 (one per SIMD) only multiply -- wave w owns row groups (tz, ty) w and w + 4 with
//     their three tx taps and one tap of row group 8 (7 / 7 / 7 / 6 accumulator tiles of one (Cin tile, Cout tile) pair); the
//     tx = 0 and tx = 2 fragments of a row group come out of ONE 12-voxel window (three transposing reads), tx = 1 from two
//     more (the first version read every tap's fragment separately: 14 reads per 7 MFMAs, 1.6 GHz -- LDS reads cost power;
//     shifting tx = 1 out of the window with 4 v_alignbyte instead costs issue slots: a SIMD fits ~5 non-MFMA
//     instructions of BOTH its waves per MFMA) -- waves 4..7 only stage (x planes into a 5-slot ring, g planes
//     into 3 buffers, two planes ahead of the multiplication; global loads four planes ahead); one barrier per plane; persistent over column segments, one
//     partial slab per workgroup (format of k_conv_wgrad_zt: KS2 = 1).
// ARITH: 0 bf16x3 (x^ and g two bf16 terms, 3 MFMAs per product), 1 one fp16 term each (mixed precision), 2 one bf16 term
// each, 3 fp16 2x1 (x^ two fp16 terms, g one fp16 term prescaled from *g_amax: the default of the fp32-class mode),
// 4 exact fp32 on v_mfma_f32_32x32x2_f32 (TEM_PRECISION=fp32): K = 2 voxels = the two lane halves, a lane holds ONE
// channel value per operand, so fp32 voxel-major records (128 B) ARE the operand layout -- plain ds_read_b32, no transpose.
#include "tem_common.h"
#include "conv_internal.h"
#include "conv_split.h"
#include "tem_act.h"
#include <set>
#include <type_traits>

#define TR_REC16 64                // bytes per voxel record: 32 channels x 16 bit (ARITH 4, exact fp32: 128)
#define TR_NXS 5                   // ring slots of x^ halo planes: 3 being multiplied + 1 prefetched by the multiplying team + 1 being written
#define TR_NGS 3                   // g planes: multiplied, prefetched, being written
#ifndef TR_UNROLL
#define TR_UNROLL 8                // iterations per trip of the 16-bit staging loop (even: two register sets alternate)
#endif
#ifndef TR_UNROLL32
#define TR_UNROLL32 4              // the same for the fp32-tensor staging loop (round 6; 8 and 16 measured: more idle padding at D = 32)
#endif
#define TR_XT_OF(rec) (TR_NXS * 100 * (rec))   // one term of x^: ring of 10 x 10 halo planes
#define TR_GT_OF(rec) (TR_NGS * 64 * (rec))    // one term of g: 8 x 8 planes

typedef short tr_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_s4* tr_lds_p;
typedef __amdgpu_buffer_rsrc_t tr_rsrc_t;
typedef float tr_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int tr_u4 __attribute__((ext_vector_type(4)));

// The per-plane barrier of the MULTIPLYING team.  __syncthreads() is fence(release) + s_barrier + fence(acquire), and the release
// costs `s_waitcnt lgkmcnt(0)`: every transposing read in flight -- the fragments of the NEXT plane, requested during the last
// MFMAs of this one precisely so that they travel across the barrier -- had to land first: one exposed LDS round trip per
// plane (round 6, found in the ISA: lgkmcnt(0) in front of every s_barrier of the plane loop).  This team writes nothing to
// LDS inside the plane loop, so it has nothing to release; what it reads after the barrier was written (and waited for) by the
// staging team before it, and the LDS serves requests in arrival order.  The empty asm keeps the compiler from moving a
// read across the barrier.  TEM_TR_SYNC=1 restores __syncthreads().
#ifndef TEM_TR_SYNC
#define TEM_TR_SYNC 0
#endif
__device__ __forceinline__ void tr_barrier_mult() {
    if (TEM_TR_SYNC) {
        __syncthreads();
        return;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ uint2 tr_read(const unsigned char* p) {
    const tr_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_p)(p));
    return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ uint4 tr_frag(const unsigned char* p) {   // 8 voxels of this lane's channel: two transposing reads
    const uint2 a = tr_read(p), b = tr_read(p + 4 * TR_REC16);
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ tr_rsrc_t tr_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 tr_load4(tr_rsrc_t r, unsigned voff) {
    const tr_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    const tr_f4 f = __builtin_bit_cast(tr_f4, v);   // whole-vector cast (element-wise casts get the load narrowed, conv_pp.hip)
    return make_float4(f.x, f.y, f.z, f.w);
}
// (e0 - h.lo, e1 - h.hi) rounded to two fp16 in one register: the lo term of the fp16 two-term split (h = the hi terms).
// v_fma_mix{lo,hi}_f16 read the fp16 operand by half, compute in fp32 (exact here) and round once (conv_zr.hip: zr_mix_lo)
__device__ __forceinline__ unsigned tr_mix_lo(unsigned h, float e0, float e1) {
    unsigned q;
    const float c = -1.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(q) : "v"(h), "s"(c), "v"(e0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(q) : "v"(h), "s"(c), "v"(e1));
    return q;
}

__device__ __forceinline__ uint4 tr_load4u(tr_rsrc_t r, unsigned voff) {
    const tr_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// pre-norm of two packed 16-bit activations, rounded once to the operand type (conv_zr.hip: zr_norm2)
template <bool F16>
__device__ __forceinline__ unsigned tr_norm2(unsigned h, float s0, float t0, float s1, float t1) {
    if constexpr (F16) {
        unsigned q;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(q) : "v"(h), "v"(s0), "v"(t0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(q) : "v"(h), "v"(s1), "v"(t1));
        return q;
    } else {
        const float a = __builtin_bit_cast(float, h << 16), b = __builtin_bit_cast(float, h & 0xffff0000u);
        return pk_bf16(fmaf(a, s0, t0), fmaf(b, s1, t1));
    }
}

// (g0 * s, g1 * s) rounded to two fp16 in one register: the prescaled gradient term (all sources fp32)
__device__ __forceinline__ unsigned tr_mix_scale(float g0, float g1, float sc) {
    unsigned q;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(q) : "v"(g0), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(q) : "v"(g1), "s"(sc));
    return q;
}

#ifndef TEM_TR_TX1_DIRECT
#define TEM_TR_TX1_DIRECT 1   // the tx = 1 fragment of a row group by two more transposing reads (0: four v_alignbyte on the window; measured +2 % kernel time: the SIMD is issue-bound at ~5 non-MFMA instructions per MFMA over both waves)
#endif
#ifndef TEM_TR_PF2
#define TEM_TR_PF2 1   // one-term modes: fragments of slab s + 2 are read during the MFMAs of slab s (four register sets; 0: s + 1, two sets)
#endif
#ifndef TEM_TR_SHARE
#define TEM_TR_SHARE 0   // one-term modes (round 6 experiment): row groups (tz, ty = 0) and (tz, ty = 2) of one wave share their x windows -- 24 % fewer
                         // LDS reads, bit-identical results, -1.5 .. 3 % in the harness on dense operands and +-0.5 % in the amp step
                         // (profiles/r06_wgrad_tr_ablations.txt): off, the PF2 loop stays the product path
#endif
#ifndef TEM_TR_RTZ
#define TEM_TR_RTZ 0    // experiment: hi term of x^ by v_cvt_pkrtz_f16_f32 (one instruction per pair; the split stays exact, lo grows to 1 ulp)
#endif
#ifndef TEM_TR_DBCOND
#define TEM_TR_DBCOND 0   // experiment: bias-gradient adds only in the workgroups that store them (cit == 0)
#endif
#ifndef TEM_TR_PRIO
#define TEM_TR_PRIO 0   // s_setprio of the multiplying team (its MFMAs and fragment reads against the staging wave of the same SIMD)
#endif
#ifndef TEM_TR_ABL
#define TEM_TR_ABL 0   // harness-only ablations (wrong results): 1 staging team idle, 2 multiplying team idle, 4 fragments read once
#endif                 // per plane (no LDS reads in the MFMA stream), 8 staging without global loads
#ifdef TEM_TR_TRACE   // developer build (scripts/wg_harness.sh ... -DTEM_TR_TRACE): shader-clock stamps of one workgroup's plane loop
#ifndef TEM_TR_TRACE_BLOCK
#define TEM_TR_TRACE_BLOCK 100
#endif
__device__ unsigned long long tem_tr_trace_buf[8][64][4];
#define TR_STAMP(it, i)                                                                               \
    do {                                                                                              \
        if (blockIdx.x == TEM_TR_TRACE_BLOCK && lane == 0 && (it) >= 0 && (it) < 64)                  \
            tem_tr_trace_buf[wv][it][i] = __builtin_amdgcn_s_memtime();                               \
    } while (0)
void tem_tr_trace_read(unsigned long long* dst) {
    (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tem_tr_trace_buf), sizeof(unsigned long long) * 8 * 64 * 4);
}
#else
#define TR_STAMP(it, i)
#endif

// TS (round 5): element type of x and g in HBM.  A 16-bit TS is the one-term mode of the same type (ARITH 1: fp16, 2: bf16) and
// changes only the staging team: a voxel record in HBM is what LDS holds (64 bytes), FOUR lanes move it with one 16-byte load
// each -- x through the packed pre-norm (tr_norm2: 4 instructions per 8 channels), g untouched -- and one ds_write_b128; an x
// plane is two rounds of loads per thread instead of four, a g plane one instead of two.
template <int ARITH, typename TS = float>
__global__ __launch_bounds__(512, 1) void k_conv_wgrad_tr(const TS* __restrict__ x, int64_t x_ld,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const TS* __restrict__ g,
                                                          int64_t g_ld, float* __restrict__ part,
                                                          float* __restrict__ dbpart, int N, int D, int H, int W,
                                                          int Cin, int Cout, int T, int nY, int nX, int zsegs,
                                                          int S, int ncz, unsigned* __restrict__ gmax,
                                                          const unsigned* __restrict__ g_amax, int64_t x_cs) {
    constexpr int NT = 27, NA = 7;
    constexpr int NX = (ARITH == 0 || ARITH == 3) ? 2 : 1;   // terms of x^
    constexpr int NG = (ARITH == 0) ? 2 : 1;                 // terms of g
    constexpr bool F16 = ARITH == 1 || ARITH == 3;
    constexpr bool H21 = ARITH == 3;
    constexpr bool T16 = sizeof(TS) == 2;
    static_assert(!T16 || ARITH == 1 || ARITH == 2, "16-bit storage: the one-term mode of the same type");
    constexpr bool FP32 = ARITH == 4;   // exact fp32: v_mfma_f32_32x32x2_f32 on fp32 records (no transpose needed: K = 2 voxels = the two lane halves)
    constexpr int TR_REC = FP32 ? 128 : 64, TR_QB = TR_REC / 8;   // bytes per voxel record / per channel quad
    constexpr int TR_XROW = 10 * TR_REC, TR_XPL = 10 * TR_XROW, TR_XT = TR_XT_OF(TR_REC);
    constexpr int TR_GPL = 64 * TR_REC, TR_GT = TR_GT_OF(TR_REC);
    auto TR_XSLOT = [](int pl) { return ((pl + 2 * TR_NXS) % TR_NXS) * TR_XPL; };   // ring slot of halo plane pl (>= -2 TR_NXS)
    auto TR_GSLOT = [](int pl) { return ((pl + 3 * TR_NGS) % TR_NGS) * TR_GPL; };
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const X0 = ldsb;                  // [term NX][slot 4][row 10][x 10][32 ch]
    unsigned char* const G0 = ldsb + NX * TR_XT;     // [term NG][buffer 2][row 8][x 8][32 ch]
    unsigned char* const TRASH = G0 + NG * TR_GT;    // 16-bit tensors: 256 x 16 bytes that threads without a halo voxel write to

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mteam = wv < 4;
    const int bid0 = tem_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = bid0 % T;
    const int sp = bid0 / T;               // partial-slab index: this workgroup walks column segments sp % S, + S, ...
    const int ncit = Cin >> 5;
    const int cit = tile % ncit, cog = tile / ncit;
    const int n = sp / S;

    float psc = 1.f, pinv = 1.f;
    if (H21) {
        // (No MODE.FP16_OVFL here: with it the conversions clamp instead of producing inf -- and, measured with
        // scripts/nan_probe.py, a NaN in g no longer reaches dw.  x^ is a norm output (|x^| <= sqrt(voxels) < 65504 without
        // an affine gain) and g is prescaled below 2^15: nothing overflows in a healthy step, and in an unhealthy one the
        // inf / NaN must stay visible in the weight gradient.)
        const int e = (int)((*g_amax >> 23) & 0xffu);                       // biased exponent of max |g| (0: all zeros)
        const int k = e == 0 ? 0 : min(max(141 - e, -100), 100);            // max |g| * 2^k in [2^14, 2^15)
        psc = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
        pinv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    }

    if (mteam) {
        // ---------------- multiplying team ----------------
        if (TEM_TR_PRIO) __builtin_amdgcn_s_setprio(TEM_TR_PRIO);
        floatx16 acc[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
        // lane part of a fragment address: half-wave = x-row of the slab, 16-lane group = channel half, lanes 4 j .. 4 j + 3 of
        // a group address voxel j of the run, 8 bytes (4 channels) each
        const int p16 = lane & 15;
        const int lane_rec = (p16 >> 2) * TR_REC + ((lane >> 4) & 1) * 32 + (p16 & 3) * 8;
        const int lane_x = (lane >> 5) * TR_XROW + (FP32 ? (lane & 31) * 4 : lane_rec);   // fp32: lane = channel, lane half = row
        const int lane_g = (lane >> 5) * (8 * TR_REC) + (FP32 ? (lane & 31) * 4 : lane_rec);
        // taps of this wave (as k_conv_wgrad_zt): accumulators 0..2 = row group (tz, ty) = wv with tx = 0, 1, 2; 3..5 = row group
        // wv + 4; accumulator 6 = (row group 8, tx = wv) -- wave 3 has no seventh tap: it repeats tx = 0 there and never stores it.
        // A row group reads ONE 12-voxel window of its x row (three transposing reads: elements k0 .. k11 of the lane's
        // channel); the tx = 0 fragment is registers 0..3 of it, tx = 2 registers 1..4, tx = 1 four v_alignbyte.
        int rgtz[3], rgoff[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int rg = a < 2 ? wv + 4 * a : 8;
            rgtz[a] = rg / 3;
            rgoff[a] = (rg % 3) * TR_XROW + (a == 2 ? (wv < 3 ? wv : 0) * TR_REC : 0);
        }
        for (int cz = sp % S; cz < ncz; cz += S) {
            const int zseg = cz % zsegs;
            const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
            for (int t = za - 6; t < za; ++t) __syncthreads();   // the staging team primes the ring (six iterations)
            // Software pipeline over the 4 k-slabs x (term) phases of a plane: the fragment reads of the NEXT phase are issued
            // between the MFMAs of the current one (sched_group_barrier), so a phase never starts by waiting for LDS -- and
            // the last phase of a plane reads the first fragments of the NEXT plane: the staging team is one plane further
            // ahead than the ring of k_conv_wgrad_zs (5 x^ slots, 3 g buffers), so that data is complete before this plane
            // starts and the barrier at its end only hands slots back (first version: 340 cycles of LDS latency per plane).
            // Phases of slab sl: [x lo * g hi] (NX == 2), [x hi * g lo] (NG == 2), [x hi * g hi]; products small first.
            uint4 gh, gl, xh[NA], xl[NA];
            auto load_g = [&](const unsigned char* gb, int sl) {
                if ((TEM_TR_ABL & 4) && sl > 0) return;
                gh = tr_frag(gb + sl * 16 * TR_REC);
                if (NG == 2) gl = tr_frag(gb + TR_GT + sl * 16 * TR_REC);
            };
            auto load_x = [&](uint4* f, const unsigned char* const* xb, int term, int sl) {
                if ((TEM_TR_ABL & 4) && sl > 0) return;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const unsigned char* p = xb[a] + term * TR_XT + sl * 2 * TR_XROW;
                    const uint2 w0 = tr_read(p), w1 = tr_read(p + 4 * TR_REC), w2 = tr_read(p + 8 * TR_REC);
                    f[3 * a + 0] = make_uint4(w0.x, w0.y, w1.x, w1.y);
                    if (TEM_TR_TX1_DIRECT)
                        f[3 * a + 1] = tr_frag(p + TR_REC16);
                    else
                        f[3 * a + 1] = make_uint4(__builtin_amdgcn_alignbyte(w0.y, w0.x, 2), __builtin_amdgcn_alignbyte(w1.x, w0.y, 2),
                                                  __builtin_amdgcn_alignbyte(w1.y, w1.x, 2), __builtin_amdgcn_alignbyte(w2.x, w1.y, 2));
                    f[3 * a + 2] = make_uint4(w0.y, w1.x, w1.y, w2.x);
                }
                f[6] = tr_frag(xb[2] + term * TR_XT + sl * 2 * TR_XROW);
            };
            auto interleave = [&]() {   // the 7 MFMAs of the phase just written and the reads / shifts for the next one:
                // reads behind the first three MFMAs, the v_alignbyte of the tx = 1 fragments behind the last three (their
                // reads have returned by then: an LDS wait inside the MFMA stream would stall the matrix pipe)
                if (TEM_TR_TX1_DIRECT) {   // 12 (+ 2) reads, no shifts: three reads behind each of the first five MFMAs
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    return;
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto bases = [&](int t, const unsigned char** xb) -> const unsigned char* {   // fragment bases of plane t
#pragma unroll
                for (int a = 0; a < 3; ++a) xb[a] = X0 + lane_x + TR_XSLOT(t + rgtz[a] - 1) + rgoff[a];
                return G0 + lane_g + TR_GSLOT(t);
            };
            // One-term x with the prefetch TWO slabs deep: a slab is only 7 MFMAs (224 cycles) and a transposing LDS read under
            // the load of eight waves takes longer than that to return -- with one slab of lead every slab still waited
            // (counters of the amp step: matrix pipe 0.44 busy at 2.2 GHz, not power-limited).  Four register sets named by the
            // slab index (four slabs per plane: static names), 170 -> ~235 VGPRs of the 256 this kernel may use.
            // ---- one-term modes, round 6: SHARED WINDOWS ------------------------------------------------------------------
            // The one-term kernel is bound by LDS bandwidth: 14 transposing reads per slab and wave x 4 waves x 512 B = 28 KB per
            // 224 cycles of MFMAs = the 128 B / clock the LDS delivers.  The fragment of tap (tz, ty = 2) at slab s -- halo rows
            // 2 s + 2, 2 s + 3 -- IS the fragment of tap (tz, ty = 0) at slab s + 1, so a wave that owns both row groups of one tz
            // reads FIVE 12-voxel windows per plane instead of eight.  Ownership: waves 0..2 = (tz = wave, ty = 0), (tz = wave,
            // ty = 2) and one tap of (tz = 2, ty = 1); wave 3 = (0, 1) and (1, 1), no shared rows.  Reads per plane and CU
            // 224 -> 171.  Window k of a plane lives in register set k & 3; a set is refilled as soon as the last MFMA group that
            // reads it has been issued, two slabs before its next use; the B group (ty = 2) of a slab is issued before the A
            // group, so that window 0 of the NEXT plane can follow window 4 into set 0 inside slab 3.
            constexpr bool SH = NX == 1 && !FP32 && TEM_TR_SHARE;
            if constexpr (SH) {
                auto wbase = [&](int t, int tz, int ro) -> const unsigned char* {
                    return X0 + lane_x + TR_XSLOT(t + tz - 1) + ro * TR_XROW;
                };
                auto rd_win = [&](uint4* f, const unsigned char* p) {   // taps tx = 0, 1, 2 of one row pair: 3 + 2 transposing reads
                    const uint2 w0 = tr_read(p), w1 = tr_read(p + 4 * TR_REC), w2 = tr_read(p + 8 * TR_REC);
                    f[0] = make_uint4(w0.x, w0.y, w1.x, w1.y);
                    f[1] = tr_frag(p + TR_REC16);
                    f[2] = make_uint4(w0.y, w1.x, w1.y, w2.x);
                };
                uint4 Bw[4][3], gq[4], eq[4];
                if (wv < 3) {
                    {
                        const unsigned char* wb = wbase(za, wv, 0);
                        rd_win(Bw[0], wb);
                        rd_win(Bw[1], wb + 2 * TR_XROW);
                        rd_win(Bw[2], wb + 4 * TR_XROW);
                        const unsigned char* eb = wbase(za, 2, 1) + wv * TR_REC;
                        eq[0] = tr_frag(eb);
                        eq[1] = tr_frag(eb + 2 * TR_XROW);
                        const unsigned char* gb = G0 + lane_g + TR_GSLOT(za);
                        gq[0] = tr_frag(gb);
                        gq[1] = tr_frag(gb + 16 * TR_REC);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll 1
                    for (int t = za; t < zb; ++t) {
                        const unsigned char *wb = wbase(t, wv, 0), *wbn = wbase(t + 1, wv, 0);
                        const unsigned char *eb = wbase(t, 2, 1) + wv * TR_REC, *ebn = wbase(t + 1, 2, 1) + wv * TR_REC;
                        const unsigned char *gb = G0 + lane_g + TR_GSLOT(t), *gbn = G0 + lane_g + TR_GSLOT(t + 1);
                        TR_STAMP(t - za, 0);
                        if (TEM_TR_ABL & 2) {
                            __syncthreads();
                            continue;
                        }
                        TR_STAMP(t - za, 1);
                        auto group = [&](int a0, const uint4* f, const uint4& gg) {
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc[a0 + j] = mfma16<F16>(f[j], gg, acc[a0 + j]);
                        };
                        auto pat9 = [&]() {   // 7 MFMAs, 9 reads
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            }
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        // slab 0: windows 0 (set 0), 1 (set 1); window 3 -> set 3
                        rd_win(Bw[3], wb + 6 * TR_XROW);
                        gq[2] = tr_frag(gb + 2 * 16 * TR_REC);
                        eq[2] = tr_frag(eb + 4 * TR_XROW);
                        group(3, Bw[1], gq[0]);
                        group(0, Bw[0], gq[0]);
                        acc[6] = mfma16<F16>(eq[0], gq[0], acc[6]);
                        pat9();
                        // slab 1: windows 1, 2; window 4 -> set 0
                        rd_win(Bw[0], wb + 8 * TR_XROW);
                        gq[3] = tr_frag(gb + 3 * 16 * TR_REC);
                        eq[3] = tr_frag(eb + 6 * TR_XROW);
                        group(3, Bw[2], gq[1]);
                        group(0, Bw[1], gq[1]);
                        acc[6] = mfma16<F16>(eq[1], gq[1], acc[6]);
                        pat9();
                        // slab 2: windows 2, 3; window 1 of the NEXT plane -> set 1
                        rd_win(Bw[1], wbn + 2 * TR_XROW);
                        gq[0] = tr_frag(gbn);
                        eq[0] = tr_frag(ebn);
                        group(3, Bw[3], gq[2]);
                        group(0, Bw[2], gq[2]);
                        acc[6] = mfma16<F16>(eq[2], gq[2], acc[6]);
                        pat9();
                        // slab 3: windows 3 (set 3), 4 (set 0); windows 2 and -- behind the MFMAs that read set 0 -- 0 of the next plane
                        rd_win(Bw[2], wbn + 4 * TR_XROW);
                        gq[1] = tr_frag(gbn + 16 * TR_REC);
                        eq[1] = tr_frag(ebn + 2 * TR_XROW);
                        group(3, Bw[0], gq[3]);
                        rd_win(Bw[0], wbn);
                        group(0, Bw[3], gq[3]);
                        acc[6] = mfma16<F16>(eq[3], gq[3], acc[6]);
#pragma unroll
                        for (int i = 0; i < 3; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        TR_STAMP(t - za, 2);
                        tr_barrier_mult();
                        TR_STAMP(t - za, 3);
                    }
                } else {
                    // wave 3: row groups (tz = 0, ty = 1) [A, accumulators 0..2] and (tz = 1, ty = 1) [B, 3..5]: different halo planes,
                    // nothing to share.  Slab s uses sets 2 (s & 1) and 2 (s & 1) + 1; each is refilled for slab s + 2 right behind
                    // the three MFMAs that read it (two slabs of lead, as the shared windows have).
                    {
                        const unsigned char *ab = wbase(za, 0, 1), *bb = wbase(za, 1, 1);
                        rd_win(Bw[0], ab);
                        rd_win(Bw[1], bb);
                        rd_win(Bw[2], ab + 2 * TR_XROW);
                        rd_win(Bw[3], bb + 2 * TR_XROW);
                        const unsigned char* gb = G0 + lane_g + TR_GSLOT(za);
                        gq[0] = tr_frag(gb);
                        gq[1] = tr_frag(gb + 16 * TR_REC);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll 1
                    for (int t = za; t < zb; ++t) {
                        const unsigned char *ab = wbase(t, 0, 1), *bb = wbase(t, 1, 1);
                        const unsigned char *abn = wbase(t + 1, 0, 1), *bbn = wbase(t + 1, 1, 1);
                        const unsigned char *gb = G0 + lane_g + TR_GSLOT(t), *gbn = G0 + lane_g + TR_GSLOT(t + 1);
                        TR_STAMP(t - za, 0);
                        if (TEM_TR_ABL & 2) {
                            __syncthreads();
                            continue;
                        }
                        TR_STAMP(t - za, 1);
#pragma unroll
                        for (int sl = 0; sl < 4; ++sl) {
                            const int nsl = (sl + 2) & 3;
                            uint4* const pa = Bw[2 * (sl & 1)];
                            uint4* const pb = Bw[2 * (sl & 1) + 1];
                            gq[nsl] = tr_frag((sl < 2 ? gb : gbn) + nsl * 16 * TR_REC);
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc[3 + j] = mfma16<F16>(pb[j], gq[sl], acc[3 + j]);
                            rd_win(pb, (sl < 2 ? bb : bbn) + nsl * 2 * TR_XROW);
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc[j] = mfma16<F16>(pa[j], gq[sl], acc[j]);
                            rd_win(pa, (sl < 2 ? ab : abn) + nsl * 2 * TR_XROW);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        TR_STAMP(t - za, 2);
                        tr_barrier_mult();
                        TR_STAMP(t - za, 3);
                    }
                }
                for (int e = zb - za + 6; e % (T16 ? TR_UNROLL : TR_UNROLL32); ++e) __syncthreads();
                continue;
            }
            constexpr bool PF2 = NX == 1 && !FP32 && TEM_TR_PF2 && !SH;
            uint4 xq[PF2 ? 4 : 1][NA], gq[PF2 ? 4 : 1];
            if constexpr (PF2) {
                const unsigned char* xb0[3];
                const unsigned char* gb0 = bases(za, xb0);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    gq[PF2 ? sl : 0] = tr_frag(gb0 + sl * 16 * TR_REC);
                    load_x(xq[PF2 ? sl : 0], xb0, 0, sl);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (!FP32) {   // first fragments of the segment's first plane
                const unsigned char* xb0[3];
                const unsigned char* gb0 = bases(za, xb0);
                load_g(gb0, 0);
                load_x(NX == 2 ? xl : xh, xb0, NX == 2 ? 1 : 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (FP32) {
                // exact fp32: a k-step = one x position of the slab's two rows (lanes 0..31 row 2 s, 32..63 row 2 s + 1); one
                // ds_read_b32 per operand, the g value serves the 7 taps; reads of k-step i + 1 fly during the MFMAs of i
#pragma unroll 1
                for (int t = za; t < zb; ++t) {
                    const unsigned char* xb[3];
                    const unsigned char* gb = bases(t, xb);
                    if (TEM_TR_ABL & 2) {
                        __syncthreads();
                        continue;
                    }
                    auto rd = [&](int ks, float* xv) -> float {   // ks = 8 sl + x
                        const int off = (ks >> 3) * 2 * TR_XROW + (ks & 7) * TR_REC;
#pragma unroll
                        for (int j = 0; j < 6; ++j) xv[j] = *reinterpret_cast<const float*>(xb[j / 3] + off + (j % 3) * TR_REC);
                        xv[6] = *reinterpret_cast<const float*>(xb[2] + off);
                        return *reinterpret_cast<const float*>(gb + (ks >> 3) * 16 * TR_REC + (ks & 7) * TR_REC);
                    };
                    float xc[NA], xn[NA];
                    float gc = rd(0, xc), gn = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 32; ++ks) {
                        if (ks + 1 < 32) gn = rd(ks + 1, xn);
#pragma unroll
                        for (int j = 0; j < NA; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[j], gc, acc[j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < NA; ++j) xc[j] = xn[j];
                        gc = gn;
                    }
                    tr_barrier_mult();
                }
                for (int e = zb - za + 6; e % TR_UNROLL32; ++e) __syncthreads();
                continue;
            }
#pragma unroll 1
            for (int t = za; t < zb; ++t) {
                const unsigned char *xb[3], *xbn[3];
                const unsigned char* gb = bases(t, xb);
                const unsigned char* gbn = bases(t + 1, xbn);
                TR_STAMP(t - za, 0);
                if (TEM_TR_ABL & 2) {
                    __syncthreads();
                    continue;
                }
                TR_STAMP(t - za, 1);
                if constexpr (PF2) {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        const int nsl = (sl + 2) & 3;
                        gq[PF2 ? nsl : 0] = tr_frag((sl < 2 ? gb : gbn) + nsl * 16 * TR_REC);
                        load_x(xq[PF2 ? nsl : 0], sl < 2 ? xb : xbn, 0, nsl);
#pragma unroll
                        for (int j = 0; j < NA; ++j) acc[j] = mfma16<F16>(xq[PF2 ? sl : 0][j], gq[PF2 ? sl : 0], acc[j]);
                        interleave();
                    }
                    TR_STAMP(t - za, 2);
                    tr_barrier_mult();
                    TR_STAMP(t - za, 3);
                    continue;
                }
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    if (NX == 2) {
                        load_x(xh, xb, 0, sl);   // for the next phase of this slab
#pragma unroll
                        for (int j = 0; j < NA; ++j) acc[j] = mfma16<F16>(xl[j], gh, acc[j]);
                        interleave();
                    }
                    if (NG == 2) {
#pragma unroll
                        for (int j = 0; j < NA; ++j) acc[j] = mfma16<F16>(xh[j], gl, acc[j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // last phase of the slab: the g fragment(s) and the first x operand of the next slab -- after slab 3: of
                    // the next PLANE -- are read meanwhile (into a second g register set: this phase still multiplies with
                    // the current one)
                    const uint4 ghc = gh;
                    load_g(sl < 3 ? gb : gbn, (sl + 1) & 3);
                    if (NX == 2) load_x(xl, sl < 3 ? xb : xbn, 1, (sl + 1) & 3);
                    if constexpr (NX == 1) {
                        // one-term x (the mixed modes): the lo register set is free, so it is the SECOND fragment buffer -- the
                        // reads of slab sl + 1 fly during the 7 MFMAs of slab sl (even slabs multiply xh and fill xl, odd ones the
                        // reverse; four slabs per plane keep the parity).  Round 4 read the next fragments into the SAME registers
                        // behind the MFMAs that used them: every slab then waited a full LDS round trip (trace of round 5, amp mode:
                        // 2200 cycles per plane for 28 MFMAs = 0.4 of the matrix pipe, unchanged by 16-bit staging).
                        uint4* const cur = (sl & 1) ? xl : xh;
                        uint4* const nxt = (sl & 1) ? xh : xl;
                        load_x(nxt, sl < 3 ? xb : xbn, 0, (sl + 1) & 3);
#pragma unroll
                        for (int j = 0; j < NA; ++j) acc[j] = mfma16<F16>(cur[j], ghc, acc[j]);
                        interleave();
                        continue;
                    }
#pragma unroll
                    for (int j = 0; j < NA; ++j) acc[j] = mfma16<F16>(xh[j], ghc, acc[j]);
                    interleave();
                }
                TR_STAMP(t - za, 2);
                tr_barrier_mult();
                TR_STAMP(t - za, 3);
            }
            // the staging team walks a segment in trips of TR_UNROLL (16-bit tensors) / TR_UNROLL32 iterations
            for (int e = zb - za + 6; e % (T16 ? TR_UNROLL : TR_UNROLL32); ++e) __syncthreads();
        }
        if (TEM_TR_PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- partial slabs: D[row = ci][col = co] ----
        if (cog * 32 < Cout) {
            const int kh = lane >> 5, r = lane & 31;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (j == 6 && wv >= 3) break;
                // shared windows (one-term modes): waves 0..2 hold row groups (wave, 0), (wave, 2) and tap tx = wave of row group (2, 1),
                // wave 3 row groups (0, 1) and (1, 1); else row groups wave, wave + 4 and tap tx = wave of row group 8
                constexpr bool SHT = NX == 1 && !FP32 && TEM_TR_SHARE;
                const int tap = SHT ? (wv < 3 ? (j < 3 ? (wv * 3) * 3 + j : j < 6 ? (wv * 3 + 2) * 3 + (j - 3) : 7 * 3 + wv)
                                              : (j < 3 ? 1 * 3 + j : 4 * 3 + (j - 3)))
                                    : (j < 6) ? (wv + 4 * (j / 3)) * 3 + (j % 3) : 8 * 3 + wv;
                float* dst = part + (((int64_t)sp * NT + tap) * Cin + cit * 32) * Cout + cog * 32 + r;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                    dst[(int64_t)row * Cout] = H21 ? acc[j][reg] * pinv : acc[j][reg];
                }
            }
        }
    }

    // ---------------- staging team ----------------
    // item = (voxel, channel quad): 8 lanes cover the 128-byte line of a voxel; x plane: 100 halo voxels = 800 items over 256
    // threads (four rounds, the last one 32 threads), g plane: 64 voxels = 512 items (two rounds).  The quad of a thread is
    // the same in every round (256 % 8 == 0), so norm parameters and the bias-gradient sums are per thread.
    const int tl = tid & 255;
    const int quad = tl & 7;
    const bool do_db = (dbpart != nullptr) && (cit == 0);
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    float gmx = 0.f;
    if constexpr (T16) {
        // ---------------- staging team, 16-bit tensors: item = (voxel, channel OCTET), 4 lanes per 64-byte record ----------------
        const int oct = tl & 3;
        float db8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!mteam) {
            uint4 xa[2][2] = {{make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)}, {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)}};
            uint4 ga[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f), sc5 = sc4, sf5 = sf4;
            if (scale) {
                sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + cit * 32 + oct * 8);
                sc5 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + cit * 32 + oct * 8 + 4);
                sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + cit * 32 + oct * 8);
                sf5 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + cit * 32 + oct * 8 + 4);
            }
            // (a chunk stride x_cs != 0 puts the 32-channel tile cit at x + cit * x_cs: planar concat halves, tem_act.h;
            //  the in-plane offsets below keep their cit * 32)
            const TS* const xn = x + (int64_t)n * D * H * W * x_ld + (x_cs ? (int64_t)cit * (x_cs - 32) : 0);
            const TS* const gn = g + (int64_t)n * D * H * W * g_ld;
            const int64_t xplane = (int64_t)H * W * x_ld, gplane = (int64_t)H * W * g_ld;
            constexpr unsigned OOB = 0x80000000u;   // >= num_records of tr_rsrc: the load returns zeros
            for (int cz = sp % S; cz < ncz; cz += S) {
                const int zseg = cz % zsegs;
                const int col = cz / zsegs;
                const int y0 = (col / nX) * 8, x0 = (col % nX) * 8;
                const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
                unsigned offx[2], offg;
                bool okx[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int hv = (tl + 256 * q) >> 2;                 // halo voxel 0..99 (round 1: threads 0..143 only)
                    const int gy = y0 + hv / 10 - 1, gx = x0 + hv % 10 - 1;
                    okx[q] = hv < 100 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                    offx[q] = okx[q] ? (unsigned)((gy * W + gx) * (int)x_ld + cit * 32 + oct * 8) * 2u : OOB;
                }
                {
                    const int gv = tl >> 2;                             // patch voxel 0..63
                    const int hy = y0 + (gv >> 3), hx = x0 + (gv & 7);
                    const bool ok = hy < H && hx < W && cog * 32 + oct * 8 < Cout;
                    offg = ok ? (unsigned)((hy * W + hx) * (int)g_ld + cog * 32 + oct * 8) * 2u : OOB;
                }
                bool zin[2] = {false, false};
                // same schedule as the fp32 staging below: iteration t stores x plane t + 3 / g plane t + 2 from the register set
                // loaded two iterations ago and loads x plane t + 5 / g plane t + 4 into it
                // Every iteration issues its three loads UNCONDITIONALLY (a plane outside the volume / the segment reads with an
                // offset beyond the buffer: zeros) and in straight-line code.  With the loads inside `if (plane in range)` blocks
                // -- round 4, and the first 16-bit version -- the compiler cannot count what is outstanding at the loop head and
                // waits with vmcnt(0) before the first conversion: the loads of the OTHER register set, issued one iteration
                // earlier, were waited for at once -- the full memory latency sat in every iteration (trace: 900 - 1900 cycles
                // from the top of an iteration to its last LDS store, the multiplying team waiting at the barrier).
                auto iteration = [&](int t, uint4(&xs_)[2], uint4& gs_, bool& zin_) {
                    TR_STAMP(t - za, 0);
                    if (TEM_TR_ABL & 1) {
                        __syncthreads();
                        return;
                    }
                    {   // (no branch in here: the pre-norm always runs -- scale 1, shift 0 without a norm reproduces the value bit for
                        // bit --, threads without a second halo voxel write a scratch record, the priming iterations write planes
                        // that are rewritten before anything reads them)
                        unsigned char* const xs = X0 + TR_XSLOT(t + 3) + oct * 16;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int hv = (tl + 256 * q) >> 2;
                            uint4 r = xs_[q];   // zeros for a plane or a voxel outside the volume
                            const bool ok = okx[q] && zin_;
                            r.x = tr_norm2<F16>(r.x, sc4.x, sf4.x, sc4.y, sf4.y);
                            r.y = tr_norm2<F16>(r.y, sc4.z, sf4.z, sc4.w, sf4.w);
                            r.z = tr_norm2<F16>(r.z, sc5.x, sf5.x, sc5.y, sf5.y);
                            r.w = tr_norm2<F16>(r.w, sc5.z, sf5.z, sc5.w, sf5.w);
                            r.x = ok ? r.x : 0u; r.y = ok ? r.y : 0u; r.z = ok ? r.z : 0u; r.w = ok ? r.w : 0u;
                            unsigned char* const dst = (q == 0 || hv < 100) ? xs + hv * TR_REC : TRASH + tl * 16;
                            *reinterpret_cast<uint4*>(dst) = r;
                        }
                    }
                    {
                        unsigned char* const gs = G0 + TR_GSLOT(t + 2) + oct * 16;
                        const uint4 v = gs_;   // zeros where out of range (load offset beyond the buffer)
                        *reinterpret_cast<uint4*>(gs + (tl >> 2) * TR_REC) = v;
                        db8[0] += act_lo<TS>(v.x); db8[1] += act_hi<TS>(v.x); db8[2] += act_lo<TS>(v.y); db8[3] += act_hi<TS>(v.y);
                        db8[4] += act_lo<TS>(v.z); db8[5] += act_hi<TS>(v.z); db8[6] += act_lo<TS>(v.w); db8[7] += act_hi<TS>(v.w);
                    }
                    TR_STAMP(t - za, 1);
                    {
                        // (plain scalar arithmetic, no short-circuit operators: a branch around a load would be back)
                        const int zx = t + 5;
                        const int ldx = (int)(zx >= za - 1) & (int)(zx <= zb) & (int)(zx >= 0) & (int)(zx < D) & (int)!(TEM_TR_ABL & 8);   // wave-uniform
                        zin_ = ldx != 0;
                        const tr_rsrc_t rsx = tr_rsrc(xn + (int64_t)(zx * ldx) * xplane);
                        const unsigned mx = ldx ? 0u : OOB;   // OR-ed into the offsets (< 2^31): beyond the buffer -> zeros
#pragma unroll
                        for (int q = 0; q < 2; ++q) xs_[q] = tr_load4u(rsx, offx[q] | mx);
                        const int zg = t + 4;
                        const int ldg = (int)(zg >= za) & (int)(zg < zb) & (int)!(TEM_TR_ABL & 8);
                        const tr_rsrc_t rsg = tr_rsrc(gn + (int64_t)(zg * ldg) * gplane);
                        gs_ = tr_load4u(rsg, offg | (ldg ? 0u : OOB));
                    }
                    TR_STAMP(t - za, 2);
                    __syncthreads();
                    TR_STAMP(t - za, 3);
                };
                // Trips of TR_UNROLL iterations, straight-line: inside a trip the compiler counts the outstanding loads exactly
                // (vmcnt(3) before a conversion: the other set's three loads stay in flight); at the loop head it does not -- it
                // waits for everything (vmcnt(0)), i.e. for the loads issued one barrier earlier: one exposed memory latency per
                // TRIP (with trips of two iterations, round 4's loop, per two planes: 900 of 2300 cycles per plane in the trace).
                // A segment is padded to whole trips with idle iterations (zero planes into free ring slots; the multiplying team
                // meets their barriers).
                const int t_end = za - 6 + (zb - za + 6 + TR_UNROLL - 1) / TR_UNROLL * TR_UNROLL;
#pragma unroll 1
                for (int t = za - 6; t < t_end; t += TR_UNROLL) {
#pragma unroll
                    for (int k = 0; k < TR_UNROLL; ++k) iteration(t + k, xa[k & 1], ga[k & 1], zin[k & 1]);
                }
            }
        }
        if (do_db) {   // bias-gradient partial of this workgroup: 64 threads per channel octet
            float* red = reinterpret_cast<float*>(ldsb);  // [64][32]
            if (!mteam) {
#pragma unroll
                for (int c = 0; c < 8; ++c) red[(tl >> 2) * 32 + oct * 8 + c] = db8[c];
            }
            __syncthreads();
            if (tid < 32 && cog * 32 + tid < Cout) {
                float a = 0.f;
                for (int rr = 0; rr < 64; ++rr) a += red[rr * 32 + tid];
                dbpart[(int64_t)sp * Cout + cog * 32 + tid] = a;
            }
        }
        return;
    } else
    if (!mteam) {
        // The staging waves share their SIMDs with the multiplying waves and get an issue slot every ~8 cycles (trace of the
        // first version: 250 instructions = 2000 - 2700 cycles per plane, the multiplying team waited for them).  So this
        // loop is written for instruction COUNT: no per-element masks (out-of-range voxels use scale = shift = 0 chosen once
        // per column and a load offset beyond the buffer, which returns zeros), no branches around loads, and
        // TWO sets of pending registers: the loads of an iteration are converted two iterations later.
        float4 xa[2][4], ga[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // (a set is stored before it is first loaded: the priming iterations write zero planes)
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            ga[i][0] = ga[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) {
            sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)n * Cin + cit * 32 + quad * 4);
            sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)n * Cin + cit * 32 + quad * 4);
        }
        const TS* const xn = x + (int64_t)n * D * H * W * x_ld;   // a load's resource starts at its z-plane: offsets stay
        const TS* const gn = g + (int64_t)n * D * H * W * g_ld;   // inside one plane (< 2 GiB, checked by the host side)
        const int64_t xplane = (int64_t)H * W * x_ld, gplane = (int64_t)H * W * g_ld;
        constexpr unsigned OOB = 0x80000000u;   // >= num_records of tr_rsrc: the load returns zeros
        const bool q3 = tl < 32;                // the fourth round of an x plane: halo voxels 96..99
        for (int cz = sp % S; cz < ncz; cz += S) {
            const int zseg = cz % zsegs;
            const int col = cz / zsegs;
            const int y0 = (col / nX) * 8, x0 = (col % nX) * 8;
            const int za = (int)(((int64_t)zseg * D) / zsegs), zb = (int)(((int64_t)(zseg + 1) * D) / zsegs);
            unsigned offx[4], offg[2];
            float4 scm[4], sfm[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int hv = (tl + 256 * q) >> 3;                 // halo voxel 0..99 (round 3: threads 0..31 only)
                const int gy = y0 + hv / 10 - 1, gx = x0 + hv % 10 - 1;
                const bool ok = hv < 100 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                offx[q] = ok ? (unsigned)((gy * W + gx) * (int)x_ld + cit * 32 + quad * 4) * 4u : OOB;
                scm[q] = ok ? sc4 : make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding applies after the pre-norm
                sfm[q] = ok ? sf4 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int gv = (tl + 256 * q) >> 3;                 // patch voxel 0..63
                const int hy = y0 + (gv >> 3), hx = x0 + (gv & 7);
                const bool ok = hy < H && hx < W && cog * 32 + quad * 4 < Cout;
                offg[q] = ok ? (unsigned)((hy * W + hx) * (int)g_ld + cog * 32 + quad * 4) * 4u : OOB;
            }
            bool zin[2] = {false, false};   // the set's x plane lies inside the volume (wave-uniform)
            // iteration t: the multiplying team works on plane t (if t >= za) and prefetches the first fragments of plane t + 1;
            // this team converts and stores the register set loaded two iterations ago (x plane t + 3 into its ring slot,
            // g plane t + 2 into its buffer), loads x plane t + 5 and g plane t + 4 into the same set, and joins the barrier.
            // Planes -1 and D are stored as zeros.
            // Round 6: the SIX LOADS of an iteration are issued unconditionally, in straight-line code, in trips of TR_UNROLL32
            // iterations (a plane outside the volume / the segment reads with an offset beyond the buffer: zeros) -- what the
            // 16-bit path above does since round 5.  With the loads inside `if (plane in range)` blocks the compiler cannot count
            // what is outstanding and waited `vmcnt(3) .. vmcnt(0)` in front of the conversions: all of this set's loads AND the
            // other set's, issued ONE plane earlier -- a full memory latency exposed in every iteration, with the multiplying
            // team waiting at the barrier (ablations of round 6, profiles/r06_wgrad_tr_ablations.txt: staging team idle
            // 0.446 -> 0.306 ms, fragment reads removed 0.446 -> 0.407 ms: the staging team WAS the critical path).
            auto iteration = [&](int t, float4(&xs_)[4], float4(&gs_)[2], bool& zin_) {
                TR_STAMP(t - za, 0);
                if (TEM_TR_ABL & 1) {
                    __syncthreads();
                    return;
                }
                {
                    unsigned char* const xs = X0 + TR_XSLOT(t + 3) + quad * TR_QB;
                    if (zin_) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int hv = (tl + 256 * q) >> 3;
                            if (q == 3 && !q3) break;
                            const float e0 = fmaf(xs_[q].x, scm[q].x, sfm[q].x), e1 = fmaf(xs_[q].y, scm[q].y, sfm[q].y);
                            const float e2 = fmaf(xs_[q].z, scm[q].z, sfm[q].z), e3 = fmaf(xs_[q].w, scm[q].w, sfm[q].w);
                            if constexpr (FP32) {
                                *reinterpret_cast<float4*>(xs + hv * TR_REC) = make_float4(e0, e1, e2, e3);
                                continue;
                            }
                            uint2 hi, lo;
                            if (H21) {
                                if (TEM_TR_RTZ)
                                    hi = make_uint2(__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e0, e1)),
                                                    __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(e2, e3)));
                                else
                                    hi = make_uint2(pk16<true>(e0, e1), pk16<true>(e2, e3));
                                lo = make_uint2(tr_mix_lo(hi.x, e0, e1), tr_mix_lo(hi.y, e2, e3));
                            } else if (ARITH == 0) {
                                split2(e0, e1, hi.x, lo.x);
                                split2(e2, e3, hi.y, lo.y);
                            } else {
                                hi = lo = make_uint2(pk16<F16>(e0, e1), pk16<F16>(e2, e3));
                            }
                            *reinterpret_cast<uint2*>(xs + hv * TR_REC) = hi;
                            if (NX == 2) *reinterpret_cast<uint2*>(xs + TR_XT + hv * TR_REC) = lo;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int hv = (tl + 256 * q) >> 3;
                            if (q == 3 && !q3) break;
                            if constexpr (FP32) {
                                *reinterpret_cast<float4*>(xs + hv * TR_REC) = make_float4(0.f, 0.f, 0.f, 0.f);
                                continue;
                            }
                            *reinterpret_cast<uint2*>(xs + hv * TR_REC) = make_uint2(0u, 0u);
                            if (NX == 2) *reinterpret_cast<uint2*>(xs + TR_XT + hv * TR_REC) = make_uint2(0u, 0u);
                        }
                    }
                }
                {   // g plane t + 2 (zeros outside the segment: the load was sent beyond the buffer; such a plane lands in a slot
                    // that is rewritten before anything multiplies with it)
                    unsigned char* const gs = G0 + TR_GSLOT(t + 2) + quad * TR_QB;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int gv = (tl + 256 * q) >> 3;
                        const float4 v = gs_[q];   // zeros where out of range (load offset beyond the buffer)
                        uint2 hi, lo;
                        if (FP32) {
                            *reinterpret_cast<float4*>(gs + gv * TR_REC) = v;
                            hi = lo = make_uint2(0u, 0u);
                        } else if (H21) {
                            hi = lo = make_uint2(tr_mix_scale(v.x, v.y, psc), tr_mix_scale(v.z, v.w, psc));
                        } else if (ARITH == 0) {
                            split2(v.x, v.y, hi.x, lo.x);
                            split2(v.z, v.w, hi.y, lo.y);
                        } else {
                            hi = lo = make_uint2(pk16<F16>(v.x, v.y), pk16<F16>(v.z, v.w));
                        }
                        if (!FP32) *reinterpret_cast<uint2*>(gs + gv * TR_REC) = hi;
                        if (NG == 2) *reinterpret_cast<uint2*>(gs + TR_GT + gv * TR_REC) = lo;
                        if (!TEM_TR_DBCOND || do_db) {
                            dbacc[0] += v.x;
                            dbacc[1] += v.y;
                            dbacc[2] += v.z;
                            dbacc[3] += v.w;
                        }
                        if (gmax)   // grid-uniform
                            gmx = __builtin_fmaxf(gmx, __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)),
                                                                       __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
                    }
                }
                TR_STAMP(t - za, 1);
                {
                    // (plain scalar arithmetic, no short-circuit operators: a branch around a load would be back)
                    const int zx = t + 5;
                    const int ldx = (int)(zx >= za - 1) & (int)(zx <= zb) & (int)(zx >= 0) & (int)(zx < D) & (int)!(TEM_TR_ABL & 8);   // wave-uniform
                    zin_ = ldx != 0;
                    // (TEM_TR_ABL & 16, timing only: every plane is read from z = 0 -- the loads stay, the HBM traffic goes)
                    const tr_rsrc_t rsx = tr_rsrc(xn + (int64_t)((TEM_TR_ABL & 16) ? 0 : zx * ldx) * xplane);
                    const unsigned mx = ldx ? 0u : OOB;   // OR-ed into the offsets (< 2^31): beyond the buffer -> zeros
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs_[q] = tr_load4(rsx, offx[q] | mx);
                    const int zg = t + 4;
                    const int ldg = (int)(zg >= za) & (int)(zg < zb) & (int)!(TEM_TR_ABL & 8);
                    const tr_rsrc_t rsg = tr_rsrc(gn + (int64_t)((TEM_TR_ABL & 16) ? 0 : zg * ldg) * gplane);
                    const unsigned mg = ldg ? 0u : OOB;
#pragma unroll
                    for (int q = 0; q < 2; ++q) gs_[q] = tr_load4(rsg, offg[q] | mg);
                }
                TR_STAMP(t - za, 2);
                __syncthreads();
                TR_STAMP(t - za, 3);
            };
            // trips of TR_UNROLL32 straight-line iterations (the compiler counts the outstanding loads exactly inside a trip and
            // waits for all of them once per trip); a segment is padded to whole trips with idle iterations whose barriers the
            // multiplying team meets
            const int t_end = za - 6 + (zb - za + 6 + TR_UNROLL32 - 1) / TR_UNROLL32 * TR_UNROLL32;
#pragma unroll 1
            for (int t = za - 6; t < t_end; t += TR_UNROLL32) {
#pragma unroll
                for (int k = 0; k < TR_UNROLL32; ++k) iteration(t + k, xa[k & 1], ga[k & 1], zin[k & 1]);
            }
        }
    }

    // ---- largest |g| (tem_conv3d_wgrad_gmax): integer max of the bit patterns, exact and order-independent ----
    if (gmax && !mteam) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmx = __builtin_fmaxf(gmx, __shfl_xor(gmx, o, 64));
        if (lane == 0) atomicMax(gmax, __builtin_bit_cast(unsigned, gmx));
    }
    // ---- bias-gradient partial of this workgroup: 32 threads per channel quad ----
    if (do_db) {
        float* red = reinterpret_cast<float*>(ldsb);  // [32][32]
        if (!mteam) {
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(tl >> 3) * 32 + quad * 4 + c] = dbacc[c];
        }
        __syncthreads();
        if (tid < 32 && cog * 32 + tid < Cout) {
            float a = 0.f;
            for (int rr = 0; rr < 32; ++rr) a += red[rr * 32 + tid];
            dbpart[(int64_t)sp * Cout + cog * 32 + tid] = a;
        }
    }
}

// h16: 0 bf16x3, 1 one fp16 term, 2 one bf16 term, 3 fp16 2x1 (g_amax required), 4 exact fp32.  Same arguments, partial-slab format and
// plan (teams: one (Cin tile, Cout tile) pair per workgroup, KS2 = 1) as k_conv_wgrad_zt.
void tem_conv_wgrad_tr_launch(int h16, unsigned nblk, const float* x, int64_t x_ld, const float* scale, const float* shift,
                              const float* g, int64_t g_ld, float* zpart, float* zdb, int N, int D, int H, int W, int Cin,
                              int Cout, int T, int nY, int nX, int zsegs, int Ss, int ncz, unsigned* gmax,
                              const unsigned* g_amax, hipStream_t s) {
    auto launch = [&](auto kern, size_t lb) {
        static std::set<const void*> sized;   // kernels whose dynamic-LDS limit was raised already
        const void* key = reinterpret_cast<const void*>(kern);
        if (!sized.count(key)) {
            (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            sized.insert(key);
        }
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lb, s, x, x_ld, scale, shift, g, g_ld, zpart, zdb, N, D, H, W, Cin, Cout,
                           T, nY, nX, zsegs, Ss, ncz, gmax, g_amax, (int64_t)0);
    };
    constexpr size_t XT = TR_XT_OF(TR_REC16), GT = TR_GT_OF(TR_REC16);
    auto launch16 = [&](auto kern, size_t lb, auto tag) {   // 16-bit x and g (the caller checked that the mode matches the type)
        using TS = decltype(tag);
        static std::set<const void*> sized;
        const void* key = reinterpret_cast<const void*>(kern);
        if (!sized.count(key)) {
            (void)hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            sized.insert(key);
        }
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lb, s, reinterpret_cast<const TS*>(x), x_ld, scale, shift,
                           reinterpret_cast<const TS*>(g), g_ld, zpart, zdb, N, D, H, W, Cin, Cout, T, nY, nX, zsegs, Ss, ncz, gmax, g_amax,
                           tem_call_cs.x);
    };
    if (tem_call_st.x == 1) launch16(&k_conv_wgrad_tr<1, tem_f16>, XT + GT + 4096, tem_f16{});
    else if (tem_call_st.x == 2) launch16(&k_conv_wgrad_tr<2, tem_bf16>, XT + GT + 4096, tem_bf16{});
    else if (h16 == 1) launch(&k_conv_wgrad_tr<1>, XT + GT);
    else if (h16 == 2) launch(&k_conv_wgrad_tr<2>, XT + GT);
    else if (h16 == 3) launch(&k_conv_wgrad_tr<3>, 2 * XT + GT);
    else if (h16 == 4) launch(&k_conv_wgrad_tr<4>, (size_t)TR_XT_OF(128) + TR_GT_OF(128));
    else launch(&k_conv_wgrad_tr<0>, 2 * XT + 2 * GT);
}
