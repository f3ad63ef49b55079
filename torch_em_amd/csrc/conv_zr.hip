// conv_zr.hip -- split-precision implicit-GEMM 3x3x3 convolution (forward and data gradient), "z-reuse" ping-pong teams.
//
// Replaces aten::convolution / the dgrad half of convolution_backward behind ConvBlock (model/unet.py:417-438) for the
// 3x3x3 layers of the levels with many patches (128^3 ... 32^3).  Arithmetic and operand layouts are those of
// k_conv_pp (conv_pp.hip): fp32 NDHWC activations, fused pre-norm while staging, operands split into 16-bit terms,
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_{bf16,f16}, two teams of four waves in opposite phases.
//
// What changed, and why.  Round-3 measurements (profiles/r03_issue_bench.txt, scripts/proto/issue_bench.hip):
// k_conv_pp's tap loop read one activation fragment from LDS and one weight fragment from L1 per MFMA triple, its
// multiplying team waited for operands (237-304 cycles per tap against 192 of MFMAs), and every other staging phase
// carried a 5500-cycle epilogue.  Here
//   * a wave owns a 4 x 8 (y, x) footprint through ALL FOUR z-planes of the team's 4 x 16 x 8 patch.  For a fixed
//     (ty, tx) the activation fragment of halo plane hz is read ONCE and multiplied with the weights of tz = 0, 1, 2 into
//     the accumulators of the output planes hz, hz - 1, hz - 2: 6 LDS reads feed 36 MFMAs (was 6 per 18), and the three
//     weight fragments of a (ty, tx) column stay in registers for 12 products (was 2).  Measured: 34.6 cycles per MFMA in
//     the tap phase against 34.0 for a bare MFMA stream -- the tap phase is matrix-pipe bound;
//   * the MFMA operands are swapped (A = weights, B = activations), so a lane ends up with 16 channels of ONE voxel: the
//     statistics are in-lane sums plus one transposing reduction per unit;
//   * the staging team is bound by INSTRUCTION ISSUE beside the partner's MFMAs (~6 cycles per plain VALU instruction,
//     ~69 per packed-fp32 one, ~32 per ds_write_b64, ~370 per 16-byte-per-line global store): the split is 14 VALU
//     instructions per 4 channels (v_cvt_pk_f16_f32, v_fma_mix{lo,hi}_f16; range guard = MODE.FP16_OVFL), this file is
//     compiled without the SLP vectoriser (no v_pk_*_f32), and outputs go through a small LDS transpose to full lines;
//   * halo records are 32 B per plane (no padding); bank conflicts are avoided by swapping the two 16-byte halves of a
//     record where bit 2 of the halo voxel index is set (8 consecutive x-voxels then cover all 8 16-byte bank groups);
//   * one kernel per epilogue mode (plain / statistics / ReLU mask): with all of them in one function hipcc spilled 100-260
//     registers, and a spill reload (vmcnt(0)) serialises the halo loads of the staging phase.
// Phase balance at 32 -> 32 channels, 2 x 128^3 (shader cycles): tap phase 11.2 k; staging 6.5 k without, 14 k with the
// epilogue of the previous unit (halo 1080 voxels, 18 16-byte loads per thread; epilogue 64 outputs per lane).
#include "tem_common.h"
#include "conv_internal.h"
#include "conv_split.h"
#include "tem_act.h"
#include <type_traits>

#ifndef TEM_ZR_PRIO
#define TEM_ZR_PRIO 1    // s_setprio of the team in its MFMA phase
#endif
#ifndef TEM_ZR_STAGE_PRIO
#define TEM_ZR_STAGE_PRIO 0   // s_setprio of the team in its staging phase (experiment)
#endif
#ifndef TEM_ZR_R0
#define TEM_ZR_R0 2      // halo planes (of 6; three 16-byte loads each) requested BEFORE the epilogue of the previous unit
#endif
#ifndef TEM_ZR_R0_16
#define TEM_ZR_R0_16 2   // ... with 16-bit tensors (three 16-byte loads per plane: all six planes fit the register ring), plain / ReLU-mask epilogues
#endif
#ifndef TEM_ZR_R0_16S
#define TEM_ZR_R0_16S 2  // ... statistics / norm-backward epilogues
#endif
#ifndef TEM_ZR_AD
#define TEM_ZR_AD 2      // activation-fragment prefetch depth (LDS reads in flight ahead of the MFMAs that use them)
#endif
#ifndef TEM_ZR_ST_AUX
#define TEM_ZR_ST_AUX 2  // cache policy of the epilogue stores: 2 = nt (streaming: the outputs do not push the halo lines that neighbouring
                         // tiles share out of the XCD's L2; measured 17.02 -> 16.82 ms fp32-class, 9.995 -> 9.89 ms amp; sc1 = 16: 16.84 / 9.94)
#endif
#ifndef TEM_ZR_ST_AUX_KS
#define TEM_ZR_ST_AUX_KS 2  // ... of split-K partial sums (read back at once by tem_splitk_epilogue)
#endif
#ifndef TEM_ZR_X32_SCHED
#define TEM_ZR_X32_SCHED 1   // exact-fp32 tap loop: 1 = (column, channel octet) steps, MFMAs round-robin over the four accumulators; 0 = plane by plane
#endif
#ifndef TEM_ZR_X32_R0_M0
#define TEM_ZR_X32_R0_M0 6   // exact fp32: halo planes requested at the start of a staging phase, per epilogue mode (plain / statistics / ReLU mask /
#define TEM_ZR_X32_R0_M1 4   // mask + norm backward): the largest counts without register spills.  (With the epilogue inside the staging phase,
#define TEM_ZR_X32_R0_M2 6   // as in the split modes, they were 6 / 2 / 5 / 3: the 18 registers of a plane overlapped the epilogue's temporaries.)
#define TEM_ZR_X32_R0_M3 5
#endif
#ifndef TEM_ZR_X32_GAP
#define TEM_ZR_X32_GAP 0     // exact-fp32 tap loop: units of 4 cycles of s_nop behind every MFMA (experiment, see zr_x32_gap: does not help)
#endif
#ifndef TEM_ZR_ABL
#define TEM_ZR_ABL 0     // harness-only ablations: 1 no halo loads, 2 no stores, 4 no weight loads, 8 no LDS writes, 16 no MFMAs
#endif

typedef float floatx4z __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4z __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t zr_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 zr_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4z v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    const floatx4z f = __builtin_bit_cast(floatx4z, v);   // whole-vector cast (element casts narrow the load)
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ uint4 zr_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4z v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int AUX>
__device__ __forceinline__ void zr_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float a, float b, float c, float d) {
    const u32x4z v = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c),
                      __builtin_bit_cast(unsigned, d)};
    // The scalar offset goes into the VGPR offset, the soffset field stays an immediate: gfx950 reads the data registers of a
    // 16-byte store after the instruction has issued, and LLVM (ROCm 7.2, GCNHazardRecognizer::createsVALUHazard) only
    // inserts the wait state before a VALU write of those registers when soffset is NOT a register.  With an SGPR soffset
    // the next row's v_cndmask overwrote the first data register in lanes 12..15 of every row of 16: the MODE 3 epilogue
    // stored the x component of row m + 1 into row m, in 1-99 % of the launches depending on register allocation
    // (scripts/race_zr_store.py; 0 of 1000 with the offset in the VGPR).
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, AUX);
}

// 8-byte variants for 16-bit activation storage (4 channels per lane in the epilogue)
__device__ __forceinline__ uint2 zr_load2u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned int u32x2z __attribute__((ext_vector_type(2)));
    const u32x2z v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_uint2(v.x, v.y);
}
template <int AUX>
__device__ __forceinline__ void zr_store2u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned a, unsigned b) {
    typedef unsigned int u32x2z __attribute__((ext_vector_type(2)));
    const u32x2z v = {a, b};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff + soff, 0, AUX);   // offset in the VGPR, as zr_store4
}
// pre-norm of two packed 16-bit activations: (h.lo * s0 + t0, h.hi * s1 + t1), computed in fp32, rounded once to the storage /
// operand type.  fp16: v_fma_mix{lo,hi}_f16 read the fp16 operand by half and round into the selected half of the destination
// (2 instructions; unpack + fma + pack would be 5).
template <bool F16>
__device__ __forceinline__ unsigned zr_norm2(unsigned h, float s0, float t0, float s1, float t1) {
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

// value of lane ^ M (M < 32): ds_swizzle in bit mode needs no index register (a __shfl_xor keeps four of them live)
template <int M>
__device__ __forceinline__ float zr_swz(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (M << 10) | 0x1f));
}

// r + (value of lane ^ 16) / r + (value of lane ^ 32) on the vector ALU: gfx950's v_permlane{16,32}_swap_b32 exchange the odd
// 16-lane rows (the upper 32-lane half) of one register with the even rows (lower half) of another; with both registers
// holding r the two results are [r0 r0 r2 r2] / [r1 r1 r3 r3] (rows), whose sum is the pairwise total in every lane.  (An LDS
// crossbar op -- ds_swizzle / ds_bpermute -- issues every ~32 cycles beside the partner team's MFMAs, a VALU op every ~6; the
// clang builtin for the swap returned the same register for both results, ROCm 7.2: hence the asm, s_nop for the VALU hazard.)
__device__ __forceinline__ float zr_xor16_sum(float r) {
    float a = r, b = r;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float zr_xor32_sum(float r) {
    float a = r, b = r;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// value of the lane a DPP control selects (row_ror / row_half_mirror / quad_perm: inside a row of 16 lanes)
template <int CTRL>
__device__ __forceinline__ float zr_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// (g0 - 2^12 h.lo, g1 - 2^12 h.hi) rounded to two fp16 in one register: h = two fp16 in one register (the hi terms),
// g = 2^12 x the fp32 values they were rounded from.  v_fma_mix{lo,hi}_f16 take the fp16 operand by half (op_sel), compute
// the fma in fp32 (exact here) and round once into the selected half of the destination: 2 instructions where
// convert / subtract / scale / convert-pack needs 7 (hipcc's vectoriser picks 4 v_cvt + v_pk_fma + v_cvt_pk instead).
__device__ __forceinline__ unsigned zr_mix_lo(unsigned h, float g0, float g1) {
    unsigned q;
    const float c = -F16_LO_SCALE;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(q) : "v"(h), "s"(c), "v"(g0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(q) : "v"(h), "s"(c), "v"(g1));
    return q;
}

#ifdef TEM_ZR_TRACE   // developer build (scripts/zr_harness.cpp): shader-clock stamps of the phases of one workgroup
#ifndef TEM_ZR_TRACE_BLOCK
#define TEM_ZR_TRACE_BLOCK 0
#endif
__device__ unsigned long long tem_zr_trace_buf[2][64][12];
#define ZR_STAMP(i)                                                                           \
    do {                                                                                      \
        if (blockIdx.x == TEM_ZR_TRACE_BLOCK && tw == 0 && lane == 0 && s < 64)               \
            tem_zr_trace_buf[team][s][i] = __builtin_amdgcn_s_memtime();                      \
    } while (0)
void tem_zr_trace_read(unsigned long long* dst) {
    (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tem_zr_trace_buf), sizeof(unsigned long long) * 2 * 64 * 12);
}
#else
#define ZR_STAMP(i) asm volatile("; ZRMARK " #i)
#endif

// Measured (scripts/proto/issue_bench.hip -DPARTNER_F32, profiles/r06_issue_bench_f32.txt): beside a wave that streams
// v_mfma_f32_32x32x2_f32 back to back, the other wave of the SIMD issues NO vector-ALU instruction at all -- 512 v_fma_f32 took
// 68 288 ticks next to a 65 896-tick MFMA loop (LDS and memory instructions are not affected; beside the 8-pass fp16 MFMAs
// a VALU instruction goes through every ~6 cycles).  The next MFMA of a dense stream waits at the vector-ALU port for the
// 64 cycles its predecessor occupies the matrix pipe, and the port is the staging team's, too: the first trace of the exact-fp32
// kernel showed the partner's staging phase starting only AFTER the tap phase (step = 55.6 k cycles of MFMAs + 10 k of staging,
// 0.84 of the pipe).  So the multiplying wave stays away from the port while its MFMA runs: s_nop for most of the 64 cycles,
// the next MFMA arrives shortly before the pipe is free, and the staging wave has the port in between.
__device__ __forceinline__ void zr_x32_gap() {
    if constexpr (TEM_ZR_X32_GAP > 0) {
        __builtin_amdgcn_sched_barrier(0);
        constexpr int FULL16 = TEM_ZR_X32_GAP / 16, REST = TEM_ZR_X32_GAP % 16;
#pragma unroll
        for (int i = 0; i < FULL16; ++i) asm volatile("s_nop 15");
        if constexpr (REST > 0) asm volatile("s_nop %0" ::"n"(REST - 1));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// a < b ? t : f on the scalar ALU, spelled out (see the exact-fp32 staging phase)
__device__ __forceinline__ unsigned zr_ssel_ltu(unsigned a, unsigned b, unsigned t, unsigned f) {
    unsigned r;
    asm volatile("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, %3, %4" : "=s"(r) : "s"(a), "s"(b), "s"(t), "s"(f) : "scc");
    return r;
}

struct ZrUnit {
    int cot, n, z0, y0, x0, ksl;
};

// NS planes per operand; F16: fp16 terms with the lo planes stored x 2^12 and their cross products in a second
// accumulator set (TEM_WL_F16X3), NS == 1 && F16: the one-term mixed mode; else bf16 terms in one accumulator set.
// MODE (launch-uniform, one epilogue per instantiation): 0 plain, 1 fused statistics (`stat`), 2 ReLU mask (`ref`),
// 3 ReLU mask + backward of the norm behind it: y = ref > 0 ? a*acc - m1 - (ref - mean)*m2r : 0 with (a, m1, m2r, mean) per
// (sample, output channel) read from `stat` (= coef[N][Cout][4], tem_norm_bwd_coef): a data gradient that lands behind a
// ReLU + norm leaves this kernel finished -- the elementwise pass over g and ref (k_norm_bwd_apply) disappears.
// KSPLIT (MODE 0 only): the input channels are cut into `ks` slices, a unit = (tile, column tile, slice) writes its raw
// partial sums into slice `ksl` of a [ks][N*D*H*W][Cout] workspace (y, y_ld = Cout) and tem_splitk_epilogue adds them up --
// the 16^3 / 32^3 levels, whose (tile, column tile) count alone cannot give every team of every CU a unit.
// WIDE (the one-term mixed modes with Cin % 32 == 0; NS == 2 then counts LDS planes, not terms): a phase stages and
// multiplies 32 input channels -- plane p of the tile holds channels 16 p .. 16 p + 15 of the chunk.  A 16-channel chunk is
// half of a 128-byte line of a 32-channel voxel record and the L2 fetches whole lines: with one chunk per phase the other
// half came back two phases (~9 MB of L2 turnover per XCD) later, from HBM again -- FETCH_SIZE showed 1.83 GB per launch
// for a 268 MB input (32 -> 32 at 2 x 128^3), and the one-term kernel ran at the HBM ceiling (5.6 TB/s, 374 us; 274 us
// with every load an L2 hit, TEM_ZR_ABL 64).  Here eight lanes request the whole line in one load instruction (two
// requests for the halves, even back to back, still fetched 0.99 GB; half a phase apart 1.38 GB).
// T (round 5): element type of x, ref and y in HBM.  A 16-bit T is the one-term mixed mode of the same type with WIDE staging
// (fp16 storage <-> fp16 operands, bf16 <-> bf16): a 32-channel voxel record is 64 bytes, FOUR lanes request it with one
// 16-byte load each (8 channels per lane, 3 load slots per thread and halo plane instead of 6), the pre-norm runs on the
// packed pairs (zr_norm2) and ONE ds_write_b128 per slot fills the tile -- without a norm (data gradients) the loaded
// registers go to LDS as they are.  Outputs are rounded once in the epilogue (8-byte pieces, 64-byte voxel rows); split-K
// partial sums stay fp32.
// X32 (round 6): EXACT fp32 on v_mfma_f32_32x32x2_f32 (use_mfma 1, TEM_WL_MFMA pack) in the same structure.  The two LDS planes of
// a team's tile hold channels 0..7 / 8..15 of the chunk as fp32 (32-byte records, the 16-byte half chosen by row parity as above):
// a staging slot writes its four normalised channels with one ds_write_b128 (no split: the staging phase gets SHORTER), a lane
// (k = lane >> 5) reads channels 8 p + 4 k + (0..3) with one ds_read_b128 per plane and feeds four MFMAs from it; the weight
// fragments are the [co/32][tap][ci/8][2][32][4] records of the exact-fp32 pack as they are (two 16-byte loads per tap and
// chunk, like the two terms of the split layouts).  Eight 16-pass MFMAs per (tap, chunk, z-plane) instead of three 8-pass ones:
// the tap phase is 5.3x as long as the fp16x3 one while staging and epilogue shrink, so the matrix pipe only idles in the
// barrier hand-overs -- this is the mode that is NOT power-limited (profiles/r06_mfma_busy_fp32.txt), where that converts.
template <int NS, bool F16, int MODE, bool KSPLIT = false, bool WIDE = false, typename T = float, bool X32 = false, bool XS = false>
__global__ __launch_bounds__(XS ? 256 : 512, XS ? 1 : 2) void k_conv_zr(
    const T* __restrict__ x, int64_t x_ld, const float* __restrict__ scale, const float* __restrict__ shift,
    const uint4* __restrict__ wp, const float* __restrict__ bias, std::conditional_t<KSPLIT, float, T>* __restrict__ y, int64_t y_ld,
    const T* __restrict__ ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout, int act, int nZ,
    int nY, int nX, float* __restrict__ stat, int nunits, const unsigned* __restrict__ in_amax, int ks, int blk,
    unsigned* __restrict__ out_amax, int64_t x_cs, int64_t y_cs) {
    static_assert(!KSPLIT || MODE == 0, "split-K units write raw partial sums");
    constexpr bool AMAX = MODE == 2 || MODE == 3;   // data gradients: max |y| as a by-product (TEM_BP_OUT_AMAX)
    float amx = 0.f;
    static_assert(!WIDE || NS == 2, "the wide one-term kernel uses the two LDS planes of the two-term layout");
    static_assert(!X32 || (NS == 2 && !F16 && !WIDE && sizeof(T) == 4), "exact fp32: two LDS planes of eight fp32 channels, fp32 tensors");
    static_assert(!XS || X32, "the one-team workgroup exists for the exact-fp32 arithmetic only");
    constexpr bool T16 = sizeof(T) == 2;         // 16-bit activations in HBM
    using TOut = std::conditional_t<KSPLIT, float, T>;
    constexpr bool Y16 = sizeof(TOut) == 2;
    constexpr int XB = (int)sizeof(T), YB = (int)sizeof(TOut);   // bytes per element of x / ref and of y
    static_assert(!T16 || WIDE, "16-bit storage: 32 input channels per phase");
    constexpr int CK = WIDE ? 2 * BCK : BCK;     // input channels per phase
    constexpr int TZ = 4, TY = 16, TX = 8;
    constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
    constexpr int HV = HZ * HY * HX;             // 1080 halo voxels
    constexpr int PLB = HV * 32;                 // bytes per plane of a team's tile (16 channels x 2 B per voxel)
    constexpr int NIT = HZ * 3;                  // float4 slots per thread: a ring of RP halo planes
    constexpr int LPV = (WIDE && !T16) ? 8 : 4;  // lanes per halo voxel: one 16-byte load each (WIDE: a whole 128-byte line; 16-bit: the 64-byte record)
    constexpr int SPP = (WIDE && !T16) ? 6 : 3;  // load slots per thread and halo plane (the last one only for part of the team)
    constexpr int CPL = T16 ? 8 : 4;             // channels per load slot
    constexpr int RP = NIT / SPP;                // planes the register ring holds
    constexpr int R0 = X32 ? (MODE == 0 ? TEM_ZR_X32_R0_M0 : MODE == 1 ? TEM_ZR_X32_R0_M1 : MODE == 2 ? TEM_ZR_X32_R0_M2 : TEM_ZR_X32_R0_M3)   // exact fp32: as much of the halo as the epilogue's registers allow, by instructions that need no vector ALU (see pvo below)
                       : (WIDE && !T16) ? ((MODE == 1 || MODE == 3) ? 1 : 2)   // (a ring of three planes: at most two ahead; the statistics / norm-backward epilogues have no room for 12 loads)
                       : T16 ? ((MODE == 1 || MODE == 3) ? TEM_ZR_R0_16S : TEM_ZR_R0_16)
                             : (TEM_ZR_R0 < HZ ? TEM_ZR_R0 : HZ);   // halo planes loaded before the epilogue
    constexpr int FR = NS * 64;                  // uint4s per (tap, 16-channel chunk) fragment group
    constexpr bool SC = F16 && NS == 2 && !WIDE;
    constexpr int ZSTEP = HY * HX * 32;          // bytes between halo z-planes: 5760 = 45 * 128 (bank-neutral)
    static_assert(HY % 2 == 0, "the record-half swizzle (row parity) is the same in every halo plane");
    extern __shared__ __attribute__((aligned(16))) unsigned char zr_lds[];   // [2 teams][NS planes][HV][32 B], [4 waves][32][144 B]

    // MODE.FP16_OVFL (bit 23): fp16 results that overflow clamp to +-65504 instead of becoming inf -- the range guard of
    // the two-term split (a normalised activation never gets there; k_conv_pp spends a v_med3 per element on the same guard)
    if (SC) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wv >> 2, tw = wv & 3, tl = tid & 255;
    const int kh = lane >> 5, v = lane & 31;
    // Footprint voxel of this lane.  ds_read_b128 is serviced in four groups of 16 lanes that are NOT contiguous
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32; MI355X_MICROARCH.md, LDS): vu numbers the lanes so that
    // each group holds two complete footprint rows (16 voxels).  With the 16-byte half of a 32-byte halo record chosen
    // by row parity the 16 reads of a group fall into 16 different bank quads: 4 LDS cycles per fragment read, where
    // lane order + the former bit-2 swizzle took 16 (SQ_LDS_BANK_CONFLICT was 67 % of SQ_LDS_IDX_ACTIVE).
    const int vu = (int)((0x73261540u >> (4 * (v >> 2))) & 7u) * 4 + (v & 3);
    const int py = vu >> 3, px = vu & 7;
    unsigned char* lds = zr_lds + team * (NS * PLB);

    const int G = (XS ? 1 : 2) * gridDim.x;   // (XS: 256 threads = one team per workgroup, team == 0)
    const int slot = XS ? tem_xcd_remap(blockIdx.x, gridDim.x) : tem_xcd_remap(blockIdx.x, gridDim.x) * 2 + team;
    const int ncot = Cout >> 5;
    const int nch_all = Cin / CK;                          // chunks of the input (16 channels; WIDE: 32)
    const int nch = KSPLIT ? nch_all / ks : nch_all;       // ... of one unit
    const int my_units = slot < nunits ? (nunits - slot + G - 1) / G : 0;
    const int P = ((nunits + G - 1) / G) * nch;

    auto decode = [&](int ui) {
        int u = slot + ui * G;
        ZrUnit t;
        t.ksl = 0;
        if (KSPLIT) { t.ksl = u % ks; u /= ks; }   // the slices of a tile are neighbours: they share its halo in L2
        t.cot = u % ncot; u /= ncot;
        // Tiles in blocks of 2^lx x 2^ly x 2^lz (x fastest inside a block, blocks x fastest): the 64 units that the teams of
        // one XCD work on at a time (tem_xcd_remap) then form a compact 3-D block whose halos overlap inside that XCD's L2
        // -- in plain x, y, z order they are a 4-voxel-thin slab whose z halo (a third of all halo lines) belongs to
        // tiles on other XCDs or in other iterations.  blk = lx | ly << 4 | lz << 8 (0: plain order).
        const int lx = blk & 15, ly = (blk >> 4) & 15, lz = blk >> 8;
        const int b = u & ((1 << (lx + ly + lz)) - 1);
        u >>= lx + ly + lz;
        const int bx = b & ((1 << lx) - 1), by = (b >> lx) & ((1 << ly) - 1), bz = b >> (lx + ly);
        const int mX = nX >> lx, mY = nY >> ly, mZ = nZ >> lz;
        t.x0 = (((u % mX) << lx) + bx) * TX; u /= mX;
        t.y0 = (((u % mY) << ly) + by) * TY; u /= mY;
        t.z0 = (((u % mZ) << lz) + bz) * TZ; u /= mZ;
        t.n = u;
        return t;
    };

    // ---- per-thread constants ----
    // staging: a halo z-plane has HY * HX = 180 voxels x 4 four-channel quarters = 720 slots: three per thread (the third
    // only for tl < 208).  Slot j of plane hz: plane voxel q_j = (tl + 256 j) >> 2, channels c4*4 .. c4*4+3 of the chunk.
    // Only the three in-plane byte offsets live in VGPRs; the plane offset is the scalar offset of the buffer load.
    // WIDE: 8 lanes per voxel (a load instruction requests whole 128-byte lines: 32 channels), six slots per thread, the sixth for
    // tl < 160; quads 4 .. 7 go to plane 1 of the tile.
    const int c4 = tl & (LPV - 1);
    unsigned poff[SPP];      // byte offset of slot j inside a halo plane (from the halo origin)
    unsigned lwj[SPP];       // LDS byte offset of slot j in any plane (16-byte half swizzled by the parity of the halo row)
#pragma unroll
    for (int j = 0; j < SPP; ++j) {
        const int q = min((tl + 256 * j) / LPV, HY * HX - 1);
        const int hy = q / HX, hx = q % HX;
        poff[j] = ((unsigned)(hy * W + hx) * (unsigned)x_ld + (unsigned)(c4 * CPL)) * (unsigned)XB;
        if (T16 || X32)   // 16-bit: channels 8 c4 .. 8 c4 + 7, exact fp32: 4 c4 .. 4 c4 + 3 -- plane c4 >> 1, 16-byte half c4 & 1 (swizzled by the row parity)
            lwj[j] = (unsigned)(q * 32) + (unsigned)((((c4 & 1) ^ (hy & 1)) << 4)) + (unsigned)((c4 >> 1) * PLB);
        else
            lwj[j] = (unsigned)(q * 32) + (unsigned)(((((c4 >> 1) & 1) ^ (hy & 1)) << 4) | ((c4 & 1) << 3)) + (unsigned)((c4 >> 2) * PLB);
    }
    // Exact fp32: the vector ALU of a SIMD is the matrix pipe of v_mfma_f32_32x32x2_f32 (zr_x32_gap above: beside the partner's
    // MFMA stream this wave issues no VALU instruction), so a staging phase must get its halo loads out WITHOUT one: the per-slot
    // offsets with the in-plane validity of the unit folded in (out-of-range slots read the always-valid plane voxel (1, 1) and
    // are zeroed later) are prepared where the unit is chosen -- behind this team's own tap phase, when the ALU is free -- and the
    // z validity is a scalar select of the plane offset.  The loads are then buffer_load + SALU only and fly during the
    // partner's MFMAs; the VALU work of the phase (epilogue, norm, LDS writes) runs as one dense burst behind them.
    const unsigned ctr_off_ = ((unsigned)(W + 1) * (unsigned)x_ld + (unsigned)(c4 * CPL)) * (unsigned)XB;
    unsigned pvo[X32 ? SPP : 1];
    unsigned yxu = (1u << SPP) - 1u;   // in-plane validity bits of the unit in `cu`
    auto unit_offsets = [&](const ZrUnit& t) {
        if constexpr (X32) {
            yxu = 0;
#pragma unroll
            for (int j = 0; j < SPP; ++j) {
                const int q = min((tl + 256 * j) / LPV, HY * HX - 1);
                const unsigned gy = (unsigned)(t.y0 - 1 + q / HX), gx = (unsigned)(t.x0 - 1 + q % HX);
                const bool ok = (gy < (unsigned)H) & (gx < (unsigned)W);
                yxu |= ok ? (1u << j) : 0u;
                pvo[X32 ? j : 0] = ok ? poff[j] : ctr_off_;
            }
        }
    };
    const bool slot2 = tl < (HY * HX * LPV - 256 * (SPP - 1));   // the last slot exists for 208 (WIDE: 160) threads
    const unsigned ctr_off = ((unsigned)(W + 1) * (unsigned)x_ld + (unsigned)(c4 * CPL)) * (unsigned)XB;  // plane voxel (1, 1): always inside
    const unsigned plane_b = (unsigned)(H * W) * (unsigned)x_ld * (unsigned)XB;   // bytes between z-planes of x
    // tap loop: halo voxel of this lane's footprint voxel at (hz, ty, tx) = (0, 0, 0)
    const int hvb = (4 * tw + py) * HX + px;
    // epilogue: this lane owns voxel (py, px) of the wave's footprint and channels 8 j + 4 kh + (0..3), j = 0..3
    // (after the LDS transpose of the epilogue: voxel row 4 tw + m with m per store, x = lane >> 3, 16-byte piece lane & 7)
    const unsigned yoff_lane = ((unsigned)(4 * tw * W + (lane >> 3)) * (unsigned)y_ld + (unsigned)(4 * (lane & 7))) * (unsigned)YB;
    const unsigned roff_lane = ((unsigned)(4 * tw * W + (lane >> 3)) * (unsigned)ref_ld + (unsigned)(4 * (lane & 7))) * (unsigned)XB;
    const __amdgpu_buffer_rsrc_t rw = zr_rsrc(wp);
    const unsigned woff_lane = (unsigned)lane * 16u;
    const float act_floor = act == TEM_ACT_RELU ? 0.f : -__builtin_inff();
    // in_amax (tem_conv3d_fwd_gscaled: a data gradient in the fp16 two-term layout): the input is an unnormalised
    // gradient, so it is multiplied by the power of two that puts its largest magnitude into [2^14, 2^15) -- exact, inside
    // fp16's range with 29 binades below the maximum before a hi term goes subnormal -- and the output by its inverse.
    float psc = 1.f, pinv = 1.f;
    if (in_amax) {
        const int e = (int)((*in_amax >> 23) & 0xffu);                      // biased exponent of max |x| (0: all zeros)
        const int k = e == 0 ? 0 : min(max(141 - e, -100), 100);            // amax * 2^k in [2^14, 2^15)
        psc = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
        pinv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    }

    floatx16 acc[TZ];
    floatx16 accl[SC ? TZ : 1];
    const int tapstride = nch_all * FR;  // uint4s between taps of the packed weights

    int ui = 0, ci = 0;
    ZrUnit cu = decode(0), eu = cu;
    unit_offsets(cu);
    bool epi_pending = false;
    // bias: lane l keeps channel (l & 31) of the unit being computed.  It is added in the epilogue: accumulators that START
    // from it measure worse -- the matrix core truncates each step's products against the larger accumulator, a one-sided
    // error that a GroupNorm backward sums over all voxels (1.5e-3 on the first layer's bias gradient).
    float bv = (bias && my_units > 0) ? bias[cu.cot * 32 + v] : 0.f;
    float bve = 0.f;   // ... of the unit whose accumulators wait for their epilogue
    auto bias16 = [&](float* b16, float bsrc) {   // the 16 channels of this lane's accumulator registers: 8 (i >> 2) + 4 kh + (i & 3)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c0 = 8 * (i >> 2) + (i & 3);
            const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bsrc), c0));
            const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bsrc), c0 + 4));
            b16[i] = kh ? hi : lo;
        }
    };
#pragma unroll
    for (int z = 0; z < TZ; ++z)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[z][i] = 0.f;
            if (SC) accl[SC ? z : 0][i] = 0.f;
        }
    unsigned wsoff = 0;

    // Exact fp32: the epilogue runs straight behind the last tap phase of a unit (see there); the same text as in the staging phase.
    auto epilogue_now = [&](unsigned yoff_l, unsigned roff_l, int s) {
        (void)s;
        if constexpr (X32) {
            // lane-derived values from a laundered lane id: left to itself hipcc computes the two dozen per-lane addresses of the
            // epilogue once in front of the unit loop, keeps them live through it and spills 60 registers to do so
            int lane_ = (int)(threadIdx.x & 63);
            asm volatile("" : "+v"(lane_));
            const int lane = lane_, kh = lane >> 5, v = lane & 31;
            const int vu = (int)((0x73261540u >> (4 * (v >> 2))) & 7u) * 4 + (v & 3);
            const int py = vu >> 3, px = vu & 7;
            (void)py; (void)px; (void)kh;
#include "conv_zr_epilogue.inc"
        }
    };
    // =====================================================================================================================
    // XS (round 6): exact fp32 with ONE team per workgroup.  Beside a stream of v_mfma_f32_32x32x2_f32 the partner wave of a SIMD
    // issues no vector-ALU instruction (zr_x32_gap above), so a second team cannot stage "behind" the MFMAs: whatever it does on
    // the vector ALU waits until they are over and then runs exposed, once per phase, with two workgroup barriers around it.  Here
    // the four waves that multiply also stage: the tile of the NEXT chunk (the other half of the same LDS) is filled from inside
    // the tap loop -- one halo slot per step of 48 MFMAs: a buffer_load four steps ahead, four fused multiply-adds, one
    // ds_write_b128 -- so the staging costs its own issue slots and nothing else, and there is one barrier per phase.  The
    // epilogue runs behind the unit's last tap phase (same text, conv_zr_epilogue.inc).
    // =====================================================================================================================
    if constexpr (XS) {
        constexpr int PFD = 4;                       // halo slots in flight ahead of their conversion
        constexpr int NSL = HZ * SPP;                // 18 slots per thread and chunk = the steps of the tap loop
        static_assert(NSL == 18, "one halo slot per step of the tap loop");
        const int nph = my_units * nch;
        uint4 wq[2][3][2], wqn[3][2];
        float4 tn[PFD + 1];
        float4 scn = make_float4(1.f, 1.f, 1.f, 1.f), sfn = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned zso[HZ];                            // scalar plane offsets of the chunk being staged (z validity folded in)
        unsigned inbn[2] = {0xffffffffu, 0xffffffffu};
        bool interior_n = true;
        __amdgpu_buffer_rsrc_t rxn = zr_rsrc(x);
        auto stage_setup = [&](const ZrUnit& t, int c) {
            const T* xb = x + ((((int64_t)t.n * D + (t.z0 - 1)) * H + (t.y0 - 1)) * W + (t.x0 - 1)) * x_ld + (int64_t)((t.ksl * nch + c) * CK);
            rxn = zr_rsrc(xb);
            interior_n = (t.z0 >= 1) & (t.z0 + HZ - 1 <= D) & (t.y0 >= 1) & (t.y0 + HY - 1 <= H) & (t.x0 >= 1) & (t.x0 + HX - 1 <= W);
#pragma unroll
            for (int hz = 0; hz < HZ; ++hz) zso[hz] = zr_ssel_ltu((unsigned)(t.z0 - 1 + hz), (unsigned)D, (unsigned)hz * plane_b, plane_b);
            if (scale) {
                scn = *reinterpret_cast<const float4*>(scale + (int64_t)t.n * Cin + (t.ksl * nch + c) * CK + c4 * CPL);
                sfn = *reinterpret_cast<const float4*>(shift + (int64_t)t.n * Cin + (t.ksl * nch + c) * CK + c4 * CPL);
            }
            inbn[0] = inbn[1] = 0xffffffffu;
            if (!interior_n) {
                inbn[0] = inbn[1] = 0;
#pragma unroll
                for (int hz = 0; hz < HZ; ++hz) {
                    const bool zok = (unsigned)(t.z0 - 1 + hz) < (unsigned)D;
#pragma unroll
                    for (int j = 0; j < SPP; ++j) inbn[hz / 3] |= (zok & ((yxu >> j) & 1u)) ? (1u << ((hz % 3) * SPP + j)) : 0u;
                }
            }
        };
        auto stage_load = [&](int s) {               // s compile-time after unrolling
            if (TEM_ZR_ABL & 1) tn[s % (PFD + 1)] = make_float4(0.5f, 0.25f, -1.f, 2.f);
            else tn[s % (PFD + 1)] = zr_load4(rxn, pvo[X32 ? s % SPP : 0], zso[s / SPP]);
        };
        auto stage_conv = [&](int s, unsigned char* tile) {
            const int hz = s / SPP, j = s % SPP;
            const float4 t4 = tn[s % (PFD + 1)];
            float e[4] = {fmaf(t4.x, scn.x, sfn.x), fmaf(t4.y, scn.y, sfn.y), fmaf(t4.z, scn.z, sfn.z), fmaf(t4.w, scn.w, sfn.w)};
            if (!interior_n) {   // zero padding comes after the norm (model/unet.py:429-438)
                const float m = ((inbn[hz / 3] >> ((hz % 3) * SPP + j)) & 1u) ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) e[c] *= m;
            }
            if (j < SPP - 1 || slot2) *reinterpret_cast<float4*>(tile + lwj[j] + hz * ZSTEP) = make_float4(e[0], e[1], e[2], e[3]);
        };
        auto wbase = [&](const ZrUnit& t, int c) { return (unsigned)((t.cot * 27 * nch_all + t.ksl * nch + c) * FR) * 16u; };
        auto comp = [](const uint4& q, int c) {
            return __builtin_bit_cast(float, c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w)));
        };
        // ---- first chunk of the first unit: staged without anything to hide behind ----
        if (nph > 0) {
            stage_setup(cu, 0);
#pragma unroll
            for (int s = 0; s < PFD; ++s) stage_load(s);
#pragma unroll
            for (int s = 0; s < NSL; ++s) {
                if (s + PFD < NSL) stage_load(s + PFD);
                stage_conv(s, zr_lds);
            }
            const unsigned wb = wbase(cu, 0);
#pragma unroll
            for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                for (int p = 0; p < 2; ++p) wq[0][tz][p] = zr_load4u(rw, woff_lane, wb + (unsigned)((tz * 9) * tapstride + p * 64) * 16u);
        }
        __syncthreads();
        for (int ph = 0; ph < nph; ++ph) {
            unsigned char* const tcur = zr_lds + (ph & 1) * (NS * PLB);
            unsigned char* const tnxt = zr_lds + ((ph + 1) & 1) * (NS * PLB);
            const bool last_chunk = ci + 1 == nch;
            // the chunk staged during this phase: the next one -- or, in the very last phase, this one again (into the free tile:
            // the loop has one shape, and the few wasted instructions of one phase per workgroup buy half the code size)
            ZrUnit tu = cu;
            int tc = ci;
            if (ph + 1 < nph) {
                if (last_chunk) {
                    tu = decode(ui + 1);
                    tc = 0;
                    unit_offsets(tu);
                } else
                    tc = ci + 1;
            }
            stage_setup(tu, tc);
#pragma unroll
            for (int s = 0; s < PFD; ++s) stage_load(s);
            wsoff = wbase(cu, ci);
            const unsigned wsoff_n = wbase(tu, tc);
            int ts = tapstride;
            asm volatile("" : "+s"(ts));
            const unsigned hsel = (unsigned)((py & 1) ^ kh) << 4;
            int hvb_ = hvb;
            asm volatile("" : "+v"(hvb_));
            unsigned a0 = 0u;
            auto col_addr = [&](int g) {
                const int hvv = hvb_ + (g / 3) * HX + (g % 3);
                a0 = (unsigned)hvv * 32u + (((g / 3) & 1) ? hsel ^ 16u : hsel);
            };
            uint4 xr[2][HZ];
            auto r_read = [&](int u) {
                if ((u & 1) == 0) col_addr(u >> 1);
#pragma unroll
                for (int hz = 0; hz < HZ; ++hz) xr[u & 1][hz] = *reinterpret_cast<const uint4*>(tcur + a0 + hz * ZSTEP + (u & 1) * PLB);
            };
            r_read(0);
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                const int g = u >> 1, p = u & 1;
                if (u + PFD < NSL) stage_load(u + PFD);
                if (u + 1 < 18) r_read(u + 1);
                if (g + 1 < 9 && !(TEM_ZR_ABL & 4)) {   // weight fragments of the next column: tz = 0, 1 in the first step, 2 in the second
                    const int gn = g + 1;
#pragma unroll
                    for (int tz = (p ? 2 : 0); tz < (p ? 3 : 2); ++tz)
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
                            wq[gn & 1][tz][pp] = zr_load4u(rw, woff_lane, wsoff + (unsigned)((tz * 9 + gn) * ts + pp * 64) * 16u);
                }
                if (u >= 15) {   // first column of the next phase, into its own registers (wq[0] is in use until the last step)
                    const int tz = u - 15;
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) wqn[tz][pp] = zr_load4u(rw, woff_lane, wsoff_n + (unsigned)((tz * 9) * ts + pp * 64) * 16u);
                }
                stage_conv(u, tnxt);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                        for (int z = 0; z < TZ; ++z)
                            acc[z] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wq[g & 1][tz][p], c), comp(xr[u & 1][z + tz], c), acc[z], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last_chunk) {
                eu = cu;
                bve = bv;
                unsigned yo = yoff_lane, ro = roff_lane;
                asm volatile("" : "+v"(yo), "+v"(ro));
                epilogue_now(yo, ro, ph);
                bv = bias ? bias[tu.cot * 32 + v] : 0.f;
                ++ui;
            }
            cu = tu;
            ci = tc;
#pragma unroll
            for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                for (int p = 0; p < 2; ++p) wq[0][tz][p] = wqn[tz][p];
            __syncthreads();
        }
        if (AMAX && out_amax) tem_amax_commit(out_amax, amx);
        return;
    }
    if (team) __syncthreads();
    for (int s = 0; s <= P; ++s) {
        const bool do_stage = ui < my_units;
        uint4 wq[2][3][NS];   // weight fragments of the (ty, tx) column in use and of the next one (declared per step: a
                              // function-scope array is loop-carried for hipcc and gets spilled across the staging phase)
        {
            // ================= staging phase (the partner team runs its tap loop) =================
            ZR_STAMP(0);
            if (TEM_ZR_STAGE_PRIO) __builtin_amdgcn_s_setprio(TEM_ZR_STAGE_PRIO);
            // keep hipcc from hoisting the 18 LDS / 18 global addresses of a phase out of the unit loop (it spills them)
#pragma unroll
            for (int j = 0; j < SPP; ++j) asm volatile("" : "+v"(lwj[j]), "+v"(poff[j]));
            unsigned yoff_l = yoff_lane, roff_l = roff_lane;
            asm volatile("" : "+v"(yoff_l), "+v"(roff_l));
            float4 tmp[NIT];
            unsigned inb[2] = {0xffffffffu, 0xffffffffu};   // validity of the slots of halo planes 0..2 / 3..5 (border patches)
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sf4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 sc5 = sc4, sf5 = sf4;   // 16-bit storage: channels 4 .. 7 of this thread's eight
            bool interior = true;
            __amdgpu_buffer_rsrc_t rx = zr_rsrc(x);
            auto load_norm = [&]() {
                if (scale) {
                    sc4 = *reinterpret_cast<const float4*>(scale + (int64_t)cu.n * Cin + (cu.ksl * nch + ci) * CK + c4 * CPL);
                    sf4 = *reinterpret_cast<const float4*>(shift + (int64_t)cu.n * Cin + (cu.ksl * nch + ci) * CK + c4 * CPL);
                    if (T16) {
                        sc5 = *reinterpret_cast<const float4*>(scale + (int64_t)cu.n * Cin + (cu.ksl * nch + ci) * CK + c4 * CPL + 4);
                        sf5 = *reinterpret_cast<const float4*>(shift + (int64_t)cu.n * Cin + (cu.ksl * nch + ci) * CK + c4 * CPL + 4);
                    }
                } else if (in_amax) {
                    sc4 = make_float4(psc, psc, psc, psc);
                }
            };
            if (do_stage) {
                if constexpr (!X32) load_norm();   // (exact fp32: behind the halo loads -- its address arithmetic is vector-ALU work)
                // the halo origin may lie outside the tensor for border patches (only in-range voxels are dereferenced)
                // (16-bit storage: a chunk stride x_cs != 0 puts chunk k at x + k * x_cs -- planar concat halves, tem_act.h)
                const int64_t xch = (T16 && x_cs) ? (int64_t)(cu.ksl * nch + ci) * x_cs : (int64_t)((cu.ksl * nch + ci) * CK);
                const T* xb = x + ((((int64_t)cu.n * D + (cu.z0 - 1)) * H + (cu.y0 - 1)) * W + (cu.x0 - 1)) * x_ld + xch;
                rx = zr_rsrc((TEM_ZR_ABL & 64) ? x + xch : xb);   // timing experiment: every unit reads the halo at the origin (L2 hits)
                interior = (cu.z0 >= 1) & (cu.z0 + HZ - 1 <= D) & (cu.y0 >= 1) & (cu.y0 + HY - 1 <= H) & (cu.x0 >= 1) &
                           (cu.x0 + HX - 1 <= W);
                if (TEM_ZR_ABL & 32) interior = true;   // timing experiment: border code paths compiled out (wrong at the faces)
                if constexpr (X32) {
                    // the whole halo: 18 buffer loads whose offsets exist already (pvo; the plane offset with the z validity is a
                    // scalar select written as SALU instructions: hipcc legalises a uniform bool that crosses a block through a
                    // VGPR -- v_cndmask / v_cmp -- and one such instruction in front of the loads holds them back for the whole
                    // tap phase of the partner) -- no vector ALU
#pragma unroll
                    for (int hz = 0; hz < R0; ++hz) {
                        const unsigned so = zr_ssel_ltu((unsigned)(cu.z0 - 1 + hz), (unsigned)D, (unsigned)hz * plane_b, plane_b);
#pragma unroll
                        for (int j = 0; j < SPP; ++j) {
                            if (TEM_ZR_ABL & 1) tmp[hz * SPP + j] = make_float4(0.5f + hz, 0.25f, -1.f, 2.f);
                            else tmp[hz * SPP + j] = zr_load4(rx, pvo[X32 ? j : 0], so);
                        }
                    }
                    load_norm();   // (scalar base + a loop-invariant lane offset: no vector ALU either)
                }
            }
            unsigned yx = (1u << SPP) - 1u;   // in-plane validity of the slots: border patches only, once per phase, from a laundered
            if constexpr (X32) {
                if (do_stage) {
                    yx = yxu;
                    if (!interior) {
                        inb[0] = inb[1] = 0;
#pragma unroll
                        for (int hz = 0; hz < HZ; ++hz) {
                            const bool zok = (unsigned)(cu.z0 - 1 + hz) < (unsigned)D;
#pragma unroll
                            for (int j = 0; j < SPP; ++j)
                                inb[hz / 3] |= (zok & ((yx >> j) & 1u)) ? (1u << ((hz % 3) * SPP + j)) : 0u;
                        }
                    }
                }
            } else
            if (do_stage && !interior) {   // copy of tl (loop-invariant code motion would keep six more registers live)
                int tl_ = tl;
                asm volatile("" : "+v"(tl_));
                yx = 0;
#pragma unroll
                for (int j = 0; j < SPP; ++j) {
                    const int q = min((tl_ + 256 * j) / LPV, HY * HX - 1);
                    const unsigned gy = (unsigned)(cu.y0 - 1 + q / HX), gx = (unsigned)(cu.x0 - 1 + q % HX);
                    yx |= ((gy < (unsigned)H) & (gx < (unsigned)W)) ? (1u << j) : 0u;
                }
            }
            // the SPP loads of halo plane hz live in slot hz % RP of the register ring
            auto issue_loads = [&](auto interior_tag, auto lo_tag, auto hi_tag) {   // halo planes LO .. HI-1
                constexpr bool INTERIOR = decltype(interior_tag)::value;
                constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
                if (INTERIOR) {
#pragma unroll
                    for (int hz = LO; hz < HI; ++hz)
#pragma unroll
                        for (int j = 0; j < SPP; ++j) {
                            if (TEM_ZR_ABL & 1) tmp[(hz % RP) * SPP + j] = make_float4(0.5f + hz, 0.25f, -1.f, 2.f);
                            else tmp[(hz % RP) * SPP + j] = zr_load4(rx, poff[j], (unsigned)hz * plane_b);
                        }
                } else {
                    if (LO == 0) inb[0] = inb[1] = 0;
#pragma unroll
                    for (int hz = LO; hz < HI; ++hz) {
                        const bool zok = (unsigned)(cu.z0 - 1 + hz) < (unsigned)D;   // wave-uniform
#pragma unroll
                        for (int j = 0; j < SPP; ++j) {
                            const bool ok = zok & ((yx >> j) & 1u);
                            inb[hz / 3] |= ok ? (1u << ((hz % 3) * SPP + j)) : 0u;
                            tmp[(hz % RP) * SPP + j] = zr_load4(rx, (ok ? poff[j] : ctr_off) + (ok ? (unsigned)hz * plane_b : plane_b), 0);
                        }
                    }
                }
            };
            if (do_stage && !X32) {
                if (interior) issue_loads(std::true_type{}, std::integral_constant<int, 0>{}, std::integral_constant<int, R0>{});
                else issue_loads(std::false_type{}, std::integral_constant<int, 0>{}, std::integral_constant<int, R0>{});
            }
            ZR_STAMP(1);
            // ---- epilogue of the unit whose last chunk this team computed in its previous phase ----
            if (!X32 && epi_pending) {
#include "conv_zr_epilogue.inc"
                epi_pending = false;
                ZR_STAMP(11);
            }
            ZR_STAMP(6);
            // ---- norm, split, LDS tile, R0 planes behind the loads; then prime the weight fragments of the first column ----
            if (do_stage) {
                // Measured (scripts/proto/issue_bench.hip): beside a wave that streams MFMAs a plain VALU instruction of this
                // wave issues every ~6 cycles, a v_fma_mix*_f16 every ~9-12, a ds_write_b64 every ~32 -- but a PACKED fp32
                // instruction (v_pk_fma_f32 / v_pk_mul_f32) only every ~69: they queue behind the matrix pipe.  So: scalar
                // fp32 math, and per 4 channels 4 v_fma (norm), 2 v_cvt_pk_f16 (hi), 4 v_mul (g = 2^12 e)
                // + 4 v_fma_mix{lo,hi}_f16 (lo' = g - 2^12 hi, exact in fp32, rounded once), 2 ds_write_b64.
                // Range: MODE.FP16_OVFL clamps instead of producing inf (set at kernel entry for the two-term layout).
                auto convert = [&](auto interior_tag, auto hz_tag) {
                    constexpr bool INTERIOR = decltype(interior_tag)::value;
                    constexpr int hz = decltype(hz_tag)::value;
#pragma unroll
                    for (int j = 0; j < SPP; ++j) {
                        const int it = (hz % RP) * SPP + j;
                        if constexpr (T16) {
                            if (j < SPP - 1 || slot2) {
                                u32x4z raw = __builtin_bit_cast(u32x4z, floatx4z{tmp[it].x, tmp[it].y, tmp[it].z, tmp[it].w});
                                if (scale) {   // launch-uniform; without a norm the operands ARE the stored values
                                    raw.x = zr_norm2<F16>(raw.x, sc4.x, sf4.x, sc4.y, sf4.y);
                                    raw.y = zr_norm2<F16>(raw.y, sc4.z, sf4.z, sc4.w, sf4.w);
                                    raw.z = zr_norm2<F16>(raw.z, sc5.x, sf5.x, sc5.y, sf5.y);
                                    raw.w = zr_norm2<F16>(raw.w, sc5.z, sf5.z, sc5.w, sf5.w);
                                }
                                if (!INTERIOR) {   // zero padding comes after the norm (model/unet.py:429-438)
                                    const bool ok = ((inb[hz / 3] >> ((hz % 3) * SPP + j)) & 1u) != 0;
                                    raw.x = ok ? raw.x : 0u; raw.y = ok ? raw.y : 0u; raw.z = ok ? raw.z : 0u; raw.w = ok ? raw.w : 0u;
                                }
                                if (!(TEM_ZR_ABL & 8) || raw.x == 0x12345678u)
                                    *reinterpret_cast<uint4*>(lds + lwj[j] + hz * ZSTEP) = make_uint4(raw.x, raw.y, raw.z, raw.w);
                            }
                        } else
                        if (j < SPP - 1 || slot2) {
                            float m = 1.f;
                            if (!INTERIOR) m = ((inb[hz / 3] >> ((hz % 3) * SPP + j)) & 1u) ? 1.f : 0.f;   // zero padding comes after the norm (model/unet.py:429-438)
                            float e[4] = {fmaf(tmp[it].x, sc4.x, sf4.x), fmaf(tmp[it].y, sc4.y, sf4.y),
                                          fmaf(tmp[it].z, sc4.z, sf4.z), fmaf(tmp[it].w, sc4.w, sf4.w)};
                            if (!INTERIOR) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) e[c] *= m;
                            }
                            unsigned char* dstp = lds + lwj[j] + hz * ZSTEP;   // HY is even: the row parity does not depend on the plane
                            if constexpr (X32) {
                                if (!(TEM_ZR_ABL & 8) || e[0] == 12345.678f) *reinterpret_cast<float4*>(dstp) = make_float4(e[0], e[1], e[2], e[3]);
                            } else
                            if (SC) {
                                const half2_t hh0 = {(_Float16)e[0], (_Float16)e[1]}, hh1 = {(_Float16)e[2], (_Float16)e[3]};
                                unsigned u0 = __builtin_bit_cast(unsigned, hh0), u1 = __builtin_bit_cast(unsigned, hh1);
                                if (!(TEM_ZR_ABL & 8) || u0 == 0x12345678u) *reinterpret_cast<uint2*>(dstp) = make_uint2(u0, u1);
                                const float g[4] = {e[0] * F16_LO_SCALE, e[1] * F16_LO_SCALE, e[2] * F16_LO_SCALE, e[3] * F16_LO_SCALE};
                                const unsigned q0 = zr_mix_lo(u0, g[0], g[1]), q1 = zr_mix_lo(u1, g[2], g[3]);
                                if (!(TEM_ZR_ABL & 8) || u0 == 0x12345678u) *reinterpret_cast<uint2*>(dstp + PLB) = make_uint2(q0, q1);
                            } else if (WIDE) {   // one term; lwj carries the plane of this thread's channel quad
#pragma unroll
                                for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(e[c]));   // one rounding after the fp32 norm, as autocast
                                const unsigned h0 = pk16<F16>(e[0], e[1]), h1 = pk16<F16>(e[2], e[3]);
                                if (!(TEM_ZR_ABL & 8) || h0 == 0x12345678u) *reinterpret_cast<uint2*>(dstp) = make_uint2(h0, h1);
                            } else {
                                if (F16 && NS == 1) {   // one rounding to fp16 after the fp32 norm, as autocast (see conv_pp.hip)
#pragma unroll
                                    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(e[c]));
                                }
#pragma unroll
                                for (int p = 0; p < NS; ++p) {
                                    const unsigned h0 = pk16<F16>(e[0], e[1]), h1 = pk16<F16>(e[2], e[3]);
                                    if (!(TEM_ZR_ABL & 8) || h0 == 0x12345678u)
                                        *reinterpret_cast<uint2*>(dstp + p * PLB) = make_uint2(h0, h1);
                                    if (p + 1 < NS) {
                                        e[0] -= lo16<F16>(h0);
                                        e[1] -= hi16<F16>(h0);
                                        e[2] -= lo16<F16>(h1);
                                        e[3] -= hi16<F16>(h1);
                                    }
                                }
                            }
                        }
                    }
                };
                auto stage_planes = [&](auto interior_tag) {
#define ZR_PLANE(HZI)                                                                                              \
    do {                                                                                                           \
        if ((HZI) + R0 < HZ)                                                                                       \
            issue_loads(interior_tag, std::integral_constant<int, ((HZI) + R0 < HZ ? (HZI) + R0 : 0)>{},           \
                        std::integral_constant<int, ((HZI) + R0 < HZ ? (HZI) + R0 + 1 : 0)>{});                    \
        convert(interior_tag, std::integral_constant<int, (HZI)>{});                                               \
    } while (0)
                    ZR_PLANE(0); ZR_PLANE(1); ZR_PLANE(2); ZR_PLANE(3); ZR_PLANE(4); ZR_PLANE(5);
#undef ZR_PLANE
                };
                if (interior) stage_planes(std::true_type{});
                else stage_planes(std::false_type{});
                ZR_STAMP(7);
                wsoff = (unsigned)((cu.cot * 27 * nch_all + cu.ksl * nch + ci) * FR) * 16u;   // (column tile, tap 0, chunk) of the packed weights
#pragma unroll
                for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        wq[0][tz][p] = zr_load4u(rw, woff_lane, wsoff + (unsigned)((tz * 9) * tapstride + p * 64) * 16u);
            }
            ZR_STAMP(2);
            if (TEM_ZR_STAGE_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        {
            ZR_STAMP(3);
            // ================= MFMA phase: 27 taps of one 16-channel chunk out of this team's tile =================
            if (do_stage) {
                int ts = tapstride;
                asm volatile("" : "+s"(ts));
                if (TEM_ZR_PRIO) __builtin_amdgcn_s_setprio(TEM_ZR_PRIO);
                // activation fragment of step st = g * 6 + hz (g = ty * 3 + tx): halo voxel hvb + (hz * HY + ty) * HX + tx,
                // 16-byte half kh ^ parity(halo row); the parity is the same in all planes (HY even), so one address per
                // column serves its six planes through immediate offsets
                constexpr int NSTEP = 9 * HZ;
                constexpr int AD = TEM_ZR_AD;
                uint4 af[AD + 1][NS];
                unsigned a0 = 0u;
                const unsigned hsel = (unsigned)((py & 1) ^ kh) << 4;   // 4 tw is even: row parity = (py + ty) & 1
                int hvb_ = hvb;
                asm volatile("" : "+v"(hvb_));   // the 18 column addresses are computed in place, not hoisted and spilled
                auto col_addr = [&](int g) {
                    const int hvv = hvb_ + (g / 3) * HX + (g % 3);
                    a0 = (unsigned)hvv * 32u + (((g / 3) & 1) ? hsel ^ 16u : hsel);
                };
                auto a_read = [&](int st) {   // st compile-time after unrolling
                    const int hz = st % HZ;
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        af[st % (AD + 1)][p] = *reinterpret_cast<const uint4*>(lds + a0 + hz * ZSTEP + p * PLB);
                };
                if constexpr (X32 && TEM_ZR_X32_SCHED == 1) {
                    // Exact fp32.  A step = (column g, channel octet p): the six halo planes of the column are read once (one
                    // ds_read_b128 each: channels 8 p + 4 k + (0..3) of this lane's voxel) and each of the four k-steps multiplies
                    // its twelve (plane, z tap) pairs ROUND-ROBIN over the four accumulators -- a v_mfma_f32_32x32x2_f32 cannot start
                    // before its predecessor on the same accumulator has written back, and this SIMD has no second multiplying wave
                    // to fill such a gap: here three other MFMAs (192 cycles) lie between two that share an accumulator.
                    // Two register sets of six fragments: the reads of step u + 1 fly during the 48 MFMAs of step u.
                    uint4 xr[2][HZ];
                    auto r_read = [&](int u) {   // compile-time after unrolling
                        if ((u & 1) == 0) col_addr(u >> 1);
#pragma unroll
                        for (int hz = 0; hz < HZ; ++hz)
                            xr[u & 1][hz] = *reinterpret_cast<const uint4*>(lds + a0 + hz * ZSTEP + (u & 1) * PLB);
                    };
                    auto comp = [](const uint4& q, int c) {
                        return __builtin_bit_cast(float, c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w)));
                    };
                    r_read(0);
#pragma unroll
                    for (int u = 0; u < 18; ++u) {
                        const int g = u >> 1, p = u & 1;
                        if (u + 1 < 18) r_read(u + 1);
                        if (g + 1 < 9 && !(TEM_ZR_ABL & 4)) {   // weight fragments of the next column: tz = 0, 1 in the first step, 2 in the second
                            const int gn = g + 1;
#pragma unroll
                            for (int tz = (p ? 2 : 0); tz < (p ? 3 : 2); ++tz)
#pragma unroll
                                for (int pp = 0; pp < 2; ++pp)
                                    wq[gn & 1][tz][pp] = zr_load4u(rw, woff_lane, wsoff + (unsigned)((tz * 9 + gn) * ts + pp * 64) * 16u);
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int tz = 0; tz < 3; ++tz)
#pragma unroll
                                for (int z = 0; z < TZ; ++z) {
                                    if (TEM_ZR_ABL & 16) {
                                        asm volatile("" ::"v"(xr[u & 1][z + tz].x), "v"(wq[g & 1][tz][p].x));
                                        continue;
                                    }
                                    acc[z] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wq[g & 1][tz][p], c), comp(xr[u & 1][z + tz], c), acc[z], 0, 0, 0);
                                    zr_x32_gap();
                                }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (X32) {
                    // (first schedule, TEM_ZR_X32_SCHED 0: kept for the A/B.)  Fragment order inside a column: halo planes 0, 5, 1, 2, 3, 4.  Planes 0 and 5 feed ONE accumulator
                    // each (output planes 0 / 3): their MFMAs alternate, so that no v_mfma_f32_32x32x2_f32 directly follows the one
                    // that writes its accumulator (a dependent one cannot start before its predecessor has written back, and this
                    // SIMD has no second multiplying wave to fill the gap); planes 1..4 feed two or three accumulators in turn.
                    // A ring of four fragments (2 x 16 bytes each), two ahead of the MFMAs; 96 MFMAs of 16 passes per column.
                    constexpr int NF = 9 * HZ, RING = 4, PF = 2;
                    uint4 xf[RING][2];
                    auto x_read = [&](int f) {   // fragment f = g * 6 + i, i-th plane of the order above (compile-time after unrolling)
                        const int i = f % HZ;
                        const int hz = i == 0 ? 0 : (i == 1 ? HZ - 1 : i - 1);
                        if (i == 0) col_addr(f / HZ);
#pragma unroll
                        for (int p = 0; p < 2; ++p) xf[f % RING][p] = *reinterpret_cast<const uint4*>(lds + a0 + hz * ZSTEP + p * PLB);
                    };
                    auto comp = [](const uint4& q, int c) {
                        return __builtin_bit_cast(float, c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w)));
                    };
#pragma unroll
                    for (int f = 0; f < 2 + PF; ++f) x_read(f);
#pragma unroll
                    for (int g = 0; g < 9; ++g) {
#pragma unroll
                        for (int sub = 0; sub < 5; ++sub) {
                            const int flast = g * HZ + (sub == 0 ? 1 : sub + 1);   // last fragment this substep multiplies
                            if (sub == 0 && g != 0 && flast + PF - 1 < NF) x_read(flast + PF - 1);   // two fragments are consumed here
                            if (!(g == 0 && sub == 0) && flast + PF < NF) x_read(flast + PF);
                            // weight fragments of the next column, one tz per substep
                            if (sub < 3 && g + 1 < 9 && !(TEM_ZR_ABL & 4)) {
                                const int gn = g + 1, tz = sub;
                                const int tap = tz * 9 + gn;
#pragma unroll
                                for (int p = 0; p < 2; ++p)
                                    wq[gn & 1][tz][p] = zr_load4u(rw, woff_lane, wsoff + (unsigned)(tap * ts + p * 64) * 16u);
                            }
                            if (sub == 0) {
                                const uint4* a_lo = xf[(g * HZ) % RING];
                                const uint4* a_hi = xf[(g * HZ + 1) % RING];
#pragma unroll
                                for (int p = 0; p < 2; ++p)
#pragma unroll
                                    for (int c = 0; c < 4; ++c) {
                                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wq[g & 1][0][p], c), comp(a_lo[p], c), acc[0], 0, 0, 0);
                                        acc[TZ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wq[g & 1][2][p], c), comp(a_hi[p], c), acc[TZ - 1], 0, 0, 0);
                                    }
                            } else {
                                const int hz = sub;   // halo planes 1 .. 4
                                const uint4* a = xf[(g * HZ + sub + 1) % RING];
#pragma unroll
                                for (int p = 0; p < 2; ++p)
#pragma unroll
                                    for (int c = 0; c < 4; ++c)
#pragma unroll
                                        for (int tz = 0; tz < 3; ++tz) {
                                            const int z = hz - tz;
                                            if (z < 0 || z >= TZ) continue;
                                            acc[z] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wq[g & 1][tz][p], c), comp(a[p], c), acc[z], 0, 0, 0);
                                        }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                col_addr(0);
#pragma unroll
                for (int st = 0; st < AD; ++st) a_read(st);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    const int g = st / HZ, hz = st % HZ;
                    // prefetch: activation fragment AD steps ahead (addresses of the next column when it crosses over)
                    if (st + AD < NSTEP) {
                        if ((st + AD) % HZ == 0) col_addr((st + AD) / HZ);
                        a_read(st + AD);
                    }
                    // weight fragments of the next column, one tz per step of the first three steps of this column
                    if (hz < 3 && g + 1 < 9 && !(TEM_ZR_ABL & 4)) {
                        const int gn = g + 1, tz = hz;
                        const int tap = tz * 9 + gn;
#pragma unroll
                        for (int p = 0; p < NS; ++p)
                            wq[gn & 1][tz][p] = zr_load4u(rw, woff_lane, wsoff + (unsigned)(tap * ts + p * 64) * 16u);
                    }
                    // products of this step: output plane z = hz - tz for tz = 0..2; smallest terms first, interleaved over z
                    const uint4* a = af[st % (AD + 1)];
                    if (WIDE) {   // plane p of both operands = channels 16 p .. 16 p + 15 of the chunk
#pragma unroll
                        for (int p = 0; p < 2; ++p)
#pragma unroll
                            for (int tz = 0; tz < 3; ++tz) {
                                const int z = hz - tz;
                                if (z < 0 || z >= TZ) continue;
                                acc[z] = mfma16<F16>(wq[g & 1][tz][p], a[p], acc[z]);
                            }
                    } else
#pragma unroll
                    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                        for (int i = 0; i <= sum; ++i)
#pragma unroll
                            for (int tz = 0; tz < 3; ++tz) {
                                const int z = hz - tz;
                                if (z < 0 || z >= TZ) continue;
                                // plane pair (weight plane i, activation plane sum - i)
                                if (TEM_ZR_ABL & 16) {
                                    asm volatile("" ::"v"(a[sum - i].x), "v"(a[sum - i].w), "v"(wq[g & 1][tz][i].x), "v"(wq[g & 1][tz][i].w));
                                    continue;
                                }
                                if (SC && sum == 1)
                                    accl[SC ? z : 0] = mfma16<F16>(wq[g & 1][tz][i], a[sum - i], accl[SC ? z : 0]);
                                else
                                    acc[z] = mfma16<F16>(wq[g & 1][tz][i], a[sum - i], acc[z]);
                            }
                    __builtin_amdgcn_sched_barrier(0);
                }
                }   // !X32
                if (TEM_ZR_PRIO) __builtin_amdgcn_s_setprio(0);
            }
            ZR_STAMP(4);
        }
        if (do_stage) {
            if (++ci == nch) {
                ci = 0;
                eu = cu;
                epi_pending = !X32;
                bve = bv;
                // Exact fp32: the unit's epilogue runs HERE, by the team that has just finished its taps, while the partner (blocked
                // during those taps: the vector ALU was the matrix pipe) works through the VALU part of its staging phase -- two
                // latency-bound instruction streams side by side instead of one after the other in front of the next tap phase; and
                // a staging phase without an epilogue has the registers to request its whole halo at once.
                if constexpr (X32) {
                    unsigned yo = yoff_lane, ro = roff_lane;
                    asm volatile("" : "+v"(yo), "+v"(ro));
                    epilogue_now(yo, ro, s);
                }
                bv = 0.f;
                if (++ui < my_units) {
                    cu = decode(ui);
                    if (bias) bv = bias[cu.cot * 32 + v];
                    unit_offsets(cu);
                }
            }
        }
        __syncthreads();
        ZR_STAMP(5);
    }
    if (!team) __syncthreads();
    if (AMAX && out_amax) tem_amax_commit(out_amax, amx);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct ZrGeom {
    int ok;
    int nZ, nY, nX;
    int64_t nunits;
};

// 3x3x3 kernels with two-plane (or the one-term mixed) layouts, 16-byte-vector friendly channel counts and at least one
// unit per team of every CU (smaller launches stay with k_conv_pp / the split-K patch kernel).
static ZrGeom zr_geometry(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit, int64_t max_ld) {
    ZrGeom g = {};
    const long long opt = tem_option(TEM_OPT_CONV_FWD_VARIANT);
    if (opt == 0 || opt == 1) return g;
    if (!(nsplit == 2 || nsplit == 4 || nsplit == 5 || nsplit == 7 || (nsplit == 1 && tem_option(TEM_OPT_FP32_ZR) && !tem_call_st.x))) return g;
    if (!(kd == 3 && kh == 3 && kw == 3)) return g;
    if (D < 4 || Cin % 16 || Cout % 32) return g;
    if (tem_call_st.x && (Cin % 32 || !(nsplit == 5 || nsplit == 7))) return g;   // 16-bit storage: whole 64-byte records per phase
    if ((int64_t)H * W * 8 * 4 * max_ld >= (1ll << 31)) return g;   // 32-bit byte offsets inside one halo / one patch
    static int ncu = 0;
    if (!ncu) {
        ncu = tem_device_cus();
        if (ncu <= 0) ncu = 256;
    }
    g.nZ = (D + 3) / 4;
    g.nY = (H + 15) / 16;
    g.nX = (W + 7) / 8;
    g.nunits = (int64_t)N * g.nZ * g.nY * g.nX * (Cout / 32);
    if (g.nunits >= (1ll << 31)) return g;
    const long long minu = tem_option(TEM_OPT_TEAM_MIN_UNITS);
    if (g.nunits < (opt == 2 ? 1 : (minu > 0 ? minu : 2ll * ncu))) return g;
    g.ok = 1;
    return g;
}

int64_t tem_conv_zr_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit) {
    const ZrGeom g = zr_geometry(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit, 1);
    if (!g.ok) return -1;
    return (int64_t)g.nZ * g.nY * g.nX * 4;
}

// tile-order blocks (see decode() in the kernel): up to 4 x 4 x 4 tiles, each extent a divisor of the tile count
static int zr_tile_blocks(const ZrGeom& g) {
    if (!tem_option(TEM_OPT_ZR_TILE_BLOCKS)) return 0;
    auto lg = [](int n) { return n % 4 == 0 ? 2 : (n % 2 == 0 ? 1 : 0); };
    return lg(g.nX) | (lg(g.nY) << 4) | (lg(g.nZ) << 8);
}

template <int NS, bool F16, int MODE, bool KSPLIT = false, bool WIDE = false, typename T = float, bool X32 = false, bool XS = false>
static void zr_launch(const ZrGeom& g, const float* x_, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                      const float* bias, float* y_, int64_t y_ld, const float* ref_, int64_t ref_ld, int N, int D, int H, int W,
                      int Cin, int Cout, int act, float* stat, const unsigned* in_amax, hipStream_t s, int ks = 1) {
    constexpr size_t ldsb = (size_t)2 * NS * 1080 * 32 + 4 * 32 * 144;   // two tiles + the epilogue's transpose scratch
    static_assert(ldsb <= 160 * 1024, "LDS budget");
    const T* x = reinterpret_cast<const T*>(x_);
    const T* ref = reinterpret_cast<const T*>(ref_);
    auto* y = reinterpret_cast<std::conditional_t<KSPLIT, float, T>*>(y_);
    auto kern = &k_conv_zr<NS, F16, MODE, KSPLIT, WIDE, T, X32, XS>;
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
    int64_t grid = XS ? g.nunits : (g.nunits + 1) / 2;   // (XS: one team per workgroup)
    if (grid > ncu) grid = ncu;
    unsigned* const out_amax = (MODE == 2 || MODE == 3) ? tem_take_output_amax() : nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(XS ? 256 : 512), ldsb, s, x, x_ld, scale, shift, reinterpret_cast<const uint4*>(wp),
                       bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, g.nZ, g.nY, g.nX, stat, (int)g.nunits, in_amax, ks,
                       zr_tile_blocks(g), out_amax, (int64_t)(sizeof(T) == 2 ? tem_call_cs.x : 0), (int64_t)(sizeof(T) == 2 && !KSPLIT ? tem_call_cs.y : 0));
}

// Split-K launch for shapes zr_geometry() declines only because they have too few (tile, column tile) units: the input
// channels are cut into ks slices so that ks x units >= two per CU, the partial sums go to the workspace and the common
// split-K epilogue (conv_mfma.hip) applies bias / activation / ReLU mask.  -> 1 launched, 0 not taken.
#ifndef TEM_ZR_KS_FILL
#define TEM_ZR_KS_FILL 4   // tenths of the tiled volume that must be real voxels for a split-K launch (measured: 8 -> 5: cfg 5 -0.2 ms, cfg 2 -0.04 ms, the 8^3 level is half padding; 5 -> 4: cfg 5 25.37 -> 25.04 ms, its 6 x 12 x 12 level is 42 % real; 3: 25.16)
#endif
// ks of the split-K launch for this shape (0: not taken): only shapes that zr_geometry() / pp_geometry() decline for
// their unit count, tiles that are not mostly padding, at least two 16-channel chunks per slice
int tem_conv_zr_splitk_ks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit) {
    const long long opt = tem_option(TEM_OPT_CONV_FWD_VARIANT);
    if (opt == 0 || opt == 1 || !tem_option(TEM_OPT_ZR_SPLITK)) return 0;
    if (!(nsplit == 2 || nsplit == 4 || nsplit == 5 || nsplit == 7 || (nsplit == 1 && tem_option(TEM_OPT_FP32_ZR) && !tem_call_st.x))) return 0;
    if (!(kd == 3 && kh == 3 && kw == 3) || D < 4 || Cin % 16 || Cout % 32) return 0;
    const bool t16 = tem_call_st.x != 0;   // 16-bit storage: slices of whole 32-channel chunks
    if (t16 && (Cin % 32 || !(nsplit == 5 || nsplit == 7))) return 0;
    if (zr_geometry(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit, 1).ok) return 0;
    if (tem_conv_pp_tiles(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit)) return 0;
    static int ncu = 0;
    if (!ncu) {
        ncu = tem_device_cus();
        if (ncu <= 0) ncu = 256;
    }
    const int nZ = (D + 3) / 4, nY = (H + 15) / 16, nX = (W + 7) / 8;
    // tiles of 4 x 16 x 8 voxels: at most 60 % of the tiled volume may be padding (an 8^3 level has 50 %, 6 x 12 x 12 has 58 %)
    if ((int64_t)D * H * W * 10 < (int64_t)nZ * 4 * nY * 16 * nX * 8 * TEM_ZR_KS_FILL) return 0;
    const int64_t units = (int64_t)N * nZ * nY * nX * (Cout / 32);
    const int nch = Cin / 16;
    for (int d = 2; d <= nch / 2; ++d)
        if (nch % d == 0 && units * d >= 2ll * ncu && !(t16 && (nch / d) % 2)) return d;
    return 0;
}

int tem_conv_fwd_zr_splitk(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp,
                           const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, void* ws,
                           int64_t ws_bytes, int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int act,
                           int nsplit, float* stat, hipStream_t s) {
    // stat: [N][tem_conv_zr_splitk_stat_blocks()][Cout][2] -- the epilogue also writes the statistics partials of y
    const int ks = ws ? tem_conv_zr_splitk_ks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit) : 0;
    if (!ks) return 0;
    if (stat && !tem_splitk_stat_blocks((int64_t)D * H * W, Cout)) return 0;
    if ((int64_t)H * W * 8 * 4 * (x_ld > Cout ? x_ld : Cout) >= (1ll << 31)) return 0;
    if ((y_ld % 4) || ((uintptr_t)y % 16) || (ref && ((ref_ld % 4) || ((uintptr_t)ref % 16))) || (bias && ((uintptr_t)bias % 16)))
        return 0;
    const int64_t NV = (int64_t)N * D * H * W;
    if (ws_bytes < (int64_t)ks * NV * Cout * 4) return 0;
    ZrGeom g = {};
    g.nZ = (D + 3) / 4;
    g.nY = (H + 15) / 16;
    g.nX = (W + 7) / 8;
    g.nunits = (int64_t)N * g.nZ * g.nY * g.nX * (Cout / 32) * ks;
    g.ok = 1;
    float* part = (float*)ws;
#define ZRKS(NS, F16, WIDE)                                                                                            \
    zr_launch<NS, F16, 0, true, WIDE>(g, x, x_ld, scale, shift, wp, nullptr, part, Cout, nullptr, 0, N, D, H, W, Cin, Cout,  \
                                      TEM_ACT_NONE, nullptr, nullptr, s, ks)
    const bool wide = (nsplit == 5 || nsplit == 7) && (Cin / 16 / ks) % 2 == 0 && (tem_option(TEM_OPT_ZR_WIDE) || tem_call_st.x);   // slices of whole 32-channel chunks
    if (tem_call_st.x == 1)
        zr_launch<2, true, 0, true, true, tem_f16>(g, x, x_ld, scale, shift, wp, nullptr, part, Cout, nullptr, 0, N, D, H, W, Cin, Cout,
                                                   TEM_ACT_NONE, nullptr, nullptr, s, ks);
    else if (tem_call_st.x == 2)
        zr_launch<2, false, 0, true, true, tem_bf16>(g, x, x_ld, scale, shift, wp, nullptr, part, Cout, nullptr, 0, N, D, H, W, Cin, Cout,
                                                     TEM_ACT_NONE, nullptr, nullptr, s, ks);
    else if (nsplit == 1 && tem_option(TEM_OPT_FP32_ZR) == 2)
        zr_launch<2, false, 0, true, false, float, true, true>(g, x, x_ld, scale, shift, wp, nullptr, part, Cout, nullptr, 0, N, D, H, W, Cin, Cout,
                                                               TEM_ACT_NONE, nullptr, nullptr, s, ks);
    else if (nsplit == 1)
        zr_launch<2, false, 0, true, false, float, true>(g, x, x_ld, scale, shift, wp, nullptr, part, Cout, nullptr, 0, N, D, H, W, Cin, Cout,
                                                         TEM_ACT_NONE, nullptr, nullptr, s, ks);
    else if (nsplit == 5 && wide) ZRKS(2, true, true);
    else if (nsplit == 7 && wide) ZRKS(2, false, true);
    else if (nsplit == 5) ZRKS(1, true, false);
    else if (nsplit == 7) ZRKS(1, false, false);
    else if (nsplit == 4) ZRKS(2, true, false);
    else ZRKS(2, false, false);
#undef ZRKS
    TemDgradSumsReq rq = {nullptr, 0, nullptr, nullptr, 0, nullptr, 0};
    if (!stat && tem_bp_wants(TEM_BP_NORM_SUMS)) {
        const TemByproducts* bp = tem_call_bp;
        rq = TemDgradSumsReq{bp->sums_x, bp->sums_x_ld, bp->sums_mean, bp->sums_rstd, bp->sums_G, bp->sums_part, bp->sums_nblk};
    }
    if (stat)
        tem_splitk_epilogue_stats(part, ks, N, (int64_t)D * H * W, Cout, bias, act, ref, ref_ld, y, y_ld, stat, s);
    else if (rq.part && rq.x && rq.mean && rq.rstd && rq.nblk == tem_splitk_stat_blocks((int64_t)D * H * W, Cout) && rq.G > 0 &&
             Cout % rq.G == 0 && rq.x_ld % 4 == 0 && ((uintptr_t)rq.x % 16 == 0)) {
        tem_bp_delivered(TEM_BP_NORM_SUMS);
        tem_splitk_epilogue_bwd_sums(part, ks, N, (int64_t)D * H * W, Cout, bias, act, ref, ref_ld, y, y_ld, rq, s);
    } else
        tem_splitk_epilogue(part, ks, NV, Cout, bias, act, ref, ref_ld, y, y_ld, s);
    return 1;
}

// statistics partial rows per sample when tem_conv_fwd_zr_splitk takes the launch with stat != NULL (-1: it does not)
int64_t tem_conv_zr_splitk_stat_blocks(int N, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int nsplit) {
    if (!tem_conv_zr_splitk_ks(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit)) return -1;
    const int64_t nb = tem_splitk_stat_blocks((int64_t)D * H * W, Cout);
    return nb > 0 ? nb : -1;
}

// tem_conv3d_fwd_gscaled (conv.hip) parks the device pointer of max |input| here around its call; the launch that honours
// it clears it (so the caller can tell that the prescale really happened)
thread_local const unsigned* tem_zr_in_amax = nullptr;
// tem_conv3d_fwd_refnorm parks coef[N][Cout][4] here the same way: the launch applies the ReLU mask of `ref` AND the backward
// of the norm behind it in its epilogue (MODE 3)
thread_local const float* tem_zr_ref_coef = nullptr;

// -> 1 launched, 0 shape not taken, -1 error (statistics sized for this kernel but the launch cannot take it)
int tem_conv_fwd_zr(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* wp, const float* bias,
                    float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H, int W, int Cin, int Cout,
                    int kd, int kh, int kw, int act, int nsplit, float* stat, hipStream_t s) {
    int64_t max_ld = x_ld > y_ld ? x_ld : y_ld;
    if (ref && ref_ld > max_ld) max_ld = ref_ld;
    const ZrGeom g = zr_geometry(N, D, H, W, Cin, Cout, kd, kh, kw, nsplit, max_ld);
    const bool strided = tem_call_cs.x != 0 || tem_call_cs.y != 0;
    if (strided && (!g.ok || !tem_call_st.x || ref || (tem_call_cs.x && Cin % 32) || (tem_call_cs.x % 8) || (tem_call_cs.y % 8))) {
        tem_set_error("tem_conv3d_fwd_ex: chunk strides (x_cs / y_cs) need 16-bit tensors on the z-reuse kernel (tem_conv3d_fwd_kernel() "
                      "== 3), no ref, strides %% 8 == 0");
        return -1;
    }
    if (!g.ok) return 0;
    if ((y_ld % 4) || ((uintptr_t)y % 16) || (ref && ((ref_ld % 4) || ((uintptr_t)ref % 16))) || (stat && ref) ||
        (bias && ((uintptr_t)bias % 16)) || act == TEM_ACT_SIGMOID) {
        if (stat || strided) {
            tem_set_error("tem_conv3d_fwd_stats / _ex: statistics (or chunk strides) were sized for the z-reuse kernel but this launch "
                          "cannot take it (y / ref / bias need 16-byte alignment and ld %% 4 == 0, no ref, no sigmoid)");
            return -1;
        }
        return 0;
    }
#define ZRGO(NS, F16, WIDE)                                                                                                   \
    do {                                                                                                                      \
        if (stat)                                                                                                             \
            zr_launch<NS, F16, 1, false, WIDE>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else if (ref && rcoef)                                                                                                \
            zr_launch<NS, F16, 3, false, WIDE>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act,        \
                                  const_cast<float*>(rcoef), in_amax, s);                                                     \
        else if (ref)                                                                                                         \
            zr_launch<NS, F16, 2, false, WIDE>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else                                                                                                                  \
            zr_launch<NS, F16, 0, false, WIDE>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
    } while (0)
    const float* rcoef = tem_zr_ref_coef;
    if (rcoef) {
        if (!ref || stat || ((uintptr_t)rcoef % 16)) {
            tem_set_error("tem_conv3d_fwd_refnorm: needs ref, no statistics, 16-byte aligned coefficients");
            return -1;
        }
        tem_zr_ref_coef = nullptr;   // consumed
    }
    const unsigned* in_amax = tem_zr_in_amax;
    if (in_amax) {
        if (nsplit != 4 || bias || scale || stat) {
            tem_set_error("tem_conv3d_fwd_gscaled: fp16 two-term layout, no bias / norm / statistics");
            return -1;
        }
        tem_zr_in_amax = nullptr;   // consumed
    }
    // one-term modes: 32 channels per phase whenever the channel count allows it (whole 128-byte lines per staging phase)
    const bool wide = (nsplit == 5 || nsplit == 7) && Cin % 32 == 0 && tem_option(TEM_OPT_ZR_WIDE);
#define ZRGO16(F16, T)                                                                                                        \
    do {                                                                                                                      \
        if (stat)                                                                                                             \
            zr_launch<2, F16, 1, false, true, T>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else if (ref && rcoef)                                                                                                \
            zr_launch<2, F16, 3, false, true, T>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act,        \
                                  const_cast<float*>(rcoef), in_amax, s);                                                     \
        else if (ref)                                                                                                         \
            zr_launch<2, F16, 2, false, true, T>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else                                                                                                                  \
            zr_launch<2, F16, 0, false, true, T>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
    } while (0)
#define ZRGO32S()                                                                                                               \
    do {                                                                                                                      \
        if (stat)                                                                                                             \
            zr_launch<2, false, 1, false, false, float, true, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else if (ref && rcoef)                                                                                                \
            zr_launch<2, false, 3, false, false, float, true, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act,        \
                                  const_cast<float*>(rcoef), in_amax, s);                                                     \
        else if (ref)                                                                                                         \
            zr_launch<2, false, 2, false, false, float, true, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else                                                                                                                  \
            zr_launch<2, false, 0, false, false, float, true, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
    } while (0)
#define ZRGO32()                                                                                                                \
    do {                                                                                                                      \
        if (stat)                                                                                                             \
            zr_launch<2, false, 1, false, false, float, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else if (ref && rcoef)                                                                                                \
            zr_launch<2, false, 3, false, false, float, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act,        \
                                  const_cast<float*>(rcoef), in_amax, s);                                                     \
        else if (ref)                                                                                                         \
            zr_launch<2, false, 2, false, false, float, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
        else                                                                                                                  \
            zr_launch<2, false, 0, false, false, float, true>(g, x, x_ld, scale, shift, wp, bias, y, y_ld, ref, ref_ld, N, D, H, W, Cin, Cout, act, stat, in_amax, s); \
    } while (0)
    if (tem_call_st.x == 1) ZRGO16(true, tem_f16);
    else if (tem_call_st.x == 2) ZRGO16(false, tem_bf16);
    else
#undef ZRGO16
    if (nsplit == 1 && tem_option(TEM_OPT_FP32_ZR) == 2) ZRGO32S();
    else if (nsplit == 1) ZRGO32();
    else if (nsplit == 5 && wide) ZRGO(2, true, true);
    else if (nsplit == 7 && wide) ZRGO(2, false, true);
    else if (nsplit == 5) ZRGO(1, true, false);
    else if (nsplit == 7) ZRGO(1, false, false);
    else if (nsplit == 4) ZRGO(2, true, false);
    else ZRGO(2, false, false);
#undef ZRGO
#undef ZRGO32
#undef ZRGO32S
    return 1;
}
