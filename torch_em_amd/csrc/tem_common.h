// tem_common.h -- shared helpers for the gfx950 kernels of libtem_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tem_hip.h"

void tem_set_error(const char* fmt, ...);

// dispatch options (capi.hip; set through tem_set_option(), never through the environment)
enum {
    TEM_OPT_WGRAD_ZS = 0,
    TEM_OPT_WGRAD_ZS_PERSIST,
    TEM_OPT_WGRAD_SUMS,
    TEM_OPT_WGRAD_SUMS_MIN_MB,
    TEM_OPT_FWD_PERSISTENT,
    TEM_OPT_CONV_FWD_VARIANT,
    TEM_OPT_CONV1X1_STREAM,
    TEM_OPT_FWD_KSPLIT_CHUNKS,
    TEM_OPT_WGRAD_CUS,
    TEM_OPT_UPSAMPLE_GENERIC,
    TEM_OPT_TEAM_MIN_UNITS,
    TEM_OPT_ZR_SPLITK,
    TEM_OPT_ZR_WIDE,
    TEM_OPT_ZR_TILE_BLOCKS,
    TEM_OPT_DICE_VOX,
    TEM_OPT_UPSAMPLE2_CH8,
    TEM_OPT_POOL_VEC8,
    TEM_OPT_FP32_ZR,
    TEM_OPT_COUNT
};
long long tem_option(int id);

#define TEM_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            tem_set_error(__VA_ARGS__);        \
            return TEM_EINVAL;                 \
        }                                      \
    } while (0)

#define TEM_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            tem_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return TEM_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

static inline int64_t tem_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t tem_align_up(int64_t a, int64_t b) { return tem_cdiv(a, b) * b; }

// Memory-bound kernels: cap the grid at ~8 blocks per CU and grid-stride the rest.
static inline int tem_grid_1d(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = tem_cdiv(work_items, block);
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    return (int)g;
}

#define TEM_WAVE 64

// By-products of the call in flight (tem_hip.h: TemByproducts, an explicit argument of the *_ex entry points; capi.hip).
extern thread_local TemByproducts* tem_call_bp;
struct TemBpScope {
    TemByproducts* prev;
    explicit TemBpScope(TemByproducts* bp) : prev(tem_call_bp) {
        tem_call_bp = bp;
        if (bp) bp->delivered = 0;
    }
    ~TemBpScope() { tem_call_bp = prev; }
};
bool tem_bp_wants(unsigned bit);        // the call in flight asks for by-product `bit` and nothing has delivered it yet
void tem_bp_delivered(unsigned bit);
// "output amax": max |y| of the tensor a launch writes, as a by-product (TEM_BP_OUT_AMAX).  A launch site that supports it
// takes the caller's device word with tem_take_output_amax() (NULL: not asked for); its kernel keeps a per-thread maximum
// (tem_amax4) and ends with tem_amax_commit() -- wave reduction, then an integer atomicMax of the bit pattern (exact,
// order-independent) that most waves skip after one plain read of the word.
unsigned* tem_take_output_amax();
__device__ __forceinline__ float tem_amax4(float m, float a, float b, float c, float d) {
    return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b))),
                           __builtin_fmaxf(__builtin_fabsf(c), __builtin_fabsf(d)));
}
__device__ __forceinline__ void tem_amax_commit(unsigned* amax, float m) {   // all lanes of a wave; amax wave-uniform, non-null
    unsigned u = __builtin_bit_cast(unsigned, m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, o, 64));
    if ((threadIdx.x & 63) == 0 && u > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, u);
}

__device__ __forceinline__ float tem_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double tem_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids
// land on the same XCD (dispatch places block b on XCD b % 8), so neighbouring
// tiles share that XCD's L2.  Speed only; any mapping is correct.
__device__ __forceinline__ int tem_xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
