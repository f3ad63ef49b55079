// capi.hip -- library-level entry points of libtem_hip.so and the layout utilities.
#include "tem_common.h"
#include "conv_internal.h"
#include "tem_act.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

// storage types of the call in flight on this thread (tem_act.h): set and restored by the entry point itself
thread_local TemCallSt tem_call_st = {0, 0};
thread_local TemCallCs tem_call_cs = {0, 0};

void tem_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tem_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// dispatch options: explicit, process-wide switches between kernel variants (the library never reads the
// environment).  Plain relaxed atomics: an option is read once per launch on the calling thread.
// ---------------------------------------------------------------------------
struct TemOption {
    const char* name;
    long long def;
};
static const TemOption g_opt_table[TEM_OPT_COUNT] = {
    {"wgrad_zs", 3},            // TEM_OPT_WGRAD_ZS: z-sliding weight gradient for 3x3x3, D >= 16 (3 k_conv_wgrad_tr: staging team + transposing LDS reads, 2 k_conv_wgrad_zt, 1 k_conv_wgrad_zs, 0 patch kernel)
    {"wgrad_zs_persist", 1},    // TEM_OPT_WGRAD_ZS_PERSIST: persistent column segments (one slab per workgroup)
    {"wgrad_sums", 1},          // TEM_OPT_WGRAD_SUMS: norm-backward sums from the weight gradient
    {"wgrad_sums_min_mb", 128}, // TEM_OPT_WGRAD_SUMS_MIN_MB: ... for layers whose norm input has at least this many MiB (measured: 256 -> 128 -0.07 ms, 64 +0.09 ms)
    {"fwd_persistent", -1},     // TEM_OPT_FWD_PERSISTENT: exact-fp32 forward, persistent variant (-1 = 64-column tiles only)
    {"conv_fwd_variant", -1},   // TEM_OPT_CONV_FWD_VARIANT: split-precision forward/dgrad kernel (-1 auto, 0 patch kernel, 1 ping-pong forced, 2 z-reuse forced)
    {"conv1x1_stream", 1},      // TEM_OPT_CONV1X1_STREAM: 1x1x1 convolutions / data gradients as a streaming GEMM (conv1x1_stream.hip)
    {"fwd_ksplit_chunks", 0},   // TEM_OPT_FWD_KSPLIT_CHUNKS: split-K forward launches: at most this many 16-channel chunks per partial (0: heuristic only)
    {"wgrad_cus", 256},         // TEM_OPT_WGRAD_CUS: workgroups the z-sliding weight gradient asks for (one per CU)
    {"upsample_generic", 0},    // TEM_OPT_UPSAMPLE_GENERIC: 1 = any-factor gather kernels also for factor (1|2, 2, 2) (A/B, tests)
    {"team_min_units", 0},      // TEM_OPT_TEAM_MIN_UNITS: units a launch needs for the team kernels (0 = 2 per CU; experiments)
    {"zr_splitk", 1},           // TEM_OPT_ZR_SPLITK: z-reuse kernel with split input channels for the 16^3 / 32^3 levels
    {"zr_wide", 1},             // TEM_OPT_ZR_WIDE: one-term z-reuse kernel stages 32 channels (whole 128-byte lines) per phase (0: 16, A/B)
    {"zr_tile_blocks", 1},      // TEM_OPT_ZR_TILE_BLOCKS: z-reuse kernel walks its tiles in 4 x 4 x 4 blocks (one block per XCD at a time; 0: x, y, z order, A/B)
    {"dice_vox", 1},            // TEM_OPT_DICE_VOX: Dice sums / gradient with one voxel per thread for C <= 16 (0: one (channel, voxel) per thread, A/B)
    {"upsample2_ch8", 1},       // TEM_OPT_UPSAMPLE2_CH8: factor-2 upsampling BACKWARD of 16-bit tensors with 8 channels per thread (0: 4, A/B)
    {"pool_vec8", 2},           // TEM_OPT_POOL_VEC8: max-pool kernels on 16-bit tensors with 8 channels (16 bytes) per thread; 2: backward on the packed-word kernel k_maxpool_bwd16 (1: generic kernel, 0: 4 channels / 8 bytes -- A/B)
    {"fp32_zr", 1},             // TEM_OPT_FP32_ZR: exact-fp32 3x3x3 forward / data gradient on the z-reuse team kernel (k_conv_zr<..., X32>; 2: one team per workgroup <..., X32, XS>; 0: k_conv_fwd_mfma[_p], A/B)
};
static long long g_opt_val[TEM_OPT_COUNT];
static bool g_opt_set[TEM_OPT_COUNT];

long long tem_option(int id) {
    if (id < 0 || id >= TEM_OPT_COUNT) return 0;
    return __atomic_load_n(&g_opt_set[id], __ATOMIC_RELAXED) ? __atomic_load_n(&g_opt_val[id], __ATOMIC_RELAXED)
                                                            : g_opt_table[id].def;
}
static int opt_find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < TEM_OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_table[i].name)) return i;
    return -1;
}
extern "C" int tem_set_option(const char* name, int64_t value) {
    const int id = opt_find(name);
    TEM_REQUIRE(id >= 0, "tem_set_option: unknown option '%s'", name ? name : "(null)");
    __atomic_store_n(&g_opt_val[id], (long long)value, __ATOMIC_RELAXED);
    __atomic_store_n(&g_opt_set[id], true, __ATOMIC_RELAXED);
    return TEM_OK;
}
extern "C" int tem_get_option(const char* name, int64_t* value) {
    const int id = opt_find(name);
    TEM_REQUIRE(id >= 0 && value, "tem_get_option: unknown option '%s'", name ? name : "(null)");
    *value = (int64_t)tem_option(id);
    return TEM_OK;
}
extern "C" int tem_version(void) { return 100; }

// ---- by-products of the call in flight (include/tem_hip.h: TemByproducts) ----------------------------------------------------
// The *_ex entry points install the struct their caller passed for the duration of the call (TemBpScope restores the previous
// pointer on exit: nothing survives a call, and a call without by-products sees NULL); a launch site that can deliver one
// takes it here, which records it in `delivered`.  Same standing as tem_call_st: plumbing between an entry point and the
// launchers it reaches, not state of the C-ABI.
thread_local TemByproducts* tem_call_bp = nullptr;
unsigned* tem_take_output_amax() {
    TemByproducts* bp = tem_call_bp;
    if (!bp || !bp->out_amax || (bp->delivered & TEM_BP_OUT_AMAX)) return nullptr;
    bp->delivered |= TEM_BP_OUT_AMAX;
    return bp->out_amax;
}
bool tem_bp_wants(unsigned bit) {
    const TemByproducts* bp = tem_call_bp;
    if (!bp || (bp->delivered & bit)) return false;
    return bit == TEM_BP_NORM_COEF ? bp->coef != nullptr : bit == TEM_BP_NORM_SUMS ? bp->sums_part != nullptr : bp->out_amax != nullptr;
}
void tem_bp_delivered(unsigned bit) {
    if (tem_call_bp) tem_call_bp->delivered |= bit;
}

extern "C" int tem_device_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TEM_ELAUNCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return TEM_ELAUNCH;
    return prop.multiProcessorCount;
}

// ---------------------------------------------------------------------------
// NCDHW <-> NDHWC.  Tiled through LDS so that both sides are coalesced:
// a tile is 64 voxels x 32 channels.
// ---------------------------------------------------------------------------
#define TT_V 64
#define TT_C 32

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst,
                                                      int64_t dst_ld, int C, int64_t V, int64_t vtiles, int ctiles) {
    __shared__ float tile[TT_C][TT_V + 1];
    int64_t ntiles = vtiles * ctiles;
    int n = blockIdx.y;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int ct = (int)(t % ctiles);
        int64_t vt = t / ctiles;
        int64_t v0 = vt * TT_V;
        int c0 = ct * TT_C;
        // read: voxel fastest
        for (int i = threadIdx.x; i < TT_C * TT_V; i += 256) {
            int c = i / TT_V, v = i % TT_V;
            float val = 0.f;
            if (c0 + c < C && v0 + v < V) val = src[((int64_t)n * C + c0 + c) * V + v0 + v];
            tile[c][v] = val;
        }
        __syncthreads();
        // write: channel fastest
        for (int i = threadIdx.x; i < TT_C * TT_V; i += 256) {
            int v = i / TT_C, c = i % TT_C;
            if (c0 + c < C && v0 + v < V) dst[((int64_t)n * V + v0 + v) * dst_ld + c0 + c] = tile[c][v];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ src, int64_t src_ld,
                                                      float* __restrict__ dst, int C, int64_t V, int64_t vtiles,
                                                      int ctiles) {
    __shared__ float tile[TT_C][TT_V + 1];
    int64_t ntiles = vtiles * ctiles;
    int n = blockIdx.y;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int ct = (int)(t % ctiles);
        int64_t vt = t / ctiles;
        int64_t v0 = vt * TT_V;
        int c0 = ct * TT_C;
        for (int i = threadIdx.x; i < TT_C * TT_V; i += 256) {
            int v = i / TT_C, c = i % TT_C;
            float val = 0.f;
            if (c0 + c < C && v0 + v < V) val = src[((int64_t)n * V + v0 + v) * src_ld + c0 + c];
            tile[c][v] = val;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TT_C * TT_V; i += 256) {
            int c = i / TT_V, v = i % TT_V;
            if (c0 + c < C && v0 + v < V) dst[((int64_t)n * C + c0 + c) * V + v0 + v] = tile[c][v];
        }
        __syncthreads();
    }
}

extern "C" int tem_nchw_to_nhwc(const float* src, float* dst, int64_t dst_ld, int N, int C, int64_t V,
                                tem_stream_t stream) {
    TEM_REQUIRE(src && dst && N > 0 && C > 0 && V > 0 && dst_ld >= C, "tem_nchw_to_nhwc: bad arguments");
    int64_t vtiles = tem_cdiv(V, TT_V);
    int ctiles = (int)tem_cdiv(C, TT_C);
    int64_t nt = vtiles * ctiles;
    dim3 grid((unsigned)(nt > 4096 ? 4096 : nt), N);
    hipLaunchKernelGGL(k_nchw_to_nhwc, grid, dim3(256), 0, (hipStream_t)stream, src, dst, dst_ld, C, V, vtiles, ctiles);
    TEM_CHECK_LAUNCH("tem_nchw_to_nhwc");
    return TEM_OK;
}

extern "C" int tem_nhwc_to_nchw(const float* src, int64_t src_ld, float* dst, int N, int C, int64_t V,
                                tem_stream_t stream) {
    TEM_REQUIRE(src && dst && N > 0 && C > 0 && V > 0 && src_ld >= C, "tem_nhwc_to_nchw: bad arguments");
    int64_t vtiles = tem_cdiv(V, TT_V);
    int ctiles = (int)tem_cdiv(C, TT_C);
    int64_t nt = vtiles * ctiles;
    dim3 grid((unsigned)(nt > 4096 ? 4096 : nt), N);
    hipLaunchKernelGGL(k_nhwc_to_nchw, grid, dim3(256), 0, (hipStream_t)stream, src, src_ld, dst, C, V, vtiles, ctiles);
    TEM_CHECK_LAUNCH("tem_nhwc_to_nchw");
    return TEM_OK;
}

// ---------------------------------------------------------------------------
// standardize: y = (x - mean) / (std + eps) per row (transform/raw.py:40-65;
// numpy mean/std => population std).  Two-stage deterministic reduction.
// ---------------------------------------------------------------------------
#define STD_BLOCKS 256

__global__ __launch_bounds__(256) void k_std_partial(const float* __restrict__ x, int64_t L, double* __restrict__ part) {
    int n = blockIdx.y;
    const float* row = x + (int64_t)n * L;
    double s = 0.0, ss = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
        double v = (double)row[i];
        s += v;
        ss += v * v;
    }
    s = tem_wave_sum_d(s);
    ss = tem_wave_sum_d(ss);
    __shared__ double sh[2][4];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((int64_t)n * gridDim.x + blockIdx.x) * 2 + 0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        part[((int64_t)n * gridDim.x + blockIdx.x) * 2 + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

__global__ __launch_bounds__(256) void k_std_apply(const float* __restrict__ x, float* __restrict__ y, int64_t L,
                                                   const double* __restrict__ part, int nblk, float eps) {
    int n = blockIdx.y;
    __shared__ float sh_mean, sh_inv;
    if (threadIdx.x == 0) {
        double s = 0.0, ss = 0.0;
        for (int b = 0; b < nblk; ++b) {
            s += part[((int64_t)n * nblk + b) * 2 + 0];
            ss += part[((int64_t)n * nblk + b) * 2 + 1];
        }
        double mean = s / (double)L;
        double var = ss / (double)L - mean * mean;
        if (var < 0.0) var = 0.0;
        sh_mean = (float)mean;
        sh_inv = (float)(1.0 / (sqrt(var) + (double)eps));
    }
    __syncthreads();
    float mean = sh_mean, inv = sh_inv;
    const float* row = x + (int64_t)n * L;
    float* out = y + (int64_t)n * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256)
        out[i] = (row[i] - mean) * inv;
}

extern "C" int tem_standardize(const float* x, float* y, int N, int64_t L, float eps, void* ws, int64_t ws_bytes,
                               tem_stream_t stream) {
    TEM_REQUIRE(x && y && ws && N > 0 && L > 0, "tem_standardize: bad arguments");
    int nblk = (int)(tem_cdiv(L, 256) < STD_BLOCKS ? tem_cdiv(L, 256) : STD_BLOCKS);
    if ((int64_t)N * nblk * 2 * (int64_t)sizeof(double) > ws_bytes) {
        tem_set_error("tem_standardize: workspace too small (%lld bytes needed)",
                      (long long)((int64_t)N * nblk * 2 * sizeof(double)));
        return TEM_EWS;
    }
    hipLaunchKernelGGL(k_std_partial, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, x, L, (double*)ws);
    hipLaunchKernelGGL(k_std_apply, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, x, y, L, (const double*)ws, nblk, eps);
    TEM_CHECK_LAUNCH("tem_standardize");
    return TEM_OK;
}

// ---------------------------------------------------------------------------
// final-activation backward (contiguous arrays)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ gy, const float* __restrict__ y,
                                                 float* __restrict__ gx, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float yv = y[i], g = gy[i];
        gx[i] = (act == TEM_ACT_SIGMOID) ? g * yv * (1.f - yv) : ((yv > 0.f) ? g : 0.f);
    }
}

extern "C" int tem_act_bwd(const float* gy, const float* y, float* gx, int64_t n, int act, tem_stream_t stream) {
    TEM_REQUIRE(gy && y && gx && n > 0, "tem_act_bwd: bad arguments");
    TEM_REQUIRE(act == TEM_ACT_RELU || act == TEM_ACT_SIGMOID, "tem_act_bwd: Invalid activation: %d", act);
    hipLaunchKernelGGL(k_act_bwd, dim3(tem_grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, gy, y, gx, n, act);
    TEM_CHECK_LAUNCH("tem_act_bwd");
    return TEM_OK;
}
