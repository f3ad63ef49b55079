// wgrad_sums.hip -- the first stage of the NORM BACKWARD without a pass over the tensors.
//
// The norm in front of a conv (z = gamma * xn + beta, fused into the conv as scale/shift) needs, per (sample, channel),
//   A = sum_v gz[v][c]            B = sum_v gz[v][c] * xn[v][c]
// where gz is the data gradient of the conv (dgrad output).  Both follow from quantities the WEIGHT gradient already
// has, because gz[v][c] = sum_{tap,co} w[co][c][tap] * g[v - tap][co]  (g = gradient w.r.t. the conv output, zero
// outside the volume):
//   sum_v gz[v][c] z[v][c] = sum_{tap,co} w[co][c][tap] * dw_n[co][c][tap]     dw_n = weight gradient of sample n
//   sum_v gz[v][c]         = sum_{tap,co} w[co][c][tap] * T_n[tap][co]         T_n[tap][co] = sum of g[.][co] over the
//                                                                              voxels u with u + tap inside the volume
// and B = (sum gz z - beta * A) / gamma.  T_n is the per-sample bias gradient minus boundary faces / edges / corners.
// The tensor pass k_norm_partial<.,1> (two reads of a 0.5-1 GB tensor per level-0 layer, 1.06 ms of a 24 ms step)
// becomes: per-sample sums in the slab merge that runs anyway, a pass over the boundary shell of g (1.5 % of it), and
// one block per sample of arithmetic on [27][Cout] numbers.
#include "tem_common.h"
#include "conv_internal.h"
#include "tem_act.h"

// ---------------------------------------------------------------------------
// boundary shell of g: sums over the 9 (y class, x class) position classes (class 0 = first index, 1 = middle, 2 = last)
// per "slot": slot s < D-2 is the RING of middle plane z = s+1 (rows 0 / H-1, columns 0 / W-1; its (1,1) entry stays
// 0: the interior total comes from the bias gradient); slots D-2 .. D-2+H-1 are the rows of plane 0, the next H those
// of plane D-1 (full rows).  planepart: [N][nslot][9][C], nslot = D-2 + 2H.
// ---------------------------------------------------------------------------
// body for one HALF (256 threads, `tid` = thread index inside it) of a 512-thread block: slot `slot` of sample n, or nothing
// (act == false: the half only meets the barrier); lsh: this half's [4 waves][9][C] floats
template <typename T>
__device__ __forceinline__ void shell_plane_sums_body(const T* __restrict__ g, int64_t g_ld, int D, int H, int W, int C,
                                                      float* __restrict__ planepart, int slot, int n, float* lsh, int tid,
                                                      bool act) {
    const int nslot = D - 2 + 2 * H;
    const int cq = C >> 2;          // power of two <= 64 (checked by the launcher)
    const int q = tid % cq, lanev = tid / cq, nlv = 256 / cq;
    const bool ringmode = slot < D - 2;
    const int z = ringmode ? slot + 1 : ((slot - (D - 2)) < H ? 0 : D - 1);
    const int yrow = ringmode ? 0 : (slot - (D - 2)) % H;
    const int count = ringmode ? 2 * W + 2 * (H - 2) : W;
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const T* gp = g + ((int64_t)n * D + z) * H * W * g_ld + q * 4;
    // four voxels per trip, loads unconditional (a clamped index, the value masked out below): the one-load-per-trip loop
    // was 16 dependent round trips for a ring slot
    constexpr int SU = 4;
    for (int idx0 = lanev; act && idx0 < count; idx0 += SU * nlv) {
        float4 v[SU];
        int cls[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int idx = idx0 + u * nlv < count ? idx0 + u * nlv : count - 1;
            int y, x;
            if (!ringmode) {
                y = yrow;
                x = idx;
            } else if (idx < W) {
                y = 0;
                x = idx;
            } else if (idx < 2 * W) {
                y = H - 1;
                x = idx - W;
            } else {
                const int j = idx - 2 * W;
                y = 1 + (j >> 1);
                x = (j & 1) ? W - 1 : 0;
            }
            cls[u] = idx0 + u * nlv < count ? ((y == 0) ? 0 : (y == H - 1) ? 2 : 1) * 3 + ((x == 0) ? 0 : (x == W - 1) ? 2 : 1) : -1;
            v[u] = act_ld4(gp + ((int64_t)y * W + x) * g_ld);
        }
#pragma unroll
        for (int u = 0; u < SU; ++u)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float m = (k == cls[u]) ? 1.f : 0.f;
                acc[k].x = fmaf(m, v[u].x, acc[k].x);
                acc[k].y = fmaf(m, v[u].y, acc[k].y);
                acc[k].z = fmaf(m, v[u].z, acc[k].z);
                acc[k].w = fmaf(m, v[u].w, acc[k].w);
            }
    }
    const int wv = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float vals[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int o = cq; o < 64; o <<= 1) vals[j] += __shfl_xor(vals[j], o, 64);  // lanes sharing q sit cq apart
            if (act && lane < cq) lsh[(wv * 9 + k) * C + lane * 4 + j] = vals[j];
        }
    }
    __syncthreads();
    for (int t = tid; act && t < 9 * C; t += 256) {
        float a = 0.f;
        for (int w4 = 0; w4 < 4; ++w4) a += lsh[w4 * 9 * C + t];
        planepart[(((int64_t)n * nslot + slot) * 9) * C + t] = a;
    }
}

struct ShellArgs {
    const void* g;   // element type: st (TEM_ST_*)
    int st;
    int64_t g_ld;
    int D, H, W;
    float* planepart;
};

// ---------------------------------------------------------------------------
// slab merge (as k_reduce_slabs_sd) that also emits, per sample and per group of 32 output channels,
//   P[n][i / 32] = sum_{co in group} w[co][ci][tap] * dw_n[tap][ci][co]
// chunks are grouped per sample: chunk c belongs to sample c / cps
// ---------------------------------------------------------------------------
#define WS_MAXN 4
__global__ __launch_bounds__(512) void k_reduce_slabs_wsum(const float* __restrict__ part, int cps, int N, int ntaps,
                                                           int Cin, int Cout, int64_t chunk_stride,
                                                           float* __restrict__ out, const float* __restrict__ w,
                                                           float* __restrict__ P, int nb_w,
                                                           const float* __restrict__ dbpart, int db_chunks,
                                                           float* __restrict__ db, int nb_db, ShellArgs sa) {
    __shared__ double sh[WS_MAXN][8][64];
    const int64_t n_out = (int64_t)ntaps * Cin * Cout;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    // block roles: the shell blocks FIRST (the longest: they must not queue behind the merge blocks), then merge, then bias
    const int nb_sh = (int)gridDim.x - nb_w - nb_db, bid = (int)blockIdx.x - nb_sh;
    if (bid < 0) {
        // the boundary-shell sums of g need nothing from the weight gradient: their blocks run NEXT TO the slab merge
        // instead of in a launch of their own behind it (20 us per layer)
        // (two slots per block, one per half: all blocks of the launch are resident at once -- with one slot per block
        // the shell blocks queued behind the merge blocks and the launch took the SUM of the two, 30 us)
        const int nslot = sa.D - 2 + 2 * sa.H, half = threadIdx.x >> 8;
        const int sb = (int)blockIdx.x * 2 + half;
        const bool act = sb < nslot * N;
        extern __shared__ float lsh[];   // [2 halves][4 waves][9][Cout]
        if (sa.st == 0)   // launch-uniform
            shell_plane_sums_body((const float*)sa.g, sa.g_ld, sa.D, sa.H, sa.W, Cout, sa.planepart, act ? sb % nslot : 0,
                                  act ? sb / nslot : 0, lsh + half * 4 * 9 * Cout, threadIdx.x & 255, act);
        else if (sa.st == 1)
            shell_plane_sums_body((const tem_f16*)sa.g, sa.g_ld, sa.D, sa.H, sa.W, Cout, sa.planepart, act ? sb % nslot : 0,
                                  act ? sb / nslot : 0, lsh + half * 4 * 9 * Cout, threadIdx.x & 255, act);
        else
            shell_plane_sums_body((const tem_bf16*)sa.g, sa.g_ld, sa.D, sa.H, sa.W, Cout, sa.planepart, act ? sb % nslot : 0,
                                  act ? sb / nslot : 0, lsh + half * 4 * 9 * Cout, threadIdx.x & 255, act);
        return;
    }
    if (bid >= nb_w) {  // the bias gradient rides along (arithmetic of k_reduce_slabs): one launch less per layer
        const int co = (bid - nb_w) * 64 + tx;
        double s = 0.0;
        if (co < Cout) {
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int c = ty;
            for (; c + 24 < db_chunks; c += 32) {
                s += (double)dbpart[(int64_t)c * Cout + co];
                s1 += (double)dbpart[(int64_t)(c + 8) * Cout + co];
                s2 += (double)dbpart[(int64_t)(c + 16) * Cout + co];
                s3 += (double)dbpart[(int64_t)(c + 24) * Cout + co];
            }
            for (; c < db_chunks; c += 8) s += (double)dbpart[(int64_t)c * Cout + co];
            s = (s + s1) + (s2 + s3);
        }
        sh[0][ty][tx] = s;
        __syncthreads();
        if (ty == 0 && co < Cout) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += sh[0][k][tx];
            db[co] = (float)a;
        }
        return;
    }
    for (int64_t i0 = (int64_t)bid * 64; i0 < n_out; i0 += (int64_t)nb_w * 64) {
        const int64_t i = i0 + tx;
        // this thread's chunks ty, ty + 8, ... of every sample: four chunks of all samples per trip, loads unconditional
        // (clamped sample / chunk / output index, the value dropped afterwards) -- the loop per sample with two loads in
        // flight was cps / 16 dependent round trips PER SAMPLE
        const int64_t ic = i < n_out ? i : n_out - 1;
        double sacc[WS_MAXN];
#pragma unroll
        for (int n = 0; n < WS_MAXN; ++n) sacc[n] = 0.0;
        for (int c0 = ty; c0 < cps; c0 += 32) {
            float v[WS_MAXN][4];
#pragma unroll
            for (int n = 0; n < WS_MAXN; ++n)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cc = c0 + 8 * u < cps ? c0 + 8 * u : cps - 1, nc = n < N ? n : N - 1;
                    v[n][u] = part[(int64_t)(nc * cps + cc) * chunk_stride + ic];
                }
#pragma unroll
            for (int n = 0; n < WS_MAXN; ++n)
#pragma unroll
                for (int u = 0; u < 4; ++u) sacc[n] += (n < N && c0 + 8 * u < cps) ? (double)v[n][u] : 0.0;
        }
#pragma unroll
        for (int n = 0; n < WS_MAXN; ++n) sh[n][ty][tx] = i < n_out ? sacc[n] : 0.0;
        __syncthreads();
        if (ty == 0) {  // wave 0: one lane per output
            double tot = 0.0;
            const bool ok = i < n_out;
            const int co = ok ? (int)(i % Cout) : 0;
            const int64_t r = ok ? i / Cout : 0;
            const int ci = (int)(r % Cin), tap = (int)(r / Cin);
            const int64_t widx = ((int64_t)co * Cin + ci) * ntaps + tap;
            const double wv = ok ? (double)w[widx] : 0.0;
#pragma unroll
            for (int n = 0; n < WS_MAXN; ++n) {
                if (n < N) {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) a += sh[n][k][tx];
                    tot += a;
                    double v = wv * a;  // lanes of a 32-group share (tap, ci): Cout % 32 == 0, i0 % 64 == 0
                    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                    if ((tx & 31) == 0 && ok) P[(int64_t)n * (n_out >> 5) + (i >> 5)] = (float)v;
                }
            }
            if (ok) out[widx] = (float)tot;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// block (ci group of 32, sample): class totals -> T[tap][co] -> A[ci], B[ci]  -> sums[n][ci][2] = (A, B)
//   taps are indexed as the kernels do: tap = (tz*3 + ty)*3 + tx, the conv reads x[v + (t - 1)], so dw[tap] pairs g[u]
//   with z[u + (t - 1)] and T[tap] sums g[u] over the u with u + (t - 1) inside the volume.
// Every reduction runs 4 lanes wide (slots) resp. 8 lanes wide (tap x co) with shuffles: no serial 100-step loops.
// ---------------------------------------------------------------------------
#ifndef TEM_NS_ABL
#define TEM_NS_ABL 0   // profiling ablations: 1 no class totals, 2 no bias-gradient total, 4 no weight loop
#endif
#define NS_WPRE 32     // weights per thread loaded before the first phase (all of them for Cout <= 32)
// The norm backward's second stage rides along (NormCoef.coef != nullptr): coef[n][c] = {a, m1, m2r, mean} exactly as
// k_norm_bwd_finalize derives them from the (A, B) rows -- the same doubles in the same order -- when the channels of a
// group lie inside one block's 32 (group size a power of two <= 32): one launch less per layer.
struct NormCoef {
    int cgn;           // channels per group (0: not requested)
    int G;
    double cnt;        // V * cgn
    const float* mean;
    const float* rstd;
    float* coef;
};
__global__ __launch_bounds__(1024) void k_norm_sums_from_wgrad(const float* __restrict__ planepart,
                                                              const float* __restrict__ dbpart, int Ss, int D, int H,
                                                              int Cin, int Cout, const float* __restrict__ w,
                                                              const float* __restrict__ P, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ sums,
                                                              NormCoef nc) {
    extern __shared__ float lt[];        // cls[27][Cout] then T[Cout][32] (tap fastest: the last phase reads it lane = tap)
    __shared__ float ab[32][2];
    float* cls = lt;
    float* T = lt + 27 * Cout;
    const int n = blockIdx.y, nslot = D - 2 + 2 * H;
    const int l4 = threadIdx.x & 3, e4 = threadIdx.x >> 2;  // bias-gradient total below: 256 entries per round, 4 row lanes each
    const int64_t n_out = (int64_t)27 * Cin * Cout;
    const int cg = Cout >> 5;
    const int ci = blockIdx.x * 32 + (threadIdx.x >> 5), l8 = threadIdx.x & 31;  // last phase: 32 lanes per input channel
    // Measured (TEM_NS_ABL builds, 32 -> 32 at 2 x 128^3, round 3): 39 us on 2-8 workgroups -- 14 for the class totals, 4 for
    // the bias-gradient total, 15 for the weight loop at the end, 5 for everything else -- each phase a chain of 2-4
    // dependent round trips to data another XCD just wrote.  The loads of the LAST phase depend on nothing computed here:
    // they are issued first (weights of this thread's (channel, tap) for the first NS_WPRE output channels, its P entries)
    // and are in registers by the time T[] exists.
    // Loads are never predicated per lane here: a lane-dependent `cond ? load : 0` compiles to a branch around the load
    // with s_waitcnt vmcnt(0) behind it -- one round trip per load.  Out-of-range lanes read a clamped (valid) address
    // and the value is dropped afterwards.
    float wpre[NS_WPRE];
    float ppre[2];
    const bool wlane = ci < Cin && l8 < 27 && !(TEM_NS_ABL & 4);
    const unsigned wbase = wlane ? (unsigned)ci * 27u + (unsigned)l8 : 0u;
#pragma unroll
    for (int k = 0; k < NS_WPRE; ++k) wpre[k] = w[(unsigned)k * (unsigned)Cin * 27u + wbase];   // Cout >= 32 = NS_WPRE
    const unsigned pbase = (unsigned)n * (unsigned)(n_out >> 5), cic = (unsigned)(ci < Cin ? ci : Cin - 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = l8 + 32 * k, jc = j < 27 * cg ? j : 27 * cg - 1;
        ppre[k] = P[pbase + ((unsigned)(jc / cg) * (unsigned)Cin + cic) * (unsigned)cg + (unsigned)(jc % cg)];
    }
    // the per-sample bias gradient (Ss partial rows) is the total over all classes; its loads, too, go out before the
    // class totals (Cout <= 256: one entry (co = e4) per group of 4 lanes)
    float tot = 0.f;
    {
        const int co = e4 < Cout ? e4 : Cout - 1;
        for (int sp = l4; sp < ((TEM_NS_ABL & 2) ? 0 : Ss); sp += 128) {
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int si = sp + 4 * k;
                v[k] = dbpart[(unsigned)(n * Ss + (si < Ss ? si : Ss - 1)) * (unsigned)Cout + (unsigned)co];
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = sp + 4 * k < Ss ? v[k] : 0.f;
#pragma unroll
            for (int k = 0; k < 32; k += 8)
                tot += ((v[k] + v[k + 1]) + (v[k + 2] + v[k + 3])) + ((v[k + 4] + v[k + 5]) + (v[k + 6] + v[k + 7]));
        }
        tot += __shfl_xor(tot, 1, 64);
        tot += __shfl_xor(tot, 2, 64);
    }
    // class totals: one thread per entry (z class, y/x class, co), consecutive lanes on consecutive floats of a slot row, 32
    // slots in flight
    for (int t = threadIdx.x; t < ((TEM_NS_ABL & 1) ? 0 : 27 * Cout); t += 1024) {
        const int cz = t / (9 * Cout), rem = t % (9 * Cout);
        const int s0 = cz == 1 ? 0 : (cz == 0 ? D - 2 : D - 2 + H), s1 = cz == 1 ? D - 2 : s0 + H;
        float a = 0.f;
        const unsigned rstride = 9u * (unsigned)Cout;   // 32-bit offsets from a uniform base: one VGPR per load in flight
        for (int sl = s0; sl < s1; sl += 32) {
            float v[32];
            const unsigned o0 = (unsigned)(n * nslot + sl) * rstride + (unsigned)rem;
            const int last = s1 - 1 - sl;
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = planepart[o0 + (unsigned)(k < last ? k : last) * rstride];
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = k <= last ? v[k] : 0.f;
#pragma unroll
            for (int k = 0; k < 32; k += 8)
                a += ((v[k] + v[k + 1]) + (v[k + 2] + v[k + 3])) + ((v[k + 4] + v[k + 5]) + (v[k + 6] + v[k + 7]));
        }
        cls[t] = a;
    }
    if (TEM_NS_ABL & 1)
        for (int t = threadIdx.x; t < 27 * Cout; t += 1024) cls[t] = 0.f;
    __syncthreads();
    // the interior class (1,1,1) = total - everything else
    if (l4 == 0 && e4 < Cout) {
        float rest = 0.f;
        for (int k = 0; k < 27; ++k)
            if (k != 13) rest += cls[k * Cout + e4];
        cls[13 * Cout + e4] = tot - rest;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 27 * Cout; t += 1024) {
        const int tap = t / Cout, co = t % Cout;
        const int d[3] = {tap / 9 - 1, (tap / 3) % 3 - 1, tap % 3 - 1};
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int c3[3] = {k / 9, (k / 3) % 3, k % 3};
            bool ok = true;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                if (d[ax] == 1 && c3[ax] == 2) ok = false;   // u is the last index: u + 1 is outside
                if (d[ax] == -1 && c3[ax] == 0) ok = false;  // u is the first index: u - 1 is outside
            }
            a += ok ? cls[k * Cout + co] : 0.f;
        }
        T[co * 32 + tap] = a;
    }
    __syncthreads();
    double A = 0.0, S2 = 0.0;
    if (ci < Cin) {
        // lane = tap, loop over co: a load instruction reads the 27 consecutive floats of w[co][ci][:] for two input
        // channels (the (tap, co)-per-lane version gathered 64 separate cache lines per instruction: 15 us)
        if (wlane) {   // Cout % 32 == 0: no bounds inside the unrolled loops (a test per k was a branch + wait per term)
#pragma unroll
            for (int k = 0; k < NS_WPRE; ++k) A += (double)wpre[k] * (double)T[k * 32 + l8];
            for (int c0 = NS_WPRE; c0 < Cout; c0 += 16) {
                float wv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) wv[k] = w[(unsigned)(c0 + k) * (unsigned)Cin * 27u + wbase];
#pragma unroll
                for (int k = 0; k < 16; ++k) A += (double)wv[k] * (double)T[(c0 + k) * 32 + l8];
            }
        }
        S2 = l8 < 27 * cg ? (double)ppre[0] : 0.0;
        if (l8 + 32 < 27 * cg) S2 += (double)ppre[1];
        for (int j = l8 + 64; j < 27 * cg; j += 32) {
            const int tap = j / cg, k = j % cg;
            S2 += (double)P[(int64_t)n * (n_out >> 5) + ((int64_t)tap * Cin + ci) * cg + k];
        }
    }
    for (int o = 1; o < 32; o <<= 1) {
        A += __shfl_xor(A, o, 64);
        S2 += __shfl_xor(S2, o, 64);
    }
    float fa = 0.f, fb = 0.f;
    if (ci < Cin && l8 == 0) {
        const double ga = gamma ? (double)gamma[ci] : 1.0, be = beta ? (double)beta[ci] : 0.0;
        const double B = fabs(ga) > 1e-30 ? (S2 - be * A) / ga : 0.0;
        fa = (float)A;
        fb = (float)B;
        sums[((int64_t)n * Cin + ci) * 2 + 0] = fa;
        sums[((int64_t)n * Cin + ci) * 2 + 1] = fb;
    }
    if (!nc.cgn) return;
    if (l8 == 0) {
        ab[threadIdx.x >> 5][0] = fa;
        ab[threadIdx.x >> 5][1] = fb;
    }
    __syncthreads();
    if (threadIdx.x < 64) {   // wave 0; lanes 0..31 = this block's channels, a group = cgn consecutive lanes
        const int t = threadIdx.x & 31, c = blockIdx.x * 32 + t;
        const bool ok = threadIdx.x < 32 && c < Cin;
        const double ga = (ok && gamma) ? (double)gamma[c] : 1.0;
        double s1 = ok ? ga * (double)ab[t][0] : 0.0, s2 = ok ? ga * (double)ab[t][1] : 0.0;
        for (int o = nc.cgn >> 1; o >= 1; o >>= 1) {   // the butterfly of k_norm_bwd_finalize's block sum
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        if (ok) {
            const int grp = c / nc.cgn;
            const double r = (double)nc.rstd[n * nc.G + grp], m = (double)nc.mean[n * nc.G + grp];
            float* o = nc.coef + ((int64_t)n * Cin + c) * 4;
            o[0] = (float)(r * ga);
            o[1] = (float)(r * s1 / nc.cnt);
            o[2] = (float)(r * r * s2 / nc.cnt);
            o[3] = (float)m;
        }
    }
}

// workspace (floats) behind the slab workspace: P [N][27*Cin*Cout/32] + planepart [N][D-2+2H][9][Cout]
int64_t tem_wgrad_sums_ws_floats(int N, int D, int H, int Cin, int Cout) {
    return tem_align_up((int64_t)N * 27 * Cin * Cout / 32, 64) + (int64_t)N * (D - 2 + 2 * H) * 9 * Cout;
}

void tem_wgrad_sums_launch(const float* zpart, int Ss, int ks2, const float* zdb, const float* g, int64_t g_ld,
                           const float* w, const float* gamma, const float* beta, float* dw, float* extra, int N, int D,
                           int H, int W, int Cin, int Cout, float* sums, int db_chunks, float* db, hipStream_t s) {
    float* P = extra;
    float* planepart = extra + tem_align_up((int64_t)N * 27 * Cin * Cout / 32, 64);
    const int64_t n_out = (int64_t)27 * Cin * Cout;
    int64_t nb = tem_cdiv(n_out, 64);
    if (nb > 4096) nb = 4096;
    const int nb_db = db ? (Cout + 63) / 64 : 0;   // zdb: [db_chunks][Cout] rows -> db
    const ShellArgs sa = {g, tem_call_st.y, g_ld, D, H, W, planepart};
    hipLaunchKernelGGL(k_reduce_slabs_wsum, dim3((unsigned)(nb + nb_db + ((int64_t)(D - 2 + 2 * H) * N + 1) / 2)), dim3(512),
                       (size_t)2 * 4 * 9 * Cout * sizeof(float), s, zpart,
                       Ss * ks2, N, 27, Cin, Cout, n_out, dw, w, P, (int)nb, zdb, db_chunks, db, nb_db, sa);
    NormCoef nc = {0, 0, 0.0, nullptr, nullptr, nullptr};
    TemWgradCoefReq rq = {0, nullptr, nullptr, nullptr};
    if (tem_bp_wants(TEM_BP_NORM_COEF)) rq = TemWgradCoefReq{tem_call_bp->coef_G, tem_call_bp->coef_mean, tem_call_bp->coef_rstd, tem_call_bp->coef};
    const int cgn = (rq.coef && rq.mean && rq.rstd && rq.G > 0 && Cin % rq.G == 0) ? Cin / rq.G : 0;
    if (cgn >= 1 && cgn <= 32 && (cgn & (cgn - 1)) == 0 && Cin % cgn == 0) {   // else: not delivered, the caller runs tem_norm_bwd_coef
        tem_bp_delivered(TEM_BP_NORM_COEF);
        nc.cgn = cgn;
        nc.G = rq.G;
        nc.cnt = (double)((int64_t)D * H * W) * (double)nc.cgn;
        nc.mean = rq.mean;
        nc.rstd = rq.rstd;
        nc.coef = rq.coef;
    }
    hipLaunchKernelGGL(k_norm_sums_from_wgrad, dim3((Cin + 31) / 32, N), dim3(1024), (size_t)(27 + 32) * Cout * sizeof(float),
                       s, planepart, zdb, Ss, D, H, Cin, Cout, w, P, gamma, beta, sums, nc);
}
