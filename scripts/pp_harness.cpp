// pp_harness.cpp -- developer harness for the forward / dgrad convolution kernels (no torch): times one launch shape
// through the library's internal entry point and, when conv_pp.hip is compiled in with -DTEM_PP_TRACE, prints the phase
// timeline of one workgroup.  Build + run: scripts/pp_harness.sh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>
#include "../torch_em_amd/csrc/tem_common.h"
#include "../torch_em_amd/csrc/conv_internal.h"
#ifdef TEM_PP_TRACE
void tem_pp_trace_read(unsigned long long* dst);
#endif
#ifdef TEM_ZR_TRACE
void tem_zr_trace_read(unsigned long long* dst);
#endif
#include "../torch_em_amd/csrc/tem_act.h"
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint64_t& s) {  // uniform [-1, 1)
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((s >> 40) & 0xffffff) / 8388608.f - 1.f;
}

int main(int argc, char** argv) {
    if (argc < 9) { printf("usage: %s N D H W Cin Cout mode variant [iters] [norm] [ref]\n", argv[0]); return 1; }
    const int N = atoi(argv[1]), D = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), Cin = atoi(argv[5]), Cout = atoi(argv[6]);
    const int mode = atoi(argv[7]), variant = atoi(argv[8]);
    const int iters = argc > 9 ? atoi(argv[9]) : 10, use_norm = argc > 10 ? atoi(argv[10]) : 1, use_ref = argc > 11 ? atoi(argv[11]) : 0;
    tem_set_option("conv_fwd_variant", variant);
    if (getenv("ZR_TILE_BLOCKS")) tem_set_option("zr_tile_blocks", atoi(getenv("ZR_TILE_BLOCKS")));   // A/B of the tile order
    if (getenv("ZR_WIDE")) tem_set_option("zr_wide", atoi(getenv("ZR_WIDE")));
    if (getenv("FP32_ZR")) tem_set_option("fp32_zr", atoi(getenv("FP32_ZR")));   // 2: exact fp32 with one team per workgroup (mode 1)
    const size_t V = (size_t)N * D * H * W;
    uint64_t seed = 1234;
    std::vector<float> hx(V * Cin), hw((size_t)Cout * Cin * 27), hb(Cout), hs((size_t)N * Cin), hf((size_t)N * Cin);
    for (auto& v : hx) v = 2.f * frand(seed);
    if (getenv("ZERO_X")) for (auto& v : hx) v = 0.f;   // power experiment: all-zero activations (run with norm 0: the staged operands are zeros)
    for (auto& v : hw) v = 0.05f * frand(seed);
    if (getenv("ZERO_W")) for (auto& v : hw) v = 0.f;
    for (auto& v : hb) v = frand(seed);
    for (auto& v : hs) v = 1.f + 0.5f * frand(seed);
    for (auto& v : hf) v = frand(seed);
    float *x, *w, *wp, *b, *sc, *sf, *y, *ref = nullptr, *stat;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&wp, hw.size() * 8));
    CK(hipMalloc(&b, Cout * 4)); CK(hipMalloc(&sc, hs.size() * 4)); CK(hipMalloc(&sf, hf.size() * 4));
    CK(hipMalloc(&y, V * Cout * 4)); CK(hipMalloc(&stat, (size_t)64 << 20));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sf, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    if (use_ref) {
        std::vector<float> hr(V * Cout);
        for (auto& v : hr) v = frand(seed);
        CK(hipMalloc(&ref, hr.size() * 4));
        CK(hipMemcpy(ref, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    }
    // ZR_ST=1|2: x, ref, y as fp16 / bf16 tensors (round 5: 16-bit activation storage; mode must be 5 / 7)
    const int st = getenv("ZR_ST") ? atoi(getenv("ZR_ST")) : 0;
    if (st) {
        auto cv = [&](float v) -> unsigned short {
            if (st == 1) { _Float16 h = (_Float16)v; unsigned short u; memcpy(&u, &h, 2); return u; }
            unsigned u; memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        };
        std::vector<unsigned short> x16(hx.size());
        for (size_t i = 0; i < hx.size(); ++i) x16[i] = cv(hx[i]);
        CK(hipMemcpy(x, x16.data(), x16.size() * 2, hipMemcpyHostToDevice));
        if (ref) {
            std::vector<unsigned short> r16(V * Cout);
            for (auto& v : r16) v = cv(frand(seed));
            CK(hipMemcpy(ref, r16.data(), r16.size() * 2, hipMemcpyHostToDevice));
        }
    }
    hipStream_t s = 0;
    if (mode == 1) {   // exact fp32 on the z-reuse kernel (round 6): the TEM_WL_MFMA pack
        if (tem_conv_pack_weights(w, wp, Cout, Cin, 3, 3, 3, 0, TEM_WL_MFMA, s)) { printf("pack failed: %s\n", tem_last_error()); return 1; }
    } else
    tem_pack_weights_bf16x3(w, wp, Cout, Cin, 3, 3, 3, 0, mode, s);
    void* ws = nullptr;
    const int64_t wsb = tem_conv_fwd_mfma_ws(N, D, H, W, Cin, Cout, 3, 3, 3);
    if (wsb) CK(hipMalloc(&ws, wsb));
    auto run = [&]() {
        TemStScope sc_(st, st);
        float* stp = use_norm && !use_ref ? stat : nullptr;
        // variants >= 1: straight into THIS executable's copy of conv_pp.hip (calls inside libtem_hip.so bind locally)
        if (variant == 2 && tem_conv_fwd_zr(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, N, D,
                                            H, W, Cin, Cout, 3, 3, 3, TEM_ACT_RELU, mode, stp, s) > 0)
            return;
        if (variant == 1 && tem_conv_fwd_pp(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, N, D,
                                            H, W, Cin, Cout, 3, 3, 3, TEM_ACT_RELU, mode, stp, s))
            return;
        int rc = tem_conv_fwd_bf16x3(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, ws, wsb, N,
                                     D, H, W, Cin, Cout, 3, 3, 3, TEM_ACT_RELU, mode, stp, s);
        if (rc) { printf("launch failed: %s\n", tem_last_error()); exit(1); }
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    if (getenv("HARNESS_CHECK") && mode != 1) {   // the kernel under test against the library's one-patch-per-workgroup kernel
        std::vector<float> ya(V * Cout), yb(V * Cout);
        CK(hipMemcpy(ya.data(), y, ya.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(y, 0xff, V * Cout * 4));
        tem_set_option("conv_fwd_variant", 0);
        int rc = tem_conv_fwd_bf16x3(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, ws, wsb, N,
                                     D, H, W, Cin, Cout, 3, 3, 3, TEM_ACT_RELU, mode, nullptr, s);
        if (rc) { printf("reference launch failed: %s\n", tem_last_error()); exit(1); }
        CK(hipDeviceSynchronize());
        tem_set_option("conv_fwd_variant", variant);
        CK(hipMemcpy(yb.data(), y, yb.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t bad = 0, nan = 0;
        for (size_t i = 0; i < ya.size(); ++i) {
            if (ya[i] != ya[i]) { ++nan; continue; }
            const double d = fabs((double)ya[i] - yb[i]);
            if (d > md) md = d;
            if (fabs(yb[i]) > mx) mx = fabs(yb[i]);
            if (d > 1e-3) ++bad;
        }
        printf("CHECK vs patch kernel: max abs diff %.3e (max |y| %.3e), %zu elements off by > 1e-3, %zu NaN\n", md, mx, bad, nan);
        run();
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, tot = 0.f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= iters; tot += ms; if (ms < best) best = ms;
    }
    const double fl = 2.0 * V * Cin * Cout * 27;
    printf("%dx%dx%dx%d %d->%d mode %d variant %d norm %d ref %d: min %.4f ms avg %.4f ms  %.0f TF alg  mfma_frac(2500) %.3f\n", N, D, H, W,
           Cin, Cout, mode, variant, use_norm, use_ref, best, tot / 3, fl / best / 1e9, fl * 3 / best / 1e9 / 2500);
#if defined(TEM_PP_TRACE) || defined(TEM_ZR_TRACE)
#ifdef TEM_ZR_TRACE
    const int TS = 12;   // stamps per step
#else
    const int TS = 8;
#endif
    std::vector<unsigned long long> tr(2 * 64 * TS);
#ifdef TEM_ZR_TRACE
    tem_zr_trace_read(tr.data());
#else
    tem_pp_trace_read(tr.data());
#endif
    for (int team = 0; team < 2; ++team) {
        printf("team %d: step  loads_issue  epilogue  convert  prime  bar1   taps   bar2   (shader cycles; t0 relative to team 0 step 0)\n", team);
        const unsigned long long base = tr[0];
        for (int st = 0; st < 20; ++st) {
            const unsigned long long* t = &tr[(team * 64 + st) * TS];
            if (!t[0]) break;
            printf("  %2d @%8llu: %6lld %6lld %6lld %6lld %6lld %6lld %6lld\n", st, t[0] - base, (long long)(t[1] - t[0]),
                   (long long)(t[6] - t[1]), (long long)(t[7] ? t[7] - t[6] : 0), (long long)(t[7] ? t[2] - t[7] : t[2] - t[6]),
                   (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(t[5] - t[4]));
            if (TS > 8 && t[8] && t[11] > t[1])   // epilogue split: fold | planes (max, statistics, stores) | reduction | re-init
                printf("        epilogue: head+fold %lld  planes %lld  stat-reduce %lld  re-init %lld\n", (long long)(t[8] - t[1]),
                       (long long)(t[9] - t[8]), (long long)(t[10] - t[9]), (long long)(t[11] - t[10]));
        }
    }
#endif
    return 0;
}
