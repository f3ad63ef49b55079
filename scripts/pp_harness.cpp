// pp_harness.cpp -- developer harness for the forward / dgrad convolution kernels (no torch): times one launch shape
// through the library's internal entry point and, when conv_pp.hip is compiled in with -DTEM_PP_TRACE, prints the phase
// timeline of one workgroup.  Build + run: scripts/pp_harness.sh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../torch_em_amd/csrc/tem_common.h"
#include "../torch_em_amd/csrc/conv_internal.h"
#ifdef TEM_PP_TRACE
void tem_pp_trace_read(unsigned long long* dst);
#endif
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
    const size_t V = (size_t)N * D * H * W;
    uint64_t seed = 1234;
    std::vector<float> hx(V * Cin), hw((size_t)Cout * Cin * 27), hb(Cout), hs((size_t)N * Cin), hf((size_t)N * Cin);
    for (auto& v : hx) v = 2.f * frand(seed);
    for (auto& v : hw) v = 0.05f * frand(seed);
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
    hipStream_t s = 0;
    tem_pack_weights_bf16x3(w, wp, Cout, Cin, 3, 3, 3, 0, mode, s);
    void* ws = nullptr;
    const int64_t wsb = tem_conv_fwd_mfma_ws(N, D, H, W, Cin, Cout, 3, 3, 3);
    if (wsb) CK(hipMalloc(&ws, wsb));
    auto run = [&]() {
        float* st = use_norm && !use_ref ? stat : nullptr;
        // variants >= 1: straight into THIS executable's copy of conv_pp.hip (calls inside libtem_hip.so bind locally)
        if (variant >= 1 && tem_conv_fwd_pp(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, N, D,
                                            H, W, Cin, Cout, 3, 3, 3, TEM_ACT_RELU, mode, st, s))
            return;
        int rc = tem_conv_fwd_bf16x3(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, ws, wsb, N,
                                     D, H, W, Cin, Cout, 3, 3, 3, TEM_ACT_RELU, mode, st, s);
        if (rc) { printf("launch failed: %s\n", tem_last_error()); exit(1); }
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
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
#ifdef TEM_PP_TRACE
    std::vector<unsigned long long> tr(2 * 64 * 8);
    tem_pp_trace_read(tr.data());
    for (int team = 0; team < 2; ++team) {
        printf("team %d: step  loads_issue  epilogue  convert  prime  bar1   taps   bar2   (shader cycles; t0 relative to team 0 step 0)\n", team);
        const unsigned long long base = tr[0];
        for (int st = 0; st < 20; ++st) {
            const unsigned long long* t = &tr[(team * 64 + st) * 8];
            if (!t[0]) break;
            printf("  %2d @%8llu: %6lld %6lld %6lld %6lld %6lld %6lld %6lld\n", st, t[0] - base, (long long)(t[1] - t[0]),
                   (long long)(t[6] - t[1]), (long long)(t[7] ? t[7] - t[6] : 0), (long long)(t[7] ? t[2] - t[7] : t[2] - t[6]),
                   (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(t[5] - t[4]));
        }
    }
#endif
    return 0;
}
