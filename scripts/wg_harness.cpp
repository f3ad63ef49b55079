// wg_harness.cpp -- developer harness for the weight-gradient kernel (no torch); see scripts/wg_harness.sh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <math.h>
#include "../torch_em_amd/csrc/tem_common.h"
#include "../torch_em_amd/csrc/conv_internal.h"
#include "../torch_em_amd/csrc/tem_act.h"
#ifdef TEM_ZS_TRACE
void tem_zs_trace_read(unsigned long long* dst);
#endif
#ifdef TEM_TR_TRACE
void tem_tr_trace_read(unsigned long long* dst);
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
static float frand(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) / 8388608.f - 1.f; }

int main(int argc, char** argv) {
    if (argc < 7) { printf("usage: %s N D H W Cin Cout [iters]\n", argv[0]); return 1; }
    const int N = atoi(argv[1]), D = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), Cin = atoi(argv[5]), Cout = atoi(argv[6]);
    const int iters = argc > 7 ? atoi(argv[7]) : 10;
    const size_t V = (size_t)N * D * H * W;
    uint64_t seed = 99;
    std::vector<float> hx(V * Cin), hg(V * Cout), hs((size_t)N * Cin), hf((size_t)N * Cin);
    for (auto& v : hx) v = 2.f * frand(seed);
    for (auto& v : hg) v = frand(seed);
    if (getenv("WG_GZERO")) {   // a fraction of exact zeros in g (ReLU masks): the step's gradients are sparser / cooler than noise
        const float frac = (float)atof(getenv("WG_GZERO"));
        for (auto& v : hg) if (0.5f * (frand(seed) + 1.f) < frac) v = 0.f;
    }
    for (auto& v : hs) v = 1.f + 0.5f * frand(seed);
    for (auto& v : hf) v = frand(seed);
    float *x, *g, *sc, *sf, *dw, *db;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&g, hg.size() * 4)); CK(hipMalloc(&sc, hs.size() * 4)); CK(hipMalloc(&sf, hf.size() * 4));
    CK(hipMalloc(&dw, (size_t)Cin * Cout * 27 * 4)); CK(hipMalloc(&db, Cout * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sf, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    tem_set_option("wgrad_zs", 1);
    int64_t wsb = tem_conv_wgrad_bf16x3_ws(N, D, H, W, Cin, Cout, 3, 3, 3);
    for (int o = 2; o <= 3; ++o) {
        tem_set_option("wgrad_zs", o);
        const int64_t wsb2 = tem_conv_wgrad_bf16x3_ws(N, D, H, W, Cin, Cout, 3, 3, 3);
        if (wsb2 > wsb) wsb = wsb2;
    }
    void* ws; CK(hipMalloc(&ws, wsb));
    hipStream_t s = 0;
    const int zsopt = argc > 8 ? atoi(argv[8]) : 2;   // option wgrad_zs: 1 k_conv_wgrad_zs, 2 _zt (staging team), 3 _tr (transposing reads)
    int h16 = getenv("WG_ONE") ? atoi(getenv("WG_ONE")) : 0;   // 0 bf16x3, 1 fp16, 2 bf16, 3 fp16 2x1 (prescaled g)
    unsigned* amax; CK(hipMalloc(&amax, 4)); CK(hipMemset(amax, 0, 4));
    if (tem_absmax(g, Cout, Cout, (int64_t)V, amax, s)) { printf("absmax failed: %s\n", tem_last_error()); return 1; }
    // WG_ST=1|2: x and g as fp16 / bf16 tensors (round 5: 16-bit activation storage; WG_ONE must be the one-term mode of the type)
    const int st = getenv("WG_ST") ? atoi(getenv("WG_ST")) : 0;
    if (st) {
        std::vector<unsigned short> x16(hx.size()), g16(hg.size());
        auto cv = [&](float v) -> unsigned short {
            if (st == 1) { _Float16 h = (_Float16)v; unsigned short u; memcpy(&u, &h, 2); return u; }
            unsigned u; memcpy(&u, &v, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        };
        for (size_t i = 0; i < hx.size(); ++i) x16[i] = cv(hx[i]);
        for (size_t i = 0; i < hg.size(); ++i) g16[i] = cv(hg[i]);
        CK(hipMemcpy(x, x16.data(), x16.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(g, g16.data(), g16.size() * 2, hipMemcpyHostToDevice));
    }
    auto run = [&]() {
        TemStScope sc_(st, st);
        tem_wgrad_gscale_source = h16 == 3 ? amax : nullptr;
        int rc = tem_conv_wgrad_bf16x3(x, Cin, sc, sf, g, Cout, dw, db, ws, wsb, N, D, H, W, Cin, Cout, 3, 3, 3, 1, h16, nullptr, nullptr,
                                       nullptr, nullptr, s);
        tem_wgrad_gscale_source = nullptr;
        if (rc) { printf("launch failed: %s\n", tem_last_error()); exit(1); }
    };
    if (getenv("HARNESS_CHECK_ARITH")) {   // the arithmetic variant WG_ONE against bf16x3 (expected: the rounding of the variant)
        std::vector<float> a((size_t)Cin * Cout * 27), b(a.size());
        const int want = h16;
        tem_set_option("wgrad_zs", 1);
        h16 = 0; run(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(a.data(), dw, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dw, 0xff, a.size() * 4));
        tem_set_option("wgrad_zs", zsopt);
        h16 = want; run(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(b.data(), dw, b.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0, l2d = 0, l2 = 0;
        for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); l2d += ((double)a[i] - b[i]) * ((double)a[i] - b[i]); l2 += (double)a[i] * a[i]; }
        printf("CHECK zs=%d h16=%d vs zs=1 bf16x3: max |dw diff| %.3e (max |dw| %.3e), relative L2 %.3e\n", zsopt, want, md, mx, sqrt(l2d / l2));
    }
    if (getenv("HARNESS_CHECK")) {   // the staging-team kernel (wgrad_zs = 2) against the round-2 kernel (1)
        std::vector<float> a((size_t)Cin * Cout * 27), b(a.size()), da(Cout), dbv(Cout);
        tem_set_option("wgrad_zs", 1); run(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(a.data(), dw, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(da.data(), db, Cout * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dw, 0xff, a.size() * 4));
        tem_set_option("wgrad_zs", 2); run(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(b.data(), dw, b.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(dbv.data(), db, Cout * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0, mdb = 0;
        for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); }
        for (int i = 0; i < Cout; ++i) mdb = fmax(mdb, fabs((double)da[i] - dbv[i]));
        printf("CHECK teams vs round-2 kernel: max |dw diff| %.3e (max |dw| %.3e), max |db diff| %.3e\n", md, mx, mdb);
    }
    tem_set_option("wgrad_zs", zsopt);
    for (int i = 0; i < 3; ++i) run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters; if (ms < best) best = ms;
    }
    const double fl = 2.0 * V * Cin * Cout * 27;
    printf("wgrad[zs=%d] %dx%dx%dx%d %d->%d: min %.4f ms (kernel + slab merge)  %.0f TF alg  mfma_frac(2500) %.3f\n", zsopt, N, D, H, W, Cin, Cout, best,
           fl / best / 1e9, fl * 3 / best / 1e9 / 2500);
#ifdef TEM_TR_TRACE
    {
        std::vector<unsigned long long> tt(8 * 64 * 4);
        tem_tr_trace_read(tt.data());
        for (int wv : {0, 3, 4, 7}) {
            printf("wave %d (%s): plane  start->%s  ->%s  barrier   plane-to-plane   (s_memtime ticks)\n", wv, wv < 4 ? "multiply" : "stage",
                   wv < 4 ? "prologue" : "stored", wv < 4 ? "mfma done" : "loads issued");
            for (int it = 4; it < 40; ++it) {
                const unsigned long long* t = &tt[(wv * 64 + it) * 4];
                const unsigned long long* tp = &tt[(wv * 64 + it - 1) * 4];
                if (!t[0] || !t[3]) break;
                printf("  %2d: %6lld %6lld %6lld   %6lld\n", it, (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]),
                       (long long)(t[0] - tp[0]));
            }
        }
    }
#endif
#ifdef TEM_ZS_TRACE
    std::vector<unsigned long long> tr(8 * 64 * 8);
    tem_zs_trace_read(tr.data());
    const unsigned long long base = tr[0];
    for (int wv : {0, 1, 4, 7}) {
        printf("wave %d: iter   mfma  stores  loads  barrier   (shader cycles)\n", wv);
        for (int it = 0; it < 24; ++it) {
            const unsigned long long* t = &tr[(wv * 64 + it) * 8];
            if (!t[0]) break;
            printf("  %2d @%8llu: %6lld %6lld %6lld %6lld\n", it, t[0] - base, (long long)(t[1] - t[0]), (long long)(t[2] - t[1]),
                   (long long)(t[3] - t[2]), (long long)(t[4] - t[3]));
        }
    }
#endif
    return 0;
}
