// wino_harness.cpp -- developer harness for conv_wino.hip (no torch): checks one shape against a float64 direct
// convolution on the host (small shapes) and times the kernel.  Build + run: scripts/wino_harness.sh.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../torch_em_amd/csrc/tem_common.h"
int64_t tem_conv_wino_pack_bytes(int Cin, int Cout);
void tem_pack_weights_wino(const float* w, void* dst, int Cw_out, int Cw_in, int transpose, int f16, hipStream_t s);
bool tem_conv_fwd_wino(const float* x, int64_t x_ld, const float* scale, const float* shift, const void* wp,
                       const float* bias, float* y, int64_t y_ld, const float* ref, int64_t ref_ld, int N, int D, int H,
                       int W, int Cin, int Cout, int act, int f16, hipStream_t s);
#ifdef WN_TRACE
void tem_wn_trace_read(unsigned long long* dst);
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(uint64_t& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((s >> 40) & 0xffffff) / 8388608.f - 1.f;
}

int main(int argc, char** argv) {
    if (argc < 8) { printf("usage: %s N D H W Cin Cout f16 [iters] [check] [norm] [ref] [transpose]\n", argv[0]); return 1; }
    const int N = atoi(argv[1]), D = atoi(argv[2]), H = atoi(argv[3]), W = atoi(argv[4]), Cin = atoi(argv[5]), Cout = atoi(argv[6]);
    const int f16 = atoi(argv[7]);
    const int iters = argc > 8 ? atoi(argv[8]) : 10, check = argc > 9 ? atoi(argv[9]) : 0, use_norm = argc > 10 ? atoi(argv[10]) : 1;
    const int use_ref = argc > 11 ? atoi(argv[11]) : 0, transpose = argc > 12 ? atoi(argv[12]) : 0;
    const size_t V = (size_t)N * D * H * W;
    uint64_t seed = 1234;
    // weight tensor as stored: [Cw_out][Cw_in][27]; transpose: the executed conv maps Cw_out -> Cw_in channels
    const int Cw_out = transpose ? Cin : Cout, Cw_in = transpose ? Cout : Cin;
    std::vector<float> hx(V * Cin), hw((size_t)Cout * Cin * 27), hb(Cout), hs((size_t)N * Cin), hf((size_t)N * Cin), hr;
    for (auto& v : hx) v = 2.f * frand(seed);
    for (auto& v : hw) v = 0.05f * frand(seed);
    for (auto& v : hb) v = frand(seed);
    for (auto& v : hs) v = 1.f + 0.5f * frand(seed);
    for (auto& v : hf) v = frand(seed);
    float *x, *w, *b, *sc, *sf, *y, *ref = nullptr;
    void* wp;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&wp, tem_conv_wino_pack_bytes(Cin, Cout)));
    CK(hipMalloc(&b, Cout * 4)); CK(hipMalloc(&sc, hs.size() * 4)); CK(hipMalloc(&sf, hf.size() * 4));
    CK(hipMalloc(&y, V * Cout * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sf, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y, 0xff, V * Cout * 4));
    if (use_ref) {
        hr.resize(V * Cout);
        for (auto& v : hr) v = frand(seed);
        CK(hipMalloc(&ref, hr.size() * 4));
        CK(hipMemcpy(ref, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s = 0;
    tem_pack_weights_wino(w, wp, Cw_out, Cw_in, transpose, f16, s);
    auto run = [&]() {
        if (!tem_conv_fwd_wino(x, Cin, use_norm ? sc : nullptr, use_norm ? sf : nullptr, wp, b, y, Cout, ref, Cout, N, D, H, W, Cin,
                               Cout, TEM_ACT_RELU, f16, s)) { printf("shape not taken\n"); exit(1); }
    };
    run();
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> hy(V * Cout);
        CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, ymax = 0;
        size_t nbad = 0;
        for (int n = 0; n < N; ++n)
            for (int z = 0; z < D; ++z)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx)
                        for (int co = 0; co < Cout; ++co) {
                            double a = hb[co];
                            for (int dz = 0; dz < 3; ++dz)
                                for (int dy = 0; dy < 3; ++dy)
                                    for (int dx = 0; dx < 3; ++dx) {
                                        const int zi = z + dz - 1, yi = yy + dy - 1, xi = xx + dx - 1;
                                        if (zi < 0 || zi >= D || yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
                                        const float* px = &hx[((((size_t)n * D + zi) * H + yi) * W + xi) * Cin];
                                        for (int ci = 0; ci < Cin; ++ci) {
                                            double v = px[ci];
                                            if (use_norm) v = (double)(float)(px[ci] * hs[n * Cin + ci] + hf[n * Cin + ci]);
                                            const double wv = transpose ? hw[((size_t)ci * Cw_in + co) * 27 + (2 - dz) * 9 + (2 - dy) * 3 + (2 - dx)]
                                                                        : hw[((size_t)co * Cw_in + ci) * 27 + dz * 9 + dy * 3 + dx];
                                            a += v * wv;
                                        }
                                    }
                            if (a < 0) a = 0;
                            const size_t o = ((((size_t)n * D + z) * H + yy) * W + xx) * Cout + co;
                            if (use_ref && !(hr[o] > 0)) a = 0;
                            const double e = fabs(a - (double)hy[o]);
                            if (!(e <= 1e30)) ++nbad;
                            if (e > emax) emax = e;
                            if (fabs(a) > ymax) ymax = fabs(a);
                        }
        printf("check %dx%dx%dx%d %d->%d f16 %d norm %d ref %d transpose %d: max err %.3e / max |y| %.3e = %.3e  (nan %zu)\n", N, D, H, W,
               Cin, Cout, f16, use_norm, use_ref, transpose, emax, ymax, emax / ymax, nbad);
    }
    if (iters > 0) {
        for (int i = 0; i < 3; ++i) run();
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) run();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= iters; if (ms < best) best = ms;
        }
        const double fl = 2.0 * V * Cin * Cout * 27;
        printf("wino %dx%dx%dx%d %d->%d f16 %d: min %.4f ms  %.0f TF alg (direct-equivalent)  executed-MFMA frac(2500) %.3f\n", N, D, H, W, Cin,
               Cout, f16, best, fl / best / 1e9, fl * 3 / 2.25 / best / 1e9 / 2500);
    }
#ifdef WN_TRACE
    {
        std::vector<unsigned long long> tr(4 * 64 * 8);
        tem_wn_trace_read(tr.data());
        const unsigned long long base = tr[0];
        for (int w = 0; w < 4; ++w) {
            printf("wave %d: step @start  loads_issue  half0  half1  stage_store  barrier  combine   (shader cycles)\n", w);
            for (int st = 0; st < 24; ++st) {
                const unsigned long long* t = &tr[(w * 64 + st) * 8];
                if (!t[0]) break;
                printf("  %2d @%8lld: %6lld %6lld %6lld %6lld %6lld %6lld\n", st, (long long)(t[0] - base), (long long)(t[1] - t[0]),
                       (long long)(t[2] - t[1]), (long long)(t[3] - t[2]), (long long)(t[4] - t[3]), (long long)(t[5] - t[4]),
                       (long long)(t[6] - t[5]));
            }
        }
    }
#endif
    return 0;
}
