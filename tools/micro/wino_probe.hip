// wino_probe.hip - where a workgroup of conv_wino_batch_kernel spends its time.  Diagnostic harness: compiles the
// library's conv_wino.hip with -DNST_WINO_STAMPS (s_memtime at the phase boundaries into a buffer of its own) and runs
// one launch shape of the L=2 closure on random data:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DNST_WINO_STAMPS [-D...] -Iartstyletransfer_amd/csrc tools/micro/wino_probe.hip -o /tmp/wp
//   /tmp/wp 256 256 256 384     (Cin Cout H W of level 0; levels 1, 2 are H/2 x W/2, H/4 x W/4)      [unpool=0] [relu=1]
// Prints the kernel time (HIP events, median of 20) and, per phase, the median / p10 / p90 over workgroups in shader
// cycles and in microseconds at the in-kernel clock (s_memtime ticks / s_memrealtime 100-MHz ticks).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../artstyletransfer_amd/csrc/conv_wino.hip"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
    } while (0)

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 256, Cout = argc > 2 ? atoi(argv[2]) : 256;
    const int H0 = argc > 3 ? atoi(argv[3]) : 256, W0 = argc > 4 ? atoi(argv[4]) : 384;
    const int unpool = argc > 5 ? atoi(argv[5]) : 0, relu = argc > 6 ? atoi(argv[6]) : 1;
    const int nobits = argc > 7 ? atoi(argv[7]) : 0, pool = argc > 8 ? atoi(argv[8]) : 0;      // experiment switches: no mask words / pooled output too
    const int nimg = 3;
    CK(nst::conv_wino_init_device());
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    nst::ConvBatch b{};
    b.n = nimg; b.Cin = Cin; b.Cout = Cout; b.relu = relu; b.unpool = unpool;
    const int nch = Cin / 32;
    const size_t wbytes = (size_t)(Cout / 128) * nch * 3 * 8 * 2 * 4 * 64 * 16;
    {
        std::vector<_Float16> w(wbytes / 2);
        for (auto& v : w) v = (_Float16)(nd(rng) * 2000.f);
        void* d; CK(hipMalloc(&d, wbytes)); CK(hipMemcpy(d, w.data(), wbytes, hipMemcpyHostToDevice));
        b.wt_wino = d; b.wt_wino_inv = 1.f / 16384.f;
    }
    { float* d; CK(hipMalloc(&d, Cout * 4)); CK(hipMemset(d, 0, Cout * 4)); b.bias = d; }
    size_t out_bytes = 0;
    for (int i = 0; i < nimg; ++i) {
        nst::ConvImage& im = b.img[i];
        im.H = H0 >> i; im.W = W0 >> i;
        const size_t px_out = (size_t)im.H * im.W, px_in = unpool ? (size_t)(im.H / 2) * (im.W / 2) : px_out;
        std::vector<float> x(px_in * Cin);
        for (auto& v : x) v = nd(rng);
        float* din; CK(hipMalloc(&din, x.size() * 4)); CK(hipMemcpy(din, x.data(), x.size() * 4, hipMemcpyHostToDevice));
        im.in = din;
        float* dout; CK(hipMalloc(&dout, px_out * Cout * 4)); im.out = dout; out_bytes += px_out * Cout * 4;
        unsigned* am; CK(hipMalloc(&am, 64 * 4));
        std::vector<unsigned> amv(64, 0x40a00000u);      // 5.0f
        CK(hipMemcpy(am, amv.data(), 256, hipMemcpyHostToDevice)); im.amax_in = am;
        unsigned* ao; CK(hipMalloc(&ao, 64 * 4)); CK(hipMemset(ao, 0, 256)); im.amax_out = ao;
        if (relu && !nobits) { unsigned* bo; CK(hipMalloc(&bo, px_out * (Cout / 32) * 4)); im.bits_out = bo; }
        if (relu && pool) {
            float* po; CK(hipMalloc(&po, (px_out / 4 + 1) * Cout * 4)); im.pool_out = po;
            unsigned* pc; CK(hipMalloc(&pc, (px_out / 4 + 1) * (Cout / 32) * 16)); im.pcode_out = pc;
        }
        else { unsigned* bi; CK(hipMalloc(&bi, px_out * (Cout / 32) * 4)); CK(hipMemset(bi, 0xFF, px_out * (Cout / 32) * 4)); im.bits_in = bi; }
        if (unpool) {
            std::vector<unsigned> code(px_in * (Cin / 32) * 4);
            for (auto& v : code) v = rng();
            unsigned* pc; CK(hipMalloc(&pc, code.size() * 4)); CK(hipMemcpy(pc, code.data(), code.size() * 4, hipMemcpyHostToDevice));
            im.pcode_in = pc;
        }
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 60; ++it) {
        CK(hipEventRecord(e0, st));
        CK(nst::launch_conv_wino_batch(b, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 40) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    int tiles = 0;
    for (int i = 0; i < nimg; ++i) tiles += ((b.img[i].W + 15) / 16) * ((b.img[i].H + 7) / 8);
    const int wgs = tiles * (Cout / 128);
    double px = 0; for (int i = 0; i < nimg; ++i) px += (double)b.img[i].H * b.img[i].W;
    const double flop = 2.0 * 9 * Cin * Cout * px;
    printf("Cin %d Cout %d level0 %dx%d unpool %d relu %d: %d workgroups (%.2f per CU), kernel %.1f us median (min %.1f): %.0f TF algorithmic, %.0f TF executed f16 MFMA\n",
           Cin, Cout, H0, W0, unpool, relu, wgs, wgs / 256.0, ms[ms.size() / 2] * 1e3, ms[0] * 1e3, flop / ms[ms.size() / 2] / 1e9, 2 * flop / ms[ms.size() / 2] / 1e9);
#ifdef NST_WINO_STAMPS
    const int n = std::min(wgs, 1 << 14);
    std::vector<unsigned long long> s((size_t)(1 << 14) * 8);
    CK(hipMemcpyFromSymbol(s.data(), HIP_SYMBOL(nst::g_wino_stamps), s.size() * 8));
    const char* names[5] = {"prologue (first patch + weights -> LDS, barrier)", "K loop", "accumulators -> LDS + barrier", "output transform + stores issued", "stores drained"};
    std::vector<double> clk;
    for (int w = 0; w < n; ++w) {
        const double cyc = (double)(s[w * 8 + 5] - s[w * 8 + 0]), real = (double)(s[w * 8 + 7] - s[w * 8 + 6]);
        if (real > 0) clk.push_back(cyc / real * 100e6);
    }
    std::sort(clk.begin(), clk.end());
    const double f = clk[clk.size() / 2];
    printf("in-kernel clock (median over workgroups): %.0f MHz\n", f / 1e6);
    double total_med = 0;
    for (int ph = 0; ph < 5; ++ph) {
        std::vector<double> d;
        for (int w = 0; w < n; ++w) d.push_back((double)(s[w * 8 + ph + 1] - s[w * 8 + ph]));
        std::sort(d.begin(), d.end());
        printf("  %-52s median %8.0f cyc = %6.2f us   (p10 %8.0f, p90 %8.0f)\n", names[ph], d[n / 2], d[n / 2] / f * 1e6, d[n / 10], d[n * 9 / 10]);
        total_med += d[n / 2];
    }
    {
        std::vector<double> d;
        for (int w = 0; w < n; ++w) d.push_back((double)(s[w * 8 + 5] - s[w * 8 + 0]));
        std::sort(d.begin(), d.end());
        printf("  %-52s median %8.0f cyc = %6.2f us; K loop per chunk %.0f cyc (MFMA-bound floor 4608)\n", "whole workgroup", d[n / 2], d[n / 2] / f * 1e6, 0.0);
    }
    {
        std::vector<double> d;
        for (int w = 0; w < n; ++w) d.push_back((double)(s[w * 8 + 2] - s[w * 8 + 1]) / nch);
        std::sort(d.begin(), d.end());
        printf("  K loop per 32-channel chunk: median %.0f cyc (p10 %.0f, p90 %.0f); 72 MFMAs x 32 cyc x 2 waves per SIMD = 4608\n", d[n / 2], d[n / 10], d[n * 9 / 10]);
    }
    // how the workgroups of one launch spread over time: first start to last end, in real time
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < n; ++w) { t0 = std::min(t0, s[w * 8 + 6]); t1 = std::max(t1, s[w * 8 + 7]); }
    printf("  launch span by s_memrealtime: %.1f us\n", (double)(t1 - t0) / 100.0);
#endif
    return 0;
}
