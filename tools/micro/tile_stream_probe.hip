// tile_stream_probe.hip - the memory floor of the conv1_1 kernels' access patterns (conv_first.hip), without their arithmetic.
//   read : a persistent workgroup of 256 threads walks 16x16-pixel tiles of an (H, W, 64) fp32 tensor and loads each tile's
//          18x18 halo patch in two 32-channel slices (11 x 16 B per thread and slice, all issued, then all consumed) - the
//          input gradient's staging
//   write: the same walk storing a tile's 16x16x64 values (64 x 4 B per thread in the forward's 32x32-accumulator layout,
//          or 16 x 16 B per thread)
// for a number of workgroups per CU (grid = wgs_per_cu * 256; registers and LDS of the probe allow 8).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/tile_stream_probe.hip -o /tmp/tsp && /tmp/tsp 1024 1536
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int T = 16, P = 18, UNITS = P * P * 8, PER_T = (UNITS + 255) / 256;

template <int SLICES_IN_FLIGHT>
__global__ __launch_bounds__(256) void read_tiles(const float* __restrict__ g, int H, int W, float* __restrict__ sink, int ntiles) {
    const int tid = threadIdx.x, tiles_x = (W + T - 1) / T;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int y0 = (tile / tiles_x) * T, x0 = (tile % tiles_x) * T;
        f32x4 st[SLICES_IN_FLIGHT][PER_T];
#pragma unroll
        for (int c0 = 0; c0 < 2; c0 += SLICES_IN_FLIGHT) {
#pragma unroll
            for (int c = 0; c < SLICES_IN_FLIGHT; ++c)
#pragma unroll
                for (int i = 0; i < PER_T; ++i) {
                    const int u = tid + i * 256, pix = u / 8, q = u % 8;
                    const int gy = y0 - 1 + pix / P, gx = x0 - 1 + pix % P;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (u < UNITS && gy >= 0 && gy < H && gx >= 0 && gx < W)
                        v = *reinterpret_cast<const f32x4*>(g + ((size_t)gy * W + gx) * 64 + (c0 + c) * 32 + q * 4);
                    st[c][i] = v;
                }
#pragma unroll
            for (int c = 0; c < SLICES_IN_FLIGHT; ++c)
#pragma unroll
                for (int i = 0; i < PER_T; ++i) acc += st[c][i];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[blockIdx.x * 256 + tid] = acc[0];
}

// WIDE = 0: 64 stores of 4 B per thread (lane l31 = channel, register r = pixel: the 32x32 accumulator layout);
// WIDE = 1: 16 stores of 16 B per thread (lane = (pixel, channel quad))
template <int WIDE>
__global__ __launch_bounds__(256) void write_tiles(float* __restrict__ out, int H, int W, int ntiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tiles_x = (W + T - 1) / T;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int y0 = (tile / tiles_x) * T, x0 = (tile % tiles_x) * T;
        if (y0 + T > H || x0 + T > W) continue;
        if (WIDE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int u = tid + i * 256, pix = u / 16, q = u % 16;
                *reinterpret_cast<f32x4*>(out + ((size_t)(y0 + pix / T) * W + x0 + pix % T) * 64 + q * 4) = f32x4{1.f, 2.f, 3.f, (float)tile};
            }
        } else {
            const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wave * 4 + (r >> 3) + 2 * mt, col = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
                        out[((size_t)(y0 + row) * W + x0 + col) * 64 + nt * 32 + l31] = (float)tile;
                    }
        }
    }
}

template <class F>
static double median_us(F&& launch, hipStream_t s) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 15; ++it) {
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 3) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1e3;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 1024, W = argc > 2 ? atoi(argv[2]) : 1536;
    const size_t n = (size_t)H * W * 64;
    const int ntiles = ((H + T - 1) / T) * ((W + T - 1) / T);
    float *g, *sink;
    CK(hipMalloc(&g, n * 4)); CK(hipMemset(g, 0, n * 4));
    CK(hipMalloc(&sink, 8 * 256 * 256 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("%d x %d x 64 fp32 = %.0f MB, %d tiles\n", H, W, n * 4 / 1e6, ntiles);
    for (int per_cu : {1, 2, 3, 4, 6, 8}) {
        const int grid = std::min(ntiles, per_cu * 256);
        const double r1 = median_us([&] { hipLaunchKernelGGL(read_tiles<1>, dim3(grid), dim3(256), 0, s, g, H, W, sink, ntiles); }, s);
        const double r2 = median_us([&] { hipLaunchKernelGGL(read_tiles<2>, dim3(grid), dim3(256), 0, s, g, H, W, sink, ntiles); }, s);
        const double w0 = median_us([&] { hipLaunchKernelGGL(write_tiles<0>, dim3(grid), dim3(256), 0, s, g, H, W, ntiles); }, s);
        const double w1 = median_us([&] { hipLaunchKernelGGL(write_tiles<1>, dim3(grid), dim3(256), 0, s, g, H, W, ntiles); }, s);
        printf("%d workgroups/CU: read (one slice in flight) %6.1f us %.2f TB/s | read (two) %6.1f us %.2f TB/s | write 4-B %6.1f us %.2f TB/s | write 16-B %6.1f us %.2f TB/s\n",
               per_cu, r1, n * 4 / r1 / 1e6, r2, n * 4 / r2 / 1e6, w0, n * 4 / w0 / 1e6, w1, n * 4 / w1 / 1e6);
    }
    return 0;
}
