// tile_shapes.hip - which workgroup / wave-tile shape sustains the most f16x2-style MFMA work on this board when the
// operands come from LDS the way the conv kernel reads them (ds_read_b128 fragments, 144-byte rows, main + cross
// accumulators, ONE barrier per 32-deep K stage, next stage's fragments requested while the current one multiplies)?
//   A  512 threads (2 waves / SIMD), wave tile 64 x 64, v_mfma_f32_32x32x16_f16: 16 reads per 24 MFMAs   (the shipped shape)
//   B  512 threads (2 waves / SIMD), wave tile 64 x 64, v_mfma_f32_16x16x32_f16: 16 reads per 48 MFMAs
//   C  256 threads (1 wave / SIMD, 512 registers: accumulators in AGPRs), wave tile 64 x 128, 16x16x32: 24 reads per 96 MFMAs
//   D  256 threads (1 wave / SIMD), wave tile 64 x 128, 32x32x16: 24 reads per 48 MFMAs
// Random fp16 operands (the clock is power bound).  Prints TFLOP/s of executed f16 MFMA work.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/tile_shapes.hip -o gpurun_out/tile_shapes && gpurun_out/tile_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_BYTES = 65536;
constexpr int ROWB = 144;

__device__ __forceinline__ f16x8 ld(const unsigned char* lds, int off) {
    return *reinterpret_cast<const f16x8*>(lds + (off & (LDS_BYTES - 1 - 15)));
}

// ---- A: the shipped form
__global__ __launch_bounds__(512, 2) void shape_a(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 512) reinterpret_cast<f16x8*>(lds)[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (wave * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    f32x16 am[4], ax[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) { am[c][r] = 0.f; ax[c][r] = 0.f; }
    f16x8 a[2][2][2], b[2][2][2];      // [set][tile][piece]
    auto req = [&](int set, int o) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) { a[set][t][p] = ld(lds, o + t * 4608 + p * 64); b[set][t][p] = ld(lds, o + 2304 + t * 4608 + p * 64); }
    };
    auto mul = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                ax[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[set][mt][1], b[set][nt][0], ax[mt * 2 + nt], 0, 0, 0);
                am[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[set][mt][0], b[set][nt][0], am[mt * 2 + nt], 0, 0, 0);
                ax[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[set][mt][0], b[set][nt][1], ax[mt * 2 + nt], 0, 0, 0);
            }
    };
    req(0, base);
    for (int it = 0; it < iters; ++it) {
        const int o = base + (it & 7) * 576;
        req(1, o + 32);        // k-step 1 while k-step 0 multiplies
        mul(0);
        req(0, o + 576);       // next stage's k-step 0
        mul(1);
        __syncthreads();
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += am[c][r] + ax[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- B / C: 16x16x32, wave tile 64 x (16 NT): NT = 4 (512 threads) or 8 (256 threads)
template <int NT, int THREADS, int WAVES_PER_EU>
__global__ __launch_bounds__(THREADS, WAVES_PER_EU) void shape_16(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += THREADS) reinterpret_cast<f16x8*>(lds)[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (wave * 16 + (lane & 15)) * ROWB + (lane >> 4) * 16;
    f32x4 am[4][NT], ax[4][NT];
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < NT; ++n)
            for (int r = 0; r < 4; ++r) { am[m][n][r] = 0.f; ax[m][n][r] = 0.f; }
    f16x8 a[2][4][2], b[2][NT][2];
    auto req = [&](int set, int o) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[set][t][p] = ld(lds, o + t * 2304 + p * 64);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[set][t][p] = ld(lds, o + 9216 + t * 2304 + p * 64);
    };
    auto mul = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ax[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[set][mt][1], b[set][nt][0], ax[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) am[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[set][mt][0], b[set][nt][0], am[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ax[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[set][mt][0], b[set][nt][1], ax[mt][nt], 0, 0, 0);
        }
    };
    req(0, base);
    for (int it = 0; it < iters; it += 2) {
        const int o = base + (it & 7) * 576;
        req(1, o + 576);
        mul(0);
        __syncthreads();
        req(0, o + 1152);
        mul(1);
        __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m)
        for (int n = 0; n < NT; ++n)
            for (int r = 0; r < 4; ++r) s += am[m][n][r] + ax[m][n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- D: 256 threads, wave tile 64 x 128 on 32x32x16 (2 x 4 tiles, two k-steps per stage)
__global__ __launch_bounds__(256, 1) void shape_d(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 256) reinterpret_cast<f16x8*>(lds)[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (wave * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    f32x16 am[2][4], ax[2][4];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 4; ++n)
            for (int r = 0; r < 16; ++r) { am[m][n][r] = 0.f; ax[m][n][r] = 0.f; }
    f16x8 a[2][2][2], b[2][4][2];
    auto req = [&](int set, int o) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[set][t][p] = ld(lds, o + t * 4608 + p * 64);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[set][t][p] = ld(lds, o + 9216 + t * 4608 + p * 64);
    };
    auto mul = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ax[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[set][mt][1], b[set][nt][0], ax[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) am[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[set][mt][0], b[set][nt][0], am[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) ax[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[set][mt][0], b[set][nt][1], ax[mt][nt], 0, 0, 0);
        }
    };
    req(0, base);
    for (int it = 0; it < iters; ++it) {
        const int o = base + (it & 7) * 576;
        req(1, o + 32);
        mul(0);
        req(0, o + 576);
        mul(1);
        __syncthreads();
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 4; ++n)
            for (int r = 0; r < 16; ++r) s += am[m][n][r] + ax[m][n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename L>
static double timed(L launch, double flops) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(8);
    (void)hipEventRecord(e0, 0);
    launch(1);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    const int iters = 40000;
    std::vector<_Float16> h(LDS_BYTES / 2);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    f16x8* ops; float* out;
    (void)hipMalloc(&ops, LDS_BYTES);
    (void)hipMalloc(&out, (size_t)256 * 2 * 512 * 4);
    (void)hipMemcpy(ops, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_a), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_16<4, 512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_16<8, 256, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_d), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int rep = 0; rep < 2; ++rep) {
        // one workgroup per CU (the conv kernel's big shape holds 140 KB of LDS); LDS_BYTES * 2 > 160 KB / 2 keeps it so
        const int blocks = 256;
        const double fa = timed([&](int d) { hipLaunchKernelGGL(shape_a, dim3(blocks), dim3(512), LDS_BYTES + 32768, 0, ops, out, iters / d); },
                                (double)blocks * 8 * iters * 24 * 2.0 * 32 * 32 * 16);
        const double fb = timed([&](int d) { hipLaunchKernelGGL((shape_16<4, 512, 2>), dim3(blocks), dim3(512), LDS_BYTES + 32768, 0, ops, out, iters / d); },
                                (double)blocks * 8 * iters * 48 * 2.0 * 16 * 16 * 32);
        const double fc = timed([&](int d) { hipLaunchKernelGGL((shape_16<8, 256, 1>), dim3(blocks), dim3(256), LDS_BYTES + 32768, 0, ops, out, iters / d); },
                                (double)blocks * 4 * iters * 96 * 2.0 * 16 * 16 * 32);
        const double fd = timed([&](int d) { hipLaunchKernelGGL(shape_d, dim3(blocks), dim3(256), LDS_BYTES + 32768, 0, ops, out, iters / d); },
                                (double)blocks * 4 * iters * 48 * 2.0 * 32 * 32 * 16);
        printf("A 512 thr, 64x64 wave tile, 32x32x16: %.0f TFLOP/s | B 512 thr, 64x64, 16x16x32: %.0f | C 256 thr (1 wave/SIMD, AGPR acc), 64x128, 16x16x32: %.0f | "
               "D 256 thr, 64x128, 32x32x16: %.0f\n", fa, fb, fc, fd);
    }
    return 0;
}
