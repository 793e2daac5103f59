// winograd_stream.hip - can the operand stream of a 1-D Winograd F(2,3) convolution be fed?
//
// In that form every wave of a 512-thread workgroup owns ONE of the four transformed-tap indices xi: its A fragments
// (transformed patch rows of its xi) come from LDS as today, but its B fragments belong to the weight slice (ky, xi) that
// no other wave of the workgroup shares with it beyond the second output-channel half - there is nothing for LDS to
// amortise, and four slices per stage instead of one do not fit beside the patch.  So B goes global (L2) -> registers in
// MFMA fragment order: 8 KB per wave and stage, 64 KB per workgroup and stage against 16 KB through LDS today.
//   A  the shipped stream: both operands from LDS (as tools/micro/tile_shapes.hip, shape A)
//   W  A fragments from LDS, B fragments by global_load_dwordx4 from a 12.6 MB weight image (4 output-channel tiles x 16
//      chunks x 3 ky stages x 64 KB, every workgroup walking the same sequence), loaded one stage ahead
//   W0 the same with every workgroup and stage reading the SAME 64 KB (L2-hit upper bound of the load path)
//   D  the direct convolution's sharing: the four waves along the pixel dimension read the same fragments (16 KB per stage,
//      9.4 MB image): the shipped kernel with its weight slices fetched L2/L1 -> registers instead of staged through LDS
// Random fp16 operands.  Prints TFLOP/s of executed f16 MFMA work and the B bytes per second of W.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/winograd_stream.hip -o /tmp/ws && /tmp/ws
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_BYTES = 65536;
constexpr int ROWB = 144;
constexpr int STAGES = 48;                      // 16 chunks x 3 ky
constexpr size_t W_UNITS = (size_t)4 * STAGES * 8 * 8 * 64;      // 16-byte units of the weight image

__device__ __forceinline__ f16x8 ld(const unsigned char* lds, int off) {
    return *reinterpret_cast<const f16x8*>(lds + (off & (LDS_BYTES - 1 - 15)));
}

__global__ __launch_bounds__(512, 2) void shape_a(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 512) reinterpret_cast<f16x8*>(lds)[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (wave * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    f32x16 am[4], ax[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) { am[c][r] = 0.f; ax[c][r] = 0.f; }
    f16x8 a[2][2][2], b[2][2][2];
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
        req(1, o + 32);
        mul(0);
        req(0, o + 576);
        mul(1);
        __syncthreads();
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += am[c][r] + ax[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// W: B fragments straight from the weight image.  MODE 0: every wave its own slice (Winograd: 48 stages x 64 KB per output-channel tile);
// MODE 1: every stage reads stage 0 of tile 0; MODE 2: the DIRECT convolution's sharing - the four waves along the pixel dimension read
// the SAME fragments (144 stages x 16 KB per output-channel tile = the 9.4 MB of a 512 -> 512 layer), i.e. the shipped kernel with its
// weight slices fetched L2/L1 -> registers instead of staged through LDS.
template <int MODE>
__global__ __launch_bounds__(512, 2) void shape_w(const f16x8* __restrict__ ops, const f16x8* __restrict__ wimg, float* __restrict__ out,
                                                 int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 512) reinterpret_cast<f16x8*>(lds)[i] = ops[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = (wave * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    constexpr bool SAME = (MODE == 1);
    constexpr int NST = (MODE == 2) ? 144 : STAGES;
    constexpr int SLOTS = (MODE == 2) ? 2 : 8;
    const int slot = (MODE == 2) ? (wave >> 2) : wave;
    const int ct = SAME ? 0 : (int)(blockIdx.x & 3);
    f32x16 am[4], ax[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) { am[c][r] = 0.f; ax[c][r] = 0.f; }
    f16x8 a[2][2][2];            // [k-step set][m tile][piece], from LDS
    f16x8 b[2][2][2][2];         // [stage set][k-step][n tile][piece], from the weight image
    auto req_a = [&](int set, int o) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[set][t][p] = ld(lds, o + t * 4608 + p * 64);
    };
    auto req_b = [&](int set, int stage) {
        const size_t u0 = (((size_t)(ct * NST + (SAME ? 0 : stage)) * SLOTS + slot) * 8) * 64 + lane;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) b[set][k][t][p] = wimg[u0 + (size_t)((k * 2 + t) * 2 + p) * 64];
    };
    auto mul = [&](int aset, int bset, int k) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                ax[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[aset][mt][1], b[bset][k][nt][0], ax[mt * 2 + nt], 0, 0, 0);
                am[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[aset][mt][0], b[bset][k][nt][0], am[mt * 2 + nt], 0, 0, 0);
                ax[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[aset][mt][0], b[bset][k][nt][1], ax[mt * 2 + nt], 0, 0, 0);
            }
    };
    req_a(0, base);
    req_b(0, 0);
    int stage = 0;
    for (int it = 0; it < iters; it += 2) {
        // two stages per trip so that the B register sets are compile-time constants
        {
            const int o = base + (it & 7) * 576;
            const int nx = (stage + 1 == NST) ? 0 : stage + 1;
            req_b(1, nx);                  // next stage's weights, a whole stage ahead
            req_a(1, o + 32);
            mul(0, 0, 0);
            req_a(0, o + 576);
            mul(1, 0, 1);
            __syncthreads();
            stage = nx;
        }
        {
            const int o = base + ((it + 1) & 7) * 576;
            const int nx = (stage + 1 == NST) ? 0 : stage + 1;
            req_b(0, nx);
            req_a(1, o + 32);
            mul(0, 1, 0);
            req_a(0, o + 576);
            mul(1, 1, 1);
            __syncthreads();
            stage = nx;
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += am[c][r] + ax[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename L>
static double timed(L launch, double flops, double* ms_out = nullptr) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(8);
    (void)hipEventRecord(e0, 0);
    launch(1);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    const int iters = 40000;
    std::vector<_Float16> h(LDS_BYTES / 2), hw(W_UNITS * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    f16x8 *ops, *wimg; float* out;
    (void)hipMalloc(&ops, LDS_BYTES);
    (void)hipMalloc(&wimg, W_UNITS * 16);
    (void)hipMalloc(&out, (size_t)256 * 2 * 512 * 4);
    (void)hipMemcpy(ops, h.data(), LDS_BYTES, hipMemcpyHostToDevice);
    (void)hipMemcpy(wimg, hw.data(), W_UNITS * 16, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_a), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 32768);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_w<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 32768);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_w<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 32768);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shape_w<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 32768);
    const int blocks = 256;
    const double flops = (double)blocks * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        double msw = 0.0;
        const double fa = timed([&](int d) { hipLaunchKernelGGL(shape_a, dim3(blocks), dim3(512), LDS_BYTES + 32768, 0, ops, out, iters / d); }, flops);
        const double fw = timed([&](int d) { hipLaunchKernelGGL(shape_w<0>, dim3(blocks), dim3(512), LDS_BYTES + 32768, 0, ops, wimg, out, iters / d); },
                                flops, &msw);
        const double f0 = timed([&](int d) { hipLaunchKernelGGL(shape_w<1>, dim3(blocks), dim3(512), LDS_BYTES + 32768, 0, ops, wimg, out, iters / d); }, flops);
        const double f2 = timed([&](int d) { hipLaunchKernelGGL(shape_w<2>, dim3(blocks), dim3(512), LDS_BYTES + 32768, 0, ops, wimg, out, iters / d); }, flops);
        printf("A both operands from LDS: %.0f TFLOP/s | W B fragments from a 12.6 MB image in L2: %.0f TFLOP/s (%.1f TB/s of B) | "
               "W0 B fragments all from one 64 KB block: %.0f TFLOP/s | D direct-conv sharing (4 waves read the same fragments, 9.4 MB image): %.0f TFLOP/s\n",
               fa, fw, (double)blocks * iters * 65536.0 / (msw * 1e-3) / 1e12, f0, f2);
    }
    return 0;
}
