// mfma_power.hip - what the fp16 matrix pipe of this board delivers when it does NOTHING but
// v_mfma_f32_32x32x16_f16 from registers: all-zero operands vs random operands (random data toggles the
// multipliers, draws more power, and the chip lowers its clock).  This is the practical ceiling the conv kernel's
// 0.44 of the nominal 2.5 PFLOP/s has to be read against.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o gpurun_out/mfma_power && gpurun_out/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512, 2) void mfma_loop(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    // 8 distinct A / B fragments per lane, CHAINS independent accumulators
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = ops[(threadIdx.x * 8 + i) % 4096];
        b[i] = ops[(threadIdx.x * 8 + 4 + i) % 4096];
    }
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[(k + c) & 3], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double run(const f16x8* ops, float* out, int iters, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(512), 0, 0, ops, out, iters / 8);      // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<8>, dim3(blocks), dim3(512), 0, 0, ops, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 /*waves*/ * iters * 8 /*chains*/ * 4 * 2.0 * 32 * 32 * 16;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    const int blocks = 256 * 2, iters = 20000;
    std::vector<_Float16> h(4096 * 8);
    f16x8* ops; float* out;
    (void)hipMalloc(&ops, h.size() * 2);
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    for (auto& v : h) v = (_Float16)0.f;
    (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const double z = run(ops, out, iters, blocks);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
    (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const double r1 = run(ops, out, iters, blocks);
    const double r2 = run(ops, out, iters * 4, blocks);       // ~4x longer: the sustained (thermal / power) figure
    printf("v_mfma_f32_32x32x16_f16 from registers, 16 waves per CU: zero operands %.0f TFLOP/s, random operands %.0f TFLOP/s "
           "(short run) / %.0f TFLOP/s (4x longer run); nominal dense peak 2500\n", z, r1, r2);
    return 0;
}
