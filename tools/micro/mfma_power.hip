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

// the same MFMA stream fed the way the conv kernel feeds it: per 24 MFMAs 16 ds_read_b128 fragments from a 144-byte-row
// LDS tile (conflict-free), no barriers, no global traffic
__global__ __launch_bounds__(512, 2) void mfma_lds_loop(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<f16x8*>(lds)[i] = ops[i];      // 64 KB
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = ((wave * 32 + (lane & 31)) * 144 + (lane >> 5) * 16) % (65536 - 8192);
    f32x16 accm[4], accx[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) { accm[c][r] = 0.f; accx[c][r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const int o = base + (it & 7) * 576;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 a[2][2], b[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    a[t][pc] = *reinterpret_cast<const f16x8*>(lds + o + t * 4608 + pc * 64 + ks * 32);
                    b[t][pc] = *reinterpret_cast<const f16x8*>(lds + o + 2304 + t * 4608 + pc * 64 + ks * 32);
                }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    accx[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][1], b[nt][0], accx[mt * 2 + nt], 0, 0, 0);
                    accm[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][0], b[nt][0], accm[mt * 2 + nt], 0, 0, 0);
                    accx[mt * 2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][0], b[nt][1], accx[mt * 2 + nt], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += accm[c][r] + accx[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- the same two loops on v_mfma_f32_16x16x32_f16: same FLOPs per fragment byte (a 64 x 64 wave tile per 32-deep K
// step = 4 A + 4 B fragments x 2 pieces = 16 ds_read_b128 for 16 blocks x 3 = 48 MFMAs), same accumulator count
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(512, 2) void mfma16_loop(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = ops[(threadIdx.x * 8 + i) % 4096];
        b[i] = ops[(threadIdx.x * 8 + 4 + i) % 4096];
    }
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k], b[(k + c) & 3], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512, 2) void mfma16_lds_loop(const f16x8* __restrict__ ops, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 4096; i += 512) reinterpret_cast<f16x8*>(lds)[i] = ops[i];      // 64 KB
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = ((wave * 16 + (lane & 15)) * 144 + (lane >> 4) * 16) % (65536 - 24576);
    f32x4 accm[16], accx[16];
    for (int c = 0; c < 16; ++c)
        for (int r = 0; r < 4; ++r) { accm[c][r] = 0.f; accx[c][r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const int o = base + (it & 7) * 576;
        f16x8 a[4][2], b[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                a[t][pc] = *reinterpret_cast<const f16x8*>(lds + o + t * 2304 + pc * 64);
                b[t][pc] = *reinterpret_cast<const f16x8*>(lds + o + 9216 + t * 2304 + pc * 64);
            }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                accx[mt * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][1], b[nt][0], accx[mt * 4 + nt], 0, 0, 0);
                accm[mt * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][0], b[nt][0], accm[mt * 4 + nt], 0, 0, 0);
                accx[mt * 4 + nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][0], b[nt][1], accx[mt * 4 + nt], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < 16; ++c)
        for (int r = 0; r < 4; ++r) s += accm[c][r] + accx[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
static double timed(K launch, double flops) {
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
static double run16(const f16x8* ops, float* out, int iters, int blocks) {
    return timed([&](int div) { hipLaunchKernelGGL(mfma16_loop<16>, dim3(blocks), dim3(512), 0, 0, ops, out, iters / div); },
                 (double)blocks * 8 * iters * 16 * 4 * 2.0 * 16 * 16 * 32);
}
static double run16_lds(const f16x8* ops, float* out, int iters, int blocks) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma16_lds_loop), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    return timed([&](int div) { hipLaunchKernelGGL(mfma16_lds_loop, dim3(blocks), dim3(512), 65536, 0, ops, out, iters / div); },
                 (double)blocks * 8 * iters * 48 * 2.0 * 16 * 16 * 32);
}

static double run_lds(const f16x8* ops, float* out, int iters, int blocks) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_lds_loop), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(mfma_lds_loop, dim3(blocks), dim3(512), 65536, 0, ops, out, iters / 8);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_lds_loop, dim3(blocks), dim3(512), 65536, 0, ops, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 24 * 2.0 * 32 * 32 * 16;
    return flops / (ms * 1e-3) / 1e12;
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
    const double l1 = run_lds(ops, out, iters * 2, 256);
    const double s1 = run16(ops, out, iters * 2, blocks);
    const double s2 = run16_lds(ops, out, iters * 2, 256);
    printf("v_mfma_f32_16x16x32_f16, random operands: from registers %.0f TFLOP/s; with the same fragment traffic per FLOP "
           "(16 ds_read_b128 per 48 MFMAs, 8 waves per CU) %.0f TFLOP/s\n", s1, s2);
    printf("with the conv kernel's fragment traffic (16 ds_read_b128 per 24 MFMAs, 8 waves per CU, random operands): %.0f TFLOP/s\n", l1);
    printf("v_mfma_f32_32x32x16_f16 from registers, 16 waves per CU: zero operands %.0f TFLOP/s, random operands %.0f TFLOP/s "
           "(short run) / %.0f TFLOP/s (4x longer run); nominal dense peak 2500\n", z, r1, r2);
    return 0;
}
