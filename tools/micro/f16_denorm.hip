// f16_denorm.hip - does the fp16 matrix pipe honour SUBNORMAL fp16 operands, and does v_cvt produce them?
// (the single-accumulator form of the f16x2 split keeps the low piece unscaled: residuals of small elements are
// fp16 subnormals)   hipcc -O3 --offload-arch=gfx950 tools/micro/f16_denorm.hip -o /tmp/f16d && /tmp/f16d
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x;
    // A[row][k]: every element = (fp16)in[0] (a value that is subnormal in fp16); B = all ones (x in[1])
    const _Float16 a = (_Float16)in[0], b = (_Float16)in[1];
    f16x8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = a; bv[j] = b; }
    f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
    c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c4, 0, 0, 0);
    f32x16 c16;
    for (int r = 0; r < 16; ++r) c16[r] = 0.f;
    c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c16, 0, 0, 0);
    if (lane == 0) { out[0] = (float)a; out[1] = c4[0]; out[2] = c16[0]; }
}
int main() {
    float *in, *out, h[2], r[3];
    (void)hipMalloc(&in, 8); (void)hipMalloc(&out, 12);
    const float vals[4] = {1.0f / 1048576.f /* 2^-20 */, 3.0f / 16777216.f /* 3 * 2^-24 */, 1.0f / 32768.f /* 2^-15 */, 1e-9f};
    for (int t = 0; t < 4; ++t) {
        h[0] = vals[t]; h[1] = 1024.f;
        (void)hipMemcpy(in, h, 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, in, out);
        (void)hipMemcpy(r, out, 12, hipMemcpyDeviceToHost);
        printf("x = %.9g: fp16(x) = %.9g (expected %.9g); 16x16x32 sum of 32 x * 1024 = %.9g, 32x32x16 sum of 16 = %.9g; exact %.9g / %.9g\n",
               vals[t], r[0], (double)(float)(_Float16)vals[t], r[1], r[2], 32.0 * (double)(float)(_Float16)vals[t] * 1024, 16.0 * (double)(float)(_Float16)vals[t] * 1024);
    }
    return 0;
}
