// vector_ops.hip - the optimiser's arithmetic over the n = 3*H0*W0 pixel floats: Adam
// (torch:optim/adam.py:457-546) and the dots / axpys of L-BFGS (torch:optim/lbfgs.py:396-488).
// All HBM-bound streaming kernels with 16-byte accesses; reductions are two-stage and ordered.
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int vblocks(size_t n4) {
    size_t b = (n4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ double vblock_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float vblock_max(float v, float* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return r;
}

// ---- dot ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          size_t n, double* __restrict__ scratch) {
    __shared__ double sh[4];
    const size_t n4 = n / 4;
    const f32x4* av = reinterpret_cast<const f32x4*>(a);
    const f32x4* bv = reinterpret_cast<const f32x4*>(b);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 x = av[i], y = bv[i];
        s0 += x[0] * y[0]; s1 += x[1] * y[1]; s2 += x[2] * y[2]; s3 += x[3] * y[3];
    }
    double s = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < n; ++i) s += (double)a[i] * (double)b[i];
    const double r = vblock_sum(s, sh);
    if (threadIdx.x == 0) scratch[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void sum_finish_kernel(const double* __restrict__ scratch, float* __restrict__ out) {
    __shared__ double sh[4];
    const double r = vblock_sum(threadIdx.x < RED_BLOCKS ? scratch[threadIdx.x] : 0.0, sh);
    if (threadIdx.x == 0) out[0] = (float)r;
}
hipError_t launch_dot(const float* a, const float* b, size_t n, double* scratch, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(dot_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, stream, a, b, n, scratch);
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, stream, scratch, out);
    return hipGetLastError();
}

// dot product left as RED_BLOCKS double partials in `scratch` (the consumer below finishes it)
hipError_t launch_dot_partial(const float* a, const float* b, size_t n, double* scratch, hipStream_t stream) {
    hipLaunchKernelGGL(dot_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, stream, a, b, n, scratch);
    return hipGetLastError();
}

// ---- one history pair of the L-BFGS two-loop recursion (torch:optim/lbfgs.py:444-460), no host round trip -------
// Finishes the dot product the previous launch left in `sin` (same order and rounding as sum_finish_kernel), forms the
// pair's coefficient, applies y += coef * x and, in the same pass, leaves the partials of the NEXT pair's dot
// product nxt . y in `sout` (same per-thread order as dot_partial_kernel, same grid).
//   first loop  (second = 0): v = dot * ro = al_i (stored to al[0]);  coef = -al_i         (q -= al_i * y_i)
//   second loop (second = 1): v = dot * ro = be_i;                     coef = al[0] - be_i  (r += (al_i - be_i) * s_i)
// With a host-side scalar per pair the two loops cost 2 * history stream synchronisations per optimiser step.
template <int NT>
__global__ __launch_bounds__(NT) void lbfgs_pair_kernel(const double* __restrict__ sin, float ro, float* __restrict__ al,
                                                         int second, const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ nxt, size_t n, double* __restrict__ sout) {
    __shared__ double sh[NT / 64];
    __shared__ float coef_sh;
    // block sum over NT / 64 waves, waves added in order
    auto block_sum = [&](double v) -> double {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        double t = 0.0;
        if (threadIdx.x == 0)
            for (int w = 0; w < NT / 64; ++w) t += sh[w];
        __syncthreads();
        return t;
    };
    const double r = block_sum(threadIdx.x < RED_BLOCKS ? sin[threadIdx.x] : 0.0);
    if (threadIdx.x == 0) {
        const float v = (float)r * ro;
        float c;
        if (second) {
            c = al[0] - v;
        } else {
            c = -v;
            if (blockIdx.x == 0) al[0] = v;
        }
        coef_sh = c;
    }
    __syncthreads();
    const float coef = coef_sh;
    const size_t n4 = n / 4;
    const f32x4* xv = reinterpret_cast<const f32x4*>(x);
    f32x4* yv = reinterpret_cast<f32x4*>(y);
    const f32x4* nv = reinterpret_cast<const f32x4*>(nxt);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four elements' loads in flight per lane (256 workgroups x 4 waves cannot cover HBM latency one load at a time);
    // the arithmetic keeps the one-element-at-a-time order, so the dot partials are those of dot_partial_kernel
    constexpr int U = (NT >= 1024) ? 2 : 4;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 a[U], b[U], c[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            a[k] = xv[i + k * stride];
            b[k] = yv[i + k * stride];
            if (nxt) c[k] = nv[i + k * stride];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            f32x4 r;
            r[0] = b[k][0] + coef * a[k][0]; r[1] = b[k][1] + coef * a[k][1];
            r[2] = b[k][2] + coef * a[k][2]; r[3] = b[k][3] + coef * a[k][3];
            yv[i + k * stride] = r;
            if (nxt) { s0 += c[k][0] * r[0]; s1 += c[k][1] * r[1]; s2 += c[k][2] * r[2]; s3 += c[k][3] * r[3]; }
        }
    }
    for (; i < n4; i += stride) {
        const f32x4 a = xv[i];
        f32x4 b = yv[i];
        b[0] = b[0] + coef * a[0]; b[1] = b[1] + coef * a[1]; b[2] = b[2] + coef * a[2]; b[3] = b[3] + coef * a[3];
        yv[i] = b;
        if (nxt) {
            const f32x4 c = nv[i];
            s0 += c[0] * b[0]; s1 += c[1] * b[1]; s2 += c[2] * b[2]; s3 += c[3] * b[3];
        }
    }
    double s = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = n4 * 4; i < n; ++i) {
            const float b = y[i] + coef * x[i];
            y[i] = b;
            if (nxt) s += (double)nxt[i] * (double)b;
        }
    if (nxt) {
        const double rr = block_sum(s);
        if (threadIdx.x == 0) sout[blockIdx.x] = rr;
    }
}
hipError_t launch_lbfgs_pair(const double* sin, float ro, float* al, int second, const float* x, float* y, const float* nxt,
                             size_t n, double* sout, hipStream_t stream) {
    constexpr int NT = 1024;     // 16 waves per CU: this pass is pure streaming and 4 waves per CU left it latency-bound
    hipLaunchKernelGGL(lbfgs_pair_kernel<NT>, dim3(RED_BLOCKS), dim3(NT), 0, stream, sin, ro, al, second, x, y, nxt, n, sout);
    return hipGetLastError();
}

// ---- L-BFGS direction from inner products (default; NST_LBFGS_GRAM=0 = the pair kernel above): every history vector is read once for all the
// dot products the recursion needs of it, and once more for the direction ----------------------------------------
// Workgroup b owns the float4 indices [1024 b, 1024 b + 1024): its slices of the three operand vectors stay in
// registers while it walks the history; per history vector it leaves three double partials.
constexpr int MD_F4 = 4;
__global__ __launch_bounds__(256) void multi_dot_kernel(const float* const* __restrict__ vecs, int nvec,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ c, size_t n, double* __restrict__ scratch) {
    __shared__ double sh[4][3];
    const size_t n4 = n / 4;
    const size_t base = (size_t)blockIdx.x * (256 * MD_F4) + threadIdx.x;
    f32x4 ra[MD_F4], rb[MD_F4], rc[MD_F4];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MD_F4; ++k) {
        const size_t i = base + (size_t)k * 256;
        const bool ok = i < n4;
        ra[k] = ok ? reinterpret_cast<const f32x4*>(a)[i] : z;
        rb[k] = ok ? reinterpret_cast<const f32x4*>(b)[i] : z;
        rc[k] = ok ? reinterpret_cast<const f32x4*>(c)[i] : z;
    }
    const bool tail = (blockIdx.x == 0 && threadIdx.x == 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < nvec; ++j) {
        const float* __restrict__ v = vecs[j];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MD_F4; ++k) {
            const size_t i = base + (size_t)k * 256;
            const f32x4 x = (i < n4) ? reinterpret_cast<const f32x4*>(v)[i] : z;
#pragma unroll
            for (int e = 0; e < 4; ++e) { s0 += x[e] * ra[k][e]; s1 += x[e] * rb[k][e]; s2 += x[e] * rc[k][e]; }
        }
        double d0 = s0, d1 = s1, d2 = s2;
        if (tail)
            for (size_t i = n4 * 4; i < n; ++i) {
                d0 += (double)v[i] * (double)a[i]; d1 += (double)v[i] * (double)b[i]; d2 += (double)v[i] * (double)c[i];
            }
        for (int off = 32; off > 0; off >>= 1) {
            d0 += __shfl_down(d0, off, 64); d1 += __shfl_down(d1, off, 64); d2 += __shfl_down(d2, off, 64);
        }
        if (lane == 0) { sh[wave][0] = d0; sh[wave][1] = d1; sh[wave][2] = d2; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int t = threadIdx.x;
            scratch[((size_t)blockIdx.x * nvec + j) * 3 + t] = ((sh[0][t] + sh[1][t]) + sh[2][t]) + sh[3][t];
        }
        __syncthreads();
    }
}
// out[j*3 + t] = sum over the workgroups' partials, in workgroup order
__global__ __launch_bounds__(256) void multi_dot_finish_kernel(const double* __restrict__ scratch, int blocks, int nvec,
                                                               float* __restrict__ out) {
    __shared__ double sh[4];
    const int o = blockIdx.x;               // = j * 3 + t
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) s += scratch[(size_t)b * nvec * 3 + o];
    const double r = vblock_sum(s, sh);
    if (threadIdx.x == 0) out[o] = (float)r;
}
int multi_dot_blocks(size_t n) { return (int)((n / 4 + 256 * MD_F4 - 1) / (256 * MD_F4)) > 0 ? (int)((n / 4 + 256 * MD_F4 - 1) / (256 * MD_F4)) : 1; }
hipError_t launch_multi_dot(const float* const* vecs_dev, int nvec, const float* a, const float* b, const float* c, size_t n,
                            double* scratch, float* out, hipStream_t stream) {
    const int blocks = multi_dot_blocks(n);
    hipLaunchKernelGGL(multi_dot_kernel, dim3(blocks), dim3(256), 0, stream, vecs_dev, nvec, a, b, c, n, scratch);
    hipLaunchKernelGGL(multi_dot_finish_kernel, dim3(nvec * 3), dim3(256), 0, stream, scratch, blocks, nvec, out);
    return hipGetLastError();
}
// d = h * q0 + sum_j coef[j] * vecs[j]  (terms added in the order j = 0, 1, ...)
__global__ __launch_bounds__(256) void multi_axpy_kernel(const float* const* __restrict__ vecs, const float* __restrict__ coef,
                                                         int nvec, const float* __restrict__ q0, float h, float* __restrict__ d,
                                                         size_t n) {
    const size_t n4 = n / 4;
    const size_t base = (size_t)blockIdx.x * (256 * MD_F4) + threadIdx.x;
    f32x4 acc[MD_F4];
#pragma unroll
    for (int k = 0; k < MD_F4; ++k) {
        const size_t i = base + (size_t)k * 256;
        const f32x4 q = (i < n4) ? reinterpret_cast<const f32x4*>(q0)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[k] = q * h;
    }
    const bool tail = (blockIdx.x == 0 && threadIdx.x == 0);
    float tacc[3] = {0.f, 0.f, 0.f};
    if (tail)
        for (size_t i = n4 * 4; i < n; ++i) tacc[i - n4 * 4] = h * q0[i];
    for (int j = 0; j < nvec; ++j) {
        const float cj = coef[j];
        const float* __restrict__ v = vecs[j];
#pragma unroll
        for (int k = 0; k < MD_F4; ++k) {
            const size_t i = base + (size_t)k * 256;
            if (i < n4) {
                const f32x4 x = reinterpret_cast<const f32x4*>(v)[i];
                acc[k] = acc[k] + x * cj;
            }
        }
        if (tail)
            for (size_t i = n4 * 4; i < n; ++i) tacc[i - n4 * 4] += cj * v[i];
    }
#pragma unroll
    for (int k = 0; k < MD_F4; ++k) {
        const size_t i = base + (size_t)k * 256;
        if (i < n4) reinterpret_cast<f32x4*>(d)[i] = acc[k];
    }
    if (tail)
        for (size_t i = n4 * 4; i < n; ++i) d[i] = tacc[i - n4 * 4];
}
hipError_t launch_multi_axpy(const float* const* vecs_dev, const float* coef_dev, int nvec, const float* q0, float h, float* d,
                             size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(multi_axpy_kernel, dim3(multi_dot_blocks(n)), dim3(256), 0, stream, vecs_dev, coef_dev, nvec, q0, h, d, n);
    return hipGetLastError();
}

// ---- max|a| and sum|a| ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void abs_partial_kernel(const float* __restrict__ a, size_t n,
                                                          double* __restrict__ scratch) {
    __shared__ double sh[4];
    __shared__ float shm[4];
    float mx = 0.f, s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = fabsf(a[i]);
        mx = fmaxf(mx, v);
        s += v;
    }
    const double rs = vblock_sum((double)s, sh);
    const float rm = vblock_max(mx, shm);
    if (threadIdx.x == 0) {
        scratch[blockIdx.x] = rs;
        scratch[RED_BLOCKS + blockIdx.x] = (double)rm;
    }
}
__global__ __launch_bounds__(256) void abs_finish_kernel(const double* __restrict__ scratch, float* __restrict__ out) {
    __shared__ double sh[4];
    __shared__ float shm[4];
    const double rs = vblock_sum(threadIdx.x < RED_BLOCKS ? scratch[threadIdx.x] : 0.0, sh);
    const float rm = vblock_max(threadIdx.x < RED_BLOCKS ? (float)scratch[RED_BLOCKS + threadIdx.x] : 0.f, shm);
    if (threadIdx.x == 0) { out[0] = rm; out[1] = (float)rs; }
}
hipError_t launch_absmax_abssum(const float* a, size_t n, double* scratch, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(abs_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, stream, a, n, scratch);
    hipLaunchKernelGGL(abs_finish_kernel, dim3(1), dim3(256), 0, stream, scratch, out);
    return hipGetLastError();
}

// ---- axpy family ----------------------------------------------------------------------------------
__global__ void axpy_kernel(float alpha, const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = y[i] + alpha * x[i];
}
__global__ void axpy_dev_kernel(const float* __restrict__ alpha_dev, float sign, const float* __restrict__ x,
                                float* __restrict__ y, size_t n) {
    const float alpha = sign * alpha_dev[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = y[i] + alpha * x[i];
}
__global__ void scale_copy_kernel(float alpha, const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = alpha * x[i];
}
__global__ void sub_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] - b[i];
}
// out = a + alpha * b (same expression as axpy_kernel, so "copy then axpy" and this give identical floats) and a
// plain copy, 16 bytes per lane: the optimiser's device-to-device moves (hipMemcpyAsync D2D runs a blit kernel at
// about a quarter of this rate)
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ void add_scaled_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b,
                                  float* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4v va = reinterpret_cast<const f32x4v*>(a)[i], vb = reinterpret_cast<const f32x4v*>(b)[i];
        f32x4v r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = va[k] + alpha * vb[k];
        reinterpret_cast<f32x4v*>(out)[i] = r;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        out[i] = a[i] + alpha * b[i];
    }
}
__global__ void copy_kernel(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4v*>(out)[i] = reinterpret_cast<const f32x4v*>(a)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = a[n4 * 4 + threadIdx.x];
}
// zero fill (n 32-bit words, 16-byte aligned base): a kernel, not hipMemsetAsync - the memset nodes of a captured closure
// were not reliably ordered against the kernels around them in a hipGraph replay, and a kernel is no slower eagerly
__global__ void zero_kernel(float* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    const f32x4v z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4v*>(out)[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = 0.f;
}
hipError_t launch_zero(void* out, size_t n_words, hipStream_t stream) {
    hipLaunchKernelGGL(zero_kernel, dim3(vblocks(n_words / 4 + 1)), dim3(256), 0, stream, static_cast<float*>(out), n_words);
    return hipGetLastError();
}
hipError_t launch_add_scaled(const float* a, float alpha, const float* b, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(add_scaled_kernel, dim3(vblocks(n / 4 + 1)), dim3(256), 0, stream, a, alpha, b, out, n);
    return hipGetLastError();
}
hipError_t launch_copy(const float* a, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(copy_kernel, dim3(vblocks(n / 4 + 1)), dim3(256), 0, stream, a, out, n);
    return hipGetLastError();
}
hipError_t launch_axpy(float alpha, const float* x, float* y, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(axpy_kernel, dim3(vblocks(n)), dim3(256), 0, stream, alpha, x, y, n);
    return hipGetLastError();
}
hipError_t launch_axpy_dev(const float* alpha_dev, float sign, const float* x, float* y, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(axpy_dev_kernel, dim3(vblocks(n)), dim3(256), 0, stream, alpha_dev, sign, x, y, n);
    return hipGetLastError();
}
hipError_t launch_scale_copy(float alpha, const float* x, float* y, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(scale_copy_kernel, dim3(vblocks(n)), dim3(256), 0, stream, alpha, x, y, n);
    return hipGetLastError();
}
hipError_t launch_sub(const float* a, const float* b, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(sub_kernel, dim3(vblocks(n)), dim3(256), 0, stream, a, b, out, n);
    return hipGetLastError();
}

// ---- Adam (single tensor, no amsgrad / weight decay) -----------------------------------------------
// torch's CPU kernels round like this (checked bit for bit against torch 2.10 on 2^20 random elements): lerp_ is
// fma(w, g - m, m); addcmul_ is fma((1-b2) g, g, v b2); the denominator is a true division plus eps; addcdiv_ is
// x + (value * m) / denom.  w and 1-b2 arrive as floats rounded from the DOUBLE differences 1 - 0.9 and 1 - 0.999
// (0.1f and 0.001f), which is what torch passes - 1.f - 0.999f is 1.3e-5 off.
__global__ void adam_kernel(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float beta2, float one_m_b1, float one_m_b2, float eps,
                            float neg_step_size, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = __fmaf_rn(one_m_b1, __fsub_rn(gi, mi), mi);                         // exp_avg.lerp_(grad, 1 - beta1)
        vi = __fmaf_rn(__fmul_rn(one_m_b2, gi), gi, __fmul_rn(vi, beta2));       // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
        x[i] = __fadd_rn(x[i], __fdiv_rn(__fmul_rn(neg_step_size, mi), denom));  // addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mi;
        v[i] = vi;
    }
}
hipError_t launch_adam(float* x, const float* g, float* m, float* v, size_t n, float beta2, float one_m_b1, float one_m_b2,
                       float eps, float step_size, float bc2_sqrt, hipStream_t stream) {
    hipLaunchKernelGGL(adam_kernel, dim3(vblocks(n)), dim3(256), 0, stream, x, g, m, v, n, beta2, one_m_b1, one_m_b2, eps,
                       -step_size, bc2_sqrt);
    return hipGetLastError();
}

}  // namespace nst
