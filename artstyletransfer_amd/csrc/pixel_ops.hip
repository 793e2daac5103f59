// pixel_ops.hip - the HBM-bound kernels of the closure: max-pool forward / backward(+ReLU mask),
// bicubic pyramid down-sample and its transpose, total variation, content MSE, image
// prepare/unprepare, layout changes and the loss assembly.  All are streaming kernels: 16-byte
// accesses where the layout allows, grid capped and grid-strided, reductions two-stage and
// ordered (no float atomics) so that a closure is bitwise reproducible run to run.
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int cap_blocks(size_t work, int threads) {
    size_t b = (work + threads - 1) / threads;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ double block_reduce_sum(double v, double* sh) {
    // 256 threads; fixed tree order
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += sh[i];
    }
    __syncthreads();
    return r;   // valid in thread 0
}

// ------------------------------------------------------------------ max pool (torchvision features[4,9,18,27])
__global__ void maxpool_fwd_kernel(const float* __restrict__ in, int H, int W, int C4, float* __restrict__ out) {
    const int oh = H >> 1, ow = W >> 1;
    const size_t total = (size_t)oh * ow * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t pix = i / C4;
        const int ox = (int)(pix % ow);
        const int oy = (int)(pix / ow);
        const f32x4* src = reinterpret_cast<const f32x4*>(in);
        const size_t r0 = ((size_t)(2 * oy) * W + 2 * ox) * C4 + c;
        const size_t r1 = r0 + (size_t)W * C4;
        const f32x4 a = src[r0], b = src[r0 + C4], d = src[r1], e = src[r1 + C4];
        f32x4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(d[k], e[k]));
        reinterpret_cast<f32x4*>(out)[i] = m;
    }
}

hipError_t launch_maxpool_fwd(const float* in, int H, int W, int C, float* out, hipStream_t stream) {
    const size_t total = (size_t)(H / 2) * (W / 2) * (C / 4);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(cap_blocks(total, 256)), dim3(256), 0, stream, in, H, W, C / 4, out);
    return hipGetLastError();
}

// backward of max_pool2d (first maximum in window scan order wins, as ATen's CPU/GPU kernels do)
// fused with the ReLU mask of the pooled activation.
__global__ void maxpool_bwd_relu_kernel(const float* __restrict__ a, const float* __restrict__ gpool, int H, int W,
                                        int C4, float* __restrict__ gin) {
    const int oh = H >> 1, ow = W >> 1;
    const int wh = (H + 1) >> 1, ww = (W + 1) >> 1;
    const size_t total = (size_t)wh * ww * C4;
    const f32x4* av = reinterpret_cast<const f32x4*>(a);
    const f32x4* gv = reinterpret_cast<const f32x4*>(gpool);
    f32x4* ov = reinterpret_cast<f32x4*>(gin);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t pix = i / C4;
        const int wx = (int)(pix % ww);
        const int wy = (int)(pix / ww);
        const int y = 2 * wy, x = 2 * wx;
        const size_t r0 = ((size_t)y * W + x) * C4 + c;
        const size_t r1 = r0 + (size_t)W * C4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (wy < oh && wx < ow) {
            const f32x4 v0 = av[r0], v1 = av[r0 + C4], v2 = av[r1], v3 = av[r1 + C4];
            const f32x4 g = gv[((size_t)wy * ow + wx) * C4 + c];
            f32x4 o0, o1, o2, o3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int arg = 0;
                float m = v0[k];
                if (v1[k] > m) { m = v1[k]; arg = 1; }
                if (v2[k] > m) { m = v2[k]; arg = 2; }
                if (v3[k] > m) { m = v3[k]; arg = 3; }
                const float gg = (m > 0.f) ? g[k] : 0.f;
                o0[k] = (arg == 0) ? gg : 0.f;
                o1[k] = (arg == 1) ? gg : 0.f;
                o2[k] = (arg == 2) ? gg : 0.f;
                o3[k] = (arg == 3) ? gg : 0.f;
            }
            ov[r0] = o0; ov[r0 + C4] = o1; ov[r1] = o2; ov[r1 + C4] = o3;
        } else {
            // odd border: pixels not covered by any window get no gradient
            ov[r0] = zero;
            if (x + 1 < W) ov[r0 + C4] = zero;
            if (y + 1 < H) {
                ov[r1] = zero;
                if (x + 1 < W) ov[r1 + C4] = zero;
            }
        }
    }
}

hipError_t launch_maxpool_bwd_relu(const float* a, const float* gpool, int H, int W, int C, float* gin,
                                   hipStream_t stream) {
    const size_t total = (size_t)((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(maxpool_bwd_relu_kernel, dim3(cap_blocks(total, 256)), dim3(256), 0, stream, a, gpool, H, W,
                       C / 4, gin);
    return hipGetLastError();
}

// ------------------------------------------------------------------ layout changes (unit-parity API only)
__global__ void chw_to_hwc_kernel(const float* __restrict__ src, int C, size_t HW, float* __restrict__ dst) {
    const size_t total = HW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t p = i / C;
        dst[i] = src[(size_t)c * HW + p];
    }
}
__global__ void hwc_to_chw_kernel(const float* __restrict__ src, int C, size_t HW, float* __restrict__ dst) {
    const size_t total = HW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW;
        const int c = (int)(i / HW);
        dst[i] = src[p * C + c];
    }
}
hipError_t launch_chw_to_hwc(const float* src, int C, int H, int W, float* dst, hipStream_t stream) {
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(chw_to_hwc_kernel, dim3(cap_blocks(HW * C, 256)), dim3(256), 0, stream, src, C, HW, dst);
    return hipGetLastError();
}
hipError_t launch_hwc_to_chw(const float* src, int C, int H, int W, float* dst, hipStream_t stream) {
    const size_t HW = (size_t)H * W;
    hipLaunchKernelGGL(hwc_to_chw_kernel, dim3(cap_blocks(HW * C, 256)), dim3(256), 0, stream, src, C, HW, dst);
    return hipGetLastError();
}

__global__ void relu_mask_kernel(const f32x4* __restrict__ act, const f32x4* __restrict__ g, size_t n4,
                                 f32x4* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = act[i], v = g[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = a[k] > 0.f ? v[k] : 0.f;
        dst[i] = o;
    }
}
hipError_t launch_relu_mask(const float* act, const float* g, size_t n, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(relu_mask_kernel, dim3(cap_blocks(n / 4, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(act), reinterpret_cast<const f32x4*>(g), n / 4,
                       reinterpret_cast<f32x4*>(dst));
    return hipGetLastError();
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}
hipError_t launch_add_inplace(float* dst, const float* src, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3(cap_blocks(n, 256)), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ bicubic (torch:include/ATen/native/UpSample.h:289-312, :373-423)
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// source position of output index o: taps idx[0..3] (clamped) and weights wt[0..3]
__device__ __forceinline__ void cubic_taps(int o, float scale, int n_in, int idx[4], float wt[4]) {
    const float A = -0.75f;
    const float src = scale * (o + 0.5f) - 0.5f;
    const float fl = floorf(src);
    const float t = src - fl;
    const int i0 = (int)fl;
    wt[0] = cubic2(t + 1.f, A);
    wt[1] = cubic1(t, A);
    wt[2] = cubic1(1.f - t, A);
    wt[3] = cubic2(2.f - t, A);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int j = i0 - 1 + k;
        j = j < 0 ? 0 : (j > n_in - 1 ? n_in - 1 : j);
        idx[k] = j;
    }
}

__global__ void bicubic_down_kernel(const float* __restrict__ x, int C, int h, int w, int oh, int ow,
                                    float* __restrict__ y) {
    const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
    const size_t total = (size_t)C * oh * ow;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % ow);
        const int oy = (int)((i / ow) % oh);
        const int c = (int)(i / ((size_t)ow * oh));
        int iy[4], ix[4];
        float wy[4], wx[4];
        cubic_taps(oy, sh, h, iy, wy);
        cubic_taps(ox, sw, w, ix, wx);
        const float* plane = x + (size_t)c * h * w;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float* row = plane + (size_t)iy[a] * w;
            float r = row[ix[0]] * wx[0];
            r += row[ix[1]] * wx[1];
            r += row[ix[2]] * wx[2];
            r += row[ix[3]] * wx[3];
            acc = (a == 0) ? r * wy[0] : acc + r * wy[a];
        }
        y[i] = acc;
    }
}

hipError_t launch_bicubic_down(const float* x, int C, int h, int w, int oh, int ow, float* y, hipStream_t stream) {
    const size_t total = (size_t)C * oh * ow;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(bicubic_down_kernel, dim3(cap_blocks(total, 256)), dim3(256), 0, stream, x, C, h, w, oh, ow, y);
    return hipGetLastError();
}

// weight with which output index o reads input index i (sum over its taps that clamp onto i)
__device__ __forceinline__ float tap_weight(int o, float scale, int n_in, int i) {
    int idx[4];
    float wt[4];
    cubic_taps(o, scale, n_in, idx, wt);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (idx[k] == i) ? wt[k] : 0.f;
    return s;
}

// gather form of the transpose: deterministic, no atomics.  Candidate outputs of input index i:
// floor(src(o)) in [i-2, i+1] unless i is a border pixel that collects clamped taps.
// outputs o whose (clamped) taps can land on input index i.  With f(o) = floor(scale*(o+.5)-.5) the raw
// taps are f-1..f+2, so interior i needs f in [i-2, i+1]; i = 0 additionally collects every tap clamped
// from below (all o with f <= 1) and i = n_in-1 every tap clamped from above (all o with f >= n_in-3).
// The range is widened by one on each side against float rounding; tap_weight() tests the real taps.
__device__ __forceinline__ void cand_range(int i, float scale, int n_in, int n_out, int& lo, int& hi) {
    lo = (i == 0) ? 0 : (int)floorf(((float)i - 1.5f) / scale - 0.5f) - 1;
    hi = (i == n_in - 1) ? n_out - 1 : (int)ceilf(((float)i + 2.5f) / scale - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n_out - 1 ? n_out - 1 : hi;
}

__global__ void bicubic_down_bwd_kernel(const float* __restrict__ gy, int C, int h, int w, int oh, int ow,
                                        float* __restrict__ gx, int accumulate) {
    const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
    // grid = (column blocks, C * h rows): no 64-bit division per element
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    if (ix >= w) return;
    for (int row = blockIdx.y; row < C * h; row += gridDim.y) {
        const int c = row / h;
        const int iy = row - c * h;
        const size_t i = (size_t)row * w + ix;
        if (h == 2 * oh && w == 2 * ow && iy >= 2 && iy < h - 2 && ix >= 2 && ix < w - 2) {
            // exact 1/2 (the pyramid's case), away from the clamped border: every output o reads 2o-1 .. 2o+2 with the
            // weights of t = 0.5, so an input index receives from exactly two outputs per axis - even i = 2m: o = m-1
            // (tap 3), o = m (tap 1); odd i = 2m+1: o = m (tap 2), o = m+1 (tap 0).  Same weights, same order and same
            // expressions as the general form below: bit-identical, 4 loads instead of ~25 candidate evaluations.
            const float A = -0.75f;
            const float wo = cubic2(1.5f, A), wi = cubic1(0.5f, A);       // outer / inner tap weight
            const int my = iy >> 1, mx = ix >> 1;
            const int oy0 = (iy & 1) ? my : my - 1, ox0 = (ix & 1) ? mx : mx - 1;
            const float wy0 = (iy & 1) ? wi : wo, wy1 = (iy & 1) ? wo : wi;
            const float wx0 = (ix & 1) ? wi : wo, wx1 = (ix & 1) ? wo : wi;
            const float* p0 = gy + (size_t)c * oh * ow + (size_t)oy0 * ow + ox0;
            float acc = 0.f;
            {
                float r = 0.f;
                r += wx0 * p0[0];
                r += wx1 * p0[1];
                acc += wy0 * r;
            }
            {
                float r = 0.f;
                r += wx0 * p0[ow];
                r += wx1 * p0[ow + 1];
                acc += wy1 * r;
            }
            gx[i] = accumulate ? gx[i] + acc : acc;
            continue;
        }
        int ylo, yhi, xlo, xhi;
        cand_range(iy, sh, h, oh, ylo, yhi);
        cand_range(ix, sw, w, ow, xlo, xhi);
        const float* plane = gy + (size_t)c * oh * ow;
        float acc = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = tap_weight(oy, sh, h, iy);
            if (wy == 0.f) continue;
            float r = 0.f;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = tap_weight(ox, sw, w, ix);
                r += wx * plane[(size_t)oy * ow + ox];
            }
            acc += wy * r;
        }
        gx[i] = accumulate ? gx[i] + acc : acc;
    }
}

hipError_t launch_bicubic_down_bwd(const float* gy, int C, int h, int w, int oh, int ow, float* gx, int accumulate,
                                   hipStream_t stream) {
    if ((size_t)C * h * w == 0) return hipSuccess;
    const int rows = C * h;
    hipLaunchKernelGGL(bicubic_down_bwd_kernel, dim3((w + 255) / 256, rows < 65535 ? rows : 65535), dim3(256), 0, stream, gy, C,
                       h, w, oh, ow, gx, accumulate);
    return hipGetLastError();
}

// ------------------------------------------------------------------ total variation (math_utils.py:37-41)
__global__ __launch_bounds__(256) void tv_partial_kernel(const float* __restrict__ y, int C, int h, int w,
                                                         double* __restrict__ partial, int row0, int rows) {
    __shared__ double sh[4];
    // block b takes the image rows (c, r) with (c h + r) % TV_BLOCKS == b, a thread every 256th column: a fixed
    // assignment (reproducible sums) without any per-element division
    float sx = 0.f, sy = 0.f;
    for (int row = blockIdx.x; row < C * h; row += gridDim.x) {
        const int r = row % h;
        if (rows > 0 && (unsigned)(r - row0) >= (unsigned)rows) continue;
        const float* line = y + (size_t)row * w;
        for (int x = threadIdx.x; x < w; x += blockDim.x) {
            const float v = line[x];
            if (x + 1 < w) sx += fabsf(v - line[x + 1]);
            if (r + 1 < h) sy += fabsf(v - line[x + w]);
        }
    }
    const double bx = block_reduce_sum((double)sx, sh);
    const double by = block_reduce_sum((double)sy, sh);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = bx;
        partial[2 * blockIdx.x + 1] = by;
    }
}

hipError_t launch_tv_partial(const float* y, int C, int h, int w, double* partial, hipStream_t stream, int row0, int rows) {
    hipLaunchKernelGGL(tv_partial_kernel, dim3(TV_BLOCKS), dim3(256), 0, stream, y, C, h, w, partial, row0, rows);
    return hipGetLastError();
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(256) void tv_finish_kernel(const float* __restrict__ y, int C, int h, int w,
                                                        const double* __restrict__ partial, float weight,
                                                        float* __restrict__ grad, int accumulate,
                                                        float* __restrict__ means, int row0, int rows,
                                                        const float* __restrict__ given_means, double gnx, double gny) {
    __shared__ double sh[4];
    __shared__ float m[2];
    // every block re-reduces the TV_BLOCKS partials in the same fixed order
    double px = 0.0, py = 0.0;
    for (int b = threadIdx.x; b < TV_BLOCKS; b += blockDim.x) { px += partial[2 * b]; py += partial[2 * b + 1]; }
    const double tx = block_reduce_sum(px, sh);
    const double ty = block_reduce_sum(py, sh);
    // (given_means: the means and element counts of a larger image of which y is a stripe)
    const double nx = given_means ? gnx : (double)C * h * (w - 1), ny = given_means ? gny : (double)C * (h - 1) * w;
    if (threadIdx.x == 0) {
        m[0] = given_means ? given_means[0] : (float)tx / (float)nx;
        m[1] = given_means ? given_means[1] : (float)ty / (float)ny;
        if (blockIdx.x == 0 && blockIdx.y == 0 && means) { means[0] = m[0]; means[1] = m[1]; }
    }
    __syncthreads();
    if (!grad) return;
    // d(mx^2 + my^2) = 2 mx d mx + 2 my d my ; d mx / d y[i] = (sign(y_i - y_{i+1}) - sign(y_{i-1} - y_i)) / nx
    const float cx = weight * 2.f * m[0] / (float)nx;
    const float cy = weight * 2.f * m[1] / (float)ny;
    // grid = (column blocks, C * h rows)
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    for (int row = blockIdx.y; row < C * h; row += gridDim.y) {
        const int r = row % h;
        if (rows > 0 && (unsigned)(r - row0) >= (unsigned)rows) continue;
        const size_t i = (size_t)row * w + x;
        const float v = y[i];
        float gx = 0.f, gyv = 0.f;
        if (x + 1 < w) gx += sgn(v - y[i + 1]);
        if (x > 0) gx -= sgn(y[i - 1] - v);
        if (r + 1 < h) gyv += sgn(v - y[i + w]);
        if (r > 0) gyv -= sgn(y[i - w] - v);
        const float g = cx * gx + cy * gyv;
        grad[i] = accumulate ? grad[i] + g : g;
    }
}

hipError_t launch_tv_finish(const float* y, int C, int h, int w, const double* partial, float weight, float* grad,
                            int accumulate, float* means, hipStream_t stream, int row0, int rows, const float* given_means,
                            double nx, double ny) {
    const int all_rows = C * h;
    // (every workgroup re-reduces the TV_BLOCKS partials first: a few rows per workgroup, but enough workgroups to
    // keep ~10 waves per SIMD in flight - with 96 row groups this kernel ran 2x slower)
    const dim3 grid = grad ? dim3((w + 255) / 256, all_rows < 768 ? all_rows : 768) : dim3(1, 1);
    hipLaunchKernelGGL(tv_finish_kernel, grid, dim3(256), 0, stream, y, C, h, w, partial, weight, grad, accumulate, means,
                       row0, rows, given_means, nx, ny);
    return hipGetLastError();
}

// ------------------------------------------------------------------ small scalar plumbing of the stripe (window) closure
// out[0] = float(sum_k p[offset + k * stride]), k < n, fixed order
__global__ __launch_bounds__(256) void sum_doubles_kernel(const double* __restrict__ p, int n, int stride, int offset,
                                                          float* __restrict__ out) {
    __shared__ double sh[4];
    double t = 0.0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) t += p[offset + (size_t)k * stride];
    const double r = block_reduce_sum(t, sh);
    if (threadIdx.x == 0) out[0] = (float)r;
}
hipError_t launch_sum_doubles(const double* p, int n, int stride, int offset, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, stream, p, n, stride, offset, out);
    return hipGetLastError();
}
// means[0] = sums[0] / nx, means[1] = sums[1] / ny (float divisions, as tv_finish does); partial[0] = sse, partial[1..n) = 0
__global__ void window_scalars_kernel(const float* __restrict__ sums, float nx, float ny, float* __restrict__ means,
                                      double* __restrict__ partial, int n) {
    if (threadIdx.x == 0) {
        means[0] = sums[1] / nx;
        means[1] = sums[2] / ny;
    }
    for (int k = threadIdx.x; k < n; k += blockDim.x) partial[k] = (k == 0) ? (double)sums[0] : 0.0;
}
hipError_t launch_window_scalars(const float* sums, double nx, double ny, float* means, double* partial, int n,
                                 hipStream_t stream) {
    hipLaunchKernelGGL(window_scalars_kernel, dim3(1), dim3(256), 0, stream, sums, (float)nx, (float)ny, means, partial, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------ content MSE (neural_style_transfer.py:95)
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                                       size_t n, float coef, float* __restrict__ g,
                                                       double* __restrict__ partial) {
    __shared__ double sh[4];
    float s = 0.f;
    const size_t n4 = n / 4;
    const f32x4* av = reinterpret_cast<const f32x4*>(a);
    const f32x4* tv = reinterpret_cast<const f32x4*>(t);
    f32x4* gv = reinterpret_cast<f32x4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 x = av[i], y = tv[i];
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) { d[k] = x[k] - y[k]; s += d[k] * d[k]; }
        if (g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] *= coef;
            gv[i] = d;
        }
    }
    const double b = block_reduce_sum((double)s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = b;
}

hipError_t launch_mse_grad(const float* a, const float* t, size_t n, float coef, float* g, double* partial,
                           hipStream_t stream) {
    if (n % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(MSE_BLOCKS), dim3(256), 0, stream, a, t, n, coef, g, partial);
    return hipGetLastError();
}

// ------------------------------------------------------------------ prepare / unprepare (neural_style_transfer.py:375-393)
__constant__ float kMean[3] = {123.675f, 116.28f, 103.53f};
__constant__ double kMeanD[3] = {123.675, 116.28, 103.53};

// hipcc contracts a*b+c into one fma by default; these three kernels restate host arithmetic that
// rounds after every operation, so contraction is switched off inside them.
__global__ void prepare_img_kernel(const float* __restrict__ hwc, size_t HW, float* __restrict__ chw) {
#pragma clang fp contract(off)
    const size_t total = HW * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i % HW;
        const int c = (int)(i / HW);
        chw[i] = hwc[p * 3 + c] * 255.f - kMean[c];     // two roundings (x.mul(255) then Normalize)
    }
}
__global__ void unprepare_img_kernel(const float* __restrict__ chw, size_t HW, float* __restrict__ hwc) {
#pragma clang fp contract(off)
    const size_t total = HW * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const size_t p = i / 3;
        // numpy: float32 += float64 array (computed in double, rounded to float32), then / 255 in float32
        const float v = (float)((double)chw[(size_t)c * HW + p] + kMeanD[c]);
        hwc[i] = v / 255.f;
    }
}
hipError_t launch_prepare_img(const float* hwc, int h, int w, float* chw, hipStream_t stream) {
    const size_t HW = (size_t)h * w;
    hipLaunchKernelGGL(prepare_img_kernel, dim3(cap_blocks(HW * 3, 256)), dim3(256), 0, stream, hwc, HW, chw);
    return hipGetLastError();
}
hipError_t launch_unprepare_img(const float* chw, int h, int w, float* hwc, hipStream_t stream) {
    const size_t HW = (size_t)h * w;
    hipLaunchKernelGGL(unprepare_img_kernel, dim3(cap_blocks(HW * 3, 256)), dim3(256), 0, stream, chw, HW, hwc);
    return hipGetLastError();
}

// ------------------------------------------------------------------ loss rows (neural_style_transfer.py:95-110, :179-185)
// one workgroup per level: the six sums of a level (content + 5 style) go through ONE pair of barriers (each sum
// keeps its own fixed order: strided per-thread share in index order, wave tree, waves in order)
__global__ __launch_bounds__(256) void loss_rows_kernel(LossAssembly la) {
#pragma clang fp contract(off)
    __shared__ double sh[6][4];
    const int l = blockIdx.x;
    const LevelLossInputs& in = la.lv[l];
    if (!in.owned) {
        if (threadIdx.x < 4) la.out[4 * l + threadIdx.x] = 0.f;
        return;
    }
    double v[6];
    v[0] = threadIdx.x < MSE_BLOCKS ? in.content_partial[threadIdx.x] : 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        // C*C / NST_GRAM_FINISH_EPB partials: each thread adds its strided share in index order, then the fixed tree
        const int nb = (in.style_c[k] * in.style_c[k] + NST_GRAM_FINISH_EPB - 1) / NST_GRAM_FINISH_EPB;
        double p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = threadIdx.x + j * 256;
            p[j] = b < nb ? in.style_partial[k][b] : 0.0;
        }
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += p[j];
        for (int b = threadIdx.x + 8 * 256; b < nb; b += 256) t += in.style_partial[k][b];
        v[k + 1] = t;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        double x = v[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) sh[q][w] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[6];
        for (int q = 0; q < 6; ++q) { r[q] = 0.0; for (int i = 0; i < 4; ++i) r[q] += sh[q][i]; }
        const float content = (float)(r[0] / (double)in.content_n);
        float style = 0.f;
        for (int k = 0; k < 5; ++k) style = style + (float)(r[k + 1] / ((double)in.style_c[k] * in.style_c[k]));
        style = style / 5.f;
        const float mx = in.tv_means[0], my = in.tv_means[1];
        const float tv = mx * mx + my * my;
        // cw*content + sw*style + tvw*tv, each product and sum rounded (contraction is off here)
        const float t0 = la.cw * content, t1 = la.sw * style, t2 = la.tvw * tv;
        la.out[4 * l + 0] = (t0 + t1) + t2;
        la.out[4 * l + 1] = content;
        la.out[4 * l + 2] = style;
        la.out[4 * l + 3] = tv;
    }
}
// total = sum of the owned levels' totals in level order (neural_style_transfer.py:179-185)
__global__ void loss_total_kernel(LossAssembly la) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool first = true;
    float tot = 0.f;
    for (int l = 0; l < la.levels; ++l) {
        if (!la.lv[l].owned) continue;
        const float total = la.out[4 * l];
        tot = first ? total : (1.0f * tot + total);
        first = false;
    }
    la.out[4 * la.levels] = first ? 0.f : tot;
}

hipError_t launch_loss_assemble(const LossAssembly& la, hipStream_t stream) {
    hipLaunchKernelGGL(loss_rows_kernel, dim3(la.levels), dim3(256), 0, stream, la);
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, stream, la);
    return hipGetLastError();
}

}  // namespace nst
