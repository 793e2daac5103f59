// image_ops.hip - job set-up on the device: what the reference does with OpenCV
// on the host before the optimisation loop (neural_style_transfer.py:211-226, :249-362, :396-439):
// pyramid resize, style-pixel noise with Gaussian envelopes, the Sobel-gradient blend weight and the
// initial image.  All O(pixels), HBM-bound; arithmetic in double where the reference computes in float64
// (cv2.Sobel(CV_64F), the Gaussian masks, the blend) so that results match the host restatement
// (host_image.py) to rounding.
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

static inline int img_blocks(size_t work) {
    size_t b = (work + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- bicubic resize of an HWC image: cv2.resize(..., INTER_CUBIC) for float images, in OpenCV's own arithmetic ----
// source coordinate fx = (float)((dx + 0.5) * scale - 0.5) with scale in DOUBLE (resize.cpp forms it so and only then
// drops to float), sx = floor(fx), t = fx - sx; weights = interpolateCubic(t) with A = -0.75 in float, the last one as
// 1 - (w0 + w1 + w2); indices clamped (replicate border); no antialiasing when shrinking.
__device__ __forceinline__ void itaps(int o, double scale, int n_in, int idx[4], float wt[4]) {
    const float A = -0.75f;
    const float src = (float)(((double)o + 0.5) * scale - 0.5);
    const float fl = floorf(src);
    const float t = src - fl;
    const int i0 = (int)fl;
    wt[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    wt[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    wt[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    wt[3] = 1.f - wt[0] - wt[1] - wt[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int j = i0 - 1 + k;
        idx[k] = j < 0 ? 0 : (j > n_in - 1 ? n_in - 1 : j);
    }
}

__global__ void resize_hwc_kernel(const float* __restrict__ src, int h, int w, int C, float* __restrict__ dst, int oh,
                                  int ow) {
    const double sh = (double)h / (double)oh, sw = (double)w / (double)ow;
    const size_t total = (size_t)oh * ow * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ox = (int)((i / C) % ow);
        const int oy = (int)(i / ((size_t)C * ow));
        int iy[4], ix[4];
        float wy[4], wx[4];
        itaps(oy, sh, h, iy, wy);
        itaps(ox, sw, w, ix, wx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float* row = src + (size_t)iy[a] * w * C + c;
            float r = row[(size_t)ix[0] * C] * wx[0];
            r += row[(size_t)ix[1] * C] * wx[1];
            r += row[(size_t)ix[2] * C] * wx[2];
            r += row[(size_t)ix[3] * C] * wx[3];
            acc = (a == 0) ? r * wy[0] : acc + r * wy[a];
        }
        dst[i] = acc;
    }
}
hipError_t launch_resize_hwc(const float* src, int h, int w, int C, float* dst, int oh, int ow, hipStream_t stream) {
    hipLaunchKernelGGL(resize_hwc_kernel, dim3(img_blocks((size_t)oh * ow * C)), dim3(256), 0, stream, src, h, w, C, dst,
                       oh, ow);
    return hipGetLastError();
}

// ---- row gather: dst[i][:] = src[perm[i]][:] (make_style_noise's np.random.permutation of RGB rows) ----
__global__ void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ perm, size_t n, int C,
                                   float* __restrict__ dst) {
    const size_t total = n * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C;
        dst[i] = src[(size_t)perm[r] * C + (i % C)];
    }
}
hipError_t launch_gather_rows(const float* src, const long long* perm, size_t n, int C, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(img_blocks(n * C)), dim3(256), 0, stream, src, perm, n, C, dst);
    return hipGetLastError();
}

// ---- acc += (src or 1) * gaussian_mask (neural_style_transfer.py:396-418) ----
// mask = p + g(y,x)/g(h//2,w//2) * (c - p), g = outer product of two normalised Gaussian kernels
// (sigma = size * dispersion): the normalisations cancel in the ratio, leaving a closed form.
// The envelope is an outer product: a thread owns one column (its exp once), a block 16 rows (their 16 exps once, through
// LDS) - two double exps per pixel (6 ms per call at 1536x1024, five calls per job) were all this kernel did.  The value
// per pixel is the same expression on the same operands as before: bit-identical.
constexpr int GM_ROWS = 16;
__global__ __launch_bounds__(256) void gauss_mask_acc_kernel(float* __restrict__ acc, const float* __restrict__ src, int h, int w, int C,
                                                             double central, double peripheral, double disp) {
    __shared__ double ey[GM_ROWS];
    const double sy = (double)h * disp, sx = (double)w * disp;
    const double cy = (h - 1) * 0.5, cx = (w - 1) * 0.5;
    const double ry = (h / 2) - cy, rx = (w / 2) - cx;
    const double ref = exp(-(ry * ry) / (2.0 * sy * sy)) * exp(-(rx * rx) / (2.0 * sx * sx));
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * GM_ROWS;
    if (threadIdx.x < GM_ROWS) {
        const double dy = (y0 + (int)threadIdx.x) - cy;
        ey[threadIdx.x] = exp(-(dy * dy) / (2.0 * sy * sy));
    }
    __syncthreads();
    if (x >= w) return;
    const double dx = x - cx;
    const double ex = exp(-(dx * dx) / (2.0 * sx * sx));
    for (int r = 0; r < GM_ROWS && y0 + r < h; ++r) {
        const double g = ey[r] * ex / ref;
        const double mask = peripheral + g * (central - peripheral);
        const size_t p = (size_t)(y0 + r) * w + x;
        for (int c = 0; c < C; ++c) {
            const size_t i = p * C + c;
            // numpy: float32 accumulator += float64 product, rounded to float32 on store
            const double add = src ? (double)src[i] * mask : mask;
            acc[i] = (float)((double)acc[i] + add);
        }
    }
}
hipError_t launch_gauss_mask_acc(float* acc, const float* src, int h, int w, int C, double central, double peripheral,
                                 double disp, hipStream_t stream) {
    hipLaunchKernelGGL(gauss_mask_acc_kernel, dim3((w + 255) / 256, (h + GM_ROWS - 1) / GM_ROWS), dim3(256), 0, stream, acc, src, h, w, C,
                       central, peripheral, disp);
    return hipGetLastError();
}

// ---- blend weight a*nf/(a + blur(clip(|sobel|))) (neural_style_transfer.py:331-343) ----
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}
// Sobel ksize 5 = [-1,-2,0,2,1] (derivative) x [1,4,6,4,1] (smoothing), BORDER_REFLECT_101, CV_64F
__global__ void sobel_mag_kernel(const float* __restrict__ img, int h, int w, int C, double* __restrict__ mag) {
    const double kd[5] = {-1, -2, 0, 2, 1}, ks[5] = {1, 4, 6, 4, 1};
    const size_t total = (size_t)h * w * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % w);
        const int y = (int)(i / ((size_t)C * w));
        double gx = 0.0, gy = 0.0;
        for (int a = 0; a < 5; ++a) {
            const int yy = reflect101(y + a - 2, h);
            for (int b = 0; b < 5; ++b) {
                const int xx = reflect101(x + b - 2, w);
                const double v = (double)img[((size_t)yy * w + xx) * C + c];
                gx += ks[a] * kd[b] * v;
                gy += kd[a] * ks[b] * v;
            }
        }
        double m = sqrt(gx * gx + gy * gy);
        mag[i] = m < 0.0 ? 0.0 : (m > 100.0 ? 100.0 : m);
    }
}
// separable Gaussian blur with sigma (taps beyond |i| <= R are below 1e-300 for the reference's sigma = 0.2 and
// are dropped; the kernel is renormalised over the kept taps exactly as cv2.getGaussianKernel normalises its 101),
// then weight = a * nf / (a + blurred)
template <int R>
__global__ void blur_weight_kernel(const double* __restrict__ mag, int h, int w, int C, double sigma, int pass,
                                   double a, double nf, double* __restrict__ out) {
    double k[2 * R + 1], ksum = 0.0;
    for (int t = -R; t <= R; ++t) { k[t + R] = exp(-(double)(t * t) / (2.0 * sigma * sigma)); ksum += k[t + R]; }
    const size_t total = (size_t)h * w * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % w);
        const int y = (int)(i / ((size_t)C * w));
        double s = 0.0;
        for (int t = -R; t <= R; ++t) {
            const int yy = pass ? reflect101(y + t, h) : y;
            const int xx = pass ? x : reflect101(x + t, w);
            s += (k[t + R] / ksum) * mag[((size_t)yy * w + xx) * C + c];
        }
        out[i] = pass ? a * nf / (a + s) : s;
    }
}
hipError_t launch_blend_weight(const float* content, int h, int w, int C, double noise_factor, double* tmp0, double* tmp1,
                               double* weight, hipStream_t stream) {
    const int blocks = img_blocks((size_t)h * w * C);
    hipLaunchKernelGGL(sobel_mag_kernel, dim3(blocks), dim3(256), 0, stream, content, h, w, C, tmp0);
    hipLaunchKernelGGL(blur_weight_kernel<3>, dim3(blocks), dim3(256), 0, stream, tmp0, h, w, C, 0.2, 0, 5.0, noise_factor,
                       tmp1);
    hipLaunchKernelGGL(blur_weight_kernel<3>, dim3(blocks), dim3(256), 0, stream, tmp1, h, w, C, 0.2, 1, 5.0, noise_factor,
                       weight);
    return hipGetLastError();
}

// ---- init = ((1 - w) * content + w * noise).astype(float32) (neural_style_transfer.py:355-358) ----
__global__ void blend_init_kernel(const float* __restrict__ content, const float* __restrict__ noise,
                                  const double* __restrict__ weight, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double wv = weight[i];
        out[i] = (float)((1.0 - wv) * (double)content[i] + wv * (double)noise[i]);
    }
}
hipError_t launch_blend_init(const float* content, const float* noise, const double* weight, size_t n, float* out,
                             hipStream_t stream) {
    hipLaunchKernelGGL(blend_init_kernel, dim3(img_blocks(n)), dim3(256), 0, stream, content, noise, weight, n, out);
    return hipGetLastError();
}

__global__ void scale_kernel(const float* __restrict__ src, float alpha, size_t n, float* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i] * alpha;
}
hipError_t launch_scale(const float* src, float alpha, size_t n, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(scale_kernel, dim3(img_blocks(n)), dim3(256), 0, stream, src, alpha, n, dst);
    return hipGetLastError();
}

}  // namespace nst
