// conv_first.hip - conv1_1 of VGG19 (3 -> 64 channels, torchvision features[0:2]) read straight
// from the planar (3,H,W) image, and its input gradient (64 -> 3) written straight into the
// planar pixel gradient.  Both are HBM-bound (13 flop/B): the 64-channel side is streamed once.
//
// Forward: implicit GEMM with K = 27 (padded to 28) on v_mfma_f32_32x32x2_f32; the 28x64 weight
// table lives in registers, the 3-plane halo patch in LDS.
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
// Block tile: 16 x 16 pixels, 4 waves of 4 rows (two 32-pixel MFMA row tiles per wave).
constexpr int F_TH = 16, F_TW = 16;
constexpr int F_MT = F_TH / 8;                // 32-pixel row tiles per wave
constexpr int F_PH = F_TH + 2, F_PW = F_TW + 2;
constexpr int F_PLANE = F_PH * F_PW;

// LDS offset (floats) of tap k = c*9 + ky*3 + kx relative to the pixel's own patch position
__host__ __device__ constexpr int tap_off(int k) {
    return (k >= 27) ? 0 : (k / 9) * F_PLANE + ((k % 9) / 3) * F_PW + (k % 3);
}
}  // namespace

// A PERSISTENT tile loop (grid = two workgroups per CU): per tile the kernel is a 64-KB store stream behind 1.7 us of
// fp32 MFMA per wave, with a set-up in front - 28 weight loads per lane and the 3-plane halo patch - that a
// one-tile-per-workgroup launch pays 6144 times at level 0 with nothing to hide it under (2.6 TB/s written, where a
// plain fill reaches 6.9; 8-row tiles for twice the occupancy made it SLOWER: twice the set-up per pixel).  Here the
// weights are loaded once per workgroup and the next tile's patch is fetched into registers before the current
// tile's MFMAs and written to LDS after its stores.
constexpr int F_PATCH_PER_T = (3 * F_PLANE + 255) / 256;

__global__ __launch_bounds__(256, 2) void conv1_1_fwd_kernel(const float* __restrict__ x, int H, int W,
                                                          const float* __restrict__ wk,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          unsigned* __restrict__ bits_out,
                                                          unsigned* __restrict__ amax_out, int ntiles) {
    __shared__ float patch[3 * F_PLANE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int tiles_x = (W + F_TW - 1) / F_TW;

    // patch element u of this thread: plane c, patch row r, patch column col (fixed for every tile)
    int pu_off[F_PATCH_PER_T];          // (c, r, col) packed
#pragma unroll
    for (int i = 0; i < F_PATCH_PER_T; ++i) {
        const int u = tid + i * 256;
        const int c = u / F_PLANE, r = (u % F_PLANE) / F_PW, col = u % F_PW;
        pu_off[i] = (u < 3 * F_PLANE) ? (c << 16) | (r << 8) | col : -1;
    }
    auto fetch = [&](int tile, float* v) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * F_TH, x0 = tx * F_TW;
#pragma unroll
        for (int i = 0; i < F_PATCH_PER_T; ++i) {
            const int pk = pu_off[i];
            const int c = pk >> 16, gy = y0 - 1 + ((pk >> 8) & 255), gx = x0 - 1 + (pk & 255);
            v[i] = (pk >= 0 && tile < ntiles && gy >= 0 && gy < H && gx >= 0 && gx < W) ? x[((size_t)c * H + gy) * W + gx] : 0.f;
        }
    };
    auto stash = [&](const float* v) {
#pragma unroll
        for (int i = 0; i < F_PATCH_PER_T; ++i)
            if (pu_off[i] >= 0) patch[tid + i * 256] = v[i];
    };

    float pv[F_PATCH_PER_T];
    fetch(blockIdx.x, pv);
    // B operand: lane (n = l31, k-half) holds W[k = 2*kk + half][nt*32 + n] - once per workgroup
    float bw[14][2];
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) {
        const int k = 2 * kk + half;
        bw[kk][0] = wk[k * 64 + l31];
        bw[kk][1] = wk[k * 64 + 32 + l31];
    }
    const float bv0 = bias[l31], bv1 = bias[32 + l31];
    stash(pv);
    __syncthreads();

    constexpr int WROWS = 2 * F_MT;                 // image rows per wave
    const int prow = wave * WROWS + (l31 >> 4);
    const int pcol = l31 & 15;
    const int base0 = prow * F_PW + pcol;           // M-tile mt: rows wave * WROWS + 2 mt + {0,1}
    float amax = 0.f;
    const bool small = (size_t)H * W * 256 < 0xFFFFFF00ull;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, small ? (unsigned)((size_t)H * W * 256) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(bits_out, 0, (bits_out && small) ? (unsigned)((size_t)H * W * 8) : 0u, 0x00020000);

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * F_TH, x0 = tx * F_TW;
        fetch(tile + gridDim.x, pv);                // the next tile's patch: in flight under this tile's MFMAs

        f32x16 acc[F_MT][2];
#pragma unroll
        for (int a = 0; a < F_MT; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) {
            const int off = half ? tap_off(2 * kk + 1) : tap_off(2 * kk);
#pragma unroll
            for (int mt = 0; mt < F_MT; ++mt) {
                const float a = patch[base0 + 2 * mt * F_PW + off];
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[kk][0], acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[kk][1], acc[mt][1], 0, 0, 0);
            }
        }

        if (y0 + F_TH <= H && x0 + F_TW <= W && small) {
            // interior tile: buffer stores = per-lane byte offset + a scalar offset per element, no bounds tests
            // (the same form as conv_h2.hip's fast epilogue)
            const int pix0 = (y0 + wave * WROWS) * W + x0 + 4 * half;
            const unsigned vbase = (unsigned)(pix0 * 256 + l31 * 4);
            const unsigned wlane = (l31 == 0) ? (unsigned)(pix0 * 8) : 0xFFFFF000u;     // (+ immediates below 4 KiB: still beyond the buffer)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float bv = nt ? bv1 : bv0;
#pragma unroll
                for (int mt = 0; mt < F_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // pixel of register r: row (r >> 3) + 2 mt (a scalar offset, four values in all), column
                        // (r & 3) + 8 ((r >> 2) & 1) (a compile-time constant: the instruction's immediate offset)
                        const int rowpix = ((r >> 3) + 2 * mt) * W;
                        const int colpix = (r & 3) + 8 * ((r >> 2) & 1);
                        const float v = fmaxf(acc[mt][nt][r] + bv, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vbase + (unsigned)(colpix * 256 + nt * 128), rowpix * 256, 0);
                        amax = fmaxf(amax, v);
                        if (bits_out) {
                            const unsigned long long bal = __ballot(v > 0.f);
                            __builtin_amdgcn_raw_buffer_store_b32(half ? (unsigned)(bal >> 32) : (unsigned)bal, rs_bits, wlane + (unsigned)(colpix * 8 + nt * 4), rowpix * 8, 0);
                        }
                    }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int co = nt * 32 + l31;
                const float bv = nt ? bv1 : bv0;
#pragma unroll
                for (int mt = 0; mt < F_MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                        const int y = y0 + wave * WROWS + mt * 2 + (m >> 4);
                        const int xx = x0 + (m & 15);
                        const bool inb = (y < H && xx < W);
                        const float v = fmaxf(acc[mt][nt][r] + bv, 0.f);
                        if (inb) {
                            out[((size_t)y * W + xx) * 64 + co] = v;
                            amax = fmaxf(amax, v);
                        }
                        if (bits_out) {
                            const unsigned long long bal = __ballot(v > 0.f);
                            if (l31 == 0 && inb) bits_out[((size_t)y * W + xx) * 2 + nt] = half ? (unsigned)(bal >> 32) : (unsigned)bal;
                        }
                    }
            }
        }
        __syncthreads();                    // every wave has read this tile's patch
        stash(pv);
        __syncthreads();
    }
    if (amax_out) {
        // absmax of the output for the fp16-piece convolution that consumes it (conv_h2.hip)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (lane == 0) atomicMax(amax_out + ((blockIdx.x * 4 + wave) & (NST_AMAX_SLOTS - 1)), __float_as_uint(amax));
    }
}

hipError_t launch_conv1_1_fwd(const float* x, int H, int W, const float* wk, const float* bias, float* out,
                              unsigned* bits_out, unsigned* amax_out, hipStream_t stream) {
    const int ntiles = ((H + F_TH - 1) / F_TH) * ((W + F_TW - 1) / F_TW);
    const int blocks = ntiles < 512 ? ntiles : 512;          // two workgroups per CU, each walking tiles b, b + 512, ...
    hipLaunchKernelGGL(conv1_1_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, H, W, wk, bias, out, bits_out, amax_out, ntiles);
    return hipGetLastError();
}

// Input gradient of conv1_1: gx[c][y][x] = sum_{tap, co} g[y+dy-1][x+dx-1][co] * wd[tap][co][c]
// (wd holds the tap-flipped weights).  N = 3 output channels is no MFMA shape (a 32-wide tile would
// waste 29/32 of the matrix pipe), so this runs on the VALU: one pixel per lane, three accumulators,
// the 64-channel gradient of the 18x18 halo patch staged through LDS in two 32-channel slices
// (144-B rows: conflict-light ds_read_b128), weights wave-uniform through the scalar path.
namespace {
constexpr int D_T = 16;                    // 16 x 16 pixel tile, 256 threads
constexpr int D_P = D_T + 2;
constexpr int D_KC = 32;
constexpr int D_RS = D_KC + 4;             // LDS row stride (floats)
constexpr int D_UNITS = D_P * D_P * (D_KC / 4);
}  // namespace

__global__ __launch_bounds__(256) void conv1_1_dgrad_kernel(const float* __restrict__ g, int H, int W,
                                                            const float* __restrict__ wd, float* __restrict__ gx) {
    __shared__ __attribute__((aligned(16))) float patch[D_P * D_P * D_RS];
    const int tid = threadIdx.x;
    const int tiles_x = (W + D_T - 1) / D_T;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int y0 = ty * D_T, x0 = tx * D_T;
    const int py = tid >> 4, px = tid & 15;
    // channels 0 and 1 accumulate as a pair (v_pk_fma_f32 with the gradient value broadcast), channel 2 alone: two
    // VALU instructions per (gradient value) instead of three on a VALU-bound kernel
    f32x2 a01 = {0.f, 0.f};
    float a2 = 0.f;
    // the halo patch of a 32-channel slice: global -> registers -> LDS; the next slice's loads are issued before this
    // slice is multiplied, so that only the first one is exposed
    constexpr int D_PER_T = (D_UNITS + 255) / 256;
    f32x4 stage[D_PER_T];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < D_PER_T; ++i) {
            const int u = tid + i * 256;
            const int pix = u / (D_KC / 4), q = u % (D_KC / 4);
            const int gy = y0 - 1 + pix / D_P, gxx = x0 - 1 + pix % D_P;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < D_UNITS && gy >= 0 && gy < H && gxx >= 0 && gxx < W)
                v = *reinterpret_cast<const f32x4*>(g + ((size_t)gy * W + gxx) * 64 + chunk * D_KC + q * 4);
            stage[i] = v;
        }
    };
    fetch(0);
    for (int chunk = 0; chunk < 64 / D_KC; ++chunk) {
        if (chunk) __syncthreads();
#pragma unroll
        for (int i = 0; i < D_PER_T; ++i) {
            const int u = tid + i * 256;
            if (u < D_UNITS) *reinterpret_cast<f32x4*>(patch + (u / (D_KC / 4)) * D_RS + (u % (D_KC / 4)) * 4) = stage[i];
        }
        __syncthreads();
        if (chunk + 1 < 64 / D_KC) fetch(chunk + 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* row = patch + ((py + t / 3) * D_P + px + t % 3) * D_RS;
            const f32x4* wv = reinterpret_cast<const f32x4*>(wd + (t * 64 + chunk * D_KC) * 4);
#pragma unroll
            for (int q = 0; q < D_KC / 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + q * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 w = wv[q * 4 + k];
                    a01 += f32x2{v[k], v[k]} * f32x2{w[0], w[1]};
                    a2 += v[k] * w[2];
                }
            }
        }
    }
    const int y = y0 + py, x = x0 + px;
    if (y < H && x < W) {
        const size_t HW = (size_t)H * W, i = (size_t)y * W + x;
        gx[i] = a01[0];
        gx[HW + i] = a01[1];
        gx[2 * HW + i] = a2;
    }
}

hipError_t launch_conv1_1_dgrad(const float* g, int H, int W, const float* wd, float* gx, hipStream_t stream) {
    const int blocks = ((H + D_T - 1) / D_T) * ((W + D_T - 1) / D_T);
    hipLaunchKernelGGL(conv1_1_dgrad_kernel, dim3(blocks), dim3(256), 0, stream, g, H, W, wd, gx);
    return hipGetLastError();
}

}  // namespace nst
