// conv_first.hip - conv1_1 of VGG19 (3 -> 64 channels, torchvision features[0:2]) read straight
// from the planar (3,H,W) image, and its input gradient (64 -> 3) written straight into the
// planar pixel gradient.  Both are HBM-bound (13 flop/B): the 64-channel side is streamed once.
//
// Forward: implicit GEMM with K = 27 (padded to 28) on v_mfma_f32_32x32x2_f32; the 28x64 weight
// table lives in registers, the 3-plane halo patch in LDS.
// Input gradient: conv1_1_dgrad_h2_kernel (two-piece fp16 arithmetic on v_mfma_f32_16x16x32_f16, the default) and
// conv1_1_dgrad_kernel (fp32 on the VALU, for the paths that record no absmax).
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

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
// (wd holds the tap-flipped weights).  The fp32 form, on the VALU: one pixel per lane, three accumulators,
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

// The same sum on the matrix pipe, in the two-piece fp16 arithmetic of conv_h2.hip (every fp32 operand = hi + 2^-11 lo in
// fp16 under a power-of-two scale; main and cross products in separate fp32 accumulators).  N = 3 wastes 13/16 of a
// 16-wide tile, so the three tap ROWS are made output columns as well: with
//     P[ky][c][t][x] = sum_{kx, co} g[t][x + kx - 1][co] * wd[ky*3 + kx][co][c]        (t = a row of the halo patch)
// the gradient is gx[c][y][x] = P[0][c][y-1][x] + P[1][c][y][x] + P[2][c][y+1][x]: one GEMM with M = the pixels of a patch
// row, N = (ky, c) = 9 of 16 columns, K = (kx, co) = 192, and three accumulators of neighbouring patch rows added across
// lanes at the end.  v_mfma_f32_16x16x32_f16: lane (l15, kg) holds A[pixel l15][k = 8 kg + j], B[k = 8 kg + j][column l15],
// D[pixel 4 kg + i][column l15].  A wave owns four image rows = six patch rows (two recomputed per wave: 54 MFMAs per
// 32-channel slice where the VALU form spends 288 fma + 288 pk_fma per lane).  The 12 weight fragments (3 kx x 2 slices
// x 2 pieces = 48 registers) stay in registers for the whole kernel; the patch is cut into pieces on its way into LDS
// (144-B pixel rows: 16 consecutive pixels on 16 distinct 16-B slots).  The gradient's scale comes from the absmax slots the
// producing launch recorded (conv_h2.hip epilogue), the weights' from their own maximum.
namespace {
constexpr int G_P = D_T + 2;
constexpr int G_KC = 32;
constexpr int G_PIECEB = G_KC * 2;
constexpr int G_ROWB = 2 * G_PIECEB + 16;
constexpr int G_UNITS = G_P * G_P * (G_KC / 4);
constexpr int G_PER_T = (G_UNITS + 255) / 256;
constexpr float G_LO_UP = 2048.f, G_LO_DOWN = 1.f / 2048.f;

// power-of-two scale bringing a tensor of largest magnitude (bit pattern) m into [2^14, 2^15), and its inverse
__device__ __forceinline__ void pow2_scale(const unsigned m, float& s, float& inv) {
    int e = (int)((m >> 23) & 0xFFu);
    e = e < 32 ? 32 : (e > 250 ? 250 : e);
    s = __uint_as_float((unsigned)(268 - e) << 23);
    inv = __uint_as_float((unsigned)(e - 14) << 23);
}
__device__ __forceinline__ unsigned wave_max(unsigned m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = o > m ? o : m;
    }
    return m;
}
}  // namespace

__global__ __launch_bounds__(256, 2) void conv1_1_dgrad_h2_kernel(const float* __restrict__ g, int H, int W,
                                                                  const float* __restrict__ wd,
                                                                  const unsigned* __restrict__ amax_g, float* __restrict__ gx,
                                                                  int ntiles) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[G_P * G_P * G_ROWB];
    __shared__ unsigned wmax[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int tiles_x = (W + D_T - 1) / D_T;

    // a 32-channel slice of a tile's halo patch: global -> registers here, registers -> LDS (cut into pieces) at the top
    // of the slice's turn; the fetch of the NEXT slice (of this tile or of the workgroup's next one) is always in flight
    // under the MFMAs of the current one
    f32x4 stage[G_PER_T];
    // (the staging is VALU work beside the MFMAs: a tile whose halo patch lies inside the image - all but the rim - takes
    // its 11 addresses from a per-lane offset table instead of deriving and bounds-testing each)
    unsigned off[G_PER_T];
#pragma unroll
    for (int i = 0; i < G_PER_T; ++i) {
        const int u = tid + i * 256;
        const int pix = u < G_UNITS ? u / (G_KC / 4) : 0, q = u % (G_KC / 4);
        off[i] = (unsigned)(((pix / G_P) * W + pix % G_P) * 64 + q * 4);
    }
    auto fetch = [&](const int tile, const int chunk) {
        const int y0 = (tile / tiles_x) * D_T, x0 = (tile % tiles_x) * D_T;
        if (tile < ntiles && y0 >= 1 && x0 >= 1 && y0 + D_T + 1 <= H && x0 + D_T + 1 <= W) {       // workgroup-uniform
            const float* base = g + ((size_t)(y0 - 1) * W + (x0 - 1)) * 64 + chunk * G_KC;
#pragma unroll
            for (int i = 0; i < G_PER_T; ++i) stage[i] = *reinterpret_cast<const f32x4*>(base + off[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < G_PER_T; ++i) {
            const int u = tid + i * 256;
            const int pix = u / (G_KC / 4), q = u % (G_KC / 4);
            const int gy = y0 - 1 + pix / G_P, gxx = x0 - 1 + pix % G_P;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < G_UNITS && tile < ntiles && gy >= 0 && gy < H && gxx >= 0 && gxx < W)
                v = *reinterpret_cast<const f32x4*>(g + ((size_t)gy * W + gxx) * 64 + chunk * G_KC + q * 4);
            stage[i] = v;
        }
    };
    fetch(blockIdx.x, 0);

    // scales: the gradient's from its recorded absmax, the weights' from their own
    float sg, inv_g, sw, inv_w;
    pow2_scale(wave_max(amax_g[lane]), sg, inv_g);
    {
        unsigned m = 0;
        for (int i = tid; i < 9 * 64 * 4; i += 256) {
            const unsigned b = __float_as_uint(wd[i]) & 0x7FFFFFFFu;
            m = b > m ? b : m;
        }
        m = wave_max(m);
        if (lane == 0) wmax[wave] = m;
        __syncthreads();
        unsigned a = wmax[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) a = wmax[i] > a ? wmax[i] : a;
        pow2_scale(a, sw, inv_w);
    }
    // weight fragments, once per workgroup: column l15 = ky*3 + c (columns 9..15 are zero), k = (kx, co = 32 slice + 8 kg + j)
    f16x8 bh[3][2], bl[3][2];
    {
        const bool valid = l15 < 9;
        const int ky = valid ? l15 / 3 : 0, c = valid ? l15 % 3 : 0;
        const float swz = valid ? sw : 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int co = sl * G_KC + kg * 8 + j;
                    const float v = wd[((ky * 3 + kx) * 64 + co) * 4 + c] * swz;       // unconditional: the 48 loads overlap
                    const _Float16 hi = (_Float16)v;
                    bh[kx][sl][j] = hi;
                    bl[kx][sl][j] = (_Float16)((v - (float)hi) * G_LO_UP);
                }
    }
    const float inv = inv_g * inv_w;
    const size_t HW = (size_t)H * W;
    const unsigned char* abase = patch + ((wave * 4) * G_P + l15) * G_ROWB + kg * 16;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x4 am[6], ax[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) { am[t] = f32x4{0.f, 0.f, 0.f, 0.f}; ax[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int chunk = 0; chunk < 64 / G_KC; ++chunk) {
            __syncthreads();                 // the previous slice's fragments have been read
#pragma unroll
            for (int i = 0; i < G_PER_T; ++i) {
                const int u = tid + i * 256;
                if (u < G_UNITS) {
                    const f32x4 v = stage[i] * sg;
                    const f32x2 x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
                    const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
                    const f32x2 r01 = (x01 - __builtin_convertvector(h01, f32x2)) * G_LO_UP;
                    const f32x2 r23 = (x23 - __builtin_convertvector(h23, f32x2)) * G_LO_UP;
                    const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
                    unsigned char* row = patch + (u / (G_KC / 4)) * G_ROWB + (u % (G_KC / 4)) * 8;
                    *reinterpret_cast<u32x2*>(row) = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
                    *reinterpret_cast<u32x2*>(row + G_PIECEB) = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
                }
            }
            __syncthreads();
            if (chunk == 0) fetch(tile, 1);          // the next slice (of this tile, or of the workgroup's next tile) is
            else fetch(tile + gridDim.x, 0);         // in flight under this slice's MFMAs
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const unsigned char* a = abase + (t * G_P + kx) * G_ROWB;
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(a);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(a + G_PIECEB);
                    ax[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[kx][chunk], ax[t], 0, 0, 0);
                    am[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[kx][chunk], am[t], 0, 0, 0);
                    ax[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[kx][chunk], ax[t], 0, 0, 0);
                }
        }
        // row r of the wave: P[0] of patch row r (column c) + P[1] of patch row r + 1 (column 3 + c) + P[2] of r + 2 (6 + c)
        const int y0 = (tile / tiles_x) * D_T, x0 = (tile % tiles_x) * D_T;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + wave * 4 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v0 = am[r][i] + ax[r][i] * G_LO_DOWN;
                const float v1 = am[r + 1][i] + ax[r + 1][i] * G_LO_DOWN;
                const float v2 = am[r + 2][i] + ax[r + 2][i] * G_LO_DOWN;
                const float s = (v0 + __shfl(v1, lane + 3) + __shfl(v2, lane + 6)) * inv;
                const int x = x0 + kg * 4 + i;
                if (l15 < 3 && y < H && x < W) gx[l15 * HW + (size_t)y * W + x] = s;
            }
        }
    }
}

hipError_t launch_conv1_1_dgrad(const float* g, int H, int W, const float* wd, const unsigned* amax_g, float* gx,
                                hipStream_t stream) {
    const int blocks = ((H + D_T - 1) / D_T) * ((W + D_T - 1) / D_T);
    if (amax_g)     // two workgroups per CU, each walking tiles b, b + grid, ...: scales and weight fragments once per workgroup
        hipLaunchKernelGGL(conv1_1_dgrad_h2_kernel, dim3(blocks < 512 ? blocks : 512), dim3(256), 0, stream, g, H, W, wd, amax_g,
                           gx, blocks);
    else
        hipLaunchKernelGGL(conv1_1_dgrad_kernel, dim3(blocks), dim3(256), 0, stream, g, H, W, wd, gx);
    return hipGetLastError();
}

}  // namespace nst
