// conv_mfma.hip - 3x3 (pad 1, stride 1) and 1x1 convolution over NHWC fp32 activations as an
// implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered
// fmaf chain).  One kernel serves
//   * VGG19 conv+bias+ReLU forward           (neural_nets.py:54-65: torchvision features[i] conv/relu),
//   * the input-gradient ("dgrad") of the same convs with tap-flipped, transposed weights
//     (autograd of those convs, neural_style_transfer.py:193), epilogue = (+ loss gradient
//     already stored at the destination) * ReLU mask of the activation it flows into,
//   * the Gram backward dF = F * S as a 1x1 conv (math_utils.py:31 through autograd).
//
// GEMM view: M = pixels of a TH x 16 spatial tile, N = BN output channels, K = TAPS * Cin walked
// as (Cin chunk of KC) x (tap).  The KC-channel slice of the tile's (TH+2)x(18) halo patch is
// staged in LDS once and re-used by all 9 taps; the per-tap KC x BN weight slice is double
// buffered in LDS.  Next-stage operands are fetched global->VGPR while the current stage's MFMAs
// run and written to LDS after them (issue-early / write-late), one barrier per stage.
//
// LDS rows are KC+4 floats (KC=32: 144 B): 16 consecutive rows then start on 16 distinct 16-B
// slots of the 256-B bank row, so the ds_read_b128 operand reads are conflict-free.
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TH, int BN, int KC, int TAPS>
struct ConvCfg {
    static constexpr int TW = 16;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PH = TH + 2 * HALO;
    static constexpr int PW = TW + 2 * HALO;
    static constexpr int WM = TH / 4;           // waves along pixels (each 4 rows x 16 cols)
    static constexpr int WN = BN / 64;          // waves along output channels
    static constexpr int NT = 64 * WM * WN;     // threads
    static constexpr int AS = KC + 4;           // LDS row stride (floats) of the activation patch
    static constexpr int BS = KC + 4;           // LDS row stride (floats) of the weight slice
    static constexpr int Q = KC / 4;            // 16-byte units per row
    static constexpr int A_UNITS = PH * PW * Q;
    static constexpr int A_PER_T = (A_UNITS + NT - 1) / NT;
    static constexpr int B_UNITS = BN * Q;
    static constexpr int B_PER_T = B_UNITS / NT;
    static constexpr int A_FLOATS = PH * PW * AS;
    static constexpr int B_FLOATS = BN * BS;
    static constexpr int LDS_BYTES = (A_FLOATS + 2 * B_FLOATS) * 4;
    static_assert(B_UNITS % NT == 0, "weight slice must divide evenly over the threads");
    static_assert(TH % 4 == 0 && BN % 64 == 0 && KC % 8 == 0, "tile shape");
};

template <int TH, int BN, int KC, int TAPS>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
    using C = ConvCfg<TH, BN, KC, TAPS>;
    static_assert(C::NT == 256, "every tile shape uses four waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ldsA = smem;
    float* ldsB = smem + C::A_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % C::WM;
    const int wn = wave / C::WM;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // block -> (channel tile, spatial tile); consecutive blocks share the activation patch
    const int n_ct = p.Cout / BN;
    const int bid = blockIdx.x;
    const int ct = bid % n_ct;
    const int sp = bid / n_ct;
    const int ty = sp / p.tiles_x;
    const int tx = sp - ty * p.tiles_x;
    const int y0 = ty * TH;
    const int x0 = tx * C::TW;
    const int n0 = ct * BN;

    // split-K: blockIdx.y owns the channel chunks [c_begin, c_end)
    const int nchunks_all = p.Cin / KC;
    const int cps = nchunks_all / p.ksplit;
    const int c_begin = blockIdx.y * cps;
    const int c_end = c_begin + cps;

    f32x4 ra[C::A_PER_T];
    f32x4 rb[C::B_PER_T];

    auto load_a = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            const int u = tid + i * C::NT;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (u < C::A_UNITS) {
                const int pix = u / C::Q;
                const int q = u - pix * C::Q;
                const int pr = pix / C::PW;
                const int pc = pix - pr * C::PW;
                const int gy = y0 - C::HALO + pr;
                const int gx = x0 - C::HALO + pc;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    const float* src = p.in + ((size_t)gy * p.W + gx) * p.Cin + chunk * KC + q * 4;
                    v = *reinterpret_cast<const f32x4*>(src);
                }
            }
            ra[i] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            const int u = tid + i * C::NT;
            if (u < C::A_UNITS) {
                const int pix = u / C::Q;
                const int q = u - pix * C::Q;
                *reinterpret_cast<f32x4*>(ldsA + pix * C::AS + q * 4) = ra[i];
            }
        }
    };
    auto load_b = [&](int chunk, int tap) {
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i) {
            const int u = tid + i * C::NT;
            const int n = u / C::Q;
            const int q = u - n * C::Q;
            const float* src = p.wt + ((size_t)tap * p.Cout + n0 + n) * p.Cin + chunk * KC + q * 4;
            rb[i] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto store_b = [&](int buf) {
        float* dst = ldsB + buf * C::B_FLOATS;
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i) {
            const int u = tid + i * C::NT;
            const int n = u / C::Q;
            const int q = u - n * C::Q;
            *reinterpret_cast<f32x4*>(dst + n * C::BS + q * 4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // this lane's two A rows (pixel of M-tile mt) inside the patch, tap (0,0), and its B row
    const int prow = wm * 4 + (l31 >> 4);
    const int pcol = l31 & 15;
    const int a_off0 = ((prow + 0) * C::PW + pcol) * C::AS + 4 * half;
    const int a_off1 = ((prow + 2) * C::PW + pcol) * C::AS + 4 * half;
    const int b_off0 = (wn * 64 + l31) * C::BS + 4 * half;
    const int b_off1 = (wn * 64 + 32 + l31) * C::BS + 4 * half;

    load_a(c_begin);
    load_b(c_begin, 0);
    int cur = 0;
    for (int c = c_begin; c < c_end; ++c) {
        if (c > c_begin) __syncthreads();     // every wave is done reading the previous patch
        store_a();
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            store_b(cur);
            __syncthreads();
            if (t + 1 < TAPS) {
                load_b(c, t + 1);
            } else if (c + 1 < c_end) {
                load_b(c + 1, 0);
                load_a(c + 1);
            }
            const int dy = (TAPS == 9) ? t / 3 : 0;
            const int dx = (TAPS == 9) ? t % 3 : 0;
            const int tap_off = (dy * C::PW + dx) * C::AS;
            const float* bsrc = ldsB + cur * C::B_FLOATS;
#pragma unroll
            for (int ks = 0; ks < KC / 8; ++ks) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(ldsA + a_off0 + tap_off + ks * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(ldsA + a_off1 + tap_off + ks * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bsrc + b_off0 + ks * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bsrc + b_off1 + ks * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
                }
            }
            cur ^= 1;
        }
    }

    // epilogue: D[m][n]: n = lane&31, m = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (p.ksplit > 1) {
        // raw partial sums; conv_splitk_finish_kernel adds them in split order and applies the epilogue
        float* dst = p.partial + (size_t)blockIdx.y * p.H * p.W * p.Cout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int y = y0 + wm * 4 + mt * 2 + (m >> 4);
                    const int x = x0 + (m & 15);
                    if (y < p.H && x < p.W) dst[((size_t)y * p.W + x) * p.Cout + co] = acc[mt][nt][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = n0 + wn * 64 + nt * 32 + l31;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int y = y0 + wm * 4 + mt * 2 + (m >> 4);
                const int x = x0 + (m & 15);
                if (y < p.H && x < p.W) {
                    const size_t idx = ((size_t)y * p.W + x) * p.Cout + co;
                    float v = acc[mt][nt][r] + bv;
                    if (p.addend) v += p.addend[idx];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.mask) v = (p.mask[idx] > 0.f) ? v : 0.f;
                    p.out[idx] = v;
                }
            }
        }
    }
}

// out = sum_s partial[s] (+ bias) (+ addend) (ReLU) (mask), 16 bytes per lane
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(ConvParams p) {
    const size_t n4 = (size_t)p.H * p.W * p.Cout / 4;
    const int c4 = p.Cout / 4;
    const f32x4* part = reinterpret_cast<const f32x4*>(p.partial);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = part[i];
        for (int s = 1; s < p.ksplit; ++s) {
            const f32x4 t = part[(size_t)s * n4 + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += t[k];
        }
        if (p.bias) {
            const f32x4 b = reinterpret_cast<const f32x4*>(p.bias)[i % c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += b[k];
        }
        if (p.addend) {
            const f32x4 a = reinterpret_cast<const f32x4*>(p.addend)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += a[k];
        }
        if (p.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (p.mask) {
            const f32x4 m = reinterpret_cast<const f32x4*>(p.mask)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (m[k] > 0.f) ? v[k] : 0.f;
        }
        reinterpret_cast<f32x4*>(p.out)[i] = v;
    }
}

hipError_t launch_conv_splitk_finish(const ConvParams& p, hipStream_t stream) {
    const size_t n4 = (size_t)p.H * p.W * p.Cout / 4;
    size_t fb = (n4 + 255) / 256;
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((int)fb), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Split the channel chunks over S blocks when the (spatial x channel) tiles alone cannot fill the chip.
// cost(S) = waves of blocks over the 256 CUs x chunks per block; smallest S within 5 % of the best.
int conv_ksplit(int H, int W, int Cin, int Cout) {
    const bool wide = (Cout % 128 == 0);
    const int th = wide ? 8 : 16, bn = wide ? 128 : 64;
    const long blocks = (long)((H + th - 1) / th) * ((W + 15) / 16) * (Cout / bn);
    const int nchunks = Cin / 32;
    if (blocks >= 384 || nchunks < 2) return 1;
    double best = 1e30;
    for (int S = 1; S <= nchunks && S <= 16; S *= 2) {
        if (nchunks % S) continue;
        const double c = (double)((blocks * S + 255) / 256) * (nchunks / S);
        if (c < best) best = c;
    }
    for (int S = 1; S <= nchunks && S <= 16; S *= 2) {
        if (nchunks % S) continue;
        const double c = (double)((blocks * S + 255) / 256) * (nchunks / S);
        if (c <= best * 1.05) return S;
    }
    return 1;
}

template <int TH, int BN, int KC, int TAPS>
static hipError_t launch_cfg(const ConvParams& p0, hipStream_t stream) {
    using C = ConvCfg<TH, BN, KC, TAPS>;
    ConvParams p = p0;
    p.tiles_x = (p.W + C::TW - 1) / C::TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    if (p.ksplit < 1) p.ksplit = 1;
    const int blocks = p.tiles_x * p.tiles_y * (p.Cout / BN);
    hipLaunchKernelGGL((conv_mfma_kernel<TH, BN, KC, TAPS>), dim3(blocks, p.ksplit), dim3(C::NT), C::LDS_BYTES, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit == 1) return e;
    return launch_conv_splitk_finish(p, stream);
}

template <int TH, int BN, int KC, int TAPS>
static hipError_t init_cfg() {
    using C = ConvCfg<TH, BN, KC, TAPS>;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<TH, BN, KC, TAPS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
}

// once per device, before the first launch there: raise the dynamic-LDS limit of every instance
hipError_t conv_mfma_init_device() {
    hipError_t e;
    if ((e = init_cfg<8, 128, 32, 9>()) != hipSuccess) return e;
    if ((e = init_cfg<16, 64, 32, 9>()) != hipSuccess) return e;
    if ((e = init_cfg<8, 128, 32, 1>()) != hipSuccess) return e;
    if ((e = init_cfg<16, 64, 32, 1>()) != hipSuccess) return e;
    return hipSuccess;
}

// Shapes must satisfy Cin % 32 == 0 and Cout % 64 == 0 (all VGG19 layers but conv1_1).
hipError_t launch_conv_mfma(const ConvParams& p0, int taps, hipStream_t stream) {
    if (p0.Cin % 32 != 0 || p0.Cout % 64 != 0 || (taps != 9 && taps != 1)) return hipErrorInvalidValue;
    ConvParams p = p0;
    p.ksplit = 1;
    if (taps == 9 && p.partial) {
        const int S = conv_ksplit(p.H, p.W, p.Cin, p.Cout);
        if (S > 1 && (size_t)S * p.H * p.W * p.Cout <= p.partial_floats) p.ksplit = S;
    }
    const bool wide = (p.Cout % 128 == 0);
    if (taps == 9) {
        if (wide) return launch_cfg<8, 128, 32, 9>(p, stream);
        return launch_cfg<16, 64, 32, 9>(p, stream);
    }
    if (wide) return launch_cfg<8, 128, 32, 1>(p, stream);
    return launch_cfg<16, 64, 32, 1>(p, stream);
}

}  // namespace nst
