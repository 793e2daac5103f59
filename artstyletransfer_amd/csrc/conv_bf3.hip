// conv_bf3.hip - the 3x3 convolutions (forward and input gradient) on the gfx950 bf16 matrix pipe at
// fp32-level accuracy.  Every fp32 operand is cut into three bf16 pieces that add up to it EXACTLY
// (8 + 8 + 8 = 24 significand bits; each residual subtraction is exact in fp32):
//       a = a0 + a1 + a2,   b = b0 + b1 + b2,      |a_i| <= 2^-8i |a|
// and the product is taken as the six terms with i + j <= 2:
//       a*b ~= a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)        dropped: <= 3 * 2^-24 |a b|
// Each bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the
// result carries the same order of error as an fp32 FMA chain, while 6 bf16 MFMAs cost 6 x 32 = 192 cycles
// per 32x32x16 block against 8 x 64 = 512 for v_mfma_f32_32x32x2_f32: 2.67x the fp32-MFMA ceiling.
//
// Same implicit-GEMM structure as conv_mfma.hip (halo patch of a 32-channel chunk staged once per 9 taps,
// per-tap weight slice double buffered, issue-early / write-late), but one 512-thread workgroup per CU:
// 16x16 (or 32x16) pixels x 128 (64) output channels, 8 waves of 64x64.  Activations stay fp32 in HBM and
// are cut while being staged into LDS (8 VALU ops per element, once per 9 taps); the frozen weights are cut
// once on the host.  LDS rows hold the three pieces of 32 channels back to back (3 x 64 B) + 16 B pad = 208 B
// = 13 x 16 B: 16 consecutive rows start on 16 distinct 16-B slots -> conflict-free ds_read_b128 fragments.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int KC = 32;            // channels per chunk
constexpr int ROWB = 208;         // LDS row bytes: 3 pieces x 64 B + 16 B pad
constexpr int WROWB = 192;        // global weight row bytes per (tap, cout, chunk): 3 pieces x 32 bf16

// NBUF = 2: one 512-thread workgroup per CU, weight slices double buffered (one barrier per stage).
// NBUF = 1: 256-thread workgroups small enough (65 KB LDS) for TWO per CU; the weight slice is single
//           buffered (two barriers per stage), and the stalls of one workgroup - stage boundaries, halo-patch
//           refills, prologue, epilogue - are covered by the other workgroup's MFMAs on the same SIMDs.
template <int TH, int BN, int NBUF>
struct BfCfg {
    static constexpr int TW = 16;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int WM = TH / 4;            // waves along pixels (4 rows x 16 cols each)
    static constexpr int WN = BN / 64;
    static constexpr int NT = 64 * WM * WN;      // 512
    static constexpr int A_UNITS = PH * PW * (KC / 4);        // float4 units of the fp32 patch
    static constexpr int A_PER_T = (A_UNITS + NT - 1) / NT;
    static constexpr int B_UNITS = BN * (WROWB / 16);         // 16-byte units of the weight slice
    static constexpr int B_PER_T = (B_UNITS + NT - 1) / NT;
    // patch-row pitch rounded up to a multiple of 256 B: the two patch rows a 32-pixel MFMA tile spans then
    // start on the same 16-B slot, which makes every ds_read_b128 lane group hit 16 distinct slots
    static constexpr int PROWB = ((PW * ROWB + 255) / 256) * 256;
    static constexpr int A_BYTES = PH * PROWB;
    static constexpr int B_BYTES = BN * ROWB;
    static constexpr int LDS_BYTES = A_BYTES + NBUF * B_BYTES;
    static_assert(NT == 512 || NT == 256, "eight or four waves per workgroup");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// a = h + m + l exactly, each piece a bf16 held in the TOP half of a 32-bit word
__device__ __forceinline__ void cut3(float a, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(a) & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2);                      // <= 8 significant bits left: exact in bf16
}
// two top-half bf16 -> one packed dword {lo16 = first, hi16 = second}
__device__ __forceinline__ unsigned pack2(unsigned first, unsigned second) {
    return __builtin_amdgcn_perm(second, first, 0x07060302u);
}

}  // namespace

// the whole workgroup program; (sp, ct, split) = spatial tile, output-channel tile, K split of this workgroup
template <int TH, int BN, int NBUF>
__device__ __forceinline__ void conv_bf3_body(const ConvParams& p, const int sp, const int ct, const int split) {
    using C = BfCfg<TH, BN, NBUF>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ldsA = smem;
    unsigned char* ldsB = smem + C::A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % C::WM;
    const int wn = wave / C::WM;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int ty = sp / p.tiles_x;
    const int tx = sp - ty * p.tiles_x;
    const int y0 = ty * TH;
    const int x0 = tx * C::TW;
    const int n0 = ct * BN;

    f32x4 ra[C::A_PER_T];
    u32x4 rb[C::B_PER_T];

    // LDS destinations of this lane's staging units (fixed for the whole kernel)
    int a_lds[C::A_PER_T];
    int a_pix_ok[C::A_PER_T];        // pixel offset (y*W + x) inside the image, or -1 outside / unused
    int a_q[C::A_PER_T];
#pragma unroll
    for (int i = 0; i < C::A_PER_T; ++i) {
        const int u = tid + i * C::NT;
        const int pix = u >> 3;
        const int q = u & 7;
        const int pr = pix / C::PW;
        const int pc = pix - pr * C::PW;
        const int gy = y0 - 1 + pr;
        const int gx = x0 - 1 + pc;
        const bool ok = (u < C::A_UNITS) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_pix_ok[i] = ok ? gy * p.W + gx : -1;
        a_q[i] = q;
        a_lds[i] = (u < C::A_UNITS) ? pr * C::PROWB + pc * ROWB + q * 8 : -1;
    }
    int b_n[C::B_PER_T], b_q[C::B_PER_T], b_lds[C::B_PER_T];
#pragma unroll
    for (int i = 0; i < C::B_PER_T; ++i) {
        const int u = tid + i * C::NT;
        b_n[i] = u / 12;
        b_q[i] = u - b_n[i] * 12;
        b_lds[i] = (u < C::B_UNITS) ? b_n[i] * ROWB + b_q[i] * 16 : -1;
    }

    // cut the staged fp32 values into their three bf16 pieces and write 8 bytes per piece
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            if (a_lds[i] >= 0) {
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) cut3(ra[i][k], h[k], m[k], l[k]);
                unsigned char* row = ldsA + a_lds[i];
                *reinterpret_cast<u32x2*>(row) = u32x2{pack2(h[0], h[1]), pack2(h[2], h[3])};
                *reinterpret_cast<u32x2*>(row + 64) = u32x2{pack2(m[0], m[1]), pack2(m[2], m[3])};
                *reinterpret_cast<u32x2*>(row + 128) = u32x2{pack2(l[0], l[1]), pack2(l[2], l[3])};
            }
        }
    };
    auto store_b = [&](int buf) {
        unsigned char* dst = ldsB + buf * C::B_BYTES;
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<u32x4*>(dst + b_lds[i]) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragment addresses: lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r]
    const int prow = wm * 4 + (l31 >> 4);
    const int pcol = l31 & 15;
    const int a_off0 = (prow + 0) * C::PROWB + pcol * ROWB + half * 16;
    const int a_off1 = (prow + 2) * C::PROWB + pcol * ROWB + half * 16;
    const int b_off0 = (wn * 64 + l31) * ROWB + half * 16;
    const int b_off1 = (wn * 64 + 32 + l31) * ROWB + half * 16;
    int cur = 0;

    // One K source = (activation tensor, its weights, NTAPS taps).  Source 0 is the 3x3 conv proper
    // (9 taps); the optional source 1 is a 1x1 product with a second tensor of the same spatial size
    // accumulated into the same tile (the Gram backward dF = F * S riding on the input-gradient launch).
    // Loads are buffer loads: a 32-bit per-lane offset that does not change over the K loop plus a scalar
    // offset per chunk/tap; out-of-image pixels get an offset beyond the buffer and read as zeros (hardware
    // range check): no address arithmetic, 64-bit pointers or divergent branches inside the loop.
    auto run_source = [&](auto ntaps_c, const float* src, int cin, const void* wts, int cb, int ce) {
        constexpr int NTAPS = decltype(ntaps_c)::value;
        const int nch = cin / KC;
        const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(src), 0, (unsigned)((size_t)p.H * p.W * cin * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(wts), 0, (unsigned)((size_t)NTAPS * p.Cout * nch * WROWB), 0x00020000);
        unsigned a_voff[C::A_PER_T], b_voff[C::B_PER_T];
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i)
            a_voff[i] = (a_pix_ok[i] >= 0) ? (unsigned)((a_pix_ok[i] * cin + a_q[i] * 4) * 4) : 0xFFFFFF00u;
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i)
            b_voff[i] = (b_lds[i] >= 0) ? (unsigned)(b_n[i] * nch * WROWB + b_q[i] * 16) : 0xFFFFFF00u;
        auto load_a = [&](int chunk) {
            const int soff = chunk * KC * 4;
#pragma unroll
            for (int i = 0; i < C::A_PER_T; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, a_voff[i], soff, 0));
        };
        // weights: [tap][Cout][chunk][piece][32] bf16, i.e. 192 contiguous bytes per (tap, cout, chunk)
        auto load_b = [&](int chunk, int tap) {
            const int soff = ((tap * p.Cout + n0) * nch + chunk) * WROWB;
#pragma unroll
            for (int i = 0; i < C::B_PER_T; ++i)
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, b_voff[i], soff, 0);
        };

        load_a(cb);
        load_b(cb, 0);
        for (int c = cb; c < ce; ++c) {
            __syncthreads();     // every wave is done reading the previous patch (and B buffers of the last stage)
            store_a();
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                if (NBUF == 1 && t > 0) __syncthreads();     // single weight buffer: everyone is done reading it
                store_b(cur);
                __syncthreads();
                if (t + 1 < NTAPS) {
                    load_b(c, t + 1);
                } else if (c + 1 < ce) {
                    load_b(c + 1, 0);
                    load_a(c + 1);
                }
                // a 1-tap source reads the centre of the halo patch
                const int tap_off = (NTAPS == 9) ? (t / 3) * C::PROWB + (t % 3) * ROWB : C::PROWB + ROWB;
                const unsigned char* bsrc = ldsB + cur * C::B_BYTES;
                // both k-steps' fragments are requested before they are needed: the second set lands while
                // the first 24 MFMAs issue
                bf16x8 fa[2][2][3], fb[2][2][3];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        fa[ks][0][s] = *reinterpret_cast<const bf16x8*>(ldsA + a_off0 + tap_off + s * 64 + ks * 32);
                        fa[ks][1][s] = *reinterpret_cast<const bf16x8*>(ldsA + a_off1 + tap_off + s * 64 + ks * 32);
                        fb[ks][0][s] = *reinterpret_cast<const bf16x8*>(bsrc + b_off0 + s * 64 + ks * 32);
                        fb[ks][1][s] = *reinterpret_cast<const bf16x8*>(bsrc + b_off1 + s * 64 + ks * 32);
                    }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // smallest terms first; the six products of one (mt, nt) form one accumulation chain
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            f32x16 v = acc[mt][nt];
                            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt][2], fb[ks][nt][0], v, 0, 0, 0);
                            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt][0], fb[ks][nt][2], v, 0, 0, 0);
                            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt][1], fb[ks][nt][1], v, 0, 0, 0);
                            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt][1], fb[ks][nt][0], v, 0, 0, 0);
                            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt][0], fb[ks][nt][1], v, 0, 0, 0);
                            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mt][0], fb[ks][nt][0], v, 0, 0, 0);
                            acc[mt][nt] = v;
                        }
                }
                // pin the order: 12 fragment reads (k-step 0), then the k-step-1 reads interleaved one per two
                // MFMAs of k-step 0, then k-step 1's MFMAs (hipcc otherwise loads fragments just in time and
                // exposes the LDS latency several times per k-step)
                __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
                if (NBUF == 2) cur ^= 1;
            }
        }
    };

    const int nchunks_all = p.Cin / KC;
    const int cps = nchunks_all / p.ksplit;
    run_source(std::integral_constant<int, 9>{}, p.in, p.Cin, p.wt_bf, split * cps, split * cps + cps);
    if (p.in2 && split == 0) run_source(std::integral_constant<int, 1>{}, p.in2, p.Cin2, p.wt2_bf, 0, p.Cin2 / KC);

    // epilogue: D[m][n]: n = lane&31, m = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (p.ksplit > 1) {
        float* dst = p.partial + (size_t)split * p.H * p.W * p.Cout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = n0 + wn * 64 + nt * 32 + l31;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int y = y0 + wm * 4 + mt * 2 + (m >> 4);
                    const int x = x0 + (m & 15);
                    if (y < p.H && x < p.W) dst[((size_t)y * p.W + x) * p.Cout + co] = acc[mt][nt][r];
                }
        }
        return;
    }
    const int words = p.Cout >> 5;          // ReLU bit-mask words per pixel
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = n0 + wn * 64 + nt * 32 + l31;
        const int cw = (n0 + wn * 64 + nt * 32) >> 5;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int y = y0 + wm * 4 + mt * 2 + (m >> 4);
                const int x = x0 + (m & 15);
                const bool inb = (y < p.H && x < p.W);
                const size_t pix = (size_t)y * p.W + x;
                const size_t idx = pix * p.Cout + co;
                float v = acc[mt][nt][r] + bv;
                if (p.addend && inb) v += p.addend[idx];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.bits_in) {
                    // the 32 lanes of a half-wave read the same word: one bit per output channel
                    const unsigned wv = inb ? p.bits_in[pix * words + cw] : 0u;
                    v = ((wv >> l31) & 1u) ? v : 0.f;
                } else if (p.mask) {
                    v = (inb && p.mask[idx] > 0.f) ? v : 0.f;
                }
                if (p.bits_out) {
                    // ReLU mask of this output for the backward pass: bit = lane, one word per half-wave
                    const unsigned long long bal = __ballot(v > 0.f);
                    if (l31 == 0 && inb) p.bits_out[pix * words + cw] = half ? (unsigned)(bal >> 32) : (unsigned)bal;
                }
                acc[mt][nt][r] = v;
                if (inb) p.out[idx] = v;
            }
            if (p.pool_out) {
                // 2x2/2 max pool of the tile rows (2 mt, 2 mt + 1): the four window elements sit in this
                // lane's registers r, r+1, r+8, r+9
                const int py = (y0 + wm * 4 + mt * 2) >> 1;
                const int PH2 = p.H >> 1, PW2 = p.W >> 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 2 * j;
                    const float mx = fmaxf(fmaxf(acc[mt][nt][r], acc[mt][nt][r + 1]),
                                           fmaxf(acc[mt][nt][r + 8], acc[mt][nt][r + 9]));
                    const int mcol = (r & 3) + 8 * (r >> 2) + 4 * half;      // column of register r (row 0 of the pair)
                    const int px = (x0 + mcol) >> 1;
                    if (py < PH2 && px < PW2) p.pool_out[((size_t)py * PW2 + px) * p.Cout + co] = mx;
                }
            }
        }
    }
}

template <int TH, int BN, int NBUF>
__global__ __launch_bounds__(512, 2) void conv_bf3_kernel(ConvParams p) {
    const int n_ct = p.Cout / BN;
    conv_bf3_body<TH, BN, NBUF>(p, blockIdx.x / n_ct, blockIdx.x % n_ct, blockIdx.y);
}

// One launch = one layer over several images (the pyramid levels of a closure): same weights, same channel
// counts, each image with its own tensors and size.  Workgroups are numbered image by image, so the few
// tiles of the small levels fill the tail of the big level's grid instead of running as under-filled
// launches of their own.
template <int TH, int BN, int NBUF>
__global__ __launch_bounds__(512, 2) void conv_bf3_batch_kernel(ConvBatch b) {
    const int n_ct = b.Cout / BN;
    const int sp_all = blockIdx.x / n_ct;
    int i = 0;
    while (i + 1 < b.n && sp_all >= b.img[i].tile_end) ++i;
    const ConvImage& im = b.img[i];
    ConvParams p;
    p.in = im.in; p.wt = nullptr; p.wt_bf = b.wt_bf; p.bias = b.bias; p.addend = im.addend; p.mask = im.mask; p.out = im.out;
    p.H = im.H; p.W = im.W; p.Cin = b.Cin; p.Cout = b.Cout; p.relu = b.relu;
    p.tiles_x = im.tiles_x; p.tiles_y = 0; p.partial = nullptr; p.partial_floats = 0; p.ksplit = 1;
    p.in2 = im.in2; p.Cin2 = b.Cin2; p.wt2_bf = im.wt2_bf; p.bits_out = im.bits_out; p.bits_in = im.bits_in;
    p.pool_out = im.pool_out;
    conv_bf3_body<TH, BN, NBUF>(p, sp_all - (i ? b.img[i - 1].tile_end : 0), blockIdx.x % n_ct, 0);
}

template <int TH, int BN, int NBUF>
static hipError_t init_one() {
    constexpr int lds = BfCfg<TH, BN, NBUF>::LDS_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf3_kernel<TH, BN, NBUF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf3_batch_kernel<TH, BN, NBUF>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

hipError_t conv_bf3_init_device() {
    hipError_t e = init_one<16, 128, 2>();
    if (e == hipSuccess) e = init_one<8, 128, 1>();
    if (e == hipSuccess) e = init_one<32, 64, 2>();
    return e;
}

// tile choice for the 128-wide layers: one 512-thread workgroup per CU (default) or two 256-thread ones
// (NST_BF3_TILE=small; measured equal: the kernel is clock/power bound, not stall bound)
static bool small_tiles() {
    static const int v = [] { const char* e = getenv("NST_BF3_TILE"); return (e && e[0] == 's') ? 1 : 0; }();
    return v != 0;
}
static int tile_rows(int Cout) { return (Cout % 128 == 0) ? (small_tiles() ? 8 : 16) : 32; }
// a launch whose 512-thread tiles cannot fill 256 CUs twice over uses the 256-thread tile (twice as many
// workgroups, two per CU): conv5_1 of the whole pyramid is 54 big tiles x 4 channel tiles
static int batch_tile_rows(const ConvBatch& b) {
    int th = tile_rows(b.Cout);
    if (th != 16) return th;
    long blocks = 0;
    for (int i = 0; i < b.n; ++i) blocks += (long)((b.img[i].H + 15) / 16) * ((b.img[i].W + 15) / 16) * (b.Cout / 128);
    return blocks < 400 ? 8 : 16;
}

template <int TH, int BN, int NBUF>
static void launch_batch_cfg(const ConvBatch& b, int blocks, hipStream_t stream) {
    constexpr int lds = BfCfg<TH, BN, NBUF>::LDS_BYTES;
    constexpr int nt = BfCfg<TH, BN, NBUF>::NT;
    hipLaunchKernelGGL((conv_bf3_batch_kernel<TH, BN, NBUF>), dim3(blocks), dim3(nt), lds, stream, b);
}
template <int TH, int BN, int NBUF>
static void launch_single_cfg(const ConvParams& p, int blocks, hipStream_t stream) {
    constexpr int lds = BfCfg<TH, BN, NBUF>::LDS_BYTES;
    constexpr int nt = BfCfg<TH, BN, NBUF>::NT;
    hipLaunchKernelGGL((conv_bf3_kernel<TH, BN, NBUF>), dim3(blocks, p.ksplit), dim3(nt), lds, stream, p);
}

// fills tiles_x / tile_end of every image and launches one grid over all of them
hipError_t launch_conv_bf3_batch(const ConvBatch& b0, hipStream_t stream) {
    if (b0.n < 1 || b0.n > 8 || b0.Cin % 32 != 0 || b0.Cout % 64 != 0 || !b0.wt_bf) return hipErrorInvalidValue;
    ConvBatch b = b0;
    const bool wide = (b.Cout % 128 == 0);
    const int th = batch_tile_rows(b), bn = wide ? 128 : 64;
    int tiles = 0;
    for (int i = 0; i < b.n; ++i) {
        if ((size_t)b.img[i].H * b.img[i].W * b.Cin * 4 >= 0xFFFFFF00ull) return hipErrorInvalidValue;
        b.img[i].tiles_x = (b.img[i].W + 15) / 16;
        tiles += b.img[i].tiles_x * ((b.img[i].H + th - 1) / th);
        b.img[i].tile_end = tiles;
    }
    const int blocks = tiles * (b.Cout / bn);
    if (!wide) launch_batch_cfg<32, 64, 2>(b, blocks, stream);
    else if (th == 8) launch_batch_cfg<8, 128, 1>(b, blocks, stream);
    else launch_batch_cfg<16, 128, 2>(b, blocks, stream);
    return hipGetLastError();
}

// cost(S) = rounds of workgroups over the CU slots x chunks per workgroup.  Split the channel chunks when that
// removes a partly filled round (e.g. 384 tiles on 256 slots: 2 rounds of 16 chunks -> 3 rounds of 8); the
// ordered finish pass costs one extra read of S partial maps, so a split must save >= 12 %.
int conv_bf3_ksplit(int H, int W, int Cin, int Cout) {
    const bool wide = (Cout % 128 == 0);
    const int th = tile_rows(Cout), bn = wide ? 128 : 64;
    const long slots = (wide && th == 8) ? 512 : 256;
    const long blocks = (long)((H + th - 1) / th) * ((W + 15) / 16) * (Cout / bn);
    const int nchunks = Cin / 32;
    if (nchunks < 2) return 1;
    auto cost = [&](int S) { return (double)((blocks * S + slots - 1) / slots) * (nchunks / S); };
    double best = cost(1);
    for (int S = 2; S <= nchunks && S <= 16; S *= 2)
        if (nchunks % S == 0 && cost(S) < best) best = cost(S);
    if (best > 0.88 * cost(1)) return 1;
    for (int S = 2; S <= nchunks && S <= 16; S *= 2)
        if (nchunks % S == 0 && cost(S) <= best * 1.05) return S;
    return 1;
}

hipError_t launch_conv_bf3(const ConvParams& p0, hipStream_t stream) {
    if (p0.Cin % 32 != 0 || p0.Cout % 64 != 0 || !p0.wt_bf) return hipErrorInvalidValue;
    // 32-bit buffer offsets: the input tensor must stay below 4 GiB (true up to L=3; callers fall back to
    // conv_mfma.hip, which addresses with 64-bit pointers, beyond that)
    if ((size_t)p0.H * p0.W * p0.Cin * 4 >= 0xFFFFFF00ull) return hipErrorInvalidValue;
    ConvParams p = p0;
    p.ksplit = 1;
    if (p.partial) {
        const int S = conv_bf3_ksplit(p.H, p.W, p.Cin, p.Cout);
        if (S > 1 && (size_t)S * p.H * p.W * p.Cout <= p.partial_floats) p.ksplit = S;
    }
    const bool wide = (p.Cout % 128 == 0);
    const int th = tile_rows(p.Cout), bn = wide ? 128 : 64;
    p.tiles_x = (p.W + 15) / 16;
    p.tiles_y = (p.H + th - 1) / th;
    const int blocks = p.tiles_x * p.tiles_y * (p.Cout / bn);
    if (!wide) launch_single_cfg<32, 64, 2>(p, blocks, stream);
    else if (th == 8) launch_single_cfg<8, 128, 1>(p, blocks, stream);
    else launch_single_cfg<16, 128, 2>(p, blocks, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || p.ksplit == 1) return e;
    return launch_conv_splitk_finish(p, stream);
}

}  // namespace nst
