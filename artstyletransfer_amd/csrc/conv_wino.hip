// conv_wino.hip - the 3x3 convolution (forward: bias + ReLU + ReLU bit mask + 2x2 max-pool with its arg-max code; input
// gradient: loss-gradient addend + ReLU mask of the map below, un-pooling loader) as a 1-D Winograd F(2,3) along x on the
// fp16 matrix pipe, in the f16x2 arithmetic of conv_h2.hip: 1.5x fewer MFMAs per output.  Default (nst_options.h2_winograd)
// for the launches with Cin >= 256, Cout a multiple of 128 and no second (Gram) source: 14 of the 24 conv launches of a closure.
//
// For an output pair (x = 2p, 2p + 1) of a row and the input columns d0..d3 = x - 1 .. x + 2 (one tap row ky):
//     t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3                    (input transform, fp32, before the cut)
//     u0 = g0, u1 = (g0 + g1 + g2) / 2, u2 = (g0 - g1 + g2) / 2, u3 = g2        (filter transform, host, fp64)
//     m_xi = sum over ky and input channels of t_xi u_xi;   y(2p) = m0 + m1 + m2,   y(2p + 1) = m1 - m2 - m3
// so a stage (ky, 32 input channels) is four GEMMs [pairs x 32] x [32 x Cout], one per xi, over HALF as many rows as
// there are pixels: 12 of them per chunk instead of 9 over all pixels.
//
// Workgroup: 512 threads, 8 rows x 16 columns of pixels = 64 pairs, 128 output channels.  Wave w owns xi = w & 3 and the
// output-channel half wn = w >> 2: a 64 x 64 wave tile of m_xi on v_mfma_f32_32x32x16_f16 with main and cross
// accumulators, exactly conv_h2.hip's products.  Its A fragments (the transformed patch rows of its xi) come from LDS;
// its B fragments belong to a weight slice no other wave shares beyond the other channel half, so they are loaded
// L2 -> registers in fragment order (host layout below), two k-steps ahead; LDS holds only the double-buffered patch
// and there is ONE barrier per chunk (3 stages).  The patch is fetched ONCE: a staging task loads its pair's own two
// pixels, takes the transform's outer columns from the neighbouring tasks' registers (ds_bpermute) and only the end pairs
// load a halo column (round 3: 18 pixel columns per row instead of 32, and the un-pooling loader one pooled pixel per pair
// instead of four).  The four xi accumulators of a pair live in four waves: the epilogue meets them in LDS, applies the
// output transform, bias and ReLU, and stores 16-byte pieces.
// Where a workgroup's time goes and what each stream of the K loop costs: profiles/r03_wino_phase_stamps.txt,
// profiles/r03_wino_kloop_ablations.txt (tools/micro/wino_probe.hip).
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Diagnostic build only (-DNST_WINO_STAMPS, tools/micro/wino_probe.hip): wave 0 of every workgroup leaves the shader clock at the
// phase boundaries in a buffer of its own that nothing else reads.  The shipped library is built without it.
#ifdef NST_WINO_STAMPS
__device__ unsigned long long g_wino_stamps[(1 << 14) * 8];
#define WSTAMP(k)                                                                                      \
    do {                                                                                               \
        if (tid == 0 && blockIdx.x < (1 << 14)) g_wino_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define WSTAMP_REAL(k)                                                                                 \
    do {                                                                                               \
        if (tid == 0 && blockIdx.x < (1 << 14)) g_wino_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define WSTAMP(k)
#define WSTAMP_REAL(k)
#endif

namespace {

constexpr int W_TH = 8, W_TW = 16, W_PAIRS = 8, W_PH = W_TH + 2;
constexpr int W_ROWB = 144;                              // 32 channels x 2 pieces x 2 B + 16 B pad
constexpr int W_PROWB = 4 * W_PAIRS * W_ROWB + 128;      // a patch row: [xi][pair] entries; + 128 B: rows y and y + 1 of a fragment
                                                         // (16 lanes = 2 rows x 8 pairs) land on complementary 16-byte slots
constexpr int W_A_BYTES = W_PH * W_PROWB;
constexpr int W_E_BYTES = 4 * 64 * 128 * 4;              // epilogue: m[xi][pair row][128 channels] fp32
constexpr int W_BITS_OFF = W_E_BYTES;                    // + 128 pixels x 4 ReLU-mask words
constexpr int W_CODE_OFF = W_BITS_OFF + 128 * 4 * 4;     // + 32 pooled pixels x 4 words x 4 window positions
constexpr int W_LDS = W_CODE_OFF + 32 * 4 * 4 * 4;       // 135 KB (the K loop uses the first 2 x W_A_BYTES = 92.5 KB of it)
static_assert(2 * W_A_BYTES <= W_LDS && W_LDS <= 160 * 1024, "LDS budget");
constexpr float LO_UP = 2048.f, LO_DOWN = 1.f / 2048.f;

__device__ __forceinline__ void cut2x4(const f32x4 v, const float s, u32x2& hi, u32x2& lo) {
    const f32x2 x01 = {v[0] * s, v[1] * s}, x23 = {v[2] * s, v[3] * s};
    const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
    const f32x2 b01 = __builtin_convertvector(h01, f32x2), b23 = __builtin_convertvector(h23, f32x2);
    const f32x2 r01 = (x01 - b01) * LO_UP, r23 = (x23 - b23) * LO_UP;
    const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
    hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}

// OR over the 8 lanes of an aligned 8-lane group (every lane gets the result): two quad permutes and a half-row mirror on
// the DPP path - no LDS traffic, where eight lanes' atomicOr into one LDS word serialised.
__device__ __forceinline__ unsigned or8(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);     // row_half_mirror: lane i <- lane 7 - i
    return v;
}

__device__ __forceinline__ int xcd_order(const int b, const int grid) {
    const int x = b & 7, k = b >> 3;
    const int q = grid >> 3, r = grid & 7;
    return x * q + (x < r ? x : r) + k;
}

}  // namespace

// UNPOOL: every image's `in` is the gradient w.r.t. a 2x2-pooled map; the loader un-pools it through the arg-max code words
// (conv_h2.hip's rule: [pooled pixel][32-channel group][window position], bit = channel & 31).
// MODE: what the epilogue may meet - 1: a forward launch (bias + ReLU, ReLU bit mask, pooled map + arg-max code; no addend, no
// incoming mask), 2: an input-gradient launch (addend, ReLU mask of the map below; nothing of the former), 0: anything.  The
// launcher picks 1 or 2 where the launch fits: the epilogue is bound by its vector instructions (profiles/
// r03_wino_phase_stamps.txt: 8 k cycles of an 80 k-cycle tile), and the specialised forms shed the other form's selects.
template <bool UNPOOL, int MODE>
__global__ __launch_bounds__(512, 2) void conv_wino_batch_kernel(ConvBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave & 3, wn = wave >> 2;
    const int half = lane >> 5, l31 = lane & 31;
    WSTAMP_REAL(6);
    WSTAMP(0);

    // tile of this workgroup
    const int n_ct = b.Cout >> 7;
    const int t = xcd_order((int)blockIdx.x, (int)gridDim.x);
    // Which (spatial tile, output-channel tile) the t-th workgroup of the XCD-contiguous order takes.  Up to 256 output
    // channels the layer's whole weight image (3.1 MB) stays in an XCD's L2 beside the patch stream: the n_ct channel tiles
    // of a patch are neighbours in t and share its lines.  A 512-channel layer's image is 12.6 MB - walked by every XCD
    // in every round of workgroups it comes back from the Infinity Cache each time (0.4 GB per launch) - so there the
    // order is channel-tile major: an XCD's share of t stays on ONE 3.1-MB slice (the patch is then fetched by n_ct XCDs:
    // 70 MB per launch).  Speed: the same (profiles/r03_wino_kloop_ablations.txt); fabric traffic: see DESIGN 6.
    int sp_all, ct;
    if (n_ct >= 4) {
        const int n_sp = (int)gridDim.x / n_ct;
        ct = t / n_sp;
        sp_all = t - ct * n_sp;
    } else {
        sp_all = t / n_ct;
        ct = t - sp_all * n_ct;
    }
    int ii = 0;
    while (ii + 1 < b.n && sp_all >= b.img[ii].tile_end) ++ii;
    ii = __builtin_amdgcn_readfirstlane(ii);      // (provably wave-uniform: the image's pointers then live in scalar registers)
    const ConvImage& im = b.img[ii];
    const int sp = sp_all - (ii ? b.img[ii - 1].tile_end : 0);
    const int H = im.H, W = im.W, Cin = b.Cin, Cout = b.Cout;
    const int ty = sp / im.tiles_x, tx = sp - ty * im.tiles_x;
    const int y0 = ty * W_TH, x0 = tx * W_TW, n0 = ct * 128;
    const int nch = Cin >> 5;

    // operand scale: the transformed values reach twice the tensor maximum, so one binade below conv_h2's scale
    float sa, ia;
    {
        unsigned m = im.amax_in[lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)m, off);
            m = o > m ? o : m;
        }
        int e = (int)((m >> 23) & 0xFFu);
        e = __builtin_amdgcn_readfirstlane(e < 32 ? 32 : (e > 250 ? 250 : e));      // (the same in every lane: scalar registers)
        sa = __uint_as_float((unsigned)(267 - e) << 23);      // 2^(13 - (e - 127))
        ia = __uint_as_float((unsigned)(e - 13) << 23);
    }
    const float inv = ia * b.wt_wino_inv;

    const int PH2 = H >> 1, PW2 = W >> 1;
    const size_t in_px = UNPOOL ? (size_t)PH2 * PW2 : (size_t)H * W;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(im.in), 0, (unsigned)(in_px * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_code = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned*>(UNPOOL ? im.pcode_in : nullptr), 0, UNPOOL ? (unsigned)(in_px * (Cin >> 5) * 16) : 0u, 0x00020000);
    // (the weight image through a buffer descriptor: per-lane offset lane * 16, everything else scalar or immediate)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(b.wt_wino), 0, (unsigned)((size_t)(Cout >> 7) * nch * 3 * 8 * 2 * 4 * 64 * 16), 0x00020000);

    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[a][c][r] = 0.f; accx[a][c][r] = 0.f; }

    // ---- patch staging.  Task (patch row, output pair p, channel quad): the pair's own two pixels E = (x0 + 2p), O = (x0 + 2p + 1)
    // are loaded ONCE; the transform's outer columns d0 = O of pair p - 1 and d3 = E of pair p + 1 come from the neighbouring
    // tasks' registers (ds_bpermute: a wave holds one patch row, 8 pairs x 8 quads), and the two halo columns x0 - 1, x0 + 16
    // from one more load that only the end pairs' lanes take part in: 18 pixel columns fetched per row instead of 32.
    // Lane -> pair: the 8-lane groups hold pairs 0, 4, 1, 5, 2, 6, 3, 7, so that the 16 lanes one ds_write_b64 pass serves
    // write rows 4 pairs = 576 B = 16 banks apart: conflict-free (neighbouring pairs, 144 B apart, collide on 12 of 16 banks).
    auto pair_of = [](int l) { const int g = (l >> 3) & 7; return (g >> 1) + 4 * (g & 1); };
    auto lane_of = [](int pr, int sub) { return ((2 * (pr & 3) + ((pr >> 2) & 1)) * 8 + sub) * 4; };      // byte address for ds_bpermute
    constexpr unsigned OOB = 0xFFFFFF00u;
    struct Stage { f32x4 e, o, h; unsigned ce[UNPOOL ? 2 : 1], ch[1]; };       // UNPOOL: e = the pooled pixel both E and O come from
    auto task_load = [&](Stage& st, int u, int chunk) {
        const int row = u >> 6, quad = u & 7, pair = pair_of(u);
        const int gy = y0 - 1 + row, gx = x0 + 2 * pair;
        const int hx = pair == 0 ? x0 - 1 : x0 + 16;                        // halo column of the end pairs
        const bool hsel = (pair == 0) | (pair == 7);
        if (!UNPOOL) {
            const bool rowok = (unsigned)gy < (unsigned)H;
            const unsigned base = ((unsigned)(gy * W + gx) * (unsigned)Cin + (unsigned)quad * 4u) * 4u;
            const unsigned hoff = ((unsigned)(gy * W + hx) * (unsigned)Cin + (unsigned)quad * 4u) * 4u;
            st.e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (rowok & (gx < W)) ? base : OOB, chunk * 128, 0));
            st.o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (rowok & (gx + 1 < W)) ? base + (unsigned)Cin * 4u : OOB, chunk * 128, 0));
            st.h = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (rowok & hsel & ((unsigned)hx < (unsigned)W)) ? hoff : OOB, chunk * 128, 0));
        } else {
            // the odd last row / column belongs to no window
            const bool rowok = ((unsigned)gy < (unsigned)H) & ((gy >> 1) < PH2);
            const int pxl = (x0 >> 1) + pair, phx = hx >> 1;                 // pooled columns (x0 is a multiple of 16)
            const bool ok = rowok & (pxl < PW2), hok = rowok & hsel & (hx >= 0) & (phx < PW2);
            const unsigned pp = (unsigned)((gy >> 1) * PW2 + pxl), hp = (unsigned)((gy >> 1) * PW2 + phx);
            const unsigned pos = (unsigned)(gy & 1) * 2u;                    // window position of E; O is the next one
            st.e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? (pp * (unsigned)Cin + (unsigned)quad * 4u) * 4u : OOB, chunk * 128, 0));
            const u32x2 c2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_code, ok ? ((pp * (unsigned)(Cin >> 5)) * 4u + pos) * 4u : OOB, chunk * 16, 0));
            st.ce[0] = c2[0];
            st.ce[UNPOOL ? 1 : 0] = c2[1];
            st.h = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, hok ? (hp * (unsigned)Cin + (unsigned)quad * 4u) * 4u : OOB, chunk * 128, 0));
            st.ch[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_code, hok ? ((hp * (unsigned)(Cin >> 5)) * 4u + pos + (unsigned)(hx & 1)) * 4u : OOB, chunk * 16, 0);
        }
    };
    auto task_store = [&](const Stage& st, int u, unsigned char* buf) {
        const int row = u >> 6, quad = u & 7, pair = pair_of(u);
        f32x4 E = st.e, O = st.o, Hh = st.h;
        if (UNPOOL) {
            // an element of the pooled gradient goes to the window position that held the (first, positive) maximum
            const unsigned be = st.ce[0] >> (quad * 4), bo = st.ce[UNPOOL ? 1 : 0] >> (quad * 4), bh = st.ch[0] >> (quad * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                O[k] = ((bo >> k) & 1u) ? st.e[k] : 0.f;
                E[k] = ((be >> k) & 1u) ? st.e[k] : 0.f;
                Hh[k] = ((bh >> k) & 1u) ? st.h[k] : 0.f;
            }
        }
        const int src0 = lane_of(pair - 1, quad), src3 = lane_of(pair + 1, quad);
        f32x4 d0, d3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d0[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src0, __float_as_int(O[k])));
            d3[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(src3, __float_as_int(E[k])));
        }
        if (pair == 0) d0 = Hh;
        if (pair == 7) d3 = Hh;
        const f32x4 tt[4] = {d0 - O, E + O, O - E, E - d3};
        unsigned char* base = buf + row * W_PROWB + pair * W_ROWB + quad * 8;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            u32x2 hi, lo;
            cut2x4(tt[x], sa, hi, lo);
            *reinterpret_cast<u32x2*>(base + x * W_PAIRS * W_ROWB) = hi;
            *reinterpret_cast<u32x2*>(base + x * W_PAIRS * W_ROWB + 64) = lo;
        }
    };

    // The last two patch rows (8 pairs x 32 channels each = 512 one-channel tasks) are every thread's second task: 4-byte
    // loads, one channel through the same exchange, transform and cut.  (As 128 more four-channel tasks they kept 12 staging
    // registers of EVERY thread busy for two waves' sake.)  Wave w: patch row 8 + (w >> 2), channels 8 (w & 3) .. + 8; its
    // lanes: the same pair order as above, channel = lane & 7.
    struct Stage1 { float e, o, h; unsigned ce[UNPOOL ? 2 : 1], ch[1]; };
    auto row_load = [&](Stage1& st, int u, int chunk) {
        const int wv = u >> 6, row = 8 + (wv >> 2), ch = (wv & 3) * 8 + (u & 7), pair = pair_of(u);
        const int gy = y0 - 1 + row, gx = x0 + 2 * pair;
        const int hx = pair == 0 ? x0 - 1 : x0 + 16;
        const bool hsel = (pair == 0) | (pair == 7);
        if (!UNPOOL) {
            const bool rowok = (unsigned)gy < (unsigned)H;
            const unsigned base = ((unsigned)(gy * W + gx) * (unsigned)Cin + (unsigned)ch) * 4u;
            const unsigned hoff = ((unsigned)(gy * W + hx) * (unsigned)Cin + (unsigned)ch) * 4u;
            st.e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, (rowok & (gx < W)) ? base : OOB, chunk * 128, 0));
            st.o = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, (rowok & (gx + 1 < W)) ? base + (unsigned)Cin * 4u : OOB, chunk * 128, 0));
            st.h = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, (rowok & hsel & ((unsigned)hx < (unsigned)W)) ? hoff : OOB, chunk * 128, 0));
        } else {
            const bool rowok = ((unsigned)gy < (unsigned)H) & ((gy >> 1) < PH2);
            const int pxl = (x0 >> 1) + pair, phx = hx >> 1;
            const bool ok = rowok & (pxl < PW2), hok = rowok & hsel & (hx >= 0) & (phx < PW2);
            const unsigned pp = (unsigned)((gy >> 1) * PW2 + pxl), hp = (unsigned)((gy >> 1) * PW2 + phx);
            const unsigned pos = (unsigned)(gy & 1) * 2u;
            st.e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, ok ? (pp * (unsigned)Cin + (unsigned)ch) * 4u : OOB, chunk * 128, 0));
            const u32x2 c2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_code, ok ? ((pp * (unsigned)(Cin >> 5)) * 4u + pos) * 4u : OOB, chunk * 16, 0));
            st.ce[0] = c2[0];
            st.ce[UNPOOL ? 1 : 0] = c2[1];
            st.h = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, hok ? (hp * (unsigned)Cin + (unsigned)ch) * 4u : OOB, chunk * 128, 0));
            st.ch[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_code, hok ? ((hp * (unsigned)(Cin >> 5)) * 4u + pos + (unsigned)(hx & 1)) * 4u : OOB, chunk * 16, 0);
        }
    };
    auto row_store = [&](const Stage1& st, int u, unsigned char* buf) {
        const int wv = u >> 6, row = 8 + (wv >> 2), ch = (wv & 3) * 8 + (u & 7), pair = pair_of(u);
        float E = st.e, O = st.o, Hh = st.h;
        if (UNPOOL) {
            O = ((st.ce[UNPOOL ? 1 : 0] >> ch) & 1u) ? st.e : 0.f;
            E = ((st.ce[0] >> ch) & 1u) ? st.e : 0.f;
            Hh = ((st.ch[0] >> ch) & 1u) ? st.h : 0.f;
        }
        float d0 = __int_as_float(__builtin_amdgcn_ds_bpermute(lane_of(pair - 1, u & 7), __float_as_int(O)));
        float d3 = __int_as_float(__builtin_amdgcn_ds_bpermute(lane_of(pair + 1, u & 7), __float_as_int(E)));
        if (pair == 0) d0 = Hh;
        if (pair == 7) d3 = Hh;
        const float tt[4] = {d0 - O, E + O, O - E, E - d3};
        unsigned char* base = buf + row * W_PROWB + pair * W_ROWB + ch * 2;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float v = tt[x] * sa;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)((v - (float)hi) * LO_UP);
            *reinterpret_cast<_Float16*>(base + x * W_PAIRS * W_ROWB) = hi;
            *reinterpret_cast<_Float16*>(base + x * W_PAIRS * W_ROWB + 64) = lo;
        }
    };

    // ---- fragments
    // A: lane (r = l31, h = half) holds A[pair row r of the m tile][k = 8 h + j]; pair row = 4 image rows x 8 pairs
    const int a_base = ((l31 >> 3)) * W_PROWB + (xi * W_PAIRS + (l31 & 7)) * W_ROWB + half * 16;
    struct AF { f16x8 v[2][2]; };       // [m tile][piece]
    auto read_a = [&](AF& f, const unsigned char* buf, int ky, int ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                f.v[mt][s] = *reinterpret_cast<const f16x8*>(buf + a_base + (mt * 4 + ky) * W_PROWB + s * 64 + ks * 32);
    };
    // B: 16-byte units [ct][chunk][ky][wave][ks][nt][piece][lane]
    struct BF { f16x8 v[2][2]; };       // [n tile][piece]
    auto load_b = [&](BF& f, int chunk, int ky, int ks) {
        const int soff = (((((ct * nch + chunk) * 3 + ky) * 8 + wave) * 2 + ks) * 4 * 64) * 16;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                f.v[nt][s] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16 + (nt * 2 + s) * 64 * 16, soff, 0));
    };
    auto multiply = [&](const AF& a, const BF& w) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[mt][1], w.v[nt][0], accx[mt][nt], 0, 0, 0);
                accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[mt][0], w.v[nt][0], accm[mt][nt], 0, 0, 0);
                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v[mt][0], w.v[nt][1], accx[mt][nt], 0, 0, 0);
            }
    };

    // ---- prologue: chunk 0 into buffer 0, the first two k-steps' weights on their way
    Stage d;                // staging of a thread's four-channel task (patch rows 0 - 7) ...
    Stage1 d1;              // ... and of its one-channel task (patch rows 8, 9)
    BF B[3];
    AF A[2];
    {
        task_load(d, tid, 0);
        load_b(B[0], 0, 0, 0);
        load_b(B[1], 0, 0, 1);
        task_store(d, tid, smem);
        row_load(d1, tid, 0);
        row_store(d1, tid, smem);
    }
    __syncthreads();
    WSTAMP(1);
    read_a(A[0], smem, 0, 0);
    // (as in conv_h2.hip: the later-dispatched half of the workgroup loses issue arbitration to the older half at the start of
    // every stage; one s_setprio for that half, wave-uniform condition)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    // ---- K loop: per chunk 3 stages (ky) of 2 k-steps; k-step q of the chunk multiplies B[q % 3] and loads the weights of
    // k-step q + 2 into B[(q + 2) % 3]; the next chunk's patch is loaded in two goes and cut into the other buffer
    for (int c = 0; c < nch; ++c) {
        const unsigned char* cur = smem + (c & 1) * W_A_BYTES;
        unsigned char* nxt = smem + ((c + 1) & 1) * W_A_BYTES;
        const int cn = (c + 1 < nch) ? c + 1 : c;          // (last chunk: a dummy re-load, nobody reads the result)
        // (the staging addresses are recomputed per chunk from a thread id the compiler cannot see through: hoisted out of
        // the loop they would sit in eight registers the accumulators and fragments need)
        int to = tid;
        asm volatile("" : "+v"(to));
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            // weights two k-steps ahead
            {
                const int q2 = q + 2;
                if (q2 < 6) load_b(B[q2 % 3], c, q2 >> 1, q2 & 1);
                else load_b(B[q2 % 3], cn, (q2 - 6) >> 1, (q2 - 6) & 1);
            }
            if (q == 0) task_load(d, to, cn);
            if (q == 1) row_load(d1, to, cn);
            // next k-step's A fragments
            if (q + 1 < 6) read_a(A[(q + 1) & 1], cur, (q + 1) >> 1, (q + 1) & 1);
            multiply(A[q & 1], B[q % 3]);
            if (q == 4) task_store(d, to, nxt);
            if (q == 5) row_store(d1, to, nxt);
        }
        __syncthreads();
        if (c + 1 < nch) read_a(A[0], nxt, 0, 0);
    }

    WSTAMP(2);
    // ---- epilogue: the four xi accumulators of every pair meet in LDS, one output-channel half at a time.  A thread then
    // owns a 2x2 pixel window (image rows 2 yp, 2 yp + 1 of the tile, output pair p) x 4 channels: output transform, bias /
    // addend, ReLU / ReLU mask, 16-byte stores, and - where a pooling layer follows - the window's maximum and arg-max code.
    constexpr bool M_FWD = MODE != 2, M_BWD = MODE != 1;      // what this instantiation has to handle
    const float* const e_bias = M_FWD ? b.bias : nullptr;
    const bool e_relu = MODE == 0 ? (b.relu != 0) : (MODE == 1);
    unsigned* const e_bits_out = M_FWD ? im.bits_out : nullptr;
    float* const e_pool_out = M_FWD ? im.pool_out : nullptr;
    unsigned* const e_pcode_out = M_FWD ? im.pcode_out : nullptr;
    const float* const e_addend = M_BWD ? im.addend : nullptr;
    const unsigned* const e_bits_in = M_BWD ? im.bits_in : nullptr;
    float* E = reinterpret_cast<float*>(smem);                               // [xi][pair row 64][128]
    unsigned* WB = reinterpret_cast<unsigned*>(smem + W_BITS_OFF);           // [pixel 128][4 words]
    unsigned* PC = reinterpret_cast<unsigned*>(smem + W_CODE_OFF);           // [pooled pixel 32][4 words][4 positions]
    const int words = Cout >> 5;
    float amax = 0.f;
    const int yp = tid >> 7, p = (tid >> 4) & 7, cq = tid & 15;
    // what the output pass reads from global memory is requested here, ahead of the exchange through LDS: the bias of this
    // thread's channels (both passes) and, in an input-gradient launch, the ReLU-mask words of its 2 x 2 pixels
    f32x4 bvp[2];
    unsigned mw[2][2][2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int co = n0 + pass * 64 + cq * 4;
        bvp[pass] = e_bias ? *reinterpret_cast<const f32x4*>(e_bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gy = y0 + 2 * yp + k, gx = x0 + 2 * p;
            const size_t pa = (size_t)gy * W + gx;
            mw[pass][k][0] = (e_bits_in && gy < H && gx < W) ? e_bits_in[pa * (Cout >> 5) + (co >> 5)] : 0u;
            mw[pass][k][1] = (e_bits_in && gy < H && gx + 1 < W) ? e_bits_in[(pa + 1) * (Cout >> 5) + (co >> 5)] : 0u;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;      // row of the 32-row tile
                E[(xi * 64 + mt * 32 + m) * 128 + wn * 64 + nt * 32 + l31] = fmaf(accx[mt][nt][r], LO_DOWN, accm[mt][nt][r]) * inv;
            }
    __syncthreads();
    WSTAMP(3);
    for (int pass = 0; pass < 2; ++pass) {      // the two 64-channel halves of the tile, one after the other per thread
        const int co = n0 + pass * 64 + cq * 4;
        const f32x4 bv = bvp[pass];
        const int w = pass * 2 + (cq >> 3), sh = (cq & 7) * 4;
        f32x4 win[2][2];                  // [row of the window][column]
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int yy = 2 * yp + k, pr = yy * 8 + p;
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(E + (0 * 64 + pr) * 128 + pass * 64 + cq * 4);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(E + (1 * 64 + pr) * 128 + pass * 64 + cq * 4);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(E + (2 * 64 + pr) * 128 + pass * 64 + cq * 4);
            const f32x4 m3 = *reinterpret_cast<const f32x4*>(E + (3 * 64 + pr) * 128 + pass * 64 + cq * 4);
            f32x4 ya = m0 + m1 + m2 + bv, yb = m1 - m2 - m3 + bv;
            const int gy = y0 + yy, gx = x0 + 2 * p;
            const bool ina = gy < H && gx < W, inb = gy < H && gx + 1 < W;
            const size_t pa = (size_t)gy * W + gx;
            // input-gradient launches: the loss gradient injected at this layer (content), then the ReLU mask of the map
            // this gradient belongs to (one bit per channel, 32 channels per word)
            if (e_addend) {
                if (ina) ya += *reinterpret_cast<const f32x4*>(e_addend + pa * Cout + co);
                if (inb) yb += *reinterpret_cast<const f32x4*>(e_addend + (pa + 1) * Cout + co);
            }
            unsigned ka = 0xFu, kb = 0xFu;
            if (e_bits_in) {
                ka = (mw[pass][k][0] >> (co & 31)) & 0xFu;
                kb = (mw[pass][k][1] >> (co & 31)) & 0xFu;
            }
            unsigned na = 0u, nb = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e_relu) { ya[e] = fmaxf(ya[e], 0.f); yb[e] = fmaxf(yb[e], 0.f); }
                ya[e] = ((ka >> e) & 1u) ? ya[e] : 0.f;
                yb[e] = ((kb >> e) & 1u) ? yb[e] : 0.f;
                na |= (ya[e] > 0.f ? 1u : 0u) << e;
                nb |= (yb[e] > 0.f ? 1u : 0u) << e;
                if (ina) amax = fmaxf(amax, fabsf(ya[e]));
                if (inb) amax = fmaxf(amax, fabsf(yb[e]));
            }
            if (ina) *reinterpret_cast<f32x4*>(im.out + pa * Cout + co) = ya;
            if (inb) *reinterpret_cast<f32x4*>(im.out + (pa + 1) * Cout + co) = yb;
            if (e_bits_out) {
                const int pix = yy * 16 + 2 * p;
                // (the 8 threads cq & 7 = 0..7 hold the 8 nibbles of one 32-channel word)
                const unsigned wa = or8(na << sh), wb = or8(nb << sh);
                if ((cq & 7) == 0) { WB[pix * 4 + w] = wa; WB[(pix + 1) * 4 + w] = wb; }
            }
            win[k][0] = ya; win[k][1] = yb;
        }
        const int py = (y0 + 2 * yp) >> 1, px = (x0 + 2 * p) >> 1;
        const bool inw = py < PH2 && px < PW2;
        if (e_pool_out) {
            // 2x2/2 max pool (+ the arg-max code the un-pooling loader of the backward pass reads: the window's FIRST maximum,
            // where it is positive - max_pool2d's backward and the ReLU mask of the pooled activation in one)
            f32x4 mx;
            unsigned cn[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float e0 = win[0][0][e], e1 = win[0][1][e], e2 = win[1][0][e], e3 = win[1][1][e];
                mx[e] = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
                int pos = 0;
                float best = e0;
                if (e1 > best) { best = e1; pos = 1; }
                if (e2 > best) { best = e2; pos = 2; }
                if (e3 > best) { best = e3; pos = 3; }
                const bool live = best > 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) cn[q] |= ((live && pos == q) ? 1u : 0u) << e;
            }
            if (inw) *reinterpret_cast<f32x4*>(e_pool_out + ((size_t)py * PW2 + px) * Cout + co) = mx;
            if (e_pcode_out) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned wq = or8(cn[q] << sh);
                    if ((cq & 7) == 0) PC[((yp * 8 + p) * 4 + w) * 4 + q] = wq;
                }
            }
        }
    }
    if (e_bits_out || e_pcode_out) {
        __syncthreads();          // the mask and code words are complete
        if (e_bits_out) {
            const int pix = tid >> 2, w2 = tid & 3;
            const int gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
            if (gy < H && gx < W) e_bits_out[((size_t)gy * W + gx) * words + (n0 >> 5) + w2] = WB[tid];
        }
        if (e_pcode_out) {
            const int pp = tid >> 4, w2 = (tid >> 2) & 3, q = tid & 3;
            const int qy = (y0 >> 1) + (pp >> 3), qx = (x0 >> 1) + (pp & 7);
            if (qy < PH2 && qx < PW2) e_pcode_out[(((size_t)qy * PW2 + qx) * words + (n0 >> 5) + w2) * 4 + q] = PC[tid];
        }
    }
    if (im.amax_out) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (lane == 0) atomicMax(im.amax_out + ((blockIdx.x * 8 + wave) & (NST_AMAX_SLOTS - 1)), __float_as_uint(amax));
    }
    WSTAMP(4);
#ifdef NST_WINO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (diagnostic build: the stores have left the wave)
#endif
    WSTAMP(5);
    WSTAMP_REAL(7);
}

namespace {
template <bool UNPOOL, int MODE>
hipError_t wino_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_batch_kernel<UNPOOL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
}
}  // namespace

hipError_t conv_wino_init_device() {
    hipError_t e = wino_attr<false, 0>();
    if (e == hipSuccess) e = wino_attr<false, 1>();
    if (e == hipSuccess) e = wino_attr<false, 2>();
    if (e == hipSuccess) e = wino_attr<true, 0>();
    if (e == hipSuccess) e = wino_attr<true, 2>();
    return e;
}

bool conv_wino_eligible(const ConvBatch& b) {
    if (!b.wt_wino || b.Cin < 128 || b.Cin % 64 != 0 || b.Cout % 128 != 0 || b.Cin2 != 0) return false;
    for (int i = 0; i < b.n; ++i) {
        const ConvImage& im = b.img[i];
        if (im.in2 || im.mask || !im.amax_in || (b.unpool != 0) != (im.pcode_in != nullptr)) return false;
        if (im.pcode_out && !im.pool_out) return false;
        if ((size_t)im.H * im.W * (b.Cin > b.Cout ? b.Cin : b.Cout) * 4 >= 0xFFFFFF00ull) return false;
    }
    return true;
}

hipError_t launch_conv_wino_batch(const ConvBatch& b0, hipStream_t stream) {
    if (b0.n < 1 || b0.n > 8 || !conv_wino_eligible(b0)) return hipErrorInvalidValue;
    ConvBatch b = b0;
    int tiles = 0;
    for (int i = 0; i < b.n; ++i) {
        b.img[i].tiles_x = (b.img[i].W + W_TW - 1) / W_TW;
        tiles += b.img[i].tiles_x * ((b.img[i].H + W_TH - 1) / W_TH);
        b.img[i].tile_end = tiles;
    }
    // the epilogue form this launch fits (see the kernel's MODE)
    bool fwd = b.relu && b.bias, bwd = !b.relu && !b.bias;
    for (int i = 0; i < b.n; ++i) {
        const ConvImage& im = b.img[i];
        fwd = fwd && !im.addend && !im.bits_in;
        bwd = bwd && !im.bits_out && !im.pool_out && !im.pcode_out;
    }
    const dim3 grid(tiles * (b.Cout / 128)), block(512);
    if (b.unpool) {
        if (bwd) hipLaunchKernelGGL((conv_wino_batch_kernel<true, 2>), grid, block, W_LDS, stream, b);
        else hipLaunchKernelGGL((conv_wino_batch_kernel<true, 0>), grid, block, W_LDS, stream, b);
    } else {
        if (fwd) hipLaunchKernelGGL((conv_wino_batch_kernel<false, 1>), grid, block, W_LDS, stream, b);
        else if (bwd) hipLaunchKernelGGL((conv_wino_batch_kernel<false, 2>), grid, block, W_LDS, stream, b);
        else hipLaunchKernelGGL((conv_wino_batch_kernel<false, 0>), grid, block, W_LDS, stream, b);
    }
    return hipGetLastError();
}

}  // namespace nst
