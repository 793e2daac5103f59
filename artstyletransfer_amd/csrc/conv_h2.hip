// conv_h2.hip - the 3x3 convolutions (forward and input gradient) on the gfx950 fp16 matrix pipe at
// fp32-level accuracy with THREE MFMAs per product block (conv_bf3.hip needs six).
//
// Every fp32 operand tensor carries one power-of-two scale s (largest magnitude -> [2^14, 2^15), taken from
// the absmax its producer recorded) and is cut into two fp16 pieces
//       hi = fp16(a s)                      (round to nearest, 11 significant bits)
//       lo = fp16((a s - hi) * 2^11)        (the residual a s - hi is exact in fp32; 11 more bits)
// so that a s = hi + lo 2^-11 to 2^-23 relative - one bit short of fp32's own 2^-24 - for every element within
// 2^-29 of the tensor maximum (smaller ones keep an absolute error below 2^-39 of the maximum).  The product is
//       a b sA sB = hi_a hi_b + 2^-11 (hi_a lo_b + lo_a hi_b)            dropped: lo_a lo_b 2^-22
// with the main term and the 2^-11-weighted cross terms accumulated in SEPARATE fp32 accumulators
// (v_mfma_f32_32x32x16_f16; every fp16 x fp16 product is exact in fp32).  The cross accumulator's rounding
// enters the result scaled by 2^-11, so the accumulation error is that of ONE fp32 chain; measured against an
// fp64 evaluation the feature maps are as accurate as an fp32 MFMA's (tests/test_hip_parity.py,
// tools/diag_accuracy.py).  3 x 32 = 96 matrix-pipe cycles per 32x32x16 block against 512 for
// v_mfma_f32_32x32x2_f32: 5.3x the fp32-MFMA ceiling.
//
// Structure as conv_bf3.hip: implicit GEMM, halo patch of a 32-channel chunk staged once per 9 taps (cut while
// being staged), per-tap weight slice double buffered, buffer loads with loop-invariant offsets, pinned
// read/MFMA interleave.  LDS rows hold the two pieces of 32 channels (2 x 64 B) + 16 B pad = 144 B = 9 x 16 B:
// 16 consecutive rows start on 16 distinct 16-B slots -> conflict-free ds_read_b128 fragments.
// The epilogue records the absmax of what it stores (one atomicMax per wave into 64 slots; max is order
// independent, so the result stays bitwise reproducible) for the launch that consumes that tensor.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

#ifndef NST_H2_DEEP_WEIGHT_PREFETCH
#define NST_H2_DEEP_WEIGHT_PREFETCH 1
#endif
#ifndef NST_H2_WEIGHT_SETS4
#define NST_H2_WEIGHT_SETS4 1
#endif
// Timing-only builds (tools/ablate_conv.sh; never shipped): a set bit gives one stream of the kernel a buffer descriptor
// with zero records, so the range check drops its loads / stores while the instruction stream, the waits and the
// barriers stay - the launch then shows what that stream's memory traffic costs.  bit 0: the patch loads of the main
// source, 1: its weight loads, 2: the stores of the fast epilogues, 3: their loads (mask words, addend), 4: the loads of
// the second (Gram) source.
#ifndef NST_H2_ABLATE
#define NST_H2_ABLATE 0
#endif
constexpr unsigned ABL_IN = (NST_H2_ABLATE & 1) ? 0u : 1u, ABL_W = (NST_H2_ABLATE & 2) ? 0u : 1u, ABL_ST = (NST_H2_ABLATE & 4) ? 0u : 1u,
                   ABL_EL = (NST_H2_ABLATE & 8) ? 0u : 1u, ABL_G = (NST_H2_ABLATE & 16) ? 0u : 1u;
constexpr float LO_UP = 2048.f;   // 2^11
constexpr float LO_DOWN = 1.f / 2048.f;

// KC  = channels per K chunk (32: two k-steps per stage; 16: one).
// NTW = 32-wide output-channel tiles per wave: 2 -> 64 x 64 wave tiles (128 accumulator registers, 8 fragment
//       reads per 12 MFMAs), 1 -> 64 x 32 (64 registers, 6 reads per 6 MFMAs).
// Shapes in use:
//   <16, 128, 2, 32>  512 threads, 16x16 pixels x 128 channels               the 128-channel-multiple layers
//   <16, 128, 4, 32>  256 threads, the same tile with ONE wave per SIMD and 64 x 128 wave tiles: 512 registers per
//                     lane (the 256 accumulators live in AGPRs), 24 fragment reads per 48 MFMAs instead of 32
//                     (nst_options.h2_wg256; tools/micro/tile_shapes.hip: +5 % on the bare MFMA + LDS stream)
//   < 8, 128, 1, 32>  512 threads,  8x16 pixels x 128 channels               their under-filled launches
//   < 8, 128, 2, 16>  256 threads,  8x16 pixels x 128 channels, 61 KB LDS    128-channel layers with Cin <= 128
//   <16,  64, 2, 16>  256 threads, 16x16 pixels x  64 channels, 70 KB LDS    the 64-channel layers: TWO workgroups
//                     per CU, so that one's prologue / epilogue (a large share with K = 576) runs under the other's
//                     MFMAs; 16-channel chunks keep the double-buffered patch of each within half the LDS
// LDS: the halo patch double buffered, the per-tap weight slice triple buffered (see the main K loop); a row holds
// the two pieces of KC channels (2 x 2 KC bytes) + 16 B pad = 144 / 80 B, an odd multiple of 16 B: 16 consecutive
// rows start on 16 distinct 16-B slots -> conflict-free ds_read_b128 fragments.
template <int TH, int BN, int NTW, int KC_>
struct H2Cfg {
    static constexpr int KC = KC_;
    static constexpr int KS = KC / 16;               // k-steps (MFMA K = 16) per stage
    static constexpr int QP = KC / 4;                // 16-byte staging units per pixel (fp32) = per weight row (fp16 x 2)
    static constexpr int PIECEB = KC * 2;            // bytes of one piece of a row
    static constexpr int ROWB = 2 * PIECEB + 16;     // LDS row bytes
    static constexpr int WROWB = 2 * PIECEB;         // global weight row bytes per (tap, cout, chunk)
    static constexpr int TW = 16;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int WM = TH / 4;            // waves along pixels (4 rows x 16 cols each)
    static constexpr int WN = BN / (32 * NTW);
    static constexpr int NT = 64 * WM * WN;
    static constexpr int A_UNITS = PH * PW * QP;              // float4 units of the fp32 patch
    static constexpr int A_PER_T = (A_UNITS + NT - 1) / NT;
    static constexpr int B_UNITS = BN * QP;                   // 16-byte units of the weight slice
    static constexpr int B_PER_T = (B_UNITS + NT - 1) / NT;
    // patch-row pitch rounded up to a multiple of 256 B (see conv_bf3.hip)
    static constexpr int PROWB = ((PW * ROWB + 255) / 256) * 256;
    static constexpr int A_BYTES = PH * PROWB;
    static constexpr int B_BYTES = BN * ROWB;
    static constexpr int LDS_BYTES = 2 * A_BYTES + 3 * B_BYTES;
    static_assert(KC == 16 || KC == 32, "one or two k-steps per stage");
    static_assert(NT == 512 || NT == 256, "eight or four waves per workgroup");
    static_assert(B_UNITS % NT == 0, "every lane stages the same number of weight units (no predication)");
    // (the 256-thread shapes with 16-channel chunks fit twice on a CU; the 4-row shape is for launches with fewer
    // workgroups than CUs and may take more than half of it)
    static_assert(LDS_BYTES <= ((NT == 512 || TH == 4 || NTW == 4) ? 160 : 80) * 1024, "LDS budget (two 256-thread workgroups share a CU)");
    // waves per SIMD the register budget is cut for: the 64 x 128 wave tile takes the whole SIMD (256 VGPRs + 256 AGPRs)
    static constexpr int WAVES_PER_EU = (NTW == 4) ? 1 : 2;
};

// Power-of-two scale that brings the recorded absmax (64 slots of non-negative float bit patterns) into
// [2^14, 2^15), and its inverse.  An all-zero tensor gets a large finite scale (0 * s = 0).
__device__ __forceinline__ void tensor_scale(const unsigned* __restrict__ slots, int lane, float& s, float& inv) {
    unsigned m = slots[lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, off);
        m = o > m ? o : m;
    }
    int e = (int)((m >> 23) & 0xFFu);
    e = e < 32 ? 32 : (e > 250 ? 250 : e);
    s = __uint_as_float((unsigned)(268 - e) << 23);      // 2^(14 - (e - 127))
    inv = __uint_as_float((unsigned)(e - 14) << 23);
}

// four scaled fp32 values -> their hi pieces and lo pieces (each 4 x fp16 = 8 bytes)
__device__ __forceinline__ void cut2x4(const f32x4 v, const float s, u32x2& hi, u32x2& lo) {
    const f32x2 x01 = {v[0] * s, v[1] * s}, x23 = {v[2] * s, v[3] * s};
    const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
    const f32x2 b01 = __builtin_convertvector(h01, f32x2), b23 = __builtin_convertvector(h23, f32x2);
    const f32x2 r01 = (x01 - b01) * LO_UP, r23 = (x23 - b23) * LO_UP;
    const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
    hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}

}  // namespace

// the whole workgroup program; (sp, ct) = spatial tile, output-channel tile of this workgroup.
// M16 (KC == 32 shapes): the products run on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16 - the same
// FLOPs per fragment byte and per matrix-pipe cycle, but the chip holds a higher clock under it (this kernel is
// power/clock bound: tools/micro/mfma_power.hip measures +7 % with this kernel's fragment traffic).  Staging and LDS
// layout are shared; the fragment addressing, the MFMA order and the epilogues (accumulator layout: a lane holds 4
// consecutive pixels of one channel instead of 16 pixel-rows of one channel) have a form of their own.
//
// `tiles` deals this workgroup its tiles: tiles.get(k, p, sp, ct) fills the launch parameters, the spatial tile and the
// output-channel tile of the k-th one (false: there is none).  A workgroup of a PERSISTENT launch (nst_options.h2_persist)
// walks several, and the software pipeline of the K loop then runs ACROSS them: the look-ahead of a tile's last chunk
// stages the first chunk of the NEXT tile (its patch, the weight slices of its first stages) instead of a dummy, so the
// next tile starts without a prologue - no load latency, no LDS staging - right behind this tile's epilogue, whose stores
// drain under its first stages.  Per tile of a one-tile-per-workgroup launch that prologue + dispatch + the write burst
// of 256 CUs finishing in lock-step cost ~11 us (fit over the 72- and 144-stage layers), 6 - 11 % of a launch.
// The arithmetic of a tile does not depend on how it was reached: both launch forms give bitwise the same tensors.
template <int TH, int BN, int NTW, int KC, bool UNPOOL, bool M16, class Tiles>
__device__ __forceinline__ void conv_h2_body(const Tiles& tiles, const int slot_seed) {
    static_assert(!M16 || KC == 32, "the 16x16x32 form takes a whole 32-channel chunk per MFMA");
    using C = H2Cfg<TH, BN, NTW, KC>;
    constexpr int ROWB = C::ROWB, WROWB = C::WROWB, QP = C::QP, PIECEB = C::PIECEB, KS = C::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the three weight-slice buffers first: every fragment address of theirs is then ONE per-lane base + an immediate
    // below 64 KiB (the ds_read offset field), instead of a base register per slice
    unsigned char* ldsB = smem;                          // 3 weight-slice buffers
    unsigned char* ldsA = smem + 3 * C::B_BYTES;         // 2 patch buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % C::WM;
    const int wn = wave / C::WM;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    // weight staging registers: they carry the slices of a chained tile's stages 2 (and 3) from one tile to the next
    u32x4 rb[C::B_PER_T];
    // (KC == 16 shapes: a second weight staging set, see the main K loop)
    constexpr bool DEEPB = (KC == 16) && NST_H2_DEEP_WEIGHT_PREFETCH;
    u32x4 rb2[DEEPB ? C::B_PER_T : 1];
    // (one 16-byte unit per lane and slice - the 64-channel shape: FOUR sets, a slice is requested four stages before it
    // goes to LDS; the sets are handed down from stage to stage)
    constexpr bool DEEPB4 = DEEPB && C::B_PER_T == 1 && NST_H2_WEIGHT_SETS4;
    u32x4 rb3[DEEPB4 ? C::B_PER_T : 1], rb4[DEEPB4 ? C::B_PER_T : 1];

    ConvParams p;
    int sp = 0, ct = 0;
    (void)tiles.get(0, p, sp, ct);
    bool primed = false;         // the first chunk of this tile is in LDS already (staged under the previous tile)
  for (int tile_k = 0;; ++tile_k) {
    const int ty_rel = sp / p.tiles_x;
    const int tx = sp - ty_rel * p.tiles_x;
    const int ty = ty_rel + p.ty0;
    const int y0 = ty * TH;
    const int x0 = tx * C::TW;
    const int n0 = ct * BN;
    // The next tile of this workgroup: here only whether the pipeline chains into it and its operand scale; what else
    // the look-ahead needs is fetched in the last chunk, and the epilogue fetches its own copy of this tile's parameters
    // (kernel arguments, re-read through an index the compiler cannot see through) - three sets of tensor pointers alive
    // over the K loop would not fit the scalar registers.
    auto opaque_s = [](int v) { asm volatile("" : "+s"(v)); return v; };
    float nx_sa = 1.f;
    bool chain = false;
    {
        ConvParams pn;
        int spn = 0, ctn = 0;
        if (tiles.get(tile_k + 1, pn, spn, ctn) && p.Cin > 0 && !p.in2 && !pn.in2) {
            chain = true;
            float unused_inv;
            tensor_scale(pn.amax_in, lane, nx_sa, unused_inv);
        }
    }

    f32x4 ra[C::A_PER_T];

    // Staging unit i of this lane: u = tid + i * NT; patch unit = (pixel u >> 3, channel quad u & 7), weight unit =
    // (row u >> 3, 16-byte piece u & 7).  Addresses are recomputed where they are used, from a thread id the compiler
    // cannot see through (`to`): kept in registers over the K loop they would push the fragment / accumulator set
    // past the 256-register budget, and the compiler then sinks the global loads to the end of the stage.
    auto a_lds_of = [&](int i, int to) -> int {
        const int u = to + i * C::NT;
        const int pix = u / QP;
        const int pr = pix / C::PW;
        const int pc = pix - pr * C::PW;
        return (u < C::A_UNITS) ? pr * C::PROWB + pc * ROWB + (u % QP) * 8 : -1;
    };
    // byte offset of patch unit i inside the image tensor; outside the image / unused: beyond the buffer (reads 0).
    // `pooled`: the tensor is the 2x2-pooled map (H/2 x W/2) and pixel (gy, gx) reads its window's element.
    // (gH, gW, gy0, gx0: image size and tile origin - this tile's, or the next one's for the look-ahead of the last chunk)
    auto a_voff_of = [&](int i, int to, int cin, const bool pooled, const int gH, const int gW, const int gy0, const int gx0) -> unsigned {
        const int u = to + i * C::NT;
        const int pix = u / QP;
        const int pr = pix / C::PW;
        const int pc = pix - pr * C::PW;
        const int gy = gy0 - 1 + pr;
        const int gx = gx0 - 1 + pc;
        // (unsigned compares fold the >= 0 tests; bitwise & keeps this a select instead of short-circuit branches)
        bool ok = (u < C::A_UNITS) & ((unsigned)gy < (unsigned)gH) & ((unsigned)gx < (unsigned)gW);
        if (!pooled) return ok ? ((unsigned)(gy * gW + gx) * (unsigned)cin + (unsigned)(u % QP) * 4u) * 4u : 0xFFFFFF00u;
        const int PH2 = gH >> 1, PW2 = gW >> 1;
        ok = ok & ((gy >> 1) < PH2) & ((gx >> 1) < PW2);      // the odd last row / column belongs to no window
        return ok ? ((unsigned)((gy >> 1) * PW2 + (gx >> 1)) * (unsigned)cin + (unsigned)(u % QP) * 4u) * 4u : 0xFFFFFF00u;
    };
    // byte offset of the arg-max code word of patch unit i: [pooled pixel][32-channel group][window position]
    auto a_coff_of = [&](int i, int to, int cin, const int gH, const int gW, const int gy0, const int gx0) -> unsigned {
        const int u = to + i * C::NT;
        const int pix = u / QP;
        const int pr = pix / C::PW;
        const int pc = pix - pr * C::PW;
        const int gy = gy0 - 1 + pr;
        const int gx = gx0 - 1 + pc;
        const int PH2 = gH >> 1, PW2 = gW >> 1;
        const bool ok = (u < C::A_UNITS) & ((unsigned)gy < (unsigned)gH) & ((unsigned)gx < (unsigned)gW) &
                        ((gy >> 1) < PH2) & ((gx >> 1) < PW2);
        return ok ? (((unsigned)((gy >> 1) * PW2 + (gx >> 1)) * (unsigned)(cin >> 5)) * 4u + (unsigned)((gy & 1) * 2 + (gx & 1))) * 4u : 0xFFFFFF00u;
    };
    auto b_ok = [&](int, int) -> bool { return true; };       // B_UNITS is a multiple of the workgroup size
    auto b_lds_of = [&](int i, int to) -> int { const int u = to + i * C::NT; return (u / QP) * ROWB + (u % QP) * 16; };

    // cut staged fp32 patch units [i0, i1) (held in r[0 .. i1-i0)) into two fp16 pieces, 8 bytes per piece
    // `code` (UNPOOL launches of the main source): r holds the POOLED gradient; element k of unit (pixel, q) survives
    // where bit (bit0 + 4 q + k) of its code word says that this pixel was its window's first positive maximum
    auto store_a = [&](unsigned char* dstA, const float s, const f32x4* r, const int i0, const int i1, const int to,
                       const unsigned* code = nullptr, const int bit0 = 0) {
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int off = a_lds_of(i, to);
            if (off >= 0) {
                u32x2 hi, lo;
                f32x4 v = r[i - i0];
                if (code) {
                    const unsigned bits = code[i - i0] >> (bit0 + ((to + i * C::NT) % QP) * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = ((bits >> k) & 1u) ? v[k] : 0.f;
                }
                cut2x4(v, s, hi, lo);
                unsigned char* row = dstA + off;
                *reinterpret_cast<u32x2*>(row) = hi;
                *reinterpret_cast<u32x2*>(row + PIECEB) = lo;
            }
        }
    };
    // pre-cut weights: 16-byte units go to LDS as they are
    auto store_b = [&](unsigned char* dst, const u32x4 (&r)[C::B_PER_T], const int to) {
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i)
            if (b_ok(i, to)) *reinterpret_cast<u32x4*>(dst + b_lds_of(i, to)) = r[i];
    };
    // fp32 weights (the Gram factor S of the second source): unit (row, q) = 4 floats, cut here
    auto store_b_f32 = [&](unsigned char* dst, const float s, const int to) {
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i)
            if (b_ok(i, to)) {
                u32x2 hi, lo;
                cut2x4(__builtin_bit_cast(f32x4, rb[i]), s, hi, lo);
                const int u = to + i * C::NT;
                unsigned char* row = dst + (u / QP) * ROWB + (u % QP) * 8;
                *reinterpret_cast<u32x2*>(row) = hi;
                *reinterpret_cast<u32x2*>(row + PIECEB) = lo;
            }
    };
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };

    f32x16 accm[2][NTW], accx[2][NTW];       // main products, cross products (weight 2^-11)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accx[a][b][r] = 0.f; }

    // fragment addresses: lane (r = lane & 31, h = lane >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r]
    const int prow = wm * 4 + (l31 >> 4);
    const int pcol = l31 & 15;
    const int a_off0 = (prow + 0) * C::PROWB + pcol * ROWB + half * 16;
    const int a_off1 = (prow + 2) * C::PROWB + pcol * ROWB + half * 16;
    const int b_off0 = (wn * 32 * NTW + l31) * ROWB + half * 16;

    // the fragments of one k-step (16 channels): [m tile][piece], [n tile][piece]
    struct Frags { f16x8 a[2][2]; f16x8 b[NTW][2]; };
    auto request = [&](Frags& f, const unsigned char* abase, const unsigned char* bbase, const int ks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f.a[0][s] = *reinterpret_cast<const f16x8*>(abase + a_off0 + s * PIECEB + ks * 32);
            f.a[1][s] = *reinterpret_cast<const f16x8*>(abase + a_off1 + s * PIECEB + ks * 32);
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                f.b[nt][s] = *reinterpret_cast<const f16x8*>(bbase + b_off0 + nt * 32 * ROWB + s * PIECEB + ks * 32);
        }
    };
    auto multiply = [&](const Frags& f) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[mt][1], f.b[nt][0], accx[mt][nt], 0, 0, 0);
                accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[mt][0], f.b[nt][0], accm[mt][nt], 0, 0, 0);
                accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[mt][0], f.b[nt][1], accx[mt][nt], 0, 0, 0);
            }
    };
    constexpr int READS = 4 + 2 * NTW, MFMAS = 6 * NTW;       // per k-step
    // one k-step: its MFMAs, with the fragment reads of the NEXT k-step issued one per MFMA in between
    auto pin_kstep = [&]() {
#pragma unroll
        for (int g = 0; g < (READS < MFMAS ? READS : MFMAS); ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (MFMAS > READS) __builtin_amdgcn_sched_group_barrier(0x008, MFMAS - READS, 0);
    };
    Frags F[2];

    // ---- M16: v_mfma_f32_16x16x32_f16.  Lane (l15 = lane & 15, kg = lane >> 4) holds A[row l15][k = 8 kg + j] and
    // B[k = 8 kg + j][col l15]; D[row = 4 kg + i][col = l15], i = 0..3.  M tile mt (0..3) = the 16 pixels of image row
    // wm*4 + mt of the tile, N tile nt (0..2 NTW - 1) = 16 output channels: a lane's accumulator (mt, nt)[i] is pixel
    // (row mt, column 4 kg + i), channel 16 nt + l15.  Fragments are handled as PAIRS of tiles (2 tiles x 2 pieces = 4
    // ds_read_b128); a stage (K = 32: one tap of one chunk) is four quadrants (A pair, B pair) of 12 MFMAs each,
    // walked as a snake so that consecutive quadrants share one pair, and the snake alternates direction from stage
    // to stage: every pair is loaded one quadrant (12 MFMAs) before its first use into the register set that the
    // previous quadrant released - 64 fragment registers in all, as in the 32x32 form.
    constexpr int NT16 = 2 * NTW, NP = NTW;          // 16-channel tiles / pairs of them per wave
    f32x4 am16[M16 ? 4 : 1][M16 ? NT16 : 1], ax16[M16 ? 4 : 1][M16 ? NT16 : 1];
    if constexpr (M16) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < NT16; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) { am16[a][b][r] = 0.f; ax16[a][b][r] = 0.f; }
    }
    const int kg = lane >> 4, l15 = lane & 15;
    const int a16_off = (wm * 4) * C::PROWB + l15 * ROWB + kg * 16;
    const int b16_off = (wn * 32 * NTW + l15) * ROWB + kg * 16;
    struct Pair { f16x8 v[2][2]; };                  // [tile of the pair][piece]
    auto read_a = [&](Pair& f, const unsigned char* abase, const int pair) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                f.v[j][s] = *reinterpret_cast<const f16x8*>(abase + a16_off + (2 * pair + j) * C::PROWB + s * PIECEB);
    };
    auto read_b = [&](Pair& f, const unsigned char* bbase, const int pair) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                f.v[j][s] = *reinterpret_cast<const f16x8*>(bbase + b16_off + (2 * pair + j) * 16 * ROWB + s * PIECEB);
    };
    // the 12 MFMAs of quadrant (A pair mp, B pair np): cross (lo x hi), main, cross (hi x lo) - the two MFMAs that
    // accumulate into one cross accumulator are eight instructions apart
    auto quad = [&](const Pair& a, const Pair& b, const int mp, const int np) {
        if constexpr (M16) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ax16[2 * mp + i][2 * np + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[i][1], b.v[j][0], ax16[2 * mp + i][2 * np + j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    am16[2 * mp + i][2 * np + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[i][0], b.v[j][0], am16[2 * mp + i][2 * np + j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    ax16[2 * mp + i][2 * np + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[i][0], b.v[j][1], ax16[2 * mp + i][2 * np + j], 0, 0, 0);
        }
    };
    // pin one quadrant: its 12 MFMAs with `reads` fragment reads (4 or 8) spread between them
    auto pin_quad = [&](const int reads) {
        if (reads == 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };
    Pair RA[2], RB[2];

    // ---- second K source: the Gram backward dF = F S (1 tap, fp32 weights cut here).  One stage per 32-channel
    // chunk; patch and weights alternate between two LDS buffers and are staged one stage ahead (global loads two
    // ahead), one barrier per stage.  Loads are buffer loads with loop-invariant per-lane offsets plus a scalar
    // offset per chunk; out-of-image pixels get an offset beyond the buffer and read as zeros.
    auto gram_source = [&](const float* src, const int cin, const float* wts, const float sa, const float sw) {
        const int nch = cin / KC;
        const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(src), 0, (unsigned)((size_t)p.H * p.W * cin * 4) * ABL_G, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(wts), 0, (unsigned)((size_t)p.Cout * cin * 4) * ABL_G, 0x00020000);
        // A 1-tap source reads the centre of the patch only: stage just the TH x 16 output pixels (no halo ring,
        // no division by the patch width) - the Gram stages are bound by this staging, not by their MFMAs.
        constexpr int G_PER_T = TH * C::TW * QP / C::NT;
        static_assert(TH * C::TW * QP % C::NT == 0 && G_PER_T <= C::A_PER_T, "centre units divide evenly");
        unsigned a_voff[G_PER_T], b_voff[C::B_PER_T];
        int g_lds[G_PER_T];
#pragma unroll
        for (int i = 0; i < G_PER_T; ++i) {
            const int u = tid + i * C::NT;
            const int pix = u / QP, q = u % QP;
            const int py = pix >> 4, px = pix & 15;
            const int gy = y0 + py, gx = x0 + px;
            bool ok = (gy < p.H) & (gx < p.W);
            // row window of the second source (a job evaluated on a stripe of a larger image adds the Gram backward on
            // the rows it owns only): rows outside read as zeros
            if (p.in2_rows > 0) ok = ok & ((unsigned)(gy - p.in2_row0) < (unsigned)p.in2_rows);
            a_voff[i] = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)cin + (unsigned)q * 4u) * 4u : 0xFFFFFF00u;
            g_lds[i] = (py + 1) * C::PROWB + (px + 1) * ROWB + q * 8;
        }
#pragma unroll
        for (int i = 0; i < C::B_PER_T; ++i) {
            const int u = tid + i * C::NT;
            b_voff[i] = b_ok(i, tid) ? (unsigned)(((u / QP) * cin + (u % QP) * 4) * 4) : 0xFFFFFF00u;
        }
        auto load = [&](int chunk) {
#pragma unroll
            for (int i = 0; i < G_PER_T; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, a_voff[i], chunk * KC * 4, 0));
#pragma unroll
            for (int i = 0; i < C::B_PER_T; ++i)
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, b_voff[i], (n0 * cin + chunk * KC) * 4, 0);
        };
        auto store_centre = [&](unsigned char* dstA) {
#pragma unroll
            for (int i = 0; i < G_PER_T; ++i) {
                u32x2 hi, lo;
                cut2x4(ra[i], sa, hi, lo);
                unsigned char* row = dstA + g_lds[i];
                *reinterpret_cast<u32x2*>(row) = hi;
                *reinterpret_cast<u32x2*>(row + PIECEB) = lo;
            }
        };
        load(0);
        store_centre(ldsA);
        store_b_f32(ldsB, sw, tid);
        if (nch > 1) load(1);
        __syncthreads();
        for (int c = 0; c < nch; ++c) {
            const int cb = c & 1;
            if (c + 1 < nch) {
                // the other buffers were last read in stage c-1, which every wave has left
                store_centre(ldsA + (cb ^ 1) * C::A_BYTES);
                store_b_f32(ldsB + (cb ^ 1) * C::B_BYTES, sw, tid);
                if (c + 2 < nch) load(c + 2);
            }
            const unsigned char* centre = ldsA + cb * C::A_BYTES + C::PROWB + ROWB;     // 1 tap: the centre of the patch
            const unsigned char* bcur = ldsB + cb * C::B_BYTES;
            if constexpr (M16) {
                read_a(RA[0], centre, 0);
                read_a(RA[1], centre, 1);
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    read_b(RB[0], bcur, np);
                    quad(RA[0], RB[0], 0, np);
                    quad(RA[1], RB[0], 1, np);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) request(F[ks], centre, bcur, ks);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) multiply(F[ks]);
            }
            __syncthreads();
        }
    };

    // ---- main K source: the 3x3 convolution, one stage per (32-channel chunk c, tap t), two k-steps per stage.
    // Software pipeline with ONE barrier per stage and no LDS latency behind it:
    //   * weights: slice (g+2) is written to LDS at the top of stage g (triple buffer) from registers loaded during
    //     stage g-1, so slice (g+1) is already visible while stage g computes;
    //   * patch: chunk c+1 is loaded at tap 2 and cut into the other patch buffer at tap 6 of chunk c;
    //   * fragments: those of k-step (g,1) are requested while (g,0) multiplies, those of (g+1,0) while (g,1)
    //     multiplies - every wave leaves the barrier with its next operands in registers.
    auto main_source = [&](const float* src, const int cin, const void* wts, const float sa) {
        const int nch = cin / KC;
        const size_t in_px = UNPOOL ? (size_t)(p.H >> 1) * (p.W >> 1) : (size_t)p.H * p.W;
        const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(src), 0, (unsigned)(in_px * cin * 4) * ABL_IN, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(wts), 0, (unsigned)((size_t)9 * p.Cout * nch * WROWB) * ABL_W, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrc_code = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned*>(UNPOOL ? p.pcode_in : nullptr), 0, UNPOOL ? (unsigned)(in_px * (cin >> 5) * 16) : 0u, 0x00020000);
        // where a chunk's patch comes from: this tile, or (look-ahead of the last chunk of a chained tile) the next one
        struct Look { __amdgpu_buffer_rsrc_t rs_in, rs_code; int H, W, y0, x0, n0; float sa; };
        const Look here{rsrc_in, rsrc_code, p.H, p.W, y0, x0, n0, sa};
        auto there = [&]() -> Look {
            ConvParams pn;
            int spn = 0, ctn = 0;
            (void)tiles.get(opaque_s(tile_k + 1), pn, spn, ctn);
            const size_t px = UNPOOL ? (size_t)(pn.H >> 1) * (pn.W >> 1) : (size_t)pn.H * pn.W;
            const int tyn = spn / pn.tiles_x;
            return Look{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pn.in), 0, (unsigned)(px * cin * 4), 0x00020000),
                        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(UNPOOL ? pn.pcode_in : nullptr), 0,
                                                          UNPOOL ? (unsigned)(px * (cin >> 5) * 16) : 0u, 0x00020000),
                        pn.H, pn.W, (tyn + pn.ty0) * TH, (spn - tyn * pn.tiles_x) * C::TW, ctn * BN, nx_sa};
        };
        unsigned rc[UNPOOL ? C::A_PER_T : 1];
        // patch units [i0, i1) of a chunk -> r[0 .. i1-i0) (+ their arg-max code words when un-pooling)
        auto load_a = [&](f32x4* r, const Look& k, int chunk, const int i0, const int i1, const int to) {
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                r[i - i0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          k.rs_in, a_voff_of(i, to, cin, UNPOOL, k.H, k.W, k.y0, k.x0), chunk * KC * 4, 0));
                if (UNPOOL)
                    rc[i - i0] = __builtin_amdgcn_raw_buffer_load_b32(k.rs_code, a_coff_of(i, to, cin, k.H, k.W, k.y0, k.x0), ((chunk * KC) >> 5) * 16, 0);
            }
        };
        // first code bit of a chunk inside its 32-channel word
        auto bit0_of = [&](int chunk) { return (chunk * KC) & 31; };
        // pre-cut weights: [tap][Cout][chunk][piece][32] fp16, i.e. 128 contiguous bytes per (tap, cout, chunk)
        auto load_b = [&](u32x4 (&r)[C::B_PER_T], int chunk, int tap, const int to, const int cout0) {
            const int soff = ((tap * p.Cout + cout0) * nch + chunk) * WROWB;
#pragma unroll
            for (int i = 0; i < C::B_PER_T; ++i) {
                const int u = to + i * C::NT;
                const unsigned voff = b_ok(i, to) ? (unsigned)((u / QP) * nch * WROWB + (u % QP) * 16) : 0xFFFFFF00u;
                r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff, soff, 0);
            }
        };
        auto tap_off = [](int t) { return (t / 3) * C::PROWB + (t % 3) * ROWB; };

        // prologue: chunk 0 and the slices of stages 0, 1 into LDS, slice 2 into registers - unless the previous tile of
        // this workgroup left exactly that state behind (`primed`)
        if (!primed) {
            u32x4 r0[C::B_PER_T], r1[C::B_PER_T];
            load_a(ra, here, 0, 0, C::A_PER_T, tid);
            load_b(r0, 0, 0, tid, n0);
            load_b(r1, 0, 1, tid, n0);
            load_b(rb, 0, 2, tid, n0);
            if constexpr (DEEPB) load_b(rb2, 0, 3, tid, n0);
            if constexpr (DEEPB4) { load_b(rb3, 0, 4, tid, n0); load_b(rb4, 0, 5, tid, n0); }
            __syncthreads();          // the previous source is done with the LDS buffers
            store_a(ldsA, sa, ra, 0, C::A_PER_T, tid, UNPOOL ? rc : nullptr, bit0_of(0));
            store_b(ldsB, r0, tid);
            store_b(ldsB + C::B_BYTES, r1, tid);
            __syncthreads();
        }
        if constexpr (M16) {
            read_a(RA[0], ldsA + tap_off(0), 0);
            read_b(RB[0], ldsB, 0);
        } else {
            request(F[0], ldsA + tap_off(0), ldsB, 0);
        }
        // Two waves per SIMD: the later-dispatched half of an 8-wave workgroup (waves 4-7) loses issue arbitration to the
        // older half at the start of every stage.  One s_setprio for that half, once, before the loop (the condition
        // must be provably wave-uniform: s_setprio ignores EXEC): conv launches -2 % (A/B on one box); flipping the
        // priority around every MFMA cluster instead gave nothing.
        if (C::NT == 512 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);

        // one chunk = 9 stages; PAR = which fragment set holds the operands of its first k-step (alternates per
        // stage when a stage is a single k-step, so chunks are then processed in pairs)
        auto chunk = [&](const int c, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            unsigned char* acur = ldsA + (c & 1) * C::A_BYTES;
            unsigned char* anext = ldsA + ((c + 1) & 1) * C::A_BYTES;
            // The stage body is branch-free so that the compiler's s_waitcnt counts stay exact (a vmcnt merged over
            // control flow waits for ALL loads, i.e. for the patch loads from HBM issued one stage earlier).  In the
            // last chunk the look-ahead stages chunk 0 of the workgroup's next tile (`there`), or - there is none, or
            // one of the two has a second source - re-loads valid addresses (chunk cn) into buffers nobody reads any more.
            const bool last = (c + 1 >= nch);
            const int cn = last ? (chain ? 0 : c) : c + 1;
            Look ahead = here;
            if (last && chain) ahead = there();
            const int to = opaque(tid);
            // The next patch is staged through registers in two halves (loaded at taps 0 / 4, cut into LDS at 3 / 7), or -
            // the 16x16x32 form on the 64 x 64 wave tile, which is four registers short otherwise and spills one staging unit
            // per half INSIDE the loop (buffer_load; s_waitcnt vmcnt(0); scratch_store) - in thirds (taps 0 / 3 / 6 -> 2 / 5 / 8)
#ifndef NST_H2_THIRDS_ALL
#define NST_H2_THIRDS_ALL 0
#endif
            constexpr int PARTS = ((NST_H2_THIRDS_ALL || (M16 && NTW == 2)) && C::A_PER_T % 3 == 0) ? 3 : 2;
            constexpr int AP = (C::A_PER_T + PARTS - 1) / PARTS;
            auto part_lo = [](int j) { return j * AP < C::A_PER_T ? j * AP : C::A_PER_T; };
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // top of stage g = 9 c + t: slice g+2 to LDS, slice g+3 on its way
                if constexpr (DEEPB4) {
                    store_b(ldsB + ((t + 2) % 3) * C::B_BYTES, rb, to);
#pragma unroll
                    for (int i = 0; i < C::B_PER_T; ++i) { rb[i] = rb2[i]; rb2[i] = rb3[i]; rb3[i] = rb4[i]; }
                    if (t + 6 < 9) load_b(rb4, c, t + 6, to, n0);
                    else load_b(rb4, cn, t + 6 - 9, to, ahead.n0);
                } else if constexpr (DEEPB) {
                    // 16-channel chunks: a stage is 12 MFMAs (~400 cycles, ~800 with the other workgroup's wave on the SIMD) -
                    // shorter than an L2 round trip, so a slice loaded ONE stage ahead arrives late and every stage
                    // waits for it.  Two register sets alternate: slice g+2 goes to LDS from the set loaded two stages
                    // ago, and that set is re-loaded with slice g+4.
                    if (((PAR + t) & 1) == 0) {
                        store_b(ldsB + ((t + 2) % 3) * C::B_BYTES, rb, to);
                        if (t + 4 < 9) load_b(rb, c, t + 4, to, n0);
                        else load_b(rb, cn, t + 4 - 9, to, ahead.n0);
                    } else {
                        store_b(ldsB + ((t + 2) % 3) * C::B_BYTES, rb2, to);
                        if (t + 4 < 9) load_b(rb2, c, t + 4, to, n0);
                        else load_b(rb2, cn, t + 4 - 9, to, ahead.n0);
                    }
                } else {
                    store_b(ldsB + ((t + 2) % 3) * C::B_BYTES, rb, to);
                    if (t + 3 < 9) load_b(rb, c, t + 3, to, n0);
                    else load_b(rb, cn, t + 3 - 9, to, ahead.n0);
                }
                if constexpr (PARTS == 2) {
                    if (t == 0) load_a(ra, ahead, cn, 0, AP, to);
                    if (t == 4) load_a(ra, ahead, cn, AP, C::A_PER_T, to);
                } else {
                    if (t % 3 == 0) load_a(ra, ahead, cn, part_lo(t / 3), part_lo(t / 3 + 1), to);
                }
                // keep the loads HERE: left free, the scheduler sinks them towards the end of the stage (their
                // registers are then shared with the fragments) and the next stage stalls on them
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PARTS == 2) {
                    if (t == 3) store_a(anext, ahead.sa, ra, 0, AP, to, UNPOOL ? rc : nullptr, bit0_of(cn));
                    if (t == 7) store_a(anext, ahead.sa, ra, AP, C::A_PER_T, to, UNPOOL ? rc : nullptr, bit0_of(cn));
                } else {
                    if (t % 3 == 2) store_a(anext, ahead.sa, ra, part_lo(t / 3), part_lo(t / 3 + 1), to, UNPOOL ? rc : nullptr, bit0_of(cn));
                }
                const unsigned char* bcur = ldsB + (t % 3) * C::B_BYTES;
                const unsigned char* anxt = (t + 1 < 9) ? acur + tap_off(t + 1) : anext + tap_off(0);
                const unsigned char* bnxt = ldsB + ((t + 1) % 3) * C::B_BYTES;
                if constexpr (M16) {
                    const unsigned char* aptr = acur + tap_off(t);
                    const int sp_ = (PAR + t) & 1;           // a constant once the tap loop is unrolled
                    if (NP == 2) {
                        // sp_ = 0: (A0,B0) (A0,B1) (A1,B1) (A1,B0), entering with RA[0] = A0, RB[0] = B0;
                        // sp_ = 1: (A0,B1) (A0,B0) (A1,B0) (A1,B1), entering with RA[0] = A0, RB[1] = B1
                        const int b0 = sp_, b1 = sp_ ^ 1;     // first / second B pair of this stage's snake
                        read_b(RB[b1], bcur, b1);                           // this stage's other B pair
                        quad(RA[0], RB[b0], 0, b0);
                        pin_quad(4);
                        read_a(RA[1], aptr, 1);                             // this stage's A pair 1
                        quad(RA[0], RB[b1], 0, b1);
                        pin_quad(4);
                        read_a(RA[0], anxt, 0);                             // next stage: A pair 0 (RA[0] is free)
                        quad(RA[1], RB[b1], 1, b1);
                        pin_quad(4);
                        read_b(RB[b1], bnxt, b1);                           // next stage starts with B pair b1 (RB[b1] is free)
                        quad(RA[1], RB[b0], 1, b0);
                        pin_quad(4);
                    } else {
                        // one B pair: (A0,B0) (A1,B0); the B set alternates from stage to stage
                        read_a(RA[1], aptr, 1);
                        quad(RA[0], RB[sp_], 0, 0);
                        pin_quad(4);
                        read_a(RA[0], anxt, 0);
                        read_b(RB[sp_ ^ 1], bnxt, 0);
                        quad(RA[1], RB[sp_], 1, 0);
                        pin_quad(8);
                    }
                } else if (KS == 2) {
                    // k-step 0, fetching k-step 1; then k-step 1, fetching the first fragments of the next stage
                    request(F[1], acur + tap_off(t), bcur, 1);
                    multiply(F[0]);
                    pin_kstep();
                    request(F[0], anxt, bnxt, 0);
                    multiply(F[1]);
                    pin_kstep();
                } else {
                    const int cur = (PAR + t) & 1;      // a constant once the tap loop is unrolled
                    request(F[cur ^ 1], anxt, bnxt, 0);
                    multiply(F[cur]);
                    pin_kstep();
                }
                __syncthreads();
            }
        };
        if (KS == 2 && !M16) {
            for (int c = 0; c < nch; ++c) chunk(c, std::integral_constant<int, 0>{});
        } else {
            // 9 stages per chunk: the fragment sets swap roles from one chunk to the next (nch is even)
            for (int c = 0; c < nch; c += 2) {
                chunk(c, std::integral_constant<int, 0>{});
                chunk(c + 1, std::integral_constant<int, 1>{});
            }
        }
    };

    // scales of the operand tensors (recorded absmax -> power of two) and of the frozen weights
    // (Cin == 0: a launch of the second source alone, e.g. the Gram backward at the top of the chain, dF = F S)
    const bool has_main = p.Cin > 0;
    float sa1 = 1.f, ia1 = 1.f;
    if (has_main) tensor_scale(p.amax_in, lane, sa1, ia1);
    const float w1_inv = has_main ? p.wt_h2_inv : 1.f;
    const float inv = ia1 * w1_inv;
    if (p.in2) {
        // The Gram backward rides on this launch with its own scales; it runs first and the accumulators are then
        // re-expressed in the scale of the main source (powers of two: exact).
        float sa2, ia2, sw2, iw2;
        tensor_scale(p.amax_in2, lane, sa2, ia2);
        tensor_scale(p.amax_w2, lane, sw2, iw2);
        gram_source(p.in2, p.Cin2, p.wt2_f32, sa2, sw2);
        const float ratio = (ia2 * sa1) * (iw2 * (1.f / w1_inv));
        if constexpr (M16) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NT16; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { am16[a][b][r] *= ratio; ax16[a][b][r] *= ratio; }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NTW; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { accm[a][b][r] *= ratio; accx[a][b][r] *= ratio; }
        }
    }
    if (has_main) main_source(p.in, p.Cin, p.wt_h2, sa1);

    // epilogue: D[m][n]: n = lane&31, m = (r&3) + 8*(r>>2) + 4*(lane>>5)
    auto epilogue = [&, &p_loop = p]() {
    // (this tile's parameters again, see the top of the tile loop; a launch over one image has them in its arguments)
    ConvParams p;
    {
        int sp_again, ct_again;
        if (Tiles::REFETCH) (void)tiles.get(opaque_s(tile_k), p, sp_again, ct_again);
        else p = p_loop;
    }
    const int words = p.Cout >> 5;          // ReLU bit-mask words per pixel
    float amax = 0.f;
    auto record_amax = [&]() {
        if (p.amax_out) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
            if (lane == 0) atomicMax(p.amax_out + ((slot_seed * (C::NT / 64) + wave) & (NST_AMAX_SLOTS - 1)), __float_as_uint(amax));
        }
    };

    if constexpr (M16) {
        // ======== epilogues of the 16x16x32 form.  Element (mt, nt, i) of a lane: pixel (y0 + wm*4 + mt, x0 + 4 kg + i),
        // channel cg0 + 16 nt + l15.  A ReLU / arg-max bit word (bit = channel & 31) covers the two 16-channel tiles
        // nt = 2 w, 2 w + 1: a ballot over one tile's register holds, for each of the four pixels 4 kg + i, the 16
        // channel bits of that pixel in bits 16 kg .. 16 kg + 15.
        const int cg0 = n0 + wn * 32 * NTW;
        const float lo_clamp = p.relu ? 0.f : -__builtin_inff();
        auto word_of = [&](const unsigned long long b_even, const unsigned long long b_odd) -> unsigned {
            return (unsigned)((b_even >> (16 * kg)) & 0xFFFFull) | ((unsigned)((b_odd >> (16 * kg)) & 0xFFFFull) << 16);
        };
        const bool interior = (y0 + TH <= p.H) && (x0 + C::TW <= p.W);
        const bool fwd_like = !p.bits_in && !p.addend;
        const bool bwd_like = p.bits_in && !p.bits_out && !p.pool_out && !p.pcode_out;
        if (interior && !p.mask && (fwd_like || bwd_like) && (size_t)p.H * p.W * p.Cout * 4 < 0xFFFFFF00ull) {
            const unsigned out_bytes = (unsigned)((size_t)p.H * p.W * p.Cout * 4);
            const unsigned bit_bytes = (unsigned)((size_t)p.H * p.W * words * 4);
            const int colB = p.Cout * 4, rowB = p.W * colB;              // bytes per pixel / per image row
            const int wcolB = words * 4, wrowB = p.W * wcolB;            // the same for the bit-mask words
            const int pix0 = (y0 + wm * 4) * p.W + x0 + 4 * kg;          // this lane's first pixel
            const unsigned vbase = (unsigned)pix0 * (unsigned)colB + (unsigned)(cg0 + l15) * 4u;
            const unsigned wbase = (unsigned)pix0 * (unsigned)wcolB + (unsigned)(cg0 >> 5) * 4u;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes * ABL_ST, 0x00020000);
            if (bwd_like) {
                const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.bits_in), 0, bit_bytes * ABL_EL, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.addend), 0, p.addend ? out_bytes * ABL_EL : 0u, 0x00020000);
                // all mask words of the tile first (the fragment registers are free now), then arithmetic and stores
                unsigned wv[4][4][NTW];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (NTW == 2) {
                            const u32x2 w2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_bits, wbase, mt * wrowB + i * wcolB, 0));
                            wv[mt][i][0] = w2[0];
                            wv[mt][i][NTW - 1] = w2[1];
                        } else {
                            wv[mt][i][0] = __builtin_amdgcn_raw_buffer_load_b32(rs_bits, wbase, mt * wrowB + i * wcolB, 0);
                        }
                    }
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) {
                    const float bv = p.bias ? p.bias[cg0 + nt * 16 + l15] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        float ad[4] = {0.f, 0.f, 0.f, 0.f};
                        if (p.addend) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                ad[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_add, vbase, mt * rowB + i * colB + nt * 64, 0));
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = fmaf(ax16[mt][nt][i], LO_DOWN, am16[mt][nt][i]) * inv + bv;
                            v = fmaxf(v + ad[i], lo_clamp);              // (+ 0 without an addend: -0 sums as in the general form)
                            v = ((wv[mt][i][nt >> 1] >> ((nt & 1) * 16 + l15)) & 1u) ? v : 0.f;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vbase, mt * rowB + i * colB + nt * 64, 0);
                            amax = fmaxf(amax, fabsf(v));
                        }
                    }
                }
            } else {
                const int PW2 = p.W >> 1;
                const unsigned pool_bytes = (unsigned)((size_t)(p.H >> 1) * PW2 * p.Cout * 4);
                const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(p.bits_out, 0, p.bits_out ? bit_bytes * ABL_ST : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_pool = __builtin_amdgcn_make_buffer_rsrc(p.pool_out, 0, p.pool_out ? pool_bytes * ABL_ST : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rs_code = __builtin_amdgcn_make_buffer_rsrc(
                    p.pcode_out, 0, p.pcode_out ? (unsigned)((size_t)(p.H >> 1) * PW2 * words * 16) * ABL_ST : 0u, 0x00020000);
                // one lane per pixel column group writes the bit words; the others carry an offset beyond the buffer (dropped)
                const unsigned wlane = (l15 == 0) ? wbase : 0xFFFFFF00u;
                const int ppix0 = ((y0 + wm * 4) >> 1) * PW2 + ((x0 + 4 * kg) >> 1);
                const unsigned pbase = (unsigned)ppix0 * (unsigned)colB + (unsigned)(cg0 + l15) * 4u;
                const unsigned cbase = (l15 == 0) ? ((unsigned)ppix0 * (unsigned)words + (unsigned)(cg0 >> 5)) * 16u : 0xFFFFFF00u;
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) {
                    const float bv = p.bias ? p.bias[cg0 + nt * 16 + l15] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = fmaf(ax16[mt][nt][i], LO_DOWN, am16[mt][nt][i]) * inv + bv;
                            v = fmaxf(v, lo_clamp);
                            am16[mt][nt][i] = v;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vbase, mt * rowB + i * colB + nt * 64, 0);
                            amax = fmaxf(amax, fabsf(v));
                        }
                }
                if (p.bits_out) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int w = 0; w < NTW; ++w) {
                                const unsigned long long b0 = __ballot(am16[mt][2 * w][i] > 0.f), b1 = __ballot(am16[mt][2 * w + 1][i] > 0.f);
                                __builtin_amdgcn_raw_buffer_store_b32(word_of(b0, b1), rs_bits, wlane, mt * wrowB + i * wcolB + w * 4, 0);
                            }
                }
                if (p.pool_out) {
                    // 2x2/2 max pool: window = image rows (mt, mt + 1), mt even, x columns (i, i + 1), i even - all four in this lane
#pragma unroll
                    for (int mt = 0; mt < 4; mt += 2)
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            const int poff = (mt >> 1) * PW2 + (i >> 1);
                            unsigned long long code[NTW][4][2];
#pragma unroll
                            for (int nt = 0; nt < NT16; ++nt) {
                                const float e0 = am16[mt][nt][i], e1 = am16[mt][nt][i + 1], e2 = am16[mt + 1][nt][i], e3 = am16[mt + 1][nt][i + 1];
                                const float mx = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx), rs_pool, pbase, poff * colB + nt * 64, 0);
                                if (p.pcode_out) {
                                    int pos = 0;
                                    float best = e0;
                                    if (e1 > best) { best = e1; pos = 1; }
                                    if (e2 > best) { best = e2; pos = 2; }
                                    if (e3 > best) { best = e3; pos = 3; }
                                    const bool live = best > 0.f;
#pragma unroll
                                    for (int q = 0; q < 4; ++q) code[nt >> 1][q][nt & 1] = __ballot(live && pos == q);
                                }
                            }
                            if (p.pcode_out) {
#pragma unroll
                                for (int w = 0; w < NTW; ++w)
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        __builtin_amdgcn_raw_buffer_store_b32(word_of(code[w][q][0], code[w][q][1]), rs_code, cbase,
                                                                              (poff * words + w) * 16 + q * 4, 0);
                            }
                        }
                }
            }
            record_amax();
            return;
        }
        // ---- general epilogue of the 16x16x32 form (edge tiles, fp32 masks, tensors from 4 GiB up)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int y = y0 + wm * 4 + mt;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + 4 * kg + i;
                const bool inb = (y < p.H && x < p.W);
                const size_t pix = (size_t)y * p.W + x;
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt) {
                    const int co = cg0 + nt * 16 + l15;
                    const size_t idx = pix * p.Cout + co;
                    float v = fmaf(ax16[mt][nt][i], LO_DOWN, am16[mt][nt][i]) * inv + (p.bias ? p.bias[co] : 0.f);
                    if (p.addend && inb) v += p.addend[idx];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.bits_in) {
                        const unsigned wv = inb ? p.bits_in[pix * words + (co >> 5)] : 0u;
                        v = ((wv >> (co & 31)) & 1u) ? v : 0.f;
                    } else if (p.mask) {
                        v = (inb && p.mask[idx] > 0.f) ? v : 0.f;
                    }
                    am16[mt][nt][i] = v;
                    if (inb) {
                        p.out[idx] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
                }
                if (p.bits_out) {
#pragma unroll
                    for (int w = 0; w < NTW; ++w) {
                        const unsigned long long b0 = __ballot(am16[mt][2 * w][i] > 0.f), b1 = __ballot(am16[mt][2 * w + 1][i] > 0.f);
                        if (l15 == 0 && inb) p.bits_out[pix * words + (cg0 >> 5) + w] = word_of(b0, b1);
                    }
                }
            }
        }
        if (p.pool_out) {
            const int PH2 = p.H >> 1, PW2 = p.W >> 1;
#pragma unroll
            for (int mt = 0; mt < 4; mt += 2)
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int py = (y0 + wm * 4 + mt) >> 1, px = (x0 + 4 * kg + i) >> 1;
                    const bool inw = (py < PH2 && px < PW2);
                    unsigned long long code[NTW][4][2];
#pragma unroll
                    for (int nt = 0; nt < NT16; ++nt) {
                        const float e0 = am16[mt][nt][i], e1 = am16[mt][nt][i + 1], e2 = am16[mt + 1][nt][i], e3 = am16[mt + 1][nt][i + 1];
                        const float mx = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
                        if (inw) p.pool_out[((size_t)py * PW2 + px) * p.Cout + cg0 + nt * 16 + l15] = mx;
                        int pos = 0;
                        float best = e0;
                        if (e1 > best) { best = e1; pos = 1; }
                        if (e2 > best) { best = e2; pos = 2; }
                        if (e3 > best) { best = e3; pos = 3; }
                        const bool live = best > 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) code[nt >> 1][q][nt & 1] = __ballot(live && pos == q);
                    }
                    if (p.pcode_out) {
#pragma unroll
                        for (int w = 0; w < NTW; ++w)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (l15 == 0 && inw)
                                    p.pcode_out[(((size_t)py * PW2 + px) * words + (cg0 >> 5) + w) * 4 + q] = word_of(code[w][q][0], code[w][q][1]);
                    }
                }
        }
        record_amax();
        return;
    }

    // ---- fast epilogue: a tile that lies inside the image (every tile when H, W are multiples of 16).  No bounds
    // tests; every access is a buffer instruction = per-lane byte offset fixed for the whole epilogue + a scalar
    // offset per element (SALU arithmetic), so the ~100 instructions per element of the general form below (64-bit
    // indices, bounds, per-element option branches) shrink to ~10.  With K = 576 the general epilogue was a third
    // of a workgroup's time.
    const bool interior = (y0 + TH <= p.H) && (x0 + C::TW <= p.W);
    const bool fwd_like = !p.bits_in && !p.addend;
    const bool bwd_like = p.bits_in && !p.bits_out && !p.pool_out && !p.pcode_out;
    if (interior && !p.mask && (fwd_like || bwd_like) && (size_t)p.H * p.W * p.Cout * 4 < 0xFFFFFF00ull) {
        const unsigned out_bytes = (unsigned)((size_t)p.H * p.W * p.Cout * 4);
        const unsigned bit_bytes = (unsigned)((size_t)p.H * p.W * words * 4);
        const int colB = p.Cout * 4, rowB = p.W * colB;              // bytes per pixel / per image row
        const int wcolB = words * 4, wrowB = p.W * wcolB;            // the same for the bit-mask words
        const int pix0 = (y0 + wm * 4) * p.W + x0 + 4 * half;        // this lane's first pixel
        const int cg0 = n0 + wn * 32 * NTW;                          // this wave's first output channel
        const unsigned vbase = (unsigned)pix0 * (unsigned)colB + (unsigned)(cg0 + l31) * 4u;      // (mod 2^32: tensors up to 4 GiB)
        const unsigned wbase = (unsigned)pix0 * (unsigned)wcolB + (unsigned)(cg0 >> 5) * 4u;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, out_bytes * ABL_ST, 0x00020000);
        // (mt, r) -> scalar offsets of the element's pixel: row (r >> 3) + 2 mt, column (r & 3) + 8 ((r >> 2) & 1)
        auto soff = [&](int mt, int r, int rb, int cb) { return ((r >> 3) + 2 * mt) * rb + ((r & 3) + 8 * ((r >> 2) & 1)) * cb; };
        if (bwd_like) {
            const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.bits_in), 0, bit_bytes * ABL_EL, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.addend), 0, p.addend ? out_bytes * ABL_EL : 0u, 0x00020000);
            const float lo_clamp = p.relu ? 0.f : -__builtin_inff();
            if (!p.addend) {
                // No addend (every input-gradient launch but the one below the content layer): ALL mask words of the
                // tile first - the fragment registers are free now - and then the arithmetic and the stores, so the
                // workgroup waits for one load latency instead of one per group of eight elements (stores to `out`
                // keep the compiler from moving later loads above them, and all eight waves of the CU's only
                // workgroup sit in this epilogue together, so nothing else hides those waits).
                unsigned wv[2][16][NTW];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (NTW == 4) {
                            const u32x4 w4 = __builtin_amdgcn_raw_buffer_load_b128(rs_bits, wbase, soff(mt, r, wrowB, wcolB), 0);
#pragma unroll
                            for (int k = 0; k < NTW; ++k) wv[mt][r][k] = w4[k & 3];
                        } else if (NTW == 2) {
                            const u32x2 w2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_bits, wbase, soff(mt, r, wrowB, wcolB), 0));
                            wv[mt][r][0] = w2[0];
                            wv[mt][r][NTW - 1] = w2[1];
                        } else {
                            wv[mt][r][0] = __builtin_amdgcn_raw_buffer_load_b32(rs_bits, wbase, soff(mt, r, wrowB, wcolB), 0);
                        }
                    }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const float bv = p.bias ? p.bias[cg0 + nt * 32 + l31] : 0.f;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = fmaf(accx[mt][nt][r], LO_DOWN, accm[mt][nt][r]) * inv + bv;
                            v = fmaxf(v + 0.f, lo_clamp);            // (+ 0: the absent addend, so that -0 sums as in the general form)
                            v = ((wv[mt][r][nt] >> l31) & 1u) ? v : 0.f;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vbase, soff(mt, r, rowB, colB) + nt * 128, 0);
                            amax = fmaxf(amax, fabsf(v));
                        }
                }
                record_amax();
                return;
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const float bv = p.bias ? p.bias[cg0 + nt * 32 + l31] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r0 = 0; r0 < 16; r0 += 8) {
                        unsigned wv[8];
                        float ad[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {          // the loads of eight elements first, then their arithmetic
                            const int r = r0 + k;
                            wv[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_bits, wbase, soff(mt, r, wrowB, wcolB) + nt * 4, 0);
                            ad[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_add, vbase, soff(mt, r, rowB, colB) + nt * 128, 0));
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int r = r0 + k;
                            float v = fmaf(accx[mt][nt][r], LO_DOWN, accm[mt][nt][r]) * inv + bv;
                            v = fmaxf(v + ad[k], lo_clamp);          // an absent addend reads as zeros
                            v = ((wv[k] >> l31) & 1u) ? v : 0.f;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vbase, soff(mt, r, rowB, colB) + nt * 128, 0);
                            amax = fmaxf(amax, fabsf(v));
                        }
                    }
                }
            }
        } else {
            const int PW2 = p.W >> 1;
            const unsigned pool_bytes = (unsigned)((size_t)(p.H >> 1) * PW2 * p.Cout * 4);
            const __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(p.bits_out, 0, p.bits_out ? bit_bytes * ABL_ST : 0u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_pool = __builtin_amdgcn_make_buffer_rsrc(p.pool_out, 0, p.pool_out ? pool_bytes * ABL_ST : 0u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_code = __builtin_amdgcn_make_buffer_rsrc(
                p.pcode_out, 0, p.pcode_out ? (unsigned)((size_t)(p.H >> 1) * PW2 * words * 16) * ABL_ST : 0u, 0x00020000);
            // one lane per half-wave writes the bit words; the others carry an offset beyond the buffer (dropped)
            const unsigned wlane = (l31 == 0) ? wbase : 0xFFFFFF00u;
            const int ppix0 = ((y0 + wm * 4) >> 1) * PW2 + ((x0 + 4 * half) >> 1);
            const unsigned pbase = (unsigned)ppix0 * (unsigned)colB + (unsigned)(cg0 + l31) * 4u;
            const unsigned cbase = (l31 == 0) ? ((unsigned)ppix0 * (unsigned)words + (unsigned)(cg0 >> 5)) * 16u : 0xFFFFFF00u;
            const float lo_clamp = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const float bv = p.bias ? p.bias[cg0 + nt * 32 + l31] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = fmaf(accx[mt][nt][r], LO_DOWN, accm[mt][nt][r]) * inv + bv;
                        v = fmaxf(v, lo_clamp);
                        accm[mt][nt][r] = v;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, vbase, soff(mt, r, rowB, colB) + nt * 128, 0);
                        amax = fmaxf(amax, fabsf(v));
                        if (p.bits_out) {
                            // ReLU mask of this output for the backward pass: bit = lane, one word per half-wave
                            const unsigned long long bal = __ballot(v > 0.f);
                            __builtin_amdgcn_raw_buffer_store_b32(half ? (unsigned)(bal >> 32) : (unsigned)bal, rs_bits, wlane,
                                                                  soff(mt, r, wrowB, wcolB) + nt * 4, 0);
                        }
                    }
                    if (p.pool_out) {
                        // 2x2/2 max pool of the tile rows (2 mt, 2 mt + 1): the four window elements sit in this lane's
                        // registers r, r+1, r+8, r+9; window column of r = 2 j: {0, 1, 4, 5}[j] (+ 2 per half-wave)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = 2 * j;
                            const int pcol = (j & 1) + 4 * (j >> 1);
                            const float mx = fmaxf(fmaxf(accm[mt][nt][r], accm[mt][nt][r + 1]),
                                                   fmaxf(accm[mt][nt][r + 8], accm[mt][nt][r + 9]));
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx), rs_pool, pbase,
                                                                  (mt * PW2 + pcol) * colB + nt * 128, 0);
                            if (p.pcode_out) {
                                // arg-max code for the un-pooling loader (see the general form below)
                                int pos = 0;
                                float best = accm[mt][nt][r];
                                if (accm[mt][nt][r + 1] > best) { best = accm[mt][nt][r + 1]; pos = 1; }
                                if (accm[mt][nt][r + 8] > best) { best = accm[mt][nt][r + 8]; pos = 2; }
                                if (accm[mt][nt][r + 9] > best) { best = accm[mt][nt][r + 9]; pos = 3; }
                                const bool live = best > 0.f;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const unsigned long long bal = __ballot(live && pos == q);
                                    __builtin_amdgcn_raw_buffer_store_b32(half ? (unsigned)(bal >> 32) : (unsigned)bal, rs_code, cbase,
                                                                          ((mt * PW2 + pcol) * words + nt) * 16 + q * 4, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
        record_amax();
        return;
    }

    // ---- general epilogue (edge tiles, fp32 masks, tensors from 4 GiB up)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = n0 + wn * 32 * NTW + nt * 32 + l31;
        const int cw = (n0 + wn * 32 * NTW + nt * 32) >> 5;
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
                float v = fmaf(accx[mt][nt][r], LO_DOWN, accm[mt][nt][r]) * inv + bv;
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
                accm[mt][nt][r] = v;
                if (inb) {
                    p.out[idx] = v;
                    amax = fmaxf(amax, fabsf(v));
                }
            }
            if (p.pool_out) {
                // 2x2/2 max pool of the tile rows (2 mt, 2 mt + 1): the four window elements sit in this
                // lane's registers r, r+1, r+8, r+9
                const int py = (y0 + wm * 4 + mt * 2) >> 1;
                const int PH2 = p.H >> 1, PW2 = p.W >> 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 2 * j;
                    const float mx = fmaxf(fmaxf(accm[mt][nt][r], accm[mt][nt][r + 1]),
                                           fmaxf(accm[mt][nt][r + 8], accm[mt][nt][r + 9]));
                    const int mcol = (r & 3) + 8 * (r >> 2) + 4 * half;      // column of register r (row 0 of the pair)
                    const int px = (x0 + mcol) >> 1;
                    const bool inw = (py < PH2 && px < PW2);
                    if (inw) p.pool_out[((size_t)py * PW2 + px) * p.Cout + co] = mx;
                    if (p.pcode_out) {
                        // arg-max code for the un-pooling input-gradient loader: one word per window position, bit =
                        // channel & 31, set where that position holds the window's FIRST maximum and it is positive
                        // (max_pool2d backward + the ReLU mask of the pooled activation)
                        int pos = 0;
                        float best = accm[mt][nt][r];
                        if (accm[mt][nt][r + 1] > best) { best = accm[mt][nt][r + 1]; pos = 1; }
                        if (accm[mt][nt][r + 8] > best) { best = accm[mt][nt][r + 8]; pos = 2; }
                        if (accm[mt][nt][r + 9] > best) { best = accm[mt][nt][r + 9]; pos = 3; }
                        const bool live = best > 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned long long bal = __ballot(live && pos == q);
                            if (l31 == 0 && inw)
                                p.pcode_out[(((size_t)py * PW2 + px) * words + cw) * 4 + q] = half ? (unsigned)(bal >> 32) : (unsigned)bal;
                        }
                    }
                }
            }
        }
    }
    record_amax();
    };
    epilogue();
    // on to the workgroup's next tile (a chained one finds its first chunk staged)
    if (!tiles.get(tile_k + 1, p, sp, ct)) break;
    primed = chain;
  }
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2.  The
// logical tile index is therefore re-dealt so that each XCD works through ONE contiguous range of tiles: the
// output-channel tiles of a spatial tile (same input patch) and its neighbours (shared halo) then meet in one L2
// instead of fetching the patch from HBM once per XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_contiguous(const int b, const int grid) {
#ifdef NST_NO_XCD_MAP
    return b;
#else
    const int x = b & 7, k = b >> 3;
    const int q = grid >> 3, r = grid & 7;
    return x * q + (x < r ? x : r) + k;
#endif
}

// one image, one tile per workgroup
struct H2OneTile {
    static constexpr bool REFETCH = false;
    const ConvParams& p;
    int n_ct;
    __device__ __forceinline__ bool get(const int k, ConvParams& q, int& sp, int& ct) const {
        if (k > 0) return false;
        const int t = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
        q = p;
        sp = t / n_ct;
        ct = t % n_ct;
        return true;
    }
};
// several images; one tile per workgroup, or (b.persist: the grid is a multiple of 8 that fills the chip once) the
// workgroup's share of its XCD's contiguous range: the order in which a one-tile-per-workgroup grid would have met them
template <bool PERSIST>
struct H2BatchTiles {
    static constexpr bool REFETCH = PERSIST;
    const ConvBatch& b;
    int n_ct;
    __device__ __forceinline__ bool get(const int k, ConvParams& p, int& sp, int& ct) const {
        int t;
        if (!PERSIST) {
            if (k > 0) return false;
            t = xcd_contiguous((int)blockIdx.x, (int)gridDim.x);
        } else {
            const int x = (int)blockIdx.x & 7, j = ((int)blockIdx.x >> 3) + k * ((int)gridDim.x >> 3);
            const int q = b.total_tiles >> 3, r = b.total_tiles & 7;
            if (j >= q + (x < r ? 1 : 0)) return false;
            t = x * q + (x < r ? x : r) + j;
        }
        const int sp_all = t / n_ct;
        ct = t - sp_all * n_ct;
        int i = 0;
        while (i + 1 < b.n && sp_all >= b.img[i].tile_end) ++i;
        const ConvImage& im = b.img[i];
        p.in = im.in; p.wt = nullptr; p.wt_bf = nullptr; p.bias = b.bias; p.addend = im.addend; p.mask = im.mask; p.out = im.out;
        p.H = im.H; p.W = im.W; p.Cin = b.Cin; p.Cout = b.Cout; p.relu = b.relu;
        p.tiles_x = im.tiles_x; p.tiles_y = 0; p.partial = nullptr; p.partial_floats = 0; p.ksplit = 1;
        p.in2 = im.in2; p.Cin2 = b.Cin2; p.wt2_bf = nullptr; p.bits_out = im.bits_out; p.bits_in = im.bits_in;
        p.pool_out = im.pool_out;
        p.wt_h2 = b.wt_h2; p.wt_h2_inv = b.wt_h2_inv; p.wt2_f32 = im.wt2_f32;
        p.amax_in = im.amax_in; p.amax_in2 = im.amax_in2; p.amax_w2 = im.amax_w2; p.amax_out = im.amax_out;
        p.pcode_in = im.pcode_in; p.pcode_out = im.pcode_out;
        p.in2_row0 = im.in2_row0; p.in2_rows = im.in2_rows; p.ty0 = 0;
        sp = sp_all - (i ? b.img[i - 1].tile_end : 0);
        return true;
    }
};

template <int TH, int BN, int NTW, int KC, bool UNPOOL, bool M16>
__global__ __launch_bounds__((H2Cfg<TH, BN, NTW, KC>::NT), (H2Cfg<TH, BN, NTW, KC>::WAVES_PER_EU)) void conv_h2_kernel(ConvParams p) {
    conv_h2_body<TH, BN, NTW, KC, UNPOOL, M16>(H2OneTile{p, p.Cout / BN}, blockIdx.x);
}

// One launch = one layer over several images (the pyramid levels of a closure), see conv_bf3.hip.
template <int TH, int BN, int NTW, int KC, bool UNPOOL, bool M16, bool PERSIST>
__global__ __launch_bounds__((H2Cfg<TH, BN, NTW, KC>::NT), (H2Cfg<TH, BN, NTW, KC>::WAVES_PER_EU)) void conv_h2_batch_kernel(ConvBatch b) {
    conv_h2_body<TH, BN, NTW, KC, UNPOOL, M16>(H2BatchTiles<PERSIST>{b, b.Cout / BN}, blockIdx.x);
}
// the shapes the persistent form is built for: the ones whose launches have more tiles than the chip holds workgroups
template <int TH, int BN, int NTW, int KC>
constexpr bool h2_persist_shape = (KC == 16) || (TH == 16 && NTW == 2 && KC == 32);

template <int TH, int BN, int NTW, int KC, bool UNPOOL, bool M16>
static hipError_t init_form() {
    constexpr int lds = H2Cfg<TH, BN, NTW, KC>::LDS_BYTES;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h2_kernel<TH, BN, NTW, KC, UNPOOL, M16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h2_batch_kernel<TH, BN, NTW, KC, UNPOOL, M16, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if constexpr (h2_persist_shape<TH, BN, NTW, KC> && !M16) {
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_h2_batch_kernel<TH, BN, NTW, KC, UNPOOL, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    return e;
}
template <int TH, int BN, int NTW, int KC, bool UNPOOL>
static hipError_t init_one() {
    hipError_t e = init_form<TH, BN, NTW, KC, UNPOOL, false>();
    if constexpr (KC == 32 && NTW <= 2) {
        if (e == hipSuccess) e = init_form<TH, BN, NTW, KC, UNPOOL, true>();
    }
    return e;
}

hipError_t conv_h2_init_device() {
    hipError_t e = init_one<16, 128, 2, 32, false>();
    if (e == hipSuccess) e = init_one<8, 128, 1, 32, false>();
    if (e == hipSuccess) e = init_one<16, 64, 2, 16, false>();
    if (e == hipSuccess) e = init_one<16, 128, 2, 32, true>();
    if (e == hipSuccess) e = init_one<8, 128, 1, 32, true>();
    if (e == hipSuccess) e = init_one<16, 64, 2, 16, true>();
    if (e == hipSuccess) e = init_one<8, 128, 2, 16, false>();
    if (e == hipSuccess) e = init_one<8, 128, 2, 16, true>();
    if (e == hipSuccess) e = init_one<4, 128, 1, 32, false>();
    if (e == hipSuccess) e = init_one<4, 128, 1, 32, true>();
    if (e == hipSuccess) e = init_one<16, 128, 4, 32, false>();
    if (e == hipSuccess) e = init_one<16, 128, 4, 32, true>();
    return e;
}

// a 128-channel launch whose 16-row tiles cannot fill 256 CUs twice over uses 8-row tiles, and 4-row tiles (256
// threads, two workgroups per CU) when even those leave most of the chip idle: the deep layers of small jobs, whose
// few workgroups each walk the whole K = 4 608 - the launch lasts as long as ONE workgroup does
static int h2_tile_rows(int Cout, long blocks16, int forced = 0) {
    if (Cout % 128 != 0) return 16;
    if (forced == 4 || forced == 8 || forced == 16) return forced;      // nst_options.h2_tile_rows (experiments)
    if (blocks16 < 100) return 4;
    return blocks16 < 400 ? 8 : 16;
}

// The 16x16x32 form exists for the 32-channel-chunk shapes (profiles/r02_mfma_shape_experiments.txt).
// nst_options.h2_mfma16: 0 = never; 1 (default) = the 8-row shape (64 x 32 wave tiles, 180 registers: +6 ... 11 % per launch);
// 2 = that and the 16-row 64 x 64 wave-tile shape where the launch does not un-pool - with the patch staged in thirds its K loop
// has no spill stores left, and over a sustained run (300 closures) the L=2 closure is 9.08 against 9.00 ms: no gain over
// 32x32x16 on that shape; 3 = wherever the form is built (also the 4-row shape: L=0 closure -3 %, and the un-pooling launches,
// whose arg-max code words do not fit beside the form: in-loop spills, -25 %).
static bool h2_use_m16(int mfma16, int th, int ntw, bool unpool) {
    if (mfma16 >= 3) return true;
    if (mfma16 == 2) return th == 8 || (th == 16 && ntw == 2 && !unpool);
    return mfma16 == 1 && th == 8;
}
template <int TH, int BN, int NTW, int KC>
static void launch_batch_cfg(const ConvBatch& b, int blocks, hipStream_t stream) {
    constexpr int lds = H2Cfg<TH, BN, NTW, KC>::LDS_BYTES;
    constexpr int nt = H2Cfg<TH, BN, NTW, KC>::NT;
    if constexpr (KC == 32 && NTW <= 2) {
        if (h2_use_m16(b.mfma16, TH, NTW, b.unpool != 0)) {
            if (b.unpool) hipLaunchKernelGGL((conv_h2_batch_kernel<TH, BN, NTW, KC, true, true, false>), dim3(blocks), dim3(nt), lds, stream, b);
            else hipLaunchKernelGGL((conv_h2_batch_kernel<TH, BN, NTW, KC, false, true, false>), dim3(blocks), dim3(nt), lds, stream, b);
            return;
        }
    }
    if constexpr (h2_persist_shape<TH, BN, NTW, KC>) {
        if (b.persist) {
            if (b.unpool) hipLaunchKernelGGL((conv_h2_batch_kernel<TH, BN, NTW, KC, true, false, true>), dim3(blocks), dim3(nt), lds, stream, b);
            else hipLaunchKernelGGL((conv_h2_batch_kernel<TH, BN, NTW, KC, false, false, true>), dim3(blocks), dim3(nt), lds, stream, b);
            return;
        }
    }
    if (b.unpool) hipLaunchKernelGGL((conv_h2_batch_kernel<TH, BN, NTW, KC, true, false, false>), dim3(blocks), dim3(nt), lds, stream, b);
    else hipLaunchKernelGGL((conv_h2_batch_kernel<TH, BN, NTW, KC, false, false, false>), dim3(blocks), dim3(nt), lds, stream, b);
}
template <int TH, int BN, int NTW, int KC>
static void launch_single_cfg(const ConvParams& p, int blocks, hipStream_t stream) {
    constexpr int lds = H2Cfg<TH, BN, NTW, KC>::LDS_BYTES;
    constexpr int nt = H2Cfg<TH, BN, NTW, KC>::NT;
    if constexpr (KC == 32 && NTW <= 2) {
        if (h2_use_m16(p.mfma16, TH, NTW, p.pcode_in != nullptr)) {
            if (p.pcode_in) hipLaunchKernelGGL((conv_h2_kernel<TH, BN, NTW, KC, true, true>), dim3(blocks), dim3(nt), lds, stream, p);
            else hipLaunchKernelGGL((conv_h2_kernel<TH, BN, NTW, KC, false, true>), dim3(blocks), dim3(nt), lds, stream, p);
            return;
        }
    }
    if (p.pcode_in) hipLaunchKernelGGL((conv_h2_kernel<TH, BN, NTW, KC, true, false>), dim3(blocks), dim3(nt), lds, stream, p);
    else hipLaunchKernelGGL((conv_h2_kernel<TH, BN, NTW, KC, false, false>), dim3(blocks), dim3(nt), lds, stream, p);
}

static bool h2_operands_ok(const void* wt, const unsigned* amax_in, int Cin, int Cout, const float* in2, const float* wt2,
                           const unsigned* a2, const unsigned* w2, int Cin2) {
    if (Cin % 32 != 0 || Cout % 64 != 0) return false;
    if (Cin > 0 ? (!wt || !amax_in) : !in2) return false;       // Cin == 0: the second source alone
    if (in2 && (!wt2 || !a2 || !w2 || Cin2 % 32 != 0 || Cin2 != Cout)) return false;
    return true;
}

// fills tiles_x / tile_end of every image and launches one grid over all of them
hipError_t launch_conv_h2_batch(const ConvBatch& b0, hipStream_t stream) {
    if (b0.n < 1 || b0.n > 8) return hipErrorInvalidValue;
    ConvBatch b = b0;
    const bool wide = (b.Cout % 128 == 0);
    long blocks16 = 0;
    for (int i = 0; i < b.n; ++i) {
        const ConvImage& im = b.img[i];
        if (!h2_operands_ok(b.wt_h2, im.amax_in, b.Cin, b.Cout, im.in2, im.wt2_f32, im.amax_in2, im.amax_w2, b.Cin2))
            return hipErrorInvalidValue;
        if ((size_t)im.H * im.W * (b.Cin > b.Cin2 ? b.Cin : b.Cin2) * 4 >= 0xFFFFFF00ull) return hipErrorInvalidValue;
        if ((b.unpool != 0) != (im.pcode_in != nullptr)) return hipErrorInvalidValue;
        blocks16 += (long)((im.H + 15) / 16) * ((im.W + 15) / 16) * (b.Cout / 128);
    }
    const bool shortk = wide && b.Cin > 0 && b.Cin <= NST_H2_SHORTK_CIN;      // 8-row tiles, two workgroups per CU
    const int th = shortk ? 8 : h2_tile_rows(b.Cout, blocks16, b.tile_rows), bn = wide ? 128 : 64;
    int tiles = 0;
    for (int i = 0; i < b.n; ++i) {
        b.img[i].tiles_x = (b.img[i].W + 15) / 16;
        tiles += b.img[i].tiles_x * ((b.img[i].H + th - 1) / th);
        b.img[i].tile_end = tiles;
    }
    int blocks = tiles * (b.Cout / bn);
    // Persistent form (nst_options.h2_persist): as many workgroups as the chip holds at once (256 CUs x 1 or 2 by the
    // shape's LDS / thread budget), each walking its share of the tiles with the K pipeline chained across them.  Not for
    // launches with a second K source (its stages use the LDS buffers the chained prologue would be staged in).
    bool second = false;
    for (int i = 0; i < b.n; ++i) second = second || (b.img[i].in2 != nullptr);
    const int resident = 256 * ((!wide || shortk || th == 4) ? 2 : 1);
    b.total_tiles = blocks;
    const bool m16 = h2_use_m16(b.mfma16, th, 2, b.unpool != 0) && !shortk && wide;      // (the 16x16x32 form has no persistent build)
    const bool pshape = !wide || shortk || (th == 16 && !b.wg256);
    b.persist = (b.persist && !second && b.Cin > 0 && blocks > resident && pshape && !m16) ? 1 : 0;
    if (b.persist) blocks = resident;
    if (!wide) launch_batch_cfg<16, 64, 2, 16>(b, blocks, stream);
    else if (shortk) launch_batch_cfg<8, 128, 2, 16>(b, blocks, stream);
    else if (th == 4) launch_batch_cfg<4, 128, 1, 32>(b, blocks, stream);
    else if (th == 8) launch_batch_cfg<8, 128, 1, 32>(b, blocks, stream);
    else if (b.wg256) launch_batch_cfg<16, 128, 4, 32>(b, blocks, stream);
    else launch_batch_cfg<16, 128, 2, 32>(b, blocks, stream);
    return hipGetLastError();
}

hipError_t launch_conv_h2(const ConvParams& p0, hipStream_t stream) {
    if (!h2_operands_ok(p0.wt_h2, p0.amax_in, p0.Cin, p0.Cout, p0.in2, p0.wt2_f32, p0.amax_in2, p0.amax_w2, p0.Cin2))
        return hipErrorInvalidValue;
    ConvParams p = p0;
    p.ksplit = 1;
    const bool wide = (p.Cout % 128 == 0);
    const long blocks16 = (long)((p.H + 15) / 16) * ((p.W + 15) / 16) * (p.Cout / 128);
    const bool shortk = wide && p.Cin > 0 && p.Cin <= NST_H2_SHORTK_CIN;
    const int th = shortk ? 8 : h2_tile_rows(p.Cout, blocks16, p.tile_rows), bn = wide ? 128 : 64;
    p.tiles_x = (p.W + 15) / 16;
    // The kernels address with 32-bit buffer offsets.  A launch whose tensors reach 4 GiB runs in bands of output rows
    // (multiples of 16, so tiles and pooling windows stay aligned): every tensor pointer is moved to the band's first
    // row minus a 16-row halo, the band "image" ends 16 rows below the band, and only the band's tile rows are
    // launched - the halo rows are read, never written.
    const int cmax = p.Cin > p.Cout ? (p.Cin > p.Cin2 ? p.Cin : p.Cin2) : (p.Cout > p.Cin2 ? p.Cout : p.Cin2);
    const size_t row_bytes = (size_t)p.W * cmax * 4;
    int band_rows = p.H;
    if ((size_t)p.H * row_bytes >= 0xFFFFFF00ull) {
        const long fit = (long)(0xFFFFFF00ull / row_bytes) - 32;
        band_rows = (int)(fit / 16) * 16;
        if (band_rows < 16) return hipErrorInvalidValue;
    }
    // test hook (nst_options.h2_band_rows): forces bands on small images, so the parity tests walk the band arithmetic
    const int forced_band = p.band_rows;
    if (forced_band >= 16 && forced_band / 16 * 16 < band_rows) band_rows = forced_band / 16 * 16;
    const int PW2 = p.W >> 1, wi = p.Cin >> 5, wo = p.Cout >> 5;
    for (int r0 = 0; r0 < p.H; r0 += band_rows) {
        const int r1 = (r0 + band_rows < p.H) ? r0 + band_rows : p.H;
        const int b0 = (r0 >= 16) ? r0 - 16 : 0, b1 = (r1 + 16 < p.H) ? r1 + 16 : p.H;
        ConvParams q = p;
        const size_t px0 = (size_t)b0 * p.W, pp0 = (size_t)(b0 >> 1) * PW2;       // first pixel / pooled pixel of the band image
        if (q.pcode_in) { q.in += pp0 * p.Cin; q.pcode_in += pp0 * wi * 4; }
        else if (q.in) q.in += px0 * p.Cin;
        if (q.in2) { q.in2 += px0 * p.Cin2; if (q.in2_rows > 0) q.in2_row0 -= b0; }
        q.out += px0 * p.Cout;
        if (q.addend) q.addend += px0 * p.Cout;
        if (q.mask) q.mask += px0 * p.Cout;
        if (q.bits_in) q.bits_in += px0 * wo;
        if (q.bits_out) q.bits_out += px0 * wo;
        if (q.pool_out) q.pool_out += pp0 * p.Cout;
        if (q.pcode_out) q.pcode_out += pp0 * wo * 4;
        q.H = b1 - b0;
        q.ty0 = (r0 - b0) / th;
        q.tiles_y = (r1 - r0 + th - 1) / th;
        const int blocks = q.tiles_x * q.tiles_y * (p.Cout / bn);
        if (!wide) launch_single_cfg<16, 64, 2, 16>(q, blocks, stream);
        else if (shortk) launch_single_cfg<8, 128, 2, 16>(q, blocks, stream);
        else if (th == 4) launch_single_cfg<4, 128, 1, 32>(q, blocks, stream);
        else if (th == 8) launch_single_cfg<8, 128, 1, 32>(q, blocks, stream);
        else if (q.wg256) launch_single_cfg<16, 128, 4, 32>(q, blocks, stream);
        else launch_single_cfg<16, 128, 2, 32>(q, blocks, stream);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- absmax of a tensor into its 64 slots (producers without an epilogue of their own) -----------------------
__global__ __launch_bounds__(256) void absmax_slots_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ slots) {
    float m = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0)
        atomicMax(slots + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (NST_AMAX_SLOTS - 1)), __float_as_uint(m));
}
hipError_t launch_absmax_slots(const float* x, size_t n, unsigned* slots, hipStream_t stream) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_slots_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, n, slots);
    return hipGetLastError();
}

}  // namespace nst
