// nst_kernels.h - internal launch interface between the C-ABI layer (nst_api.cpp) and the
// gfx950 kernels.  Activations inside the network are NHWC fp32 ("pixel-major": [y][x][C]);
// the optimised image, its pyramid levels and their gradients are planar (3,h,w) fp32, the
// storage of the torch tensor the reference optimises.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace nst {

struct ConvParams {
    const float* in;      // [H][W][Cin]
    const float* wt;      // [TAPS][Cout][Cin]
    const void* wt_bf;    // conv_bf3 only: [9][Cout][Cin/32][3 pieces][32] bf16 (see conv_bf3.hip)
    const float* bias;    // [Cout] or nullptr
    const float* addend;  // [H][W][Cout] or nullptr (may alias out)
    const float* mask;    // [H][W][Cout] or nullptr: out = mask > 0 ? v : 0
    float* out;           // [H][W][Cout]
    int H, W, Cin, Cout;
    int relu;
    int tiles_x, tiles_y; // filled by the launcher
    float* partial;       // split-K workspace (nullable): [ksplit][H][W][Cout]
    size_t partial_floats;
    int ksplit;           // filled by the launcher
    // conv_bf3 only ------------------------------------------------------------------------------
    const float* in2;     // optional second K source [H][W][Cin2], accumulated as a 1x1 product with wt2_bf
    int Cin2;
    const void* wt2_bf;   // [1][Cout][Cin2/32][3][32] bf16
    unsigned* bits_out;   // optional: ReLU bit-mask of the output, [H*W][Cout/32] words (bit = channel & 31)
    const unsigned* bits_in;  // optional: replaces `mask` (same layout, of the tensor the gradient flows into)
    float* pool_out;      // optional: 2x2/2 max-pooled output [H/2][W/2][Cout]
    // conv_h2 only (bits_out / bits_in / pool_out / in2 / Cin2 above apply too) -------------------------
    const void* wt_h2;    // [9][Cout][Cin/32][2 pieces][32] fp16, the weights times 1/wt_h2_inv (see conv_h2.hip)
    float wt_h2_inv;      // power of two: true weight = (hi + lo 2^-11) * wt_h2_inv
    const float* wt2_f32; // weights of the second K source, fp32 [Cout][Cin2] (cut in the kernel)
    const unsigned* amax_in;   // NST_AMAX_SLOTS words: absmax of `in` (float bit patterns, max over the slots)
    const unsigned* amax_in2;  // ... of `in2`
    const unsigned* amax_w2;   // ... of `wt2_f32`
    unsigned* amax_out;        // optional: absmax of `out` is recorded here (atomic max; zero it beforehand)
    // un-pooling fused into the operand loader (input-gradient launches that follow a max-pool): `in` is then the
    // gradient w.r.t. the POOLED map, [H/2][W/2][Cin], and pcode_in the arg-max code of that pool
    // ([H/2*W/2][Cin/32][4 window positions] words, bit = channel & 31: set where that position held the window's
    // first maximum and it was positive); pcode_out: a forward launch with pool_out writes that code
    const unsigned* pcode_in;
    unsigned* pcode_out;
    int in2_row0, in2_rows;    // in2_rows > 0: the second source contributes on output rows [in2_row0, in2_row0 + in2_rows) only
    int ty0;                   // first tile row of this launch (filled by the launcher: tensors from 4 GiB up run in row bands)
    int band_rows;             // conv_h2: >= 16 forces row bands of that many rows (nst_options.h2_band_rows; 0 = only when needed)
    int mfma16;                // conv_h2: the 32-channel-chunk shapes run on v_mfma_f32_16x16x32_f16 (nst_options.h2_mfma16)
    int wg256;                 // conv_h2: the 16x16 x 128 tile as 4 waves of 64 x 128 (nst_options.h2_wg256)
    int tile_rows;             // conv_h2: 4 / 8 / 16 forces that tile height on the 128-channel shapes (nst_options.h2_tile_rows)
};

constexpr int NST_AMAX_SLOTS = 64;
// conv_h2: launches with at most this many input channels (K <= 1152) use the 16-channel-chunk shapes, and the
// host lays their pre-cut weights out in 16-channel chunks (make_h2)
#ifndef NST_H2_SHORTK_CIN_VALUE
#define NST_H2_SHORTK_CIN_VALUE 128
#endif
constexpr int NST_H2_SHORTK_CIN = NST_H2_SHORTK_CIN_VALUE;      // (-DNST_H2_SHORTK_CIN_VALUE=512: experiment builds)

// One image (pyramid level) of a batched conv_bf3 launch; the layer's weights / channel counts are shared.
struct ConvImage {
    const float* in;
    const float* addend;
    const float* mask;
    float* out;
    const float* in2;
    const void* wt2_bf;
    unsigned* bits_out;
    const unsigned* bits_in;
    float* pool_out;
    int H, W;
    int tiles_x, tile_end;   // filled by the launcher
    // conv_h2 only
    const float* wt2_f32;
    const unsigned* amax_in;
    const unsigned* amax_in2;
    const unsigned* amax_w2;
    unsigned* amax_out;
    const unsigned* pcode_in;
    unsigned* pcode_out;
    int in2_row0, in2_rows;
};
struct ConvBatch {
    ConvImage img[8];
    int n;
    const void* wt_bf;
    const float* bias;
    int Cin, Cout, Cin2, relu;
    const void* wt_h2;       // conv_h2 only
    float wt_h2_inv;
    int unpool;              // every image's `in` is a pooled gradient to be un-pooled through pcode_in
    int mfma16, wg256, tile_rows;   // conv_h2: see ConvParams
    int persist;             // conv_h2: in: nst_options.h2_persist; the launcher clears it where the persistent form does not apply
    int total_tiles;         // conv_h2: filled by the launcher (tiles x output-channel tiles)
    const void* wt_wino;     // conv_wino: the layer's transformed weights in fragment order (nullptr: none)
    float wt_wino_inv;
};

// conv_wino.hip: forward 3x3 convolution as 1-D Winograd F(2,3) in the f16x2 arithmetic (nst_options.h2_winograd)
hipError_t conv_wino_init_device();
bool conv_wino_eligible(const ConvBatch& b);
hipError_t launch_conv_wino_batch(const ConvBatch& b, hipStream_t stream);

// conv_mfma.hip
hipError_t conv_mfma_init_device();
hipError_t launch_conv_mfma(const ConvParams& p, int taps, hipStream_t stream);
// number of channel-chunk splits launch_conv_mfma uses for a 3x3 layer of this shape (1 = none)
int conv_ksplit(int H, int W, int Cin, int Cout);

hipError_t launch_conv_splitk_finish(const ConvParams& p, hipStream_t stream);

// conv_bf3.hip: the same 3x3 convolution on the bf16 matrix pipe with 3-piece operands (fp32-level accuracy)
hipError_t conv_bf3_init_device();
hipError_t launch_conv_bf3(const ConvParams& p, hipStream_t stream);
hipError_t launch_conv_bf3_batch(const ConvBatch& b, hipStream_t stream);
int conv_bf3_ksplit(int H, int W, int Cin, int Cout);

// conv_h2.hip: the same on the fp16 matrix pipe with 2-piece scaled operands (3 MFMAs per product block)
hipError_t conv_h2_init_device();
hipError_t launch_conv_h2(const ConvParams& p, hipStream_t stream);
hipError_t launch_conv_h2_batch(const ConvBatch& b, hipStream_t stream);
// absmax of n floats into NST_AMAX_SLOTS slots (atomic max; zero them beforehand)
hipError_t launch_absmax_slots(const float* x, size_t n, unsigned* slots, hipStream_t stream);

// conv_first.hip: conv1_1 (3 -> 64) forward from the planar image, and its input gradient
// wk: [28][64] (k = c*9 + ky*3 + kx, row 27 zero); bias [64]; out NHWC 64, ReLU applied.
// bits_out (nullable): ReLU bit-mask of the output, [H*W][2] words; amax_out (nullable): NST_AMAX_SLOTS words
// receiving the absmax of the output (atomic max)
hipError_t launch_conv1_1_fwd(const float* x, int H, int W, const float* wk, const float* bias, float* out,
                              unsigned* bits_out, unsigned* amax_out, hipStream_t stream);
// g: [H][W][64] gradient w.r.t. the pre-ReLU conv1_1 output; wd: [9][64][4] flipped taps
// (wd[t][co][c] = W[co][c][2-ky][2-kx], c = 3 unused 0); gx planar (3,H,W), overwritten.
// amax_g: the absmax slots of g (the matrix-pipe form), or null (fp32 on the VALU)
hipError_t launch_conv1_1_dgrad(const float* g, int H, int W, const float* wd, const unsigned* amax_g, float* gx,
                                hipStream_t stream);

// pixel_ops.hip ---------------------------------------------------------------------------------
// 2x2/2 max pool (floor) over NHWC, C % 4 == 0
hipError_t launch_maxpool_fwd(const float* in, int H, int W, int C, float* out, hipStream_t stream);
// gin[y][x][c] = (a[y][x][c] is the first maximum of its window and a > 0) ? gpool[y/2][x/2][c] : 0
// (max_pool2d backward fused with the ReLU mask of the activation `a` that was pooled)
hipError_t launch_maxpool_bwd_relu(const float* a, const float* gpool, int H, int W, int C, float* gin,
                                   hipStream_t stream);
// planar (C,h,w) <-> NHWC
hipError_t launch_chw_to_hwc(const float* src, int C, int H, int W, float* dst, hipStream_t stream);
hipError_t launch_hwc_to_chw(const float* src, int C, int H, int W, float* dst, hipStream_t stream);
// dst = (src > 0 ? g : 0) elementwise over n floats (n % 4 == 0)
hipError_t launch_relu_mask(const float* act, const float* g, size_t n, float* dst, hipStream_t stream);
// dst += src over n floats
hipError_t launch_add_inplace(float* dst, const float* src, size_t n, hipStream_t stream);

// general bicubic (A=-0.75, align_corners=False, clamped taps) down-sample of planar (C,h,w) to
// (C,oh,ow) and its transpose; the pyramid uses oh=h/2, ow=w/2.
hipError_t launch_bicubic_down(const float* x, int C, int h, int w, int oh, int ow, float* y, hipStream_t stream);
// gx (C,h,w): accumulate != 0 -> gx += transpose(gy), else gx = transpose(gy)
hipError_t launch_bicubic_down_bwd(const float* gy, int C, int h, int w, int oh, int ow, float* gx, int accumulate,
                                   hipStream_t stream);

// total variation: partial sums of |dx| and |dy| (NST_TV_BLOCKS x 2 doubles in `partial`)
constexpr int TV_BLOCKS = 1024;
// row window [row0, row0 + rows) of every channel (rows <= 0: all rows): the sums / the gradient of those rows only
hipError_t launch_tv_partial(const float* y, int C, int h, int w, double* partial, hipStream_t stream, int row0 = 0,
                             int rows = 0);
// reduces the partials (fixed order), writes means to scal[0..1]; if grad: grad (+)= weight * d tv/dy
hipError_t launch_tv_finish(const float* y, int C, int h, int w, const double* partial, float weight, float* grad,
                            int accumulate, float* means, hipStream_t stream, int row0 = 0, int rows = 0,
                            const float* given_means = nullptr, double nx = 0, double ny = 0);

// content: sum((a - t)^2) partials and, if g != nullptr, g = coef * (a - t) (coef = cw*2/n)
constexpr int MSE_BLOCKS = 256;
hipError_t launch_mse_grad(const float* a, const float* t, size_t n, float coef, float* g, double* partial,
                           hipStream_t stream);

// scalar plumbing of the stripe closure (pixel_ops.hip)
hipError_t launch_sum_doubles(const double* p, int n, int stride, int offset, float* out, hipStream_t stream);
hipError_t launch_window_scalars(const float* sums, double nx, double ny, float* means, double* partial, int n,
                                 hipStream_t stream);

// prepare / unprepare
hipError_t launch_prepare_img(const float* hwc, int h, int w, float* chw, hipStream_t stream);
hipError_t launch_unprepare_img(const float* chw, int h, int w, float* hwc, hipStream_t stream);

// loss assembly -----------------------------------------------------------------------------------
struct LevelLossInputs {
    const double* content_partial;   // MSE_BLOCKS doubles
    size_t content_n;
    const double* style_partial[5];  // gram_finish_blocks(C) doubles each: partial sums of (G-Gt)^2
    int style_c[5];                  // C of each style layer (mse mean over C*C)
    const float* tv_means;           // 2 floats (mean_x, mean_y)
    int owned;                       // 0: level computed by another rank, its row is written as zeros
};
struct LossAssembly {
    LevelLossInputs lv[8];
    int levels;
    float cw, sw, tvw;
    float* out;                      // 4*levels + 1
};
hipError_t launch_loss_assemble(const LossAssembly& la, hipStream_t stream);

// gram.hip -----------------------------------------------------------------------------------------
// partial Gram of NHWC f (N pixels x C): slabs part[s][C][C] (upper-triangle tiles only)
hipError_t gram_init_device();
int gram_nsplit(int C, size_t N);
// amax (nullable): NST_AMAX_SLOTS-word absmax record of f -> the fp16-piece kernel (3 MFMAs per product block)
hipError_t launch_gram_partial(const float* f, size_t N, int C, int nsplit, const unsigned* amax, float* part,
                               hipStream_t stream);
// G = (sum_s part[s]) / divisor (fixed order).  If target: mse_out[0] = sum((G-Gt)^2) (double) and
// S = coef * (G - Gt) (C x C, for the backward 1x1 conv).  gram_out / target / S / mse_partial nullable;
// mse_partial: gram_finish_blocks(C) doubles.  `nslabs` = gram_nslabs(C, nsplit).
int gram_nslabs(int C, int nsplit);
// Several Gram matrices in two partial launches (one per tile shape) + one finish launch.  Per item the caller sets
// f, N, C, amax, part (its own gram_nsplit(C, N) x C x C floats) and the finish arguments; the rest is filled in.
constexpr int NST_GRAM_BATCH_MAX = 16;
struct GramItem {
    const float* f; size_t N; int C; const unsigned* amax; float* part;
    float divisor; const float* target; float coef; float* gram_out; float* S; unsigned short* S_bf; unsigned* S_amax;
    double* mse_partial;
    int nsplit; size_t pix_per_split; int part_end, finish_end;     // filled by the launcher
};
struct GramBatch { GramItem it[NST_GRAM_BATCH_MAX]; int n; };
hipError_t launch_gram_batch(const GramBatch& b, hipStream_t stream);
#define NST_GRAM_FINISH_EPB 128      // elements of G a block of the finish pass owns (one partial sum of (G-Gt)^2 each)
int gram_finish_blocks(int C);
// S_amax (nullable): NST_AMAX_SLOTS words receiving the absmax of S (atomic max; zero them beforehand).
hipError_t launch_gram_finish(const float* part, int nslabs, int C, float divisor, const float* target, float coef,
                              float* gram_out, float* S, unsigned short* S_bf, unsigned* S_amax, double* mse_partial,
                              hipStream_t stream);

// image_ops.hip: job set-up on the device (pyramid resize, structured-noise initial image) ---------------------
hipError_t launch_resize_hwc(const float* src, int h, int w, int C, float* dst, int oh, int ow, hipStream_t stream);
hipError_t launch_gather_rows(const float* src, const long long* perm, size_t n, int C, float* dst, hipStream_t stream);
hipError_t launch_gauss_mask_acc(float* acc, const float* src, int h, int w, int C, double central, double peripheral,
                                 double disp, hipStream_t stream);
// weight (double, h*w*C) = 5 nf / (5 + blur(clip(|sobel5(content)|, 0, 100))); tmp0/tmp1: h*w*C doubles each
hipError_t launch_blend_weight(const float* content, int h, int w, int C, double noise_factor, double* tmp0, double* tmp1,
                               double* weight, hipStream_t stream);
hipError_t launch_blend_init(const float* content, const float* noise, const double* weight, size_t n, float* out,
                             hipStream_t stream);
hipError_t launch_scale(const float* src, float alpha, size_t n, float* dst, hipStream_t stream);

// vector_ops.hip: optimiser arithmetic over the n pixel floats ---------------------------------------
constexpr int RED_BLOCKS = 256;
// out[0] = sum(a*b) as float (double accumulation inside), deterministic two-stage
hipError_t launch_dot(const float* a, const float* b, size_t n, double* scratch, float* out, hipStream_t stream);
// out[0] = max|a|, out[1] = sum|a|
hipError_t launch_absmax_abssum(const float* a, size_t n, double* scratch, float* out, hipStream_t stream);
// y = alpha * x + beta * y'  variants
hipError_t launch_dot_partial(const float* a, const float* b, size_t n, double* scratch, hipStream_t stream);   // RED_BLOCKS partials
// one history pair of the L-BFGS two-loop recursion on the device (see vector_ops.hip)
hipError_t launch_lbfgs_pair(const double* sin, float ro, float* al, int second, const float* x, float* y, const float* nxt,
                             size_t n, double* sout, hipStream_t stream);
// L-BFGS direction from inner products (vector_ops.hip): out[j*3 + {0,1,2}] = vecs[j] . {a, b, c};  d = h q0 + sum coef[j] vecs[j]
int multi_dot_blocks(size_t n);                                       // scratch: multi_dot_blocks(n) * nvec * 3 doubles
hipError_t launch_multi_dot(const float* const* vecs_dev, int nvec, const float* a, const float* b, const float* c, size_t n,
                            double* scratch, float* out, hipStream_t stream);
hipError_t launch_multi_axpy(const float* const* vecs_dev, const float* coef_dev, int nvec, const float* q0, float h, float* d,
                             size_t n, hipStream_t stream);
hipError_t launch_axpy(float alpha, const float* x, float* y, size_t n, hipStream_t stream);               // y += alpha*x
hipError_t launch_axpy_dev(const float* alpha_dev, float sign, const float* x, float* y, size_t n, hipStream_t stream);
hipError_t launch_scale_copy(float alpha, const float* x, float* y, size_t n, hipStream_t stream);         // y = alpha*x
hipError_t launch_add_scaled(const float* a, float alpha, const float* b, float* out, size_t n, hipStream_t stream);  // out = a + alpha*b
hipError_t launch_zero(void* out, size_t n_words, hipStream_t stream);                                      // out[0..n) = 0 (32-bit words, 16-byte aligned)
hipError_t launch_copy(const float* a, float* out, size_t n, hipStream_t stream);                          // out = a (16-byte aligned)
hipError_t launch_sub(const float* a, const float* b, float* out, size_t n, hipStream_t stream);           // out = a-b
hipError_t launch_adam(float* x, const float* g, float* m, float* v, size_t n, float beta2, float one_m_b1, float one_m_b2,
                       float eps, float step_size, float bc2_sqrt, hipStream_t stream);

}  // namespace nst
