/*
 * nst_hip.h - C ABI of libnst_hip.so, the MI355X (gfx950) implementation of the pyramid
 * neural-style-transfer hot path.
 *
 * The reference (irenemizus/ArtStyleTransfer) has no FFI of its own: its hot path is Python on top
 * of torch ops.  Each entry point below names the reference interface it replaces
 * (file:line in the reference checkout; "torch:" = the torch package the reference calls into).
 *
 * Conventions
 *   - every function returns 0 on success, a negative NST_E_* code on failure, never throws;
 *     nst_last_error(ctx) returns a description of the last failure on that context
 *     (nst_last_error(NULL): the last failure of a call that had no context, thread-local).
 *   - the caller owns every image / gradient / optimiser buffer and passes raw DEVICE pointers
 *     plus the hipStream_t (as void*) the work must be ordered on; the context owns the VGG
 *     weights (pre-transformed), the per-level targets and the activation workspace.
 *   - images on the device are fp32 planar (3, H, W) in the reference's "prepared" domain
 *     (RGB * 255 - ImageNet mean: neural_style_transfer.py:375-383), exactly the storage of the
 *     (1,3,H,W) torch tensor the reference optimises.
 *   - every entry point binds the context's device itself (callers hop between thread-pool
 *     threads: neural_style_transfer.py:206) and is re-entrant across contexts.
 */
#ifndef NST_HIP_H
#define NST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NST_OK 0
#define NST_E_ARG (-1)      /* bad argument (null pointer, bad size, unknown enum) */
#define NST_E_STATE (-2)    /* call order violated (e.g. closure before targets) */
#define NST_E_HIP (-3)      /* a HIP runtime call failed; see nst_last_error */
#define NST_E_NOMEM (-4)

#define NST_VGG19_CONVS 13  /* conv1_1 ... conv5_1 (torchvision features[0:30]) */
#define NST_MAX_LEVELS 8
#define NST_LOSS_ROW 4      /* per level: total, content, style, tv */

typedef struct nst_ctx nst_ctx;
typedef struct nst_opt nst_opt;
typedef struct nst_comm nst_comm;

/* library / device ------------------------------------------------------------------------ */
int nst_version(void);
const char* nst_last_error(const nst_ctx* ctx);
/* number of visible HIP devices (does not initialise a context on any of them) */
int nst_device_count(int* count);

/* context: VGG19 feature network of math_utils.prepare_model (math_utils.py:9-23) and
 * neural_nets.Vgg19.__init__ (neural_nets.py:17-51).  weights[i]: HOST pointer, (Cout,Cin,3,3)
 * fp32 in torchvision order conv1_1..conv5_1; biases[i]: HOST pointer (Cout).  The context
 * keeps device copies re-laid-out for the forward and the input-gradient kernels. */
int nst_ctx_create(int device, const float* const* weights, const float* const* biases, nst_ctx** out);
void nst_ctx_destroy(nst_ctx* ctx);

/* The same with the execution options as ARGUMENTS (nst_ctx_create = nst_ctx_create_ex with opts == NULL).  A field
 * left at -1 takes the environment variable named beside it, read ONCE here, and otherwise the stated default; the
 * environment is never consulted again during the life of the context.  There is no counterpart in the reference
 * (torch picks its kernels by itself); the options exist for the parity tests, which hold the alternative
 * arithmetic / schedules against the default one. */
#define NST_CONV_F32 0      /* fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference's arithmetic type, exactly */
#define NST_CONV_BF16X3 1   /* bf16 matrix pipe, three bf16 pieces that sum to the fp32 value exactly, 6 MFMAs */
#define NST_CONV_F16X2 2    /* fp16 matrix pipe, two scaled fp16 pieces per operand, 3 MFMAs per product block (default) */
typedef struct nst_options {
    int struct_size;      /* sizeof(nst_options), filled by nst_options_default */
    int conv_mode;        /* NST_CONV_*; -1: env NST_CONV (f32 | bf16x3 | f16x2), default f16x2 */
    int batched;          /* 1: one conv launch per layer covering every pyramid level; 0: per level; -1: env NST_BATCH, default 1 */
    int single_stream;    /* per-level schedule only - 1: all levels on the caller's stream; -1: env NST_SINGLE_STREAM, default 0 */
    int use_graph;        /* 1: replay the closure as a hipGraph; -1: env NST_GRAPH, default 0 */
    int h2_band_rows;     /* >= 16: run the f16x2 per-level launches in row bands of that many rows (the path tensors beyond
                             4 GiB take, forced onto small images); -1: env NST_H2_BAND_ROWS, default 0 = only when needed */
    int lbfgs_gram;       /* 1: L-BFGS direction from inner products (two passes over the history); 0: the sequential
                             recursion; -1: env NST_LBFGS_GRAM, default 1 */
    int h2_mfma16;        /* f16x2 convolutions with 32-channel chunks: v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16
                             (same products, same accumulation chains per output; the chip clocks higher under it): 0 = never,
                             1 = on the 8-row x 128-channel shape (under-filled launches: +6 ... 11 % per launch),
                             2 = also on the 16-row shape where the launch does not un-pool (no gain measured), 3 = on every
                             32-channel-chunk shape (slower; both kept for experiments);
                             -1: env NST_H2_MFMA16, default 1 */
    int h2_wg256;         /* f16x2 convolutions, the 16x16-pixel x 128-channel tile: 1 = 256-thread workgroups, one wave per SIMD
                             with a 64 x 128 wave tile (accumulators in AGPRs), 0 = 512 threads with 64 x 64 wave tiles;
                             -1: env NST_H2_WG256, default 0 */
    int h2_tile_rows;     /* f16x2 convolutions with 128-channel tiles and 32-channel chunks: 4 / 8 / 16 = pixel rows per workgroup tile,
                             0 = chosen per launch from the number of workgroups; -1: env NST_H2_TILE_ROWS, default 0 */
    int gram_overlap;     /* f16x2 closure: 1 = the Gram matrices of relu1_1 ... relu3_1 (HBM-bound) run on a side stream of the
                             context under the MFMA-bound convolutions of conv3_2 ... conv5_1, joined before the backward pass;
                             0 = everything in order on the caller's stream; -1: env NST_GRAM_OVERLAP, default 0 (measured: no gain) */
    int h2_persist;       /* f16x2 batched launches: 1 = persistent workgroups (one grid that fills the chip once, each workgroup
                             walks its share of the tiles with the K pipeline chained from one tile into the next);
                             0 = one workgroup per tile; -1: env NST_H2_PERSIST, default 0 (measured: the chained form's extra scalar state
                             costs more than the hidden prologues gain) */
    int level_split;      /* f16x2 batched closure: 1 = the top pyramid level's chain on the caller's stream and the lower levels'
                             chain on a side stream of the context, joined before the gradients are merged; 0 = one launch per
                             layer over all levels; -1: env NST_LEVEL_SPLIT, default 0 */
    int h2_winograd;      /* f16x2 batched closure: 1 = the forward and input-gradient launches with Cin >= 256 and Cout a
                             multiple of 128 that have no second (Gram) source - 14 of the 24 of a closure - run as a 1-D Winograd
                             F(2,3) along x (conv_wino.hip: 1.5x fewer MFMAs there; the feature maps are no further from an fp64
                             evaluation than the direct path's); 0 = direct convolution everywhere;
                             -1: env NST_H2_WINOGRAD, default 1 */
} nst_options;
void nst_options_default(nst_options* opts);
int nst_ctx_create_ex(int device, const float* const* weights, const float* const* biases, const nst_options* opts,
                      nst_ctx** out);

/* pyramid geometry of one job: levels_num levels, level 0 = (H0, W0), level l = previous // 2
 * (neural_style_transfer.py:170-176).  Allocates the activation workspace of every level. */
int nst_job_configure(nst_ctx* ctx, int levels_num, int H0, int W0);

/* LossBuilder.__init__ (neural_style_transfer.py:68-82): target content representation
 * ReLU(conv4_2) of the content image and the 5 target Gram matrices of the style image of one
 * level.  content: device (3,h,w) of that level's size; style: device (3,hs,ws), any size. */
int nst_level_set_targets(nst_ctx* ctx, int level, const float* content, const float* style,
                          int hs, int ws, void* stream);

/* optimizer_step_callback without its LR decay and prints (neural_style_transfer.py:152-199) =
 * sum over levels of LossBuilder.build (:84-112) on the bicubic 1/2 chain of x (:170-176),
 * then backward (:193).  x, grad: device (3,H0,W0).  losses: device, NST_LOSS_ROW*levels+1
 * floats = per level (total, content, style, tv) unweighted components as the reference
 * prints them, then the grand total.  Asynchronous on `stream`. */
int nst_closure(nst_ctx* ctx, const float* x, float content_weight, float style_weight,
                float tv_weight, float* grad, float* losses, void* stream);

/* The same closure restricted to the pyramid levels whose bit is set in level_mask: the level-sharded
 * form of BASELINE config 4 (rank r owns some levels; loss = sum over levels, so the pixel gradients and
 * loss rows of the ranks add up: one all-reduce(sum) of `grad` (3*H0*W0 floats) and `losses`).  Rows of
 * levels not in the mask are zeros; the last element is the sum of the owned level totals. */
int nst_closure_levels(nst_ctx* ctx, const float* x, float content_weight, float style_weight,
                       float tv_weight, unsigned level_mask, float* grad, float* losses, void* stream);

/* torch.optim.Adam / torch.optim.LBFGS as constructed at neural_style_transfer.py:134-136,
 * driving nst_closure, including the closure's `lr *= 0.999` (:155-158).  kind: 0 = adam
 * (torch:optim/adam.py:457-546, betas (0.9,0.999), eps 1e-8), 1 = lbfgs
 * (torch:optim/lbfgs.py:332-537: max_iter 1, strong_wolfe, history 100; lbfgs_max_eval is the
 * constructor's max_eval: 1 = torch 2.10 behaviour of the reference's arguments, 26 = legacy
 * line search). */
#define NST_OPT_ADAM 0
#define NST_OPT_LBFGS 1
int nst_opt_create(nst_ctx* ctx, int kind, float lr_start, int lbfgs_max_eval, nst_opt** out);
void nst_opt_destroy(nst_opt* opt);

typedef struct nst_step_info {
    int closures;        /* closure evaluations made by this step (Adam 1, L-BFGS >= 1) */
    int total_closures;  /* the reference's `step` counter after this call */
    int accepted;        /* L-BFGS: 1 if x moved, 0 if the trial was rejected; Adam: 1 */
    float loss;          /* loss of the FIRST closure of this step (what optimizer.step returns) */
    float lr;            /* learning rate after this step's decays */
    float t;             /* L-BFGS step length taken (0 when rejected) */
    int history;         /* L-BFGS: curvature pairs held after this step (0..100) */
} nst_step_info;

/* one optimizer.step(closure) (neural_style_transfer.py:205-206).  x: device (3,H0,W0), updated
 * in place.  losses_host (nullable): HOST buffer of closures_capacity*(NST_LOSS_ROW*levels+1)
 * floats receiving the loss rows of every closure made.  Synchronous for L-BFGS (host control
 * flow needs the loss), asynchronous on `stream` for Adam when losses_host is NULL. */
int nst_opt_step(nst_opt* opt, float* x, float content_weight, float style_weight, float tv_weight,
                 float* losses_host, int closures_capacity, nst_step_info* info, void* stream);

/* Level sharding inside the optimiser drivers: every closure the driver evaluates covers only
 * `level_mask`, then calls hook(user) - which must all-reduce(sum) the `grad` and `losses` DEVICE buffers
 * given here over the ranks, ordered on the stream passed to nst_opt_step - before the driver reads
 * them.  The update itself is replicated (deterministic), so no broadcast is needed.  grad: 3*H0*W0
 * floats, losses: NST_LOSS_ROW*levels+1 floats, both owned by the caller and alive as long as `opt`. */
typedef void (*nst_reduce_hook)(void* user);
int nst_opt_shard_levels(nst_opt* opt, unsigned level_mask, float* grad, float* losses, nst_reduce_hook hook,
                         void* user);

/* The same with the collective behind the ABI: every closure the driver evaluates covers `level_mask`, then ONE
 * ncclAllReduce(sum, fp32) over `comm` of a single buffer holding the 3*H0*W0 gradient floats followed by the
 * NST_LOSS_ROW*levels+1 loss scalars, on the stream passed to nst_opt_step; the grand total is re-formed from the level rows
 * in level order, so L-BFGS' accept test takes the same branch as the unsharded run.  comm == NULL switches sharding off. */
int nst_opt_shard_levels_comm(nst_opt* opt, unsigned level_mask, nst_comm* comm);

/* curvature pairs currently held by L-BFGS and the optimiser's iteration count (Adam: its step count k) */
int nst_opt_history(const nst_opt* opt, int* pairs, int* n_iter);

/* ---- RCCL communicator (one rank per GPU; the reference has no collective: neural_style_transfer.py:236-245).
 * librccl is resolved at run time; without it these return NST_E_STATE.  Bootstrap: rank 0 calls nst_comm_unique_id and
 * hands the NST_COMM_ID_BYTES bytes to the other ranks by whatever channel the host program has (file, socket, MPI,
 * torch.distributed); then every rank calls nst_comm_create (collective). */
#define NST_COMM_ID_BYTES 128
int nst_comm_unique_id(void* id);
int nst_comm_create(int device, int rank, int world, const void* id, nst_comm** out);
void nst_comm_destroy(nst_comm* comm);
/* rank / world of the communicator and what it has carried so far (any pointer may be NULL) */
int nst_comm_info(const nst_comm* comm, int* rank, int* world, long* calls, double* bytes);
/* in-place all-reduce(sum) of n floats at the DEVICE pointer buf, ordered on `stream` */
int nst_comm_allreduce_sum(nst_comm* comm, float* buf, size_t n, void* stream);

/* ---- the optimisers' update arithmetic alone, exported for unit parity -------------------------------
 * One torch.optim.Adam update (torch:optim/adam.py:457-546, betas (0.9, 0.999), eps 1e-8) of the n floats at x from the
 * gradient g and the state (m = exp_avg, v = exp_avg_sq), all DEVICE pointers updated in place; k: the step count of
 * this update (1-based), lr: the group's learning rate at this update (neural_style_transfer.py:134, :155-158). */
int nst_adam_step(nst_ctx* ctx, float* x, const float* g, float* m, float* v, size_t n, int k, double lr, void* stream);
/* The L-BFGS direction d = -H g of torch:optim/lbfgs.py:396-442 from m curvature pairs: y[i] = old_dirs[i],
 * s[i] = old_stps[i] (HOST arrays of m DEVICE pointers, oldest first), ro[i] = 1 / (y_i . s_i) (HOST floats),
 * h_diag = H_diag; g, d: DEVICE, n floats.  form 0: from inner products (what the driver runs by default), form 1: the
 * sequential two-loop recursion in torch's arithmetic order.  Synchronous. */
int nst_lbfgs_direction(nst_ctx* ctx, const float* g, const float* const* y, const float* const* s, const float* ro, int m,
                        float h_diag, size_t n, int form, float* d, void* stream);

/* ---- standalone pieces of the path, exported for unit parity -------------------------------- */

/* Vgg19.forward (neural_nets.py:53-68): x device (3,h,w) -> the six maps, each written as
 * (C,h_i,w_i) planar fp32 (reference layout) into outs[i] (device; NULL to skip). */
int nst_vgg_features(nst_ctx* ctx, const float* x, int h, int w, float* const* outs, void* stream);
/* The same forward pass, all 13 post-ReLU conv outputs conv1_1 ... conv5_1 as (C,h_l,w_l) planar fp32 into outs[l]
 * (device; NULL to skip): the decisions (unit on / pooling arg-max) nst_vgg_features_backward of the same x takes. */
int nst_vgg_activations(nst_ctx* ctx, const float* x, int h, int w, float* const* outs, void* stream);
/* d(sum_i <outs_i, gouts_i>)/dx through the network: gouts[i] device (C,h_i,w_i) or NULL. */
int nst_vgg_features_backward(nst_ctx* ctx, const float* x, int h, int w, const float* const* gouts,
                              float* gx, void* stream);
/* The post-ReLU output of conv layer `layer` (0 = conv1_1 ... 12 = conv5_1) that the LAST closure / window pass left in the
 * workspace of pyramid level `level`, written as (C,h_l,w_l) planar fp32 to out (device).  The parity tests read the
 * ReLU and max-pool DECISIONS of the device pass from it (a unit is on where the value is > 0; a pooling window passes its
 * gradient to its first maximum) and hand them to the oracle, so that gradients are compared under equal decisions. */
int nst_level_activation(nst_ctx* ctx, int level, int layer, float* out, void* stream);
/* The image of pyramid level `level` >= 1 that the last closure evaluated - the bicubic 1/2 chain of x
 * (neural_style_transfer.py:170-176) - as (3,h_l,w_l) planar fp32 to out (device).  The total-variation term takes
 * sign(y_i - y_j) of neighbouring pixels: on flat image regions those differences are rounding noise of the down-sampling,
 * so the parity tests read the signs the device pass took from this image. */
int nst_level_image(nst_ctx* ctx, int level, float* out, void* stream);
/* math_utils.gram_matrix (math_utils.py:26-34): f device (C,h,w) -> gram device (C,C). */
int nst_gram(nst_ctx* ctx, const float* f, int C, int h, int w, int normalize, float* gram, void* stream);
/* math_utils.total_variation (math_utils.py:37-41): value (device scalar) and, if grad != NULL,
 * grad (C,h,w) = d tv / d y. */
int nst_total_variation(nst_ctx* ctx, const float* y, int C, int h, int w, float* value, float* grad,
                        void* stream);
/* F.interpolate(x, size=(h//2,w//2), mode='bicubic') (neural_style_transfer.py:173-176) and its
 * transpose (autograd backward); x (C,h,w) -> y (C,h/2,w/2); gy -> gx (overwritten). */
int nst_bicubic_half(nst_ctx* ctx, const float* x, int C, int h, int w, float* y, void* stream);
int nst_bicubic_half_backward(nst_ctx* ctx, const float* gy, int C, int h, int w, float* gx, void* stream);
/* prepare_img / unprepare_img (neural_style_transfer.py:375-393): HWC [0,1] RGB <-> planar
 * prepared, both on the device. */
int nst_prepare_img(nst_ctx* ctx, const float* hwc, int h, int w, float* chw, void* stream);
int nst_unprepare_img(nst_ctx* ctx, const float* chw, int h, int w, float* hwc, void* stream);

/* ---- job set-up on the device (the host-side OpenCV work of the reference's job driver) ----------- */

/* cv2.resize(img, (nw, nh), interpolation=cv2.INTER_CUBIC) for float32 HWC images (bicubic, A = -0.75, half-pixel
 * centres, replicate border, no antialias) - `resize` (neural_style_transfer.py:211-226) and the noise-map
 * up-sampling (:304-305); src (h,w,channels) -> dst (nh,nw,channels), both on the device. */
int nst_resize_bicubic(nst_ctx* ctx, const float* src, int h, int w, int channels, float* dst, int nh, int nw, void* stream);
/* dst[i][:] = src[perm[i]][:]: the row shuffle of make_style_noise (:422-432) with the permutation drawn on the
 * host by np.random.permutation (so the reference's RNG stream is reproduced); perm: device int64[rows]. */
int nst_gather_rows(nst_ctx* ctx, const float* src, const long long* perm, size_t rows, int channels, float* dst, void* stream);
/* acc += (src ? src : 1) * gaussian_mask((h,w), central, peripheral, dispersion) (gaussian_mask :396-418 and the
 * accumulation at :283-284, :311-313); acc, src: device float32 (h,w,channels); mask evaluated in double. */
int nst_gaussian_mask_accumulate(nst_ctx* ctx, float* acc, const float* src, int h, int w, int channels, double central,
                                 double peripheral, double dispersion, void* stream);
/* out = ((1 - nr) * content + nr * noise).astype(float32), nr = 5 nf / (5 + GaussianBlur_101,0.2(clip(|Sobel_5|, 0, 100)))
 * computed in double (:331-343, :355-358); all (h,w,channels) device float32.  Synchronous. */
int nst_noise_blend(nst_ctx* ctx, const float* content, const float* noise, int h, int w, int channels, double noise_factor,
                    float* out, void* stream);
/* dst = alpha * src (init_method 'random': 0.5 * noise, :351) */
int nst_scale(nst_ctx* ctx, const float* src, float alpha, size_t n, float* dst, void* stream);

/* ---- spatial sharding of one pyramid level: a context evaluates a horizontal STRIPE of a larger image ------------
 * (SURVEY 8(e) partition B with halo recompute.  There is no counterpart in the reference; the quantities are those of
 * neural_style_transfer.py:84-112 restricted to the rows a rank owns.)
 * The context is configured with nst_job_configure(ctx, 1, ext_rows, W0) and its targets set with
 * nst_level_set_targets(ctx, 0, <the same rows of the content image>, <the whole style image>, ...).  Its image `xs`
 * (3, ext_rows, W0) is rows [e0, e0 + ext_rows) of the (3, H0, W0) image: the rows the rank owns,
 * [row0, row0 + rows) in stripe coordinates, plus a halo on each interior side that covers the receptive field of
 * relu5_1 (78 rows; 96 keeps the boundaries multiples of 16).  row0 and the boundaries between stripes are multiples
 * of 16 (pooling alignment); only the bottom stripe may own a ragged last row group.
 *   nst_window_begin: forward pass; writes to `sums` (nst_window_sums_count floats, device) the un-normalised Gram
 *     sums of the five style maps, the content sum of squares and the two TV sums OVER THE OWNED ROWS.
 *   The caller adds the `sums` of all stripes (one all-reduce).
 *   nst_window_end: turns the summed `sums` into the style / content / TV terms of the FULL image (its normalisers),
 *     runs the backward pass for the loss terms of the owned rows and writes d loss / d xs to gxs (3, ext_rows, W0);
 *     the caller adds the stripes' gradients into the full image (overlap-add, one all-reduce).  losses[0..3] = (total,
 *     content, style, tv) of the level, losses[4] = total - identical on every rank.
 * Nothing else may run on the context between begin and end. */
int nst_window_sums_count(size_t* count);
int nst_window_begin(nst_ctx* ctx, const float* xs, int row0, int rows, int H0, float* sums, void* stream);
int nst_window_end(nst_ctx* ctx, const float* xs, int row0, int rows, int H0, float content_weight, float style_weight,
                   float tv_weight, float* sums, float* gxs, float* losses, void* stream);

/* arithmetic of the 3x3 convolutions of this context (nst_options.conv_mode): NST_CONV_F16X2 (default: both operands
 * cut into two scaled fp16 pieces, main and cross terms in separate fp32 accumulators; error measured against fp64 =
 * an fp32 MFMA's), NST_CONV_BF16X3 or NST_CONV_F32. */
int nst_conv_mode(const nst_ctx* ctx);

/* workspace bytes currently held by the context (activations, gradients, history, targets) */
int nst_ctx_bytes(const nst_ctx* ctx, size_t* bytes);

/* wall time in ms of the kernels of the last nst_closure on `ctx`, measured with HIP events on
 * the streams the kernels ran on (0 if timing was not enabled with nst_set_timing). */
int nst_set_timing(nst_ctx* ctx, int enabled);   /* 0 off, 1 whole closure, 2 + every kernel launch, 3 + only the 3x3 conv
                                                    launches, 4 + those of every fourth closure only */
int nst_last_closure_ms(nst_ctx* ctx, float* ms);
/* last closure, per kernel class: summed launch durations (ms), launches, algorithmic flops.
 * cls: 0 = 3x3 MFMA convolutions (forward + input gradient), 1 = Gram forward + its 1x1 backward,
 * 2 = conv1_1 forward + input gradient, 3 = streaming kernels (pool, bicubic, TV, MSE, reductions). */
int nst_last_closure_class(nst_ctx* ctx, int cls, float* ms, int* launches, double* flops);
/* the same accumulated over every closure since the last reset (timing mode 2); cls = -1: whole
 * closures (ms = summed closure time on the caller's stream, launches = closures). */
int nst_timing_totals(nst_ctx* ctx, int cls, double* ms, long* launches, double* flops, int reset);
/* executed matrix-pipe FLOPs of the launches accumulated in nst_timing_totals(cls): algorithmic FLOPs x the MFMAs the
 * arithmetic spends per product (f16x2: 3, bf16x3: 6, f32: 1), x 2/3 for the launches that ran as Winograd F(2,3). */
int nst_timing_mfma_flops(nst_ctx* ctx, int cls, double* mfma_flops);
/* debugging aid: prints one line per timed launch of the last closure (timing mode 2) to stderr */
int nst_dump_last_closure(nst_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* NST_HIP_H */
