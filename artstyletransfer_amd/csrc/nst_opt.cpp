// nst_opt.cpp - the two optimisers the reference constructs (neural_style_transfer.py:134-136)
// driving nst_closure on the device-resident pixel buffer:
//   Adam   torch:optim/adam.py:457-546  (single tensor, betas (0.9, 0.999), eps 1e-8)
//   L-BFGS torch:optim/lbfgs.py:332-537 (max_iter 1, strong_wolfe, history 100, tolerance_grad 1e-7,
//          tolerance_change 1e-9; max_eval is a parameter, see include/nst_hip.h)
// including the closure's own `lr *= 0.999` (neural_style_transfer.py:155-158).
// Vector arithmetic runs in vector_ops.hip; only scalars (loss, dots) come back to the host,
// because L-BFGS' control flow is host-side scalar logic in the reference too.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nst_hip.h"
#include "nst_kernels.h"

using namespace nst;

extern "C" int nst_internal_device(const nst_ctx* ctx);
extern "C" int nst_internal_levels(const nst_ctx* ctx);
extern "C" size_t nst_internal_pixels(const nst_ctx* ctx);
extern "C" int nst_internal_fail(nst_ctx* ctx, int code, const char* msg);
extern "C" void nst_internal_poison(void* p, size_t bytes);
extern "C" int nst_internal_zero_now(void* p, size_t bytes);      // a zero fill that has RUN when it returns (nst_api.cpp)
extern "C" int nst_internal_lbfgs_gram(const nst_ctx* ctx);
extern "C" void nst_internal_mark(nst_ctx* ctx, void* stream);

struct nst_opt {
    nst_ctx* ctx = nullptr;
    int kind = 0;
    size_t n = 0;
    int levels = 0;
    double lr = 10.0;            // python float in the reference
    int total_closures = 0;
    // device
    float* g = nullptr;          // gradient of the latest closure
    float* losses = nullptr;     // 4*levels+1
    double* scratch = nullptr;   // 2*RED_BLOCKS
    float* scal = nullptr;       // 4 floats
    float* al_dev = nullptr;     // L-BFGS: the al_i of the two-loop recursion, one per history pair
    // direction from inner products (default; NST_LBFGS_GRAM=0 selects the sequential recursion): SY[i][j] = s_i . y_j and
    // YY[i][j] = y_i . y_j kept on the host, one multi-dot pass and one multi-axpy pass over the history per step
    // instead of 2 launches per pair (update at 100 pairs, L=2: 3.5 -> 1.5 ms per step)
    bool gram_mode = false;
    const float** vec_dev = nullptr;   // 2*history device pointers: y_0..y_{m-1}, s_0..s_{m-1}
    float* coef_dev = nullptr;         // 2*history coefficients of the direction
    double* md_scratch = nullptr;      // multi_dot_blocks(n) * 2*history * 3
    float* md_out = nullptr;           // 2*history * 3
    unsigned char* pin2 = nullptr;     // page-locked staging: pointers | coefficients | results
    std::vector<double> SY, YY;        // history x history, row-major
    float* pinned = nullptr;     // page-locked host staging for the scalar read-backs (pageable targets make
                                 // hipMemcpyAsync stage through an internal buffer: ~50 us per read-back)
    // adam
    float* m = nullptr; float* v = nullptr; int k = 0;
    // lbfgs
    int max_eval = 1;
    int history = 100;
    int n_iter = 0;
    float* d = nullptr; float* prev_g = nullptr; float* xinit = nullptr; float* q = nullptr;
    std::vector<float*> old_dirs, old_stps;
    std::vector<float> ro;
    std::vector<float*> spare;   // free history vectors: slices of `pool`
    float* pool = nullptr;       // ONE allocation made by nst_opt_create holding every history vector (2*history + 2
                                 // of them): no hipMalloc - an implicit device synchronisation - inside a step
    hipEvent_t tail = nullptr;   // recorded after the last launch of every step: what nst_opt_destroy waits for
    hipStream_t tail_stream = nullptr;
    bool tail_set = false;
    bool want_rows = false;      // this step keeps the loss rows of its closures for the caller
    bool H_is_one = true; float H_diag = 1.f;
    bool t_is_float = true;      // t held as fp32 tensor value vs python double
    double t = 0.0;
    bool have_prev = false;
    // level sharding (BASELINE config 4)
    unsigned level_mask = 0xFFFFFFFFu;
    nst_reduce_hook hook = nullptr;
    void* hook_user = nullptr;
    float* own_g = nullptr; float* own_losses = nullptr;   // buffers this object allocated
    nst_comm* comm = nullptr;    // RCCL communicator (nst_opt_shard_levels_comm): all-reduce of `pack` per closure
    float* pack = nullptr;       // gradient (n floats, padded to 64) followed by the loss row: ONE collective buffer
    size_t pack_floats = 0;
    // per-step outputs
    std::vector<float> loss_rows;   // host copy of every closure's loss rows in this step
};

namespace {

#define OHIP(o, expr)                                                                                \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return nst_internal_fail((o)->ctx, NST_E_HIP, (std::string(#expr) + ": " + hipGetErrorString(_e)).c_str()); \
    } while (0)
#define OCHK(expr)                   \
    do {                             \
        int _r = (expr);             \
        if (_r != NST_OK) return _r; \
    } while (0)

int oalloc(nst_opt* o, float** p, size_t n) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float));
    if (e != hipSuccess) return nst_internal_fail(o->ctx, NST_E_NOMEM, "hipMalloc failed in optimiser");
    nst_internal_poison(*p, n * sizeof(float));
    return NST_OK;
}

// Scalars the host needs about the new gradient ride on the closure's own synchronisation (every separate read-back
// is a stream sync that leaves the GPU idle until the host has enqueued the next launches).
struct GradStats {
    bool want_abs = false;            // max|g| and sum|g|  (lbfgs.py: opt_cond / first-step t)
    const float* dot_with = nullptr;  // g . dot_with       (lbfgs.py: gtd_new of the line search)
    float gmax = 0.f, gsum = 0.f, gdot = 0.f;
};

// one closure evaluation at x: decays lr, fills o->g, returns the total loss (host) - synchronises once
int eval_closure(nst_opt* o, const float* x, float cw, float sw, float tvw, hipStream_t s, float* loss_out,
                 GradStats* gs = nullptr) {
    o->lr *= 0.999;                                                          // neural_style_transfer.py:155-158
    OCHK(nst_closure_levels(o->ctx, x, cw, sw, tvw, o->level_mask, o->g, o->losses, s));
    if (o->comm) OCHK(nst_comm_allreduce_sum(o->comm, o->pack, o->pack_floats, s));   // gradient + loss row, one call
    else if (o->hook) o->hook(o->hook_user);                                 // all-reduce(sum) over the ranks
    const size_t row = (size_t)NST_LOSS_ROW * o->levels + 1;
    const size_t off = o->loss_rows.size();
    o->loss_rows.resize(off + row);
    float* hrow = o->pinned + 8;                                             // [0..8): scalars, [8..): the loss row
    OHIP(o, hipMemcpyAsync(hrow, o->losses, row * sizeof(float), hipMemcpyDeviceToHost, s));
    float* r = o->pinned;
    if (gs && gs->want_abs) OHIP(o, launch_absmax_abssum(o->g, o->n, o->scratch, o->scal, s));
    if (gs && gs->dot_with) OHIP(o, launch_dot(o->g, gs->dot_with, o->n, o->scratch, o->scal + 2, s));
    if (gs) OHIP(o, hipMemcpyAsync(r, o->scal, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    OHIP(o, hipStreamSynchronize(s));
    if (o->comm) {
        // every level row has exactly one non-zero contributor, so the rows are exact; the grand total is re-formed from
        // them in level order - the association of the unsharded closure (neural_style_transfer.py:179-185) - so that
        // L-BFGS' `f_new < f` takes the same branch as the unsharded run
        float total = hrow[0];
        for (int l = 1; l < o->levels; ++l) total = total + hrow[(size_t)NST_LOSS_ROW * l];
        hrow[row - 1] = total;
    }
    std::memcpy(o->loss_rows.data() + off, hrow, row * sizeof(float));
    if (gs) { gs->gmax = r[0]; gs->gsum = r[1]; gs->gdot = r[2]; }
    o->total_closures += 1;                                                  // :198
    *loss_out = o->loss_rows[off + row - 1];
    return NST_OK;
}

int dot(nst_opt* o, const float* a, const float* b, hipStream_t s, float* out) {
    OHIP(o, launch_dot(a, b, o->n, o->scratch, o->scal, s));
    OHIP(o, hipMemcpyAsync(o->pinned, o->scal, sizeof(float), hipMemcpyDeviceToHost, s));
    OHIP(o, hipStreamSynchronize(s));
    *out = o->pinned[0];
    return NST_OK;
}
// two dot products, one synchronisation
int dot2(nst_opt* o, const float* a0, const float* b0, const float* a1, const float* b1, hipStream_t s, float* out0,
         float* out1) {
    float* r = o->pinned;
    OHIP(o, launch_dot(a0, b0, o->n, o->scratch, o->scal, s));
    OHIP(o, launch_dot(a1, b1, o->n, o->scratch, o->scal + 1, s));
    OHIP(o, hipMemcpyAsync(r, o->scal, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    OHIP(o, hipStreamSynchronize(s));
    *out0 = r[0]; *out1 = r[1];
    return NST_OK;
}
// g . d together with max|d|, sum|d|, one synchronisation
int dot_and_absstats(nst_opt* o, const float* g, const float* d, hipStream_t s, float* gtd, float* mx, float* sum) {
    float* r = o->pinned;
    OHIP(o, launch_absmax_abssum(d, o->n, o->scratch, o->scal, s));
    OHIP(o, launch_dot(g, d, o->n, o->scratch, o->scal + 2, s));
    OHIP(o, hipMemcpyAsync(r, o->scal, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    OHIP(o, hipStreamSynchronize(s));
    *mx = r[0]; *sum = r[1]; *gtd = r[2];
    return NST_OK;
}
int absstats(nst_opt* o, const float* a, hipStream_t s, float* mx, float* sum) {
    float* r = o->pinned;
    OHIP(o, launch_absmax_abssum(a, o->n, o->scratch, o->scal, s));
    OHIP(o, hipMemcpyAsync(r, o->scal, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    OHIP(o, hipStreamSynchronize(s));
    *mx = r[0]; *sum = r[1];
    return NST_OK;
}

// torch:optim/lbfgs.py:12-37 on host scalars
double cubic_interpolate(double x1, double f1, double g1, double x2, double f2, double g2, bool has_bounds,
                         double lo, double hi) {
    double xmin = has_bounds ? lo : std::min(x1, x2);
    double xmax = has_bounds ? hi : std::max(x1, x2);
    const double d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2);
    const double d2sq = d1 * d1 - g1 * g2;
    if (d2sq >= 0) {
        const double d2 = std::sqrt(d2sq);
        double min_pos;
        if (x1 <= x2) min_pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2));
        else min_pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2));
        return std::min(std::max(min_pos, xmin), xmax);
    }
    return (xmin + xmax) / 2.0;
}

struct LsResult { double t; float f; int evals; };

// torch:optim/lbfgs.py:40-209.  Gradients of bracket points are never needed by the caller with
// max_iter == 1 (only f, g.d and t are consumed), so no gradient clones are kept.
int strong_wolfe(nst_opt* o, float* x, double t, float f, float gtd, float d_norm, int max_ls, float cw, float sw,
                 float tvw, hipStream_t s, LsResult* res) {
    const double c1 = 1e-4, c2 = 0.9, tol_change = 1e-9;
    auto eval_at = [&](double tt, float* f_new, float* gtd_new) -> int {
        // x = x_init + t*d ; closure ; (x restored by the caller at the end)
        OHIP(o, launch_add_scaled(o->xinit, (float)tt, o->d, x, o->n, s));
        GradStats gs;
        gs.dot_with = o->d;
        OCHK(eval_closure(o, x, cw, sw, tvw, s, f_new, &gs));
        *gtd_new = gs.gdot;
        return NST_OK;
    };
    float f_new, gtd_new;
    OCHK(eval_at(t, &f_new, &gtd_new));
    int evals = 1;
    double t_prev = 0; float f_prev = f; float gtd_prev = gtd;
    bool done = false;
    int ls_iter = 0;
    double br[2] = {0, 0}; float br_f[2] = {0, 0}; float br_gtd[2] = {0, 0};
    int nbr = 0;
    while (ls_iter < max_ls) {
        if (f_new > (float)(f + (float)(c1 * t) * gtd) || (ls_iter > 1 && f_new >= f_prev)) {
            br[0] = t_prev; br[1] = t; br_f[0] = f_prev; br_f[1] = f_new; br_gtd[0] = gtd_prev; br_gtd[1] = gtd_new; nbr = 2;
            break;
        }
        if (std::fabs(gtd_new) <= -(float)c2 * gtd) {
            br[0] = t; br_f[0] = f_new; br_gtd[0] = gtd_new; nbr = 1; done = true;
            break;
        }
        if (gtd_new >= 0) {
            br[0] = t_prev; br[1] = t; br_f[0] = f_prev; br_f[1] = f_new; br_gtd[0] = gtd_prev; br_gtd[1] = gtd_new; nbr = 2;
            break;
        }
        const double min_step = t + 0.01 * (t - t_prev);
        const double max_step = t * 10;
        const double tmp = t;
        t = cubic_interpolate(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, true, min_step, max_step);
        t_prev = tmp; f_prev = f_new; gtd_prev = gtd_new;
        OCHK(eval_at(t, &f_new, &gtd_new));
        evals += 1;
        ls_iter += 1;
    }
    if (ls_iter == max_ls && nbr == 0) {
        br[0] = 0; br[1] = t; br_f[0] = f; br_f[1] = f_new; br_gtd[0] = gtd; br_gtd[1] = gtd_new; nbr = 2;
    }
    bool insuf = false;
    int low = (br_f[0] <= br_f[nbr - 1]) ? 0 : 1, high = 1 - low;
    if (nbr == 1) { low = 0; high = 0; }
    while (!done && ls_iter < max_ls) {
        if (std::fabs(br[1] - br[0]) * d_norm < tol_change) break;
        t = cubic_interpolate(br[0], br_f[0], br_gtd[0], br[1], br_f[1], br_gtd[1], false, 0, 0);
        const double bmax = std::max(br[0], br[1]), bmin = std::min(br[0], br[1]);
        const double eps = 0.1 * (bmax - bmin);
        if (std::min(bmax - t, t - bmin) < eps) {
            if (insuf || t >= bmax || t <= bmin) {
                t = (std::fabs(t - bmax) < std::fabs(t - bmin)) ? bmax - eps : bmin + eps;
                insuf = false;
            } else {
                insuf = true;
            }
        } else {
            insuf = false;
        }
        OCHK(eval_at(t, &f_new, &gtd_new));
        evals += 1;
        ls_iter += 1;
        if (f_new > (float)(f + (float)(c1 * t) * gtd) || f_new >= br_f[low]) {
            br[high] = t; br_f[high] = f_new; br_gtd[high] = gtd_new;
            low = (br_f[0] <= br_f[1]) ? 0 : 1; high = 1 - low;
        } else {
            if (std::fabs(gtd_new) <= -(float)c2 * gtd) {
                done = true;
            } else if (gtd_new * (br[high] - br[low]) >= 0) {
                br[high] = br[low]; br_f[high] = br_f[low]; br_gtd[high] = br_gtd[low];
            }
            br[low] = t; br_f[low] = f_new; br_gtd[low] = gtd_new;
        }
    }
    res->t = br[low]; res->f = br_f[low]; res->evals = evals;
    return NST_OK;
}

// the two recurrences of the inner-product form below on host scalars: coef[j] multiplies y_j, coef[m + j] multiplies s_j
void direction_coefficients(int m, int ld, const double* SYm, const double* YYm, const float* ro, float Hd,
                            const double* sq, const double* yq, float* coef) {
    std::vector<double> al(m), be(m);
    for (int i = m - 1; i >= 0; --i) {
        double v = sq[i];                                             // s_i . q0
        for (int j = i + 1; j < m; ++j) v -= al[j] * SYm[(size_t)i * ld + j];
        al[i] = (double)ro[i] * v;
    }
    for (int i = 0; i < m; ++i) {
        double t = yq[i];                                             // y_i . q0
        for (int j = 0; j < m; ++j) t -= al[j] * YYm[(size_t)j * ld + i];
        double v = (double)Hd * t;
        for (int j = 0; j < i; ++j) v += (al[j] - be[j]) * SYm[(size_t)j * ld + i];
        be[i] = (double)ro[i] * v;
    }
    for (int j = 0; j < m; ++j) { coef[j] = (float)(-(double)Hd * al[j]); coef[m + j] = (float)(al[j] - be[j]); }
}

// d = -H g by the two-loop recursion (lbfgs.py:444-460) carried out on inner products: with q0 = -g,
//   al_i = ro_i (s_i.q0 - sum_{j>i} al_j s_i.y_j)                                   (first loop, i = m-1 .. 0)
//   be_i = ro_i (H (y_i.q0 - sum_j al_j y_j.y_i) + sum_{j<i} (al_j - be_j) s_j.y_i)  (second loop, i = 0 .. m-1)
//   d = H q0 - H sum_j al_j y_j + sum_j (al_j - be_j) s_j
// One multi-dot pass gives s_i.q0, y_i.q0 and, for a new pair k, the new row / column of SY and YY.
int gram_direction(nst_opt* o, bool new_pair, hipStream_t s) {
    const int m = (int)o->old_dirs.size();
    const int Hh = o->history;
    const float Hd = o->H_is_one ? 1.f : o->H_diag;
    OHIP(o, launch_scale_copy(-1.f, o->g, o->q, o->n, s));       // q0 = -g
    if (m == 0) {
        OHIP(o, launch_scale_copy(Hd, o->q, o->d, o->n, s));
        return NST_OK;
    }
    const float** hp = reinterpret_cast<const float**>(o->pin2);
    float* hcoef = reinterpret_cast<float*>(o->pin2 + 2 * (size_t)Hh * sizeof(float*));
    float* hres = hcoef + 2 * (size_t)Hh;
    for (int j = 0; j < m; ++j) { hp[j] = o->old_dirs[j]; hp[m + j] = o->old_stps[j]; }
    OHIP(o, hipMemcpyAsync(o->vec_dev, hp, 2 * (size_t)m * sizeof(float*), hipMemcpyHostToDevice, s));
    const float* a = new_pair ? o->old_stps[m - 1] : o->q;
    const float* b = new_pair ? o->old_dirs[m - 1] : o->q;
    OHIP(o, launch_multi_dot(o->vec_dev, 2 * m, a, b, o->q, o->n, o->md_scratch, o->md_out, s));
    OHIP(o, hipMemcpyAsync(hres, o->md_out, 2 * (size_t)m * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    OHIP(o, hipStreamSynchronize(s));
    auto SY = [&](int i, int j) -> double& { return o->SY[(size_t)i * Hh + j]; };
    auto YY = [&](int i, int j) -> double& { return o->YY[(size_t)i * Hh + j]; };
    if (new_pair) {
        const int k = m - 1;
        for (int j = 0; j < m; ++j) {
            SY(k, j) = hres[(size_t)j * 3 + 0];                       // s_k . y_j
            YY(k, j) = YY(j, k) = hres[(size_t)j * 3 + 1];            // y_k . y_j
            SY(j, k) = hres[(size_t)(m + j) * 3 + 1];                 // s_j . y_k
        }
    }
    std::vector<double> sq(m), yq(m);
    for (int i = 0; i < m; ++i) { sq[i] = hres[(size_t)(m + i) * 3 + 2]; yq[i] = hres[(size_t)i * 3 + 2]; }
    direction_coefficients(m, Hh, o->SY.data(), o->YY.data(), o->ro.data(), Hd, sq.data(), yq.data(), hcoef);
    OHIP(o, hipMemcpyAsync(o->coef_dev, hcoef, 2 * (size_t)m * sizeof(float), hipMemcpyHostToDevice, s));
    OHIP(o, launch_multi_axpy(o->vec_dev, o->coef_dev, 2 * m, o->q, Hd, o->d, o->n, s));
    return NST_OK;
}

int lbfgs_step(nst_opt* o, float* x, float cw, float sw, float tvw, hipStream_t s, nst_step_info* info) {
    const double lr = o->lr;                                     // read before the closure decays it (lbfgs.py:349)
    float loss;
    GradStats g0;
    g0.want_abs = true;
    OCHK(eval_closure(o, x, cw, sw, tvw, s, &loss, &g0));
    info->loss = loss;
    const float gmax = g0.gmax, gsum = g0.gsum;
    if (gmax <= 1e-7f) { info->accepted = 0; info->t = 0.f; return NST_OK; }
    o->n_iter += 1;
    if (o->n_iter == 1) {
        OHIP(o, launch_scale_copy(-1.f, o->g, o->d, o->n, s));   // d = -g
        for (float* p : o->old_dirs) o->spare.push_back(p);
        for (float* p : o->old_stps) o->spare.push_back(p);
        o->old_dirs.clear(); o->old_stps.clear(); o->ro.clear();
        o->H_is_one = true; o->H_diag = 1.f;
        if (o->gram_mode) { std::fill(o->SY.begin(), o->SY.end(), 0.0); std::fill(o->YY.begin(), o->YY.end(), 0.0); }
    } else {
        // y = g - prev_g ; s = d * t
        float* y; float* st;
        auto take = [&](float** p) -> int {
            if (o->spare.empty()) return nst_internal_fail(o->ctx, NST_E_STATE, "L-BFGS history pool exhausted");
            *p = o->spare.back(); o->spare.pop_back();
            return NST_OK;
        };
        OCHK(take(&y)); OCHK(take(&st));
        OHIP(o, launch_sub(o->g, o->prev_g, y, o->n, s));
        OHIP(o, launch_scale_copy((float)o->t, o->d, st, o->n, s));
        float ys, yy;
        OCHK(dot2(o, y, st, y, y, s, &ys, &yy));
        bool new_pair = false;
        if (ys > 1e-10f) {
            if ((int)o->old_dirs.size() == o->history) {
                o->spare.push_back(o->old_dirs.front()); o->spare.push_back(o->old_stps.front());
                o->old_dirs.erase(o->old_dirs.begin()); o->old_stps.erase(o->old_stps.begin()); o->ro.erase(o->ro.begin());
                if (o->gram_mode) {
                    // the oldest pair leaves: rows / columns move up-left
                    const int Hh = o->history;
                    for (int i = 0; i + 1 < Hh; ++i)
                        for (int j = 0; j + 1 < Hh; ++j) {
                            o->SY[(size_t)i * Hh + j] = o->SY[(size_t)(i + 1) * Hh + j + 1];
                            o->YY[(size_t)i * Hh + j] = o->YY[(size_t)(i + 1) * Hh + j + 1];
                        }
                }
            }
            o->old_dirs.push_back(y); o->old_stps.push_back(st); o->ro.push_back(1.0f / ys);
            o->H_diag = ys / yy; o->H_is_one = false;
            new_pair = true;
        } else {
            o->spare.push_back(y); o->spare.push_back(st);
        }
        if (o->gram_mode) {
            OCHK(gram_direction(o, new_pair, s));
        } else {
        // two-loop recursion (lbfgs.py:444-460) without a host round trip per pair: every launch finishes the previous
        // launch's dot product, applies its pair's update and leaves the partials of the next pair's dot product
        // (vector_ops.hip::lbfgs_pair_kernel).  The two halves of `scratch` alternate as the partials buffer.
        const int num_old = (int)o->old_dirs.size();
        double* pp[2] = {o->scratch, o->scratch + RED_BLOCKS};
        int cur = 0;
        OHIP(o, launch_scale_copy(-1.f, o->g, o->q, o->n, s));   // q = -g
        if (num_old > 0) OHIP(o, launch_dot_partial(o->old_stps[num_old - 1], o->q, o->n, pp[cur], s));
        for (int i = num_old - 1; i >= 0; --i) {
            // al_i = (s_i . q) ro_i ;  q -= al_i y_i ;  partials of s_{i-1} . q
            OHIP(o, launch_lbfgs_pair(pp[cur], o->ro[i], o->al_dev + i, 0, o->old_dirs[i], o->q,
                                      i > 0 ? o->old_stps[i - 1] : nullptr, o->n, pp[cur ^ 1], s));
            cur ^= 1;
        }
        OHIP(o, launch_scale_copy(o->H_is_one ? 1.f : o->H_diag, o->q, o->d, o->n, s));   // d = r = q * H_diag
        if (num_old > 0) OHIP(o, launch_dot_partial(o->old_dirs[0], o->d, o->n, pp[cur], s));
        for (int i = 0; i < num_old; ++i) {
            // be_i = (y_i . r) ro_i ;  r += (al_i - be_i) s_i ;  partials of y_{i+1} . r
            OHIP(o, launch_lbfgs_pair(pp[cur], o->ro[i], o->al_dev + i, 1, o->old_stps[i], o->d,
                                      i + 1 < num_old ? o->old_dirs[i + 1] : nullptr, o->n, pp[cur ^ 1], s));
            cur ^= 1;
        }
        }
    }
    OHIP(o, launch_copy(o->g, o->prev_g, o->n, s));
    o->have_prev = true;
    double t;
    if (o->n_iter == 1) {
        const float inv = 1.0f / gsum;
        t = (inv < 1.0f) ? (double)(inv * (float)lr) : lr;       // min(1., 1./|g|_1) * lr
    } else {
        t = lr;
    }
    float gtd, d_norm, d_sum;
    OCHK(dot_and_absstats(o, o->g, o->d, s, &gtd, &d_norm, &d_sum));
    info->accepted = 0; info->t = 0.f;
    if (!(gtd > -1e-9f)) {
        OHIP(o, launch_copy(x, o->xinit, o->n, s));
        LsResult r;
        OCHK(strong_wolfe(o, x, t, loss, gtd, d_norm, o->max_eval - 1, cw, sw, tvw, s, &r));
        t = r.t;
        if (t != 0.0) OHIP(o, launch_add_scaled(o->xinit, (float)t, o->d, x, o->n, s));
        else OHIP(o, launch_copy(o->xinit, x, o->n, s));
        info->accepted = (t != 0.0) ? 1 : 0;
        info->t = (float)t;
    }
    o->t = t;
    return NST_OK;
}

}  // namespace

extern "C" {

int nst_opt_create(nst_ctx* ctx, int kind, float lr_start, int lbfgs_max_eval, nst_opt** out) {
    if (!ctx || !out) return nst_internal_fail(ctx, NST_E_ARG, "null argument");
    if (kind != NST_OPT_ADAM && kind != NST_OPT_LBFGS) return nst_internal_fail(ctx, NST_E_ARG, "Unknown optimizer");
    if (nst_internal_levels(ctx) < 1) return nst_internal_fail(ctx, NST_E_STATE, "nst_job_configure has not been called");
    if (hipSetDevice(nst_internal_device(ctx)) != hipSuccess) return nst_internal_fail(ctx, NST_E_HIP, "hipSetDevice failed");
    nst_opt* o = new (std::nothrow) nst_opt();
    if (!o) return nst_internal_fail(ctx, NST_E_NOMEM, "out of host memory");
    o->ctx = ctx; o->kind = kind; o->lr = (double)lr_start;
    o->levels = nst_internal_levels(ctx);
    o->n = 3 * nst_internal_pixels(ctx);
    o->max_eval = lbfgs_max_eval < 1 ? 1 : lbfgs_max_eval;
    const size_t row = (size_t)NST_LOSS_ROW * o->levels + 1;
    // gradient and loss row live in ONE allocation (gradient padded to 64 floats): the sharded closure all-reduces both
    // with a single collective (nst_opt_shard_levels_comm)
    const size_t n_pad = (o->n + 63) & ~(size_t)63;
    o->pack_floats = n_pad + row;
    int r = oalloc(o, &o->pack, o->pack_floats);
    if (r == NST_OK && nst_internal_zero_now(o->pack, o->pack_floats * sizeof(float))) r = NST_E_HIP;
    o->g = o->own_g = o->pack;
    o->losses = o->own_losses = o->pack ? o->pack + n_pad : nullptr;
    if (r == NST_OK) r = oalloc(o, &o->scal, 4);
    if (r == NST_OK && hipEventCreateWithFlags(&o->tail, hipEventDisableTiming) != hipSuccess) r = NST_E_HIP;
    if (r == NST_OK && kind == NST_OPT_LBFGS) r = oalloc(o, &o->al_dev, (size_t)o->history);
    // nst_options.lbfgs_gram (env NST_LBFGS_GRAM as the default): 0 = the sequential recursion, one fused launch per pair and loop
    if (r == NST_OK && kind == NST_OPT_LBFGS && nst_internal_lbfgs_gram(ctx)) {
        const size_t H2 = 2 * (size_t)o->history;
        o->gram_mode = true;
        o->SY.assign((size_t)o->history * o->history, 0.0);
        o->YY.assign((size_t)o->history * o->history, 0.0);
        if (hipMalloc(reinterpret_cast<void**>(&o->vec_dev), H2 * sizeof(float*)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&o->md_scratch), (size_t)multi_dot_blocks(o->n) * H2 * 3 * sizeof(double)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&o->pin2), H2 * (sizeof(float*) + 4 * sizeof(float)), hipHostMallocDefault) != hipSuccess)
            r = NST_E_NOMEM;
        if (r == NST_OK) r = oalloc(o, &o->coef_dev, H2);
        if (r == NST_OK) r = oalloc(o, &o->md_out, H2 * 3);
    }
    if (r == NST_OK && hipHostMalloc(reinterpret_cast<void**>(&o->pinned), (8 + (size_t)NST_LOSS_ROW * NST_MAX_LEVELS + 1) * sizeof(float), hipHostMallocDefault) != hipSuccess) r = NST_E_NOMEM;
    if (r == NST_OK && hipMalloc(reinterpret_cast<void**>(&o->scratch), 2 * RED_BLOCKS * sizeof(double)) != hipSuccess) r = NST_E_NOMEM;
    if (r == NST_OK && kind == NST_OPT_ADAM) {
        r = oalloc(o, &o->m, o->n);
        if (r == NST_OK) r = oalloc(o, &o->v, o->n);
        if (r == NST_OK && (nst_internal_zero_now(o->m, o->n * 4) || nst_internal_zero_now(o->v, o->n * 4))) r = NST_E_HIP;
    }
    if (r == NST_OK && kind == NST_OPT_LBFGS) {
        r = oalloc(o, &o->d, o->n);
        if (r == NST_OK) r = oalloc(o, &o->prev_g, o->n);
        if (r == NST_OK) r = oalloc(o, &o->xinit, o->n);
        if (r == NST_OK) r = oalloc(o, &o->q, o->n);
        // the whole curvature history up front: history pairs (y, s) + the pair being formed.  3.8 GB at L=2, 15 GB at
        // L=3 of the 288 GB; a job that cannot have it fails here, not in the middle of its run
        const size_t nvec = 2 * (size_t)o->history + 2;
        if (r == NST_OK) r = oalloc(o, &o->pool, nvec * n_pad);
        if (r == NST_OK) {
            o->spare.reserve(nvec);
            for (size_t i = nvec; i-- > 0;) o->spare.push_back(o->pool + i * n_pad);
        }
    }
    if (r != NST_OK) { nst_opt_destroy(o); return nst_internal_fail(ctx, r, "optimiser allocation failed"); }
    *out = o;
    return NST_OK;
}

void nst_opt_destroy(nst_opt* o) {
    if (!o) return;
    (void)hipSetDevice(nst_internal_device(o->ctx));
    // wait for THIS optimiser's last launches only (a device-wide synchronisation would stall the other job that
    // shares the GPU in the two-jobs-per-GPU serving mode)
    if (o->tail) { (void)hipEventSynchronize(o->tail); (void)hipEventDestroy(o->tail); }
    float* ptrs[] = {o->pack, o->scal, o->al_dev, o->m, o->v, o->d, o->prev_g, o->xinit, o->q, o->pool};
    for (float* p : ptrs) if (p) (void)hipFree(p);
    if (o->scratch) (void)hipFree(o->scratch);
    if (o->pinned) (void)hipHostFree(o->pinned);
    if (o->pin2) (void)hipHostFree(o->pin2);
    if (o->vec_dev) (void)hipFree(o->vec_dev);
    if (o->md_scratch) (void)hipFree(o->md_scratch);
    if (o->coef_dev) (void)hipFree(o->coef_dev);
    if (o->md_out) (void)hipFree(o->md_out);
    delete o;
}

int nst_opt_shard_levels(nst_opt* o, unsigned level_mask, float* grad, float* losses, nst_reduce_hook hook, void* user) {
    if (!o) return nst_internal_fail(nullptr, NST_E_ARG, "null optimiser");
    if ((hook != nullptr) != (grad != nullptr && losses != nullptr))
        return nst_internal_fail(o->ctx, NST_E_ARG, "a reduce hook needs caller-owned grad and losses buffers (and vice versa)");
    o->level_mask = level_mask;
    o->hook = hook; o->hook_user = user;
    o->comm = nullptr;
    o->g = grad ? grad : o->own_g;
    o->losses = losses ? losses : o->own_losses;
    return NST_OK;
}

int nst_opt_shard_levels_comm(nst_opt* o, unsigned level_mask, nst_comm* comm) {
    if (!o) return nst_internal_fail(nullptr, NST_E_ARG, "null optimiser");
    o->level_mask = comm ? level_mask : 0xFFFFFFFFu;
    o->hook = nullptr; o->hook_user = nullptr;
    o->comm = comm;
    o->g = o->own_g; o->losses = o->own_losses;
    return NST_OK;
}

int nst_opt_history(const nst_opt* o, int* pairs, int* n_iter) {
    if (!o) return nst_internal_fail(nullptr, NST_E_ARG, "null optimiser");
    if (pairs) *pairs = (int)o->old_dirs.size();
    if (n_iter) *n_iter = o->kind == NST_OPT_ADAM ? o->k : o->n_iter;
    return NST_OK;
}

// torch:optim/adam.py:457-546 for one tensor, step count k (1-based), group lr `lr` (a python double in the reference)
static int adam_update(nst_ctx* ctx, float* x, const float* g, float* m, float* v, size_t n, int k, double lr, hipStream_t s,
                       float* step_size_out) {
    const double bc1 = 1.0 - std::pow(0.9, k);
    const double bc2 = 1.0 - std::pow(0.999, k);
    const double step_size = lr / bc1;
    // 1 - beta as torch forms them: double differences, rounded to fp32 when they meet the fp32 tensors
    const hipError_t e = launch_adam(x, g, m, v, n, 0.999f, (float)(1.0 - 0.9), (float)(1.0 - 0.999), 1e-8f, (float)step_size,
                                     (float)std::sqrt(bc2), s);
    if (e != hipSuccess) return nst_internal_fail(ctx, NST_E_HIP, hipGetErrorString(e));
    if (step_size_out) *step_size_out = (float)step_size;
    return NST_OK;
}

int nst_adam_step(nst_ctx* ctx, float* x, const float* g, float* m, float* v, size_t n, int k, double lr, void* stream) {
    if (!ctx || !x || !g || !m || !v || k < 1) return nst_internal_fail(ctx, NST_E_ARG, "bad argument");
    if (hipSetDevice(nst_internal_device(ctx)) != hipSuccess) return nst_internal_fail(ctx, NST_E_HIP, "hipSetDevice failed");
    return adam_update(ctx, x, g, m, v, n, k, lr, static_cast<hipStream_t>(stream), nullptr);
}

int nst_lbfgs_direction(nst_ctx* ctx, const float* g, const float* const* y, const float* const* sv, const float* ro, int m,
                        float h_diag, size_t n, int form, float* d, void* stream) {
    if (!ctx || !g || !d || m < 0 || (m > 0 && (!y || !sv || !ro)) || (form != 0 && form != 1))
        return nst_internal_fail(ctx, NST_E_ARG, "bad argument");
    if (hipSetDevice(nst_internal_device(ctx)) != hipSuccess) return nst_internal_fail(ctx, NST_E_HIP, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(stream);
    nst_opt tmp;                       // only for the error macros
    tmp.ctx = ctx;
    nst_opt* o = &tmp;
    float* q = nullptr; float* al_dev = nullptr; float* coef_dev = nullptr; float* md_out = nullptr;
    double* scratch = nullptr; double* md_scratch = nullptr; const float** vec_dev = nullptr;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        for (void* p : {(void*)q, (void*)al_dev, (void*)coef_dev, (void*)md_out, (void*)scratch, (void*)md_scratch, (void*)vec_dev})
            if (p) (void)hipFree(p);
    };
    auto body = [&]() -> int {
        OHIP(o, hipMalloc(reinterpret_cast<void**>(&q), n * sizeof(float)));
        OHIP(o, launch_scale_copy(-1.f, g, q, n, s));                               // q0 = -g
        if (m == 0) { OHIP(o, launch_scale_copy(h_diag, q, d, n, s)); return NST_OK; }
        if (form == 1) {
            // sequential form: the arithmetic order of torch's two loops, one fused launch per pair and loop
            OHIP(o, hipMalloc(reinterpret_cast<void**>(&al_dev), (size_t)m * sizeof(float)));
            OHIP(o, hipMalloc(reinterpret_cast<void**>(&scratch), 2 * RED_BLOCKS * sizeof(double)));
            double* pp[2] = {scratch, scratch + RED_BLOCKS};
            int cur = 0;
            OHIP(o, launch_dot_partial(sv[m - 1], q, n, pp[cur], s));
            for (int i = m - 1; i >= 0; --i) {
                OHIP(o, launch_lbfgs_pair(pp[cur], ro[i], al_dev + i, 0, y[i], q, i > 0 ? sv[i - 1] : nullptr, n, pp[cur ^ 1], s));
                cur ^= 1;
            }
            OHIP(o, launch_scale_copy(h_diag, q, d, n, s));
            OHIP(o, launch_dot_partial(y[0], d, n, pp[cur], s));
            for (int i = 0; i < m; ++i) {
                OHIP(o, launch_lbfgs_pair(pp[cur], ro[i], al_dev + i, 1, sv[i], d, i + 1 < m ? y[i + 1] : nullptr, n, pp[cur ^ 1], s));
                cur ^= 1;
            }
            return NST_OK;
        }
        // inner-product form: S^T Y and Y^T Y by m multi-dot passes (the optimiser driver keeps them incrementally, one
        // pass per step), then the same recurrences and the same multi-axpy pass as the driver
        const size_t M2 = 2 * (size_t)m;
        OHIP(o, hipMalloc(reinterpret_cast<void**>(&vec_dev), M2 * sizeof(float*)));
        OHIP(o, hipMalloc(reinterpret_cast<void**>(&md_scratch), (size_t)multi_dot_blocks(n) * M2 * 3 * sizeof(double)));
        OHIP(o, hipMalloc(reinterpret_cast<void**>(&md_out), M2 * 3 * sizeof(float)));
        OHIP(o, hipMalloc(reinterpret_cast<void**>(&coef_dev), M2 * sizeof(float)));
        std::vector<const float*> hp(M2);
        for (int j = 0; j < m; ++j) { hp[j] = y[j]; hp[m + j] = sv[j]; }
        OHIP(o, hipMemcpyAsync(vec_dev, hp.data(), M2 * sizeof(float*), hipMemcpyHostToDevice, s));
        std::vector<double> SY((size_t)m * m), YY((size_t)m * m), sq(m), yq(m);
        std::vector<float> res(M2 * 3), coef(M2);
        for (int k = 0; k < m; ++k) {
            OHIP(o, launch_multi_dot(vec_dev, (int)M2, sv[k], y[k], q, n, md_scratch, md_out, s));
            OHIP(o, hipMemcpyAsync(res.data(), md_out, M2 * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
            OHIP(o, hipStreamSynchronize(s));
            for (int j = 0; j < m; ++j) {
                SY[(size_t)k * m + j] = res[(size_t)j * 3 + 0];            // s_k . y_j
                YY[(size_t)k * m + j] = res[(size_t)j * 3 + 1];            // y_k . y_j
            }
            if (k == 0) for (int j = 0; j < m; ++j) { yq[j] = res[(size_t)j * 3 + 2]; sq[j] = res[(size_t)(m + j) * 3 + 2]; }
        }
        direction_coefficients(m, m, SY.data(), YY.data(), ro, h_diag, sq.data(), yq.data(), coef.data());
        OHIP(o, hipMemcpyAsync(coef_dev, coef.data(), M2 * sizeof(float), hipMemcpyHostToDevice, s));
        OHIP(o, launch_multi_axpy(vec_dev, coef_dev, (int)M2, q, h_diag, d, n, s));
        OHIP(o, hipStreamSynchronize(s));      // `coef` is pageable host memory: keep it alive until the copy has run
        return NST_OK;
    };
    const int rc = body();
    cleanup();
    return rc;
}

static int opt_step_body(nst_opt* o, float* x, float cw, float sw, float tvw, nst_step_info* info, hipStream_t s) {
    if (o->kind == NST_OPT_ADAM) {
        float loss = NAN;
        if (o->want_rows) {
            OCHK(eval_closure(o, x, cw, sw, tvw, s, &loss));
        } else {
            o->lr *= 0.999;
            OCHK(nst_closure_levels(o->ctx, x, cw, sw, tvw, o->level_mask, o->g, o->losses, s));
            if (o->comm) OCHK(nst_comm_allreduce_sum(o->comm, o->pack, o->pack_floats, s));
            else if (o->hook) o->hook(o->hook_user);
            o->total_closures += 1;
        }
        o->k += 1;
        float step_size = 0.f;
        OCHK(adam_update(o->ctx, x, o->g, o->m, o->v, o->n, o->k, o->lr, s, &step_size));   // lr already decayed by the closure
        info->loss = loss; info->accepted = 1; info->t = step_size;
    } else {
        OCHK(lbfgs_step(o, x, cw, sw, tvw, s, info));
    }
    return NST_OK;
}

int nst_opt_step(nst_opt* o, float* x, float cw, float sw, float tvw, float* losses_host, int closures_capacity,
                 nst_step_info* info, void* stream) {
    if (!o || !x || !info) return nst_internal_fail(o ? o->ctx : nullptr, NST_E_ARG, "null argument");
    if (hipSetDevice(nst_internal_device(o->ctx)) != hipSuccess) return nst_internal_fail(o->ctx, NST_E_HIP, "hipSetDevice failed");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a step issued on another stream than the previous one is ordered behind it (one tail event then covers the optimiser)
    if (o->tail_set && s != o->tail_stream) OHIP(o, hipStreamWaitEvent(s, o->tail, 0));
    std::memset(info, 0, sizeof(*info));
    o->loss_rows.clear();
    const int before = o->total_closures;
    o->want_rows = losses_host != nullptr;
    const int rc = opt_step_body(o, x, cw, sw, tvw, info, s);
    // recorded on EVERY path out of the step - a step that failed half way has launches in flight too - so that
    // nst_opt_destroy never frees the optimiser's buffers under them
    if (hipEventRecord(o->tail, s) == hipSuccess) { o->tail_stream = s; o->tail_set = true; }
    if (rc != NST_OK) return rc;
    info->closures = o->total_closures - before;
    info->total_closures = o->total_closures;
    info->lr = (float)o->lr;
    info->history = (int)o->old_dirs.size();
    if (losses_host) {
        const size_t row = (size_t)NST_LOSS_ROW * o->levels + 1;
        const size_t have = o->loss_rows.size() / row;
        const size_t cp = std::min(have, (size_t)std::max(closures_capacity, 0));
        std::memcpy(losses_host, o->loss_rows.data(), cp * row * sizeof(float));
    }
    return NST_OK;
}

}  // extern "C"
