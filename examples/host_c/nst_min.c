/* Minimal C host of the MI355X style-transfer engine: no Python, no torch - only include/nst_hip.h,
 * libnst_hip.so and the HIP runtime for the caller-owned device buffers.  It is the call sequence a
 * non-Python front-end binds: context -> job geometry -> targets ->
 * optimiser -> steps, on a small synthetic job (seeded pseudo-random VGG19 weights and images), and it
 * prints the loss of every optimiser step.
 *
 *   gcc -std=c99 -O2 -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/host_c/nst_min.c \
 *       -Lartstyletransfer_amd -lnst_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/artstyletransfer_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/nst_min
 *   /tmp/nst_min [adam|lbfgs] [steps]
 *
 * Level sharding over several GPUs without Python (BASELINE config 4): start one process per GPU,
 *   /tmp/nst_min lbfgs 6 <rank> <world> <id-file>
 * Rank 0 writes the communicator id (nst_comm_unique_id) to <id-file>, the others read it; every rank then evaluates the
 * pyramid levels dealt to it (largest first onto the least-loaded rank) and the driver all-reduces the packed gradient + loss row over RCCL per closure.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime_api.h>

#include "nst_hip.h"

static const int kCin[NST_VGG19_CONVS] = {3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512};
static const int kCout[NST_VGG19_CONVS] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512};

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float uniform01(void) { /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (float)((rng_state * 0x2545F4914F6CDD1Dull) >> 40) / 16777216.0f;
}
static float gaussian(void) {
    float u1 = uniform01(), u2 = uniform01();
    if (u1 < 1e-7f) u1 = 1e-7f;
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
}

#define CHECK_NST(call)                                                                          \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_ != NST_OK) {                                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nst_last_error(ctx));                  \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)
#define CHECK_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)

/* smooth synthetic HWC image in [0,1], prepared as the reference's prepare_img does: x*255 - mean, CHW */
static float* prepared_image(nst_ctx* ctx, int h, int w, float phase) {
    static const float mean[3] = {123.675f, 116.28f, 103.53f};
    size_t n = (size_t)3 * h * w;
    float* host = (float*)malloc(n * sizeof(float));
    float* dev = NULL;
    (void)ctx;
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float v = 0.5f + 0.25f * sinf(0.11f * x + phase + c) * cosf(0.07f * y - phase) + 0.2f * (uniform01() - 0.5f);
                host[((size_t)c * h + y) * w + x] = v * 255.0f - mean[c];
            }
    if (hipMalloc((void**)&dev, n * sizeof(float)) != hipSuccess) { free(host); return NULL; }
    if (hipMemcpy(dev, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { free(host); return NULL; }
    free(host);
    return dev;
}

/* the NST_COMM_ID_BYTES of rank 0's nst_comm_unique_id, handed over through a file (any channel would do) */
static int exchange_id(const char* path, int rank, unsigned char* id) {
    if (rank == 0) {
        char tmp[1024];
        FILE* f;
        if (nst_comm_unique_id(id) != NST_OK) return 1;
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, NST_COMM_ID_BYTES, f) != NST_COMM_ID_BYTES) return 1;
        fclose(f);
        return rename(tmp, path) != 0;            /* atomic: readers never see a partial file */
    }
    for (int tries = 0; tries < 6000; ++tries) {
        FILE* f = fopen(path, "rb");
        if (f) {
            size_t n = fread(id, 1, NST_COMM_ID_BYTES, f);
            fclose(f);
            if (n == NST_COMM_ID_BYTES) return 0;
        }
        struct timespec ts = {0, 10 * 1000 * 1000};
        nanosleep(&ts, NULL);
    }
    return 1;
}

int main(int argc, char** argv) {
    const int kind = (argc > 1 && strcmp(argv[1], "adam") == 0) ? NST_OPT_ADAM : NST_OPT_LBFGS;
    const int steps = argc > 2 ? atoi(argv[2]) : 6;
    const int rank = argc > 5 ? atoi(argv[3]) : 0, world = argc > 5 ? atoi(argv[4]) : 1;
    const int H = 96, W = 144, levels = 2;
    nst_ctx* ctx = NULL;
    nst_opt* opt = NULL;
    nst_comm* comm = NULL;
    int ndev = 0;

    if (nst_device_count(&ndev) != NST_OK || ndev < 1) { fprintf(stderr, "no GPU: the engine has no CPU path\n"); return 2; }

    /* seeded synthetic weights: N(0, 2 / (Cout * 9)), zero biases (torchvision order conv1_1..conv5_1) */
    float* w[NST_VGG19_CONVS];
    float* b[NST_VGG19_CONVS];
    for (int l = 0; l < NST_VGG19_CONVS; ++l) {
        size_t nw = (size_t)kCout[l] * kCin[l] * 9;
        float sd = sqrtf(2.0f / (kCout[l] * 9.0f));
        w[l] = (float*)malloc(nw * sizeof(float));
        b[l] = (float*)calloc((size_t)kCout[l], sizeof(float));
        for (size_t i = 0; i < nw; ++i) w[l][i] = sd * gaussian();
    }
    /* options as arguments (fields left at -1 fall back to the environment, then to the defaults) */
    nst_options opts;
    nst_options_default(&opts);
    opts.conv_mode = NST_CONV_F16X2;
    CHECK_NST(nst_ctx_create_ex(world > 1 ? rank % ndev : 0, (const float* const*)w, (const float* const*)b, &opts, &ctx));
    for (int l = 0; l < NST_VGG19_CONVS; ++l) { free(w[l]); free(b[l]); }

    CHECK_NST(nst_job_configure(ctx, levels, H, W));
    for (int l = 0; l < levels; ++l) {
        int h = H >> l, ww = W >> l;
        float* content = prepared_image(ctx, h, ww, 0.3f);
        float* style = prepared_image(ctx, h + 10, ww - 6, 1.7f);       /* a style image of its own size */
        if (!content || !style) { fprintf(stderr, "device allocation failed\n"); return 1; }
        CHECK_NST(nst_level_set_targets(ctx, l, content, style, h + 10, ww - 6, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipFree(content));
        CHECK_HIP(hipFree(style));
    }
    float* x = prepared_image(ctx, H, W, 0.9f);                          /* the optimised pixels: caller-owned */
    if (!x) { fprintf(stderr, "device allocation failed\n"); return 1; }

    CHECK_NST(nst_opt_create(ctx, kind, kind == NST_OPT_ADAM ? 10.0f : 1.0f, 26, &opt));
    if (world > 1) {
        unsigned char id[NST_COMM_ID_BYTES];
        unsigned mask = 0;
        if (exchange_id(argv[5], rank, id)) { fprintf(stderr, "communicator id exchange failed: %s\n", nst_last_error(NULL)); return 1; }
        CHECK_NST(nst_comm_create(rank % ndev, rank, world, id, &comm));
        {   /* levels dealt largest first onto the least-loaded rank (work of level l = 4^-l): the rule of
             * artstyletransfer_amd/sharding.py deal_levels */
            double load[64] = {0};
            double wl = 1.0;
            for (int l = 0; l < levels; ++l, wl *= 0.25) {
                int r = 0;
                for (int i = 1; i < world && i < 64; ++i) if (load[i] < load[r]) r = i;
                load[r] += wl;
                if (r == rank) mask |= 1u << l;
            }
        }
        CHECK_NST(nst_opt_shard_levels_comm(opt, mask, comm));
    }
    float first = 0.f, last = 0.f;
    for (int s = 0; s < steps; ++s) {
        nst_step_info info;
        float rows[26 * (NST_LOSS_ROW * 2 + 1)];
        CHECK_NST(nst_opt_step(opt, x, 1e3f, 4e5f, 1e2f, rows, 26, &info, NULL));
        CHECK_HIP(hipDeviceSynchronize());
        printf("step %d: loss %.6e closures %d (total %d) accepted %d lr %.4f\n", s, info.loss, info.closures,
               info.total_closures, info.accepted, info.lr);
        if (s == 0) first = info.loss;
        last = info.loss;
        if (!(info.loss == info.loss) || info.loss > 3.0e38f) { fprintf(stderr, "non-finite loss\n"); return 1; }
    }
    printf("first %.6e last %.6e %s\n", first, last, last < first ? "DECREASED" : "NOT-DECREASED");
    nst_opt_destroy(opt);
    nst_comm_destroy(comm);
    nst_ctx_destroy(ctx);
    CHECK_HIP(hipFree(x));
    return last < first ? 0 : 3;
}
