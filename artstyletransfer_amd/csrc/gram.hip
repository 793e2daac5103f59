// gram.hip - Gram matrix G = F F^T / (C h w) of an NHWC feature map (math_utils.py:26-34) on the
// gfx950 fp32 matrix cores, split over the pixel dimension with an ordered two-stage reduction
// (bitwise reproducible; no float atomics).
//
// With pixel-major activations A[n][c] the Gram is A^T A: both MFMA operands are rows of the same
// LDS tile and lane l of v_mfma_f32_32x32x2_f32 wants A[k = l>>5][c = l&31], i.e. 32 consecutive
// floats per half-wave: conflict-free ds_read_b32, no transpose anywhere.  Only tiles on or above
// the diagonal are computed; the finish kernel mirrors them.
#include <hip/hip_runtime.h>

#include "nst_kernels.h"

namespace nst {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TS>
struct GramCfg {
    static constexpr int KP = (TS == 128) ? 32 : 128;     // pixels per staged chunk
    static constexpr int SIDES = (TS == 128) ? 2 : 1;
    static constexpr int UNITS = KP * TS / 4;              // 16-byte units per side per chunk
    static constexpr int PER_T = UNITS / 256;
    static constexpr int SIDE_FLOATS = KP * TS;
    static constexpr int BUF_FLOATS = SIDES * SIDE_FLOATS;
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;   // 64 KiB in both shapes
};

template <int TS>
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ f, size_t N, int C, int nsplit,
                                                           size_t pix_per_split, float* __restrict__ part) {
    using G = GramCfg<TS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    const int split = blockIdx.x % nsplit;
    int tp = blockIdx.x / nsplit;
    const int T = C / TS;
    int ti = 0;
    while (tp >= T - ti) { tp -= T - ti; ++ti; }
    const int tj = ti + tp;
    const bool diag = (ti == tj);

    const size_t p0 = (size_t)split * pix_per_split;
    size_t p1 = p0 + pix_per_split;
    if (p1 > N) p1 = N;
    const int nchunks = (p0 < p1) ? (int)((p1 - p0 + G::KP - 1) / G::KP) : 0;

    const int wm = (TS == 128) ? (wave & 1) : 0;
    const int wn = (TS == 128) ? (wave >> 1) : 0;
    const int kbase = (TS == 128) ? 0 : 32 * wave;

    f32x4 ra[G::PER_T];
    f32x4 rb[G::PER_T];

    auto load = [&](int chunk) {
        const size_t pc = p0 + (size_t)chunk * G::KP;
#pragma unroll
        for (int i = 0; i < G::PER_T; ++i) {
            const int u = tid + i * 256;
            const int pix = u / (TS / 4);
            const int q = u - pix * (TS / 4);
            const size_t gp = pc + pix;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (gp < p1) {
                va = *reinterpret_cast<const f32x4*>(f + gp * C + ti * TS + q * 4);
                if (G::SIDES == 2 && !diag) vb = *reinterpret_cast<const f32x4*>(f + gp * C + tj * TS + q * 4);
            }
            ra[i] = va;
            rb[i] = vb;
        }
    };
    auto store = [&](int buf) {
        float* base = smem + buf * G::BUF_FLOATS;
#pragma unroll
        for (int i = 0; i < G::PER_T; ++i) {
            const int u = tid + i * 256;
            *reinterpret_cast<f32x4*>(base + u * 4) = ra[i];
            if (G::SIDES == 2 && !diag) *reinterpret_cast<f32x4*>(base + G::SIDE_FLOATS + u * 4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (nchunks > 0) load(0);
    int cur = 0;
    for (int c = 0; c < nchunks; ++c) {
        store(cur);
        __syncthreads();
        if (c + 1 < nchunks) load(c + 1);
        const float* abase = smem + cur * G::BUF_FLOATS;
        const float* bbase = (G::SIDES == 2 && !diag) ? abase + G::SIDE_FLOATS : abase;
        const float* arow = abase + (kbase + half) * TS + wm * 64 + l31;
        const float* brow = bbase + (kbase + half) * TS + wn * 64 + l31;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const float a0 = arow[k2 * 2 * TS];
            const float a1 = arow[k2 * 2 * TS + 32];
            const float b0 = brow[k2 * 2 * TS];
            const float b1 = brow[k2 * 2 * TS + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        cur ^= 1;
    }

    float* slab = part + (size_t)split * C * C;
    if (TS == 128) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int row = ti * TS + wm * 64 + mt * 32 + m;
                    const int col = tj * TS + wn * 64 + nt * 32 + l31;
                    slab[(size_t)row * C + col] = acc[mt][nt][r];
                }
    } else {
        // the four waves hold partial sums over disjoint pixel rows: combine through LDS in wave order
        __syncthreads();
        float* mine = smem + wave * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    mine[(mt * 32 + m) * 64 + nt * 32 + l31] = acc[mt][nt][r];
                }
        __syncthreads();
        for (int e = tid; e < 4096; e += 256) {
            const float s = ((smem[e] + smem[4096 + e]) + smem[8192 + e]) + smem[12288 + e];
            const int row = ti * TS + (e >> 6), col = tj * TS + (e & 63);
            slab[(size_t)row * C + col] = s;
        }
    }
}

// ---- the same partial Gram on the fp16 matrix pipe (see conv_h2.hip for the arithmetic) ------------------------------
// Both operands are the SAME tensor with one power-of-two scale s (from its recorded absmax), cut into two fp16
// pieces, hi = fp16(x s), lo = fp16((x s - hi) 2^11); G s^2 = sum hi hi' + 2^-11 sum (hi lo' + lo hi') with the main
// and the cross products in separate fp32 accumulators: 3 v_mfma_f32_32x32x16_f16 per 32x32x16 block.  That MFMA
// wants 8 consecutive k (= pixels) per lane, so the NHWC map is transposed while it is staged: a lane loads 8
// pixels of ONE channel (a wave's 64 lanes = 64 consecutive channels of a pixel: 256-byte segments) and writes
// their 8 hi and 8 lo pieces as two 16-byte words into the channel's LDS row
// [row = channel][piece][32 pixels] (128 B + 16 B pad: conflict-free ds_read_b128 fragments).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TS>
struct GramH2Cfg {
    static constexpr int KP = (TS == 128) ? 32 : 128;          // pixels per staged chunk
    static constexpr int SIDES = (TS == 128) ? 2 : 1;
    static constexpr int ROWB = 144;
    static constexpr int ROWS = (TS == 128) ? 256 : 256;       // 2 sides x 128 channels | 4 pixel slices x 64 channels
    static constexpr int BUF_BYTES = ROWS * ROWB;
    static constexpr int UNITS = TS * (KP / 8);                // (channel, 8-pixel group) units per side per chunk
    static constexpr int PER_T = UNITS / 256;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;            // 72 KiB: two workgroups per CU
};

template <int TS>
__device__ __forceinline__ void gram_h2_body(const float* __restrict__ f, size_t N, int C, int nsplit, size_t pix_per_split,
                                             const unsigned* __restrict__ amax, float* __restrict__ part, const int bid) {
    using G = GramH2Cfg<TS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h2[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;

    const int split = bid % nsplit;
    int tp = bid / nsplit;
    const int T = C / TS;
    int ti = 0;
    while (tp >= T - ti) { tp -= T - ti; ++ti; }
    const int tj = ti + tp;
    const bool diag = (ti == tj);

    const size_t p0 = (size_t)split * pix_per_split;
    size_t p1 = p0 + pix_per_split;
    if (p1 > N) p1 = N;
    const int nchunks = (p0 < p1) ? (int)((p1 - p0 + G::KP - 1) / G::KP) : 0;

    // scale from the recorded absmax (same rule as conv_h2.hip::tensor_scale)
    float sc, inv;
    {
        unsigned m = amax[lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)m, off);
            m = o > m ? o : m;
        }
        int e = (int)((m >> 23) & 0xFFu);
        e = e < 32 ? 32 : (e > 250 ? 250 : e);
        sc = __uint_as_float((unsigned)(268 - e) << 23);
        inv = __uint_as_float((unsigned)(e - 14) << 23);
    }

    // pixels beyond p1 read as zeros (buffer range check).  The descriptor starts at this split's first pixel, so the
    // 32-bit offsets span one split only and the map itself may exceed 4 GiB.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(f) + p0 * (size_t)C, 0, (p0 < p1) ? (unsigned)((p1 - p0) * (size_t)C * 4) : 0u, 0x00020000);
    float st[G::SIDES][G::PER_T][8];
    auto load = [&](int chunk) {
        const unsigned base = (unsigned)((size_t)chunk * G::KP * (size_t)C * 4);
#pragma unroll
        for (int sd = 0; sd < G::SIDES; ++sd) {
            if (sd == 1 && diag) break;
            const int cb = (sd == 0 ? ti : tj) * TS;
#pragma unroll
            for (int i = 0; i < G::PER_T; ++i) {
                const int u = tid + i * 256;
                const int ch = u % TS, gg = u / TS;
                const unsigned voff = (unsigned)(((gg * 8) * C + cb + ch) * 4);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    st[sd][i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, base + (unsigned)(j * C * 4), 0));
            }
        }
    };
    auto store = [&](int buf) {
        unsigned char* base = smem_h2 + buf * G::BUF_BYTES;
#pragma unroll
        for (int sd = 0; sd < G::SIDES; ++sd) {
            if (sd == 1 && diag) break;
#pragma unroll
            for (int i = 0; i < G::PER_T; ++i) {
                const int u = tid + i * 256;
                const int ch = u % TS, gg = u / TS;
                const int row = (TS == 128) ? sd * 128 + ch : (gg >> 2) * 64 + ch;
                unsigned hi[4], lo[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x2 x = {st[sd][i][2 * k] * sc, st[sd][i][2 * k + 1] * sc};
                    const f16x2 h = __builtin_convertvector(x, f16x2);
                    const f32x2 r = (x - __builtin_convertvector(h, f32x2)) * 2048.f;
                    const f16x2 l = __builtin_convertvector(r, f16x2);
                    hi[k] = __builtin_bit_cast(unsigned, h);
                    lo[k] = __builtin_bit_cast(unsigned, l);
                }
                unsigned char* dst = base + row * G::ROWB + (gg & 3) * 16;
                *reinterpret_cast<u32x4*>(dst) = u32x4{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<u32x4*>(dst + 64) = u32x4{lo[0], lo[1], lo[2], lo[3]};
            }
        }
    };

    const int wm = (TS == 128) ? (wave & 1) : 0;
    const int wn = (TS == 128) ? (wave >> 1) : 0;
    const int arow = (TS == 128) ? wm * 64 : wave * 64;
    const int brow = (TS == 128) ? (diag ? 0 : 128) + wn * 64 : wave * 64;
    const int a_off = (arow + l31) * G::ROWB + half * 16;
    const int b_off = (brow + l31) * G::ROWB + half * 16;

    f32x16 accm[2][2], accx[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.f; accx[a][b][r] = 0.f; }

    if (nchunks > 0) {
        load(0);
        store(0);
        if (nchunks > 1) load(1);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cb = c & 1;
        if (c + 1 < nchunks) {
            store(cb ^ 1);              // the other buffer was last read in chunk c-1, which every wave has left
            if (c + 2 < nchunks) load(c + 2);
        }
        const unsigned char* base = smem_h2 + cb * G::BUF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    fa[t][pc] = *reinterpret_cast<const f16x8*>(base + a_off + t * 32 * G::ROWB + pc * 64 + ks * 32);
                    fb[t][pc] = *reinterpret_cast<const f16x8*>(base + b_off + t * 32 * G::ROWB + pc * 64 + ks * 32);
                }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][1], fb[nt][0], accx[mt][nt], 0, 0, 0);
                    accm[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][0], accm[mt][nt], 0, 0, 0);
                    accx[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][0], fb[nt][1], accx[mt][nt], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    const float inv2 = inv * inv;
    float* slab = part + (size_t)split * C * C;
    if (TS == 128) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int row = ti * TS + wm * 64 + mt * 32 + m;
                    const int col = tj * TS + wn * 64 + nt * 32 + l31;
                    slab[(size_t)row * C + col] = fmaf(accx[mt][nt][r], 1.f / 2048.f, accm[mt][nt][r]) * inv2;
                }
    } else {
        // the four waves hold partial sums over disjoint pixel slices: combine through LDS in wave order
        float* red = reinterpret_cast<float*>(smem_h2);
        float* mine = red + wave * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    mine[(mt * 32 + m) * 64 + nt * 32 + l31] = fmaf(accx[mt][nt][r], 1.f / 2048.f, accm[mt][nt][r]);
                }
        __syncthreads();
        for (int e = tid; e < 4096; e += 256) {
            const float sum = ((red[e] + red[4096 + e]) + red[8192 + e]) + red[12288 + e];
            const int row = ti * TS + (e >> 6), col = tj * TS + (e & 63);
            slab[(size_t)row * C + col] = sum * inv2;
        }
    }
}

template <int TS>
__global__ __launch_bounds__(256, 2) void gram_h2_kernel(const float* __restrict__ f, size_t N, int C, int nsplit,
                                                         size_t pix_per_split, const unsigned* __restrict__ amax,
                                                         float* __restrict__ part) {
    gram_h2_body<TS>(f, N, C, nsplit, pix_per_split, amax, part, blockIdx.x);
}
// several maps (the style taps of every pyramid level of a closure) in one launch: their workgroups are numbered
// item by item, so the short ones fill the tail of the long ones
template <int TS>
__global__ __launch_bounds__(256, 2) void gram_h2_batch_kernel(GramBatch b) {
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i].part_end) ++i;
    const GramItem& it = b.it[i];
    gram_h2_body<TS>(it.f, it.N, it.C, it.nsplit, it.pix_per_split, it.amax, it.part,
                     (int)blockIdx.x - (i ? b.it[i - 1].part_end : 0));
}

// generic fallback for channel counts that are not a multiple of 64 (unit-parity API only)
__global__ void gram_generic_kernel(const float* __restrict__ f, size_t N, int C, float* __restrict__ part) {
    const int i = blockIdx.x / C, j = blockIdx.x % C;
    __shared__ float sh[256];
    float s = 0.f;
    for (size_t n = threadIdx.x; n < N; n += blockDim.x) s += f[n * C + i] * f[n * C + j];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)i * C + j] = sh[0];
}

// 64-wide tiles have one operand side only, so that shape serves C == 64 alone
static inline int gram_ts(int C) { return (C % 128 == 0) ? 128 : ((C == 64) ? 64 : 0); }

int gram_nsplit(int C, size_t N) {
    const int ts = gram_ts(C);
    if (ts == 0) return 1;
    const int kp = (ts == 128) ? 32 : 128;
    const int T = C / ts;
    const int pairs = T * (T + 1) / 2;
    const size_t chunks = (N + kp - 1) / kp;
    size_t ns = 256 / pairs;          // (with 512 the partial slabs cost more HBM traffic than the maps themselves)
    const size_t by_bytes = ((size_t)32 << 20) / ((size_t)C * C * 4);
    if (ns > by_bytes) ns = by_bytes;
    if (ns > chunks) ns = chunks;
    if (ns < 1) ns = 1;
    const size_t cps = (chunks + ns - 1) / ns;
    return (int)((chunks + cps - 1) / cps);
}

int gram_nslabs(int C, int nsplit) { (void)C; return nsplit; }

hipError_t gram_init_device() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_partial_kernel<128>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GramCfg<128>::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_partial_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GramCfg<64>::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_h2_kernel<128>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GramH2Cfg<128>::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_h2_batch_kernel<128>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GramH2Cfg<128>::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_h2_batch_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, GramH2Cfg<64>::LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_h2_kernel<64>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GramH2Cfg<64>::LDS_BYTES);
}

hipError_t launch_gram_partial(const float* f, size_t N, int C, int nsplit, const unsigned* amax, float* part,
                               hipStream_t stream) {
    const int ts = gram_ts(C);
    if (ts == 0) {
        hipLaunchKernelGGL(gram_generic_kernel, dim3(C * C), dim3(256), 0, stream, f, N, C, part);
        return hipGetLastError();
    }
    const int kp = (ts == 128) ? 32 : 128;
    const int T = C / ts;
    const int pairs = T * (T + 1) / 2;
    const size_t chunks = (N + kp - 1) / kp;
    const size_t cps = (chunks + nsplit - 1) / nsplit;
    const size_t pix_per_split = cps * kp;
    // fp16-piece kernel when the absmax record of f is available (32-bit buffer offsets inside one split)
    const bool h2 = amax != nullptr && pix_per_split * (size_t)C * 4 < 0xFFFFFF00ull;
    if (h2 && ts == 128) {
        hipLaunchKernelGGL(gram_h2_kernel<128>, dim3(pairs * nsplit), dim3(256), GramH2Cfg<128>::LDS_BYTES, stream, f, N, C,
                           nsplit, pix_per_split, amax, part);
    } else if (h2) {
        hipLaunchKernelGGL(gram_h2_kernel<64>, dim3(pairs * nsplit), dim3(256), GramH2Cfg<64>::LDS_BYTES, stream, f, N, C,
                           nsplit, pix_per_split, amax, part);
    } else if (ts == 128) {
        hipLaunchKernelGGL(gram_partial_kernel<128>, dim3(pairs * nsplit), dim3(256), GramCfg<128>::LDS_BYTES, stream, f,
                           N, C, nsplit, pix_per_split, part);
    } else {
        hipLaunchKernelGGL(gram_partial_kernel<64>, dim3(pairs * nsplit), dim3(256), GramCfg<64>::LDS_BYTES, stream, f,
                           N, C, nsplit, pix_per_split, part);
    }
    return hipGetLastError();
}

// G = (sum of the slabs, in slab order) / divisor; optional target: S = coef * (G - Gt) and the partial sums
// of (G - Gt)^2.  A block owns 128 consecutive elements, four per lane (one 16-byte read per slab); its 8 lane groups each
// add every 8th slab, then the 8 group sums are added in group order - a fixed order, so the result is reproducible.
// (Blocks of 32 elements - 57 000 of them for an L=2 closure, each a handful of 128-byte reads, a barrier and a serial tail -
// made this pass launch- and latency-bound: 0.129 ms for 273 MB.)
constexpr int GF_EPB = NST_GRAM_FINISH_EPB;      // elements per block
__device__ __forceinline__ void gram_finish_body(const float* __restrict__ part, int nslabs, int C, int ts, float divisor,
                                                 const float* __restrict__ target, float coef, float* __restrict__ gram_out,
                                                 float* __restrict__ S, unsigned short* __restrict__ S_bf,
                                                 unsigned* __restrict__ S_amax, double* __restrict__ mse_partial,
                                                 const unsigned bid) {
    __shared__ f32x4 sh[8][32];
    __shared__ double shd[32];
    const size_t CC = (size_t)C * C;
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t e0 = (size_t)bid * GF_EPB + (size_t)el * 4;      // this lane's first element (C % 4 == 0: one row)
    const int i = (int)(e0 / C), j0 = (int)(e0 % C);
    // The tiled kernels compute the tiles on or above the diagonal only, and inside a diagonal tile the fp16-piece
    // kernel adds the two cross products of (i,j) and (j,i) in opposite orders: only elements with j >= i are read
    // and each result is written to (i,j) AND (j,i), which makes G exactly symmetric.  With C a multiple of 128 a
    // block's segment lies in one row; segments entirely below the diagonal do nothing.
    const bool tri = ts > 0;
    if (tri && C % GF_EPB == 0 && (int)((size_t)bid * GF_EPB % C) + GF_EPB - 1 < (int)((size_t)bid * GF_EPB / C)) {
        if (threadIdx.x == 0 && mse_partial) mse_partial[bid] = 0.0;
        return;
    }
    // the lane reads its four elements when any of them is on or above the diagonal (the rest are discarded below)
    const bool any = e0 < CC && (!tri || j0 + 3 >= i);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (!tri) {
        // the generic shapes (unit-parity API: any channel count): element by element, bounds checked
        for (int k = grp; k < nslabs; k += 8)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (e0 + q < CC) s[q] += part[(size_t)k * CC + e0 + q];
    } else if (any) {
        // eight slab reads in flight per lane (the adds stay in slab order)
        int k = grp;
        for (; k + 56 < nslabs; k += 64) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(part + (size_t)(k + 8 * q) * CC + e0);
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        // the rest four at a time, slabs beyond the last read as +0 (x + 0 = x: same sum, same order)
        for (; k < nslabs; k += 32) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[q] = (k + 8 * q < nslabs) ? *reinterpret_cast<const f32x4*>(part + (size_t)(k + 8 * q) * CC + e0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) s += v[q];
        }
    }
    sh[grp][el] = s;
    __syncthreads();
    if (grp == 0) {
        f32x4 t = sh[0][el];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += sh[g][el];
        double sq = 0.0;
        float sabs = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t e = e0 + q;
            const int i = (int)(e / C), j = (int)(e % C);      // (a tiled shape's four elements share the row)
            if (!(e < CC && (!tri || j >= i))) continue;
            const bool both = tri && j > i;
            const size_t em = (size_t)j * C + i;      // the mirrored element
            const float g = t[q] / divisor;      // torch: gram /= ch*h*w
            if (gram_out) { gram_out[e] = g; if (both) gram_out[em] = g; }
            if (target) {
                const float d = g - target[e];
                sq += (double)d * (double)d * (both ? 2.0 : 1.0);
                const float sv = coef * d;
                sabs = fmaxf(sabs, fabsf(sv));
                if (S) { S[e] = sv; if (both) S[em] = sv; }
                if (S_bf) {
                    // the same value cut into three bf16 pieces in conv_bf3's weight layout
                    // [row][chunk col/32][piece][col%32] (a 1x1 "conv" weight with Cout = Cin = C)
                    const unsigned uh = __float_as_uint(sv) & 0xFFFF0000u;
                    const float r1 = sv - __uint_as_float(uh);
                    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
                    const float r2 = r1 - __uint_as_float(um);
                    const size_t base = (((size_t)i * (C / 32) + j / 32) * 3) * 32 + (j & 31);
                    S_bf[base] = (unsigned short)(uh >> 16);
                    S_bf[base + 32] = (unsigned short)(um >> 16);
                    S_bf[base + 64] = (unsigned short)(__float_as_uint(r2) >> 16);
                    if (both) {
                        const size_t bm = (((size_t)j * (C / 32) + i / 32) * 3) * 32 + (i & 31);
                        S_bf[bm] = (unsigned short)(uh >> 16);
                        S_bf[bm + 32] = (unsigned short)(um >> 16);
                        S_bf[bm + 64] = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                }
            }
        }
        shd[el] = sq;
        if (S_amax) {
            // absmax of S for the fp16-piece convolution that multiplies by it (conv_h2.hip); lanes 0..31 here
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) sabs = fmaxf(sabs, __shfl_xor(sabs, off));
            if (el == 0) atomicMax(S_amax + (bid & (NST_AMAX_SLOTS - 1)), __float_as_uint(sabs));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && mse_partial) {
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += shd[k];
        mse_partial[bid] = t;
    }
}
__global__ __launch_bounds__(256) void gram_finish_kernel(const float* __restrict__ part, int nslabs, int C, int ts,
                                                          float divisor, const float* __restrict__ target, float coef,
                                                          float* __restrict__ gram_out, float* __restrict__ S,
                                                          unsigned short* __restrict__ S_bf,
                                                          unsigned* __restrict__ S_amax,
                                                          double* __restrict__ mse_partial) {
    gram_finish_body(part, nslabs, C, ts, divisor, target, coef, gram_out, S, S_bf, S_amax, mse_partial, blockIdx.x);
}
__global__ __launch_bounds__(256) void gram_finish_batch_kernel(GramBatch b) {
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i].finish_end) ++i;
    const GramItem& it = b.it[i];
    gram_finish_body(it.part, it.nsplit, it.C, (it.C % 128 == 0) ? 128 : 64, it.divisor, it.target, it.coef, it.gram_out, it.S,
                     it.S_bf, it.S_amax, it.mse_partial, blockIdx.x - (unsigned)(i ? b.it[i - 1].finish_end : 0));
}

int gram_finish_blocks(int C) { return (int)(((size_t)C * C + GF_EPB - 1) / GF_EPB); }

// Splits of an item inside a batch: the largest map of each channel count (the top pyramid level) takes gram_nsplit and
// fills the chip by itself; a smaller map of the same channel count gets the same PIXELS per workgroup instead of the same
// number of workgroups - its slabs (nsplit x C x C floats, written by the partial pass and read back by the finish pass)
// shrink with its size.  L=2 closure: 273 -> 134 MB of slabs.  Never more than gram_nsplit(C, N): the partial buffers are
// sized for that.
static int gram_nsplit_in_batch(const GramBatch& b, int i) {
    size_t nmax = b.it[i].N;
    for (int k = 0; k < b.n; ++k)
        if (b.it[k].C == b.it[i].C && b.it[k].N > nmax) nmax = b.it[k].N;
    const int own = gram_nsplit(b.it[i].C, b.it[i].N);
    const size_t scaled = ((size_t)gram_nsplit(b.it[i].C, nmax) * b.it[i].N + nmax - 1) / nmax;
    int ns = (int)(scaled < 1 ? 1 : scaled);
    if (ns > own) ns = own;
    // (whole chunks per split, as gram_nsplit rounds)
    const int kp = (gram_ts(b.it[i].C) == 128) ? 32 : 128;
    const size_t chunks = (b.it[i].N + kp - 1) / kp;
    const size_t cps = (chunks + ns - 1) / ns;
    return (int)((chunks + cps - 1) / cps);
}

// Batched forms (fp16-piece kernels only: every item needs its absmax record, C = 64 or a multiple of 128, and its
// own partial buffer of nsplit x C x C floats).  Fills nsplit / pix_per_split / the block prefixes.
hipError_t launch_gram_batch(const GramBatch& b0, hipStream_t stream) {
    if (b0.n < 1 || b0.n > NST_GRAM_BATCH_MAX) return hipErrorInvalidValue;
    // partial products: one launch per tile shape
    for (int ts = 128; ts >= 64; ts -= 64) {
        GramBatch b{};
        for (int i = 0; i < b0.n; ++i) {
            const GramItem& src = b0.it[i];
            if (gram_ts(src.C) != ts) continue;
            if (!src.amax) return hipErrorInvalidValue;
            GramItem& it = b.it[b.n];
            it = src;
            const int kp = (ts == 128) ? 32 : 128;
            const int T = it.C / ts, pairs = T * (T + 1) / 2;
            const size_t chunks = (it.N + kp - 1) / kp;
            it.nsplit = gram_nsplit_in_batch(b0, i);
            it.pix_per_split = ((chunks + it.nsplit - 1) / it.nsplit) * kp;
            // 32-bit buffer offsets inside ONE split (the map itself may be larger)
            if (it.pix_per_split * (size_t)it.C * 4 >= 0xFFFFFF00ull) return hipErrorInvalidValue;
            it.part_end = (b.n ? b.it[b.n - 1].part_end : 0) + pairs * it.nsplit;
            ++b.n;
        }
        if (b.n == 0) continue;
        const int blocks = b.it[b.n - 1].part_end;
        if (ts == 128) hipLaunchKernelGGL(gram_h2_batch_kernel<128>, dim3(blocks), dim3(256), GramH2Cfg<128>::LDS_BYTES, stream, b);
        else hipLaunchKernelGGL(gram_h2_batch_kernel<64>, dim3(blocks), dim3(256), GramH2Cfg<64>::LDS_BYTES, stream, b);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // ordered finish of all of them
    GramBatch b = b0;
    for (int i = 0; i < b.n; ++i) {
        if (gram_ts(b.it[i].C) == 0) return hipErrorInvalidValue;
        b.it[i].nsplit = gram_nsplit_in_batch(b0, i);               // = number of slabs
        b.it[i].finish_end = (i ? b.it[i - 1].finish_end : 0) + gram_finish_blocks(b.it[i].C);
        if (b.it[i].C % 32 != 0) b.it[i].S_bf = nullptr;
    }
    hipLaunchKernelGGL(gram_finish_batch_kernel, dim3(b.it[b.n - 1].finish_end), dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t launch_gram_finish(const float* part, int nslabs, int C, float divisor, const float* target, float coef,
                              float* gram_out, float* S, unsigned short* S_bf, unsigned* S_amax, double* mse_partial,
                              hipStream_t stream) {
    hipLaunchKernelGGL(gram_finish_kernel, dim3(gram_finish_blocks(C)), dim3(256), 0, stream, part, nslabs, C,
                       gram_ts(C), divisor, target, coef, gram_out, S, (C % 32 == 0) ? S_bf : nullptr, S_amax, mse_partial);
    return hipGetLastError();
}

}  // namespace nst
