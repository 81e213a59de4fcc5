// The whole first block of Darknet-19 -- conv1 (3 -> 32, 3x3) + BatchNorm(train) + LeakyReLU + MaxPool(2,2) -- forward
// and backward WITHOUT ever storing the block's full-resolution tensors.
//
// At B = 64 the raw output of conv1 is 709 MB of fp16 (64 x 416 x 416 x 32) for 0.4 % of the network's FLOPs; the
// unfused path writes it, reads it for BatchNorm + pool, reads it twice more in the BatchNorm backward, writes the
// equally large dY and reads that again for the weight gradient: ~4.5 GB of HBM traffic and 1.05 ms of a 10.8 ms step.
// Both tensors are redundant: the layer has only K = 27 inputs per pixel, so everything the BatchNorm needs from y is
// a function of the 27 x 27 Gram matrix of the image windows, and y itself can be recomputed from the 88 MB image
// faster than it can be read back.
//
//   v_m      = im2col row of output pixel m (27 values; slot (ty, tx, c)),   y[m][n] = W[n] . v_m
//   C        = sum_m v_m v_m^T,   S = sum_m v_m                  (stem_gram_kernel: image only, MFMA F^T F)
//   mean_n   = W[n] . S / M,      E[y^2]_n = W[n]^T C W[n] / M   (stem_coeffs_kernel, double)
//   forward  : image -> y (registers) -> scale/shift -> 2x2 max -> LeakyReLU -> pooled fp16 output   (one pass)
//   backward : recompute y and the window argmax from the image, g_z = G * leaky'(z) at the argmax position,
//              T[n][k] = sum_m g_z[m][n] v_m[k] (MFMA), dbeta_n = sum_m g_z[m][n]                     (one pass)
//              dgamma_n = invstd_n (W[n] . T[n] - mean_n dbeta_n)
//              dW[n][k] = gamma_n invstd_n (T[n][k] - dbeta_n/M S[k] - dgamma_n/M invstd_n ((W C)[n][k] - mean_n S[k]))
//   (the last line is dW = sum_m dy[m][n] v_m[k] with the BatchNorm backward dy = gamma invstd (g_z - mean(g_z) -
//   xhat mean(g_z xhat)) expanded: sum_m xhat[m][n] v_m[k] = invstd_n ((W C)[n][k] - mean_n S[k]).)
//
// HBM traffic per step: image 3 x 88 MB + pooled output 177 MB + its gradient 177 MB.  y is never rounded to fp16,
// so the block is closer to the fp32 reference than the unfused path.
//
// Replaces, for the first block, F.conv2d (reference src/pruning/weightPruning/layers.py:60-64), nn.BatchNorm2d,
// nn.LeakyReLU(0.1), nn.MaxPool2d(2, 2) (src/nets.py:798-821) and their autograd backward.
#include "kernels.h"
#include "tr_frag.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct StemBlockArgs {
    const half_t* x;        // padded NHWC4 image, pixel (0, 0, 0)
    const half_t* w;        // packed stem weights [>= 32][96], k = ty*32 + tx*4 + c
    const half_t* xl;       // split operands: lo image, fp16(x - fp16(x)), same layout (NULL = plain operands)
    const half_t* wl;       // split operands: lo weights, fp16(w - fp16(w)), same packing
    float* stats;           // stats pass: fp32 [gridDim.x][2][stats_ld] per-channel sums / sums of squares of y
    int stats_ld;
    const float* scale;     // [32]
    const float* shift;     // [32]
    half_t* out;            // pooled output, padded NHWC pixel (0, 0, 0)
    const half_t* g;        // gradient wrt the pooled output [B*H2*W2][g_ld]
    float* slab;            // per workgroup partial results
    int out_ld, out_choff, g_ld, g_choff;
    int B, H, W, H2, W2, Wb;   // Wb = W / 32
    int cout;                  // filters computed (<= 32; the forward pass writes zeros behind them)
    long long nunits;
    float slope;
};

__device__ __forceinline__ h4_t lo4(h8_t v) { return h4_t{v[0], v[1], v[2], v[3]}; }

// A wave-uniform 64-bit element offset pinned to scalar registers: base pointer + this + a 32-bit lane byte offset
// compiles to the scalar-base form of global_load (no 64-bit address arithmetic per lane).
__device__ __forceinline__ long long uniform_off(long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// Position of a work unit (image b, unit row h, 32-column block w) in 32-bit arithmetic, stepped incrementally: a
// 64-bit decode of the unit index per step cost more instructions than the step's MFMAs.  Units are numbered DOWN a
// 32-pixel column strip first (h fastest): consecutive units of a wave share image rows (3 of 3 + 1 for the Gram steps,
// 2 of 4 for the two-row units), which then come from its own L1 / L2 instead of being fetched again by another XCD.
struct UnitPos {
    int b, h, w;
};
__device__ __forceinline__ UnitPos unit_decode(int u, int Wb, int Hn) {
    UnitPos p;
    p.h = u % Hn;
    const int t = u / Hn;
    p.w = t % Wb;
    p.b = t / Wb;
    return p;
}
__device__ __forceinline__ void unit_advance(UnitPos& p, const UnitPos& d, int Wb, int Hn) {   // p += d (d from unit_decode)
    p.h += d.h;
    if (p.h >= Hn) p.h -= Hn, ++p.w;
    p.w += d.w;
    if (p.w >= Wb) p.w -= Wb, ++p.b;
    p.b += d.b;
}

// Pitch of a staged window row (34 pixels x 8 bytes = 272 used).  384 = 96 banks: the four 16-lane groups of a
// transposing read (row 0 | 1 at pitch distance, pixels +8 at 64 bytes) then touch disjoint bank ranges
// (0-13, 32-45, 16-29, 48-61); at 272 bytes they overlapped and the reads of a step took longer than its MFMAs.
constexpr int XROW = 384;

// Window rows from the registers that feed the forward MFMAs into LDS; `ones`: channel 3 of every INTERIOR pixel is set
// to 1.0 (the NHWC4 image keeps 0 there), so that column (tap, 3) of the Gram matrix holds the plain sums S.
// Lane (pl, kg) holds pixels pl + 2 kg, pl + 2 kg + 1 of each row and writes the first of them: lanes kg = 0 cover pixels
// 0-31, lanes (30, 1), (31, 1) pixels 32, 33, the other kg = 1 lanes repeat what lane (pl + 2, 0) writes (no branches).
template <int NR>
__device__ __forceinline__ void stage_window(char* win, const h8_t (&xr)[NR], int pl, int kg, int prow0, int pcol0, int H,
                                             int W, bool ones) {
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        h4_t v = lo4(xr[rr]);
        if (ones) {
            const int c0 = pcol0 + pl + 2 * kg;
            v[3] = (prow0 + rr >= 1 && prow0 + rr <= H && c0 >= 1 && c0 <= W) ? (half_t)1.f : (half_t)0.f;
        }
        *(h4_t*)(win + rr * XROW + (pl + 2 * kg) * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Gram matrix of the image windows.  A step = 32 consecutive output pixels of one row; its window (3 rows x 34 pixels)
// goes to LDS and the MFMA fragments are gathered with the transposing read exactly as in wgrad_stem_kernel: F01 =
// columns (ty in {0, 1}) x (tx, c), F2 = ty = 2.  C = F^T F: the SAME registers serve as the A and the B operand
// (A[i][k] = F[k][i]).  Slab per workgroup: fp32 [48][48], slot = ty*16 + tx*4 + c (tx = 3 is a don't-care column).
__global__ __launch_bounds__(512, 4) void stem_gram_kernel(StemBlockArgs a) {     // 4 waves per SIMD = 2 workgroups per CU: <= 128 registers
    constexpr int PERW = 3 * XROW + 128, NW = 8;     // two window buffers of PERW bytes per wave
    __shared__ __attribute__((aligned(16))) char smem[NW * 16 * 64 * 4];   // the final reduction needs 32 KB
    static_assert(NW * 2 * PERW <= NW * 16 * 64 * 4, "window buffers");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, kg = lane >> 5;
    char* my = smem + wave * (2 * PERW);
    const unsigned my_addr = lds_addr_of(my);
    for (int i = lane; i < 2 * PERW / 4; i += 64) ((int*)my)[i] = 0;    // the fragment reads touch bytes past the staged pixels
    const int g4 = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int krow = 8 * (g4 >> 1) + q;
    const unsigned b01 = my_addr + (g4 & 1) * XROW + (krow + p) * 8;
    const unsigned b2 = my_addr + 2 * XROW + (krow + p) * 8;

    f32x16_t cbb, cbc, ccc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cbb[r] = cbc[r] = ccc[r] = 0.f;

    // A wave takes batches of GD consecutive steps: all image loads of a batch are in flight before its first window
    // is staged, and inside a batch the windows alternate between two LDS buffers so that the LDS round trip of window
    // d + 1 (write, transposing reads) runs under the MFMAs of window d.
    constexpr int GD = 4;
    const int nunits = (int)a.nunits;
    const int wstride = gridDim.x * NW * GD;
    int u0 = (blockIdx.x * NW + wave) * GD;
    UnitPos pos = unit_decode(u0 < nunits ? u0 : 0, a.Wb, a.H);
    const UnitPos dstep = unit_decode(wstride - (GD - 1), a.Wb, a.H), one = {0, 1, 0};
    const unsigned row_elems = (a.W + 2) * 4, lane_off = (pl + 2 * kg) * 4;
    auto reads = [&](int buf, Frag (&fb)[2], Frag (&fc)[2]) {
        const unsigned o = buf * PERW;
        fb[0].lo = tr_read4<0>(b01 + o), fb[0].hi = tr_read4<4 * 8>(b01 + o);
        fc[0].lo = tr_read4<0>(b2 + o), fc[0].hi = tr_read4<4 * 8>(b2 + o);
        fb[1].lo = tr_read4<16 * 8>(b01 + o), fb[1].hi = tr_read4<16 * 8 + 4 * 8>(b01 + o);
        fc[1].lo = tr_read4<16 * 8>(b2 + o), fc[1].hi = tr_read4<16 * 8 + 4 * 8>(b2 + o);
    };
    for (; u0 < nunits; u0 += wstride) {
        h8_t xr[GD][3];
        int uh[GD], uw[GD];
#pragma unroll
        for (int d = 0; d < GD; ++d) {
            uh[d] = pos.h, uw[d] = pos.w;
            // wave-uniform base (scalar registers) + a 32-bit lane offset
            const half_t* xb = a.x + uniform_off(((long long)(pos.b * (a.H + 2) + pos.h) * (a.W + 2) + pos.w * 32) * 4);
            if (u0 + d < nunits) {
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) xr[d][rr] = *(const h8_t*)((const char*)xb + (lane_off + rr * row_elems) * 2u);
            }
            if (d + 1 < GD) unit_advance(pos, one, a.Wb, a.H);
        }
        unit_advance(pos, dstep, a.Wb, a.H);
        Frag fb[2][2], fc[2][2];
        stage_window<3>(my, xr[0], pl, kg, uh[0], uw[0] * 32, a.H, a.W, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        reads(0, fb[0], fc[0]);
#pragma unroll
        for (int d = 0; d < GD; ++d) {
            if (u0 + d < nunits) {          // wave-uniform
                const bool more = d + 1 < GD && u0 + d + 1 < nunits;
                if (more) stage_window<3>(my + ((d + 1) & 1) * PERW, xr[d + 1 < GD ? d + 1 : d], pl, kg, uh[d + 1 < GD ? d + 1 : d],
                                          uw[d + 1 < GD ? d + 1 : d] * 32, a.H, a.W, true);
                lds_wait_all(fb[d & 1][0]);     // the reads of window d (and the writes of window d + 1) have landed
                if (more) reads((d + 1) & 1, fb[(d + 1) & 1], fc[(d + 1) & 1]);
#pragma unroll
                for (int k16 = 0; k16 < 2; ++k16) {
                    tie(fb[d & 1][k16]), tie(fc[d & 1][k16]);
                    cbb = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[d & 1][k16].v(), fb[d & 1][k16].v(), cbb, 0, 0, 0);
                    cbc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[d & 1][k16].v(), fc[d & 1][k16].v(), cbc, 0, 0, 0);
                    ccc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fc[d & 1][k16].v(), fc[d & 1][k16].v(), ccc, 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    // sum the waves in a fixed order, one accumulator at a time: red[wave][r][lane]; D[row i][col j], i / j = fragment
    // columns.  (fp32 throughout: measured at B=64 the batch mean / invstd derived from C are within 1.2e-7 of float64.)
    float* red = (float*)smem;
    float* out = a.slab + (long long)blockIdx.x * 2304;
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = which == 0 ? cbb[r] : (which == 1 ? cbc[r] : ccc[r]);
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 512) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) v += red[wv * 1024 + idx];
            const int r = idx >> 6, ln = idx & 63;
            const int i = mfma32_row(r, ln), j = ln & 31;
            if (which == 0) out[i * 48 + j] = v;
            else if (which == 1) {
                if (j < 16) out[i * 48 + 32 + j] = v, out[(32 + j) * 48 + i] = v;
            } else if (i < 16 && j < 16) out[(32 + i) * 48 + 32 + j] = v;
        }
        __syncthreads();
    }
}

// out[e] = sum over slabs of slab[s][e], in double, fixed order (16 groups of slabs per entry, combined through LDS)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* slab, int nslabs, int n, double* out) {
    __shared__ double red[16][16];
    const int e16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + e16;
    double s = 0.0;
    if (e < n)
        for (int k = grp; k < nslabs; k += 16) s += (double)slab[(long long)k * n + e];
    red[grp][e16] = s;
    __syncthreads();
    if (grp == 0 && e < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e16];
        out[e] = t;
    }
}

__device__ __forceinline__ int slot48(int ty, int tx, int c) { return ty * 16 + tx * 4 + c; }

// Batch statistics of y = W v from the Gram matrix -> BatchNorm coefficients; keeps S and W C for the backward pass.
// ctx (double): [0, 27) S, [32, 32 + 32*27) (W C)[n][k27], k27 = (ty*3 + tx)*3 + c.
__global__ __launch_bounds__(1024) void stem_coeffs_kernel(const double* csum, const half_t* wp, double count, const float* gamma,
                                                           const float* beta, float* rmean, float* rvar, float momentum,
                                                           float eps, float* scale, float* shift, float* save_mean,
                                                           float* save_invstd, double* ctx) {
    __shared__ double C[48 * 48];
    __shared__ double Wd[32][28], WC[32][28], S[28];
    const int t = threadIdx.x;
    for (int i = t; i < 2304; i += 1024) C[i] = csum[i];
    if (t < 864) {
        const int n = t / 27, k = t - n * 27;
        const int ty = k / 9, tx = (k / 3) % 3, c = k % 3;
        Wd[n][k] = (double)(float)wp[n * 96 + ty * 32 + tx * 4 + c];
    }
    __syncthreads();
    if (t < 27) {
        const int ty = t / 9, tx = (t / 3) % 3, c = t % 3;
        S[t] = C[slot48(ty, tx, c) * 48 + slot48(ty, tx, 3)];
        ctx[t] = S[t];
    }
    if (t < 864) {
        const int n = t / 27, k = t - n * 27;
        const int sk = slot48(k / 9, (k / 3) % 3, k % 3);
        double acc = 0.0;
        for (int k2 = 0; k2 < 27; ++k2) acc += Wd[n][k2] * C[slot48(k2 / 9, (k2 / 3) % 3, k2 % 3) * 48 + sk];
        WC[n][k] = acc;
        ctx[32 + n * 27 + k] = acc;
    }
    __syncthreads();
    if (t < 32) {
        double m1 = 0.0, m2 = 0.0;
        for (int k = 0; k < 27; ++k) {
            m1 += Wd[t][k] * S[k];
            m2 += Wd[t][k] * WC[t][k];
        }
        const double mean = m1 / count;
        double var = m2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double sc = (double)gamma[t] * invstd;
        scale[t] = (float)sc;
        shift[t] = (float)((double)beta[t] - mean * sc);
        save_mean[t] = (float)mean;
        save_invstd[t] = (float)invstd;
        if (rmean) rmean[t] = (float)((1.0 - momentum) * (double)rmean[t] + momentum * mean);
        if (rvar) rvar[t] = (float)((1.0 - momentum) * (double)rvar[t] + momentum * var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward: a unit = 32 columns x 2 rows of the conv output = 16 pooled pixels.  Y = X W^T on v_mfma_f32_32x32x16_f16
// with the image fragment as the A operand (row = pixel, the same 16-byte loads as stem_fwd_kernel) and the
// register-resident weights as B: an accumulator's lane is a CHANNEL and its registers are pixels
// (pixel = (r & 3) + 8 (r >> 2) + 4 kg), so a 2x2 pool window is four registers of one lane -- no cross-lane traffic,
// one scale / shift pair per lane.  Four image rows per unit feed both conv rows.  LeakyReLU is monotonic, so it is
// applied to the window maximum.  The 16 x 32 pooled tile is turned through 1 KB of LDS per wave and leaves as whole
// 16-byte pieces (1 KB contiguous when dst_ld == 32).
__device__ __forceinline__ int pooled_of(int q, int kg) { return (q & 1) + 4 * (q >> 1) + 2 * kg; }   // pooled pixel of register pair q

// SPLIT: the convolution on split operands, y = x_hi w_hi + x_lo w_hi + x_hi w_lo in the fp32 accumulators (three MFMAs
// per product; the dropped x_lo w_lo term is 2^-22): the first block of the "mixed" TRAINING precision, whose operand
// rounding alone would cost 1.9e-2 on the train-mode logits (engine.py MIXED_BUDGET_TRAIN).  The pass stays HBM-bound.
template <int UN, int PL, bool SPLIT = false>   // PL = 3: split (hi | lo | hi) storage of the pooled output (32 channels per plane, adjacent planes)
__global__ __launch_bounds__(256) void stem_block_fwd_kernel(StemBlockArgs a) {
    constexpr int TW = 32 * PL;                       // halfs per pooled pixel in the tile
    __shared__ __attribute__((aligned(16))) half_t tile[4][16 * TW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, kg = lane >> 5;
    h8_t wf[3], wlo[SPLIT ? 3 : 1];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
        wf[ty] = *(const h8_t*)(a.w + (long long)pl * 96 + ty * 32 + 8 * kg);
        if (SPLIT) wlo[ty] = *(const h8_t*)(a.wl + (long long)pl * 96 + ty * 32 + 8 * kg);
    }
    const float sc = pl < a.cout ? a.scale[pl] : 0.f, sh = pl < a.cout ? a.shift[pl] : 0.f;   // slim models: fewer than 32 filters
    half_t* tw = tile[wave];
    const int nunits = (int)a.nunits;
    const int wstride = gridDim.x * 4 * UN;
    int u0 = (blockIdx.x * 4 + wave) * UN;
    UnitPos pos = unit_decode(u0 < nunits ? u0 : 0, a.Wb, a.H2);
    const UnitPos dstep = unit_decode(wstride - (UN - 1), a.Wb, a.H2), one = {0, 1, 0};
    const unsigned row_elems = (a.W + 2) * 4, lane_off = (pl + 2 * kg) * 4;
    for (; u0 < nunits; u0 += wstride) {
        h8_t xr[UN][4], xlo[SPLIT ? UN : 1][4];
        int ub[UN], uh[UN], uw[UN];
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            ub[i] = pos.b, uh[i] = pos.h, uw[i] = pos.w;
            const long long xoff = uniform_off(((long long)(pos.b * (a.H + 2) + 2 * pos.h) * (a.W + 2) + pos.w * 32) * 4);
            const half_t* xb = a.x + xoff;
            if (u0 + i < nunits) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) xr[i][rr] = *(const h8_t*)((const char*)xb + (lane_off + rr * row_elems) * 2u);
                if (SPLIT) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) xlo[i][rr] = *(const h8_t*)((const char*)(a.xl + xoff) + (lane_off + rr * row_elems) * 2u);
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    xr[i][rr] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
                    if (SPLIT) xlo[i][rr] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
            if (i + 1 < UN) unit_advance(pos, one, a.Wb, a.H2);
        }
        unit_advance(pos, dstep, a.Wb, a.H2);
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            f32x16_t acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
            if (SPLIT) {     // the small terms first, so that the large products round last
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xlo[i][ty], wf[ty], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xlo[i][ty + 1], wf[ty], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty], wlo[ty], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty + 1], wlo[ty], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty], wf[ty], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty + 1], wf[ty], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float z00 = __builtin_fmaf(acc0[2 * q], sc, sh), z01 = __builtin_fmaf(acc0[2 * q + 1], sc, sh);
                const float z10 = __builtin_fmaf(acc1[2 * q], sc, sh), z11 = __builtin_fmaf(acc1[2 * q + 1], sc, sh);
                float m = fmaxf(fmaxf(z00, z01), fmaxf(z10, z11));
                m = m > 0.f ? m : m * a.slope;
                m = fminf(fmaxf(m, -65504.f), 65504.f);
                const half_t hi = (half_t)m;
                half_t* tp = tw + pooled_of(q, kg) * TW + pl;
                tp[0] = hi;
                if (PL >= 2) tp[32] = (half_t)(m - (float)hi);   // exact difference, one rounding; a consumer with split operands reads hi and lo
                if (PL == 3) tp[64] = hi;                          // (PL == 2: the consumer wraps its third K part onto the hi plane)
            }
#pragma unroll
            for (int k = 0; k < PL; ++k) {
                const int piece = lane + 64 * k;
                const int prow = piece / (4 * PL), pc = piece - prow * (4 * PL);
                const h8_t v = *(const h8_t*)(tw + prow * TW + pc * 8);
                if (u0 + i < nunits)
                    *(h8_t*)(a.out + ((long long)(ub[i] * (a.H2 + 2) + uh[i] + 1) * (a.W2 + 2) + uw[i] * 16 + prow + 1) * a.out_ld +
                             a.out_choff + pc * 8) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Statistics pass of the split-operand first block: the same units and the same accumulators as the forward pass, but
// nothing is written except per-channel sums and sums of squares of the fp32 conv output (a lane is a channel, its
// registers are pixels: the sums are plain register adds).  One slab row per workgroup, mcamd_bn_coeffs finishes them in
// double as it does for every other block -- batch statistics of the UNROUNDED y, as the reference forms them
// (nn.BatchNorm2d on the fp32 conv output, src/nets.py:802).  The Gram route of the plain-operand block is exact for the
// fp16-rounded image only; a real 8-bit image has a deterministic rounding error per grey level, which need not average out.
template <bool SPLIT>
__global__ __launch_bounds__(256) void stem_block_stats_kernel(StemBlockArgs a) {
    constexpr int UN = 2;
    __shared__ float red[4][2][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, kg = lane >> 5;
    h8_t wf[3], wlo[SPLIT ? 3 : 1];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
        wf[ty] = *(const h8_t*)(a.w + (long long)pl * 96 + ty * 32 + 8 * kg);
        if (SPLIT) wlo[ty] = *(const h8_t*)(a.wl + (long long)pl * 96 + ty * 32 + 8 * kg);
    }
    float s1 = 0.f, s2 = 0.f;
    const int nunits = (int)a.nunits;
    const int wstride = gridDim.x * 4 * UN;
    int u0 = (blockIdx.x * 4 + wave) * UN;
    UnitPos pos = unit_decode(u0 < nunits ? u0 : 0, a.Wb, a.H2);
    const UnitPos dstep = unit_decode(wstride - (UN - 1), a.Wb, a.H2), one = {0, 1, 0};
    const unsigned row_elems = (a.W + 2) * 4, lane_off = (pl + 2 * kg) * 4;
    for (; u0 < nunits; u0 += wstride) {
        h8_t xr[UN][4], xlo[SPLIT ? UN : 1][4];
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            const long long xoff = uniform_off(((long long)(pos.b * (a.H + 2) + 2 * pos.h) * (a.W + 2) + pos.w * 32) * 4);
            if (u0 + i < nunits) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    xr[i][rr] = *(const h8_t*)((const char*)(a.x + xoff) + (lane_off + rr * row_elems) * 2u);
                    if (SPLIT) xlo[i][rr] = *(const h8_t*)((const char*)(a.xl + xoff) + (lane_off + rr * row_elems) * 2u);
                }
            } else {        // (zero operands: y = 0 adds nothing to either sum)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    xr[i][rr] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
                    if (SPLIT) xlo[i][rr] = h8_t{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
            if (i + 1 < UN) unit_advance(pos, one, a.Wb, a.H2);
        }
        unit_advance(pos, dstep, a.Wb, a.H2);
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            f32x16_t acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
            if (SPLIT) {
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xlo[i][ty], wf[ty], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xlo[i][ty + 1], wf[ty], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty], wlo[ty], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty + 1], wlo[ty], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty], wf[ty], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[i][ty + 1], wf[ty], acc1, 0, 0, 0);
            }
            // the unit's 32 values of this lane first (short chains), then onto the running sums
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                u1 += acc0[r] + acc1[r];
                u2 = __builtin_fmaf(acc0[r], acc0[r], u2);
                u2 = __builtin_fmaf(acc1[r], acc1[r], u2);
            }
            s1 += u1, s2 += u2;
        }
    }
    red[wave][0][lane] = s1, red[wave][1][lane] = s2;
    __syncthreads();
    if (tid < 64) {           // fixed order: waves 0..3, pixel halves kg = 0, 1
        const int which = tid >> 5, ch = tid & 31;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) v += red[wv][which][ch] + red[wv][which][32 + ch];
        a.stats[((long long)blockIdx.x * 2 + which) * a.stats_ld + ch] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward: per unit recompute both conv rows in the same orientation, find the window argmax (first maximum in
// (h, w) scan order, torch's max_pool2d rule: row 0 wins a tie between the rows, column 0 inside a row), route
// G * leaky'(z) to it, and accumulate T = Gd^T V.  Because a lane's registers are pixels of ONE channel, the routed
// gradients are already the A operand of that product (row = channel, k = pixel (j & 3) + 8 (j >> 2) + 4 kg + 16 s):
// they never touch LDS; the B operand is the image window, staged from the registers and gathered with the
// transposing read as in wgrad_stem_kernel, with the row addresses following the same pixel order.
// Slab per workgroup: fp32 [32][96] (T in the stem K layout) + [32] (dbeta).
template <int OFF>
__device__ __forceinline__ void win_frags(unsigned b01, unsigned b2, Frag& fb, Frag& fc) {
    fb.lo = tr_read4<OFF>(b01), fb.hi = tr_read4<OFF + 8 * 8>(b01);
    fc.lo = tr_read4<OFF>(b2), fc.hi = tr_read4<OFF + 8 * 8>(b2);
}

__global__ __launch_bounds__(512) void stem_block_bwd_kernel(StemBlockArgs a) {
    constexpr int WIN = 4 * XROW + 128, NW = 8;
    __shared__ __attribute__((aligned(16))) char smem[NW * 16 * 64 * 4];      // windows: NW x WIN bytes; reduction: 32 KB
    static_assert(NW * WIN <= NW * 16 * 64 * 4, "window buffers");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, kg = lane >> 5;
    char* my = smem + wave * WIN;
    const unsigned my_addr = lds_addr_of(my);
    for (int i = lane; i < WIN / 4; i += 64) ((int*)my)[i] = 0;       // the fragment reads touch bytes past the staged pixels
    h8_t wf[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) wf[ty] = *(const h8_t*)(a.w + (long long)pl * 96 + ty * 32 + 8 * kg);
    const float sc = a.scale[pl], sh = a.shift[pl];
    float sb = 0.f;
    f32x16_t t01, t2;
#pragma unroll
    for (int r = 0; r < 16; ++r) t01[r] = t2[r] = 0.f;
    // transposing reads: lane 4 qq + p of a 16-lane group supplies the address of k-row qq at tap column p; the k-rows
    // of a half-wave are pixels 4 kg + qq (lo) and 8 + 4 kg + qq (hi), + 16 for the second half of the row
    const int g4 = lane >> 4, qq = (lane & 15) >> 2, p = lane & 3;
    const unsigned b01 = my_addr + (g4 & 1) * XROW + (4 * (g4 >> 1) + qq + p) * 8;
    const unsigned b2 = my_addr + 2 * XROW + (4 * (g4 >> 1) + qq + p) * 8;

    // one unit per iteration; the image rows and G of the NEXT unit are loaded before the current one is processed
    const int nunits = (int)a.nunits;
    const int wstride = gridDim.x * NW;
    const unsigned row_elems = (a.W + 2) * 4;
    h8_t xn[4];
    half_t gn[8];
    int u = blockIdx.x * NW + wave;
    UnitPos pos = unit_decode(u < nunits ? u : 0, a.Wb, a.H2);
    const UnitPos dstep = unit_decode(wstride, a.Wb, a.H2);
    const unsigned lane_off = (pl + 2 * kg) * 4;
    unsigned goff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) goff[q] = (pooled_of(q, kg) * a.g_ld + a.g_choff + pl) * 2u;   // bytes
    auto fetch = [&]() {     // wave-uniform bases (scalar registers) + 32-bit lane offsets
        const half_t* xb = a.x + uniform_off(((long long)(pos.b * (a.H + 2) + 2 * pos.h) * (a.W + 2) + pos.w * 32) * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) xn[rr] = *(const h8_t*)((const char*)xb + (lane_off + rr * row_elems) * 2u);
        const half_t* gb = a.g + uniform_off(((long long)(pos.b * a.H2 + pos.h) * a.W2 + pos.w * 16) * a.g_ld);
#pragma unroll
        for (int q = 0; q < 8; ++q) gn[q] = *(const half_t*)((const char*)gb + goff[q]);
    };
    if (u < nunits) fetch();
    for (; u < nunits; u += wstride) {
        h8_t xr[4];
        half_t gq[8];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) xr[rr] = xn[rr];
#pragma unroll
        for (int q = 0; q < 8; ++q) gq[q] = gn[q];
        unit_advance(pos, dstep, a.Wb, a.H2);
        if (u + wstride < nunits) fetch();

        f32x16_t acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[ty], wf[ty], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[ty + 1], wf[ty], acc1, 0, 0, 0);
        }
        stage_window<4>(my, xr, pl, kg, 0, 0, a.H, a.W, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        Frag fb[4], fc[4];
        win_frags<0>(b01, b2, fb[0], fc[0]);                     // conv row 0, pixels 0-15
        win_frags<16 * 8>(b01, b2, fb[1], fc[1]);                // conv row 0, pixels 16-31
        win_frags<XROW>(b01, b2, fb[2], fc[2]);                  // conv row 1
        win_frags<XROW + 16 * 8>(b01, b2, fb[3], fc[3]);
        // the routed gradients: word q of a row = (pixel 2q | pixel 2q + 1) as fp16, only the winner non-zero
        unsigned g0[8], g1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float z00 = __builtin_fmaf(acc0[2 * q], sc, sh), z01 = __builtin_fmaf(acc0[2 * q + 1], sc, sh);
            const float z10 = __builtin_fmaf(acc1[2 * q], sc, sh), z11 = __builtin_fmaf(acc1[2 * q + 1], sc, sh);
            const float m0 = fmaxf(z00, z01), m1 = fmaxf(z10, z11);
            const float best = fmaxf(m0, m1);
            const float gz = (float)gq[q] * (best > 0.f ? 1.f : a.slope);
            sb += gz;
            const unsigned gh = (unsigned)__builtin_bit_cast(unsigned short, (half_t)gz);
            const unsigned w0 = z00 >= z01 ? gh : gh << 16, w1 = z10 >= z11 ? gh : gh << 16;
            g0[q] = m0 >= m1 ? w0 : 0u;
            g1[q] = m0 >= m1 ? 0u : w1;
        }
        lds_wait_all(fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            tie(fb[s]), tie(fc[s]);
            const unsigned* gs = (s < 2 ? g0 : g1) + 4 * (s & 1);
            union { unsigned u[4]; h8_t h; } fa;
            fa.u[0] = gs[0], fa.u[1] = gs[1], fa.u[2] = gs[2], fa.u[3] = gs[3];
            t01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.h, fb[s].v(), t01, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.h, fc[s].v(), t2, 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = (float*)smem;
    float* out = a.slab + (long long)blockIdx.x * 3104;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = which == 0 ? t01[r] : t2[r];
        __syncthreads();
        for (int idx = tid; idx < 1024; idx += 512) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) v += red[wv * 1024 + idx];
            const int r = idx >> 6, ln = idx & 63;
            const int n = mfma32_row(r, ln), col = ln & 31;
            if (which == 0) out[n * 96 + (col >> 4) * 32 + (col & 15)] = v;
            else if (col < 16) out[n * 96 + 64 + col] = v;
        }
        __syncthreads();
    }
    // dbeta: lane (n, kg) summed G * leaky' over its pooled pixels of channel n
    red[wave * 64 + lane] = sb;
    __syncthreads();
    if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) v += red[wv * 64 + tid] + red[wv * 64 + 32 + tid];
        out[3072 + tid] = v;
    }
    // columns the accumulators do not cover (tx slots 4-7 of every filter row) are never read
}

// tsum (double [3104]) -> dW (OIHW, x mask), dgamma, dbeta; see the file header.
__global__ __launch_bounds__(1024) void stem_bwd_finish_kernel(const double* tsum, const double* ctx, const half_t* wp,
                                                               double count, const float* gamma, const float* save_mean,
                                                               const float* save_invstd, const float* mask, double inv_scale,
                                                               float* dw, float* dgamma, float* dbeta) {
    __shared__ double Wd[32][28], T[32][28], dB[32], dG[32];
    const int t = threadIdx.x;
    if (t < 864) {
        const int n = t / 27, k = t - n * 27;
        const int ty = k / 9, tx = (k / 3) % 3, c = k % 3;
        Wd[n][k] = (double)(float)wp[n * 96 + ty * 32 + tx * 4 + c];
        T[n][k] = tsum[n * 96 + ty * 32 + tx * 4 + c] * inv_scale;
    }
    if (t < 32) dB[t] = tsum[3072 + t] * inv_scale;
    __syncthreads();
    if (t < 32) {
        double acc = 0.0;
        for (int k = 0; k < 27; ++k) acc += Wd[t][k] * T[t][k];
        dG[t] = (double)save_invstd[t] * (acc - (double)save_mean[t] * dB[t]);
        if (dbeta) dbeta[t] = (float)dB[t];
        if (dgamma) dgamma[t] = (float)dG[t];
    }
    __syncthreads();
    if (t < 864) {
        const int n = t / 27, k = t - n * 27;
        const int ty = k / 9, tx = (k / 3) % 3, c = k % 3;
        const double istd = (double)save_invstd[n], mu = (double)save_mean[n];
        const double S = ctx[k], WC = ctx[32 + n * 27 + k];
        double v = (double)gamma[n] * istd * (T[n][k] - dB[n] / count * S - dG[n] / count * istd * (WC - mu * S));
        const int dst = ((n * 3 + c) * 3 + ty) * 3 + tx;
        if (mask) v *= (double)mask[dst];
        dw[dst] = (float)v;
    }
}

struct Carve {      // workspace layout (bytes), all offsets multiples of 256
    size_t gram_slab, gram_sum, ctx, bwd_slab, bwd_sum, total;
};
constexpr int kGramWgs = 512, kBwdWgs = 256;   // slab capacity; the launches use one 8-wave workgroup per CU
Carve carve() {
    Carve c;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    c.gram_slab = take((size_t)kGramWgs * 2304 * sizeof(float));
    c.gram_sum = take(2304 * sizeof(double));
    c.ctx = take((32 + 32 * 27) * sizeof(double));
    c.bwd_slab = take((size_t)kBwdWgs * 3104 * sizeof(float));
    c.bwd_sum = take(3104 * sizeof(double));
    c.total = off;
    return c;
}

int check_desc(const mcamd_stem_block_desc* d, const char* what) {
    MCAMD_REQUIRE(d, "%s: null descriptor", what);
    MCAMD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "%s: non-positive dimension", what);
    MCAMD_REQUIRE(d->W % 32 == 0 && d->H % 2 == 0, "%s: needs W %% 32 == 0 and an even H (got %d x %d)", what, d->H, d->W);
    MCAMD_REQUIRE((long long)d->B * d->H * d->W < (1ll << 31), "%s: more than 2^31 output pixels", what);
    MCAMD_REQUIRE(d->x && d->wp && d->scale && d->shift, "%s: null argument", what);
    MCAMD_REQUIRE(d->cout == 0 || d->cout == 32 || (d->cout > 0 && d->cout < 32 && d->cout % 8 == 0 && !d->training),
                  "%s: cout %d (32; 8 / 16 / 24 for the inference-mode forward pass only)", what, d->cout);
    return MCAMD_OK;
}

void fill_args(StemBlockArgs& a, const mcamd_stem_block_desc* d) {
    memset(&a, 0, sizeof(a));
    a.x = (const half_t*)d->x;
    a.w = (const half_t*)d->wp;
    a.scale = d->scale, a.shift = d->shift;
    a.B = d->B, a.H = d->H, a.W = d->W, a.H2 = d->H / 2, a.W2 = d->W / 2, a.Wb = d->W / 32;
    a.slope = d->slope;
    a.cout = d->cout > 0 ? d->cout : 32;
    a.xl = (const half_t*)d->x_lo;
    a.wl = (const half_t*)d->wp_lo;
}

// fp32 NCHW image -> two padded NHWC4 fp16 images: hi = fp16(v), lo = fp16(v - hi) (channel 3 zero in both).
// Consecutive threads take consecutive pixels: three coalesced plane reads, one 8-byte store per image.
__global__ __launch_bounds__(256) void nhwc4_split_kernel(const float* src, int B, int H, int W, half_t* hi, half_t* lo) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
        const int b = (int)(pix / HW), rem = (int)(pix - (long long)b * HW);
        const int h = rem / W, w = rem - h * W;
        const float* sp = src + (long long)b * 3 * HW + rem;
        h4_t vh, vl;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = fminf(fmaxf(sp[(long long)c * HW], -65504.f), 65504.f);
            vh[c] = (half_t)v;
            vl[c] = (half_t)(v - (float)vh[c]);
        }
        vh[3] = vl[3] = (half_t)0.f;
        const long long o = (((long long)b * (H + 2) + h + 1) * (W + 2) + w + 1) * 4;
        *(h4_t*)(hi + o) = vh;
        *(h4_t*)(lo + o) = vl;
    }
}

int stats_grid(const mcamd_stem_block_desc* d) {
    const long long nunits = (long long)d->B * (d->H / 2) * (d->W / 32);
    const long long want = (nunits + 7) / 8;      // >= one pass of 2 units per wave
    return (int)(want < 1024 ? (want < 1 ? 1 : want) : 1024);
}

}  // namespace

extern "C" size_t mcamd_stem_block_workspace_bytes(void) { return carve().total; }

extern "C" int mcamd_stem_block_fwd(const mcamd_stem_block_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "stem_block_fwd: null descriptor");
        const mcamd_stem_block_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_stem_block_fwd(&d_, workspace, workspace_bytes, s); });
    }
    if (check_desc(d, "stem_block_fwd")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(d->planes >= 0 && d->planes <= 3, "stem_block_fwd: planes must be 1, 2 or 3 (got %d)", d->planes);
    const int span = d->planes >= 2 ? 32 * d->planes : 32;
    const bool split = d->x_lo != nullptr || d->wp_lo != nullptr;
    MCAMD_REQUIRE(!split || (d->x_lo && d->wp_lo), "stem_block_fwd: split operands need both x_lo and wp_lo");
    MCAMD_REQUIRE(!split || (!d->training && (d->cout == 0 || d->cout == 32)),
                  "stem_block_fwd: split operands take scale / shift from mcamd_stem_block_stats + mcamd_bn_coeffs (training must "
                  "be 0) and 32 filters");
    // dst == NULL with training != 0: statistics only (Gram sums, coefficients and the context the backward pass reads)
    MCAMD_REQUIRE(d->dst || d->training, "stem_block_fwd: null dst (allowed with training != 0 only: statistics, no output)");
    MCAMD_REQUIRE(!d->dst || (d->dst_ld % 8 == 0 && d->dst_choff % 8 == 0 && d->dst_choff + span <= d->dst_ld),
                  "stem_block_fwd: output slice [%d, %d) does not fit dst_ld %d", d->dst_choff, d->dst_choff + span, d->dst_ld);
    const Carve c = carve();
    hipStream_t st = (hipStream_t)stream;
    StemBlockArgs a;
    fill_args(a, d);
    if (d->training) {
        MCAMD_REQUIRE(workspace && d->gamma && d->beta && d->save_mean && d->save_invstd, "stem_block_fwd: null argument (training)");
        if (workspace_bytes < c.total) {
            mcamd_set_error("stem_block_fwd: workspace %zu < %zu bytes", workspace_bytes, c.total);
            return MCAMD_EWORKSPACE;
        }
        char* ws = (char*)workspace;
        StemBlockArgs g = a;
        g.nunits = (long long)d->B * d->H * g.Wb;
        g.slab = (float*)(ws + c.gram_slab);
        long long want = (g.nunits + 31) / 32;    // >= one batch of 4 steps per wave
        const int gmax = kGramWgs;   // 2 workgroups per CU: 223 vs 239 us
        const int grid = (int)(want < gmax ? (want < 1 ? 1 : want) : gmax);
        hipLaunchKernelGGL(stem_gram_kernel, dim3(grid), dim3(512), 0, st, g);
        MCAMD_LAUNCH_CHECK("stem_gram");
        hipLaunchKernelGGL(slab_sum_kernel, dim3(2304 / 16), dim3(256), 0, st, (const float*)g.slab, grid, 2304,
                           (double*)(ws + c.gram_sum));
        hipLaunchKernelGGL(stem_coeffs_kernel, dim3(1), dim3(1024), 0, st, (const double*)(ws + c.gram_sum), a.w,
                           (double)d->B * d->H * d->W, d->gamma, d->beta, d->running_mean, d->running_var, d->momentum, d->eps,
                           d->scale, d->shift, d->save_mean, d->save_invstd, (double*)(ws + c.ctx));
        MCAMD_LAUNCH_CHECK("stem_coeffs");
    }
    if (!d->dst) return MCAMD_OK;
    a.out = (half_t*)d->dst;
    a.out_ld = d->dst_ld, a.out_choff = d->dst_choff;
    a.nunits = (long long)d->B * a.H2 * a.Wb;
    long long want = (a.nunits + 7) / 8;          // >= one pass of 2 units per wave
    const int grid = (int)(want < 2048 ? (want < 1 ? 1 : want) : 2048);
    if (split) {
        if (d->planes == 3) hipLaunchKernelGGL((stem_block_fwd_kernel<2, 3, true>), dim3(grid), dim3(256), 0, st, a);
        else if (d->planes == 2) hipLaunchKernelGGL((stem_block_fwd_kernel<2, 2, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((stem_block_fwd_kernel<2, 1, true>), dim3(grid), dim3(256), 0, st, a);
    } else if (d->planes == 3) hipLaunchKernelGGL((stem_block_fwd_kernel<2, 3>), dim3(grid), dim3(256), 0, st, a);
    else if (d->planes == 2) hipLaunchKernelGGL((stem_block_fwd_kernel<2, 2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((stem_block_fwd_kernel<4, 1>), dim3(grid), dim3(256), 0, st, a);   // 4 units in flight per wave (1 / 2 / 4: 66 / 62 / 60 us)
    MCAMD_LAUNCH_CHECK("stem_block_fwd");
    return MCAMD_OK;
}

extern "C" int mcamd_nchw_f32_to_nhwc4_split(const float* src, int32_t B, int32_t H, int32_t W, void* hi, void* lo, void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_nchw_f32_to_nhwc4_split(src, B, H, W, hi, lo, s); });
    MCAMD_REQUIRE(src && hi && lo && B > 0 && H > 0 && W > 0, "nchw_to_nhwc4_split: bad argument");
    MCAMD_REQUIRE((long long)B * H * W < (1ll << 31), "nchw_to_nhwc4_split: more than 2^31 pixels");
    const long long total = (long long)B * H * W;
    long long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(nhwc4_split_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, B, H, W, (half_t*)hi, (half_t*)lo);
    MCAMD_LAUNCH_CHECK("nchw_to_nhwc4_split");
    return MCAMD_OK;
}

extern "C" int32_t mcamd_stem_block_stats_rows(const mcamd_stem_block_desc* d) {
    if (!d || d->B <= 0 || d->H <= 0 || d->W <= 0) return 0;
    return stats_grid(d);
}

extern "C" int mcamd_stem_block_stats(const mcamd_stem_block_desc* d, float* stats, int32_t stats_rows, int32_t stats_ld,
                                      void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "stem_block_stats: null descriptor");
        const mcamd_stem_block_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_stem_block_stats(&d_, stats, stats_rows, stats_ld, s); });
    }
    MCAMD_REQUIRE(d, "stem_block_stats: null descriptor");
    MCAMD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0, "stem_block_stats: non-positive dimension");
    MCAMD_REQUIRE(d->W % 32 == 0 && d->H % 2 == 0, "stem_block_stats: needs W %% 32 == 0 and an even H (got %d x %d)", d->H, d->W);
    MCAMD_REQUIRE((long long)d->B * d->H * d->W < (1ll << 31), "stem_block_stats: more than 2^31 output pixels");
    MCAMD_REQUIRE(d->x && d->wp && stats, "stem_block_stats: null argument");
    MCAMD_REQUIRE(d->cout == 0 || d->cout == 32, "stem_block_stats: 32 filters (got %d)", d->cout);
    const bool split = d->x_lo != nullptr || d->wp_lo != nullptr;
    MCAMD_REQUIRE(!split || (d->x_lo && d->wp_lo), "stem_block_stats: split operands need both x_lo and wp_lo");
    const int grid = stats_grid(d);
    MCAMD_REQUIRE(stats_rows == grid && stats_ld >= 32, "stem_block_stats: slab must be [%d][2][>= 32] (got %d rows, ld %d)",
                  grid, stats_rows, stats_ld);
    StemBlockArgs a;
    fill_args(a, d);
    a.nunits = (long long)d->B * a.H2 * a.Wb;
    a.stats = stats, a.stats_ld = stats_ld;
    hipStream_t st = (hipStream_t)stream;
    if (split) hipLaunchKernelGGL((stem_block_stats_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((stem_block_stats_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    MCAMD_LAUNCH_CHECK("stem_block_stats");
    return MCAMD_OK;
}

extern "C" int mcamd_stem_block_bwd(const mcamd_stem_block_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "stem_block_bwd: null descriptor");
        const mcamd_stem_block_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_stem_block_bwd(&d_, workspace, workspace_bytes, s); });
    }
    if (check_desc(d, "stem_block_bwd")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(workspace && d->g && d->dw && d->gamma && d->save_mean && d->save_invstd, "stem_block_bwd: null argument");
    MCAMD_REQUIRE(d->cout == 0 || d->cout == 32, "stem_block_bwd: training needs 32 filters (got %d)", d->cout);
    MCAMD_REQUIRE(d->g_ld % 8 == 0 && d->g_choff % 8 == 0 && d->g_choff + 32 <= d->g_ld,
                  "stem_block_bwd: gradient slice [%d, %d) does not fit g_ld %d", d->g_choff, d->g_choff + 32, d->g_ld);
    MCAMD_REQUIRE(d->grad_scale > 0.f, "stem_block_bwd: grad_scale must be positive");
    const Carve c = carve();
    if (workspace_bytes < c.total) {
        mcamd_set_error("stem_block_bwd: workspace %zu < %zu bytes", workspace_bytes, c.total);
        return MCAMD_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    StemBlockArgs a;
    fill_args(a, d);
    a.g = (const half_t*)d->g;
    a.g_ld = d->g_ld, a.g_choff = d->g_choff;
    a.nunits = (long long)d->B * a.H2 * a.Wb;
    a.slab = (float*)(ws + c.bwd_slab);
    long long want = (a.nunits + 31) / 32;        // >= 4 units per wave
    const int grid = (int)(want < kBwdWgs ? (want < 1 ? 1 : want) : kBwdWgs);
    hipLaunchKernelGGL(stem_block_bwd_kernel, dim3(grid), dim3(512), 0, st, a);
    MCAMD_LAUNCH_CHECK("stem_block_bwd");
    hipLaunchKernelGGL(slab_sum_kernel, dim3(3104 / 16), dim3(256), 0, st, (const float*)a.slab, grid, 3104,
                       (double*)(ws + c.bwd_sum));
    hipLaunchKernelGGL(stem_bwd_finish_kernel, dim3(1), dim3(1024), 0, st, (const double*)(ws + c.bwd_sum),
                       (const double*)(ws + c.ctx), a.w, (double)d->B * d->H * d->W, d->gamma, d->save_mean, d->save_invstd,
                       d->mask, 1.0 / (double)d->grad_scale, d->dw, d->dgamma, d->dbeta);
    MCAMD_LAUNCH_CHECK("stem_bwd_finish");
    return MCAMD_OK;
}
