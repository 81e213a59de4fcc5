// BatchNorm2d + LeakyReLU + MaxPool2d(2,2) / Reorg(2) / route, fused around the conv kernels.
// HBM-bound streaming kernels: 16 bytes (8 fp16 channels) per lane, channel-fastest thread
// mapping so every wave touches whole contiguous pixel rows.
//
// Reference ops replaced: nn.BatchNorm2d (nets.py:802, torch defaults eps 1e-5 / momentum 0.1),
// nn.LeakyReLU(0.1) (nets.py:809), nn.MaxPool2d(2,2) (nets.py:821), Reorg (nets.py:648-667),
// torch.cat route (nets.py:738-746), and their autograd backward.
#include "common.h"

// ------------------------------------------------------------------------------------
// batch statistics -> affine coefficients
// ------------------------------------------------------------------------------------
// Slab reductions.  One block per 8 channels (the early layers have 32-64 channels and up to 2048 slab rows: a
// block per 32 channels was one or two CUs reading 0.5 MB each, 25-29 us), 128 row groups x 8 channels per block:
// rows are summed in double per thread, then over the 8 row groups of a wave by shuffles, then over the 16 waves
// through LDS -- a fixed order, so the result does not depend on scheduling.
constexpr int RED_CPB = 8, RED_RG = 128;

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

// returns (for threads of wave 0 .. any) the block total in threads with ry == 0: call from all 1024 threads
__device__ __forceinline__ void block_reduce2(double& a, double& b, double (*red)[16][RED_CPB]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cx = threadIdx.x & (RED_CPB - 1);
#pragma unroll
    for (int m = RED_CPB; m < 64; m <<= 1) {   // lanes cx, cx+8, ..., cx+56 hold the same channel
        a += shfl_xor_f64(a, m);
        b += shfl_xor_f64(b, m);
    }
    if (lane < RED_CPB) {
        red[0][wave][cx] = a;
        red[1][wave][cx] = b;
    }
    __syncthreads();
    if (threadIdx.x < RED_CPB) {
        a = b = 0.0;
        for (int w = 0; w < 16; ++w) {
            a += red[0][w][cx];
            b += red[1][w][cx];
        }
    }
}

__global__ __launch_bounds__(1024) void bn_coeffs_kernel(const float* stats, int rows, int ld, int C, double count,
                                                         const float* gamma, const float* beta, float* rmean,
                                                         float* rvar, float momentum, float eps, int training,
                                                         float* scale, float* shift, float* save_mean,
                                                         float* save_invstd, const int* perm, int ones_channel) {
    __shared__ double red[2][16][RED_CPB];
    const int cx = threadIdx.x & (RED_CPB - 1), ry = threadIdx.x / RED_CPB;
    const int c = blockIdx.x * RED_CPB + cx;
    double s1 = 0.0, s2 = 0.0;
    if (training && c < C) {
        for (int p = ry; p < rows; p += RED_RG) {
            s1 += (double)stats[((long long)p * 2 + 0) * ld + c];
            s2 += (double)stats[((long long)p * 2 + 1) * ld + c];
        }
    }
    block_reduce2(s1, s2, red);
    if (ry != 0 || c >= C) return;
    // `perm` (optional): the statistics / coefficient vectors are in the kernels' PHYSICAL channel order
    // (kept filters first, engine.py filter compaction); the module's parameter vectors are in the
    // reference's order.  pc = this physical channel's index into gamma/beta/running_*.
    const int pc = perm ? perm[c] : c;
    double mean, var;
    if (training) {
        mean = s1 / count;
        var = s2 / count - mean * mean;  // biased variance (normalisation)
        if (var < 0.0) var = 0.0;
        double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[pc] = (float)((1.0 - momentum) * (double)rmean[pc] + (double)momentum * mean);
        rvar[pc] = (float)((1.0 - momentum) * (double)rvar[pc] + (double)momentum * unbiased);
    } else {
        mean = rmean[pc];
        var = rvar[pc];
    }
    double invstd = 1.0 / sqrt(var + (double)eps);
    float sc = (float)((double)gamma[pc] * invstd);
    scale[c] = sc;
    shift[c] = (float)((double)beta[pc] - mean * (double)sc);
    if (c == ones_channel) scale[c] = 0.f, shift[c] = 1.f;      // the activation pass writes ones here (fold.hip)
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = (float)invstd;
}

// ------------------------------------------------------------------------------------
// forward: y(raw fp16) -> leaky(y*scale+shift) -> {plain | 2x2 max pool | reorg} -> padded NHWC
// ------------------------------------------------------------------------------------
struct ActArgs {
    const void* y;       // fp16 or fp32 (template parameter Y32) raw conv output
    const float* scale;
    const float* shift;
    half_t* dst;
    half_t* dst2;
    int B, H, W, C;
    int y_ld, y_choff, dst_ld, dst_choff, dst2_ld, dst2_choff;
    float slope;
    long long items;
    const float* border;  // optional [16][C]: added to the raw conv output by border class (slim models)
    int dst_plane, dst2_plane;  // plane strides of the split (hi | lo | hi) storage, PL == 3
    int dst_pw, dst2_pw;        // halo pixels per row / rows per image of dst, dst2: 2 (padded form) or 1 (shared-halo form)
    int dst2_pl;                // storage form of dst2 (mcamd_act_desc.planes2)
    half_t* pool_act;           // optional (pool): full-resolution fp16 activation for the backward pass (mcamd_act_desc.pool_act)
    int pool_act_ld, pool_act_pw;
};

__device__ __forceinline__ void load8(const half_t* p, float* v) {
    h8_t h = *(const h8_t*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
}
__device__ __forceinline__ half_t sat_half(float v) {  // saturate instead of overflowing to inf (inf * 0 = NaN downstream)
    return (half_t)fminf(fmaxf(v, -65504.f), 65504.f);
}
__device__ __forceinline__ void store8(half_t* p, const float* v) {
    h8_t h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = sat_half(v[i]);
    *(h8_t*)p = h;
}
__device__ __forceinline__ void loadf8(const float* p, float* v) {
    f32x4_t a = *(const f32x4_t*)p, b = *(const f32x4_t*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] = a[i];
        v[4 + i] = b[i];
    }
}
// raw conv output, fp16 or fp32 storage
template <bool Y32>
__device__ __forceinline__ void load_y(const void* y, long long idx, float* v) {
    if (Y32) loadf8((const float*)y + idx, v);
    else load8((const half_t*)y + idx, v);
}
// activation store: one fp16 plane, or the split form of the "fp16x3" precision mode (include/mcamd.h,
// mcamd_act_desc.planes): hi = fp16(v), lo = fp16(v - hi), and hi again, `plane` channels apart
// PL == 4 (round 4): hi = fp16(v) and, `plane` fp16 units further, the e4m3 correction bytes of the consumer with
// mcamd_conv_geom.x_f8: [lo8 = e4m3((v - hi) * 2^12) : plane bytes | x8 = e4m3(v * 2) : plane bytes] (common.h); `ci` =
// channel of v[0] in the buffer (the consumer's slice starts at channel 0: a concat member writes at its offset inside the
// e4m3 strings too).  The conversion instruction returns NaN above 448: clamped first.
template <int PL>
__device__ __forceinline__ void store_act(half_t* p, int plane, const float* v, int ci = 0) {
    h8_t hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = sat_half(v[i]);
    *(h8_t*)p = hi;
    if (PL == 4) {
        float ql[8], q8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ql[i] = fminf(fmaxf((v[i] - (float)hi[i]) * (float)(1 << MCAMD_F8_SXL), -448.f), 448.f);
            q8[i] = fminf(fmaxf(v[i] * (float)(1 << MCAMD_F8_SX8), -448.f), 448.f);
        }
        int wl[2], w8[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wl[i] = __builtin_amdgcn_cvt_pk_fp8_f32(ql[4 * i], ql[4 * i + 1], 0, false);
            wl[i] = __builtin_amdgcn_cvt_pk_fp8_f32(ql[4 * i + 2], ql[4 * i + 3], wl[i], true);
            w8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(q8[4 * i], q8[4 * i + 1], 0, false);
            w8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(q8[4 * i + 2], q8[4 * i + 3], w8[i], true);
        }
        char* b = (char*)(p - ci + plane) + ci;      // byte ci of the e4m3 region
        typedef int i32x2_t __attribute__((ext_vector_type(2)));
        *(i32x2_t*)b = i32x2_t{wl[0], wl[1]};
        *(i32x2_t*)(b + plane) = i32x2_t{w8[0], w8[1]};
    } else if (PL >= 2) {
        h8_t lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) lo[i] = (half_t)(v[i] - (float)hi[i]);   // exact difference, one rounding; |lo| <= ulp(hi)/2
        *(h8_t*)(p + plane) = lo;
        if (PL == 3) *(h8_t*)(p + 2 * plane) = hi;      // (PL == 2: the consumer wraps its third K part onto the hi plane)
    }
}
// the largest fp16 value below h (bit pattern; h finite, not the most negative value)
__device__ __forceinline__ unsigned short half_prev_bits(unsigned short b) {
    if (b == 0x0000 || b == 0x8000) return 0x8001;          // +-0 -> the smallest negative subnormal
    return (b & 0x8000) ? (unsigned short)(b + 1) : (unsigned short)(b - 1);
}
// dst2 may be stored in another form than dst (its consumer is another convolution): wave-uniform switch
template <int PL>
__device__ __forceinline__ void store_act2(int pl2, half_t* p, int plane, const float* v, int ci) {
    if (PL < 2 || pl2 == PL) store_act<PL>(p, plane, v, ci);
    else if (pl2 == 4) store_act<4>(p, plane, v, ci);
    else if (pl2 == 3) store_act<3>(p, plane, v, ci);
    else if (pl2 == 2) store_act<2>(p, plane, v, ci);
    else store_act<1>(p, plane, v, ci);
}
// offset of pixel (b, h, w) from the buffer pointer; pw = 2: padded NHWC, pw = 1: shared-halo form (include/mcamd.h)
__device__ __forceinline__ long long pad_off(int b, int h, int w, int H, int W, int ld, int pw = 2) {
    return (((long long)b * (H + pw) + h + 1) * (W + pw) + w + 1) * ld;
}

// Border class of pixel (h, w): which 3x3 taps fall into the zero padding (bit 0 top, 1 bottom, 2 left, 3 right).
__device__ __forceinline__ int border_class(int h, int w, int H, int W) {
    return (h == 0 ? 1 : 0) | (h == H - 1 ? 2 : 0) | (w == 0 ? 4 : 0) | (w == W - 1 ? 8 : 0);
}

// FIXED: C/8 divides 256, so a thread keeps one channel group for the whole grid-stride loop and its BN
// coefficients stay in registers.  !FIXED: any C % 8 == 0 (slim models), coefficients re-read per item (L1/L2 hits).
template <int MODE, bool FIXED, bool Y32, int PL>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(ActArgs a) {
    const int CH = a.C >> 3;
    int c8 = (threadIdx.x % CH) * 8;
    float sc[8], sh[8];
    if (FIXED) {
        loadf8(a.scale + c8, sc);
        loadf8(a.shift + c8, sh);
    }
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    for (long long item = (long long)blockIdx.x * 256 + threadIdx.x; item < a.items; item += (long long)gridDim.x * 256) {
        long long pix = item / CH;
        if (!FIXED) {
            c8 = (int)(item - pix * CH) * 8;
            loadf8(a.scale + c8, sc);
            loadf8(a.shift + c8, sh);
        }
        if (MODE == MCAMD_DST_PLAIN) {
            int b = (int)(pix / (a.H * a.W));
            int rem = (int)(pix - (long long)b * a.H * a.W);
            int h = rem / a.W, w = rem - h * a.W;
            float v[8];
            load_y<Y32>(a.y, pix * a.y_ld + a.y_choff + c8, v);
            if (a.border) {
                float bb[8];
                loadf8(a.border + border_class(h, w, a.H, a.W) * a.C + c8, bb);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bb[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float z = v[i] * sc[i] + sh[i];
                v[i] = z > 0.f ? z : z * a.slope;
            }
            store_act<PL>(a.dst + pad_off(b, h, w, a.H, a.W, a.dst_ld, a.dst_pw) + a.dst_choff + c8, a.dst_plane, v, a.dst_choff + c8);
        } else {
            int b = (int)(pix / (Ho * Wo));
            int rem = (int)(pix - (long long)b * Ho * Wo);
            int ho = rem / Wo, wo = rem - ho * Wo;
            float act[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int h = 2 * ho + (k >> 1), w = 2 * wo + (k & 1);
                long long sp = ((long long)b * a.H + h) * a.W + w;
                load_y<Y32>(a.y, sp * a.y_ld + a.y_choff + c8, act[k]);
                if (a.border) {
                    float bb[8];
                    loadf8(a.border + border_class(h, w, a.H, a.W) * a.C + c8, bb);
#pragma unroll
                    for (int i = 0; i < 8; ++i) act[k][i] += bb[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float z = act[k][i] * sc[i] + sh[i];
                    act[k][i] = z > 0.f ? z : z * a.slope;
                }
                if (a.dst2) store_act2<PL>(a.dst2_pl, a.dst2 + pad_off(b, h, w, a.H, a.W, a.dst2_ld, a.dst2_pw) + a.dst2_choff + c8, a.dst2_plane, act[k], a.dst2_choff + c8);
            }
            long long dp = pad_off(b, ho, wo, Ho, Wo, a.dst_ld, a.dst_pw) + a.dst_choff;
            if (MODE == MCAMD_DST_POOL) {
                float m[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) m[i] = fmaxf(fmaxf(act[0][i], act[1][i]), fmaxf(act[2][i], act[3][i]));
                store_act<PL>(a.dst + dp + c8, a.dst_plane, m, a.dst_choff + c8);
                if (a.pool_act) {
                    // the four activations rounded to fp16, with the window's argmax (first maximum of the UNROUNDED values,
                    // what was just pooled) as their strict maximum: a neighbour that rounds to the same fp16 value is stored
                    // one step lower, so the backward pass finds the argmax the forward pass took from the stored values alone
                    typedef unsigned short us8_t __attribute__((ext_vector_type(8)));
                    us8_t hb[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        int arg = 0;
                        float best = act[0][i];
#pragma unroll
                        for (int k = 1; k < 4; ++k) {
                            const bool gt = act[k][i] > best;
                            best = gt ? act[k][i] : best, arg = gt ? k : arg;
                        }
                        const unsigned short top = __builtin_bit_cast(unsigned short, sat_half(best));
                        const unsigned short below = half_prev_bits(top);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned short hk = __builtin_bit_cast(unsigned short, sat_half(act[k][i]));
                            hb[k][i] = (k != arg && hk == top) ? below : hk;
                        }
                    }
                    half_t* q0 = a.pool_act + pad_off(b, 2 * ho, 2 * wo, a.H, a.W, a.pool_act_ld, a.pool_act_pw) + c8;
                    const long long qrow = (long long)(a.W + a.pool_act_pw) * a.pool_act_ld;
                    *(us8_t*)q0 = hb[0];
                    *(us8_t*)(q0 + a.pool_act_ld) = hb[1];
                    *(us8_t*)(q0 + qrow) = hb[2];
                    *(us8_t*)(q0 + qrow + a.pool_act_ld) = hb[3];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) store_act<PL>(a.dst + dp + k * a.C + c8, a.dst_plane, act[k], a.dst_choff + k * a.C + c8);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------
struct ActBwdArgs {
    const void* y;       // fp16 or fp32 (template parameter Y32) saved raw conv output
    const float* scale;
    const float* shift;
    const float* mean;
    const float* invstd;
    const half_t* g;
    const half_t* g2;
    half_t* dy;
    float* slab;         // [nblocks][2][C]
    const float* coef;   // [2][C]: c1 = sum(gz)/count, c2 = sum(gz*xhat)/count
    const float* dy_keep;  // optional [C]: 0 for fully pruned filters (their dY is not needed and is zeroed)
    int* overflow;         // optional: set to 1 when a dY value was clamped to +-65504
    int B, H, W, C;
    int y_ld, y_choff, g_ld, g_choff, g2_ld, g2_choff, dy_ld, dy_choff;
    float slope;
    long long items;
    int dy_pw;             // 2: dy in the padded form, 1: shared-halo form
    const half_t* act;     // optional (bn_plain_bwd_act_kernel): the block's stored activation, padded NHWC fp16
    int act_ld, act_choff, act_pw;
};

// PHASE 0: per-channel sums of g_z and g_z*xhat -> slab.  PHASE 1: dy -> padded NHWC.
// Pass 0 of the BatchNorm backward kernels: every thread holds partial sums sb[8] / sg[8] of its 8 channels; threads t and
// t + CH (+ 2 CH ...) hold the same channel group.  The block's sums go to slab rows (2 block, 2 block + 1) in a fixed
// order (deterministic).  LDS scratch [16 values][256 threads] with a row pitch of 264 floats: writes are lane-consecutive
// and the reads of 64 consecutive channels (8 groups x 8 values) hit 64 distinct banks ((8 value + group) mod 64).  The
// layout of rounds 1-2, red[thread][16], put the 64 lanes of a store on 4 banks: SQ_LDS_BANK_CONFLICT 443 M cycles per 16
// launches of bn_plain_bwd_kernel<0>, the largest count in the profile (profiles/r02c_pmc_per_kernel.json); the slab
// stores were 4-byte pieces at a 32-byte stride and are whole lines now.
__device__ __forceinline__ void block_partials_to_slab(const float (&sb)[8], const float (&sg)[8], int CH, float* slab, int C) {
    constexpr int PITCH = 264;
    __shared__ float red[16 * PITCH];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[i * PITCH + threadIdx.x] = sb[i];
        red[(8 + i) * PITCH + threadIdx.x] = sg[i];
    }
    __syncthreads();
    const int reps = 256 / CH;
    for (int o = threadIdx.x; o < 16 * CH; o += 256) {
        const int which = o / (8 * CH), c = o - which * 8 * CH;      // c = channel inside the block's C = 8 CH channels
        const int ch = c >> 3, v = which * 8 + (c & 7);
        float s = 0.f;
        for (int k = 0; k < reps; ++k) s += red[v * PITCH + k * CH + ch];
        slab[((long long)blockIdx.x * 2 + which) * C + c] = s;
    }
}

template <int MODE, int PHASE, bool Y32>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(ActBwdArgs a) {
    constexpr int NP = MODE == MCAMD_DST_PLAIN ? 1 : 4;
    const int CH = a.C >> 3;
    const int c8 = (threadIdx.x % CH) * 8;
    float sc[8], sh[8], mu[8], is[8], c1[8], c2[8], dm[8];
    loadf8(a.scale + c8, sc);
    loadf8(a.shift + c8, sh);
    loadf8(a.mean + c8, mu);
    loadf8(a.invstd + c8, is);
    if (PHASE == 1) {
        loadf8(a.coef + c8, c1);
        loadf8(a.coef + a.C + c8, c2);
        if (a.dy_keep) {
            float kp[8];
            loadf8(a.dy_keep + c8, kp);
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = kp[i] != 0.f ? sc[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = sc[i];
        }
    }
    float sb[8], sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = sg[i] = 0.f;
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    bool sat = false;

    for (long long item = (long long)blockIdx.x * 256 + threadIdx.x; item < a.items; item += (long long)gridDim.x * 256) {
        long long pix = item / CH;
        int b, hh[NP], ww[NP];
        long long gp;  // pixel index into g
        if (MODE == MCAMD_DST_PLAIN) {
            b = (int)(pix / (a.H * a.W));
            int rem = (int)(pix - (long long)b * a.H * a.W);
            hh[0] = rem / a.W;
            ww[0] = rem - hh[0] * a.W;
            gp = pix;
        } else {
            b = (int)(pix / (Ho * Wo));
            int rem = (int)(pix - (long long)b * Ho * Wo);
            int ho = rem / Wo, wo = rem - ho * Wo;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                hh[k] = 2 * ho + (k >> 1);
                ww[k] = 2 * wo + (k & 1);
            }
            gp = pix;
        }
        float yv[NP][8], zz[NP][8];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            long long sp = ((long long)b * a.H + hh[k]) * a.W + ww[k];
            load_y<Y32>(a.y, sp * a.y_ld + a.y_choff + c8, yv[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) zz[k][i] = yv[k][i] * sc[i] + sh[i];
        }
        float ga[NP][8];
        if (MODE == MCAMD_DST_PLAIN) {
            load8(a.g + gp * a.g_ld + a.g_choff + c8, ga[0]);
        } else if (MODE == MCAMD_DST_POOL) {
            float gv[8];
            load8(a.g + gp * a.g_ld + a.g_choff + c8, gv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // argmax over the window of the activation AS STORED by the forward pass (fp16; unrounded with the
                // split hi | lo storage that goes with an fp32 y), first maximum in (h, w) scan order -- torch's
                // max_pool2d tie rule on those values
                auto stored = [&](float z) {
                    const float av = z > 0.f ? z : z * a.slope;
                    return Y32 ? av : (float)(half_t)av;
                };
                float best = stored(zz[0][i]);
                int arg = 0;
#pragma unroll
                for (int k = 1; k < NP; ++k) {
                    float av = stored(zz[k][i]);
                    if (av > best) {
                        best = av;
                        arg = k;
                    }
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) ga[k][i] = (k == arg) ? gv[i] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) load8(a.g + gp * a.g_ld + a.g_choff + k * a.C + c8, ga[k]);
        }
        if (a.g2) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                long long sp = ((long long)b * a.H + hh[k]) * a.W + ww[k];
                float t[8];
                load8(a.g2 + sp * a.g2_ld + a.g2_choff + c8, t);
#pragma unroll
                for (int i = 0; i < 8; ++i) ga[k][i] += t[i];
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            float out[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float gz = zz[k][i] > 0.f ? ga[k][i] : ga[k][i] * a.slope;
                float xh = (yv[k][i] - mu[i]) * is[i];
                if (PHASE == 0) {
                    sb[i] += gz;
                    sg[i] += gz * xh;
                } else {
                    out[i] = dm[i] * (gz - c1[i] - xh * c2[i]);
                }
            }
            if (PHASE == 1) {
                store8(a.dy + pad_off(b, hh[k], ww[k], a.H, a.W, a.dy_ld, a.dy_pw) + a.dy_choff + c8, out);
#pragma unroll
                for (int i = 0; i < 8; ++i) sat |= fabsf(out[i]) > 65504.f;
            }
        }
    }
    if (PHASE == 1 && sat && a.overflow) atomicOr(a.overflow, 1);

    if (PHASE == 0) {
        block_partials_to_slab(sb, sg, CH, a.slab, a.C);
    }
}

// PLAIN blocks (16 of YOLOv2's 21 BatchNorm blocks) without a second gradient: the same hoisted form as bn_pool_bwd_kernel
// below, and no position arithmetic beyond one incremental (b, h, w) for the padded dY address.  The generic kernel's two
// 64-bit divisions per 48-byte item made its loop 450-640 instructions long; on the 13x13 layers (5 items per thread,
// 4 waves per SIMD) that is ~17 us of pure instruction issue for a pass that moves 44 MB -- 14.7 us measured against
// 9 us for the forward pass over the same bytes.
template <int PHASE, bool Y32>
__global__ __launch_bounds__(256) void bn_plain_bwd_kernel(ActBwdArgs a) {
    const int CH = a.C >> 3, lg = __ffs(CH) - 1;        // C / 8 is a power of two (check_c)
    const int c8 = (threadIdx.x & (CH - 1)) * 8;
    float sc[8], sh[8], mu[8], is[8], A[8], Bc[8], dm[8];
    loadf8(a.scale + c8, sc);
    loadf8(a.shift + c8, sh);
    loadf8(a.mean + c8, mu);
    loadf8(a.invstd + c8, is);
    if (PHASE == 1) {
        float c1[8], c2[8];
        loadf8(a.coef + c8, c1);
        loadf8(a.coef + a.C + c8, c2);
        if (a.dy_keep) {
            float kp[8];
            loadf8(a.dy_keep + c8, kp);
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = kp[i] != 0.f ? sc[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = sc[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Bc[i] = dm[i] * c2[i] * is[i];
            A[i] = dm[i] * c1[i] - Bc[i] * mu[i];
        }
    }
    float sb[8], sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = sg[i] = 0.f;
    float satmax = 0.f;
    const int HW = a.H * a.W;
    const unsigned npix = (unsigned)(a.items >> lg);
    const unsigned stride = (gridDim.x * 256u) >> lg;                      // pixels per grid stride
    unsigned pix = (blockIdx.x * 256u + threadIdx.x) >> lg;
    int b = (int)(pix / (unsigned)HW), rem = (int)(pix - (unsigned)b * (unsigned)HW);
    int h = rem / a.W, w = rem - h * a.W;
    const int sb_ = (int)(stride / (unsigned)HW), srem = (int)(stride - (unsigned)sb_ * (unsigned)HW);
    const int sh_ = srem / a.W, sw_ = srem - sh_ * a.W;
    for (; pix < npix; pix += stride) {
        float yv[8], gv[8];
        load_y<Y32>(a.y, (long long)pix * a.y_ld + a.y_choff + c8, yv);
        load8(a.g + (long long)pix * a.g_ld + a.g_choff + c8, gv);
        float out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float z = yv[i] * sc[i] + sh[i];
            const float gz = z > 0.f ? gv[i] : gv[i] * a.slope;
            if (PHASE == 0) {
                sb[i] += gz;
                sg[i] += gz * ((yv[i] - mu[i]) * is[i]);
            } else {
                const float o = dm[i] * gz - (Bc[i] * yv[i] + A[i]);
                out[i] = o;
                satmax = fmaxf(satmax, fabsf(o));
            }
        }
        if (PHASE == 1) {
            store8(a.dy + pad_off(b, h, w, a.H, a.W, a.dy_ld, a.dy_pw) + a.dy_choff + c8, out);
            w += sw_;
            if (w >= a.W) w -= a.W, ++h;
            h += sh_;
            if (h >= a.H) h -= a.H, ++b;
            b += sb_;
        }
    }
    if (PHASE == 1 && satmax > 65504.f && a.overflow) atomicOr(a.overflow, 1);
    if (PHASE == 0) {
        block_partials_to_slab(sb, sg, CH, a.slab, a.C);
    }
}

// The same two passes WITHOUT the saved raw output: LeakyReLU is invertible, so a PLAIN block's pre-activation is
// recovered from the activation the forward pass stored for the consumer (fp16, the hi plane of split storage):
// z = a > 0 ? a : a / slope, xhat = (z - beta) / gamma.  With the split-operand precisions the saved y is fp32: the two
// passes read 2 instead of 4 bytes per element for it (0.89 -> 0.65 ms per "mixed" B=64 step), at the operand precision
// every backward pass has anyway (G and dY are fp16).  With dm = gamma invstd (0 for a pruned filter):
//     out = dm g_z - (P z + Q),   P = dm c2 / gamma,  Q = dm c1 - P beta     (per channel, hoisted)
// A channel with gamma == 0 has no xhat to recover (its dY is 0 either way, dm = 0): the threads that hold such a channel
// read the saved fp32 y for their dgamma sums in pass 0, as the y kernel does -- a branch nobody takes on a trained network.
template <int PHASE>
__global__ __launch_bounds__(256) void bn_plain_bwd_act_kernel(ActBwdArgs a) {
    const int CH = a.C >> 3, lg = __ffs(CH) - 1;        // C / 8 is a power of two (check_c)
    const int c8 = (threadIdx.x & (CH - 1)) * 8;
    float sc[8], sh[8], mu[8], is[8], beta[8], rg[8], P[8], Q[8], dm[8];
    loadf8(a.scale + c8, sc);
    loadf8(a.shift + c8, sh);
    loadf8(a.mean + c8, mu);
    loadf8(a.invstd + c8, is);
    bool need_y = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        beta[i] = sh[i] + mu[i] * sc[i];                 // shift = beta - mean scale
        rg[i] = sc[i] != 0.f ? is[i] / sc[i] : 0.f;      // 1 / gamma (scale = gamma invstd)
        need_y |= sc[i] == 0.f;
    }
    need_y = need_y && PHASE == 0 && a.y != nullptr;
    if (PHASE == 1) {
        float c1[8], c2[8];
        loadf8(a.coef + c8, c1);
        loadf8(a.coef + a.C + c8, c2);
        if (a.dy_keep) {
            float kp[8];
            loadf8(a.dy_keep + c8, kp);
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = kp[i] != 0.f ? sc[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = sc[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            P[i] = dm[i] * c2[i] * rg[i];
            Q[i] = dm[i] * c1[i] - P[i] * beta[i];
        }
    }
    float sb[8], sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = sg[i] = 0.f;
    float satmax = 0.f;
    const float inv_slope = 1.0f / a.slope;
    const int HW = a.H * a.W;
    const unsigned npix = (unsigned)(a.items >> lg);
    const unsigned stride = (gridDim.x * 256u) >> lg;                      // pixels per grid stride
    unsigned pix = (blockIdx.x * 256u + threadIdx.x) >> lg;
    int b = (int)(pix / (unsigned)HW), rem = (int)(pix - (unsigned)b * (unsigned)HW);
    int h = rem / a.W, w = rem - h * a.W;
    const int sb_ = (int)(stride / (unsigned)HW), srem = (int)(stride - (unsigned)sb_ * (unsigned)HW);
    const int sh_ = srem / a.W, sw_ = srem - sh_ * a.W;
    for (; pix < npix; pix += stride) {
        float av[8], gv[8];
        load8(a.act + pad_off(b, h, w, a.H, a.W, a.act_ld, a.act_pw) + a.act_choff + c8, av);
        load8(a.g + (long long)pix * a.g_ld + a.g_choff + c8, gv);
        float out[8], yv[8];
        if (need_y) load_y<true>(a.y, (long long)pix * a.y_ld + a.y_choff + c8, yv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool pos = av[i] > 0.f;
            const float z = pos ? av[i] : av[i] * inv_slope;
            const float gz = pos ? gv[i] : gv[i] * a.slope;
            if (PHASE == 0) {
                sb[i] += gz;
                float xh = (z - beta[i]) * rg[i];
                if (need_y && sc[i] == 0.f) xh = (yv[i] - mu[i]) * is[i];
                sg[i] += gz * xh;
            } else {
                const float o = dm[i] * gz - (P[i] * z + Q[i]);
                out[i] = o;
                satmax = fmaxf(satmax, fabsf(o));
            }
        }
        if (PHASE == 1) store8(a.dy + pad_off(b, h, w, a.H, a.W, a.dy_ld, a.dy_pw) + a.dy_choff + c8, out);
        w += sw_;
        if (w >= a.W) w -= a.W, ++h;
        h += sh_;
        if (h >= a.H) h -= a.H, ++b;
        b += sb_;
    }
    if (PHASE == 1 && satmax > 65504.f && a.overflow) atomicOr(a.overflow, 1);
    if (PHASE == 0) {
        block_partials_to_slab(sb, sg, CH, a.slab, a.C);
    }
}

// MaxPool blocks (conv2 / conv5 / conv8 / conv13 of YOLOv2: 0.7 of the 1.5 ms of BatchNorm backward).
// The generic kernel above spends ~1 300 instructions per item there (SQ_ACTIVE_INST_ANY 0.62 of the wave cycles, 3.6 TB/s:
// it is issue-bound, not HBM-bound) because it treats the four window pixels as four full backward elements.  Only the
// argmax pixel receives a gradient, so with  out = dm (g_z - c1 - xhat c2),  xhat = (y - mu) is:
//     out_k = [k == arg] dm g_z  -  (Bc y_k + A),      Bc = dm c2 is,  A = dm c1 - Bc mu     (per channel, hoisted)
// i.e. one multiply-add per non-argmax element; the sums of pass 0 need the argmax element only.  The argmax itself is
// taken exactly as the generic kernel takes it (first maximum of the activations AS STORED by the forward pass, compared
// as fp16 values here instead of converting them back).  32-bit positions stepped incrementally (no division in the loop).
// G2: a second gradient at full resolution (the route that reads conv13 beside its pool) is added to every window pixel,
// so every pixel has a g_z -- still without the generic kernel's per-pixel xhat / select chains.
template <int PHASE, bool Y32, bool G2>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(ActBwdArgs a) {
    const int CH = a.C >> 3, lg = __ffs(CH) - 1;        // C / 8 is a power of two (check_c)
    const int c8 = (threadIdx.x & (CH - 1)) * 8;
    float sc[8], sh[8], mu[8], is[8], A[8], Bc[8], dm[8];
    loadf8(a.scale + c8, sc);
    loadf8(a.shift + c8, sh);
    loadf8(a.mean + c8, mu);
    loadf8(a.invstd + c8, is);
    if (PHASE == 1) {
        float c1[8], c2[8];
        loadf8(a.coef + c8, c1);
        loadf8(a.coef + a.C + c8, c2);
        if (a.dy_keep) {
            float kp[8];
            loadf8(a.dy_keep + c8, kp);
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = kp[i] != 0.f ? sc[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = sc[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Bc[i] = dm[i] * c2[i] * is[i];
            A[i] = dm[i] * c1[i] - Bc[i] * mu[i];
        }
    }
    float sb[8], sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = sg[i] = 0.f;
    float satmax = 0.f;
    const int Ho = a.H >> 1, Wo = a.W >> 1, HoWo = Ho * Wo;
    const unsigned npix = (unsigned)(a.items >> lg);                       // pooled pixels
    const unsigned stride = (gridDim.x * 256u) >> lg;                      // pooled pixels per grid stride
    unsigned pix = (blockIdx.x * 256u + threadIdx.x) >> lg;
    // (b, ho, wo) of `pix` and of the stride, advanced with carries
    int b = (int)(pix / (unsigned)HoWo), rem = (int)(pix - (unsigned)b * (unsigned)HoWo);
    int ho = rem / Wo, wo = rem - ho * Wo;
    const int sb_ = (int)(stride / (unsigned)HoWo), srem = (int)(stride - (unsigned)sb_ * (unsigned)HoWo);
    const int sho = srem / Wo, swo = srem - sho * Wo;
    const long long yrow = (long long)a.W * a.y_ld;                         // elements per image row of y
    for (; pix < npix; pix += stride) {
        const half_t* yp = (const half_t*)a.y;
        const long long y0 = (((long long)b * a.H + 2 * ho) * a.W + 2 * wo) * a.y_ld + a.y_choff + c8;
        float yv[4][8], gv[8];
        if (Y32) {
            const float* yf = (const float*)a.y;
            loadf8(yf + y0, yv[0]);
            loadf8(yf + y0 + a.y_ld, yv[1]);
            loadf8(yf + y0 + yrow, yv[2]);
            loadf8(yf + y0 + yrow + a.y_ld, yv[3]);
        } else {
            load8(yp + y0, yv[0]);
            load8(yp + y0 + a.y_ld, yv[1]);
            load8(yp + y0 + yrow, yv[2]);
            load8(yp + y0 + yrow + a.y_ld, yv[3]);
        }
        load8(a.g + (long long)pix * a.g_ld + a.g_choff + c8, gv);
        float g2v[G2 ? 4 : 1][8];
        if (G2) {
            const half_t* q0 = a.g2 + (((long long)b * a.H + 2 * ho) * a.W + 2 * wo) * a.g2_ld + a.g2_choff + c8;
            const long long g2row = (long long)a.W * a.g2_ld;
            load8(q0, g2v[0]);
            load8(q0 + a.g2_ld, g2v[G2 ? 1 : 0]);
            load8(q0 + g2row, g2v[G2 ? 2 : 0]);
            load8(q0 + g2row + a.g2_ld, g2v[G2 ? 3 : 0]);
        }
        float out[4][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = yv[k][i] * sc[i] + sh[i];
            // first maximum in (h, w) scan order of the activations as the forward pass stored them
            int arg = 0;
            float zs = z[0], ys = yv[0][i];
            if (Y32) {
                float best = z[0] > 0.f ? z[0] : z[0] * a.slope;
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const float av = z[k] > 0.f ? z[k] : z[k] * a.slope;
                    const bool gt = av > best;
                    best = gt ? av : best, arg = gt ? k : arg, zs = gt ? z[k] : zs, ys = gt ? yv[k][i] : ys;
                }
            } else {
                half_t best = (half_t)(z[0] > 0.f ? z[0] : z[0] * a.slope);
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const half_t av = (half_t)(z[k] > 0.f ? z[k] : z[k] * a.slope);
                    const bool gt = av > best;
                    best = gt ? av : best, arg = gt ? k : arg, zs = gt ? z[k] : zs, ys = gt ? yv[k][i] : ys;
                }
            }
            if (G2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ga = (k == arg ? gv[i] : 0.f) + g2v[G2 ? k : 0][i];
                    const float gzk = z[k] > 0.f ? ga : ga * a.slope;
                    if (PHASE == 0) {
                        sb[i] += gzk;
                        sg[i] += gzk * ((yv[k][i] - mu[i]) * is[i]);
                    } else {
                        const float o = dm[i] * gzk - (Bc[i] * yv[k][i] + A[i]);
                        out[k][i] = o;
                        satmax = fmaxf(satmax, fabsf(o));
                    }
                }
            } else {
                const float gz = zs > 0.f ? gv[i] : gv[i] * a.slope;
                if (PHASE == 0) {
                    sb[i] += gz;
                    sg[i] += gz * ((ys - mu[i]) * is[i]);
                } else {
                    const float t = dm[i] * gz;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float o = (k == arg ? t : 0.f) - (Bc[i] * yv[k][i] + A[i]);
                        out[k][i] = o;
                        satmax = fmaxf(satmax, fabsf(o));
                    }
                }
            }
        }
        if (PHASE == 1) {
            half_t* d0 = a.dy + pad_off(b, 2 * ho, 2 * wo, a.H, a.W, a.dy_ld, a.dy_pw) + a.dy_choff + c8;
            const long long drow = (long long)(a.W + a.dy_pw) * a.dy_ld;
            store8(d0, out[0]);
            store8(d0 + a.dy_ld, out[1]);
            store8(d0 + drow, out[2]);
            store8(d0 + drow + a.dy_ld, out[3]);
        }
        wo += swo;
        if (wo >= Wo) wo -= Wo, ++ho;
        ho += sho;
        if (ho >= Ho) ho -= Ho, ++b;
        b += sb_;
    }
    if (PHASE == 1 && satmax > 65504.f && a.overflow) atomicOr(a.overflow, 1);
    if (PHASE == 0) {
        block_partials_to_slab(sb, sg, CH, a.slab, a.C);
    }
}

// MaxPool blocks WITHOUT the saved raw output (second session of round 4): as bn_plain_bwd_act_kernel recovers a PLAIN block's
// pre-activation from the stored activation, a MaxPool block's comes from a FULL-RESOLUTION fp16 copy of its activation
// (the forward pass writes one beside the pooled output: mcamd_act_desc.pool_act).  z = a > 0 ? a : a / slope, xhat =
// (z - beta) / gamma; the window's argmax is the maximum of the four STORED values -- the forward pass stores them so that
// the element it pooled (first maximum of the unrounded activations) is their strict maximum -- and the LeakyReLU side of
// every element is the sign of its stored value (rounding keeps the sign).  Split-operand engines save y as fp32: the two passes read 2 instead of 4 bytes per
// element (conv2 / 5 / 8 / 13 at B = 64: 0.66 GB less per step) for 2 bytes more written by the forward pass where the copy
// did not exist.  out_k = [k == arg] dm g_z - (P z_k + Q) with P, Q of bn_plain_bwd_act_kernel; gamma == 0 channels read the
// saved y for their dgamma sums (pass 0 only), as there.
template <int PHASE, bool G2>
__global__ __launch_bounds__(256) void bn_pool_bwd_act_kernel(ActBwdArgs a) {
    const int CH = a.C >> 3, lg = __ffs(CH) - 1;        // C / 8 is a power of two (check_c)
    const int c8 = (threadIdx.x & (CH - 1)) * 8;
    float sc[8], sh[8], mu[8], is[8], beta[8], rg[8], P[8], Q[8], dm[8];
    loadf8(a.scale + c8, sc);
    loadf8(a.shift + c8, sh);
    loadf8(a.mean + c8, mu);
    loadf8(a.invstd + c8, is);
    bool need_y = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        beta[i] = sh[i] + mu[i] * sc[i];                 // shift = beta - mean scale
        rg[i] = sc[i] != 0.f ? is[i] / sc[i] : 0.f;      // 1 / gamma (scale = gamma invstd)
        need_y |= sc[i] == 0.f;
    }
    need_y = need_y && PHASE == 0 && a.y != nullptr;
    if (PHASE == 1) {
        float c1[8], c2[8];
        loadf8(a.coef + c8, c1);
        loadf8(a.coef + a.C + c8, c2);
        if (a.dy_keep) {
            float kp[8];
            loadf8(a.dy_keep + c8, kp);
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = kp[i] != 0.f ? sc[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dm[i] = sc[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            P[i] = dm[i] * c2[i] * rg[i];
            Q[i] = dm[i] * c1[i] - P[i] * beta[i];
        }
    }
    float sb[8], sg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sb[i] = sg[i] = 0.f;
    float satmax = 0.f;
    const float inv_slope = 1.0f / a.slope;
    const int Ho = a.H >> 1, Wo = a.W >> 1, HoWo = Ho * Wo;
    const unsigned npix = (unsigned)(a.items >> lg);                       // pooled pixels
    const unsigned stride = (gridDim.x * 256u) >> lg;                      // pooled pixels per grid stride
    unsigned pix = (blockIdx.x * 256u + threadIdx.x) >> lg;
    int b = (int)(pix / (unsigned)HoWo), rem = (int)(pix - (unsigned)b * (unsigned)HoWo);
    int ho = rem / Wo, wo = rem - ho * Wo;
    const int sb_ = (int)(stride / (unsigned)HoWo), srem = (int)(stride - (unsigned)sb_ * (unsigned)HoWo);
    const int sho = srem / Wo, swo = srem - sho * Wo;
    const long long arow = (long long)(a.W + a.act_pw) * a.act_ld;          // elements per padded image row of the activation
    const long long yrow = (long long)a.W * a.y_ld;
    for (; pix < npix; pix += stride) {
        const half_t* ap = a.act + pad_off(b, 2 * ho, 2 * wo, a.H, a.W, a.act_ld, a.act_pw) + a.act_choff + c8;
        float av[4][8], gv[8];
        load8(ap, av[0]);
        load8(ap + a.act_ld, av[1]);
        load8(ap + arow, av[2]);
        load8(ap + arow + a.act_ld, av[3]);
        load8(a.g + (long long)pix * a.g_ld + a.g_choff + c8, gv);
        float g2v[G2 ? 4 : 1][8];
        if (G2) {
            const half_t* q0 = a.g2 + (((long long)b * a.H + 2 * ho) * a.W + 2 * wo) * a.g2_ld + a.g2_choff + c8;
            const long long g2row = (long long)a.W * a.g2_ld;
            load8(q0, g2v[0]);
            load8(q0 + a.g2_ld, g2v[G2 ? 1 : 0]);
            load8(q0 + g2row, g2v[G2 ? 2 : 0]);
            load8(q0 + g2row + a.g2_ld, g2v[G2 ? 3 : 0]);
        }
        float yv[4][8];
        if (need_y) {
            const float* yf = (const float*)a.y;
            const long long y0 = (((long long)b * a.H + 2 * ho) * a.W + 2 * wo) * a.y_ld + a.y_choff + c8;
            loadf8(yf + y0, yv[0]);
            loadf8(yf + y0 + a.y_ld, yv[1]);
            loadf8(yf + y0 + yrow, yv[2]);
            loadf8(yf + y0 + yrow + a.y_ld, yv[3]);
        }
        float out[4][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = av[k][i] > 0.f ? av[k][i] : av[k][i] * inv_slope;
            // first maximum in (h, w) scan order of the activations as the forward pass stored them
            int arg = 0;
            float best = av[0][i];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const bool gt = av[k][i] > best;
                best = gt ? av[k][i] : best, arg = gt ? k : arg;
            }
            if (G2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ga = (k == arg ? gv[i] : 0.f) + g2v[G2 ? k : 0][i];
                    const float gzk = av[k][i] > 0.f ? ga : ga * a.slope;
                    if (PHASE == 0) {
                        sb[i] += gzk;
                        float xh = (z[k] - beta[i]) * rg[i];
                        if (need_y && sc[i] == 0.f) xh = (yv[k][i] - mu[i]) * is[i];
                        sg[i] += gzk * xh;
                    } else {
                        const float o = dm[i] * gzk - (P[i] * z[k] + Q[i]);
                        out[k][i] = o;
                        satmax = fmaxf(satmax, fabsf(o));
                    }
                }
            } else {
                const float gz = best > 0.f ? gv[i] : gv[i] * a.slope;
                if (PHASE == 0) {
                    float zs = z[0], ys = need_y ? yv[0][i] : 0.f;
#pragma unroll
                    for (int k = 1; k < 4; ++k) zs = k == arg ? z[k] : zs, ys = (need_y && k == arg) ? yv[k][i] : ys;
                    sb[i] += gz;
                    float xh = (zs - beta[i]) * rg[i];
                    if (need_y && sc[i] == 0.f) xh = (ys - mu[i]) * is[i];
                    sg[i] += gz * xh;
                } else {
                    const float t = dm[i] * gz;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float o = (k == arg ? t : 0.f) - (P[i] * z[k] + Q[i]);
                        out[k][i] = o;
                        satmax = fmaxf(satmax, fabsf(o));
                    }
                }
            }
        }
        if (PHASE == 1) {
            half_t* d0 = a.dy + pad_off(b, 2 * ho, 2 * wo, a.H, a.W, a.dy_ld, a.dy_pw) + a.dy_choff + c8;
            const long long drow = (long long)(a.W + a.dy_pw) * a.dy_ld;
            store8(d0, out[0]);
            store8(d0 + a.dy_ld, out[1]);
            store8(d0 + drow, out[2]);
            store8(d0 + drow + a.dy_ld, out[3]);
        }
        wo += swo;
        if (wo >= Wo) wo -= Wo, ++ho;
        ho += sho;
        if (ho >= Ho) ho -= Ho, ++b;
        b += sb_;
    }
    if (PHASE == 1 && satmax > 65504.f && a.overflow) atomicOr(a.overflow, 1);
    if (PHASE == 0) {
        block_partials_to_slab(sb, sg, CH, a.slab, a.C);
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* slab, int nblocks, int C, double count,
                                                               float inv_scale, float* dgamma, float* dbeta,
                                                               float* coef, const int* perm, int skip_from) {
    __shared__ double red[2][16][RED_CPB];
    const int cx = threadIdx.x & (RED_CPB - 1), ry = threadIdx.x / RED_CPB;
    const int c = blockIdx.x * RED_CPB + cx;
    double sb = 0.0, sg = 0.0;
    if (c < C) {
        for (int p = ry; p < nblocks; p += RED_RG) {
            sb += (double)slab[((long long)p * 2 + 0) * C + c];
            sg += (double)slab[((long long)p * 2 + 1) * C + c];
        }
    }
    block_reduce2(sb, sg, red);
    if (ry != 0 || c >= C) return;
    const int pc = perm ? perm[c] : c;   // parameter-order index of this physical channel
    const bool skip = skip_from > 0 && c >= skip_from;   // a folded dead channel: its consumer delivers these gradients (fold.hip)
    if (dbeta && !skip) dbeta[pc] = (float)(sb * inv_scale);
    if (dgamma && !skip) dgamma[pc] = (float)(sg * inv_scale);
    coef[c] = (float)(sb / count);
    coef[C + c] = (float)(sg / count);
}

// ------------------------------------------------------------------------------------
// NCHW fp32 -> padded NHWC fp16 (model boundary)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* src, int B, int C, int H, int W, float mul,
                                                           half_t* dst, int ld, int choff, int cgroups, int* overflow,
                                                           int plane, int pw) {
    bool sat = false;
    // item = (pixel, channel group of up to 8); consecutive threads -> consecutive pixels (coalesced NCHW reads)
    const long long HW = (long long)H * W;
    const long long items = (long long)B * HW * cgroups;
    for (long long item = (long long)blockIdx.x * 256 + threadIdx.x; item < items; item += (long long)gridDim.x * 256) {
        long long pg = item / ((long long)B * HW);  // channel group slowest
        long long pix = item - pg * (long long)B * HW;
        int b = (int)(pix / HW);
        int rem = (int)(pix - (long long)b * HW);
        int h = rem / W, w = rem - h * W;
        int c0 = (int)pg * 8;
        half_t* d = dst + pad_off(b, h, w, H, W, ld, pw) + choff + c0;
        const float* s = src + ((long long)b * C + c0) * HW + rem;
        int nc = C - c0 < 8 ? C - c0 : 8;
        if (overflow)
            for (int i = 0; i < nc; ++i) sat |= fabsf(s[i * HW] * mul) > 65504.f;
        if (plane > 0) {  // split storage: hi | lo | hi planes, `plane` channels apart (mcamd_act_desc.planes)
            for (int i = 0; i < nc; ++i) {
                const float v = s[i * HW] * mul;
                const half_t hi = sat_half(v);
                d[i] = hi;
                d[plane + i] = (half_t)(v - (float)hi);
                d[2 * plane + i] = hi;
            }
        } else if (ld == 4) {  // stem image: 3 channels + one zero, one 8-byte store per pixel
            h4_t q;
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = (i < nc) ? sat_half(s[i * HW] * mul) : (half_t)0.f;
            *(h4_t*)d = q;
        } else if (nc == 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = s[i * HW] * mul;
            store8(d, v);
        } else {
            for (int i = 0; i < nc; ++i) d[i] = sat_half(s[i * HW] * mul);
        }
    }
    if (sat && overflow) atomicOr(overflow, 1);
}

// ------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------
static int stream_grid(long long items) {
    long long g = (items + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int mcamd_bn_coeffs_ex(const float* stats, int32_t stats_rows, int32_t stats_ld, int32_t C, int64_t count,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  float momentum, float eps, int32_t training, float* scale, float* shift,
                                  float* save_mean, float* save_invstd, const int32_t* chan_perm, int32_t ones_channel,
                                  void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) {
            return mcamd_bn_coeffs_ex(stats, stats_rows, stats_ld, C, count, gamma, beta, running_mean, running_var, momentum, eps,
                                      training, scale, shift, save_mean, save_invstd, chan_perm, ones_channel, s);
        });
    MCAMD_REQUIRE(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "bn_coeffs: null argument");
    MCAMD_REQUIRE(!training || (stats && stats_rows > 0 && stats_ld >= C && count > 0), "bn_coeffs: bad statistics slab");
    MCAMD_REQUIRE(ones_channel < C, "bn_coeffs: ones_channel %d outside the %d channels", ones_channel, C);
    hipLaunchKernelGGL(bn_coeffs_kernel, dim3((C + RED_CPB - 1) / RED_CPB), dim3(1024), 0, (hipStream_t)stream, stats, stats_rows,
                       stats_ld, C, (double)count, gamma, beta, running_mean, running_var, momentum, eps, training,
                       scale, shift, save_mean, save_invstd, (const int*)chan_perm, ones_channel < 0 ? -1 : ones_channel);
    MCAMD_LAUNCH_CHECK("bn_coeffs");
    return MCAMD_OK;
}

extern "C" int mcamd_bn_coeffs(const float* stats, int32_t stats_rows, int32_t stats_ld, int32_t C, int64_t count,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int32_t training, float* scale, float* shift,
                               float* save_mean, float* save_invstd, const int32_t* chan_perm, void* stream) {
    return mcamd_bn_coeffs_ex(stats, stats_rows, stats_ld, C, count, gamma, beta, running_mean, running_var, momentum, eps,
                              training, scale, shift, save_mean, save_invstd, chan_perm, -1, stream);
}

static int check_c(int C, const char* what) {
    int CH = C / 8;
    if (C <= 0 || C % 8 != 0 || CH > 256 || 256 % CH != 0) {
        mcamd_set_error("%s: channel count %d must be 8 * (a power of two <= 256)", what, C);
        return MCAMD_EINVAL;
    }
    return MCAMD_OK;
}

extern "C" int mcamd_bn_act_fwd(const mcamd_act_desc* d, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "bn_act_fwd: null descriptor");
        const mcamd_act_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_bn_act_fwd(&d_, s); });
    }
    MCAMD_REQUIRE(d && d->y && d->dst && d->scale && d->shift, "bn_act_fwd: null argument");
    MCAMD_REQUIRE(d->C > 0 && d->C % 8 == 0, "bn_act_fwd: channel count %d must be a positive multiple of 8", d->C);
    MCAMD_REQUIRE(d->y_ld % 8 == 0 && d->y_choff % 8 == 0 && d->dst_ld % 8 == 0 && d->dst_choff % 8 == 0 &&
                      d->dst2_ld % 8 == 0 && d->dst2_choff % 8 == 0,
                  "bn_act_fwd: leading dimensions / channel offsets must be multiples of 8");
    MCAMD_REQUIRE(d->mode == MCAMD_DST_PLAIN || (d->H % 2 == 0 && d->W % 2 == 0), "bn_act_fwd: pool/reorg need even H, W");
    MCAMD_REQUIRE(d->mode != MCAMD_DST_PLAIN || !d->dst2, "bn_act_fwd: dst2 only with pool/reorg");
    MCAMD_REQUIRE(d->y_dtype == 0 || d->y_dtype == 1, "bn_act_fwd: y_dtype %d (0 = fp16, 1 = fp32)", d->y_dtype);
    const int planes = d->planes == 0 ? 1 : d->planes;
    const int planes2 = d->planes2 == 0 ? planes : d->planes2;
    MCAMD_REQUIRE(planes >= 1 && planes <= 4 && planes2 >= 1 && planes2 <= 4 && (planes == 1) == (planes2 == 1),
                  "bn_act_fwd: planes / planes2 must be 1, or both of 2, 3, 4 (got %d, %d)", d->planes, d->planes2);
    if (planes >= 2) {
        const int span = d->mode == MCAMD_DST_REORG ? 4 * d->C : d->C;
        // (4: the e4m3 region is one plane stride of fp16 units, like a lo plane)
        MCAMD_REQUIRE(d->dst_plane % 8 == 0 && d->dst_plane >= span &&
                          d->dst_choff + (planes == 4 ? 2 * d->dst_plane : (planes - 1) * d->dst_plane + span) <= d->dst_ld,
                      "bn_act_fwd: %d planes of stride %d (+ offset %d, %d channels) do not fit dst_ld %d", planes, d->dst_plane,
                      d->dst_choff, span, d->dst_ld);
        MCAMD_REQUIRE(!d->dst2 || (d->dst2_plane % 8 == 0 && d->dst2_plane >= d->C &&
                                   d->dst2_choff + (planes2 == 4 ? 2 * d->dst2_plane : (planes2 - 1) * d->dst2_plane + d->C) <= d->dst2_ld),
                      "bn_act_fwd: %d planes of stride %d do not fit dst2_ld %d", planes2, d->dst2_plane, d->dst2_ld);
    }
    ActArgs a;
    a.dst2_pl = planes2;
    a.dst_plane = d->dst_plane, a.dst2_plane = d->dst2_plane;
    MCAMD_REQUIRE((d->dst_pad == 0 || d->dst_pad == 1) && (d->dst2_pad == 0 || d->dst2_pad == 1), "bn_act_fwd: dst_pad / dst2_pad must be 0 or 1");
    a.dst_pw = d->dst_pad ? 1 : 2, a.dst2_pw = d->dst2_pad ? 1 : 2;
    a.y = d->y;
    a.scale = d->scale;
    a.shift = d->shift;
    a.dst = (half_t*)d->dst;
    a.dst2 = (half_t*)d->dst2;
    a.B = d->B, a.H = d->H, a.W = d->W, a.C = d->C;
    a.y_ld = d->y_ld, a.y_choff = d->y_choff, a.dst_ld = d->dst_ld, a.dst_choff = d->dst_choff;
    a.dst2_ld = d->dst2_ld, a.dst2_choff = d->dst2_choff;
    a.slope = d->slope;
    long long pixels = (long long)d->B * d->H * d->W;
    if (d->mode != MCAMD_DST_PLAIN) pixels /= 4;
    a.items = pixels * (d->C / 8);
    int grid = stream_grid(a.items);
    hipStream_t st = (hipStream_t)stream;
    a.border = d->border;
    MCAMD_REQUIRE(!d->pool_act || (d->mode == MCAMD_DST_POOL && d->pool_act_ld % 8 == 0 && d->pool_act_ld >= d->C &&
                                   (d->pool_act_pad == 0 || d->pool_act_pad == 1)),
                  "bn_act_fwd: pool_act goes with mode pool, pool_act_ld (%d) a multiple of 8 >= C, pool_act_pad 0 or 1", d->pool_act_ld);
    a.pool_act = (half_t*)d->pool_act;
    a.pool_act_ld = d->pool_act_ld, a.pool_act_pw = d->pool_act_pad ? 1 : 2;
    const int CH = d->C / 8;
    const bool fixed = CH <= 256 && 256 % CH == 0;
    const bool y32 = d->y_dtype == 1;
    MCAMD_REQUIRE(planes != 4 || y32, "bn_act_fwd: planes 4 (e4m3 corrections) goes with an fp32 y");
#define ACT_INST(MODE_, FIXED_)                                                                                         \
    do {                                                                                                                \
        if (y32 && planes == 4) hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, true, 4>), dim3(grid), dim3(256), 0, st, a);       \
        else if (y32 && planes == 3) hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, true, 3>), dim3(grid), dim3(256), 0, st, a);  \
        else if (y32 && planes == 2) hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, true, 2>), dim3(grid), dim3(256), 0, st, a);  \
        else if (y32) hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, true, 1>), dim3(grid), dim3(256), 0, st, a);                 \
        else if (planes == 3) hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, false, 3>), dim3(grid), dim3(256), 0, st, a);        \
        else if (planes == 2) hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, false, 2>), dim3(grid), dim3(256), 0, st, a);        \
        else hipLaunchKernelGGL((bn_act_fwd_kernel<MODE_, FIXED_, false, 1>), dim3(grid), dim3(256), 0, st, a);                         \
    } while (0)
#define ACT_CASE(MODE_)                   \
    do {                                  \
        if (fixed) ACT_INST(MODE_, true); \
        else ACT_INST(MODE_, false);      \
    } while (0)
    if (d->mode == MCAMD_DST_PLAIN) ACT_CASE(MCAMD_DST_PLAIN);
    else if (d->mode == MCAMD_DST_POOL) ACT_CASE(MCAMD_DST_POOL);
    else if (d->mode == MCAMD_DST_REORG) ACT_CASE(MCAMD_DST_REORG);
    else MCAMD_REQUIRE(false, "bn_act_fwd: bad mode %d", d->mode);
#undef ACT_CASE
#undef ACT_INST
    MCAMD_LAUNCH_CHECK("bn_act_fwd");
    return MCAMD_OK;
}

static const int kBwdBlocks = 1024;

extern "C" size_t mcamd_bn_act_bwd_workspace_bytes(const mcamd_act_bwd_desc* d) {
    return ((size_t)kBwdBlocks * 2 * d->C + 2 * (size_t)d->C) * sizeof(float);
}

extern "C" int mcamd_bn_act_bwd(const mcamd_act_bwd_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "bn_act_bwd: null descriptor");
        const mcamd_act_bwd_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_bn_act_bwd(&d_, workspace, workspace_bytes, s); });
    }
    MCAMD_REQUIRE(d && (d->y || d->act) && d->g && d->dy && d->scale && d->shift && d->mean && d->invstd && workspace,
                  "bn_act_bwd: null argument");
    if (check_c(d->C, "bn_act_bwd")) return MCAMD_EINVAL;
    if (workspace_bytes < mcamd_bn_act_bwd_workspace_bytes(d)) {
        mcamd_set_error("bn_act_bwd: workspace %zu < %zu bytes", workspace_bytes, mcamd_bn_act_bwd_workspace_bytes(d));
        return MCAMD_EWORKSPACE;
    }
    MCAMD_REQUIRE(d->y_ld % 8 == 0 && d->y_choff % 8 == 0 && d->g_ld % 8 == 0 && d->g_choff % 8 == 0 &&
                      d->g2_ld % 8 == 0 && d->g2_choff % 8 == 0 && d->dy_ld % 8 == 0 && d->dy_choff % 8 == 0,
                  "bn_act_bwd: leading dimensions / channel offsets must be multiples of 8");
    MCAMD_REQUIRE(d->mode == MCAMD_DST_PLAIN || (d->H % 2 == 0 && d->W % 2 == 0), "bn_act_bwd: pool/reorg need even H, W");
    MCAMD_REQUIRE(d->grad_scale > 0.f, "bn_act_bwd: grad_scale must be positive");
    MCAMD_REQUIRE(d->y_dtype == 0 || d->y_dtype == 1, "bn_act_bwd: y_dtype %d (0 = fp16, 1 = fp32)", d->y_dtype);
    const bool y32 = d->y_dtype == 1;
    ActBwdArgs a;
    a.y = d->y;
    a.scale = d->scale, a.shift = d->shift, a.mean = d->mean, a.invstd = d->invstd;
    a.g = (const half_t*)d->g;
    a.g2 = (const half_t*)d->g2;
    a.dy = (half_t*)d->dy;
    a.slab = (float*)workspace;
    float* coef = (float*)workspace + (size_t)kBwdBlocks * 2 * d->C;
    a.coef = coef;
    a.dy_keep = d->dy_keep;
    a.overflow = (int*)d->overflow;
    a.B = d->B, a.H = d->H, a.W = d->W, a.C = d->C;
    a.y_ld = d->y_ld, a.y_choff = d->y_choff, a.g_ld = d->g_ld, a.g_choff = d->g_choff;
    a.g2_ld = d->g2_ld, a.g2_choff = d->g2_choff, a.dy_ld = d->dy_ld, a.dy_choff = d->dy_choff;
    MCAMD_REQUIRE(d->dy_pad == 0 || d->dy_pad == 1, "bn_act_bwd: dy_pad must be 0 or 1");
    a.dy_pw = d->dy_pad ? 1 : 2;
    a.slope = d->slope;
    a.act = (const half_t*)d->act;
    a.act_ld = d->act_ld, a.act_choff = d->act_choff;
    a.act_pw = d->act_pad ? 1 : 2;
    if (d->act) {
        MCAMD_REQUIRE((d->mode == MCAMD_DST_PLAIN && !d->g2) || d->mode == MCAMD_DST_POOL,
                      "bn_act_bwd: `act` (backward from the stored activation) is for PLAIN blocks without a second gradient "
                      "and for MaxPool blocks (a full-resolution copy of the activation)");
        MCAMD_REQUIRE(d->act_ld % 8 == 0 && d->act_choff % 8 == 0 && d->act_choff + d->C <= d->act_ld &&
                          (d->act_pad == 0 || d->act_pad == 1),
                      "bn_act_bwd: activation slice [%d, %d) does not fit act_ld %d", d->act_choff, d->act_choff + d->C, d->act_ld);
        MCAMD_REQUIRE(d->slope > 0.f, "bn_act_bwd: `act` needs an invertible activation (slope > 0)");
        MCAMD_REQUIRE(!d->y || d->y_dtype == 1, "bn_act_bwd: with `act`, `y` is NULL or the fp32 raw output (read for channels "
                                                "whose gamma is 0 only)");
    }
    long long pixels = (long long)d->B * d->H * d->W;
    double count = (double)pixels;
    if (d->mode != MCAMD_DST_PLAIN) pixels /= 4;
    a.items = pixels * (d->C / 8);
    int grid = stream_grid(a.items);
    if (grid > kBwdBlocks) grid = kBwdBlocks;
    hipStream_t st = (hipStream_t)stream;
#define BWD_INST(MODE_, PHASE)                                                                                     \
    do {                                                                                                          \
        if (y32) hipLaunchKernelGGL((bn_act_bwd_kernel<MODE_, PHASE, true>), dim3(grid), dim3(256), 0, st, a);    \
        else hipLaunchKernelGGL((bn_act_bwd_kernel<MODE_, PHASE, false>), dim3(grid), dim3(256), 0, st, a);       \
    } while (0)
#define BWD_LAUNCH(PHASE)                                                                                         \
    if (d->mode == MCAMD_DST_PLAIN)                                                                               \
        BWD_INST(MCAMD_DST_PLAIN, PHASE);                                                                         \
    else if (d->mode == MCAMD_DST_POOL)                                                                           \
        BWD_INST(MCAMD_DST_POOL, PHASE);                                                                          \
    else if (d->mode == MCAMD_DST_REORG)                                                                          \
        BWD_INST(MCAMD_DST_REORG, PHASE);                                                                         \
    else                                                                                                          \
        MCAMD_REQUIRE(false, "bn_act_bwd: bad mode %d", d->mode);
    // MaxPool blocks: the argmax form (bn_pool_bwd_kernel); MCAMD_BN_POOL_FAST=0: generic kernel
    const bool pool_fast_on = MCAMD_ENV_INT("MCAMD_BN_POOL_FAST", 1) != 0;
    const bool pool_fast = pool_fast_on && d->mode == MCAMD_DST_POOL && a.items < (1ll << 31) &&
                           (unsigned long long)grid * 256ull < (1ull << 31);
#define POOL_INST(PHASE, Y32_, G2_) hipLaunchKernelGGL((bn_pool_bwd_kernel<PHASE, Y32_, G2_>), dim3(grid), dim3(256), 0, st, a)
#define POOL_LAUNCH(PHASE)                                                                                        \
    do {                                                                                                          \
        if (a.act && d->g2) hipLaunchKernelGGL((bn_pool_bwd_act_kernel<PHASE, true>), dim3(grid), dim3(256), 0, st, a);   \
        else if (a.act) hipLaunchKernelGGL((bn_pool_bwd_act_kernel<PHASE, false>), dim3(grid), dim3(256), 0, st, a);      \
        else if (y32 && d->g2) POOL_INST(PHASE, true, true);                                                      \
        else if (y32) POOL_INST(PHASE, true, false);                                                              \
        else if (d->g2) POOL_INST(PHASE, false, true);                                                            \
        else POOL_INST(PHASE, false, false);                                                                      \
    } while (0)
    // PLAIN blocks without a second gradient: bn_plain_bwd_kernel (MCAMD_BN_PLAIN_FAST=0: generic kernel)
    const bool plain_fast = MCAMD_ENV_INT("MCAMD_BN_PLAIN_FAST", 1) != 0 &&
                            d->mode == MCAMD_DST_PLAIN && !d->g2 && a.items < (1ll << 31) &&
                            (unsigned long long)grid * 256ull < (1ull << 31);
    MCAMD_REQUIRE(!d->act || plain_fast || pool_fast, "bn_act_bwd: `act` needs the plain / pool fast path (fewer than 2^31 items, "
                                                      "MCAMD_BN_POOL_FAST / MCAMD_BN_PLAIN_FAST not 0)");
#define PLAIN_LAUNCH(PHASE)                                                                                       \
    do {                                                                                                          \
        if (a.act) hipLaunchKernelGGL((bn_plain_bwd_act_kernel<PHASE>), dim3(grid), dim3(256), 0, st, a);         \
        else if (y32) hipLaunchKernelGGL((bn_plain_bwd_kernel<PHASE, true>), dim3(grid), dim3(256), 0, st, a);    \
        else hipLaunchKernelGGL((bn_plain_bwd_kernel<PHASE, false>), dim3(grid), dim3(256), 0, st, a);            \
    } while (0)
    if (pool_fast) POOL_LAUNCH(0);
    else if (plain_fast) PLAIN_LAUNCH(0);
    else {
        BWD_LAUNCH(0)
    }
    MCAMD_LAUNCH_CHECK("bn_act_bwd reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((d->C + RED_CPB - 1) / RED_CPB), dim3(1024), 0, st, (const float*)a.slab, grid, d->C,
                       count, 1.0f / d->grad_scale, d->dgamma, d->dbeta, coef, (const int*)d->chan_perm,
                       d->skip_dead_param_grads);
    MCAMD_LAUNCH_CHECK("bn_act_bwd finalize");
    if (pool_fast) POOL_LAUNCH(1);
    else if (plain_fast) PLAIN_LAUNCH(1);
    else {
        BWD_LAUNCH(1)
    }
    MCAMD_LAUNCH_CHECK("bn_act_bwd apply");
#undef POOL_LAUNCH
#undef PLAIN_LAUNCH
#undef POOL_INST
#undef BWD_LAUNCH
#undef BWD_INST
    return MCAMD_OK;
}

extern "C" int mcamd_nchw_f32_to_padded_nhwc_f16(const float* src, int32_t B, int32_t C, int32_t H, int32_t W, float mul,
                                                 void* dst, int32_t dst_ld, int32_t dst_choff, int32_t* overflow,
                                                 void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) {
            return mcamd_nchw_f32_to_padded_nhwc_f16(src, B, C, H, W, mul, dst, dst_ld, dst_choff, overflow, s);
        });
    MCAMD_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad argument");
    MCAMD_REQUIRE(dst_choff + C <= dst_ld || dst_ld == 4, "nchw_to_nhwc: channel slice exceeds ld");
    MCAMD_REQUIRE((C >= 8) ? (dst_ld % 8 == 0 && dst_choff % 8 == 0) : true, "nchw_to_nhwc: alignment");
    int cgroups = (C + 7) / 8;
    long long items = (long long)B * H * W * cgroups;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_grid(items)), dim3(256), 0, (hipStream_t)stream, src, B, C, H, W,
                       mul, (half_t*)dst, dst_ld, dst_choff, cgroups, (int*)overflow, 0, 2);
    MCAMD_LAUNCH_CHECK("nchw_to_nhwc");
    return MCAMD_OK;
}

extern "C" int mcamd_nchw_f32_to_padded_nhwc_f16_pad(const float* src, int32_t B, int32_t C, int32_t H, int32_t W, float mul,
                                                     void* dst, int32_t dst_ld, int32_t dst_choff, int32_t pad,
                                                     int32_t* overflow, void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) {
            return mcamd_nchw_f32_to_padded_nhwc_f16_pad(src, B, C, H, W, mul, dst, dst_ld, dst_choff, pad, overflow, s);
        });
    MCAMD_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && (pad == 0 || pad == 1), "nchw_to_nhwc_pad: bad argument");
    MCAMD_REQUIRE(dst_choff + C <= dst_ld && dst_ld != 4, "nchw_to_nhwc_pad: channel slice exceeds ld (and not the stem image)");
    MCAMD_REQUIRE((C >= 8) ? (dst_ld % 8 == 0 && dst_choff % 8 == 0) : true, "nchw_to_nhwc_pad: alignment");
    int cgroups = (C + 7) / 8;
    long long items = (long long)B * H * W * cgroups;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_grid(items)), dim3(256), 0, (hipStream_t)stream, src, B, C, H, W,
                       mul, (half_t*)dst, dst_ld, dst_choff, cgroups, (int*)overflow, 0, pad ? 1 : 2);
    MCAMD_LAUNCH_CHECK("nchw_to_nhwc_pad");
    return MCAMD_OK;
}

extern "C" int mcamd_nchw_f32_to_padded_nhwc_f16_split(const float* src, int32_t B, int32_t C, int32_t H, int32_t W,
                                                       void* dst, int32_t dst_ld, int32_t dst_choff, int32_t plane,
                                                       void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) {
            return mcamd_nchw_f32_to_padded_nhwc_f16_split(src, B, C, H, W, dst, dst_ld, dst_choff, plane, s);
        });
    MCAMD_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc_split: bad argument");
    MCAMD_REQUIRE(plane >= C && dst_choff + 2 * plane + C <= dst_ld, "nchw_to_nhwc_split: three planes of %d channels, %d apart, exceed ld %d",
                  C, plane, dst_ld);
    int cgroups = (C + 7) / 8;
    long long items = (long long)B * H * W * cgroups;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_grid(items)), dim3(256), 0, (hipStream_t)stream, src, B, C, H, W,
                       1.0f, (half_t*)dst, dst_ld, dst_choff, cgroups, (int*)nullptr, plane, 2);
    MCAMD_LAUNCH_CHECK("nchw_to_nhwc_split");
    return MCAMD_OK;
}
