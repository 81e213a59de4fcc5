// Weight-gradient convolution for gfx950 (MI355X).
//
//   dW[n][tap*cin + c] = sum_{m} dY[pixel(m)][n] * X[pixel(m) + tap][c]
//
// A GEMM whose reduction index is the pixel: both operands (padded NHWC fp16) have the
// reduction index as the SLOW dimension, so MFMA fragments (8 consecutive k per lane) are
// gathered with gfx950's transposing LDS read ds_read_b64_tr_b16 -- no transposed copy of
// the activations is ever made.  Tiles of 32 pixels x TMo output channels (dY) and
// 32 pixels x TNc input channels (X shifted by the tap) are DMA'd into LDS
// (global_load_lds_dwordx4, per-lane source addresses), two stages.
// The pixel range is split over grid.y; each split writes its own fp32 slab and
// wgrad_finish_kernel sums the slabs in a fixed order (deterministic, no atomics), applies
// the pruning mask and 1/grad_scale, and writes fp32 OIHW -- i.e. autograd's gradient of
// `self.weight * mask` followed by F.conv2d (reference layers.py:59-64).
#include "kernels.h"
#include <stdlib.h>


__device__ __forceinline__ h8_t tr_frag(const char* tile, int rowbytes, int s, int colbase, int lane) {
    // Fragment for a 32x32x16 MFMA operand whose k index is the LDS row:
    // lane l gets T[k = 16*s + 8*(l>>5) + j][colbase + (l&31)], j = 0..7.
    // ds_read_b64_tr_b16: within each 16-lane group, lane 4q+p supplies the address of row q,
    // columns 4p..4p+3 of a 4x16 block; lane i receives column i of the 4 rows.
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int kb = 16 * s + 8 * (g >> 1);
    const int cb = colbase + 16 * (g & 1);
    const char* p0 = tile + (kb + q) * rowbytes + (cb + 4 * p) * 2;
    const char* p1 = p0 + 4 * rowbytes;
    union {
        fp16x4_t h[2];
        h8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p1);
    return u.v;
}

struct PixState {
    int b, h, w, m;
};

template <int TMo, int TNc>
__global__ __launch_bounds__((TMo >= 64 ? 2 : 1) * (TNc >= 64 ? 2 : 1) * 64) void wgrad_kernel(WgradArgs a) {
    constexpr int WAVES_M = TMo >= 64 ? 2 : 1, WAVES_N = TNc >= 64 ? 2 : 1;
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int WMo = TMo / WAVES_M, WNc = TNc / WAVES_N;
    constexpr int TI = WMo / 32, TJ = WNc / 32;
    constexpr int A_CH = TMo / 8, B_CH = TNc / 8;
    constexpr int A_SLOTS = 32 * A_CH, B_SLOTS = 32 * B_CH;
    constexpr int A_IT = (A_SLOTS + NT - 1) / NT, B_IT = (B_SLOTS + NT - 1) / NT;
    constexpr int STAGE_BYTES = (A_SLOTS + B_SLOTS) * 16;
    static_assert(A_SLOTS % 64 == 0 && B_SLOTS % 64 == 0, "whole waves per DMA instruction");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int per_o = a.ntaps * a.n_ctiles;
    const int ot = blockIdx.x / per_o;
    const int rem = blockIdx.x - ot * per_o;
    const int tap = rem / a.n_ctiles;
    const int ct = rem - tap * a.n_ctiles;
    const int split = blockIdx.y;
    const int k0 = split * a.pix_per_split;
    int k1 = k0 + a.pix_per_split;
    if (k1 > a.M) k1 = a.M;
    const int nsteps = (k1 - k0 + 31) / 32;
    const int x_tap_off = a.tap_off[tap] + a.x_off + ct * TNc;
    const int dy_col_off = ot * TMo;

    // Pixel state per DMA slot (row of the 32-pixel chunk this thread fetches).
    PixState pa[A_IT], pb[B_IT];
    int cha[A_IT], chb[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int slot = it * NT + tid;
        int row = slot / A_CH;
        cha[it] = (slot - row * A_CH) * 8;
        int m = k0 + row;
        int mm = m < a.M ? m : 0;
        pa[it].m = m;
        pa[it].b = mm / a.HW;
        int r2 = mm - pa[it].b * a.HW;
        pa[it].h = r2 / a.W;
        pa[it].w = r2 - pa[it].h * a.W;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int slot = it * NT + tid;
        int row = slot / B_CH;
        chb[it] = (slot - row * B_CH) * 8;
        int m = k0 + row;
        int mm = m < a.M ? m : 0;
        pb[it].m = m;
        pb[it].b = mm / a.HW;
        int r2 = mm - pb[it].b * a.HW;
        pb[it].h = r2 / a.W;
        pb[it].w = r2 - pb[it].h * a.W;
    }
    auto advance = [&](PixState& p) {
        p.m += 32;
        p.w += 32;
        while (p.w >= a.W) {
            p.w -= a.W;
            p.h += 1;
        }
        while (p.h >= a.H) {
            p.h -= a.H;
            p.b += 1;
        }
    };

    f32x16_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_SLOTS * 16;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int wslot = it * NT + (tid & ~63);
            if (wslot < A_SLOTS) {
                const half_t* src;
                if (pa[it].m < k1)
                    src = a.dy + (long long)pa[it].b * a.dy_img_stride + (long long)pa[it].h * a.dy_row_stride +
                          (long long)pa[it].w * a.dy_ld + a.dy_off + dy_col_off + cha[it];
                else
                    src = a.dy + a.dy_zero_off + dy_col_off + cha[it];  // zero halo pixel: contributes nothing
                glds16(src, sa + wslot * 16);
            }
            advance(pa[it]);
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int wslot = it * NT + (tid & ~63);
            if (wslot < B_SLOTS) {
                const half_t* src;
                if (pb[it].m < k1)
                    src = a.x + (long long)pb[it].b * a.x_img_stride + (long long)pb[it].h * a.x_row_stride +
                          (long long)pb[it].w * a.x_ld + x_tap_off + chb[it];
                else
                    src = a.x + a.x_off + chb[it];  // any finite data; its dY partner is zero
                glds16(src, sb + wslot * 16);
            }
            advance(pb[it]);
        }
    };

    if (nsteps > 0) stage(0);
    for (int st = 0; st < nsteps; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nsteps) stage((st + 1) & 1);
        const char* sa = smem + (st & 1) * STAGE_BYTES;
        const char* sb = sa + A_SLOTS * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h8_t af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = tr_frag(sa, TMo * 2, s, wm * WMo + i * 32, lane);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = tr_frag(sb, TNc * 2, s, wn * WNc + j * 32, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    float* out = a.slab + (long long)split * a.rows_pad * a.ktot;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = ot * TMo + wm * WMo + i * 32 + mfma32_row(r, lane);
                int k = tap * a.cin_tap + ct * TNc + wn * WNc + j * 32 + (lane & 31);
                out[(long long)n * a.ktot + k] = acc[i][j][r];
            }
}

// Sum the split slabs in order, apply mask and 1/grad_scale, write fp32 OIHW.
// grid (ceil(Cin/256), Cout), 256 threads; KK = ksize*ksize.
__global__ void wgrad_finish_kernel(const float* slab, int nsplit, int rows_pad, int ktot, int cin_tap, int stem,
                                    int Cout, int Cin, int KK, const float* mask, float inv_scale, float* dw) {
    __shared__ float buf[256 * 9];
    const int n = blockIdx.y;
    const int c0 = blockIdx.x * 256;
    const int c = c0 + threadIdx.x;
    const long long split_stride = (long long)rows_pad * ktot;
    if (c < Cin) {
        for (int t = 0; t < KK; ++t) {
            int kidx = stem ? (t / 3) * 32 + (t % 3) * 4 + c : t * cin_tap + c;
            const float* p = slab + (long long)n * ktot + kidx;
            float v = 0.f;
            for (int s = 0; s < nsplit; ++s) v += p[s * split_stride];
            buf[threadIdx.x * KK + t] = v;
        }
    }
    __syncthreads();
    int cnt = Cin - c0;
    if (cnt > 256) cnt = 256;
    const long long base = ((long long)n * Cin + c0) * KK;
    for (int e = threadIdx.x; e < cnt * KK; e += 256) {
        float v = buf[e] * inv_scale;
        if (mask) v *= mask[base + e];
        dw[base + e] = v;
    }
}

// dbias[n] = sum over every padded pixel of dy[pixel][choff + n] / grad_scale (halo rows are zero).
__global__ void colsum_kernel(const half_t* dy, long long rows, int ld, int choff, int C, float inv_scale, float* out) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    float s = 0.f;
    for (long long r = threadIdx.x; r < rows; r += 256) s += (float)dy[r * ld + choff + n];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && n < C) out[n] = red[0] * inv_scale;
}

// ---------------------------------------------------------------------------------------
static int pick_t(int n) {  // largest of 128/64/32 dividing round_up(n, 32)
    int p = round_up_int(n, 32);
    if (p % 128 == 0) return 128;
    if (p % 64 == 0) return 64;
    return 32;
}


static int env_int_w(const char* name, int dflt) {
    const char* s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

WgradPlan mcamd_wgrad_plan(long long M, int cout, int cin_tap, int ntaps) {
    WgradPlan p;
    p.tmo = pick_t(cout);
    p.tnc = pick_t(cin_tap);
    p.rows_pad = round_up_int(cout, p.tmo);
    p.n_otiles = p.rows_pad / p.tmo;
    p.n_ctiles = cin_tap / p.tnc;
    long long tiles = (long long)p.n_otiles * ntaps * p.n_ctiles;
    long long target = env_int_w("MCAMD_WGRAD_WGS", 1024);
    long long ns = (target + tiles - 1) / tiles;
    long long max_by_work = (M + 255) / 256;  // at least 8 steps of 32 pixels per split
    if (ns > max_by_work) ns = max_by_work;
    if (ns < 1) ns = 1;
    long long pps = ((M + ns - 1) / ns + 31) / 32 * 32;
    ns = (M + pps - 1) / pps;
    p.nsplit = (int)ns;
    p.pix_per_split = (int)pps;
    p.bytes = (size_t)p.nsplit * p.rows_pad * ((size_t)ntaps * cin_tap) * sizeof(float);
    return p;
}

template <int TMo, int TNc>
static void launch_w(const WgradArgs& a, int gx, int gy, hipStream_t st) {
    constexpr int NT = (TMo >= 64 ? 2 : 1) * (TNc >= 64 ? 2 : 1) * 64;
    size_t lds = 2 * (size_t)(32 * (TMo / 8) + 32 * (TNc / 8)) * 16;
    hipLaunchKernelGGL((wgrad_kernel<TMo, TNc>), dim3(gx, gy), dim3(NT), lds, st, a);
}

int mcamd_wgrad_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st) {
    a.rows_pad = p.rows_pad;
    a.n_ctiles = p.n_ctiles;
    a.pix_per_split = p.pix_per_split;
    int gx = p.n_otiles * a.ntaps * p.n_ctiles, gy = p.nsplit;
#define W_CASE(TM_, TN_) \
    if (p.tmo == TM_ && p.tnc == TN_) launch_w<TM_, TN_>(a, gx, gy, st);
    W_CASE(128, 128) W_CASE(128, 64) W_CASE(128, 32) W_CASE(64, 128) W_CASE(64, 64) W_CASE(64, 32) W_CASE(32, 128)
    W_CASE(32, 64) W_CASE(32, 32)
#undef W_CASE
    MCAMD_LAUNCH_CHECK("wgrad");
    return MCAMD_OK;
}

int mcamd_wgrad_finish_launch(const float* slab, const WgradPlan& p, int ktot, int cin_tap, int stem, int Cout, int Cin,
                              int ksize, const float* mask, float inv_scale, float* dw, hipStream_t st) {
    dim3 grid((Cin + 255) / 256, Cout);
    hipLaunchKernelGGL(wgrad_finish_kernel, grid, dim3(256), 0, st, slab, p.nsplit, p.rows_pad, ktot, cin_tap, stem,
                       Cout, Cin, ksize * ksize, mask, inv_scale, dw);
    MCAMD_LAUNCH_CHECK("wgrad_finish");
    return MCAMD_OK;
}

int mcamd_colsum_launch(const half_t* dy, long long rows, int ld, int choff, int C, float inv_scale, float* out,
                        hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, st, dy, rows, ld, choff, C, inv_scale, out);
    MCAMD_LAUNCH_CHECK("colsum");
    return MCAMD_OK;
}
