// Weight gradient of the 3-channel first layer (conv1: 64 x 416 x 416 pixels, 32 filters).
//
//   dW[n][ty][tx*4 + c] = sum_p dY[p][n] * X[p + (ty, tx)][c]          (X: padded NHWC4 image, 8 bytes per pixel)
//
// The generic wgrad_kernel stages, per 32 pixels and filter row ty, a 32-pixel x 32-"channel" tile of X -- 8
// neighbouring pixels per pixel, 64 bytes where 24 are needed -- so 3/4 of its LDS-DMA traffic is the SAME image
// bytes over and over (6 KB of X against 2 KB of dY per step) and the launch runs at that DMA volume, not at
// HBM speed.  Here a step stages the RAW image window instead: 3 rows x 34 pixels x 8 bytes (816 bytes), and
// the MFMA B fragments are gathered from it with overlapping rows: element (pixel k, column tx*4 + c) of the
// im2col matrix lives at byte (k + tx) * 8 + 2 c of the window row, i.e. the im2col matrix IS the window read
// with a row stride of 8 bytes.  ds_read_b64_tr_b16 takes one address per lane (4 halfs = one pixel), so the
// overlap costs nothing.  One 32x32x16 MFMA covers filter rows ty = 0 (columns 0-15) and 1 (columns 16-31), a
// second one ty = 2.
//
// Waves run independently (no workgroup barrier in the loop): each owns a contiguous run of 32-pixel steps and a
// private NS-stage LDS ring (3 KB per stage: 2 KB dY + 1 KB window), three LDS-DMA instructions per step.  The
// four waves' accumulators are summed through LDS at the end; one fp32 slab per workgroup, summed in a fixed
// order by wgrad_finish_kernel (deterministic, no atomics).
//
// Replaces autograd's weight gradient of F.conv2d at reference src/pruning/weightPruning/layers.py:60-64.
#include "kernels.h"
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ fp16x4_t tr_read4(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p);
}

template <int NS>
__global__ __launch_bounds__(256) void wgrad_stem_kernel(WgradArgs a) {
    constexpr int DY_BYTES = 2048, X_ROW = 272, STAGE = 3072;
    __shared__ __attribute__((aligned(16))) char smem[4 * NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + wave * (NS * STAGE);

    // the wave's run of 32-pixel steps
    const int nsteps = a.M / 32;
    const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int per = (nsteps + nwaves - 1) / nwaves;
    const int s_begin = gw * per;
    const int s_end = s_begin + per < nsteps ? s_begin + per : nsteps;

    // DMA roles.  dY: piece = it*64 + lane -> pixel piece >> 2, 16-byte chunk piece & 3 (tile rows of 64 bytes).
    // window: lane -> row lane / 17, chunk lane % 17; lanes 51-63 repeat a valid address into unused LDS.
    const int xr = lane < 51 ? lane / 17 : 2;
    const int xc = lane < 51 ? lane - 17 * xr : 16;
    const long long x_lane = (long long)xr * a.x_row_stride + xc * 8 + a.x_off;
    const long long dy_lane0 = (long long)(lane >> 2) * a.dy_ld + (lane & 3) * 8 + a.dy_off;
    const long long dy_lane1 = dy_lane0 + 16ll * a.dy_ld;

    // issue position (b, h, w0) of the next step to stage
    int ib = 0, ih = 0, iw = 0, istep = s_begin;
    if (s_begin < s_end) {
        const long long m0 = (long long)s_begin * 32;
        ib = (int)(m0 / a.HW);
        const int rem = (int)(m0 - (long long)ib * a.HW);
        ih = rem / a.W;
        iw = rem - ih * a.W;
    }
    auto issue = [&](int slot) {
        char* st = ring + slot * STAGE;
        const half_t* dyp = a.dy + (long long)ib * a.dy_img_stride + (long long)ih * a.dy_row_stride + (long long)iw * a.dy_ld;
        glds16(dyp + dy_lane0, st);
        glds16(dyp + dy_lane1, st + 1024);
        const half_t* xp = a.x + (long long)ib * a.x_img_stride + (long long)ih * a.x_row_stride + (long long)iw * a.x_ld;
        glds16(xp + x_lane, st + DY_BYTES);
        if (istep + 1 < s_end) {     // the tail re-stages the last step (keeps the vmcnt arithmetic constant)
            ++istep;
            iw += 32;
            if (iw >= a.W) {
                iw = 0;
                if (++ih >= a.H) ih = 0, ++ib;
            }
        }
    };

    f32x16_t acc01, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc01[r] = acc2[r] = 0.f;

    // fragment addressing (see tr_frag in conv_wgrad.hip): lane l reads rows kb + q and kb + q + 4
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int krow = 8 * (g >> 1) + q;
    const int a_off = krow * 64 + (16 * (g & 1) + 4 * p) * 2;              // dY tile: 64-byte rows, columns = filters
    const int b01_off = DY_BYTES + (g & 1) * X_ROW + (krow + p) * 8;       // window row 0 | 1, pixel k + tx
    const int b2_off = DY_BYTES + 2 * X_ROW + (krow + p) * 8;

    if (s_begin < s_end) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue(s);
        int slot = 0;
        for (int s = s_begin; s < s_end; ++s) {
            int nslot = slot + NS - 1;
            if (nslot >= NS) nslot -= NS;
            issue(nslot);
            // three DMA instructions per stage, NS - 1 younger stages may stay in flight
            if (NS == 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (NS == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            const char* st = ring + slot * STAGE;
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                union { fp16x4_t h[2]; h8_t v; } fa, fb, fc;
                fa.h[0] = tr_read4(st + a_off + k16 * 16 * 64);
                fa.h[1] = tr_read4(st + a_off + k16 * 16 * 64 + 4 * 64);
                fb.h[0] = tr_read4(st + b01_off + k16 * 16 * 8);
                fb.h[1] = tr_read4(st + b01_off + k16 * 16 * 8 + 4 * 8);
                fc.h[0] = tr_read4(st + b2_off + k16 * 16 * 8);
                fc.h[1] = tr_read4(st + b2_off + k16 * 16 * 8 + 4 * 8);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.v, fb.v, acc01, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa.v, fc.v, acc2, 0, 0, 0);
            }
            // the fragments are in registers (the MFMAs above waited for them) before the slot is re-staged
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // sum the four waves: red[wave][acc][r][lane]
    float* red = (float*)smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[((wave * 2 + 0) * 16 + r) * 64 + lane] = acc01[r];
        red[((wave * 2 + 1) * 16 + r) * 64 + lane] = acc2[r];
    }
    __syncthreads();
    float* out = a.slab + (long long)blockIdx.x * a.rows_pad * a.ktot;
    for (int idx = tid; idx < 2 * 16 * 64; idx += 256) {
        const float v = red[idx] + red[2048 + idx] + red[4096 + idx] + red[6144 + idx];
        const int which = idx >> 10, r = (idx >> 6) & 15, ln = idx & 63;
        const int n = mfma32_row(r, ln), col = ln & 31;
        // slab columns: filter row ty at ty * 32, then tx * 4 + c (wgrad_finish_kernel, stem form)
        if (which == 0) out[n * a.ktot + (col >> 4) * 32 + (col & 15)] = v;
        else if (col < 16) out[n * a.ktot + 64 + col] = v;
    }
}

bool mcamd_wgrad_stem_ok(int stem, int cout, int W, long long M) {
    const char* e = getenv("MCAMD_WGRAD_STEM");
    if (e && atoi(e) == 0) return false;
    return stem && cout == 32 && W % 32 == 0 && M >= 4096;
}

WgradPlan mcamd_wgrad_stem_plan(long long M) {
    WgradPlan p;
    memset(&p, 0, sizeof(p));
    p.stemw = 1;
    p.tmo = 32, p.tnc = 16, p.taps = 3, p.kp = 32;
    p.rows_pad = 32;
    p.n_otiles = p.n_ctiles = p.n_tapgroups = 1;
    long long steps = M / 32, wgs = (steps + 31) / 32;     // at least 8 steps per wave
    p.nsplit = (int)(wgs < 768 ? wgs : 768);                // 3 workgroups of 48 KB LDS per CU
    p.pix_per_split = 0;
    p.bytes = (size_t)p.nsplit * 32 * 96 * sizeof(float);
    return p;
}

int mcamd_wgrad_stem_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st) {
    if (a.W % 32 != 0 || a.M % 32 != 0 || a.x_ld != 4 || a.ktot != 96) {
        mcamd_set_error("wgrad_stem: needs W %% 32 == 0, NHWC4 input, ktot 96 (W %d, x_ld %d, ktot %d)", a.W, a.x_ld, a.ktot);
        return MCAMD_EINVAL;
    }
    a.rows_pad = p.rows_pad;
    a.nsplit = p.nsplit;
    hipLaunchKernelGGL(wgrad_stem_kernel<4>, dim3(p.nsplit), dim3(256), 0, st, a);
    MCAMD_LAUNCH_CHECK("wgrad_stem");
    return MCAMD_OK;
}
