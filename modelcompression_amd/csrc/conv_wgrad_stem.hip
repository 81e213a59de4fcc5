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
#include "tr_frag.h"
#include <stdlib.h>
#include <string.h>

// (transposing reads: inline assembly, see tr_frag.h)
template <int NS>
__global__ __launch_bounds__(256) void wgrad_stem_kernel(WgradArgs a) {
    constexpr int DY_BYTES = 2048, X_ROW = 272, STAGE = 3072;
    __shared__ __attribute__((aligned(16))) char smem[4 * NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + wave * (NS * STAGE);
    const unsigned ring_addr = lds_addr_of(ring);

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
            const unsigned st = ring_addr + slot * STAGE;
            Frag fa[2], fb[2], fc[2];
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                if (k16 == 0) {
                    fa[0].lo = tr_read4<0>(st + a_off), fa[0].hi = tr_read4<4 * 64>(st + a_off);
                    fb[0].lo = tr_read4<0>(st + b01_off), fb[0].hi = tr_read4<4 * 8>(st + b01_off);
                    fc[0].lo = tr_read4<0>(st + b2_off), fc[0].hi = tr_read4<4 * 8>(st + b2_off);
                } else {
                    fa[1].lo = tr_read4<16 * 64>(st + a_off), fa[1].hi = tr_read4<16 * 64 + 4 * 64>(st + a_off);
                    fb[1].lo = tr_read4<16 * 8>(st + b01_off), fb[1].hi = tr_read4<16 * 8 + 4 * 8>(st + b01_off);
                    fc[1].lo = tr_read4<16 * 8>(st + b2_off), fc[1].hi = tr_read4<16 * 8 + 4 * 8>(st + b2_off);
                }
            }
            lds_wait_all(fa[0]);
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                tie(fa[k16]), tie(fb[k16]), tie(fc[k16]);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k16].v(), fb[k16].v(), acc01, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k16].v(), fc[k16].v(), acc2, 0, 0, 0);
            }
            // the fragments are in registers (lds_wait_all above) before the slot is re-staged
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


// ---------------------------------------------------------------------------------------
// The same idea for the 3x3 layers with 32 (padded) input channels on huge images (conv2: 64 x 208 x 208
// pixels, 32 -> 64 channels).  wgrad_kernel<64, 32, 9> stages nine shifted 32-channel tiles of X per step
// (18 KB + 4 KB of dY per 32 pixels); here a 16-pixel step stages 2 KB of dY and the raw window of 3 rows x
// 18 pixels x 64 bytes (3.4 KB), and the B fragment of tap (ty, tx) is the window read at row ty, pixel
// k + tx.  A wave owns ALL of dW (NI filter blocks x 9 taps = 18 accumulators at NI = 2, 288 registers; one wave
// per SIMD) and runs independently on a contiguous run of steps with a private NS-stage ring of 6 KB stages
// (6 DMA instructions per step).  W % 16 == 0 keeps a step inside one image row.
template <int NI, int NS>
__global__ __launch_bounds__(256, 1) void wgrad_win_kernel(WgradArgs a) {
    constexpr int RBA = NI * 64;                 // dY tile row: NI*32 filters
    constexpr int DY_BYTES = 16 * RBA;           // 16 pixels
    constexpr int DY_INSTR = DY_BYTES / 1024;    // 1 | 2
    constexpr int X_ROW = 18 * 64, X_PIECES = 3 * 18 * 4, X_INSTR = 4;   // 216 pieces in 4 instructions (256 slots)
    constexpr int STAGE = DY_BYTES + 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + wave * (NS * STAGE);

    const int nsteps = a.M / 16;
    const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    const int per = (nsteps + nwaves - 1) / nwaves;
    const int s_begin = gw * per;
    const int s_end = s_begin + per < nsteps ? s_begin + per : nsteps;

    // DMA roles.  dY piece -> pixel piece / (RBA/16), chunk (swizzled against the transposing read).
    long long dy_lane[DY_INSTR], x_lane[X_INSTR];
#pragma unroll
    for (int it = 0; it < DY_INSTR; ++it) {
        const int piece = it * 64 + lane, row = piece / (RBA / 16), ch = piece % (RBA / 16);
        dy_lane[it] = (long long)row * a.dy_ld + ((ch ^ tr_swz<RBA>(row)) * 8) + a.dy_off;
    }
#pragma unroll
    for (int it = 0; it < X_INSTR; ++it) {
        int piece = it * 64 + lane;
        if (piece > X_PIECES - 1) piece = X_PIECES - 1;          // spare slots repeat a valid address
        const int r = piece / 72, j = piece - 72 * r;
        x_lane[it] = (long long)r * a.x_row_stride + (long long)(j >> 2) * a.x_ld + (j & 3) * 8 + a.x_off;
    }

    int ib = 0, ih = 0, iw = 0, istep = s_begin;
    if (s_begin < s_end) {
        const long long m0 = (long long)s_begin * 16;
        ib = (int)(m0 / a.HW);
        const int rem = (int)(m0 - (long long)ib * a.HW);
        ih = rem / a.W;
        iw = rem - ih * a.W;
    }
    auto issue = [&](int slot) {
        char* st = ring + slot * STAGE;
        const half_t* dyp = a.dy + (long long)ib * a.dy_img_stride + (long long)ih * a.dy_row_stride + (long long)iw * a.dy_ld;
#pragma unroll
        for (int it = 0; it < DY_INSTR; ++it) glds16(dyp + dy_lane[it], st + it * 1024);
        const half_t* xp = a.x + (long long)ib * a.x_img_stride + (long long)ih * a.x_row_stride + (long long)iw * a.x_ld;
#pragma unroll
        for (int it = 0; it < X_INSTR; ++it) glds16(xp + x_lane[it], st + DY_BYTES + it * 1024);
        if (istep + 1 < s_end) {
            ++istep;
            iw += 16;
            if (iw >= a.W) {
                iw = 0;
                if (++ih >= a.H) ih = 0, ++ib;
            }
        }
    };

    f32x16_t acc[NI][9];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    // fragment addressing (tr_frag in tr_frag.h, k16 step 0): lane l reads rows krow and krow + 4
    const unsigned ring_addr = lds_addr_of(ring);
    unsigned a_off[NI], b_off;
    {
        const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
        const int krow = 8 * (g >> 1) + q;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int off = (i * 32 + 16 * (g & 1) + 4 * p4) * 2;
            a_off[i] = krow * RBA + ((((off >> 4) ^ tr_swz<RBA>(krow)) << 4) | (off & 15));
        }
        b_off = krow * 64 + (16 * (g & 1) + 4 * p4) * 2;     // 64-byte rows need no swizzle
    }

    if (s_begin < s_end) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) issue(s);
        int slot = 0;
        for (int s = s_begin; s < s_end; ++s) {
            int nslot = slot + NS - 1;
            if (nslot >= NS) nslot -= NS;
            issue(nslot);
            constexpr int PER = DY_INSTR + X_INSTR;          // DMA instructions per stage; NS - 1 stages stay in flight
            static_assert(PER * (NS - 1) <= 63, "vmcnt range");
            __builtin_amdgcn_s_waitcnt(0x0F70 | ((PER * (NS - 1)) & 15) | (((PER * (NS - 1)) >> 4) << 14));
            const unsigned st = ring_addr + slot * STAGE;
            Frag af[NI], bf[9];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i].lo = tr_read4<0>(st + a_off[i]), af[i].hi = tr_read4<4 * RBA>(st + a_off[i]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // tap (ty, tx): window row ty, pixel k + tx
                if (t == 0) bf[0].lo = tr_read4<DY_BYTES + 0 * X_ROW + 0 * 64>(st + b_off), bf[0].hi = tr_read4<DY_BYTES + 0 * X_ROW + 0 * 64 + 256>(st + b_off);
                if (t == 1) bf[1].lo = tr_read4<DY_BYTES + 0 * X_ROW + 1 * 64>(st + b_off), bf[1].hi = tr_read4<DY_BYTES + 0 * X_ROW + 1 * 64 + 256>(st + b_off);
                if (t == 2) bf[2].lo = tr_read4<DY_BYTES + 0 * X_ROW + 2 * 64>(st + b_off), bf[2].hi = tr_read4<DY_BYTES + 0 * X_ROW + 2 * 64 + 256>(st + b_off);
                if (t == 3) bf[3].lo = tr_read4<DY_BYTES + 1 * X_ROW + 0 * 64>(st + b_off), bf[3].hi = tr_read4<DY_BYTES + 1 * X_ROW + 0 * 64 + 256>(st + b_off);
                if (t == 4) bf[4].lo = tr_read4<DY_BYTES + 1 * X_ROW + 1 * 64>(st + b_off), bf[4].hi = tr_read4<DY_BYTES + 1 * X_ROW + 1 * 64 + 256>(st + b_off);
                if (t == 5) bf[5].lo = tr_read4<DY_BYTES + 1 * X_ROW + 2 * 64>(st + b_off), bf[5].hi = tr_read4<DY_BYTES + 1 * X_ROW + 2 * 64 + 256>(st + b_off);
                if (t == 6) bf[6].lo = tr_read4<DY_BYTES + 2 * X_ROW + 0 * 64>(st + b_off), bf[6].hi = tr_read4<DY_BYTES + 2 * X_ROW + 0 * 64 + 256>(st + b_off);
                if (t == 7) bf[7].lo = tr_read4<DY_BYTES + 2 * X_ROW + 1 * 64>(st + b_off), bf[7].hi = tr_read4<DY_BYTES + 2 * X_ROW + 1 * 64 + 256>(st + b_off);
                if (t == 8) bf[8].lo = tr_read4<DY_BYTES + 2 * X_ROW + 2 * 64>(st + b_off), bf[8].hi = tr_read4<DY_BYTES + 2 * X_ROW + 2 * 64 + 256>(st + b_off);
            }
            lds_wait_all(af[0]);
#pragma unroll
            for (int i = 0; i < NI; ++i) tie(af[i]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                tie(bf[t]);
#pragma unroll
                for (int i = 0; i < NI; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i].v(), bf[t].v(), acc[i][t], 0, 0, 0);
            }
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // sum the four waves, one accumulator at a time: red[wave][r][lane]
    float* red = (float*)smem;
    float* out = a.slab + (long long)blockIdx.x * a.rows_pad * a.ktot;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[i][t][r];
            __syncthreads();
            for (int idx = tid; idx < 1024; idx += 256) {
                const float v = red[idx] + red[1024 + idx] + red[2048 + idx] + red[3072 + idx];
                const int r = idx >> 6, ln = idx & 63;
                out[(long long)(i * 32 + mfma32_row(r, ln)) * a.ktot + t * 32 + (ln & 31)] = v;   // slab K order: tap-major
            }
            __syncthreads();
        }
}

bool mcamd_wgrad_win_ok(int ksize, int stem, int cout, int cin_tap, int W, long long M) {
    return ksize == 3 && !stem && cin_tap == 32 && round_up_int(cout, 32) <= 64 && W % 16 == 0 && M >= 4096;
}

WgradPlan mcamd_wgrad_win_plan(long long M, int cout) {
    WgradPlan p;
    memset(&p, 0, sizeof(p));
    p.stemw = 2;
    p.rows_pad = round_up_int(cout, 32);
    p.tmo = p.rows_pad, p.tnc = 32, p.taps = 9, p.kp = 16;
    p.n_otiles = p.n_ctiles = p.n_tapgroups = 1;
    long long steps = M / 16, wgs = (steps + 31) / 32;     // at least 8 steps per wave
    p.nsplit = (int)(wgs < 256 ? wgs : 256);                // one workgroup per CU
    p.bytes = (size_t)p.nsplit * p.rows_pad * 288 * sizeof(float);
    return p;
}

template <int NI>
static void launch_win(const WgradArgs& a, int grid, hipStream_t st) {
    constexpr int NS = 6;                                     // 5 stages (27 KB per wave) in flight: latency-bound below that
    const size_t lds = 4 * NS * (16 * NI * 64 + 4096);
    MCAMD_LDS_OPT_IN((wgrad_win_kernel<NI, NS>), lds);   // lds is a per-instance constant
    hipLaunchKernelGGL((wgrad_win_kernel<NI, NS>), dim3(grid), dim3(256), lds, st, a);
}

int mcamd_wgrad_win_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st) {
    if (a.W % 16 != 0 || a.cin_tap != 32 || a.ktot != 288 || (p.rows_pad != 32 && p.rows_pad != 64)) {
        mcamd_set_error("wgrad_win: needs W %% 16 == 0, 32 padded input channels, <= 64 filters (W %d, cin_tap %d, rows %d)",
                        a.W, a.cin_tap, p.rows_pad);
        return MCAMD_EINVAL;
    }
    a.rows_pad = p.rows_pad;
    a.nsplit = p.nsplit;
    if (p.rows_pad == 32) launch_win<1>(a, p.nsplit, st);
    else launch_win<2>(a, p.nsplit, st);
    MCAMD_LAUNCH_CHECK("wgrad_win");
    return MCAMD_OK;
}
