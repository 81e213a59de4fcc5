// 3x3 convolution from a ROLLING LDS WINDOW for the narrow layers on huge images whose input has 64 (padded) channels
// and at most 32 outputs: the input gradient of conv2 (64 x 208 x 208 pixels, dY 64 channels -> dX 32 channels).
//
// The implicit-GEMM kernels stage nine shifted copies of every activation line (LDS-DMA-bound: 0.25 ms for 0.1
// TFLOP); small3x3_kernel reads the nine taps straight from global memory and is bound by the number of 1 KB
// load instructions (0.35 ms for this shape).  Here every activation line goes through the vector-memory path ONCE:
//  * a wave owns a strip of 32 output columns of one image and walks down its rows; its private LDS ring holds
//    NR padded input rows of 34 pixels (one 4.25 KB row = 5 LDS-DMA instructions per output row, two rows ahead,
//    counted vmcnt; no workgroup barrier anywhere: waves free-run);
//  * the nine taps of an output row are `ds_read_b128` reads of ring rows h, h+1, h+2 at pixel + tx -- the MFMA B
//    operand of v_mfma_f32_16x16x32_f16 (lane = pixel, 8 channels), two 16-pixel groups per row;
//  * the weights (9 taps x 64 channels x 32 outputs = 36 fragments, 144 registers) are the A operand and stay in
//    registers for the whole kernel (transposed product, as small3x3_kernel);
//  * the 16-pixel x 32-channel result leaves through 1 KB of LDS as whole cache lines.
// Units of work = (image, strip, segment of rows), dealt round-robin to the waves of the grid.
//
// Replaces autograd's input gradient of F.conv2d at reference src/pruning/weightPruning/layers.py:60-64.
#include "kernels.h"
#include <stdlib.h>

namespace {
constexpr int NR = 6;                 // ring rows: 3 under the taps + 2 in flight + 1 being overwritten
constexpr int ROW_BYTES = 5 * 1024;   // 34 pixels x 128 bytes = 272 pieces, staged by 5 DMA instructions (320 slots)
constexpr int LA = 2;                 // rows staged ahead of the row being multiplied

template <int N>
__device__ __forceinline__ void wait_vmc() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

// Every wave has its own units and computes all (<= 32) output channels: the input gradient of conv2.  (A four-wave form
// for the forward of conv3/conv5, 64 -> 128 channels with wave w on channels [32 w, 32 w + 32), measured 0.183 ms against
// 0.170 ms for the 128x128 implicit GEMM and was removed in round 3: DESIGN.md, negative findings.)
template <int NB>   // 16-channel output blocks per wave (1 | 2); 64 input channels per tap
__global__ __launch_bounds__(256, 1) void win3x3_kernel(IgemmArgs a, int seg_rows, int nseg, int nstrips) {
    constexpr int CT = 64, KK = 2, NC = NB * 16, PXB = CT * 2;      // bytes per pixel in the ring
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = smem + wave * (NR * ROW_BYTES + 1024);
    half_t* tile = (half_t*)(ring + NR * ROW_BYTES);                // [16 pixels][NC channels] of this wave
    const int pl = lane & 15, kg = lane >> 4;
    constexpr int n0 = 0;                                           // first output channel of this wave

    // weights: A fragment (row n = n0 + nb*16 + lane & 15, k = t*64 + 32 kk + 8 kg .. +7)
    h8_t wf[9][KK][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                wf[t][kk][nb] = *(const h8_t*)(a.w + (long long)(n0 + nb * 16 + pl) * a.ktot + t * CT + 32 * kk + 8 * kg);

    // DMA roles: piece = it*64 + lane -> pixel piece / 8, 16-byte chunk piece % 8; pieces past 271 repeat the last one
    long long src_off[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        int piece = it * 64 + lane;
        if (piece > 271) piece = 271;
        src_off[it] = (long long)(piece >> 3) * a.x_ld + (piece & 7) * 8;
    }
    auto stage_row = [&](const half_t* rowp, int slot) {            // rowp: padded pixel (b, row, 32 s), channel x_off
        char* dst = ring + slot * ROW_BYTES;
#pragma unroll
        for (int it = 0; it < 5; ++it) glds16(rowp + src_off[it], dst + it * 1024);
    };
    const int b_off = pl * PXB + kg * 16;                           // lane's place inside a 16-pixel group of a ring row

    const int nunits = a.M / (a.H * a.W) * nstrips * nseg;
    const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    for (int unit = gw; unit < nunits; unit += nwaves) {
        const int seg = unit % nseg, rest = unit / nseg;
        const int strip = rest % nstrips, b = rest / nstrips;
        const int h_lo = seg * seg_rows;
        int h_hi = h_lo + seg_rows;
        if (h_hi > a.H) h_hi = a.H;
        const int c0 = strip * 32;
        const int ngroups = (a.W - c0 > 16) ? 2 : 1;                // the last strip of a row may be narrower
        // padded row r of this strip starts at padded pixel (b, r, c0)
        const half_t* xs = a.x + (long long)b * a.x_img_stride + (long long)c0 * a.x_ld + a.x_off;
        auto rowptr = [&](int r) {
            if (r > a.H + 1) r = a.H + 1;                            // past the image: re-stage the bottom halo row (never multiplied)
            return xs + (long long)r * a.x_row_stride;
        };
        // prologue: padded rows h_lo .. h_lo + 2 + LA - 1
        int slot_in = 0;                                             // ring slot of the next row to stage
#pragma unroll
        for (int r = 0; r < 2 + LA; ++r) {
            stage_row(rowptr(h_lo + r), slot_in);
            slot_in = slot_in + 1 == NR ? 0 : slot_in + 1;
        }
        int slot0 = 0;                                               // ring slot of padded row h (top tap row)
        half_t* yrow = (half_t*)a.y + ((long long)(b * a.H + h_lo) * a.W + c0) * a.y_ld + a.y_choff + n0;
        for (int h = h_lo; h < h_hi; ++h) {
            stage_row(rowptr(h + 2 + LA), slot_in);
            slot_in = slot_in + 1 == NR ? 0 : slot_in + 1;
            // row h + 2 must have landed: behind it are LA rows of 5 DMA instructions and the stores of the output
            // rows computed since it was issued (ngroups each) -- none in a unit's first row, one row's worth in its
            // second, LA rows' worth from then on.  (Counting stores that were never issued would let the wait pass
            // before the row is in: wrong results under load, B = 64.)
            static_assert(LA == 2, "the three cases below are written out for LA = 2");
            const int k = h - h_lo;
            if (k == 0) wait_vmc<10>();
            else if (k == 1) {
                if (ngroups == 2) wait_vmc<12>();
                else wait_vmc<11>();
            } else {
                if (ngroups == 2) wait_vmc<14>();
                else wait_vmc<12>();
            }
            int s1 = slot0 + 1, s2 = slot0 + 2;
            if (s1 >= NR) s1 -= NR;
            if (s2 >= NR) s2 -= NR;
            const char* r0 = ring + slot0 * ROW_BYTES + b_off;
            const char* r1 = ring + s1 * ROW_BYTES + b_off;
            const char* r2 = ring + s2 * ROW_BYTES + b_off;
            // All 18 fragments of a group are read before its MFMAs, and the second group's reads are issued before the
            // first group's MFMAs (one wave per SIMD: nobody else hides the LDS latency).  sched_barrier keeps hipcc from
            // sinking the reads back to their uses.
            auto read_group = [&](int gi, h8_t (&xf)[9][KK]) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const char* rp = (t / 3 == 0 ? r0 : t / 3 == 1 ? r1 : r2) + (gi * 16 + t % 3) * PXB;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) xf[t][kk] = *(const h8_t*)(rp + kk * 64);
                }
            };
            auto mfma_group = [&](int gi, const h8_t (&xf)[9][KK]) {
                f32x4_t acc[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][kk][nb], xf[t][kk], acc[nb], 0, 0, 0);
                // accumulator: column = lane & 15 = pixel, row = channel 4 kg + r
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    h4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)fminf(fmaxf(acc[nb][e], -65504.f), 65504.f);
                    *(h4_t*)(tile + pl * NC + nb * 16 + 4 * kg) = v;
                }
                // 16 pixels x NC channels = 16 * NC / 8 pieces of 16 bytes, consecutive pixels: whole cache lines
                constexpr int RC = NC / 8;
                if (lane < 16 * RC) {
                    const int prow = lane / RC, pc = lane - prow * RC;
                    const h8_t v = *(const h8_t*)(tile + prow * NC + pc * 8);
                    if (n0 + pc * 8 < a.N && c0 + gi * 16 + prow < a.W) {     // (a ragged last group of a row)
                        *(h8_t*)(yrow + (long long)(gi * 16 + prow) * a.y_ld + pc * 8) = v;
                        if (a.overflow) {       // a clamped value reads back as exactly +-65504
                            bool s_ = false;
#pragma unroll
                            for (int e = 0; e < 8; ++e) s_ |= fabsf((float)v[e]) >= 65504.f;
                            if (s_) atomicOr(a.overflow, 1);
                        }
                    }
                }
            };
            h8_t xf0[9][KK], xf1[9][KK];
            read_group(0, xf0);
            __builtin_amdgcn_sched_barrier(0);
            if (ngroups == 2) read_group(1, xf1);
            mfma_group(0, xf0);
            __builtin_amdgcn_sched_barrier(0);
            if (ngroups == 2) mfma_group(1, xf1);
            yrow += (long long)a.W * a.y_ld;
            slot0 = slot0 + 1 == NR ? 0 : slot0 + 1;
        }
        wait_vmc<0>();     // the look-ahead rows of this unit must not land in the next unit's ring
    }
}

// dgrad-shaped problems: 3x3, 64 padded input channels, <= 32 outputs, raw fp16 epilogue without statistics
static bool win_enabled() { return true; }

bool mcamd_win3x3_ok(const IgemmArgs& a) {
    if (!win_enabled()) return false;
    if (a.mode != MCAMD_EPI_RAW_F16 || a.bias || a.kb != 64 || a.W < 32 || a.H < 8 || a.M % (a.H * a.W) != 0) return false;
    return !a.stats && a.ktot == 9 * 64 && a.cin_tap == 64 && a.N % 8 == 0 && a.N <= 32 && a.W % 16 == 0 && a.M >= 65536;
}

bool mcamd_win3x3_shape(long long M, int n, int cin_tap, int ktot, int W) {   // for mcamd_conv_tile_info (dgrad)
    if (!win_enabled()) return false;
    return ktot == 9 * 64 && cin_tap == 64 && n % 8 == 0 && n <= 32 && W % 16 == 0 && W >= 32 && M >= 65536;
}

template <int NB>
static void launch_win3(const IgemmArgs& a, int grid, size_t lds, int seg_rows, int nseg, int nstrips, hipStream_t st) {
    MCAMD_LDS_OPT_IN((win3x3_kernel<NB>), 160 * 1024);   // the window size follows the image width: opt in to the whole LDS
    hipLaunchKernelGGL((win3x3_kernel<NB>), dim3(grid), dim3(256), lds, st, a, seg_rows, nseg, nstrips);
}

int mcamd_win3x3_launch(const IgemmArgs& a, hipStream_t st) {
    const int nstrips = (a.W + 31) / 32;
    // Row segments: a wave runs `rounds` units of seg_rows + 4 staged rows (4 = the taps' halo + the look-ahead of the
    // prologue).  Pick the segment length whose units fill whole rounds of the 1024 waves (208 rows, 64 images, 7
    // strips: 13 rows -> exactly 7 units per wave, 0.150 ms; 16 rows -> 5.7, 0.165 ms).
    const long long runners = 1024;
    int want = 0;
    if (want <= 0) {
        want = 0;
        const long long per_row_units = (long long)(a.M / (a.H * a.W)) * nstrips;
        double best = 1e30;
        for (int sg = 8; sg <= 32; ++sg) {
            const int ns = (a.H + sg - 1) / sg, rows = (a.H + ns - 1) / ns;
            const long long rounds = (per_row_units * ns + runners - 1) / runners;
            const double cost = (double)rounds * (rows + 4);
            if (cost < best - 1e-9) best = cost, want = sg;
        }
    }
    const int nseg = (a.H + want - 1) / want, seg_rows = (a.H + nseg - 1) / nseg;
    const long long units = (long long)(a.M / (a.H * a.W)) * nstrips * nseg;
    const size_t lds = 4 * (size_t)(NR * ROW_BYTES + 1024);
    long long wgs = (units + 3) / 4;
    const int grid = (int)(wgs < 256 ? wgs : 256);                         // one workgroup per CU (124 KB of LDS)
    if ((a.N + 15) / 16 == 1) launch_win3<1>(a, grid, lds, seg_rows, nseg, nstrips, st);
    else launch_win3<2>(a, grid, lds, seg_rows, nseg, nstrips, st);
    MCAMD_LAUNCH_CHECK("win3x3");
    return MCAMD_OK;
}
