// 9-tap implicit-GEMM convolution over PADDED pixels: forward and dgrad of 3x3 layers.
//
// Rows of the GEMM are the padded pixels p' of the (zero-halo) input, so tap t is the constant row
// shift (ty-1)*(W+2) + (tx-1).  For one 64-channel chunk the nine taps are nine shifted views of ONE
// LDS window of activation rows [p0 - S, p0 + 128 + S): the window is DMA'd once per chunk and only
// the 16 KB weight tile changes per tap -- 18 KB of LDS-DMA per 16 MFMAs/wave instead of 32 KB in the
// one-tap-at-a-time kernel (conv_igemm.hip), and no pixel decomposition in the main loop.  Price: the
// halo rows are computed too (25 % at 13x13, 14 % at 26x26) and dropped in the epilogue.
// Tile 128 (padded pixels) x 128 (channels), 4 waves of 64x64, v_mfma_f32_32x32x16_f16, two weight
// stages + two window stages, workgroups persistent over M tiles; BN partial sums are gathered in the
// store loop (where halo rows are known) and written once per workgroup.
#include "kernels.h"
#include <stdlib.h>

template <int EPI>
__global__ __launch_bounds__(256, 2) void igemm9_kernel(Igemm9Args a) {
    constexpr int BM = 128, BN = 128, BK = 64, CPR = 8, NT = 256;
    constexpr int B_BYTES = BN * BK * 2;                  // 16 KB weight tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int R = BM + 2 * a.S;                           // window rows (multiple of 8)
    const int a_bytes = R * BK * 2;
    char* const abuf = smem;                              // 2 windows
    char* const bbuf = smem + 2 * a_bytes;                // 2 weight tiles

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nt = jb % a.num_ntiles;
    const int pslot = (jb / a.num_ntiles) * 8 + xcd;
    if (pslot >= a.num_pslots) return;
    const int nch = a.cin_tap / BK;
    const int nunits = nch * 9;
    const int a_iters = (R * CPR + NT - 1) / NT;

    // weight-tile DMA: 1024 slots, 4 per thread, fixed rows
    long long bbase[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int slot = it * NT + tid;
        int row = slot >> 3, phys = slot & 7;
        bbase[it] = (long long)(nt * BN + row) * a.ktot + (phys ^ swz<CPR>(row)) * 8;
    }
    float st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st1[i] = st2[i] = 0.f;

    for (int mt = pslot; mt < a.num_mtiles; mt += a.num_pslots) {
        const long long p0 = (long long)mt * BM;
        const half_t* xwin = a.x + (p0 - a.S) * a.x_ld + a.x_off;   // row 0 of the window, channel chunk 0
        f32x16_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        auto stage_b = [&](int u, int buf) {
            // packed K index of unit u = (tap, chunk)
            const int c = u / 9, tap = u - c * 9;
            const long long koff = ((long long)c * 9 + tap) * BK;   // packed K order [64-channel block][tap][64]
            char* sb = bbuf + buf * B_BYTES;
#pragma unroll
            for (int it = 0; it < 4; ++it) glds16(a.w + bbase[it] + koff, sb + (it * NT + wave * 64) * 16);
        };
        auto stage_a = [&](int c, int buf) {
            char* sa = abuf + buf * a_bytes;
            for (int it = 0; it < a_iters; ++it) {
                const int wslot = it * NT + wave * 64;
                if (wslot < R * CPR) {
                    const int slot = wslot + lane;
                    const int row = slot >> 3, phys = slot & 7;
                    glds16(xwin + (long long)row * a.x_ld + c * BK + (phys ^ swz<CPR>(row)) * 8, sa + wslot * 16);
                }
            }
        };

        __syncthreads();  // previous tile's epilogue is done with the LDS
        stage_a(0, 0);
        stage_b(0, 0);
        for (int u = 0; u < nunits; ++u) {
            const int c = u / 9, tap = u - c * 9;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (u + 1 < nunits) stage_b(u + 1, (u + 1) & 1);
            if (tap == 0 && c + 1 < nch) stage_a(c + 1, (c + 1) & 1);
            const char* sa = abuf + (c & 1) * a_bytes;
            const char* sb = bbuf + (u & 1) * B_BYTES;
            const int shift = a.S + (tap / 3 - 1) * a.W2 + (tap % 3 - 1);   // window row of output row 0
#pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                const int chunk = 2 * s + (lane >> 5);
                h8_t af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int row = wm * 64 + i * 32 + (lane & 31) + shift;
                    af[i] = *(const h8_t*)(sa + (row * CPR + (chunk ^ swz<CPR>(row))) * 16);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int row = wn * 64 + j * 32 + (lane & 31);
                    bf[j] = *(const h8_t*)(sb + (row * CPR + (chunk ^ swz<CPR>(row))) * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }

        // ---- epilogue: fp16 tile through LDS, halo rows dropped, coalesced 16-byte row stores ----
        __syncthreads();
        half_t* ct = (half_t*)smem;  // [128][128]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wn * 64 + j * 32 + (lane & 31);
            float sc = 1.f, sh = 0.f;
            if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                int n = nt * BN + col;
                if (n < a.N) {
                    if (a.scale) sc = a.scale[n];
                    if (a.shift) sh = a.shift[n];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + i * 32 + mfma32_row(r, lane);
                    float v = acc[i][j][r];
                    if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                        v = v * sc + sh;
                        v = v > 0.f ? v : v * a.slope;
                    }
                    ct[row * BN + col] = (half_t)fminf(fmaxf(v, -65504.f), 65504.f);
                }
        }
        __syncthreads();
        half_t* y = (half_t*)a.y;
        const int ch = tid & 15;                 // 16-byte chunk (8 channels) this thread always handles
        const int n0 = nt * BN + ch * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (tid >> 4) + 16 * k;
            const long long pp = p0 + row;
            // padded pixel -> (b, hp, wp); only interior pixels are real outputs
            const int b = (int)(pp / a.HW2);
            const int rem = (int)(pp - (long long)b * a.HW2);
            const int hp = rem / a.W2, wp = rem - hp * a.W2;
            const bool real = pp < a.P && hp >= 1 && hp <= a.H && wp >= 1 && wp <= a.W;
            if (real && n0 < a.N) {
                const h8_t v = *(const h8_t*)(ct + row * BN + ch * 8);
                long long off;
                if constexpr (EPI == MCAMD_EPI_PAD_F16) off = pp * a.y_ld;
                else off = ((long long)(b * a.H + hp - 1) * a.W + wp - 1) * a.y_ld;
                *(h8_t*)(y + off + a.y_choff + n0) = v;
                if (EPI == MCAMD_EPI_RAW_F16 && a.stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float f = (float)v[e];
                        st1[e] += f;
                        st2[e] += f * f;
                    }
                }
            }
        }
    }

    if (EPI == MCAMD_EPI_RAW_F16 && a.stats) {
        __syncthreads();
        float* red = (float*)smem;   // [16 row groups][2][128]
        const int ch = tid & 15, rg = tid >> 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(rg * 2 + 0) * BN + ch * 8 + e] = st1[e];
            red[(rg * 2 + 1) * BN + ch * 8 + e] = st2[e];
        }
        __syncthreads();
        {
            const int which = tid >> 7, col = tid & 127;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += red[(k * 2 + which) * BN + col];
            a.stats[((long long)pslot * 2 + which) * a.stats_ld + nt * BN + col] = v;
        }
    }
}

static int env_i9(const char* name, int dflt) {
    const char* s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}

bool mcamd_igemm9_ok(int ksize, int stem, int n, int cin_tap, int W, int mode) {
    if (!env_i9("MCAMD_IGEMM9", 0)) return false;   // measured slower than igemm_kernel at 13x13..52x52 (halo-row work): opt-in
    if (ksize != 3 || stem || cin_tap % 64 != 0 || n % 8 != 0 || mode == MCAMD_EPI_NCHW_F32 || mode == MCAMD_EPI_RAW_F32) return false;
    if (n % 128 != 0 && n < 128) return false;
    return W <= env_i9("MCAMD_IGEMM9_MAXW", 26);
}

int mcamd_igemm9_S(int W) { return round_up_int(W + 3, 4); }

int mcamd_igemm9_rows(long long P, int n) {
    int ntiles = (n + 127) / 128;
    int mtiles = (int)((P + 127) / 128);
    int p = env_i9("MCAMD_IGEMM_WGS", 2048) / ntiles;
    if (p < 1) p = 1;
    if (p > mtiles) p = mtiles;
    return p;
}

int mcamd_igemm9_launch(Igemm9Args& a, hipStream_t st) {
    a.num_ntiles = (a.N + 127) / 128;
    a.num_mtiles = (int)(((long long)a.P + 127) / 128);
    a.num_pslots = mcamd_igemm9_rows(a.P, a.N);
    const int R = 128 + 2 * a.S;
    size_t lds = 2 * (size_t)R * 128 + 2 * 16384;
    if (lds < 32768 + 0) lds = 32768;
    const int grid = round_up_int(a.num_pslots, 8) * a.num_ntiles;
#define L9(EPI_)                                                                                                    \
    {                                                                                                               \
        if (lds > 64 * 1024) MCAMD_LDS_OPT_IN(igemm9_kernel<EPI_>, 160 * 1024);                                     \
        hipLaunchKernelGGL((igemm9_kernel<EPI_>), dim3(grid), dim3(256), lds, st, a);                               \
    }
    if (a.mode == MCAMD_EPI_PAD_F16) L9(MCAMD_EPI_PAD_F16) else L9(MCAMD_EPI_RAW_F16)
#undef L9
    MCAMD_LAUNCH_CHECK("igemm9");
    return MCAMD_OK;
}
