// 3x3 convolution with the WEIGHTS RESIDENT IN REGISTERS and one activation window per M tile: forward of the layers whose
// packed weight matrix is small (conv3 / conv5 of YOLOv2: 64 -> 128 channels at 104x104, K = 9 x 64 = 576, 147 KB of fp16).
//
// Why: under the counters (tools/pmc_one.sh) the 128x128 implicit GEMM runs these layers at 25 % MFMA
// busy, 2.2 TB/s of HBM traffic and 8.2 TB/s of L2 -> LDS staging -- the LDS-DMA rate for its access pattern.  Half of what
// it stages is the SAME 147 KB weight matrix, once per M tile (5 408 times per launch), the other half is each activation
// line nine times, once per tap.  Here
//   * a wave (2 x 2 waves, wave tile 64 x 64) loads its 64 filter columns x 576 K of weights ONCE into 288 VGPRs as MFMA
//     B fragments and keeps them for the whole kernel (one wave per SIMD, 512 registers);
//   * the M tile is 128 consecutive PADDED pixels (zero halo), so tap t is the constant row shift
//     (ty - 1)(W + 2) + (tx - 1) of ONE LDS window of activation rows [p0 - S, p0 + 128 + S), S >= W + 3: the window is
//     staged once per tile (44 KB at W = 104 against 295 KB staged per tile before), double-buffered across the persistent
//     workgroup's tiles, and the nine taps are ds_read_b128 fragment reads at precomputed lane offsets (the XOR swizzle of a
//     shifted row differs per tap; the offsets do not change from tile to tile);
//   * halo pixels are multiplied too (2 / (W + 2) of the rows: 1.9 % at 104 x 104) and dropped by the store loop, which also
//     gathers the BatchNorm partial sums (deterministic: per-thread sums in tile order, one slab row per workgroup).
// The accumulation order over K ([tap][channel], one 64-channel block) equals igemm_kernel's, so the raw fp16 outputs and
// the partial sums agree with it to the summation order of the statistics slab.
//
// Replaces F.conv2d at reference src/pruning/weightPruning/layers.py:60-64 for these shapes.
#include <type_traits>

#include "kernels.h"
#include "tr_frag.h"

namespace {
constexpr int BM = 128, BN = 128, CPR = 8, NT = 256, RB = 128;   // window rows of 64 channels = 128 bytes = 8 chunks
constexpr int CTP = BN + 8;      // halfs per row of the fp16 output tile in LDS (272 bytes: 8-byte writes of 32 pixels 2-way)

__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

// The fragment reads are inline assembly with hand-counted lgkmcnt (as in tr_frag.h): left to hipcc, every ds_read_b128 of
// the unrolled loop was followed by `s_waitcnt lgkmcnt(0)` and its two MFMAs -- one wave per SIMD, so nothing hid the LDS
// latency.  Here the two reads of step u + 1 are issued before the wait for step u's.
template <int OFF>
__device__ __forceinline__ h8_t lds_read16(unsigned addr) {
    h8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// at most N younger LDS operations in flight; `a`, `b` are results that must have landed (ties the MFMAs below the wait)
// s_waitcnt vmcnt(n) for a wave-uniform n in 0..8 (the instruction takes an immediate)
__device__ __forceinline__ void wait_vm_le(int n) {
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
template <int N>
__device__ __forceinline__ void lds_wait1(h8_t& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(h8_t& a, h8_t& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
}  // namespace

template <int EPI>
__global__ __launch_bounds__(256, 1) void wres_kernel(IgemmArgs a, int S, int P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int W2 = a.W + 2, HW2 = (a.H + 2) * W2;
    const int R = BM + 2 * S;                          // window rows (S is a multiple of 4)
    const int win_bytes = R * RB;
    char* const win0 = smem;                           // two windows, then two fp16 output tiles [128][CTP]
    half_t* const ct = (half_t*)(smem + 2 * win_bytes);

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nt = jb % a.num_ntiles;
    const int pslot = (jb / a.num_ntiles) * 8 + xcd;
    if (pslot >= a.num_pslots) return;

    // ---- weights: B fragments of this wave's 64 columns, all of K, resident for the whole kernel ----
    // packed row n, K index (tap t) * 64 + 16 s + 8 (lane >> 5) .. + 7  (one 64-channel block: K order [tap][channel])
    h8_t wf[9][4][2];
    {
        const half_t* wrow0 = a.w + (long long)(nt * BN + wn * 64 + (lane & 31)) * a.ktot + (lane >> 5) * 8;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) wf[t][s][j] = *(const h8_t*)(wrow0 + (long long)j * 32 * a.ktot + t * 64 + s * 16);
    }
    // ---- fragment read offsets into a window (bytes), sub-step 0; sub-step s flips address bits 5-6: addr ^ (s << 5) ----
    // (the second 32-row block is 32 rows = 4096 bytes further with the same swizzle: an immediate of the read)
    int aoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int row = wm * 64 + (lane & 31) + S + (t / 3 - 1) * W2 + (t % 3 - 1);
        aoff[t] = row * RB + (((lane >> 5) ^ swz8(row)) << 4);
    }
    // ---- window DMA: slot = row * 8 + chunk, lane-linear in LDS; the swizzle goes on the source chunk ----
    const int a_iters = (R * CPR + NT - 1) / NT;
    auto stage = [&](int mt, int buf) {
        const long long p0 = (long long)mt * BM;
        const half_t* xwin = a.x + (p0 - S) * a.x_ld + a.x_off;
        char* sa = win0 + buf * win_bytes;
        for (int it = 0; it < a_iters; ++it) {
            const int wslot = it * NT + wave * 64;
            if (wslot < R * CPR) {
                const int slot = wslot + lane;
                const int row = slot >> 3, phys = slot & 7;
                glds16(xwin + (long long)row * a.x_ld + ((phys ^ swz8(row)) << 3), sa + wslot * 16);
            }
        }
    };

    float st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st1[i] = st2[i] = 0.f;
    float satmax = 0.f;

    // The epilogue rides BETWEEN the MFMAs (one wave per SIMD: nothing else overlaps it; serialised it cost 3.3 us of a
    // tile's 7.6 us).  A tile is multiplied in two PHASES, pixel block i = 0 (rows 0-31 of the wave tile), then i = 1, each
    // 36 k16 steps x 2 MFMAs; while phase i runs, the OTHER block's finished accumulators are converted and written to the LDS
    // output tile, one register quad every fourth step: phase 1 converts this tile's block 0, phase 0 of the NEXT tile
    // converts this tile's block 1.  The store loop of tile k follows phase 0 of tile k + 1; two output tiles in LDS (by tile
    // parity) so that phase 1 may write the next one while a slower wave still reads this one.
    f32x16_t acc[2][2];

    // one register quad of pixel block i: channels col .. col + 3 of pixel prow -> one 8-byte LDS write
    auto convert_quad = [&](const f32x16_t (&pa)[2], int i, int c, half_t* ctile) {
        const int j = c >> 2, q = c & 3;
        const int prow = wm * 64 + i * 32 + (lane & 31);
        const int col = wn * 64 + j * 32 + 8 * q + 4 * (lane >> 5);              // mfma32_row(4 q, lane)
        h4_t hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = pa[j][4 * q + e];
            if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                const int n = nt * BN + col + e;
                const float sc = (n < a.N && a.scale) ? a.scale[n] : 1.f, sh = (n < a.N && a.shift) ? a.shift[n] : 0.f;
                v = v * sc + sh;
                v = v > 0.f ? v : v * a.slope;
            }
            satmax = fmaxf(satmax, fabsf(v));
            hv[e] = (half_t)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
        }
        *(h4_t*)(ctile + prow * CTP + col) = hv;
    };

    // phase I of one tile (window at LDS address sa) into acc[I]; CONV: convert acc[1 - I] into `ctile` on the way
    auto mfma_phase = [&](unsigned sa, auto i_tag, auto conv_tag, half_t* ctile) {
        constexpr int I = decltype(i_tag)::value;
        constexpr bool CONV = decltype(conv_tag)::value;
        // four steps ahead (a step is 2 MFMAs = 64 cycles): 4 reads in flight
        constexpr int D = 4;
        h8_t af[D];
#pragma unroll
        for (int u = 0; u < D; ++u) af[u] = lds_read16<I * 32 * RB>(sa + (aoff[u >> 2] ^ ((u & 3) << 5)));
#pragma unroll
        for (int u = 0; u < 36; ++u) {
            const int t = u >> 2, s = u & 3;
            // reads of steps u .. min(u + D - 1, 35) are in flight: this step's is the oldest
            // (a ct write of the interleaved epilogue is younger than it: the count stays conservative)
            if (36 - u >= D) lds_wait1<D - 1>(af[u % D]);
            else if (36 - u == 3) lds_wait1<2>(af[u % D]);
            else if (36 - u == 2) lds_wait1<1>(af[u % D]);
            else lds_wait1<0>(af[u % D]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (u == 0)      // the first step starts from zero (no accumulator initialisation pass)
                    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t][s][j], af[u % D], f32x16_t{0}, 0, 0, 0);
                else
                    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t][s][j], af[u % D], acc[I][j], 0, 0, 0);
            }
            if (u + D < 36) af[u % D] = lds_read16<I * 32 * RB>(sa + (aoff[(u + D) >> 2] ^ (((u + D) & 3) << 5)));
            if (CONV && u >= 2 && u < 34 && (u & 3) == 2) convert_quad(acc[1 - I], 1 - I, (u - 2) >> 2, ctile);
        }
    };

    // store loop of tile mt: ct -> global (whole 16-byte pieces per (pixel, 8 channels)), halo pixels dropped, BatchNorm sums
    int nstores = 0;          // output store instructions this wave issued AFTER the DMA of the window it will wait for
    auto store_tile = [&](int mt, const half_t* ctile) {
        half_t* y = (half_t*)a.y;
        const int ch = tid & 15;                 // 16-byte chunk (8 channels) this thread always handles
        const int n0 = nt * BN + ch * 8;
        // padded pixel -> (b, hp, wp), only interior pixels are real outputs: ONE 32-bit decomposition per tile, then steps of
        // 16 pixels with carries
        int pp = mt * BM + (tid >> 4);
        int b = pp / HW2;
        int rem = pp - b * HW2;
        int hp = rem / W2, wp = rem - hp * W2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (tid >> 4) + 16 * k;
            const bool real = pp < P && hp >= 1 && hp <= a.H && wp >= 1 && wp <= a.W && n0 < a.N;
            nstores += __builtin_amdgcn_ballot_w64(real) != 0 ? 1 : 0;
            if (real) {
                const h8_t v = *(const h8_t*)(ctile + row * CTP + ch * 8);
                long long off;
                if constexpr (EPI == MCAMD_EPI_PAD_F16) off = (long long)pp * a.y_ld;
                else off = ((long long)(b * a.H + hp - 1) * a.W + wp - 1) * a.y_ld;
                *(h8_t*)(y + off + a.y_choff + n0) = v;
                if (EPI == MCAMD_EPI_RAW_F16 && a.stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        st1[e] += f;
                        st2[e] += f * f;
                    }
                }
            }
            pp += 16, wp += 16;
            while (wp >= W2) wp -= W2, ++hp;      // W2 >= 10: at most two carries
            if (hp >= a.H + 2) hp -= a.H + 2, ++b;
        }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using NO = std::integral_constant<bool, false>;
    using YES = std::integral_constant<bool, true>;
    half_t* const ct0 = ct;
    half_t* const ct1 = ct + BM * CTP;
    int buf = 0, li = 0, prev_mt = -1;
    if (pslot < a.num_mtiles) stage(pslot, 0);
    for (int mt = pslot; mt < a.num_mtiles; mt += a.num_pslots, ++li) {
        // The window of this tile was issued BEFORE the store loop that ran since: wait for it, not for those stores (vmcnt
        // is in order; the stores' round trip to HBM would otherwise be exposed once per tile).  The eval epilogue loads its
        // coefficients in between: plain wait there.
        if (EPI == MCAMD_EPI_PAD_F16) wait_vm_le(0);
        else wait_vm_le(nstores);
        nstores = 0;
        __syncthreads();      // this tile's window landed for every wave; everybody is done with the window before the last
        if (mt + a.num_pslots < a.num_mtiles) stage(mt + a.num_pslots, buf ^ 1);   // the next window lands under this tile's MFMAs
        const unsigned sa = lds_addr_of(win0 + buf * win_bytes);
        buf ^= 1;
        half_t* const ct_cur = (li & 1) ? ct1 : ct0;
        half_t* const ct_prev = (li & 1) ? ct0 : ct1;
        if (li == 0) {
            mfma_phase(sa, I0{}, NO{}, ct_prev);
        } else {
            mfma_phase(sa, I0{}, YES{}, ct_prev);      // ... and block 1 of the previous tile
            __syncthreads();                           // the previous tile's output tile is complete
            store_tile(prev_mt, ct_prev);
        }
        mfma_phase(sa, I1{}, YES{}, ct_cur);           // ... and block 0 of this tile
        prev_mt = mt;
    }
    if (prev_mt >= 0) {       // drain: block 1 of the last tile
        half_t* const ct_last = ((li - 1) & 1) ? ct1 : ct0;
#pragma unroll
        for (int c = 0; c < 8; ++c) convert_quad(acc[1], 1, c, ct_last);
        __syncthreads();
        store_tile(prev_mt, ct_last);
    }

    const bool sat = satmax > 65504.f;
    if (sat && a.overflow) atomicOr(a.overflow, 1);
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
            if (nt * BN + col < a.stats_ld) a.stats[((long long)pslot * 2 + which) * a.stats_ld + nt * BN + col] = v;
        }
    }
}

static int wres_S(int W) { return round_up_int(W + 3, 4); }

// 3x3, ONE 64-channel input block (K = 576), at least 128 output columns, fp16 epilogues; the two windows + the output
// tile must fit the LDS (W <= 160), and the launch must have enough tiles to amortise the weight load (72 fragments).
bool mcamd_wres_ok(int ksize, int stem, int n, int cin_tap, int ktot, int B, int H, int W, int mode) {
    if (MCAMD_ENV_INT("MCAMD_WRES", 1) == 0) return false;
    if (ksize != 3 || stem || cin_tap != 64 || ktot != 576 || n < 128 || n % 8 != 0) return false;
    if (mode != MCAMD_EPI_RAW_F16 && mode != MCAMD_EPI_PAD_F16) return false;
    if (B <= 0 || H <= 0 || W < 8) return false;
    const size_t lds = 2 * (size_t)(BM + 2 * wres_S(W)) * RB + 2 * (size_t)BM * CTP * 2;
    if (lds > 160 * 1024) return false;
    return (long long)B * (H + 2) * (W + 2) >= 256ll * BM * MCAMD_ENV_INT("MCAMD_WRES_MIN_ROUNDS", 4);
}

int mcamd_wres_rows(int n, int B, int H, int W) {
    const long long P = (long long)B * (H + 2) * (W + 2);
    const int ntiles = (n + BN - 1) / BN;
    const long long mtiles = (P + BM - 1) / BM;
    long long p = 256 / ntiles;                      // one workgroup per CU (512 registers per wave), persistent
    if (p < 1) p = 1;
    if (p > mtiles) p = mtiles;
    return (int)p;
}

int mcamd_wres_launch(IgemmArgs& a, int B, hipStream_t st) {
    const int S = wres_S(a.W);
    const long long P = (long long)B * (a.H + 2) * (a.W + 2);
    a.num_ntiles = (a.N + BN - 1) / BN;
    a.num_mtiles = (int)((P + BM - 1) / BM);
    a.num_pslots = mcamd_wres_rows(a.N, B, a.H, a.W);
    const size_t lds = 2 * (size_t)(BM + 2 * S) * RB + 2 * (size_t)BM * CTP * 2;
    const int grid = round_up_int(a.num_pslots, 8) * a.num_ntiles;
    if (a.mode == MCAMD_EPI_PAD_F16) {
        MCAMD_LDS_OPT_IN(wres_kernel<MCAMD_EPI_PAD_F16>, 160 * 1024);
        hipLaunchKernelGGL((wres_kernel<MCAMD_EPI_PAD_F16>), dim3(grid), dim3(NT), lds, st, a, S, (int)P);
    } else {
        MCAMD_LDS_OPT_IN(wres_kernel<MCAMD_EPI_RAW_F16>, 160 * 1024);
        hipLaunchKernelGGL((wres_kernel<MCAMD_EPI_RAW_F16>), dim3(grid), dim3(NT), lds, st, a, S, (int)P);
    }
    MCAMD_LAUNCH_CHECK("wres");
    return MCAMD_OK;
}
