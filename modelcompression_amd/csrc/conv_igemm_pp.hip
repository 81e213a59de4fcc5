// Implicit-GEMM convolution, 256x256 "ping-pong" form (forward and dgrad of the wide layers).
//
// Why a second form: the L2 -> LDS operand stream of a CU tops out near 70 GB/s (MI355X_MICROARCH.md,
// "Indexed rows: gather into LDS"), so a tile's MFMA work per staged byte, BM*BN/(BM+BN), decides
// whether the matrix cores or the DMA path is the bound.  128x128 (64) and 192x128 (77) are DMA-bound;
// 256x256 (128) is not -- but it is one workgroup per CU, so nothing outside the workgroup hides its
// stalls.  Here the workgroup hides them itself:
//
//   * 8 waves = 2 (M) x 4 (N), wave tile 128x64 (128 accumulator registers), one wave of each GROUP
//     (waves 0-3 / waves 4-7) per SIMD;
//   * K advances in chunks of 32; a chunk is two barrier-delimited PHASES per wave:
//         memory phase : wait (counted vmcnt) until chunk p+1 has landed, read the 12 fragments of
//                        chunk p, retire them (lgkmcnt 0)
//         s_barrier
//         matrix phase : 16 x v_mfma_f32_32x32x16_f16, with the 4 LDS-DMA instructions of chunk p+3
//                        issued between them (all 8 waves feed the DMA path continuously: 13-17 % faster
//                        than a burst at the start of the memory phase; s_setprio and other placements
//                        of the four instructions measured within 2 %)
//         s_barrier
//   * group 1 executes ONE extra s_barrier before its first phase (and group 0 one after its last), so
//     the groups run half a chunk apart: while one group's waves multiply, the other group's waves on
//     the same SIMDs issue their DMAs and LDS reads.
//   * 4-deep LDS ring of 32 KB stages (128 KB); three chunks are in flight across the barriers.
//
// Ordering rules that make the ring safe (one barrier episode = every wave of the workgroup):
//   RAW  chunk c is read in memory phase c.  Every wave waited for its own DMA pieces of chunk c in
//        memory phase c-1, i.e. before the first barrier of that phase; the read happens after the
//        second barrier of phase c-1.  For the group that runs half a chunk later the same two
//        barriers are one episode further on, still in that order.
//   WAR  the DMA of chunk p+3 overwrites the ring slot of chunk p-1.  Both groups retired their reads
//        of chunk p-1 (lgkmcnt 0) BEFORE the first barrier of their phase p-1; the DMA is issued in
//        matrix phase p, three (group 0) or more barrier episodes later.
// The accumulation order over K equals igemm_kernel's, so raw outputs are bit-identical to it
// (tests/test_kernels_gpu.py::test_pingpong_*).
//
// F8 (round 4, the split-operand forward of the default precision): the K loop's tail is the fp8 CORRECTION part of
// the product.  x*w = x_hi*w_hi + (x_lo*w_hi + x_hi*w_lo) + O(2^-22): the bracket is 2^-11 of the result, so its operands
// need 3-4 bits, not 11 -- e4m3 copies [lo8 | x8] of the activation and [w8 | wlo8] of the weights (common.h: scales),
// multiplied by v_mfma_scale_f32_32x32x64_f8f6f4 INTO THE SAME ACCUMULATORS (the C/D layout is shape-determined), whose
// e8m0 scale operands take the 2^(12 + wexp) back.  A 64-byte LDS row that holds 32 fp16 k's of an fp16 chunk holds 64 e4m3 k's
// of an fp8 chunk; the concatenation of the two k16-step fragments a lane reads from it (bytes [16h, 16h+16) and
// [32+16h, 32+16h+16) of row r) is a valid 32-byte operand of the 64-k instruction because A and B are gathered the same way
// (any k permutation common to both sums the same products; tools/f8_probe.hip).  DMA, swizzle, ring, barriers and the
// fragment reads are byte-identical to the fp16 chunks: only the matrix phase differs -- ONE 32x32x64 fp8 MFMA per block
// instead of two 32x32x16 fp16 ones, the same cycles for twice the k's.  Three products cost 2 instead of 3 units of
// MFMA time and of staged bytes.  Residual: the e4m3 rounding of the bracket's operands, ~4 % of what plain fp16
// operands lose (measured per kernel and per block: DESIGN.md 3d).
//
// Operand addressing, LDS swizzle (on the DMA source), persistent M tiles per N tile, BatchNorm partial
// sums and the epilogues are those of conv_igemm.hip.  Replaces F.conv2d at reference
// src/pruning/weightPruning/layers.py:60-64 and its autograd input gradient.
#include "kernels.h"
#include "epi_pool.h"
#include <stdlib.h>

namespace {

constexpr int BK = 32, NST = 4;
constexpr int WAVES_N = 4;                                    // 2 (M) x 4 (N) waves
constexpr int NT = 512;
constexpr int CPR = BK / 8;                                   // 16-byte chunks per LDS row (64-byte rows)

// MFMA shape of the wave tile: MS = 32 -> v_mfma_f32_32x32x16_f16 (two k16 steps per chunk, 16 accumulator
// registers per block), MS = 16 -> v_mfma_f32_16x16x32_f16 (one k32 step, 4 registers per block; the same flops
// per cycle on paper, a higher sustained clock in practice: MI355X_MICROARCH.md "DVFS give-back" (7)).
template <int MS> struct Shape;
template <> struct Shape<32> {
    typedef f32x16_t acc_t;
    static constexpr int AR = 16, KS = 2;
    __device__ static __forceinline__ int swz4(int row) { return (row >> 2) & 3; }
    __device__ static __forceinline__ int rowof(int r, int lane) { return mfma32_row(r, lane); }
    __device__ static __forceinline__ acc_t mfma(h8_t a, h8_t b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Shape<16> {
    typedef f32x4_t acc_t;
    static constexpr int AR = 4, KS = 1;
    // 64-byte rows read 16 rows x 4 chunks per instruction (lane = row + 16 * chunk): the ds_read_b128 lane groups
    // {0-3,12-15,20-27}, ... hit 16 distinct 16-byte slots when chunk ^= {0,2,3,1}[(row >> 2) & 3]
    __device__ static __forceinline__ int swz4(int row) {
        const int q = (row >> 2) & 3;
        return (((q & 1) ^ (q >> 1)) << 1) | (q >> 1);
    }
    __device__ static __forceinline__ int rowof(int r, int lane) { return (lane >> 4) * 4 + r; }
    __device__ static __forceinline__ acc_t mfma(h8_t a, h8_t b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
// F8 kernels keep the two k16-step fragments of a 64-byte row as ONE 8-register value from the moment they are read: it is
// the 32-byte operand of the 64-k fp8 instruction as it stands, and its halves are the operands of the fp16 instruction
// (built at the use instead -- concatenating two h8_t values in the matrix phase -- hipcc copies every fragment: +56
// registers and spills)
__device__ __forceinline__ i32x8_t read_pair(const char* p0, const char* p1) {
    const i32x4_t lo = *(const i32x4_t*)p0, hi = *(const i32x4_t*)p1;
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ h8_t half_of(i32x8_t v, int s) {
    const i32x4_t q = s ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3);
    return __builtin_bit_cast(h8_t, q);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// BM = 256 (wave tile 128x64) or 192 (wave tile 96x64: 228 tiles instead of 172 for the 13x13 layers at B=64).
// With BM = 192 the A tile is 1.5 DMA instructions per thread: waves 0-3 (= group 0) issue two A pieces, waves 4-7
// one, so the counted waits differ per group.
// BN = 256 or 128 columns (128: the layers with 128 output channels and the dgrads whose 256-column tiles
// would be too few; staged bytes per flop are 1.3x those of 256 columns, still 0.75x those of 128x128).
template <int EPI, int BM, int BN, int MS, bool F8 = false>
__global__ __launch_bounds__(512, 1) void igemm_pp_kernel(IgemmArgs a) {
    static_assert(!F8 || MS == 32, "the fp8 correction chunks are 32x32x64 blocks");
    typedef Shape<MS> SH;
    typedef typename SH::acc_t acc_v;
    constexpr int AR = SH::AR, KS = SH::KS;
    constexpr int WN = BN / WAVES_N, B_SLOTS = BN * CPR, B_IT = B_SLOTS / NT;
    static_assert(BN == 256 || BN == 128, "wave tiles of 64 or 32 columns");
    constexpr int WM = BM / 2, TM = WM / MS, TN = WN / MS;
    constexpr int A_SLOTS = BM * CPR;
    constexpr int A_IT = (A_SLOTS + NT - 1) / NT;             // 2 (the second one only for waves 0-3 when BM = 192)
    constexpr bool RAGGED = A_SLOTS % NT != 0;
    constexpr int DPC = A_IT + B_IT;                          // DMA instructions per chunk of waves 0-3
    constexpr int STAGE_BYTES = (A_SLOTS + B_SLOTS) * 16;     // 32 KB / 28 KB
    static_assert(BM == 256 || BM == 192, "wave tiles of 128 or 96 rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int group = wave >> 2;   // waves 0-3 / 4-7: one of each per SIMD
    // counted waits: "at most n chunks' worth of this wave's DMA instructions still in flight"
    auto wait_chunks = [&](int n) {
        if (RAGGED && group == 1) {
            if (n == 2) wait_vm<2 * (DPC - 1)>();
            else if (n == 1) wait_vm<DPC - 1>();
            else wait_vm<0>();
        } else {
            if (n == 2) wait_vm<2 * DPC>();
            else if (n == 1) wait_vm<DPC>();
            else wait_vm<0>();
        }
    };

    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int nt = jb % a.num_ntiles;
    const int pslot = (jb / a.num_ntiles) * 8 + xcd;
    if (pslot >= a.num_pslots) return;
    const int nchunks = a.ktot / BK;
    const int cpt = a.cin_tap / BK;

    long long bbase[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int slot = it * NT + tid;
        const int row = slot / CPR, phys = slot % CPR;
        bbase[it] = (long long)(nt * BN + row) * a.ktot + (phys ^ SH::swz4(row)) * 8;
    }

    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
    bool sat = false;   // an fp16 output was clamped (reported through a.overflow)

    for (int mt = pslot; mt < a.num_mtiles; mt += a.num_pslots) {
        long long abase[A_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int slot = it * NT + tid;
            const int row = (slot / CPR) % BM, phys = slot % CPR;   // (slots beyond the tile, BM = 192: never issued)
            int m = mt * BM + row;
            if (m > a.M - 1) m = a.M - 1;   // tail rows re-read the last pixel; their results are masked
            int b, h, w;
            if (EPI == MCAMD_EPI_PAD_F16 && a.dst_mode != 0) {   // pooled order: four consecutive rows = one 2x2 window
                pooled_pixel(a, m, b, h, w);
            } else {
                b = m / a.HW;
                const int rem = m - b * a.HW;
                h = rem / a.W;
                w = rem - h * a.W;
            }
            abase[it] = (long long)b * a.x_img_stride + (long long)h * a.x_row_stride + (long long)w * a.x_ld + a.x_off +
                        (phys ^ SH::swz4(row)) * 8;
        }

        acc_v acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

        // Byte pointers of this thread's DMA pieces at chunk 0 (swizzle folded in); a chunk adds a wave-uniform
        // offset: activations tap_off[tap] + channel offset, weights q * BK.
        const char* aptr[A_IT];
        const char* bptr[B_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) aptr[it] = (const char*)(a.x + abase[it]);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) bptr[it] = (const char*)(a.w + bbase[it]);
        // the next chunk to stage, in the packed K order [channel block of a.kb][tap][a.kb channels]:
        // channel offset of its block, tap, offset inside the block, linear chunk index
        int st_cb = 0, st_tap = 0, st_c = 0, st_q = 0;
        auto stage_next = [&]() {
            const int koff = (a.tap_off[st_tap] + (st_cb >= a.wrap ? st_cb - a.wrap : st_cb) + st_c) * 2;   // (x_wrap: hi plane again)
            const int woff = st_q * (BK * 2);
            char* sa = smem + (st_q & (NST - 1)) * STAGE_BYTES;
            char* sb = sa + A_SLOTS * 16;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                if (it * NT + wave * 64 < A_SLOTS) glds16(aptr[it] + koff, sa + (it * NT + wave * 64) * 16);
#pragma unroll
            for (int it = 0; it < B_IT; ++it) glds16(bptr[it] + woff, sb + (it * NT + wave * 64) * 16);
            ++st_q;
            st_c += BK;
            if (st_c == a.kb) {
                st_c = 0;
                if (++st_tap == a.ntaps) {
                    st_tap = 0;
                    st_cb += a.kb;
                }
            }
        };
        // Fragment addresses: row = tile row block + (lane & (MS-1)), so the swizzle term depends on the lane only.
        // MS = 32: lane -> (row, k8 group lane >> 5), the k16 sub-step s flips bit 1 of the chunk index;
        // MS = 16: lane -> (row, chunk lane >> 4), one read covers the whole 64-byte row.
        const int lrow = lane & (MS - 1);
        const int c0 = (lane / MS) ^ SH::swz4(lrow);
        const int a_off0 = ((wm * WM + lrow) * CPR + c0) * 16, a_off1 = ((wm * WM + lrow) * CPR + (c0 ^ 2)) * 16;
        const int b_off0 = A_SLOTS * 16 + ((wn * WN + lrow) * CPR + c0) * 16;
        const int b_off1 = A_SLOTS * 16 + ((wn * WN + lrow) * CPR + (c0 ^ 2)) * 16;

        __syncthreads();   // previous tile's epilogue is done with the LDS (and drained every DMA)
#pragma unroll
        for (int q = 0; q < NST - 1; ++q)
            if (q < nchunks) stage_next();
        // chunk 0 landed for every wave before anybody's phase 0
        wait_chunks(nchunks >= 3 ? 2 : nchunks - 1);
        __builtin_amdgcn_s_barrier();
        if (group == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one episode behind

        // one chunk = one memory phase + one matrix phase; IS8: a chunk of the fp8 correction part (F8 kernels run two loops,
        // fp16 chunks then fp8 chunks: with one loop and a branch around the two kinds of matrix phase hipcc kept two
        // register sets for the accumulators and copied between them)
        // (IS8 is a literal at both call sites and the lambda is inlined: the branch folds.  A generic lambda with a tag type
        // made hipcc drop the kernels' host stubs.)
        auto chunk_phases = [&](const int p, const bool IS8) __attribute__((always_inline)) {
            // ---------------- memory phase: wait for chunk p+1, read the fragments of chunk p ----------------
            __builtin_amdgcn_sched_barrier(0);   // nothing of this phase is scheduled above the barrier that opens it
            const bool more = p + NST - 1 < nchunks;
            const int koff = (a.tap_off[st_tap] + (st_cb >= a.wrap ? st_cb - a.wrap : st_cb) + st_c) * 2;   // (x_wrap: hi plane again)
            const int woff = st_q * (BK * 2);
            char* sa = smem + (st_q & (NST - 1)) * STAGE_BYTES;
            char* sb = sa + A_SLOTS * 16;
            auto issue_piece = [&](int piece) {   // 0..3: A0, A1, B0, B1 of chunk p+3
                if (piece < A_IT) {
                    if (piece * NT + wave * 64 < A_SLOTS) glds16(aptr[piece] + koff, sa + (piece * NT + wave * 64) * 16);
                } else if (piece - A_IT < B_IT) {
                    glds16(bptr[piece - A_IT] + woff, sb + ((piece - A_IT) * NT + wave * 64) * 16);
                }
            };
            // chunk p+1 must have landed; chunk p+2 (issued in matrix phase p-1) may still be in flight
            wait_chunks(p + 2 < nchunks ? 1 : 0);
            const char* sbase = smem + (p & (NST - 1)) * STAGE_BYTES;
            h8_t af[F8 ? 1 : KS][F8 ? 1 : TM], bf[F8 ? 1 : KS][F8 ? 1 : TN];
            i32x8_t af8[F8 ? TM : 1], bf8[F8 ? TN : 1];
            if constexpr (F8) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bf8[j] = read_pair(sbase + b_off0 + j * (MS * CPR * 16), sbase + b_off1 + j * (MS * CPR * 16));
#pragma unroll
                for (int i = 0; i < TM; ++i) af8[i] = read_pair(sbase + a_off0 + i * (MS * CPR * 16), sbase + a_off1 + i * (MS * CPR * 16));
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bf[0][j] = *(const h8_t*)(sbase + b_off0 + j * (MS * CPR * 16));
                    if (KS == 2) bf[KS - 1][j] = *(const h8_t*)(sbase + b_off1 + j * (MS * CPR * 16));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    af[0][i] = *(const h8_t*)(sbase + a_off0 + i * (MS * CPR * 16));
                    if (KS == 2) af[KS - 1][i] = *(const h8_t*)(sbase + a_off1 + i * (MS * CPR * 16));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads retired BEFORE the barrier (WAR rule above)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- matrix phase: 16 MFMAs with the 4 DMA instructions of chunk p+3 between them ----------------
            // (issued here, not in a burst at the start of the memory phase: every wave feeds the DMA path all the
            // time and the issue cost hides behind this wave's own MFMAs)
            __builtin_amdgcn_s_setprio(1);
            bool did8 = false;
            if constexpr (F8) {
                if (IS8) {                 // fp8 correction chunk: one 64-k MFMA per block
                    did8 = true;
                    // e8m0 scales 1 and 2^-(12 + wexp) (a VGPR of four scale bytes each; op_sel picks byte 0).  Inline assembly: through
                    // the builtin hipcc does not accumulate in place (no tied form of the scaled instruction: every block got a
                    // second set of 16 registers and 16 v_mov back -- 256 registers + 254 spilled for the 256 x 256 tile).
                    // Hazards the assembler does not see: none inside the loop (the blocks are independent, the operands come
                    // from LDS reads retired before the barrier, the next reader of an accumulator is an MFMA two barriers
                    // later); the epilogue's VALU reads are padded after the loop.
                    const int SA = 127 * 0x01010101, SB = a.f8_sb;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                                         : "+v"(acc[i][j])
                                         : "v"(af8[i]), "v"(bf8[j]), "v"(SA), "v"(SB));
                            const int m = i * TN + j;                       // the 4 DMA instructions spread over the TM * TN MFMAs
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (more && q * (TM * TN) / 4 == m) issue_piece(q);
                        }
                }
            }
            if (!did8) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (F8) acc[i][j] = SH::mfma(half_of(af8[i], s), half_of(bf8[j], s), acc[i][j]);
                            else acc[i][j] = SH::mfma(af[s][i], bf[s][j], acc[i][j]);
                            const int m = (s * TM + i) * TN + j;            // 0 .. NM-1
                            // the 4 DMA instructions spread over the NM MFMAs, the first one behind the second MFMA
                            // (NM = 6, the 192 x 128 tile on 32 x 32 blocks: one per MFMA from the first -- `m % (NM / 4) == 1`
                            // never fired there and the ring was never refilled; found with the fp8 form, round 4)
                            constexpr int NM = KS * TM * TN;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (more && q * NM / 4 + (NM >= 8 ? 1 : 0) == m) issue_piece(q);
                        }
            }
            __builtin_amdgcn_s_setprio(0);
            if (more) {
                ++st_q;
                st_c += BK;
                if (st_c == a.kb) {
                    st_c = 0;
                    if (++st_tap == a.ntaps) {
                        st_tap = 0;
                        st_cb += a.kb;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        {
            const int p8 = F8 ? (a.f8_from < nchunks ? a.f8_from : nchunks) : nchunks;
            int p = 0;
            for (; p < p8; ++p) chunk_phases(p, false);
            if constexpr (F8)
                for (; p < nchunks; ++p) chunk_phases(p, true);
        }
        if (group == 0) __builtin_amdgcn_s_barrier();   // pairs with group 1's last barrier: both groups aligned again
        if constexpr (F8) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last (assembly) MFMA -> VALU reads of the accumulators

        // ------------------------------- epilogue (as conv_igemm.hip) -------------------------------
        if constexpr (EPI == MCAMD_EPI_NCHW_F32) {
            float* y = (float*)a.y;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < AR; ++r) {
                    const int m = mt * BM + wm * WM + i * MS + SH::rowof(r, lane);
                    if (m < a.M) {
                        const int b = m / a.HW;
                        const int hw = m - b * a.HW;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int n = nt * BN + wn * WN + j * MS + (lane & (MS - 1));
                            if (n < a.N) {
                                float v = acc[i][j][r];
                                if (a.bias) v += a.bias[n];
                                y[((long long)b * a.N + n) * a.HW + hw] = v;
                            }
                        }
                    }
                }
        } else if constexpr (EPI == MCAMD_EPI_RAW_F32) {
            // unrounded accumulators, fp32 [M][y_ld], straight from the registers (conv_igemm.hip); statistics from fp32
            float* y = (float*)a.y;
            const int mlim = a.M - mt * BM;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nt * BN + wn * WN + j * MS + (lane & (MS - 1));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < AR; ++r) {
                        const int row = wm * WM + i * MS + SH::rowof(r, lane);
                        const float v = acc[i][j][r];
                        if (row < mlim) {
                            if (n < a.N) y[(long long)(mt * BM + row) * a.y_ld + a.y_choff + n] = v;
                            s1[j] += v;
                            s2[j] += v * v;
                        }
                    }
            }
        } else {
            __syncthreads();   // every wave is done with the stage buffers
            half_t* ct = (half_t*)smem;   // [BM][BN] fp16 output tile (128 KB)
            const int mlim = a.M - mt * BM;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * WN + j * MS + (lane & (MS - 1));
                float sc = 1.f, sh = 0.f;
                if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                    const int n = nt * BN + col;
                    if (n < a.N) {
                        if (a.scale) sc = a.scale[n];
                        if (a.shift) sh = a.shift[n];
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < AR; ++r) {
                        const int row = wm * WM + i * MS + SH::rowof(r, lane);
                        float v = acc[i][j][r];
                        if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                            v = v * sc + sh;
                            v = v > 0.f ? v : v * a.slope;
                        }
                        const half_t hv = (half_t)fminf(fmaxf(v, -65504.f), 65504.f);
                        sat |= fabsf(v) > 65504.f;
                        ct[row * BN + col] = hv;
                        if (EPI == MCAMD_EPI_RAW_F16 && a.stats) {
                            const float fv = (row < mlim) ? (float)hv : 0.f;
                            s1[j] += fv;
                            s2[j] += fv * fv;
                        }
                    }
            }
            __syncthreads();
            constexpr int CH = BN / 8;
            half_t* y = (half_t*)a.y;
            if (EPI == MCAMD_EPI_PAD_F16 && a.dst_mode != 0) {
                store_pad_pooled<BM, BN, NT>(a, ct, mt, nt, tid);
                continue;
            }
            for (int slot = tid; slot < BM * CH; slot += NT) {
                const int row = slot / CH, ch = slot - row * CH;
                const int m = mt * BM + row;
                const int n0 = nt * BN + ch * 8;
                if (m < a.M && n0 < a.N) {
                    long long off;
                    if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                        const int b = m / a.HW;
                        const int rem = m - b * a.HW;
                        const int h = rem / a.W;
                        const int w = rem - h * a.W;
                        off = (((long long)b * (a.H + 2) + h + 1) * (a.W + 2) + w + 1) * a.y_ld;
                    } else {
                        off = (long long)m * a.y_ld;
                    }
                    *(h8_t*)(y + off + a.y_choff + n0) = *(const h8_t*)(ct + row * BN + ch * 8);
                }
            }
        }
    }

    if (sat && a.overflow) atomicOr(a.overflow, 1);
    if ((EPI == MCAMD_EPI_RAW_F16 || EPI == MCAMD_EPI_RAW_F32) && a.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {   // lanes that hold the same column: l ^ 32 (and l ^ 16 for 16x16 blocks)
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (MS == 16) {
                s1[j] += __shfl_xor(s1[j], 16);
                s2[j] += __shfl_xor(s2[j], 16);
            }
        }
        __syncthreads();
        float* red = (float*)smem;   // [BM/WM][2][BN]
        if (lane < MS) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                red[(wm * 2 + 0) * BN + wn * WN + j * MS + lane] = s1[j];
                red[(wm * 2 + 1) * BN + wn * WN + j * MS + lane] = s2[j];
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += NT) {
            const int which = t / BN, col = t - which * BN;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < BM / WM; ++k) v += red[(k * 2 + which) * BN + col];
            a.stats[((long long)pslot * 2 + which) * a.stats_ld + nt * BN + col] = v;
        }
    }
}

template <int EPI, int BM, int BN, int MS, bool F8 = false>
static void launch_pp(const IgemmArgs& a, int rows, int ntiles, hipStream_t st) {
    constexpr size_t ring = (size_t)NST * (BM + BN) * CPR * 16;   // <= 128 KB: the ring, then the fp16 output tile
    static_assert((size_t)BM * BN * 2 <= ring, "output tile fits the ring");
    MCAMD_LDS_OPT_IN((igemm_pp_kernel<EPI, BM, BN, MS, F8>), ring);
    hipLaunchKernelGGL((igemm_pp_kernel<EPI, BM, BN, MS, F8>), dim3(round_up_int(rows, 8) * ntiles + 8), dim3(NT), ring, st, a);
}

// a.* filled as for mcamd_igemm_launch; the packed weights are padded to 256 rows, cin_tap % 32 == 0, at least one
// chunk.  bm = 256 or 192, bn = 256 or 128.
int mcamd_igemm_pp_launch(const IgemmArgs& a, int bm, int bn, int rows, int ntiles, hipStream_t st) {
    if (a.cin_tap % BK != 0 || a.ktot < BK || a.kb % BK != 0 || (bm != 256 && bm != 192) || (bn != 256 && bn != 128)) {
        mcamd_set_error("igemm_pp: K per tap (%d) must be a multiple of %d, tile (%d x %d) 256|192 x 256|128", a.cin_tap, BK, bm, bn);
        return MCAMD_EINVAL;
    }
    if (a.f8_from != 0x7fffffff) {       // fp8 correction part (mcamd_conv_geom.x_f8): fp32 output with BatchNorm sums only
        if (a.mode != MCAMD_EPI_RAW_F32 || a.f8_from <= 0 || a.f8_from >= a.ktot / BK || a.wrap != 0x7fffffff) {
            mcamd_set_error("igemm_pp: the fp8 correction form needs epilogue mode 3, no x_wrap and 0 < f8_from (%d) < chunks", a.f8_from);
            return MCAMD_EINVAL;
        }
        if (bm == 256 && bn == 256) launch_pp<MCAMD_EPI_RAW_F32, 256, 256, 32, true>(a, rows, ntiles, st);
        else if (bm == 192 && bn == 256) launch_pp<MCAMD_EPI_RAW_F32, 192, 256, 32, true>(a, rows, ntiles, st);
        else if (bm == 256) launch_pp<MCAMD_EPI_RAW_F32, 256, 128, 32, true>(a, rows, ntiles, st);
        else launch_pp<MCAMD_EPI_RAW_F32, 192, 128, 32, true>(a, rows, ntiles, st);
        MCAMD_LAUNCH_CHECK("igemm_pp(f8)");
        return MCAMD_OK;
    }
    const int ms = MCAMD_ENV_INT("MCAMD_PP_MFMA", 16) == 32 ? 32 : 16;
#define PP_SHAPE(EPI_, BM_, BN_)                                           \
    do {                                                                   \
        if (ms == 32) launch_pp<EPI_, BM_, BN_, 32>(a, rows, ntiles, st);  \
        else launch_pp<EPI_, BM_, BN_, 16>(a, rows, ntiles, st);           \
    } while (0)
#define PP_CASE(EPI_)                                                      \
    do {                                                                   \
        if (bm == 256 && bn == 256) PP_SHAPE(EPI_, 256, 256);              \
        else if (bm == 192 && bn == 256) PP_SHAPE(EPI_, 192, 256);         \
        else if (bm == 256) PP_SHAPE(EPI_, 256, 128);                      \
        else PP_SHAPE(EPI_, 192, 128);                                     \
    } while (0)
    if (a.mode == MCAMD_EPI_NCHW_F32) PP_CASE(MCAMD_EPI_NCHW_F32);
    else if (a.mode == MCAMD_EPI_RAW_F32) PP_CASE(MCAMD_EPI_RAW_F32);
    else if (a.mode == MCAMD_EPI_PAD_F16) PP_CASE(MCAMD_EPI_PAD_F16);
    else PP_CASE(MCAMD_EPI_RAW_F16);
#undef PP_CASE
#undef PP_SHAPE
    MCAMD_LAUNCH_CHECK("igemm_pp");
    return MCAMD_OK;
}
