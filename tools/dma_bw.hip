// Microbenchmark: LDS-DMA (global_load_lds_dwordx4) operand-stream rate of an implicit-GEMM workgroup, by the width
// of the contiguous piece fetched per tile row: 64 bytes (K chunks of 32 halfs, the two halves of a 128-byte line
// fetched in consecutive steps) against 128 bytes (K chunks of 64 halfs, whole lines).  8 waves, A rows private
// to the workgroup and streamed once (192 rows x 2 KB per pass), B rows shared by all workgroups (256 rows x 2 KB).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/dma_bw tools/dma_bw.hip ; run: tools/dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int PIECE, bool PAIR = false, bool SWAP = false>   // bytes per row per step: 64 | 128; PAIR: the two 64-byte halves of a line back to back
__global__ __launch_bounds__(512, 1) void dma_kernel(const char* A, const char* B, int passes, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS_A = 192, ROWS_B = 256, STRIDE = 2048, LPR = PIECE / 16;
    constexpr int INSTR_A = ROWS_A * LPR / 64, INSTR_B = ROWS_B * LPR / 64, INSTR = INSTR_A + INSTR_B;
    constexpr int STEPS = STRIDE / PIECE;                       // steps per pass (K = 1024 halfs)
    constexpr int STAGE = (ROWS_A + ROWS_B) * PIECE, NST = 112 * 1024 / STAGE;   // 4 stages of 28 KB | 2 of 56 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* a_wg = A + (size_t)blockIdx.x * passes * ROWS_A * STRIDE;
    int slot = 0;
    for (int p = 0; p < passes; ++p) {
        const char* a_p = a_wg + (size_t)p * ROWS_A * STRIDE;
        if (PAIR) {
            for (int s = 0; s < STEPS; s += 2) {
                char* st0 = smem + slot * STAGE;
                char* st1 = smem + ((slot + 1) % NST) * STAGE;
#pragma unroll
                for (int i = 0; i < INSTR; ++i) {
                    if ((i & 7) != wave) continue;
                    const int piece = (i < INSTR_A ? i : i - INSTR_A) * 64 + lane;
                    const int row = piece / LPR, ch = piece % LPR;
                    const char* src = (i < INSTR_A ? a_p : B) + (size_t)row * STRIDE + s * PIECE + ch * 16;
                    glds16(src, st0 + i * 1024);
                    glds16(src + PIECE, st1 + i * 1024);          // the other half of the same lines, right behind
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                slot = (slot + 2) % NST;
            }
            continue;
        }
        for (int s = 0; s < STEPS; ++s) {
            char* st = smem + slot * STAGE;
#pragma unroll
            for (int ii = 0; ii < INSTR; ++ii) {
                // SWAP: even steps fetch B then A, odd steps A then B: the A lines are re-touched after 24 KB, not 56 KB
                const int i = (SWAP && !(s & 1)) ? (ii + INSTR_A) % INSTR : ii;
                if ((i & 7) != wave) continue;
                const int piece = (i < INSTR_A ? i : i - INSTR_A) * 64 + lane;
                const int row = piece / LPR, ch = piece % LPR;
                const char* src = (i < INSTR_A ? a_p : B) + (size_t)row * STRIDE + s * PIECE + ch * 16;
                glds16(src, st + i * 1024);
            }
            // at most one younger step in flight behind the one being waited for (same bytes in flight in both modes)
            if (PIECE == 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            slot = slot + 1 == NST ? 0 : slot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[blockIdx.x & 1023] == 123) sink[0] = 1;
}

int main() {
    const int grid = 256, passes = 12;
    size_t abytes = (size_t)grid * passes * 192 * 2048, bbytes = 256 * 2048;
    char *A, *B;
    int* sink;
    hipMalloc(&A, abytes), hipMalloc(&B, bbytes), hipMalloc(&sink, 4);
    hipMemset(A, 1, abytes), hipMemset(B, 2, bbytes);
    hipFuncSetAttribute((const void*)dma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    hipFuncSetAttribute((const void*)dma_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)dma_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    hipFuncSetAttribute((const void*)dma_kernel<64, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(dma_kernel<64>, dim3(grid), dim3(512), 112 * 1024, 0, A, B, passes, sink);
            else if (mode == 1) hipLaunchKernelGGL(dma_kernel<128>, dim3(grid), dim3(512), 112 * 1024, 0, A, B, passes, sink);
            else if (mode == 2) hipLaunchKernelGGL((dma_kernel<64, true>), dim3(grid), dim3(512), 112 * 1024, 0, A, B, passes, sink);
            else hipLaunchKernelGGL((dma_kernel<64, false, true>), dim3(grid), dim3(512), 112 * 1024, 0, A, B, passes, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double bytes = (double)grid * passes * (192 + 256) * 2048;
            printf("%s piece %3d B: %.3f ms  %.2f TB/s chip  %.1f GB/s per CU (%s)\n", mode == 2 ? "paired" : mode == 3 ? "swap  " : "plain ", mode == 1 ? 128 : 64, ms, bytes / ms / 1e9,
                   bytes / ms / 1e6 / 256, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
