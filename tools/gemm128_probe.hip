// Probe for the next tile form: C[M][N] = A[M][K] B[N][K]^T in fp16 -> fp32 with a 256 x 256
// workgroup tile on FOUR waves (one per SIMD), wave tile 128 x 128 = 16 accumulator blocks in the 256 AGPRs, the MFMAs as
// inline assembly accumulating in place ("+a").  Per K chunk of 32 a wave reads 16 fragments for 32 MFMAs (the ping-pong
// kernel of libmcamd: 10 for 12), so the CU's LDS port carries 768 cycles per chunk against 1 024 MFMA cycles per SIMD.
// One wave per SIMD has nobody to hide behind: the reads of chunk p+1 and the LDS-DMA of chunk p+3 are issued between the
// MFMAs of chunk p (fragments double-buffered in VGPRs), one barrier per chunk.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/gemm128_probe tools/gemm128_probe.hip && gpurun -- ./tools/gemm128_probe
// -DBUFLDS (second session of round 4): the LDS-DMA pieces as `buffer_load_dwordx4 ... offen lds` (SGPR resource + a fixed 32-bit
// lane offset + the chunk offset as the scalar offset) instead of global_load_lds_dwordx4 with 64-bit lane addresses: 153-166 us
// per launch either way on one box (three timed repetitions each, two runs) -- the issue cost of a piece is not its address
// arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, BK = 32, NST = 4, CPR = 4;
constexpr int STAGE_BYTES = (BM + BN) * CPR * 16;      // 32 KB
constexpr int NT = 256;

__device__ __forceinline__ int swz4(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// -DBUFLDS: the same copy as buffer_load_dwordx4 ... lds -- base in an SGPR resource, a 32-bit per-lane offset that never
// changes, the chunk offset as the instruction's scalar offset: no 64-bit address arithmetic per piece
#ifdef BUFLDS
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
#endif
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16_t& acc, h8_t a, h8_t b) {
    if (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm128_kernel(const half_t* A, const half_t* B, float* C,
                                                                                                 int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
    const int nchunks = K / BK;

    // DMA sources: slot = it * 256 + tid -> (row, physical 16-byte chunk); the swizzle is applied on the source side
    const char* asrc[4];
    const char* bsrc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int slot = it * NT + tid, row = slot / CPR, phys = slot % CPR;
        asrc[it] = (const char*)(A + (long long)(mt * BM + row) * K + ((phys ^ swz4(row)) * 8));
        bsrc[it] = (const char*)(B + (long long)(nt * BN + row) * K + ((phys ^ swz4(row)) * 8));
    }
#ifdef BUFLDS
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    int aoff[4], boff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) aoff[it] = (int)(asrc[it] - (const char*)A), boff[it] = (int)(bsrc[it] - (const char*)B);
#endif
    auto stage = [&](int q) {
        char* sa = smem + (q & (NST - 1)) * STAGE_BYTES;
        char* sb = sa + BM * CPR * 16;
#ifdef BUFLDS
#pragma unroll
        for (int it = 0; it < 4; ++it) blds16(ra, aoff[it], q * (BK * 2), sa + (it * NT + wave * 64) * 16);
#pragma unroll
        for (int it = 0; it < 4; ++it) blds16(rb, boff[it], q * (BK * 2), sb + (it * NT + wave * 64) * 16);
#else
#pragma unroll
        for (int it = 0; it < 4; ++it) glds16(asrc[it] + q * (BK * 2), sa + (it * NT + wave * 64) * 16);
#pragma unroll
        for (int it = 0; it < 4; ++it) glds16(bsrc[it] + q * (BK * 2), sb + (it * NT + wave * 64) * 16);
#endif
    };
    // fragment addresses: lane -> (row lane & 31, k8 group lane >> 5); k16 step s flips bit 1 of the chunk index
    const int lrow = lane & 31, c0 = (lane >> 5) ^ swz4(lrow);
    const int a_off0 = ((wm * 128 + lrow) * CPR + c0) * 16, a_off1 = ((wm * 128 + lrow) * CPR + (c0 ^ 2)) * 16;
    const int b_off0 = BM * CPR * 16 + ((wn * 128 + lrow) * CPR + c0) * 16;
    const int b_off1 = BM * CPR * 16 + ((wn * 128 + lrow) * CPR + (c0 ^ 2)) * 16;

    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    h8_t fa[2][2][4], fb[2][2][4];   // [buffer][k16 step][block]
    auto read_frag = [&](int buf, int idx, const char* sbase) {     // idx 0..15: A s0 i, A s1 i, B s0 j, B s1 j
        const int blk = idx & 3, s = (idx >> 2) & 1;
        if (idx < 8) fa[buf][s][blk] = *(const h8_t*)(sbase + (s ? a_off1 : a_off0) + blk * (32 * CPR * 16));
        else fb[buf][s][blk] = *(const h8_t*)(sbase + (s ? b_off1 : b_off0) + blk * (32 * CPR * 16));
    };

    stage(0);
    if (nchunks > 1) stage(1);
    if (nchunks > 2) stage(2);
    if (nchunks > 2) wait_vm<16>(); else if (nchunks > 1) wait_vm<8>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) read_frag(0, idx, smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    auto iter = [&](const int p, const int cur) __attribute__((always_inline)) {
        const int nxt = cur ^ 1;
#ifdef NODMA
        const bool have_next = p + 1 < nchunks, more = false;
#else
        const bool have_next = p + 1 < nchunks, more = p + 3 < nchunks;
#endif
        // chunk p+1: this wave's pieces have landed (chunk p+2 may still be in flight), then everybody's
        if (p + 2 < nchunks) wait_vm<8>(); else wait_vm<0>();
#ifndef NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        const char* snext = smem + ((p + 1) & (NST - 1)) * STAGE_BYTES;
        char* sa = smem + ((p + 3) & (NST - 1)) * STAGE_BYTES;
        char* sb = sa + BM * CPR * 16;
        const int koff = (p + 3) * (BK * 2);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const int s = m >> 4, i = (m >> 2) & 3, j = m & 3;
            mfma<true>(acc[i][j], fa[cur][s][i], fb[cur][s][j]);
#ifndef NOREAD
            if (have_next && (m & 1) == 0) read_frag(nxt, m >> 1, snext);
#endif
            if (more && (m & 3) == 1) {
                const int piece = m >> 2;      // 0..7
#ifdef BUFLDS
                if (piece < 4) blds16(ra, aoff[piece], koff, sa + (piece * NT + wave * 64) * 16);
                else blds16(rb, boff[piece - 4], koff, sb + ((piece - 4) * NT + wave * 64) * 16);
#else
                if (piece < 4) glds16(asrc[piece] + koff, sa + (piece * NT + wave * 64) * 16);
                else glds16(bsrc[piece - 4] + koff, sb + ((piece - 4) * NT + wave * 64) * 16);
#endif
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int p = 0;
    for (; p + 1 < nchunks; p += 2) {
        iter(p, 0);
        iter(p + 1, 1);
    }
    if (p < nchunks) iter(p, 0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // last (assembly) MFMA -> accumulator reads

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * BM + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int col = nt * BN + wn * 128 + j * 32 + (lane & 31);
                C[(long long)row * N + col] = acc[i][j][r];
            }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 4608;
    if (M % BM || N % BN || K % BK) { printf("M %% 256, N %% 256, K %% 32\n"); return 1; }
    std::vector<half_t> hA((size_t)M * K), hB((size_t)N * K);
    srand(3);
    for (auto& v : hA) v = (half_t)((rand() % 2001 - 1000) * 1e-3f);
    for (auto& v : hB) v = (half_t)((rand() % 2001 - 1000) * 2e-5f);
    half_t *dA, *dB; float* dC;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, (size_t)M * N * 4));
    const size_t lds = (size_t)NST * STAGE_BYTES;
    CK(hipFuncSetAttribute((const void*)gemm128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (M / BM) * (N / BN);
    gemm128_kernel<<<grid, NT, lds>>>(dA, dB, dC, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4096; ++t) {
        const int r = rand() % M, c = rand() % N;
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)(float)hA[(size_t)r * K + k] * (double)(float)hB[(size_t)c * K + k];
        const double e = fabs(hC[(size_t)r * N + c] - s) / (fabs(s) + 1e-6);
        if (e > worst) worst = e;
    }
    printf("gemm128: M %d N %d K %d, %d workgroups; worst relative error of 4096 sampled outputs %.2e\n", M, N, K, grid, worst);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) gemm128_kernel<<<grid, NT, lds>>>(dA, dB, dC, M, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gemm128: %.1f us per launch, %.1f TFLOP/s\n", ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20) * 1e-9);
    }
    return 0;
}
