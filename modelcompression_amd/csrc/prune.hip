// Pruning kernels for gfx950: global magnitude order statistic + mask (weight_prune,
// reference methods.py:9-26) and per-filter scores in numpy's float32 summation order
// (quick_filter_prune, methods.py:28-78).  Pure HBM scans: 4 bytes read per weight per pass.
//
// This file is compiled with -ffp-contract=off: every multiply and add below is a separately
// rounded IEEE fp32 operation (numpy squares, then sums), and division / sqrt are the
// correctly rounded forms, so the scores are bit-identical to the reference's numpy result.
#include "common.h"

// ------------------------------------------------------------------------------------
// k-th smallest |w|: 3-pass radix select on the 31-bit magnitude (11 + 11 + 9 bits)
// ------------------------------------------------------------------------------------
struct SelectState {
    unsigned long long k_rem;  // rank still to resolve inside the current prefix
    unsigned prefix;           // magnitude bits resolved so far
    unsigned pad;
};

__device__ __forceinline__ void pass_bits(int pass, int& shift, int& nbits, int& prefix_shift) {
    if (pass == 0) { shift = 20; nbits = 11; prefix_shift = 31; }
    else if (pass == 1) { shift = 9; nbits = 11; prefix_shift = 20; }
    else { shift = 0; nbits = 9; prefix_shift = 9; }
}

__global__ __launch_bounds__(256) void select_hist_kernel(const float* w, long long n, const SelectState* st, int pass,
                                                          unsigned* hist) {
    __shared__ unsigned lh[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    int shift, nbits, pshift;
    pass_bits(pass, shift, nbits, pshift);
    const unsigned prefix = st->prefix;
    const unsigned binmask = (1u << nbits) - 1u;
    const unsigned* u = (const unsigned*)w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        unsigned key = u[i] & 0x7fffffffu;
        bool match = pass == 0 ? true : (key >> pshift) == prefix;
        if (match) atomicAdd(&lh[(key >> shift) & binmask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << nbits); i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

__global__ void select_scan_kernel(unsigned* hist, SelectState* st, int pass, float* out) {
    // one thread: 2048 bins at most
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int shift, nbits, pshift;
    pass_bits(pass, shift, nbits, pshift);
    unsigned long long k = st->k_rem, cum = 0;
    int bin = (1 << nbits) - 1;
    for (int i = 0; i < (1 << nbits); ++i) {
        unsigned long long c = hist[i];
        if (cum + c > k) {
            bin = i;
            break;
        }
        cum += c;
    }
    for (int i = 0; i < (1 << nbits); ++i) hist[i] = 0;
    st->k_rem = k - cum;
    st->prefix = (st->prefix << nbits) | (unsigned)bin;
    if (pass == 2) *(unsigned*)out = st->prefix;
}

__global__ void select_init_kernel(SelectState* st, unsigned long long k, unsigned* hist) {
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) {
        st->k_rem = k;
        st->prefix = 0;
    }
}

extern "C" size_t mcamd_kth_magnitude_workspace_bytes(void) { return 2048 * sizeof(unsigned) + sizeof(SelectState); }

extern "C" int mcamd_kth_magnitude(const float* const* ptrs, const int64_t* counts, int32_t nseg, int64_t k,
                                   float* out2, void* workspace, size_t workspace_bytes, void* stream) {
    MCAMD_REQUIRE(ptrs && counts && nseg > 0 && out2 && workspace, "kth_magnitude: null argument");
    if (workspace_bytes < mcamd_kth_magnitude_workspace_bytes()) {
        mcamd_set_error("kth_magnitude: workspace too small");
        return MCAMD_EWORKSPACE;
    }
    long long total = 0;
    for (int s = 0; s < nseg; ++s) {
        MCAMD_REQUIRE(counts[s] >= 0 && (counts[s] == 0 || ptrs[s]), "kth_magnitude: bad segment %d", s);
        total += counts[s];
    }
    MCAMD_REQUIRE(k >= 0 && k < total, "kth_magnitude: k=%lld out of range for %lld elements", (long long)k, total);
    hipStream_t st = (hipStream_t)stream;
    unsigned* hist = (unsigned*)workspace;
    SelectState* state = (SelectState*)(hist + 2048);
    for (int which = 0; which < 2; ++which) {
        long long kk = k + which;
        if (kk > total - 1) kk = total - 1;
        hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, st, state, (unsigned long long)kk, hist);
        for (int pass = 0; pass < 3; ++pass) {
            for (int s = 0; s < nseg; ++s) {
                if (counts[s] == 0) continue;
                long long g = (counts[s] + 256 * 16 - 1) / (256 * 16);
                if (g > 2048) g = 2048;
                hipLaunchKernelGGL(select_hist_kernel, dim3((int)g), dim3(256), 0, st, ptrs[s], (long long)counts[s],
                                   (const SelectState*)state, pass, hist);
            }
            hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(64), 0, st, hist, state, pass, out2 + which);
        }
    }
    MCAMD_LAUNCH_CHECK("kth_magnitude");
    return MCAMD_OK;
}

__global__ __launch_bounds__(256) void magnitude_mask_kernel(const float* w, long long n, const float* thr, float* mask) {
    const float t = *thr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        mask[i] = fabsf(w[i]) > t ? 1.f : 0.f;
}

extern "C" int mcamd_magnitude_mask(const float* w, int64_t n, const float* threshold, float* mask, void* stream) {
    MCAMD_REQUIRE(w && threshold && mask && n >= 0, "magnitude_mask: bad argument");
    if (n == 0) return MCAMD_OK;
    long long g = (n + 256 * 8 - 1) / (256 * 8);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(magnitude_mask_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, w, (long long)n, threshold,
                       mask);
    MCAMD_LAUNCH_CHECK("magnitude_mask");
    return MCAMD_OK;
}

// ------------------------------------------------------------------------------------
// filter scores
// ------------------------------------------------------------------------------------
// numpy pairwise_sum_FLOAT over f(p[i*stride]), f = square (SQ) or identity.
template <bool SQ>
__device__ __forceinline__ float elem(const float* p, long long i, long long stride) {
    float v = p[i * stride];
    return SQ ? v * v : v;
}

template <bool SQ>
__device__ float pw_block(const float* p, int n, long long stride) {  // n <= 128
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = r + elem<SQ>(p, i, stride);
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = elem<SQ>(p, j, stride);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + elem<SQ>(p, i + j, stride);
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + elem<SQ>(p, i, stride);
    return res;
}

template <bool SQ, int DEPTH>
__device__ float pw_sum(const float* p, int n, long long stride) {
    if (n <= 128) return pw_block<SQ>(p, n, stride);
    if constexpr (DEPTH == 0) {
        return pw_block<SQ>(p, n, stride);  // unreachable for n <= 128 << MAXDEPTH (checked on the host)
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        float a = pw_sum<SQ, DEPTH - 1>(p, n2, stride);
        float b = pw_sum<SQ, DEPTH - 1>(p + (long long)n2 * stride, n - n2, stride);
        return a + b;
    }
}
#define PW_MAXDEPTH 8  // n <= 32768

// KK > 1: partial[o][t] = sequential sum over cin of w[o][c][t]^2.  One thread per (o, t).
__global__ void filter_partial_kernel(const float* w, int O, int I, int KK, float* partial) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= O * KK) return;
    int o = idx / KK, t = idx - o * KK;
    const float* p = w + (long long)o * I * KK + t;
    float v0 = p[0];
    float acc = v0 * v0;
    for (int c = 1; c < I; ++c) {
        float v = p[(long long)c * KK];
        acc = acc + v * v;
    }
    partial[idx] = acc;
}

// mean square per filter: KK > 1 combines the partials (sequential over kh, then kw);
// KK == 1 does the pairwise sum over the contiguous cin axis.
__global__ void filter_meansq_kernel(const float* w, const float* partial, int O, int I, int kh, int kw, float* ms) {
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    float s;
    if (kh * kw == 1) {
        s = pw_sum<true, PW_MAXDEPTH>(w + (long long)o * I, I, 1);
    } else {
        const float* p = partial + (long long)o * kh * kw;
        s = 0.f;
        for (int x = 0; x < kw; ++x) {
            float col = p[x];
            for (int y = 1; y < kh; ++y) col = col + p[y * kw + x];
            s = (x == 0) ? col : s + col;
        }
    }
    ms[o] = s / (float)(I * kh * kw);
}

// scores[o] = (ms[o] / sqrt(pairwise_sum(ms^2))) / max(...).  One block.
__global__ void filter_normalize_kernel(const float* ms, int O, float* scores) {
    __shared__ float sh_norm, sh_max;
    __shared__ float red[256];
    if (threadIdx.x == 0) {
        float ss = pw_sum<true, PW_MAXDEPTH>(ms, O, 1);
        sh_norm = sqrtf(ss);
    }
    __syncthreads();
    const float norm = sh_norm;
    float mx = -INFINITY;
    for (int o = threadIdx.x; o < O; o += 256) {
        float v = ms[o] / norm;
        scores[o] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) sh_max = red[0];
    __syncthreads();
    const float m = sh_max;
    for (int o = threadIdx.x; o < O; o += 256) scores[o] = scores[o] / m;
}

extern "C" size_t mcamd_filter_scores_workspace_bytes(int32_t cout) { return (size_t)cout * 10 * sizeof(float); }

static int filter_args_ok(const float* w, int32_t cout, int32_t cin, int32_t kh, int32_t kw, const void* out,
                          const void* workspace, size_t workspace_bytes, const char* what) {
    MCAMD_REQUIRE(w && out && workspace && cout > 0 && cin > 0 && kh > 0 && kw > 0, "%s: bad argument", what);
    MCAMD_REQUIRE(kh * kw <= 9, "%s: kernel larger than 3x3 unsupported", what);
    MCAMD_REQUIRE(cin <= (128 << PW_MAXDEPTH) && cout <= (128 << PW_MAXDEPTH), "%s: tensor too large", what);
    if (workspace_bytes < mcamd_filter_scores_workspace_bytes(cout)) {
        mcamd_set_error("%s: workspace too small", what);
        return MCAMD_EWORKSPACE;
    }
    return MCAMD_OK;
}

static void launch_mean_square(const float* w, int cout, int cin, int kh, int kw, float* partial, float* ms,
                               hipStream_t st) {
    int KK = kh * kw;
    if (KK > 1) {
        int total = cout * KK;
        hipLaunchKernelGGL(filter_partial_kernel, dim3((total + 63) / 64), dim3(64), 0, st, w, cout, cin, KK, partial);
    }
    hipLaunchKernelGGL(filter_meansq_kernel, dim3((cout + 63) / 64), dim3(64), 0, st, w, (const float*)partial, cout, cin,
                       kh, kw, ms);
}

extern "C" int mcamd_filter_mean_square(const float* w_oihw, int32_t cout, int32_t cin, int32_t kh, int32_t kw,
                                        float* mean_sq, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = filter_args_ok(w_oihw, cout, cin, kh, kw, mean_sq, workspace, workspace_bytes, "filter_mean_square");
    if (rc) return rc;
    launch_mean_square(w_oihw, cout, cin, kh, kw, (float*)workspace, mean_sq, (hipStream_t)stream);
    MCAMD_LAUNCH_CHECK("filter_mean_square");
    return MCAMD_OK;
}

extern "C" int mcamd_filter_scores(const float* w_oihw, int32_t cout, int32_t cin, int32_t kh, int32_t kw, float* scores,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    int rc = filter_args_ok(w_oihw, cout, cin, kh, kw, scores, workspace, workspace_bytes, "filter_scores");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* ms = partial + (size_t)cout * 9;
    launch_mean_square(w_oihw, cout, cin, kh, kw, partial, ms, st);
    hipLaunchKernelGGL(filter_normalize_kernel, dim3(1), dim3(256), 0, st, (const float*)ms, cout, scores);
    MCAMD_LAUNCH_CHECK("filter_scores");
    return MCAMD_OK;
}

__global__ __launch_bounds__(256) void filter_mask_kernel(const int* keep, long long per_filter, long long n, float* mask) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        mask[i] = keep[i / per_filter] ? 1.f : 0.f;
}

extern "C" int mcamd_filter_mask(const int32_t* keep, int32_t cout, int64_t per_filter, float* mask, void* stream) {
    MCAMD_REQUIRE(keep && mask && cout > 0 && per_filter > 0, "filter_mask: bad argument");
    long long n = (long long)cout * per_filter;
    long long g = (n + 256 * 8 - 1) / (256 * 8);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(filter_mask_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, keep, (long long)per_filter, n,
                       mask);
    MCAMD_LAUNCH_CHECK("filter_mask");
    return MCAMD_OK;
}

__global__ __launch_bounds__(256) void count_zeros_kernel(const float* w, long long n, unsigned long long* out) {
    unsigned long long c = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        c += (w[i] == 0.f) ? 1ull : 0ull;
    __shared__ unsigned long long red[256];
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0]) atomicAdd(out, red[0]);
}

extern "C" int mcamd_count_zeros(const float* w, int64_t n, unsigned long long* out, void* stream) {
    MCAMD_REQUIRE(w && out && n >= 0, "count_zeros: bad argument");
    if (n == 0) return MCAMD_OK;
    long long g = (n + 256 * 16 - 1) / (256 * 16);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(count_zeros_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, w, (long long)n, out);
    MCAMD_LAUNCH_CHECK("count_zeros");
    return MCAMD_OK;
}

__global__ __launch_bounds__(256) void masked_residual_kernel(const float* w, const float* mask, long long n, float* out) {
    float c = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        c += fabsf(w[i] * fabsf(mask[i] - 1.f));  // |.|: zero iff every term is zero (no cancellation)
    __shared__ float red[256];
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && red[0] != 0.f) atomicAdd(out, red[0]);
}

extern "C" int mcamd_masked_residual(const float* w, const float* mask, int64_t n, float* out, void* stream) {
    MCAMD_REQUIRE(w && mask && out && n >= 0, "masked_residual: bad argument");
    if (n == 0) return MCAMD_OK;
    long long g = (n + 256 * 16 - 1) / (256 * 16);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(masked_residual_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, w, mask, (long long)n, out);
    MCAMD_LAUNCH_CHECK("masked_residual");
    return MCAMD_OK;
}
