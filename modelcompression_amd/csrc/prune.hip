// Pruning kernels for gfx950: global magnitude order statistic + mask (weight_prune,
// reference methods.py:9-26) and per-filter scores in numpy's float32 summation order
// (quick_filter_prune, methods.py:28-78).  Pure HBM scans: 4 bytes read per weight per pass.
//
// This file is compiled with -ffp-contract=off: every multiply and add below is a separately
// rounded IEEE fp32 operation (numpy squares, then sums), and division / sqrt are the
// correctly rounded forms, so the scores are bit-identical to the reference's numpy result.
#include "common.h"
#include <string.h>

// ------------------------------------------------------------------------------------
// k-th smallest |w|: 3-pass radix select on the 31-bit magnitude (11 + 11 + 9 bits)
// ------------------------------------------------------------------------------------
// Both order statistics np.percentile interpolates between (ranks k and k+1) are resolved by the SAME three
// scans: while their prefixes agree one histogram serves both, afterwards each keeps its own.  All weight tensors
// are scanned by one launch per pass through a segment table passed by value (one launch per tensor and pass was
// 138 launches of mostly tiny grids: 0.37 ms per pass, 7 % of the HBM rate).
constexpr int SEL_MAXSEG = 128;
constexpr int SEL_CHUNK = 4096;          // elements per work item (16 KB: whole float4 loads, tensors are >= 16-byte aligned)
struct SegTable {
    const float* p[SEL_MAXSEG];
    long long count[SEL_MAXSEG];
    int chunk0[SEL_MAXSEG + 1];          // first work item of segment s; chunk0[nseg] = total
    int nseg;
};
struct SelectState {
    unsigned long long k_rem[2];  // rank still to resolve inside the current prefix, for k and k+1
    unsigned prefix[2];           // magnitude bits resolved so far
};

__device__ __forceinline__ void pass_bits(int pass, int& shift, int& nbits, int& prefix_shift) {
    if (pass == 0) { shift = 20; nbits = 11; prefix_shift = 31; }
    else if (pass == 1) { shift = 9; nbits = 11; prefix_shift = 20; }
    else { shift = 0; nbits = 9; prefix_shift = 9; }
}

__global__ __launch_bounds__(256) void select_hist_kernel(SegTable t, const SelectState* st, int pass, unsigned* hist) {
    // two copies of each histogram, picked by lane parity: halves the same-address serialisation of the LDS atomics
    // (trained / initialised weights fall into a few dozen exponent bins)
    __shared__ unsigned lh[2][2][2048];
    for (int i = threadIdx.x; i < 2 * 2 * 2048; i += 256) (&lh[0][0][0])[i] = 0;
    __syncthreads();
    int shift, nbits, pshift;
    pass_bits(pass, shift, nbits, pshift);
    const unsigned p0 = st->prefix[0], p1 = st->prefix[1];
    const bool same = p0 == p1;
    const unsigned binmask = (1u << nbits) - 1u;
    const int cp = threadIdx.x & 1;
    const int total = t.chunk0[t.nseg];
    for (int item = blockIdx.x; item < total; item += gridDim.x) {
        int s = 0;
        while (item >= t.chunk0[s + 1]) ++s;                  // <= 64 entries, wave-uniform
        const long long off = (long long)(item - t.chunk0[s]) * SEL_CHUNK;
        const long long left = t.count[s] - off;
        const unsigned* u = (const unsigned*)t.p[s] + off;
        auto tally = [&](unsigned raw) {
            const unsigned key = raw & 0x7fffffffu;
            const unsigned bin = (key >> shift) & binmask;
            if (pass == 0) {
                atomicAdd(&lh[0][cp][bin], 1u);
            } else {
                const unsigned pre = key >> pshift;
                if (pre == p0) atomicAdd(&lh[0][cp][bin], 1u);
                if (!same && pre == p1) atomicAdd(&lh[1][cp][bin], 1u);
            }
        };
        if (left >= SEL_CHUNK && (((size_t)u) & 15) == 0) {
#pragma unroll
            for (int r = 0; r < SEL_CHUNK / (256 * 4); ++r) {
                const uint4 v = *(const uint4*)(u + (r * 256 + threadIdx.x) * 4);
                tally(v.x), tally(v.y), tally(v.z), tally(v.w);
            }
        } else {
            const int n = left < SEL_CHUNK ? (int)left : SEL_CHUNK;
            for (int i = threadIdx.x; i < n; i += 256) tally(u[i]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * (1 << nbits); i += 256) {
        const int which = i >> nbits, bin = i & ((1 << nbits) - 1);
        const unsigned c = lh[which][0][bin] + lh[which][1][bin];
        if (c) atomicAdd(&hist[which * 2048 + bin], c);
    }
}

// One block: locate the bins of both ranks, advance the state, clear the histograms for the next pass.
__global__ __launch_bounds__(256) void select_scan_kernel(unsigned* hist, SelectState* st, int pass, float* out2) {
    __shared__ unsigned long long part[2][256];
    __shared__ unsigned long long base[2][256];
    int shift, nbits, pshift;
    pass_bits(pass, shift, nbits, pshift);
    const int nb = 1 << nbits, per = (nb + 255) / 256;
    const bool same = pass == 0 || st->prefix[0] == st->prefix[1];
    const int tid = threadIdx.x;
    for (int which = 0; which < 2; ++which) {
        const unsigned* h = hist + (same ? 0 : which) * 2048;
        unsigned long long s = 0;
        for (int i = tid * per; i < (tid + 1) * per && i < nb; ++i) s += h[i];
        part[which][tid] = s;
    }
    __syncthreads();
    if (tid < 2) {
        unsigned long long c = 0;
        for (int i = 0; i < 256; ++i) {
            base[tid][i] = c;
            c += part[tid][i];
        }
    }
    __syncthreads();
    __shared__ unsigned newbin[2];
    __shared__ unsigned long long newrem[2];
    if (tid < 2) newbin[tid] = nb - 1, newrem[tid] = 0;
    __syncthreads();
    for (int which = 0; which < 2; ++which) {
        const unsigned* h = hist + (same ? 0 : which) * 2048;
        const unsigned long long k = st->k_rem[which];
        unsigned long long cum = base[which][tid];
        if (k >= cum && k < cum + part[which][tid]) {       // the rank falls into this thread's run of bins
            for (int i = tid * per; i < (tid + 1) * per && i < nb; ++i) {
                const unsigned long long c = h[i];
                if (cum + c > k) {
                    newbin[which] = (unsigned)i;
                    newrem[which] = k - cum;
                    break;
                }
                cum += c;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * 2048; i += 256) hist[i] = 0;
    if (tid < 2) {
        st->k_rem[tid] = newrem[tid];
        const unsigned pre = (st->prefix[tid] << nbits) | newbin[tid];
        st->prefix[tid] = pre;
        if (pass == 2) ((unsigned*)out2)[tid] = pre;
    }
}

__global__ void select_init_kernel(SelectState* st, unsigned long long k0, unsigned long long k1, unsigned* hist) {
    for (int i = threadIdx.x; i < 2 * 2048; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) {
        st->k_rem[0] = k0, st->k_rem[1] = k1;
        st->prefix[0] = st->prefix[1] = 0;
    }
}

extern "C" size_t mcamd_kth_magnitude_workspace_bytes(void) { return 2 * 2048 * sizeof(unsigned) + sizeof(SelectState); }

extern "C" int mcamd_kth_magnitude(const float* const* ptrs, const int64_t* counts, int32_t nseg, int64_t k,
                                   float* out2, void* workspace, size_t workspace_bytes, void* stream) {
    MCAMD_REQUIRE(ptrs && counts && nseg > 0 && out2 && workspace, "kth_magnitude: null argument");
    if (workspace_bytes < mcamd_kth_magnitude_workspace_bytes()) {
        mcamd_set_error("kth_magnitude: workspace too small");
        return MCAMD_EWORKSPACE;
    }
    MCAMD_REQUIRE(nseg <= SEL_MAXSEG, "kth_magnitude: at most %d tensors per call (got %d)", SEL_MAXSEG, nseg);
    SegTable t;
    memset(&t, 0, sizeof(t));
    long long total = 0, chunks = 0;
    int ns = 0;
    for (int s = 0; s < nseg; ++s) {
        MCAMD_REQUIRE(counts[s] >= 0 && (counts[s] == 0 || ptrs[s]), "kth_magnitude: bad segment %d", s);
        if (counts[s] == 0) continue;
        t.p[ns] = ptrs[s];
        t.count[ns] = counts[s];
        t.chunk0[ns] = (int)chunks;
        chunks += (counts[s] + SEL_CHUNK - 1) / SEL_CHUNK;
        total += counts[s];
        ++ns;
    }
    MCAMD_REQUIRE(chunks < (1ll << 31), "kth_magnitude: too many elements");
    t.chunk0[ns] = (int)chunks;
    t.nseg = ns;
    MCAMD_REQUIRE(k >= 0 && k < total, "kth_magnitude: k=%lld out of range for %lld elements", (long long)k, total);
    hipStream_t st = (hipStream_t)stream;
    unsigned* hist = (unsigned*)workspace;
    SelectState* state = (SelectState*)(hist + 2 * 2048);
    const long long k1 = k + 1 > total - 1 ? total - 1 : k + 1;
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, st, state, (unsigned long long)k, (unsigned long long)k1, hist);
    long long g = chunks < 2048 ? chunks : 2048;              // 8 resident blocks per CU
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(select_hist_kernel, dim3((int)g), dim3(256), 0, st, t, (const SelectState*)state, pass, hist);
        hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(256), 0, st, hist, state, pass, out2);
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

// KK > 1: partial[o][t] = sequential sum over cin of w[o][c][t]^2.  One thread per (o, t) -- the fallback for
// tiny or oddly shaped tensors (its loads are 36 bytes apart per lane: ~0.1 TB/s).
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

// The same sums for 3x3 layers at the HBM rate.  numpy's order pins each (filter, tap) sum to ONE sequential chain
// over the input channels, so the parallelism is across the O * 9 chains, not inside them.  One wave per filter:
// its 9 * I weights are one contiguous run, streamed in chunks of CC channels (CC * 36 bytes) by all 64 lanes with
// 16-byte loads into LDS, the NEXT chunk's loads in flight (registers) while lanes 0-8 walk the current chunk, each
// along its own tap (LDS stride 9 words: conflict-free).  Squares are rounded before they are added (this file is
// compiled with -ffp-contract=off), exactly like the one-thread-per-chain form.
template <int CC>
__global__ __launch_bounds__(64) void filter_partial_lds_kernel(const float* w, int O, int I, float* partial) {
    constexpr int CHUNK = CC * 9;                 // floats per chunk
    constexpr int V4 = CHUNK / 4;                 // float4 pieces per chunk
    constexpr int PER = (V4 + 63) / 64;           // pieces per lane
    __shared__ __attribute__((aligned(16))) float buf[CHUNK];
    const int o = blockIdx.x, lane = threadIdx.x;
    const float* row = w + (long long)o * I * 9;  // 16-byte aligned: I % 4 == 0 (checked on the host)
    const int nchunks = I / CC;                   // I % CC == 0 (host)
    f32x4_t pre[PER];
    auto fetch = [&](int q) {
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int i = r * 64 + lane;
            if (i < V4) pre[r] = *(const f32x4_t*)(row + (long long)q * CHUNK + i * 4);
        }
    };
    fetch(0);
    float acc = 0.f;
    for (int q = 0; q < nchunks; ++q) {
        __syncthreads();                          // the chain lanes are done with the previous chunk
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int i = r * 64 + lane;
            if (i < V4) *(f32x4_t*)(buf + i * 4) = pre[r];
        }
        __syncthreads();
        if (q + 1 < nchunks) fetch(q + 1);        // in flight while the chains run
        if (lane < 9) {
            // eight LDS reads and squares at a time, then the eight dependent adds (the chain itself is the
            // only serial part); chunk 0 starts the chain with its first square, as numpy's accumulate does
#pragma unroll 1
            for (int c0 = 0; c0 < CC; c0 += 8) {
                float sq[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = buf[(c0 + j) * 9 + lane];
                    sq[j] = v * v;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = (q == 0 && c0 == 0 && j == 0) ? sq[0] : acc + sq[j];
            }
        }
    }
    if (lane < 9) partial[o * 9 + lane] = acc;
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
    if (KK == 9 && cin % 64 == 0 && (((size_t)w) & 15) == 0) {
        hipLaunchKernelGGL(filter_partial_lds_kernel<64>, dim3(cout), dim3(64), 0, st, w, cout, cin, partial);
    } else if (KK == 9 && cin % 32 == 0 && (((size_t)w) & 15) == 0) {
        hipLaunchKernelGGL(filter_partial_lds_kernel<32>, dim3(cout), dim3(64), 0, st, w, cout, cin, partial);
    } else if (KK > 1) {
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
