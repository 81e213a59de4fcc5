// Skipping the INPUT channels that filter pruning killed (north_star: "a sparse-masked ... path that skips zeroed
// filters"; the reference computes them all, layers.py:59-64).
//
// A filter whose mask row is zero produces y = 0 everywhere, so after BatchNorm + LeakyReLU its channel is the constant
// v_c = leaky(beta_c) at every interior pixel (0 in the zero-padding halo; a 2x2 max pool of a constant is the same
// constant).  The consumer convolution therefore sees, from ALL dead channels together,
//
//     sum_{c dead} v_c sum_{taps inside the image} W[n][c][tap]  =  conv of ONE channel that is 1 at every interior pixel
//                                                                    with the folded filter W'[n][tap] = sum_c v_c W[n][c][tap]
//
// -- border effects included, because the ones-channel has the same zero halo.  The engine keeps such a channel right
// behind the kept channels of the producer's output (mcamd_bn_coeffs_ex: scale 0 / shift 1 for that physical channel, so
// the BatchNorm pass itself writes the ones) and runs the consumer's forward, dgrad and wgrad on  kept + 1  input
// channels with the AUGMENTED weight tensor built here once per step:
//
// (Waug / dWaug are [n][cin_aug][k][k] with cin_aug >= cin_k + 1: the kernels want a channel count that is a multiple
// of 8; the padding columns hold zero weights and their gradients are ignored.)
//     fold    Waug[n][j]      = W[rows[n]][cols[j]] * mask                       j <  cin_k   (kept inputs, physical order)
//             Waug[n][cin_k]  = sum_{j >= cin_k} v_{cols[j]} W[rows[n]][cols[j]] * mask        (the folded filter)
//     unfold  dW[rows[n]][cols[j]]  = dWaug[n][j] * mask                         j <  cin_k
//             dW[rows[n]][cols[j]]  = v_{cols[j]} dWaug[n][cin_k] * mask         j >= cin_k   (X is the constant v: chain rule)
//             dbeta_prod[c] (+)= leaky'(beta_c) sum_{n, tap} W[rows[n]][c][tap] mask dWaug[n][cin_k][tap]   c dead
//             dgamma_prod[c]   = 0                                               (xhat = 0 for a dead filter)
//
// dWaug[n][cin_k][tap] = sum_p dY[p][n] [p + tap inside the image] is exactly the quantity both gradients need: the
// gradient wrt a dead channel's constant is sum_p G[p][c] = sum_{n, tap} W[n][c][tap] dWaug[n][cin_k][tap].  Everything
// is exact algebra on the reference's semantics (BatchNorm parameters of pruned filters and the consumer's weights on
// dead inputs keep training, as they do in the reference); nothing is approximated.  Deterministic reductions.
#include "common.h"

namespace {

__device__ __forceinline__ float leaky_of(float b, float slope) { return b > 0.f ? b : b * slope; }

__device__ __forceinline__ void fold_one_filter(float* red, int n, const float* w, const float* mask, const int* rows,
                                                const int* cols, const float* beta, float slope, int cin_t, int cin_k, int cin_aug,
                                                int kk, float* waug) {
    const int tid = threadIdx.x;
    const long long rbase = (long long)(rows ? rows[n] : n) * cin_t * kk;
    float* dst = waug + (long long)n * cin_aug * kk;
    for (int idx = (cin_k + 1) * kk + tid; idx < cin_aug * kk; idx += 256) dst[idx] = 0.f;   // alignment padding of the channel count
    for (int idx = tid; idx < cin_k * kk; idx += 256) {
        const int j = idx / kk, t = idx - j * kk;
        const long long src = rbase + (long long)(cols ? cols[j] : j) * kk + t;
        dst[idx] = mask ? w[src] * mask[src] : w[src];
    }
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int j = cin_k + tid; j < cin_t; j += 256) {
        const int c = cols ? cols[j] : j;
        const float v = leaky_of(beta[c], slope);
        const long long src = rbase + (long long)c * kk;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (t < kk) acc[t] += v * (mask ? w[src + t] * mask[src + t] : w[src + t]);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) red[t * 256 + tid] = acc[t];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int t = 0; t < 9; ++t) red[t * 256 + tid] += red[t * 256 + tid + o];
        }
        __syncthreads();
    }
    if (tid < kk) dst[cin_k * kk + tid] = red[tid * 256];
}

__global__ __launch_bounds__(256) void fold_weights_kernel(const float* w, const float* mask, const int* rows, const int* cols,
                                                           const float* beta, float slope, int cin_t, int cin_k, int cin_aug,
                                                           int kk, float* waug) {
    __shared__ float red[256 * 9];
    fold_one_filter(red, blockIdx.x, w, mask, rows, cols, beta, slope, cin_t, cin_k, cin_aug, kk, waug);
}

// every folding layer of a network in one launch: block -> (job, filter) through the jobs' first_block prefix sums
__global__ __launch_bounds__(256) void fold_weights_many_kernel(const mcamd_fold_job* jobs, int njobs) {
    __shared__ float red[256 * 9];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (long long)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const mcamd_fold_job j = jobs[lo];
    fold_one_filter(red, (int)(blockIdx.x - j.first_block), j.d.w, j.d.mask, (const int*)j.d.rows, (const int*)j.d.cols, j.d.beta,
                    j.d.slope, j.d.cin_t, j.d.cin_k, j.d.cin_aug, j.d.ksize * j.d.ksize, j.waug);
}

__global__ __launch_bounds__(256) void unfold_wgrad_kernel(const float* dwaug, const float* w, const float* mask, const int* rows,
                                                           const int* cols, const float* beta, float slope, int cin_t, int cin_k,
                                                           int cin_aug, int kk, int N, float* dw, float* prod_dbeta,
                                                           float* prod_dgamma, int accumulate) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < N) {        // one physical filter: scatter its row back to OIHW, expand the folded column
        const int n = blockIdx.x;
        const long long rbase = (long long)(rows ? rows[n] : n) * cin_t * kk;
        const float* src = dwaug + (long long)n * cin_aug * kk;
        for (int idx = tid; idx < cin_t * kk; idx += 256) {
            const int j = idx / kk, t = idx - j * kk;
            const int c = cols ? cols[j] : j;
            const long long d = rbase + (long long)c * kk + t;
            float v = j < cin_k ? src[idx] : leaky_of(beta[c], slope) * src[cin_k * kk + t];
            if (mask) v *= mask[d];
            dw[d] = v;
        }
        return;
    }
    // one dead input channel: gradient wrt its constant, summed over the consumer's filters and taps
    const int j = cin_k + ((int)blockIdx.x - N);
    const int c = cols ? cols[j] : j;
    float acc = 0.f;
    for (int idx = tid; idx < N * kk; idx += 256) {
        const int n = idx / kk, t = idx - n * kk;
        const long long s = ((long long)(rows ? rows[n] : n) * cin_t + c) * kk + t;
        const float wv = mask ? w[s] * mask[s] : w[s];
        acc += wv * dwaug[((long long)n * cin_aug + cin_k) * kk + t];
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        const float g = (beta[c] > 0.f ? 1.f : slope) * red[0];
        if (accumulate) prod_dbeta[c] += g;
        else {
            prod_dbeta[c] = g;
            if (prod_dgamma) prod_dgamma[c] = 0.f;
        }
    }
}

int check(const mcamd_fold_desc* d, const char* what) {
    MCAMD_REQUIRE(d && d->w && d->beta, "%s: null argument", what);
    MCAMD_REQUIRE(d->n > 0 && d->cin_t > 0 && d->cin_k >= 0 && d->cin_k < d->cin_t, "%s: need 0 <= cin_k < cin_t and n > 0 (got %d, %d, %d)",
                  what, d->cin_k, d->cin_t, d->n);
    MCAMD_REQUIRE(d->ksize == 1 || d->ksize == 3, "%s: ksize %d unsupported (1 or 3)", what, d->ksize);
    MCAMD_REQUIRE(d->cin_aug > d->cin_k, "%s: cin_aug %d must exceed cin_k %d", what, d->cin_aug, d->cin_k);
    return MCAMD_OK;
}

}  // namespace

extern "C" int mcamd_fold_weights(const mcamd_fold_desc* d, float* waug, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "fold_weights: null descriptor");
        const mcamd_fold_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_fold_weights(&d_, waug, s); });
    }
    if (check(d, "fold_weights")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(waug, "fold_weights: null output");
    hipLaunchKernelGGL(fold_weights_kernel, dim3(d->n), dim3(256), 0, (hipStream_t)stream, d->w, d->mask, (const int*)d->rows,
                       (const int*)d->cols, d->beta, d->slope, d->cin_t, d->cin_k, d->cin_aug, d->ksize * d->ksize, waug);
    MCAMD_LAUNCH_CHECK("fold_weights");
    return MCAMD_OK;
}

extern "C" int mcamd_fold_weights_many(const mcamd_fold_job* jobs_dev, int32_t njobs, int64_t total_blocks, void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_fold_weights_many(jobs_dev, njobs, total_blocks, s); });
    MCAMD_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "fold_weights_many: empty job table");
    hipLaunchKernelGGL(fold_weights_many_kernel, dim3((int)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    MCAMD_LAUNCH_CHECK("fold_weights_many");
    return MCAMD_OK;
}

extern "C" int mcamd_unfold_wgrad(const mcamd_fold_desc* d, const float* dwaug, float* dw_oihw, float* prod_dbeta,
                                  float* prod_dgamma, int32_t accumulate, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(d, "unfold_wgrad: null descriptor");
        const mcamd_fold_desc d_ = *d;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_unfold_wgrad(&d_, dwaug, dw_oihw, prod_dbeta, prod_dgamma, accumulate, s); });
    }
    if (check(d, "unfold_wgrad")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(dwaug && dw_oihw && prod_dbeta, "unfold_wgrad: null argument");
    const int dead = d->cin_t - d->cin_k;
    hipLaunchKernelGGL(unfold_wgrad_kernel, dim3(d->n + dead), dim3(256), 0, (hipStream_t)stream, dwaug, d->w, d->mask,
                       (const int*)d->rows, (const int*)d->cols, d->beta, d->slope, d->cin_t, d->cin_k, d->cin_aug, d->ksize * d->ksize, d->n,
                       dw_oihw, prod_dbeta, prod_dgamma, accumulate);
    MCAMD_LAUNCH_CHECK("unfold_wgrad");
    return MCAMD_OK;
}
