// C-ABI entry points of libmcamd.so (see include/mcamd.h): argument validation, geometry ->
// launch descriptors, weight packing.  No allocation, no synchronisation.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.h"

static thread_local char g_err[512] = "";

void mcamd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int g_mcamd_env_generation = 0;
int McamdEnvSlot::get(const char* name, int dflt) {
    const int g = __atomic_load_n(&g_mcamd_env_generation, __ATOMIC_ACQUIRE);
    if (gen != g) {       // (a race between host threads re-reads the same variable twice: harmless)
        const char* s = getenv(name);
        has = s && *s ? 1 : 0;
        val = has ? atoi(s) : 0;
        __atomic_store_n(&gen, g, __ATOMIC_RELEASE);
    }
    return has ? val : dflt;
}
extern "C" void mcamd_reload_config(void) { __atomic_add_fetch(&g_mcamd_env_generation, 1, __ATOMIC_ACQ_REL); }

extern "C" int mcamd_version(void) { return 101; }
extern "C" const char* mcamd_arch(void) { return "gfx950"; }
extern "C" const char* mcamd_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------
// geometry helpers
// ---------------------------------------------------------------------------------------
static int cin_tap_of(const mcamd_conv_geom* g) { return g->stem ? 32 : round_up_int(g->cin, 32); }
static int ntaps_of(const mcamd_conv_geom* g) { return g->stem ? 3 : g->ksize * g->ksize; }
static int cout_p_of(const mcamd_conv_geom* g) { return round_up_int(g->cout, 32); }
// padded pixels per row / rows per image of the operands (2: zero halo on every side; 1: shared-halo form, mcamd.h), and
// the number of padded pixels the 9-tap weight gradient enumerates
static int pw_of(const mcamd_conv_geom* g) { return g->pad ? 1 : 2; }
static long long padded_pixels(const mcamd_conv_geom* g) {
    const int pw = pw_of(g);
    return (long long)g->B * (g->H + pw) * (g->W + pw) + (pw == 1 ? g->W + 2 : 0);
}

static int check_geom(const mcamd_conv_geom* g, const char* what) {
    MCAMD_REQUIRE(g, "%s: null geometry", what);
    MCAMD_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->cin > 0 && g->cout > 0, "%s: non-positive dimension", what);
    MCAMD_REQUIRE(g->ksize == 1 || g->ksize == 3, "%s: ksize %d unsupported (1 or 3)", what, g->ksize);
    MCAMD_REQUIRE((long long)g->B * g->H * g->W < (1ll << 31), "%s: more than 2^31 output pixels", what);
    MCAMD_REQUIRE(g->pad == 0 || (g->pad == 1 && !g->stem), "%s: pad must be 0 or 1 (and 0 for the stem layer)", what);
    if (g->stem) {
        MCAMD_REQUIRE(g->cin == 3 && g->ksize == 3 && g->x_ld == 4 && g->x_choff == 0,
                      "%s: stem form needs cin=3, ksize=3, x_ld=4, x_choff=0", what);
    } else {
        MCAMD_REQUIRE(g->x_ld % 8 == 0 && g->x_choff % 8 == 0, "%s: x_ld / x_choff must be multiples of 8", what);
        const int span = g->x_wrap > 0 ? g->x_wrap : cin_tap_of(g);
        MCAMD_REQUIRE(g->x_choff + span <= g->x_ld, "%s: x channel slice [%d, %d) exceeds x_ld %d", what,
                      g->x_choff, g->x_choff + span, g->x_ld);
    }
    if (g->x_f8 != 0) {
        MCAMD_REQUIRE(!g->stem && g->x_wrap == 0 && g->x_f8 > 0 && g->x_f8 % 64 == 0 && g->cin == 2 * g->x_f8 && g->x_choff == 0,
                      "%s: x_f8 %d needs cin = 2 P with P %% 64 == 0, no x_wrap, x_choff 0 (cin %d, x_choff %d)", what, g->x_f8, g->cin,
                      g->x_choff);
        MCAMD_REQUIRE(g->x_f8_wexp >= -24 && g->x_f8_wexp <= 40, "%s: x_f8_wexp %d outside [-24, 40]", what, g->x_f8_wexp);
    }
    if (g->x_wrap != 0) {
        const int ct = cin_tap_of(g), kb = ct % 64 == 0 ? 64 : 32;   // kblock_of(ct)
        MCAMD_REQUIRE(!g->stem && g->x_wrap > 0 && g->x_wrap % 64 == 0 && g->x_wrap % kb == 0 && g->cin == g->x_wrap / 2 * 3 && ct == g->cin,
                      "%s: x_wrap %d needs cin = 3 P, x_wrap = 2 P, P %% 32 == 0 (cin %d)", what, g->x_wrap, g->cin);
    }
    return MCAMD_OK;
}

// tap table relative to the row base = padded pixel (h, w) of output pixel (h, w), i.e. the
// top-left corner of its 3x3 window.
static void fill_taps(int ksize, int stem, int row_stride, int ld, int* taps) {
    if (stem) {
        for (int ty = 0; ty < 3; ++ty) taps[ty] = ty * row_stride;
    } else if (ksize == 3) {
        for (int ty = 0; ty < 3; ++ty)
            for (int tx = 0; tx < 3; ++tx) taps[ty * 3 + tx] = ty * row_stride + tx * ld;
    } else {
        taps[0] = row_stride + ld;  // centre pixel
    }
}

extern "C" int64_t mcamd_packed_elems_fwd(const mcamd_conv_geom* g) {
    if (!g) return 0;
    return (int64_t)round_up_int(g->cout, 256) * ntaps_of(g) * cin_tap_of(g);
}
extern "C" int64_t mcamd_packed_elems_dgrad(const mcamd_conv_geom* g) {
    if (!g || g->stem) return 0;
    return (int64_t)round_up_int(g->cin, 256) * g->ksize * g->ksize * cout_p_of(g);
}

// ---------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int round_up_dev(int v, int m) { return (v + m - 1) / m * m; }
// K order of the packed matrices: [channel block][tap][channel inside the block].  All taps of a 64-channel
// block are consecutive K chunks, so the nine shifted reads of the same activation lines follow each other
// closely and hit L2 instead of streaming the whole activation tile nine times from the Infinity Cache.
__host__ __device__ __forceinline__ int kblock_of(int ch_padded) { return ch_padded % 64 == 0 ? 64 : 32; }
// position of (tap t, channel c) inside one packed row of `taps` taps x `chp` padded channels
__host__ __device__ __forceinline__ int kpos(int t, int c, int taps, int chp) {
    const int kb = kblock_of(chp);
    return (c / kb) * taps * kb + t * kb + c % kb;
}

// `wlo` (may be NULL): the residual fp16(v - fp16(v)) of every entry in the same layout -- the lo half of split operands
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* w, const float* mask, half_t* wp, int Cout, int Cin,
                                                       int ks, int cin_tap, int stem, long long total, int ktot,
                                                       const int* rmap, const int* cmap, half_t* wlo = nullptr) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        int n = (int)(idx / ktot);
        int k = (int)(idx - (long long)n * ktot);
        float v = 0.f;
        if (n < Cout) {
            int c, ty, tx;
            bool ok;
            if (stem) {
                ty = k >> 5;
                int r = k & 31;
                tx = r >> 2;
                c = r & 3;
                ok = tx < 3 && c < 3;
            } else {
                const int kb = kblock_of(cin_tap), blk = ks * ks * kb;
                const int cb = k / blk, r = k - cb * blk;
                const int t = r / kb;
                c = cb * kb + (r - t * kb);
                ty = t / ks;
                tx = t - ty * ks;
                ok = c < Cin;
            }
            if (ok) {
                const int ns = rmap ? rmap[n] : n, cs = cmap ? cmap[c] : c;
                long long src = (((long long)ns * Cin + cs) * ks + ty) * ks + tx;
                v = w[src];
                if (mask) v *= mask[src];
            }
        }
        const half_t hv = (half_t)v;
        wp[idx] = hv;
        if (wlo) wlo[idx] = (half_t)(v - (float)hv);
    }
}

__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* w, const float* mask, half_t* wp, int Cout, int Cin,
                                                         int ks, int cout_p, long long total, int ktot,
                                                         const int* rmap, const int* cmap) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        int c = (int)(idx / ktot);
        int k = (int)(idx - (long long)c * ktot);
        const int kb = kblock_of(cout_p), blk = ks * ks * kb;
        const int nb = k / blk, r = k - nb * blk;
        int t = r / kb;
        int n = nb * kb + (r - t * kb);
        float v = 0.f;
        if (c < Cin && n < Cout) {
            int ty = t / ks, tx = t - ty * ks;
            const int ns = rmap ? rmap[n] : n, cs = cmap ? cmap[c] : c;
            long long src = (((long long)ns * Cin + cs) * ks + (ks - 1 - ty)) * ks + (ks - 1 - tx);
            v = w[src];
            if (mask) v *= mask[src];
        }
        wp[idx] = (half_t)v;
    }
}

extern "C" int mcamd_pack_weights(const mcamd_conv_geom* g, const float* w_oihw, const float* mask_oihw,
                                  const mcamd_chan_map* map, void* wp_fwd, void* wp_dgrad, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(g, "pack_weights: null geometry");
        const mcamd_conv_geom g_ = *g;
        const bool has_map = map != nullptr;
        const mcamd_chan_map m_ = has_map ? *map : mcamd_chan_map{nullptr, nullptr};
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_pack_weights(&g_, w_oihw, mask_oihw, has_map ? &m_ : nullptr, wp_fwd, wp_dgrad, s); });
    }
    if (check_geom(g, "pack_weights")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(w_oihw, "pack_weights: null weights");
    const int* rmap = map ? (const int*)map->rows : nullptr;
    const int* cmap = map ? (const int*)map->cols : nullptr;
    MCAMD_REQUIRE(!(g->stem && cmap), "pack_weights: the stem layer takes no input-channel map");
    hipStream_t st = (hipStream_t)stream;
    if (wp_fwd) {
        int ktot = ntaps_of(g) * cin_tap_of(g);
        long long total = mcamd_packed_elems_fwd(g);
        long long grid = (total + 256 * 4 - 1) / (256 * 4);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(pack_fwd_kernel, dim3((int)grid), dim3(256), 0, st, w_oihw, mask_oihw, (half_t*)wp_fwd, g->cout,
                           g->cin, g->ksize, cin_tap_of(g), g->stem, total, ktot, rmap, cmap);
    }
    if (wp_dgrad) {
        MCAMD_REQUIRE(!g->stem, "pack_weights: the stem layer has no dgrad packing");
        int ktot = g->ksize * g->ksize * cout_p_of(g);
        long long total = mcamd_packed_elems_dgrad(g);
        long long grid = (total + 256 * 4 - 1) / (256 * 4);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(pack_dgrad_kernel, dim3((int)grid), dim3(256), 0, st, w_oihw, mask_oihw, (half_t*)wp_dgrad,
                           g->cout, g->cin, g->ksize, cout_p_of(g), total, ktot, rmap, cmap);
    }
    MCAMD_LAUNCH_CHECK("pack_weights");
    return MCAMD_OK;
}

extern "C" int mcamd_pack_stem_split(const float* w_oihw, const float* mask_oihw, int32_t cout, void* wp_hi, void* wp_lo,
                                     void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_pack_stem_split(w_oihw, mask_oihw, cout, wp_hi, wp_lo, s); });
    MCAMD_REQUIRE(w_oihw && wp_hi && wp_lo && cout > 0, "pack_stem_split: bad argument");
    mcamd_conv_geom g = {};
    g.B = 1, g.H = 2, g.W = 32, g.ksize = 3, g.cin = 3, g.cout = cout, g.x_ld = 4, g.stem = 1;
    const int ktot = ntaps_of(&g) * cin_tap_of(&g);
    const long long total = mcamd_packed_elems_fwd(&g);
    long long grid = (total + 256 * 4 - 1) / (256 * 4);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_fwd_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, w_oihw, mask_oihw, (half_t*)wp_hi,
                       cout, 3, 3, cin_tap_of(&g), 1, total, ktot, (const int*)nullptr, (const int*)nullptr, (half_t*)wp_lo);
    MCAMD_LAUNCH_CHECK("pack_stem_split");
    return MCAMD_OK;
}

// All layers in one launch: a workgroup packs a 32-filter x 32-channel tile of one layer into both layouts.
// Wide accesses on both sides: a tile row (32 channels x k*k taps of one filter) is one contiguous run of the OIHW master
// and is read as float4s; both fp16 layouts keep 8 consecutive channels (forward) / 8 consecutive filters (dgrad) of one
// tap adjacent, so a lane writes 16 bytes.  (2-byte stores and nine 4-byte loads per lane: 164 us for the 405 MB of a
// YOLOv2 re-pack = 2.5 TB/s, the instruction count being the bound.)
__global__ __launch_bounds__(256) void pack_tiles_kernel(const mcamd_pack_job* jobs, int njobs, long long total) {
    constexpr int TS = 32, LDW = TS * 9 + 1;           // row stride of the LDS tile (odd: column reads spread over banks)
    __shared__ float tile[TS * LDW];                    // [n][c * kk + t], 37 KB
    for (long long item = blockIdx.x; item < total; item += gridDim.x) {
        int lo = 0, hi = njobs - 1;                     // last job whose first_tile <= item
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_tile <= item) lo = mid;
            else hi = mid - 1;
        }
        const mcamd_pack_job j = jobs[lo];
        const int ks = j.ksize, kk = ks * ks;
        const int ctiles = (j.cin + TS - 1) / TS;
        const int r = (int)(item - j.first_tile);
        const int n0 = (r / ctiles) * TS, c0 = (r - (r / ctiles) * ctiles) * TS;
        __syncthreads();                                // the previous item's readers are done with the tile
        // 1. gather.  Without an input-channel map a full tile row is TS * kk contiguous floats of the master
        const bool rows_contig = !j.cols && c0 + TS <= j.cin && ((long long)j.cin * kk) % 4 == 0 && (c0 * kk) % 4 == 0;
        if (rows_contig) {
            const int q4 = TS * kk / 4;                 // float4s per row (72 or 8)
            for (int e = threadIdx.x; e < TS * q4; e += 256) {
                const int nl = e / q4, q = (e - nl * q4) * 4;
                const int n = n0 + nl;
                f32x4_t v = {0.f, 0.f, 0.f, 0.f};
                if (n < j.cout) {
                    const int ns = j.rows ? j.rows[n] : n;
                    const long long src = ((long long)ns * j.cin + c0) * kk + q;
                    v = *(const f32x4_t*)(j.w + src);
                    if (j.mask) {
                        const f32x4_t m = *(const f32x4_t*)(j.mask + src);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = v[i] * m[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) tile[nl * LDW + q + i] = v[i];
            }
        } else {
            // thread -> (n, c) pairs, c fastest: the k*k taps of a pair are one contiguous run
            for (int pair = threadIdx.x; pair < TS * TS; pair += 256) {
                const int nl = pair / TS, cl = pair - nl * TS;
                const int n = n0 + nl, c = c0 + cl;
                float v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = 0.f;
                if (n < j.cout && c < j.cin) {
                    const int ns = j.rows ? j.rows[n] : n, cs = j.cols ? j.cols[c] : c;
                    const long long src = ((long long)ns * j.cin + cs) * kk;
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        if (t < kk) v[t] = j.mask ? j.w[src + t] * j.mask[src + t] : j.w[src + t];
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (t < kk) tile[nl * LDW + cl * kk + t] = v[t];
            }
        }
        __syncthreads();
        // 2. forward layout [n][kpos(t, c)]: 8 consecutive channels of one (filter, tap) per lane = one 16-byte store.
        // j.split: the split-operand packing [w_hi | w_hi | w_lo] along the input channels (w_hi = fp16(w * mask),
        // w_lo = fp16(w * mask - w_hi)), i.e. channel c is written at c, cin + c and 2 cin + c of a 3 cin wide row.
        if (j.dst_fwd && j.split == 2) {
            // fp8 correction packing (mcamd_conv_geom.x_f8): a row of 2 cin fp16 units per tap = [w_hi: cin fp16 | w8: cin
            // e4m3 bytes | wlo8: cin bytes], the byte string cut into the same [channel block][tap][64] K order as the
            // activations' [x_hi | lo8 | x8].  cin % 64 == 0 (host-checked).
            half_t* dst = (half_t*)j.dst_fwd;
            const int cin_tap = 2 * j.cin;
            for (int e = threadIdx.x; e < TS * kk * (TS / 8); e += 256) {
                const int g8 = e % (TS / 8), t = (e / (TS / 8)) % kk, nl = e / ((TS / 8) * kk);
                const int n = n0 + nl, c = c0 + g8 * 8;
                if (n >= j.cout || c >= j.cin) continue;
                const float* src = tile + nl * LDW + (g8 * 8) * kk + t;
                half_t* row = dst + (long long)n * kk * cin_tap;
                h8_t h;
                float q8[8], ql[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = src[i * kk];
                    h[i] = (half_t)v;
                    q8[i] = fminf(fmaxf(ldexpf((float)h[i], j.f8_wexp), -448.f), 448.f);
                    ql[i] = fminf(fmaxf(ldexpf(v - (float)h[i], j.f8_wexp + 11), -448.f), 448.f);
                }
                *(h8_t*)(row + kpos(t, c, kk, cin_tap)) = h;
                int w8[2], wl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    w8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(q8[4 * i], q8[4 * i + 1], 0, false);
                    w8[i] = __builtin_amdgcn_cvt_pk_fp8_f32(q8[4 * i + 2], q8[4 * i + 3], w8[i], true);
                    wl[i] = __builtin_amdgcn_cvt_pk_fp8_f32(ql[4 * i], ql[4 * i + 1], 0, false);
                    wl[i] = __builtin_amdgcn_cvt_pk_fp8_f32(ql[4 * i + 2], ql[4 * i + 3], wl[i], true);
                }
                // e4m3 element e of the tap's [w8 | wlo8] string lives in byte (e & 1) of fp16 unit cin + e / 2
                int* d8 = (int*)(row + kpos(t, j.cin + c / 2, kk, cin_tap));
                d8[0] = w8[0], d8[1] = w8[1];
                int* dl = (int*)(row + kpos(t, j.cin + (j.cin + c) / 2, kk, cin_tap));
                dl[0] = wl[0], dl[1] = wl[1];
            }
        } else if (j.dst_fwd) {
            half_t* dst = (half_t*)j.dst_fwd;
            const int parts = j.split ? 3 : 1;
            const int cin_tap = round_up_dev(j.cin * parts, 32);
            const bool vec_ok = parts == 1 || j.cin % 8 == 0;
            for (int e = threadIdx.x; e < TS * kk * (TS / 8); e += 256) {
                const int g8 = e % (TS / 8), t = (e / (TS / 8)) % kk, nl = e / ((TS / 8) * kk);
                const int n = n0 + nl, c = c0 + g8 * 8;
                if (n >= j.cout || c >= j.cin) continue;
                const float* src = tile + nl * LDW + (g8 * 8) * kk + t;
                const int lim = c + 8 <= j.cin ? 8 : j.cin - c;
                for (int part = 0; part < parts; ++part) {
                    half_t* d = dst + (long long)n * kk * cin_tap + kpos(t, c + part * j.cin, kk, cin_tap);
                    h8_t h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = i < lim ? src[i * kk] : 0.f;
                        const half_t hi = (half_t)v;
                        h[i] = part < 2 ? hi : (half_t)(v - (float)hi);
                    }
                    if (lim == 8 && vec_ok) {
                        *(h8_t*)d = h;
                    } else {
                        for (int i = 0; i < lim; ++i) dst[(long long)n * kk * cin_tap + kpos(t, c + i + part * j.cin, kk, cin_tap)] = h[i];
                    }
                }
            }
        }
        // 3. dgrad layout [c][kpos(t', n)], flipped taps: 8 consecutive filters of one (channel, tap) per lane
        if (j.dst_dgrad) {
            half_t* dst = (half_t*)j.dst_dgrad;
            const int cout_p = round_up_dev(j.cout, 32);
            for (int e = threadIdx.x; e < TS * kk * (TS / 8); e += 256) {
                const int g8 = e % (TS / 8), t = (e / (TS / 8)) % kk, cl = e / ((TS / 8) * kk);
                const int n = n0 + g8 * 8, c = c0 + cl;
                if (n >= j.cout || c >= j.cin) continue;
                half_t* d = dst + (long long)c * kk * cout_p + kpos(t, n, kk, cout_p);
                const float* src = tile + (g8 * 8) * LDW + cl * kk + (kk - 1 - t);
                if (n + 8 <= j.cout) {
                    h8_t h;
#pragma unroll
                    for (int i = 0; i < 8; ++i) h[i] = (half_t)src[i * LDW];
                    *(h8_t*)d = h;
                } else {
                    for (int i = 0; n + i < j.cout; ++i) d[i] = (half_t)src[i * LDW];
                }
            }
        }
    }
}

extern "C" int mcamd_pack_weights_many(const mcamd_pack_job* jobs_dev, int32_t njobs, int64_t total_tiles, void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_pack_weights_many(jobs_dev, njobs, total_tiles, s); });
    MCAMD_REQUIRE(jobs_dev && njobs > 0 && total_tiles > 0, "pack_weights_many: empty job table");
    long long grid = total_tiles;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs,
                       (long long)total_tiles);
    MCAMD_LAUNCH_CHECK("pack_weights_many");
    return MCAMD_OK;
}

// ---------------------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------------------
static int fill_epilogue(IgemmArgs& a, const mcamd_conv_epilogue* e, int n_out, long long M, int cin_tap, int ktot,
                         const char* what, int expected_rows = -1) {
    MCAMD_REQUIRE(e && e->y, "%s: null output", what);
    a.y = e->y;
    a.mode = e->mode;
    a.bias = nullptr;
    a.stats = nullptr;
    a.scale = nullptr;
    a.shift = nullptr;
    a.slope = 1.f;
    a.y_ld = e->y_ld;
    a.y_choff = e->y_choff;
    a.stats_ld = 0;
    a.overflow = (int*)e->overflow;
    if (e->mode == MCAMD_EPI_NCHW_F32) {
        a.bias = e->bias;
    } else if (e->mode == MCAMD_EPI_RAW_F16 || e->mode == MCAMD_EPI_PAD_F16) {
        MCAMD_REQUIRE(n_out % 8 == 0, "%s: fp16 output needs a channel count that is a multiple of 8 (got %d)", what, n_out);
        MCAMD_REQUIRE(e->y_ld % 8 == 0 && e->y_choff % 8 == 0 && e->y_choff + n_out <= e->y_ld,
                      "%s: output slice [%d, %d) does not fit y_ld %d", what, e->y_choff, e->y_choff + n_out, e->y_ld);
        if (e->mode == MCAMD_EPI_RAW_F16 && e->stats) {
            int rows = expected_rows >= 0 ? expected_rows : mcamd_igemm_rows(M, n_out, cin_tap, ktot);
            MCAMD_REQUIRE(e->stats_rows == rows, "%s: stats_rows must be mcamd_conv_stats_rows() = %d (got %d)", what, rows,
                          e->stats_rows);
            MCAMD_REQUIRE(e->stats_ld >= round_up_int(n_out, 256), "%s: stats_ld must be >= %d", what,
                          round_up_int(n_out, 256));
            a.stats = e->stats;
            a.stats_ld = e->stats_ld;
        }
        if (e->mode == MCAMD_EPI_PAD_F16) {
            a.scale = e->scale;
            a.shift = e->shift;
            a.slope = e->slope;
            MCAMD_REQUIRE(e->dst_mode == MCAMD_DST_PLAIN || e->dst_mode == MCAMD_DST_POOL || e->dst_mode == MCAMD_DST_REORG,
                          "%s: bad dst_mode %d", what, e->dst_mode);
            if (e->dst_mode != MCAMD_DST_PLAIN) {
                MCAMD_REQUIRE(a.H % 2 == 0 && a.W % 2 == 0 && a.ntaps > 0 && !a.stats, "%s: pooled / reorg epilogue needs even H and W", what);
                const int span = e->dst_mode == MCAMD_DST_REORG ? 4 * n_out : n_out;
                MCAMD_REQUIRE(e->y_choff + span <= e->y_ld, "%s: pooled output slice [%d, %d) does not fit y_ld %d", what, e->y_choff,
                              e->y_choff + span, e->y_ld);
                MCAMD_REQUIRE(!e->y2 || (e->dst_mode == MCAMD_DST_POOL && e->y2_ld % 8 == 0 && e->y2_choff % 8 == 0 &&
                                         e->y2_choff + n_out <= e->y2_ld),
                              "%s: y2 (full-resolution copy) needs dst_mode POOL and a fitting slice", what);
                a.dst_mode = e->dst_mode;
                a.y2 = e->y2;
                a.y2_ld = e->y2_ld, a.y2_choff = e->y2_choff;
            }
        } else {
            MCAMD_REQUIRE(e->dst_mode == 0 && !e->y2, "%s: dst_mode / y2 belong to epilogue mode 2 (MCAMD_EPI_PAD_F16)", what);
        }
    } else if (e->mode == MCAMD_EPI_RAW_F32) {
        MCAMD_REQUIRE(e->y_ld % 4 == 0 && e->y_choff % 4 == 0 && e->y_choff + n_out <= e->y_ld,
                      "%s: fp32 output slice [%d, %d) does not fit y_ld %d", what, e->y_choff, e->y_choff + n_out, e->y_ld);
        if (e->stats) {
            // the fp32 epilogue lives in the LDS-staged implicit-GEMM kernels only (mcamd_conv_stats_rows_mode)
            int rows = expected_rows >= 0 ? expected_rows : mcamd_igemm_rows(M, n_out, cin_tap, ktot, false);
            MCAMD_REQUIRE(e->stats_rows == rows, "%s: stats_rows must be mcamd_conv_stats_rows_mode(g, 3) = %d (got %d)", what,
                          rows, e->stats_rows);
            MCAMD_REQUIRE(e->stats_ld >= round_up_int(n_out, 256), "%s: stats_ld must be >= %d", what,
                          round_up_int(n_out, 256));
            a.stats = e->stats;
            a.stats_ld = e->stats_ld;
        }
    } else {
        MCAMD_REQUIRE(false, "%s: bad epilogue mode %d", what, e->mode);
    }
    return MCAMD_OK;
}

extern "C" int32_t mcamd_conv_stats_rows(const mcamd_conv_geom* g) {
    if (!g) return 0;
    if (mcamd_stem_direct_ok(g->stem, g->cout, MCAMD_EPI_RAW_F16)) return mcamd_stem_rows((long long)g->B * g->H * g->W);
    if (g->pad == 0 && mcamd_wres_ok(g->ksize, g->stem, g->cout, cin_tap_of(g), ntaps_of(g) * cin_tap_of(g), g->B, g->H, g->W, MCAMD_EPI_RAW_F16))
        return mcamd_wres_rows(g->cout, g->B, g->H, g->W);
    return mcamd_igemm_rows((long long)g->B * g->H * g->W, g->cout, cin_tap_of(g), ntaps_of(g) * cin_tap_of(g));
}

extern "C" int32_t mcamd_conv_stats_rows_mode(const mcamd_conv_geom* g, int32_t mode) {
    if (!g) return 0;
    if (mode == MCAMD_EPI_RAW_F32) {
        if (g->pad == 0 && mcamd_small3x3_split_ok((long long)g->B * g->H * g->W, g->cout, cin_tap_of(g), ntaps_of(g) * cin_tap_of(g),
                                                   g->x_wrap, mode))
            return mcamd_small3x3_rows((long long)g->B * g->H * g->W);
        const int ct = g->x_f8 > 0 ? cin_tap_of(g) / 2 * 3 : cin_tap_of(g);      // (x_f8: the tile of the three-product problem)
        return mcamd_igemm_rows((long long)g->B * g->H * g->W, g->cout, ct, ntaps_of(g) * ct, false);
    }
    return mcamd_conv_stats_rows(g);
}

extern "C" int32_t mcamd_conv_fwd_f8_ok(const mcamd_conv_geom* g) {
    if (!g || g->x_f8 <= 0 || g->x_f8 % 64 != 0 || g->cin != 2 * g->x_f8 || g->stem || g->x_wrap != 0 || g->x_choff != 0) return 0;
    return mcamd_igemm_f8_ok((long long)g->B * g->H * g->W, g->cout, cin_tap_of(g), ntaps_of(g) * cin_tap_of(g)) ? 1 : 0;
}

extern "C" int mcamd_conv_tile_info(const mcamd_conv_geom* g, int32_t dgrad, int32_t out[4]) {
    if (check_geom(g, "conv_tile_info")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(out, "conv_tile_info: null output");
    out[3] = 0;
    if (!dgrad && g->x_f8 > 0) {
        const int ct = cin_tap_of(g) / 2 * 3;
        mcamd_igemm_tile((long long)g->B * g->H * g->W, g->cout, ct, ntaps_of(g) * ct, out, false);
        return MCAMD_OK;
    }
    if (!dgrad && mcamd_stem_direct_ok(g->stem, g->cout, MCAMD_EPI_RAW_F16)) {
        out[0] = 32, out[1] = g->cout, out[2] = 48, out[3] = 1;   // stem_fwd_kernel (conv_stem.hip)
        return MCAMD_OK;
    }
    if (!dgrad && g->pad == 0 && mcamd_wres_ok(g->ksize, g->stem, g->cout, cin_tap_of(g), ntaps_of(g) * cin_tap_of(g), g->B, g->H, g->W, MCAMD_EPI_RAW_F16)) {
        out[0] = 128, out[1] = 128, out[2] = 64, out[3] = 6;   // wres_kernel (conv_wres.hip): weights resident in registers
        return MCAMD_OK;
    }
    if (dgrad && mcamd_win3x3_shape((long long)g->B * g->H * g->W, g->cin, cout_p_of(g), g->ksize * g->ksize * cout_p_of(g), g->W)) {
        out[0] = 32, out[1] = round_up_int(g->cin, 16), out[2] = 64, out[3] = 5;   // win3x3_kernel (conv_win.hip)
        return MCAMD_OK;
    }
    mcamd_igemm_tile((long long)g->B * g->H * g->W, dgrad ? g->cin : g->cout, dgrad ? cout_p_of(g) : cin_tap_of(g),
                     dgrad ? g->ksize * g->ksize * cout_p_of(g) : ntaps_of(g) * cin_tap_of(g), out,
                     dgrad == 2);
    return MCAMD_OK;
}

extern "C" int mcamd_conv_fwd(const mcamd_conv_geom* g, const void* x, const void* wp_fwd, const mcamd_conv_epilogue* epi,
                              void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(g && epi, "conv_fwd: null geometry / epilogue");
        const mcamd_conv_geom g_ = *g;
        const mcamd_conv_epilogue e_ = *epi;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_conv_fwd(&g_, x, wp_fwd, &e_, s); });
    }
    if (check_geom(g, "conv_fwd")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(x && wp_fwd, "conv_fwd: null input");
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const half_t*)x;
    a.w = (const half_t*)wp_fwd;
    a.x_ld = g->x_ld;
    a.x_row_stride = (g->W + pw_of(g)) * g->x_ld;
    a.x_img_stride = (long long)(g->H + pw_of(g)) * a.x_row_stride;
    a.x_off = g->x_choff;
    a.H = g->H, a.W = g->W, a.HW = g->H * g->W;
    a.M = g->B * g->H * g->W;
    a.N = g->cout;
    a.cin_tap = cin_tap_of(g);
    a.ntaps = ntaps_of(g);
    a.kb = kblock_of(a.cin_tap);
    a.ktot = a.ntaps * a.cin_tap;
    fill_taps(g->ksize, g->stem, a.x_row_stride, g->x_ld, a.tap_off);
    a.wrap = g->x_wrap > 0 ? g->x_wrap : 0x7fffffff;
    MCAMD_REQUIRE(g->x_wrap == 0 || (epi && (epi->mode == MCAMD_EPI_RAW_F32 || epi->mode == MCAMD_EPI_NCHW_F32)),
                  "conv_fwd: x_wrap goes with the fp32 epilogues (modes 3 and 1)");
    // x_f8: channel blocks [P, 2 P) of the slice are e4m3 bytes; K order [channel block][tap][kb] -> the fp8 chunks are the tail
    a.f8_from = g->x_f8 > 0 ? (g->x_f8 / a.kb) * a.ntaps * (a.kb / 32) : 0x7fffffff;
    a.f8_sb = (127 - (MCAMD_F8_SXL + g->x_f8_wexp)) * 0x01010101;
    if (g->x_f8 > 0) {
        MCAMD_REQUIRE(epi && epi->mode == MCAMD_EPI_RAW_F32, "conv_fwd: x_f8 goes with the fp32 epilogue (mode 3)");
        MCAMD_REQUIRE(mcamd_igemm_f8_ok(a.M, g->cout, a.cin_tap, a.ktot), "conv_fwd: no fp8-correction kernel for this shape (mcamd_conv_fwd_f8_ok)");
    }
    const bool stem_direct = epi && mcamd_stem_direct_ok(g->stem, g->cout, epi->mode);
    const bool wres = epi && epi->dst_mode == MCAMD_DST_PLAIN && g->pad == 0 && g->x_f8 == 0 &&
                      mcamd_wres_ok(g->ksize, g->stem, g->cout, a.cin_tap, a.ktot, g->B, g->H, g->W, epi->mode);
    const bool small_split = epi && g->pad == 0 && !g->stem && g->x_choff == 0 && g->x_f8 == 0 &&
                             mcamd_small3x3_split_ok(a.M, g->cout, a.cin_tap, a.ktot, g->x_wrap, epi->mode);
    if (fill_epilogue(a, epi, g->cout, a.M, a.cin_tap, a.ktot, "conv_fwd",
                      stem_direct ? mcamd_stem_rows(a.M) : (wres ? mcamd_wres_rows(g->cout, g->B, g->H, g->W)
                                                                 : (small_split ? mcamd_small3x3_rows(a.M)
                                                                                : (g->x_f8 > 0 ? mcamd_conv_stats_rows_mode(g, MCAMD_EPI_RAW_F32) : -1)))))
        return MCAMD_EINVAL;
    if (small_split) return mcamd_small3x3_split_launch(a, (hipStream_t)stream);   // conv_small.hip: weights resident, split operands
    if (wres) return mcamd_wres_launch(a, g->B, (hipStream_t)stream);
    if (stem_direct) {       // conv_stem.hip: weights in registers, image fragments straight from global memory
        StemArgs q;
        q.x = a.x, q.w = a.w, q.y = (half_t*)a.y, q.stats = a.stats;
        q.y_ld = a.y_ld, q.y_choff = a.y_choff, q.stats_ld = a.stats_ld;
        q.H = g->H, q.W = g->W, q.HW = a.HW, q.M = a.M;
        return mcamd_stem_launch(q, g->cout, (hipStream_t)stream);
    }
    return mcamd_igemm_launch(a, (hipStream_t)stream);
}

extern "C" int mcamd_conv_dgrad(const mcamd_conv_geom* g, const void* dy, int32_t dy_ld, int32_t dy_choff,
                                const void* wp_dgrad, const mcamd_conv_epilogue* epi, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(g && epi, "conv_dgrad: null geometry / epilogue");
        const mcamd_conv_geom g_ = *g;
        const mcamd_conv_epilogue e_ = *epi;
        return mcamd_rec_push(stream, [=](void* s) { return mcamd_conv_dgrad(&g_, dy, dy_ld, dy_choff, wp_dgrad, &e_, s); });
    }
    if (check_geom(g, "conv_dgrad")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(g->x_wrap == 0 && g->x_f8 == 0, "conv_dgrad: x_wrap / x_f8 are forward-only fields");
    MCAMD_REQUIRE(!g->stem, "conv_dgrad: the stem layer has no input gradient");
    MCAMD_REQUIRE(dy && wp_dgrad, "conv_dgrad: null input");
    int cout_p = cout_p_of(g);
    MCAMD_REQUIRE(dy_ld % 8 == 0 && dy_choff % 8 == 0 && dy_choff + cout_p <= dy_ld,
                  "conv_dgrad: dy slice [%d, %d) does not fit dy_ld %d", dy_choff, dy_choff + cout_p, dy_ld);
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const half_t*)dy;
    a.w = (const half_t*)wp_dgrad;
    a.x_ld = dy_ld;
    a.x_row_stride = (g->W + pw_of(g)) * dy_ld;
    a.x_img_stride = (long long)(g->H + pw_of(g)) * a.x_row_stride;
    a.x_off = dy_choff;
    a.H = g->H, a.W = g->W, a.HW = g->H * g->W;
    a.M = g->B * g->H * g->W;
    a.N = g->cin;
    a.cin_tap = cout_p;
    a.wrap = 0x7fffffff;
    a.f8_from = 0x7fffffff;
    a.ntaps = g->ksize * g->ksize;
    a.kb = kblock_of(a.cin_tap);
    a.ktot = a.ntaps * a.cin_tap;
    fill_taps(g->ksize, 0, a.x_row_stride, dy_ld, a.tap_off);
    MCAMD_REQUIRE(epi && epi->mode != MCAMD_EPI_PAD_F16 && !epi->stats && epi->dst_mode == 0 && !epi->y2,
                  "conv_dgrad: epilogue must be mode 0 (no stats) or 1");
    if (fill_epilogue(a, epi, g->cin, a.M, a.cin_tap, a.ktot, "conv_dgrad")) return MCAMD_EINVAL;
    a.concurrent = epi->concurrent != 0;
    return mcamd_igemm_launch(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// wgrad
// ---------------------------------------------------------------------------------------
static WgradPlan wgrad_plan_for(const mcamd_conv_geom* g) {
    if (mcamd_wgrad_stem_ok(g->stem, g->cout, g->W, (long long)g->B * g->H * g->W))
        return mcamd_wgrad_stem_plan((long long)g->B * g->H * g->W);
    if (mcamd_wgrad_win_ok(g->ksize, g->stem, g->cout, cin_tap_of(g), g->W, (long long)g->B * g->H * g->W))
        return mcamd_wgrad_win_plan((long long)g->B * g->H * g->W, g->cout);
    if (mcamd_wgrad_use9(g->ksize, g->stem, g->cout, cin_tap_of(g), g->W))
        return mcamd_wgrad_plan9(padded_pixels(g), g->cout, cin_tap_of(g), g->W, g->W + pw_of(g), g->B);
    return mcamd_wgrad_plan((long long)g->B * g->H * g->W, g->cout, cin_tap_of(g), ntaps_of(g));
}

extern "C" size_t mcamd_conv_wgrad_workspace_bytes(const mcamd_conv_geom* g) {
    if (!g) return 0;
    return wgrad_plan_for(g).bytes;
}

extern "C" int mcamd_conv_wgrad(const mcamd_conv_geom* g, const void* x, const void* dy, int32_t dy_ld, int32_t dy_choff,
                                const float* mask_oihw, const mcamd_chan_map* map, float grad_scale, float* dw_oihw,
                                float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    if (mcamd_recording()) {
        MCAMD_REQUIRE(g, "conv_wgrad: null geometry");
        const mcamd_conv_geom g_ = *g;
        const bool has_map = map != nullptr;
        const mcamd_chan_map m_ = has_map ? *map : mcamd_chan_map{nullptr, nullptr};
        return mcamd_rec_push(stream, [=](void* s) {
            return mcamd_conv_wgrad(&g_, x, dy, dy_ld, dy_choff, mask_oihw, has_map ? &m_ : nullptr, grad_scale, dw_oihw, dbias,
                                    workspace, workspace_bytes, s);
        });
    }
    if (check_geom(g, "conv_wgrad")) return MCAMD_EINVAL;
    MCAMD_REQUIRE(g->x_wrap == 0 && g->x_f8 == 0, "conv_wgrad: x_wrap / x_f8 are forward-only fields");
    MCAMD_REQUIRE(x && dy && dw_oihw && workspace, "conv_wgrad: null argument");
    MCAMD_REQUIRE(grad_scale > 0.f, "conv_wgrad: grad_scale must be positive");
    const int* rmap = map ? (const int*)map->rows : nullptr;
    const int* cmap = map ? (const int*)map->cols : nullptr;
    MCAMD_REQUIRE(!(g->stem && cmap), "conv_wgrad: the stem layer takes no input-channel map");
    const int cin_tap = cin_tap_of(g), ntaps = ntaps_of(g);
    const long long M = (long long)g->B * g->H * g->W;
    WgradPlan p = wgrad_plan_for(g);
    MCAMD_REQUIRE(dy_ld % 8 == 0 && dy_choff % 8 == 0 && dy_choff + p.rows_pad <= dy_ld,
                  "conv_wgrad: dy slice [%d, %d) does not fit dy_ld %d", dy_choff, dy_choff + p.rows_pad, dy_ld);
    if (workspace_bytes < p.bytes) {
        mcamd_set_error("conv_wgrad: workspace %zu < %zu bytes", workspace_bytes, p.bytes);
        return MCAMD_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const half_t*)x;
    a.dy = (const half_t*)dy;
    a.slab = (float*)workspace;
    a.x_ld = g->x_ld;
    a.x_row_stride = (g->W + pw_of(g)) * g->x_ld;
    a.x_img_stride = (long long)(g->H + pw_of(g)) * a.x_row_stride;
    a.x_off = g->x_choff;
    a.dy_ld = dy_ld;
    a.dy_row_stride = (g->W + pw_of(g)) * dy_ld;
    a.dy_img_stride = (long long)(g->H + pw_of(g)) * a.dy_row_stride;
    a.dy_off = dy_choff + a.dy_row_stride + dy_ld;
    a.dy_zero_off = dy_choff;
    a.H = g->H, a.W = g->W, a.HW = g->H * g->W;
    a.M = (int)M;
    a.cin_tap = cin_tap;
    a.ntaps = ntaps;
    a.ktot = ntaps * cin_tap;
    fill_taps(g->ksize, g->stem, a.x_row_stride, g->x_ld, a.tap_off);
    int rc = p.nine    ? mcamd_wgrad9_launch(a, p, g->W + pw_of(g), padded_pixels(g), st)
             : p.stemw == 1 ? mcamd_wgrad_stem_launch(a, p, st)
             : p.stemw == 2 ? mcamd_wgrad_win_launch(a, p, st)
                       : mcamd_wgrad_launch(a, p, st);
    if (rc) return rc;
    rc = mcamd_wgrad_finish_launch((const float*)workspace, p, a.ktot, cin_tap, g->stem, g->cout, g->cin, g->ksize,
                                   mask_oihw, 1.0f / grad_scale, dw_oihw, rmap, cmap, st);
    if (rc) return rc;
    if (dbias) {
        long long rows = padded_pixels(g);
        // (the split-K slabs are consumed by the finish pass enqueued above: the workspace is free again, in stream order)
        rc = mcamd_colsum_launch((const half_t*)dy, rows, dy_ld, dy_choff, g->cout, 1.0f / grad_scale, dbias, st, workspace,
                                 workspace_bytes);
    }
    return rc;
}
