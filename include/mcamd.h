/* mcamd.h -- C ABI of libmcamd.so, the MI355X (gfx950) hot path of
 * modelcompression_amd: YOLOv2/Darknet-19 convolution forward/backward and the
 * src/pruning mask kernels of AnishDelft/ModelCompression.
 *
 * The reference has no FFI layer: its operator API is
 * Python calling torch.  Each entry point below names the reference call site it
 * replaces.  Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no C++/torch types;
 *   - every pointer is a DEVICE pointer into memory owned by the caller
 *     (PyTorch-ROCm's allocator in practice); the library never allocates,
 *     frees or retains device memory -- workspaces are caller-provided and sized
 *     by the *_workspace_bytes queries;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     calls only enqueue work on it: no host synchronisation, graph-capturable;
 *   - returns 0 on success, a negative MCAMD_E* code otherwise; the message is
 *     available per host thread from mcamd_last_error(); nothing throws.
 *
 * Device data layouts
 *   activation ("padded NHWC"): fp16 [B][H+2][W+2][ld], a one-pixel zero halo on
 *     every side, `ld` channels per pixel (ld >= channels used, multiple of 8;
 *     a tensor may be a channel slice [choff, choff+C) of a wider buffer -- that
 *     is how route/concat is expressed).  The pointer passed is the address of
 *     padded pixel (b=0, hp=0, wp=0), channel 0.  The halo must be zero and is
 *     never written by the library.  Buffers must be preceded AND followed by a
 *     zeroed guard band of (round_up(W+3, 4) + 128) pixels (+ 64 elements): the 9-tap
 *     kernels read whole row windows around their pixel tiles, and the stem
 *     layer reads 32 contiguous halfs per pixel.
 *   raw conv output / gradient wrt a block output: fp16 [B*H*W][ld] (no halo).
 *   stem input (first layer, Cin = 3): padded NHWC with ld = 4 (channel 3 zero).
 *   activation, SHARED-HALO form (`pad` = 1 in the descriptors below; the engine uses it for its small images, W <= 26):
 *     the same tensor with ONE zero pixel between consecutive rows and ONE zero row between consecutive images --
 *     pixel (b, h, w) at ((b (H + 1) + h + 1) (W + 1) + w + 1) ld from a pointer that is the address of pixel
 *     (0, -1, -1); B (H + 1) (W + 1) + W + 2 pixels in all (the last image's bottom halo row and corner), guard bands as
 *     above.  The right halo of a row IS the left halo of the next one, the bottom halo row of an image the top one of the
 *     next: every 3x3 tap is still a constant shift, (ty - 1)(W + 1) + (tx - 1) pixels, and every kernel's addressing stays
 *     linear -- but the padded-pixel enumeration of the 9-tap weight gradient shrinks from (H + 2)(W + 2) to
 *     (H + 1)(W + 1) rows per image (13x13: 225 -> 196, -13 % of its MFMA work; conv19's weight gradient -19 %).
 *   packed weights: fp16 [Npad][K], see mcamd_pack_weights.
 *   master weights, masks, weight gradients: fp32 OIHW exactly as torch holds them.
 */
#ifndef MCAMD_H
#define MCAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCAMD_OK 0
#define MCAMD_EINVAL (-1)   /* bad argument / unsupported geometry */
#define MCAMD_ELAUNCH (-2)  /* HIP launch error */
#define MCAMD_EWORKSPACE (-3) /* workspace too small */

int mcamd_version(void);              /* 100*major + minor */
/* The MCAMD_* tuning switches are read from the environment once per process, at their first
 * use -- never per launch.  A process that changes one afterwards (tests, A/B runs) calls this to have them re-read. */
void mcamd_reload_config(void);
const char* mcamd_arch(void);         /* "gfx950" */
const char* mcamd_last_error(void);   /* per-thread, never NULL */

/* ------------------------------------------------------------------------- *
 * Convolution geometry: k x k cross-correlation, stride 1, zero pad (k-1)/2,
 * dilation 1, groups 1 -- the only form F.conv2d is called with on the hot
 * path (reference src/pruning/weightPruning/layers.py:60-64 via nets.py:796-806).
 * ------------------------------------------------------------------------- */
typedef struct mcamd_conv_geom {
    int32_t B, H, W;      /* batch, spatial size (output == input)            */
    int32_t ksize;        /* 1 or 3                                           */
    int32_t cin;          /* input channels of the weight tensor (any > 0).  The kernels consume
                             round_up(cin, 32) channels per pixel: the buffer slice must be that wide
                             and the extra channels must hold zeros.  3 when stem != 0. */
    int32_t cout;         /* output channels (any > 0)                        */
    int32_t x_ld;         /* channels per pixel of the input buffer           */
    int32_t x_choff;      /* first input channel inside the buffer            */
    int32_t stem;         /* 1: first-layer form, x is NHWC4 (x_ld == 4), cin == 3, ksize == 3 */
    int32_t pad;          /* 0: padded NHWC; 1: shared-halo form (above) of EVERY padded operand of the call -- x in
                             mcamd_conv_fwd, dy in mcamd_conv_dgrad, x and dy in mcamd_conv_wgrad.  stem == 0 only. */
    int32_t x_wrap;       /* mcamd_conv_fwd only, 0 = none.  Input channels >= x_wrap are read from channel - x_wrap of the
                             pixel: the split-operand forward on TWO activation planes [x_hi | x_lo], plane stride P,
                             with cin = 3 P, the K-concatenated weights [w_hi | w_hi | w_lo] (mcamd_pack_job.split) and
                             x_wrap = 2 P: the third part multiplies the hi plane again without a third copy of it.
                             P % 32 == 0; epilogue modes MCAMD_EPI_RAW_F32 / MCAMD_EPI_NCHW_F32; the buffer slice must hold
                             x_wrap channels. */
    int32_t x_f8;         /* mcamd_conv_fwd only, 0 = none.  P > 0 (P % 64 == 0, cin = 2 P, x_wrap = 0): the split-operand forward
                             with fp8 CORRECTION terms.  Channels [0, P) of the slice are the fp16 hi plane; the next P fp16
                             units hold 2 P OCP-e4m3 bytes [lo8 = e4m3(x_lo * 2^12) | x8 = e4m3(x * 2)] (mcamd_act_desc.planes 4),
                             the packed weights [w_hi | w8 = e4m3(w_hi * 2^e) | wlo8 = e4m3(w_lo * 2^(e + 11))], e = x_f8_wexp (mcamd_pack_job.split 2):
                             y = x_hi w_hi (fp16 MFMA) + 2^-(12 + e) (lo8 w8 + x8 wlo8) (block-scaled fp8 MFMA at twice the fp16
                             rate, same fp32 accumulators) -- x w to ~2^-15 instead of plain fp16's 2^-11, at 2/3 of the
                             x_wrap form's MFMA time and staged bytes.  x_choff = 0 (the e4m3 strings are addressed from the
                             pixel's first channel, also by a concat member writing at its offset).  Epilogue mode MCAMD_EPI_RAW_F32; only shapes for
                             which mcamd_conv_fwd_f8_ok() returns 1 (the ping-pong implicit-GEMM tiles).  Replaces the same
                             F.conv2d (reference src/pruning/weightPruning/layers.py:60-64). */
    int32_t x_f8_wexp;    /* with x_f8: the exponent the weight bytes were packed with, w8 = e4m3(w_hi * 2^x_f8_wexp), wlo8 =
                             e4m3(w_lo * 2^(x_f8_wexp + 11)) (mcamd_pack_job.f8_wexp; [-24, 40]).  Chosen per layer so that the
                             largest |w| lands in the upper binades of e4m3 (448): BatchNorm makes a layer's weight scale
                             arbitrary.  5 suits initialisation-sized weights (|w| <= 14). */
} mcamd_conv_geom;
int32_t mcamd_conv_fwd_f8_ok(const mcamd_conv_geom* g);   /* 1: mcamd_conv_fwd accepts this x_f8 geometry */

/* Output side of a convolution launch. */
#define MCAMD_EPI_RAW_F16 0   /* y: fp16 [M][y_ld] + optional per-channel partial sums (BN batch statistics) */
#define MCAMD_EPI_NCHW_F32 1  /* y: fp32 [B][cout][H][W] (+ bias)  -- the model's returned logits */
#define MCAMD_EPI_PAD_F16 2   /* y: padded NHWC fp16, leaky(acc*scale[c]+shift[c]) (inference, BN folded) */
#define MCAMD_EPI_RAW_F32 3   /* y: fp32 [M][y_ld], the accumulators unrounded (+ optional partial sums taken from the
                                 fp32 values) -- the "fp16x3" precision mode, see mcamd_act_desc.planes */
typedef struct mcamd_conv_epilogue {
    int32_t mode;
    int32_t y_ld, y_choff;     /* modes 0, 2 and 3 */
    void* y;
    const float* bias;         /* mode 1, may be NULL */
    float* stats;              /* modes 0 and 3, may be NULL: fp32 [stats_rows][2][stats_ld]; row p holds the
                                  per-channel sums (index 0) and sums of squares (index 1) over the pixels
                                  that persistent workgroup p processed (fixed order: deterministic) */
    int32_t stats_rows;        /* must equal mcamd_conv_stats_rows(geom) */
    int32_t stats_ld;          /* >= round_up(cout, 256) */
    const float* scale;        /* mode 2, may be NULL (=1) */
    const float* shift;        /* mode 2, may be NULL (=0) */
    float slope;               /* mode 2: negative-side slope (0.1 leaky, 1.0 linear) */
    int32_t* overflow;         /* modes 0 and 2, may be NULL: device flag, set to 1 when a value had to be clamped to
                                  the fp16 range (+-65504) on its way out.  fp16 outputs saturate instead of becoming
                                  inf; the scaled gradients of the backward pass (grad_scale x dX) are where that can
                                  happen, and the caller decides (train.py skips the step and halves grad_scale). */
    int32_t dst_mode;          /* mode 2 only: MCAMD_DST_PLAIN (0), MCAMD_DST_POOL or MCAMD_DST_REORG -- the inference
                                  epilogue fused with the MaxPool(2,2) / Reorg(2) that follows the block (nets.py:821,
                                  648-667): `y` is then the padded NHWC buffer at the POOLED resolution (H/2 x W/2; reorg:
                                  4 x cout channels from y_choff), H and W must be even, and the forward launch enumerates
                                  its output pixels window by window.  Forward only. */
    void* y2;                  /* dst_mode POOL, may be NULL: a second, FULL-resolution padded copy of leaky(bn(conv)) (the
                                  route that reads the block beside its pool: conv13 of yolov2-voc.cfg) */
    int32_t y2_ld, y2_choff;
    /* (mode 2 writes the standard padded form: inference engines do not use the shared-halo one) */
    int32_t concurrent;        /* mcamd_conv_dgrad only, 0 / 1: the caller runs other kernels on another stream at the
                                  same time (the training engine: the weight gradients of the layers behind).  The launch
                                  then picks the workgroup tile with the least CU-time even if it fills only 40-80 % of the
                                  CUs, instead of the tile with the shortest launch on an otherwise idle GPU. */
} mcamd_conv_epilogue;

/* Rows of the BatchNorm partial-sum slab a forward launch of this geometry writes (epilogue mode 0). */
int32_t mcamd_conv_stats_rows(const mcamd_conv_geom* g);
/* The same for a given epilogue mode (MCAMD_EPI_RAW_F16 or MCAMD_EPI_RAW_F32: the fp32 epilogue only exists in
 * the LDS-staged implicit-GEMM kernels, so the slab shape differs for the layers that otherwise take a streaming kernel). */
int32_t mcamd_conv_stats_rows_mode(const mcamd_conv_geom* g, int32_t mode);

/* Workgroup tile {BM, BN, BK, kernel} the forward (dgrad == 0) or dgrad launch (1; 2 = with epilogue.concurrent set) of
 * this geometry uses:
 * kernel 0 = igemm_kernel<BM,BN,..,BK,..> (one tap per K chunk),
 * 2 = igemm_pp_kernel (ping-pong, one workgroup per CU), 1 = stem_fwd_kernel (first layer, no LDS staging),
 * 4 = small3x3_kernel (narrow 3x3 layers on huge images, no LDS staging), 5 = win3x3_kernel (rolling LDS window),
 * 6 = wres_kernel (3x3 layers with one 64-channel input block: weights resident in registers, one activation window per
 * M tile of 128 padded pixels). */
int mcamd_conv_tile_info(const mcamd_conv_geom* g, int32_t dgrad, int32_t out[4]);

/* Packed-weight sizes (elements of fp16) for a geometry. */
int64_t mcamd_packed_elems_fwd(const mcamd_conv_geom* g);
int64_t mcamd_packed_elems_dgrad(const mcamd_conv_geom* g);

/* Physical channel order of one convolution (NULL pointers = identity).  With filter pruning the engine keeps
 * the surviving filters first and computes only those: `g->cout` / `g->cin` then describe the PHYSICAL problem
 * and the weight / mask / gradient tensors keep the module's OIHW order with `g->cin` input channels per filter:
 * physical filter n is tensor row rows[n], physical input channel c is tensor column cols[c]. */
typedef struct mcamd_chan_map {
    const int32_t* rows;   /* device int32[g->cout] or NULL */
    const int32_t* cols;   /* device int32[g->cin] or NULL */
} mcamd_chan_map;

/* OIHW fp32 master (optionally * mask) -> fp16 kernel layouts.  Replaces the per-forward
 * `self.weight * mask_var` of layers.py:59 (done once per optimizer step here).
 *   fwd  : [Npad][kpos(t, c)] = w[n][c][ty][tx],  t = ty*k + tx;  Npad = roundup(cout,256); pad rows zero
 *          (stem: [Npad][ty*32 + tx*4 + c], other slots zero)
 *   dgrad: [Cpad][kpos(t, n)] = w[n][c][k-1-ty][k-1-tx]; cout_p = roundup(cout,32); Cpad = roundup(cin,256)
 *   kpos(t, c) = (c / kb) * k*k*kb + t * kb + c % kb over the padded channel count chp (cin_tap = roundup(cin,32)
 *   or cout_p), kb = 64 if chp % 64 == 0 else 32: the K axis runs [channel block][tap][channel in block], so the
 *   k*k shifted reads of one activation line are consecutive K chunks (L2 hits instead of k*k streams).
 * Either destination may be NULL.  `map` (may be NULL): gather rows / columns of w and mask, see mcamd_chan_map. */
int mcamd_pack_weights(const mcamd_conv_geom* g, const float* w_oihw, const float* mask_oihw,
                       const mcamd_chan_map* map, void* wp_fwd, void* wp_dgrad, void* stream);

/* All layers of a network in ONE launch (the per-step re-pack of engine.py): `jobs_dev` is a DEVICE array of
 * `njobs` descriptors, one per layer.  A workgroup takes a 32-filter x 32-channel tile of one layer: the
 * k*k taps of every (filter, channel) pair are read as one contiguous run (x mask), transposed through LDS
 * and written to BOTH packed layouts in 64-byte pieces, so the fp32 master and the mask are read once.
 * Only real entries are written -- pad rows / pad channels of the destinations must already be zero
 * (they never change).  The stem layer is not supported here (use mcamd_pack_weights). */
typedef struct mcamd_pack_job {
    const float* w;            /* OIHW fp32 master */
    const float* mask;         /* OIHW fp32 or NULL */
    void* dst_fwd;             /* fp16 forward layout, or NULL */
    void* dst_dgrad;           /* fp16 dgrad layout, or NULL */
    const int32_t* rows;       /* channel maps as in mcamd_chan_map, or NULL */
    const int32_t* cols;
    int64_t first_tile;        /* sum of ceil(cout/32)*ceil(cin/32) over the preceding jobs */
    int32_t cout, cin, ksize;  /* physical geometry */
    int32_t split;             /* 0: plain fp16 forward packing.  1: the split-operand forward packing of the "fp16x3" /
                                  "mixed" precisions, [w_hi | w_hi | w_lo] along the input channels of a 3 * cin wide row
                                  (w_hi = fp16(w * mask), w_lo = fp16(w * mask - w_hi)), to be multiplied with
                                  [x_hi | x_lo | x_hi] activation planes; dst_fwd then has the size of a geometry with
                                  3 * cin input channels.  2: the fp8-correction packing of mcamd_conv_geom.x_f8,
                                  [w_hi fp16 | w8 | wlo8 e4m3 bytes] in a row of 2 * cin fp16 units per tap (cin % 64 == 0).
                                  The dgrad packing is plain in every case. */
    int32_t f8_wexp;           /* split 2: w8 = e4m3(w_hi * 2^f8_wexp), wlo8 = e4m3(w_lo * 2^(f8_wexp + 11)); the consumer's
                                  mcamd_conv_geom.x_f8_wexp must say the same */
} mcamd_pack_job;
int mcamd_pack_weights_many(const mcamd_pack_job* jobs_dev, int32_t njobs, int64_t total_tiles, void* stream);

/* y = conv(x, w) -- replaces F.conv2d at layers.py:60-64. */
int mcamd_conv_fwd(const mcamd_conv_geom* g, const void* x, const void* wp_fwd,
                   const mcamd_conv_epilogue* epi, void* stream);

/* dx = conv_transpose(dy, w) -- autograd's input gradient of the same call.
 * `dy` is padded NHWC fp16 [B][H+2][W+2][dy_ld] (zero halo); g->cin/cout keep their forward
 * meaning; the result has g->cin channels.  epi->mode 0 (fp16 [M][y_ld]) or 1 (fp32 NCHW). */
int mcamd_conv_dgrad(const mcamd_conv_geom* g, const void* dy, int32_t dy_ld, int32_t dy_choff,
                     const void* wp_dgrad, const mcamd_conv_epilogue* epi, void* stream);

/* dW = wgrad(x, dy) * mask / grad_scale, written as fp32 OIHW -- autograd's weight gradient of
 * `self.weight * mask_var` followed by F.conv2d.  Deterministic (slab reduction, no atomics).
 * Fully pruned filters are skipped through `map` (may be NULL): the caller keeps the surviving filters first
 * in its channel order and passes a geometry whose `cout` is the kept count, so that forward, dgrad and
 * wgrad all run on the kept filters only (modelcompression_amd/engine.py, "filter compaction").  With a map,
 * dW[rows[n]][cols[c]] receives the gradient of physical (n, c) (the mask is read at the same place) and
 * tensor rows that are not in rows[] are NOT written: the caller zeroes dw_oihw first.
 * `dbias` (fp32[cout], may be NULL) receives sum over pixels of dy / grad_scale. */
size_t mcamd_conv_wgrad_workspace_bytes(const mcamd_conv_geom* g);
int mcamd_conv_wgrad(const mcamd_conv_geom* g, const void* x, const void* dy, int32_t dy_ld,
                     int32_t dy_choff, const float* mask_oihw, const mcamd_chan_map* map,
                     float grad_scale, float* dw_oihw, float* dbias, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * BatchNorm2d (eps, momentum as given; torch defaults 1e-5 / 0.1 at nets.py:802)
 * + LeakyReLU(0.1) (nets.py:809) + MaxPool2d(2,2) (nets.py:821) + Reorg(2)
 * (nets.py:648-667) + route/concat (nets.py:738-746), fused around the convs.
 * ------------------------------------------------------------------------- */
/* Batch statistics from the conv epilogue's partial sums -> affine coefficients.
 * scale = gamma*invstd, shift = beta - mean*scale.  training != 0: batch stats (biased var),
 * running stats updated with the unbiased var; training == 0: running stats. */
/* `chan_perm` (device int32[C], may be NULL = identity): statistics and the output vectors are in the
 * kernels' physical channel order, gamma/beta/running_* in the module's order; physical channel c reads and
 * updates index chan_perm[c] of them.  The engine keeps the filters that survive filter pruning first (its
 * convolutions then run on the kept filters only) and passes that order here. */
int mcamd_bn_coeffs(const float* stats, int32_t stats_rows, int32_t stats_ld, int32_t C, int64_t count,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int32_t training,
                    float* scale, float* shift, float* save_mean, float* save_invstd,
                    const int32_t* chan_perm, void* stream);

/* As mcamd_bn_coeffs; `ones_channel` >= 0 names ONE physical channel whose coefficients are forced to scale 0 /
 * shift 1 after the statistics update: the BatchNorm + LeakyReLU pass then writes 1.0 at every interior pixel of that
 * channel (0 stays in the halo).  The engine uses the first dead channel behind the kept filters of a filter-pruned
 * layer this way when the consumer folds the dead inputs (mcamd_fold_weights); -1 = none. */
int mcamd_bn_coeffs_ex(const float* stats, int32_t stats_rows, int32_t stats_ld, int32_t C, int64_t count,
                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                       float momentum, float eps, int32_t training,
                       float* scale, float* shift, float* save_mean, float* save_invstd,
                       const int32_t* chan_perm, int32_t ones_channel, void* stream);

#define MCAMD_DST_PLAIN 0  /* same resolution */
#define MCAMD_DST_POOL 1   /* 2x2/2 max pool */
#define MCAMD_DST_REORG 2  /* reorg stride 2: out channel = (hs*2+ws)*C + c at (h/2, w/2) */
typedef struct mcamd_act_desc {
    int32_t B, H, W, C;        /* conv output size */
    const void* y; int32_t y_ld, y_choff;   /* raw conv output fp16 [B*H*W][y_ld] */
    const float* scale; const float* shift; /* per channel */
    float slope;               /* 0.1 (leaky) or 1.0 (linear) */
    int32_t mode;              /* MCAMD_DST_* for dst */
    void* dst; int32_t dst_ld, dst_choff;   /* padded NHWC fp16 at the mode's resolution */
    void* dst2; int32_t dst2_ld, dst2_choff;/* optional second copy, PLAIN resolution (route of a pooled layer) */
    int32_t y_dtype;           /* 0: y is fp16 (MCAMD_EPI_RAW_F16), 1: y is fp32 (MCAMD_EPI_RAW_F32) */
    int32_t planes;            /* 1 (0 is read as 1), or 3 = split storage for the "fp16x3" precision mode: the
                                  activation v is written as the fp16 pair hi = fp16(v), lo = fp16(v - hi) plus a second
                                  copy of hi, at channels choff + {0, 1, 2} * plane stride.  A convolution whose input
                                  is the 3*C-channel run [hi | lo | hi] and whose packed weights are [w_hi | w_hi | w_lo]
                                  (w_hi = fp16(w), w_lo = fp16(w - w_hi)) accumulates x_hi*w_hi + x_lo*w_hi + x_hi*w_lo
                                  in fp32 on the fp16 MFMA path: operand rounding drops from 2^-11 to ~2^-21, which
                                  is what the reference's fp32 F.conv2d (layers.py:60-64) needs over 23 layers for
                                  1e-3 logits (tools/error_budget.py).  Reading channels [0, C) alone is the plain
                                  fp16 activation.
                                  2 = hi and lo only: the consumer reads the hi plane twice (mcamd_conv_geom.x_wrap),
                                  4 instead of 6 bytes written per activation.
                                  4 = hi and, one plane stride further, the e4m3 correction bytes [lo8 | x8] of a consumer
                                  with mcamd_conv_geom.x_f8 (each `plane` BYTES long; fp32 y only). */
    int32_t dst_plane, dst2_plane; /* plane strides (channels, multiples of 8) of dst / dst2 when planes >= 2 */
    int32_t dst_pad, dst2_pad; /* 0 / 1: dst, dst2 are in the padded / the shared-halo form (each at its own resolution) */
    const float* border;       /* optional fp32 [16][C], NULL = none: added to the raw conv output before the
                                  affine step, row = border class of the pixel (bit 0: h == 0, bit 1: h == H-1,
                                  bit 2: w == 0, bit 3: w == W-1).  Physically slim filter-pruned models fold the
                                  constant output of their removed input channels into this table, because zero
                                  padding clips it differently at the borders (BASELINE config 5; the reference
                                  only states slim convs as a conclusion, README.md:19). */
    int32_t planes2;           /* storage form of dst2 when it differs from dst's (its consumer is another convolution);
                                  0 = `planes` */
    void* pool_act;            /* optional (mode MCAMD_DST_POOL), NULL = none: a FULL-RESOLUTION fp16 copy of the activation,
                                  padded NHWC / shared-halo form per pool_act_pad, pool_act_ld channels per pixel, channels
                                  [0, C) -- what mcamd_bn_act_bwd reads instead of the raw output (mcamd_act_bwd_desc.act)
                                  in the backward pass of a MaxPool block.  The element the pool took (the first maximum
                                  of the four UNROUNDED activations in (h, w) scan order, nn.MaxPool2d(2, 2), reference
                                  src/nets.py:821) is stored as the STRICT maximum of the window: a neighbour that rounds to
                                  the same fp16 value is written one fp16 step lower (~6e-4 of the elements of a random
                                  tensor), so that the backward pass routes the gradient where the fp32 forward did. */
    int32_t pool_act_ld, pool_act_pad;
} mcamd_act_desc;
int mcamd_bn_act_fwd(const mcamd_act_desc* d, void* stream);

typedef struct mcamd_act_bwd_desc {
    int32_t B, H, W, C;
    const void* y; int32_t y_ld, y_choff;   /* saved raw conv output */
    const float* scale; const float* shift; const float* mean; const float* invstd; /* from mcamd_bn_coeffs */
    float slope;
    int32_t mode;              /* how `g` maps onto this layer's output (MCAMD_DST_*) */
    const void* g; int32_t g_ld, g_choff;   /* fp16 gradient wrt dst, [pixels at mode's resolution][g_ld] */
    const void* g2; int32_t g2_ld, g2_choff;/* optional gradient wrt dst2 (PLAIN resolution) */
    void* dy; int32_t dy_ld, dy_choff;      /* out: padded NHWC fp16 gradient wrt raw conv output */
    float* dgamma; float* dbeta;            /* out fp32 [C], already divided by grad_scale (may be NULL) */
    float grad_scale;          /* the incoming gradients are grad_scale x the true ones (fp16 range) */
    const float* dy_keep;      /* optional fp32 [C]: 0 marks a fully pruned filter; its dY channel is written as
                                  zero (its weights are zero, so dgrad/wgrad never need it -- and a dead filter has
                                  zero batch variance, which would otherwise blow dY up by 1/sqrt(eps)) */
    const int32_t* chan_perm;  /* optional device int32[C]: dgamma / dbeta of physical channel c are written to
                                  index chan_perm[c] (see mcamd_bn_coeffs) */
    int32_t y_dtype;           /* 0: y is fp16, 1: y is fp32 (and the pooled argmax is taken on unrounded activations,
                                  as the split-storage forward keeps them) */
    int32_t* overflow;         /* optional device flag, set to 1 when a dY value was clamped to +-65504 (see
                                  mcamd_conv_epilogue.overflow) */
    int32_t dy_pad;            /* 0 / 1: dy is in the padded / the shared-halo form */
    int32_t skip_dead_param_grads; /* n > 0: dgamma / dbeta of the physical channels c >= n are NOT written -- the
                                  consumer that folded those dead channels delivers their gradients
                                  (mcamd_unfold_wgrad) and `g` holds nothing for them; 0 = write all */
    const void* act;           /* optional, PLAIN blocks without g2: the activation the forward pass stored for the consumer
                                  (fp16, padded NHWC / shared-halo form per act_pad, the hi plane of split storage) at channels
                                  [act_choff, act_choff + C) of rows of act_ld.  LeakyReLU is invertible: z = act > 0 ? act :
                                  act / slope, xhat = (z - beta) / gamma -- so `y` (fp32, y_dtype 1, or NULL) is read only by
                                  the threads that hold a channel with gamma == 0, where xhat cannot be recovered (NULL: the
                                  dgamma of such a channel is written as 0).  With an fp32 `y` (split-operand precisions) the
                                  two passes read half the bytes; the result carries the fp16 rounding of the stored
                                  activation, as every backward tensor does (nn.BatchNorm2d + nn.LeakyReLU backward,
                                  reference src/nets.py:802-809 under autograd).
                                  Mode MCAMD_DST_POOL (with or without g2): `act` is the FULL-RESOLUTION fp16 copy of the block's
                                  activation the forward pass wrote as mcamd_act_desc.pool_act (H x W, act_choff 0); the
                                  window's argmax is the maximum of the four stored values (strict by construction there)
                                  and every element's LeakyReLU side is the sign of its stored value (nn.MaxPool2d(2, 2)
                                  backward, reference src/nets.py:821). */
    int32_t act_ld, act_choff, act_pad;
} mcamd_act_bwd_desc;
size_t mcamd_bn_act_bwd_workspace_bytes(const mcamd_act_bwd_desc* d);
int mcamd_bn_act_bwd(const mcamd_act_bwd_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Dead INPUT channels of a filter-pruned network (csrc/fold.hip).  A fully pruned filter
 * feeds its consumer the constant leaky(beta) (0 in the halo); all such channels together
 * act like ONE channel of ones convolved with the folded filter sum_c leaky(beta_c) W[n][c].
 * The engine runs the consumer's forward / dgrad / wgrad on the producer's kept channels + that
 * one channel with the augmented weights built by mcamd_fold_weights, and maps the augmented
 * weight gradient back with mcamd_unfold_wgrad (which also yields dbeta of the producer's dead
 * channels).  Exact on the reference's semantics: F.conv2d(x, weight * mask) at layers.py:59-64
 * and the BatchNorm / LeakyReLU / MaxPool of nets.py:798-821 between the two convolutions.
 *   w, mask : the consumer's OIHW fp32 master [cout_t][cin_t][k][k] and mask (or NULL)
 *   rows    : device int32[n] physical filter -> tensor row (NULL = identity; n = filters computed)
 *   cols    : device int32[cin_t] physical input channel -> tensor column, kept channels first
 *             (NULL = identity); the first cin_k are taken as they are, the rest are folded
 *   beta    : the PRODUCER's BatchNorm bias in the module's channel order, fp32[cin_t]
 *   waug / dwaug : fp32 [n][cin_aug][k][k], cin_aug >= cin_k + 1 (the convolution kernels want a channel count
 *             that is a multiple of 8): column cin_k is the folded filter, the columns behind it are zero
 * ------------------------------------------------------------------------- */
typedef struct mcamd_fold_desc {
    const float* w; const float* mask;
    const int32_t* rows; const int32_t* cols;
    const float* beta; float slope;
    int32_t n, cin_t, cin_k, cin_aug, ksize;
} mcamd_fold_desc;
int mcamd_fold_weights(const mcamd_fold_desc* d, float* waug, void* stream);
/* Every folding layer of a network in ONE launch (the per-step rebuild of engine.py): `jobs_dev` is a DEVICE array of
 * `njobs` descriptors; first_block = sum of d.n over the preceding jobs, total_blocks = sum over all jobs. */
typedef struct mcamd_fold_job {
    mcamd_fold_desc d;
    float* waug;
    int64_t first_block;
} mcamd_fold_job;
int mcamd_fold_weights_many(const mcamd_fold_job* jobs_dev, int32_t njobs, int64_t total_blocks, void* stream);
/* dw_oihw rows outside rows[] are not written (the caller zeroes them).  prod_dbeta / prod_dgamma: the producer's
 * gradient vectors in the module's order; dead channels receive dbeta (`accumulate` != 0: added to what an earlier
 * consumer of the same producer wrote) and dgamma = 0. */
int mcamd_unfold_wgrad(const mcamd_fold_desc* d, const float* dwaug, float* dw_oihw, float* prod_dbeta,
                       float* prod_dgamma, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------- *
 * The first block as ONE unit: conv1 (3 -> 32 filters, 3x3) + BatchNorm2d + LeakyReLU
 * + MaxPool2d(2,2) (reference src/nets.py:798-821 for the first [convolutional] +
 * [maxpool] pair of yolov2-voc.cfg, F.conv2d at layers.py:60-64) without the block's
 * full-resolution tensors: neither the raw conv output (709 MB at B=64) nor its
 * gradient ever exists in HBM.  Batch statistics come from the 27x27 Gram matrix of
 * the image windows (y = W v is linear in the 27-value window v), the forward pass
 * recomputes nothing, the backward pass recomputes y from the image and reduces the
 * weight gradient algebraically (csrc/conv_stem_block.hip).  Deterministic.
 *   x      : stem input, padded NHWC4 fp16 (as mcamd_conv_geom.stem)
 *   wp     : packed stem weights from mcamd_pack_weights (mask already applied)
 *   dst    : pooled output, padded NHWC fp16 [B][H/2+2][W/2+2][dst_ld], pixel (0,0,0)
 *   g      : gradient wrt dst, fp16 [B*(H/2)*(W/2)][g_ld], grad_scale x the true one
 * Requires W % 32 == 0 and an even H.  `workspace` (mcamd_stem_block_workspace_bytes())
 * carries S and W*C from a training-mode forward call to the backward call of the same
 * step: the caller must not touch it in between.
 * training != 0: batch statistics -> scale/shift/save_mean/save_invstd are WRITTEN and the
 * running statistics updated; training == 0: scale/shift are READ (mcamd_bn_coeffs with
 * stats == NULL provides them from the running statistics).
 * ------------------------------------------------------------------------- */
typedef struct mcamd_stem_block_desc {
    int32_t B, H, W;                      /* conv resolution; cin = 3 */
    const void* x;
    const void* wp;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;   /* may be NULL */
    float momentum, eps;
    int32_t training;
    float* scale; float* shift; float* save_mean; float* save_invstd;   /* fp32 [32] each */
    float slope;                          /* 0.1 (leaky) or 1.0 (linear) */
    void* dst; int32_t dst_ld, dst_choff;
    /* backward only */
    const void* g; int32_t g_ld, g_choff;
    const float* mask;                    /* OIHW fp32 [32][3][3][3] or NULL */
    float grad_scale;
    float* dw;                            /* out: OIHW fp32, x mask, / grad_scale */
    float* dgamma; float* dbeta;          /* out fp32 [32], / grad_scale (may be NULL) */
    int32_t cout;                         /* filters: 32 (0 is read as 32); 8, 16 or 24 are accepted by the forward pass with
                                             training == 0 (physically slim models): `dst` still receives 32 channels, the
                                             ones past cout as zeros, and scale / shift hold cout entries */
    int32_t planes;                       /* forward: 1 (0 is read as 1), or 3 = split storage of the pooled output as in
                                             mcamd_act_desc.planes: hi | lo | hi in three ADJACENT 32-channel planes
                                             [dst_choff, dst_choff + 96) -- the "mixed" precision mode keeps plain fp16
                                             operands on this block (image and weights: 5.2e-4 -> 5.4e-4 on the logits,
                                             tools/error_budget.py) but hands its consumer an unrounded activation;
                                             2 = hi | lo in [dst_choff, dst_choff + 64) (consumer with x_wrap = 64) */
    /* SPLIT OPERANDS (round 4; both NULL = plain fp16 operands): the reference multiplies fp32 by fp32 (F.conv2d,
     * layers.py:60-64), and in TRAINING mode the operand rounding of this block alone moves the region-layer logits by
     * 1.9e-2.  With x_lo = fp16(x - fp16(x)) in a second NHWC4 image (mcamd_nchw_f32_to_nhwc4_split) and wp_lo =
     * fp16(w - fp16(w)) in the same packing as wp (mcamd_pack_stem_split) the pass accumulates
     * x_hi w_hi + x_lo w_hi + x_hi w_lo in fp32.  Honoured by mcamd_stem_block_stats and by mcamd_stem_block_fwd with
     * training == 0 (scale / shift read); the backward pass multiplies plain operands, as every backward pass does. */
    const void* x_lo;
    const void* wp_lo;
} mcamd_stem_block_desc;
size_t mcamd_stem_block_workspace_bytes(void);
/* dst == NULL with training != 0: the statistics half only -- Gram sums, scale / shift / save_mean / save_invstd and the
 * workspace context a later mcamd_stem_block_bwd reads; nothing is written to an output. */
int mcamd_stem_block_fwd(const mcamd_stem_block_desc* d, void* workspace, size_t workspace_bytes, void* stream);
int mcamd_stem_block_bwd(const mcamd_stem_block_desc* d, void* workspace, size_t workspace_bytes, void* stream);
/* Batch statistics of the block's conv output WITHOUT storing it (the training-mode forward on split operands is two
 * passes over the image: this one, mcamd_bn_coeffs on its slab, then mcamd_stem_block_fwd with training == 0):
 * stats[row][0][c] = sum over the pixels of workgroup `row` of y[.][c], stats[row][1][c] = sum of squares, fp32
 * [mcamd_stem_block_stats_rows(d)][2][stats_ld >= 32], every row written, fixed order (deterministic) -- the slab
 * mcamd_bn_coeffs takes, as mcamd_conv_epilogue.stats.  nn.BatchNorm2d's batch statistics (src/nets.py:802). */
int32_t mcamd_stem_block_stats_rows(const mcamd_stem_block_desc* d);
int mcamd_stem_block_stats(const mcamd_stem_block_desc* d, float* stats, int32_t stats_rows, int32_t stats_ld, void* stream);

/* ------------------------------------------------------------------------- *
 * Layout conversion at the model boundary (Darknet.forward takes/returns NCHW fp32,
 * nets.py:720-774).
 * ------------------------------------------------------------------------- */
/* src fp32 [B][C][H][W] * mul -> dst padded NHWC fp16 channels [choff, choff+C).  `overflow` (may be NULL): device
 * flag set to 1 when a value was clamped to +-65504 (the incoming logit gradient x grad_scale). */
int mcamd_nchw_f32_to_padded_nhwc_f16(const float* src, int32_t B, int32_t C, int32_t H, int32_t W,
                                      float mul, void* dst, int32_t dst_ld, int32_t dst_choff, int32_t* overflow,
                                      void* stream);
/* As mcamd_nchw_f32_to_padded_nhwc_f16 with the destination in the shared-halo form when pad == 1. */
int mcamd_nchw_f32_to_padded_nhwc_f16_pad(const float* src, int32_t B, int32_t C, int32_t H, int32_t W,
                                          float mul, void* dst, int32_t dst_ld, int32_t dst_choff, int32_t pad,
                                          int32_t* overflow, void* stream);
/* The same into split storage (mcamd_act_desc.planes == 3): channel c of the image is written as hi = fp16(v) at
 * dst_choff + c, lo = fp16(v - hi) at dst_choff + plane + c and hi again at dst_choff + 2 * plane + c -- the network
 * input of the "fp16x3" / "mixed" precision modes (nets.py:720 takes the image as fp32 NCHW). */
int mcamd_nchw_f32_to_padded_nhwc_f16_split(const float* src, int32_t B, int32_t C, int32_t H, int32_t W,
                                            void* dst, int32_t dst_ld, int32_t dst_choff, int32_t plane, void* stream);

/* The image for the split-operand first block (mcamd_stem_block_desc.x / x_lo): src fp32 [B][3][H][W] -> two padded
 * NHWC4 fp16 images [B][H+2][W+2][4], hi = fp16(v) and lo = fp16(v - hi), channel 3 zero; the halo is not written
 * (zero it once).  nets.py:720 takes the image as fp32 NCHW. */
int mcamd_nchw_f32_to_nhwc4_split(const float* src, int32_t B, int32_t H, int32_t W, void* hi, void* lo, void* stream);
/* The stem packing of mcamd_pack_weights (g->stem) for split operands: wp_hi = fp16(w * mask), wp_lo =
 * fp16(w * mask - wp_hi), both [Npad][96]. */
int mcamd_pack_stem_split(const float* w_oihw, const float* mask_oihw, int32_t cout, void* wp_hi, void* wp_lo, void* stream);

/* The first convolution of the split-operand precisions in fp32 on the vector ALUs, straight from the image
 * (F.conv2d(x, weight * mask, None, 1, 1) at layers.py:60-64 for a 3-channel input, 3x3 kernel, 32 filters):
 *   x_nchw fp32 [B][3][H][W] -> y fp32 [B*H*W][y_ld] (channels 0..31), exact fp32 products and sums;
 *   stats (may be NULL): fp32 [stats_rows][2][stats_ld] partial sums / sums of squares of y per filter, one row per
 *   workgroup, stats_rows == mcamd_stem_conv_f32_stats_rows(); mcamd_bn_coeffs adds the rows (training mode);
 *   weff_scratch: 32 * 27 floats of device memory the call may overwrite (weight * mask).
 * Other filter counts are refused (the caller then multiplies split operands through mcamd_conv_fwd). */
int32_t mcamd_stem_conv_f32_stats_rows(void);
int mcamd_stem_conv_f32(const float* x_nchw, int32_t B, int32_t H, int32_t W, const float* w_oihw,
                        const float* mask_oihw, int32_t cout, float* weff_scratch, float* y, int32_t y_ld,
                        float* stats, int32_t stats_rows, int32_t stats_ld, void* stream);

/* ------------------------------------------------------------------------- *
 * Region loss of the training step (reference src/nets.py:282-635: build_targets + RegionLoss.forward, called at
 * train.py:224): loss AND d(loss)/d(output) in one pass over the logits; the targets are built per image in LDS.
 *   output : fp32 [B][num_anchors * (5 + num_classes)][H][W] -- what Darknet.forward returns (nets.py:720-774)
 *   target : fp32 [B][max_boxes * 5] rows of (class, x, y, w, h), normalised to [0, 1]; a row list ends at the first x == 0
 *   anchors: num_anchors (w, h) pairs in grid units (the [region] block of the cfg)
 *   loss   : device fp32 scalar = the value RegionLoss.forward returns (sum of the six terms / B)
 *   grad   : device fp32, shaped like output: d(loss)/d(output)
 *   counts : optional device int32[2] (nGT, nCorrect; the reference prints them), ADDED to -- zero them first
 * The reference's arithmetic quirks are kept (modelcompression_amd/region_loss.py lists them).  Deterministic.
 * ------------------------------------------------------------------------- */
typedef struct mcamd_region_desc {
    const float* output; const float* target;
    int32_t B, H, W, num_anchors, num_classes, max_boxes;   /* max_boxes must be 50 (nets.py:312) */
    float anchors[16];
    float coord_scale, noobject_scale, object_scale, class_scale, thresh;
} mcamd_region_desc;
size_t mcamd_region_loss_workspace_bytes(int32_t B);
int mcamd_region_loss(const mcamd_region_desc* d, float* loss, float* grad, int32_t* counts, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Launch plans: a whole forward or backward pass as ONE library call.
 * The reference executes Darknet.forward as one Python call per torch module (src/nets.py:720-774) and autograd
 * replays them; here the caller walks its layer list ONCE between mcamd_plan_begin and mcamd_plan_end: every
 * launch-type entry point above (pack / fold / conv fwd / dgrad / wgrad / bn_coeffs / bn_act fwd + bwd / stem block /
 * layout conversion / mcamd_stream_wait / mcamd_memset_zero) then RECORDS its arguments -- descriptors copied by
 * value -- instead of launching, on the recording host thread only.  mcamd_plan_run replays them: same launches, same
 * order, same streams (bit-identical results), without the per-launch host overhead of a language binding.
 * A plan holds the raw device pointers it was recorded with: the caller keeps those buffers in place and records a
 * new plan when one moves.  Argument errors of a recorded call surface from mcamd_plan_run.  Queries
 * (*_workspace_bytes, tile_info, ...) are never recorded.
 *   streams  : the hipStream_t's (as void*) the recording may see, slot 0 first; mcamd_plan_run takes the streams to
 *              replay on in the same slot order (normally the same ones)
 *   segments : mcamd_plan_mark() closes a segment; mcamd_plan_run(p, lo, hi, ...) replays segments [lo, hi) -- the
 *              data-parallel reducer is called between segments, when a gradient slice has become final
 * ------------------------------------------------------------------------- */
typedef struct mcamd_plan mcamd_plan;
int mcamd_plan_begin(void* const* streams, int32_t nstreams);
int32_t mcamd_plan_mark(void);                    /* -> index of the segment that starts here */
mcamd_plan* mcamd_plan_end(void);                 /* NULL when a recorded call failed (mcamd_last_error) */
int32_t mcamd_plan_segments(const mcamd_plan* p);
int32_t mcamd_plan_launches(const mcamd_plan* p); /* recorded calls (a call may launch several kernels) */
int mcamd_plan_run(mcamd_plan* p, int32_t seg_lo, int32_t seg_hi, void* const* streams, int32_t nstreams);
void mcamd_plan_destroy(mcamd_plan* p);
/* `waiter` waits for all work enqueued so far on `signal` (event record + stream wait; the two-stream backward pass
 * hands weight gradients to its second stream this way).  Recordable. */
int mcamd_stream_wait(void* waiter_stream, void* signal_stream);
/* hipMemsetAsync(dst, 0, bytes) -- the zero rows of a filter-pruned layer's weight gradient.  Recordable. */
int mcamd_memset_zero(void* dst, size_t bytes, void* stream);

/* The overflow / non-finite policy of a training step as ONE launch (modelcompression_amd/train.py StepGuard; the
 * reference checks the loss on the host, train.py:226-231):
 *   flags[0] = any of the n_engine device flags engine_overflow[i] (mcamd_conv_epilogue.overflow & co.) is set,
 *   flags[1] = *loss is not finite (loss may be NULL), flags[2] = *transport_overflow != 0 (may be NULL);
 *   the int flags that were read are reset to 0; found (may be NULL) = flags[0] + flags[1] + flags[2] -- what torch's
 *   fused SGD takes as `found_inf`.
 * With n_engine == 0, loss == NULL and transport_overflow == NULL only `found` is recomputed from `flags` (after the
 * MAX all-reduce of the flags over the data-parallel ranks).  engine_overflow: host array of <= 8 device pointers. */
int mcamd_step_flags(const int32_t* const* engine_overflow, int32_t n_engine, const float* loss,
                     int32_t* transport_overflow, float* flags, float* found, void* stream);

/* ------------------------------------------------------------------------- *
 * Pruning (reference src/pruning/weightPruning/methods.py).
 * ------------------------------------------------------------------------- */
/* k-th smallest |w| (0-based, ascending) over `nseg` fp32 tensors -- the order statistic
 * np.percentile selects at methods.py:18.  Writes s[k] and s[min(k+1, n-1)] as two fp32 values to
 * `out2` (device).  ptrs/counts are HOST arrays of device pointers / element counts. */
size_t mcamd_kth_magnitude_workspace_bytes(void);
int mcamd_kth_magnitude(const float* const* ptrs, const int64_t* counts, int32_t nseg, int64_t k,
                        float* out2, void* workspace, size_t workspace_bytes, void* stream);
/* mask[i] = |w[i]| > *threshold ? 1.f : 0.f  (strict, methods.py:24-25); threshold is a device fp32. */
int mcamd_magnitude_mask(const float* w, int64_t n, const float* threshold, float* mask, void* stream);
/* Per-filter score of methods.py:43-51 in numpy's fp32 summation order:
 * out[o] = (mean_sq[o] / sqrt(sum_o mean_sq^2)) / max_o(...), one launch per conv layer. */
size_t mcamd_filter_scores_workspace_bytes(int32_t cout);
int mcamd_filter_scores(const float* w_oihw, int32_t cout, int32_t cin, int32_t kh, int32_t kw,
                        float* scores, void* workspace, size_t workspace_bytes, void* stream);
/* Only the first stage: out[o] = sum(w[o]^2) / (cin*kh*kw) in numpy's summation order
 * (prune_one_filter ranks before the /max step, methods.py:104-109).  Same workspace size. */
int mcamd_filter_mean_square(const float* w_oihw, int32_t cout, int32_t cin, int32_t kh, int32_t kw,
                             float* mean_sq, void* workspace, size_t workspace_bytes, void* stream);
/* mask[o][...] = keep[o] ? 1.f : 0.f over a [cout][per_filter] tensor (methods.py:74). */
int mcamd_filter_mask(const int32_t* keep, int32_t cout, int64_t per_filter, float* mask, void* stream);
/* count of exact zeros in an fp32 tensor, added to *out (device int64) -- prune_rate, utils.py:76-80. */
int mcamd_count_zeros(const float* w, int64_t n, unsigned long long* out, void* stream);
/* sum(p * |m - 1|) accumulated in fp32 into *out -- are_masks_consistent, utils.py:122-133. */
int mcamd_masked_residual(const float* w, const float* mask, int64_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCAMD_H */
