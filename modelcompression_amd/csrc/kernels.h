// Internal launch descriptors shared between the kernel translation units and api.hip.
#pragma once
#include "common.h"

struct IgemmArgs {
    const half_t* x;
    const half_t* w;
    void* y;
    const float* bias;
    float* stats;
    const float* scale;
    const float* shift;
    long long x_img_stride;  // elements between images of x
    int x_row_stride;        // elements between padded rows of x
    int x_ld;                // elements per pixel of x
    int x_off;               // channel offset added to every row base
    int H, W, HW, M, N;
    int ktot;                // ntaps * cin_tap
    int cin_tap;
    int ntaps;
    int kb;                  // channel-block size of the packed K order [cb][tap][kb]: 64 when cin_tap % 64 == 0, else 32
    int tap_off[9];          // element offset of tap t from the row base
    int mode;
    int y_ld, y_choff;
    int stats_ld;
    float slope;
    int num_mtiles;
    int num_pslots;  // persistent workgroups along M (= rows of the statistics slab)
    int num_ntiles;
    int xcd_order;
    int* overflow;   // optional: set to 1 when an fp16 output was clamped to +-65504
    // inference epilogue fused with MaxPool(2,2) / Reorg(2) (MCAMD_EPI_PAD_F16 only, epi_pool.h): dst_mode != 0 enumerates
    // the M tile in pooled order and `y` is at the pooled resolution; y2 = optional full-resolution copy (POOL)
    int dst_mode;
    void* y2;
    int y2_ld, y2_choff;
    int concurrent;  // mcamd_conv_epilogue.concurrent (dgrad): tiles chosen for CU-time, see pick_tile
    int wrap;        // mcamd_conv_geom.x_wrap (INT_MAX = none): channel blocks >= wrap are read `wrap` channels lower
    int f8_from;     // mcamd_conv_geom.x_f8: first K chunk (of 32 fp16 = 64 e4m3 values) of the fp8 correction part; INT_MAX = none
    int f8_sb;       // e8m0 scale of the B operand of the fp8 MFMAs in all four bytes: 2^-(F8_SXL + x_f8_wexp) (A: 1.0)
};


struct StemArgs {        // conv_stem.hip: forward of the 3-channel first layer
    const half_t* x;     // padded NHWC4 image
    const half_t* w;     // packed stem weights [Npad][96]
    half_t* y;           // raw output [M][y_ld]
    float* stats;        // [grid][2][stats_ld] or NULL
    int y_ld, y_choff, stats_ld;
    int H, W, HW, M;
};

struct WgradArgs {
    const half_t* x;
    const half_t* dy;
    float* slab;
    long long x_img_stride, dy_img_stride;
    int x_row_stride, dy_row_stride, x_ld, dy_ld;
    int x_off;        // channel offset of x
    int dy_off;       // channel offset of dy + offset of padded pixel (1,1)
    int dy_zero_off;  // channel offset of dy at padded pixel (0,0): a zero row
    int H, W, HW, M;
    int rows_pad;     // slab rows
    int ktot, cin_tap, ntaps;
    int tap_off[9];
    int n_ctiles;     // cin tiles per tap
    int n_otiles, n_tapgroups, nsplit;
    int pix_per_split;
};

struct WgradPlan {
    int tmo, tnc, taps, kp, rows_pad, n_otiles, n_ctiles, n_tapgroups, nsplit, pix_per_split;
    int nine;   // 1: padded-pixel 9-tap kernel (wgrad9_kernel), 2: its wide form (wgrad9w_kernel)
    int stemw;  // 1: raw-window first-layer kernel (wgrad_stem_kernel), 2: raw-window 32-channel kernel (wgrad_win_kernel)
    size_t bytes;
};

void mcamd_igemm_tile(long long M, int n, int cin_tap, int ktot, int out[4], bool concurrent = false);   // {BM, BN, BK, kind}: kind 0 igemm_kernel, 2 igemm_pp_kernel
int mcamd_igemm_pp_launch(const IgemmArgs& a, int bm, int bn, int rows, int ntiles, hipStream_t st);
bool mcamd_igemm_f8_ok(long long M, int n, int cin_tap, int ktot);   // the fp8-correction form exists for the tile this shape takes
int mcamd_igemm_rows(long long M, int n, int cin_tap, int ktot, bool raw_epilogue = true, bool concurrent = false);
int mcamd_igemm_launch(IgemmArgs& a, hipStream_t st);

WgradPlan mcamd_wgrad_plan(long long M, int cout, int cin_tap, int ntaps);
int mcamd_wgrad_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st);
bool mcamd_wgrad_stem_ok(int stem, int cout, int W, long long M);   // conv_wgrad_stem.hip
WgradPlan mcamd_wgrad_stem_plan(long long M);
int mcamd_wgrad_stem_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st);
bool mcamd_wgrad_win_ok(int ksize, int stem, int cout, int cin_tap, int W, long long M);
WgradPlan mcamd_wgrad_win_plan(long long M, int cout);
int mcamd_wgrad_win_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st);
bool mcamd_wgrad_use9(int ksize, int stem, int cout, int cin_tap, int W);
WgradPlan mcamd_wgrad_plan9(long long P, int cout, int cin_tap, int W, int pitch, int B);
int mcamd_wgrad9_launch(const WgradArgs& w, const WgradPlan& p, int pitch, long long P, hipStream_t st);
int mcamd_wgrad_finish_launch(const float* slab, const WgradPlan& p, int ktot, int cin_tap, int stem, int Cout, int Cin,
                              int ksize, const float* mask, float inv_scale, float* dw, const int* rmap, const int* cmap,
                              hipStream_t st);
int mcamd_colsum_launch(const half_t* dy, long long rows, int ld, int choff, int C, float inv_scale, float* out,
                        hipStream_t st, void* scratch, size_t scratch_bytes);   // scratch: reusable once the finish pass is enqueued

bool mcamd_win3x3_ok(const IgemmArgs& a);                             // conv_win.hip
bool mcamd_win3x3_shape(long long M, int n, int cin_tap, int ktot, int W);
int mcamd_win3x3_launch(const IgemmArgs& a, hipStream_t st);
bool mcamd_small3x3_ok(long long M, int n, int cin_tap, int ktot);   // conv_small.hip
int mcamd_small3x3_rows(long long M);
int mcamd_small3x3_launch(const IgemmArgs& a, hipStream_t st);
bool mcamd_small3x3_split_ok(long long M, int n, int cin_tap, int ktot, int wrap, int mode);   // split operands, fp32 output
int mcamd_small3x3_split_launch(const IgemmArgs& a, hipStream_t st);

bool mcamd_wres_ok(int ksize, int stem, int n, int cin_tap, int ktot, int B, int H, int W, int mode);   // conv_wres.hip
int mcamd_wres_rows(int n, int B, int H, int W);
int mcamd_wres_launch(IgemmArgs& a, int B, hipStream_t st);

bool mcamd_stem_direct_ok(int stem, int cout, int mode);
int mcamd_stem_rows(long long M);
int mcamd_stem_launch(const StemArgs& a, int cout, hipStream_t st);
