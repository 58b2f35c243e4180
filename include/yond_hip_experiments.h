/* yond_hip_experiments.h -- entry points that exist ONLY in experiment builds of libyond_hip.so
 * (hipcc -DYOND_EXPERIMENTS; `python -m yond_public_amd.build --experiments` writes tools/probe/libyond_exp.so).
 * They are not part of the product's C ABI (include/yond_hip.h): measured alternatives kept for A/B runs of tools/. */
#ifndef YOND_HIP_EXPERIMENTS_H
#define YOND_HIP_EXPERIMENTS_H
#include "yond_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* K5' the same maps in ONE pass over the frame(s) -- the B19 map of the self mode never leaves the chip -- together with
 * sweep 1 of the threshold selection (see yond_nle_stats_f32 below: level-1 histogram of lap, per-mean-bin minimum of lap,
 * resolve of the percentile ranks) and the frame maximum (lr.max() for the bias LUT grid, YOND_SIDD.py:256/393; in the
 * workspace head as an order-preserving key).  Replaces yond_box_stats_self1/self2 (or _collab) + yond_nle_stats_f32 on
 * the hot path; continue with yond_nle_threshold_f32(lap, ...) on the same workspace.  k <= 29, k2 <= k. */
int yond_box_stats_self_fused_f32(const float* bayer, int H, int W, int k, int k2, int tile_w, float* mean, float* var,
                                  float* lap, const double* q_host, int nq, void* ws, void* stream);
int yond_box_stats_collab_fused_f32(const float* bayer_lr, const float* bayer_hr, int H, int W, int k, int tile_w,
                                    float* mean, float* var, float* lap, const double* q_host, int nq, void* ws, void* stream);

/* K2f: a whole LEVEL-0 residual block of the guided U-Net (32 -> 32 -> 32 channels; archs/modules.py:186-196, archs/Unet.py:433, 461)
 * in ONE launch:   out = conv2( SiLU( conv1(SiLU(x)) * s1 + t1 ) ) * s2 + t2 + x.   The tensor between the two 3x3 convolutions never
 * leaves the chip (csrc/block0_fused.hip); split-operand arithmetic as yond_conv2d_f32 algo 3 (fp32-accurate products on the fp16 MFMA).
 *   x        block input, in_fmt YOND_FMT_PLANES4 ([N][8][H*W][4]) or YOND_FMT_NHWC_F32 ([N][H][W][32]); also the residual
 *   w1, w2   the two layers' weights as yond_pack_block0_weight_f32 lays them out (36,864 bytes each, device memory)
 *   s1..t2   [N][32] (ebatch != 0) or [32] float32 FiLM / bias vectors (the conv biases folded into t1 / t2 as yond_film_f32 does); NULL: 1 / 0
 *   dst      split planes (YOND_FMT_SPLIT_PLANES, raw values), or NULL with the fused output projection:
 *   out4_*   as in YondConvDesc: out4_dst[N][H][W][4] = (W4 . out + b4 + x4 / ub) * ub  (archs/Unet.py:463-468); the 32-channel tensor is not stored
 *   status   bit 0 set when a staged value leaves fp16's range (|a| > 65504) or is NaN. */
typedef struct YondBlock0Desc {
    const float* x;
    int in_fmt, N, H, W;
    const void *w1, *w2;
    const float *s1, *t1, *s2, *t2;
    int ebatch;
    void* dst;
    const float *out4_w, *out4_b, *out4_x, *out4_ub;
    float* out4_dst;
    int* status;
} YondBlock0Desc;
int yond_block0_fused_f32(const YondBlock0Desc* d /* host */, void* stream);
/* Host-side packer: OIHW float32 [cout <= 32][cin <= 32][3][3] (host memory) -> 36,864 bytes (host memory) in the kernel's LDS order
 * [tap][part h, l][group of 8 input channels][32 output channels] x 8 halves; YOND_EUNSUPPORTED for a weight beyond fp16's range. */
int yond_pack_block0_weight_f32(const float* w_oihw, int cout, int cin, void* out);
/* (round 5, measured no-go: +1.0 % per forward although it moves 1.16 GB less per block -- level 0 is bound by its vector work, not by
 * bytes: profiles/r05_experiments/README.md.  Kept for A/B runs: tools/block0_ab.py, tools/b0_dbg.py, engine.FUSE_BLOCK0.) */

#ifdef __cplusplus
}
#endif
#endif
