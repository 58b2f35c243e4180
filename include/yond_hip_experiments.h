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

#ifdef __cplusplus
}
#endif
#endif
