// The per-frame parameters from the estimator's moment sums (frame_chain.hip's kernel and the one-launch chain of bias_lut.hip).
#pragma once
#include "common.h"
#include "nle_common.h"

// parameter block (doubles): see YOND_PRM_* in include/yond_hip.h
__device__ __forceinline__ double vst_scalar_d(double x, double sigma, double gain) {
    // utils/isp_algos.py:7-9 on scalars: gain*x + (3/8) gain**2 + sigma**2, max(., 0), 2/gain * sqrt
    double fz = gain * x + 0.375 * (gain * gain) + sigma * sigma;
    fz = fz > 0.0 ? fz : 0.0;
    return 2.0 / gain * sqrt(fz);
}

// np.linspace(start, stop, num) in float32 (NumPy 2: a float32 stop with Python-scalar start / num computes in float32):
// step = (stop - start) / (num - 1);  y[i] = i * step + start (two roundings);  y[num - 1] = stop
__device__ __forceinline__ float linspace_f32(int i, int num, float start, float stop) {
    if (i == num - 1) return stop;
    const float step = __fdiv_rn(__fsub_rn(stop, start), (float)(num - 1));
    return __fadd_rn(__fmul_rn((float)i, step), start);
}
// ... and in float64 (Python-int start and stop)
__device__ __forceinline__ double linspace_f64(int i, int num, double start, double stop) {
    if (i == num - 1) return stop;
    const double step = (stop - start) / (double)(num - 1);
    return __dadd_rn(__dmul_rn((double)i, step), start);
}

struct FrameParams {
    double b1, b2, gain, sigma, lo, hi, nsr, nsel, th, pct;
    float tf, ub, mxv;
    int flags, nk, n1, n2, n3;
};

// one thread: (beta1, beta2) -> (K, sigma) -> lower / upper / t, and the shape of get_bias' knot grid
__device__ inline void frame_params_compute(FrameParams& P, const NleState* __restrict__ st, const float* __restrict__ max_dev, int mode,
                                            double scale_est, double scale, double tfac, int lut_cap) {
    int flags = 0;
    const double* all = st->mom;
    const double* ns = st->mom + 5;
    if (!(all[0] > 0.0)) flags |= YOND_PRM_FLAG_NO_FLAT_AREA;                  // YOND_SIDD.py:79-84: the host path takes over
    const double* use = ns[0] > 0.01 * all[0] ? ns : all;                       // utils/isp_algos.py:348-350
    const double n = use[0], sx = use[1], sy = use[2], sxx = use[3], sxy = use[4];
    const double det = n * sxx - sx * sx;
    double b1, b2;
    const double lim = n * sxx > 1e-300 ? n * sxx : 1e-300;
    if (n < 2.0 || det <= 1e-12 * lim) {                                        // rank-deficient: minimum-norm solution (pipeline._fit_from_moments)
        if (n <= 0.0) { b1 = 0.0; b2 = 0.0; }
        else {
            const double mbar = sx / n, vbar = sy / n;
            b1 = vbar * mbar / (mbar * mbar + 1.0);
            b2 = vbar / (mbar * mbar + 1.0);
        }
    } else {
        b1 = (n * sxy - sx * sy) / det;
        b2 = (sxx * sy - sx * sxy) / det;
    }
    double gain, sigma;
    if (mode == 0) {                                                             // :356
        gain = b1 * scale_est;
        sigma = sqrt(b2 > 0.0 ? b2 : 0.0) * scale_est;
    } else {                                                                     // :438-447
        if (b2 < 0.0) b2 = b1 * b1;
        gain = b1 * scale_est;
        sigma = sqrt(b2) * scale_est;
        if (b1 < 0.0) flags |= YOND_PRM_FLAG_ROUND_ABORTED;
    }
    if (!(gain > 0.0) || !(sigma >= 0.0)) flags |= YOND_PRM_FLAG_BAD_ESTIMATE;  // (K1 / the LUT need K > 0: the consumers skip their work)
    const double lo = vst_scalar_d(0.0, sigma, gain), hi = vst_scalar_d(scale, sigma, gain);      // :263-264
    const double nsr = 1.0 / (hi - lo);
    const float tf = (float)(nsr * tfac);                                        // :284-285
    // knot grid of get_bias (utils/isp_algos.py:101-108) for ub = ceil(float32(max) * float32(scale)) + 1 (float32)
    const float mxv = max_dev ? *max_dev : key2f(st->frame_max_key);
    const float mx = __fmul_rn(mxv, (float)scale);
    const float ub = __fadd_rn(ceilf(mx), 1.0f);
    int n1, n2 = 0, n3 = 0;
    if (ub < 50.0f) n1 = (int)__fdiv_rn(ub, 0.1f) + 2;
    else if (ub < 500.0f) { n1 = 501; n2 = (int)__fsub_rn(ub, 50.0f) + 2; }
    else { n1 = 501; n2 = 451; n3 = (int)__fsub_rn(ub, 500.0f) / 10 + 2; }
    int nk = n1 + n2 + n3;
    if (nk > lut_cap || !(ub >= 0.0f)) { flags |= YOND_PRM_FLAG_LUT_CAPACITY; nk = 0; n1 = n2 = n3 = 0; }
    P.b1 = b1; P.b2 = b2; P.gain = gain; P.sigma = sigma; P.lo = lo; P.hi = hi; P.nsr = nsr; P.tf = tf;
    P.flags = flags; P.nk = nk; P.n1 = n1; P.n2 = n2; P.n3 = n3; P.ub = ub; P.mxv = mxv;
    P.nsel = all[0]; P.th = st->sel[1]; P.pct = st->sel[2];
}

__device__ inline void frame_params_store(const FrameParams& P, double* __restrict__ prm, float* __restrict__ t_out) {
    prm[YOND_PRM_BETA1] = P.b1; prm[YOND_PRM_BETA2] = P.b2; prm[YOND_PRM_GAIN] = P.gain; prm[YOND_PRM_SIGMA] = P.sigma;
    prm[YOND_PRM_LO] = P.lo; prm[YOND_PRM_HI] = P.hi; prm[YOND_PRM_NSR] = P.nsr; prm[YOND_PRM_T] = (double)P.tf;
    prm[YOND_PRM_FLAGS] = (double)P.flags; prm[YOND_PRM_LUT_N] = (double)P.nk; prm[YOND_PRM_NSEL] = P.nsel;
    prm[YOND_PRM_TH] = P.th; prm[YOND_PRM_PCT] = P.pct; prm[YOND_PRM_FRAME_MAX] = (double)P.mxv;
    if (t_out) *t_out = P.tf;
}

// knot i of the grid (float32 runs where NumPy 2 computes in float32)
__device__ __forceinline__ double frame_knot(const FrameParams& P, int i) {
    const int n1 = P.n1, n2 = P.n2, n3 = P.n3;
    const float ub = P.ub;
    if (ub < 50.0f) return (double)linspace_f32(i, n1, 0.0f, ub);                // float32 knots (NumPy 2)
    if (i < n1) return linspace_f64(i, n1, 0.0, 50.0);
    if (ub < 500.0f) return (double)linspace_f32(i - n1, n2, 50.0f, ub);
    if (i < n1 + n2) return linspace_f64(i - n1, n2, 50.0, 500.0);
    return (double)linspace_f32(i - n1 - n2, n3, 500.0f, ub);
}
