// The per-frame parameter chain ON THE DEVICE (YOND_SIDD.py:341-356, 392-397, 438-454; utils/isp_algos.py:98-108, 345-365):
//     moment sums of the estimator  ->  (beta1, beta2)  ->  (K, sigma)  ->  lower = VST(0), upper = VST(scale), t = 1.03 / (upper - lower)
//     frame maximum                 ->  the knot grid of get_bias
// One small launch writes a parameter block and the knots; the bias-LUT kernel, K1, the FiLM kernel and K4 read them from
// there, so a frame needs NO host round trip between the estimator and the network (the reference does all of this in
// NumPy on the host, and so did this build until round 3: one synchronisation per round and ~0.15 ms of idle GPU per frame).
// The host reads the block when the frame is finished; the rare branches that need more kernels (no flat area: nsel == 0,
// YOND_SIDD.py:79-84) or more table space are FLAGGED here and recomputed on the host path.
// Arithmetic: float64 in the operation order of the NumPy / Python expressions (the library is built with
// -ffp-contract=off), float32 where NumPy 2 computes in float32 (the knot grid of a float32 maximum).
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

__global__ __launch_bounds__(256) void frame_params_kernel(const NleState* __restrict__ st, const float* __restrict__ max_dev,
                                                           int mode, double scale_est, double scale, double tfac, int lut_cap,
                                                           double* __restrict__ prm, float* __restrict__ t_out,
                                                           double* __restrict__ lut_x) {
    __shared__ int s_n, s_n1, s_n2, s_n3;
    __shared__ float s_ub;
    const int tid = threadIdx.x;
    if (tid == 0) {
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
        prm[YOND_PRM_BETA1] = b1; prm[YOND_PRM_BETA2] = b2; prm[YOND_PRM_GAIN] = gain; prm[YOND_PRM_SIGMA] = sigma;
        prm[YOND_PRM_LO] = lo; prm[YOND_PRM_HI] = hi; prm[YOND_PRM_NSR] = nsr; prm[YOND_PRM_T] = (double)tf;
        prm[YOND_PRM_FLAGS] = (double)flags; prm[YOND_PRM_LUT_N] = (double)nk; prm[YOND_PRM_NSEL] = all[0];
        prm[YOND_PRM_TH] = st->sel[1]; prm[YOND_PRM_PCT] = st->sel[2]; prm[YOND_PRM_FRAME_MAX] = (double)mxv;
        if (t_out) *t_out = tf;
        s_n = nk; s_n1 = n1; s_n2 = n2; s_n3 = n3; s_ub = ub;
    }
    __syncthreads();
    const int n1 = s_n1, n2 = s_n2, n3 = s_n3;
    const float ub = s_ub;
    for (int i = tid; i < s_n; i += 256) {
        double v;
        if (ub < 50.0f) v = (double)linspace_f32(i, n1, 0.0f, ub);                   // float32 knots (NumPy 2)
        else if (i < n1) v = linspace_f64(i, n1, 0.0, 50.0);
        else if (ub < 500.0f) v = (double)linspace_f32(i - n1, n2, 50.0f, ub);
        else if (i < n1 + n2) v = linspace_f64(i - n1, n2, 50.0, 500.0);
        else v = (double)linspace_f32(i - n1 - n2, n3, 500.0f, ub);
        lut_x[i] = v;
    }
}

extern "C" int yond_frame_params_f64(const void* nle_ws, const float* max_dev, int mode, double scale_est, double scale, double tfac,
                                     int lut_cap, double* prm, float* t_out, double* lut_x, void* stream) {
    if (!nle_ws || !prm || !lut_x || (mode != 0 && mode != 1) || !(scale > 0.0) || !(scale_est > 0.0) || lut_cap < 2) return YOND_EINVAL;
    hipLaunchKernelGGL(frame_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const NleState*)nle_ws, max_dev, mode, scale_est,
                       scale, tfac, lut_cap, prm, t_out, lut_x);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
