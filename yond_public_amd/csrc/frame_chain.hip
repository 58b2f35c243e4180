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

#include "frame_params.h"

__global__ __launch_bounds__(256) void frame_params_kernel(const NleState* __restrict__ st, const float* __restrict__ max_dev,
                                                           int mode, double scale_est, double scale, double tfac, int lut_cap,
                                                           double* __restrict__ prm, float* __restrict__ t_out,
                                                           double* __restrict__ lut_x) {
    __shared__ FrameParams P;
    const int tid = threadIdx.x;
    if (tid == 0) {
        frame_params_compute(P, st, max_dev, mode, scale_est, scale, tfac, lut_cap);
        frame_params_store(P, prm, t_out);
    }
    __syncthreads();
    for (int i = tid; i < P.nk; i += 256) lut_x[i] = frame_knot(P, i);
}

extern "C" int yond_frame_params_f64(const void* nle_ws, const float* max_dev, int mode, double scale_est, double scale, double tfac,
                                     int lut_cap, double* prm, float* t_out, double* lut_x, void* stream) {
    if (!nle_ws || !prm || !lut_x || (mode != 0 && mode != 1) || !(scale > 0.0) || !(scale_est > 0.0) || lut_cap < 2) return YOND_EINVAL;
    hipLaunchKernelGGL(frame_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const NleState*)nle_ws, max_dev, mode, scale_est,
                       scale, tfac, lut_cap, prm, t_out, lut_x);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
