// Row H: construction of the VST bias LUT on the device (utils/isp_algos.py:49-140).
//
// For every knot lam <= th the reference builds the Poisson(lam/K) (*) N(0, sigma/K) density on a grid
// of 1/pho electrons over [-r, r] (scipy.stats pmf/pdf + scipy.signal.convolve(mode='same')),
// renormalises it and integrates the VST against it; above th it uses Foi's closed form.  Here one
// workgroup owns one knot: the Gaussian table and the Poisson masses live in LDS (float64) and every
// thread evaluates grid points of the convolution directly (no FFT).  ~1 GFMA of float64 for a 1000-knot
// LUT: well under a millisecond, against 0.07-0.3 s of SciPy per call in the reference.
//
// The grid is np.linspace(-r, r, l): x_j = fl(fl(j*step) + start), last point = stop.  The emulation is
// exact (separately rounded multiply and add) because scipy's poisson.pmf is zero wherever x_j is not
// exactly an integer -- a rounding artefact of the reference that has to be reproduced, not fixed.
#include "common.h"
#include "frame_params.h"
#include "lut_table.h"

// The frame chain as ONE launch (yond_frame_chain_f64): every workgroup derives the frame's parameters from the estimator's moment
// sums itself (a few hundred scalar float64 operations: cheaper than a launch of their own), integrates its knot, and the workgroup
// that arrives last turns knots and ordinates into K1's prepared table.  st == nullptr: the plain LUT kernel.
struct FrameChain {
    const NleState* st;
    const float* max_dev;
    double scale_est, scale, tfac;
    double* lut_x;
    float* t_out;
    void* lut_ws;
    unsigned int* ticket;
    int mode, lut_cap, smem_bytes;
};

__device__ __forceinline__ double linspace_pt(int j, int l, double start, double stop, double step) {
    if (j == l - 1) return stop;
    return __dadd_rn(__dmul_rn((double)j, step), start);
}

__device__ __forceinline__ double vst_d(double x, double sigma, double gain) {
    // utils/isp_algos.py:7-9 with mu = 0
    double fz = gain * x + 0.375 * gain * gain + sigma * sigma;
    fz = fz > 0.0 ? fz : 0.0;
    return 2.0 / gain * sqrt(fz);
}

__device__ __forceinline__ double block_sum(double v, double* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__global__ __launch_bounds__(256) void bias_lut_kernel(const double* __restrict__ lams, int n, double K, double sigma,
                                                       double th, int pho, int lmax, float* __restrict__ bias,
                                                       double* __restrict__ prm, int smem_doubles, double* __restrict__ gbuf,
                                                       double* __restrict__ bias64, FrameChain fc) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double s_red[4];
    __shared__ FrameParams s_fp;
    const bool chain = fc.st != nullptr;
    if (chain) {
        if (threadIdx.x == 0) {
            frame_params_compute(s_fp, fc.st, fc.max_dev, fc.mode, fc.scale_est, fc.scale, fc.tfac, fc.lut_cap);
            // the derived constants as the host entry computes them; more LDS than this launch has: host path
            K = s_fp.gain; sigma = s_fp.sigma;
            if (!(s_fp.flags & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY))) {
                int ph = (int)sqrt(K);
                if (ph < 1) ph = 1;
                const double t = K < 1.0 ? 50.0 * K : 50.0 * sqrt(K);
                const int rmax = (int)(t * (1.0 / K) * 2.0 + sigma * 2.0 + t + 10.0) + 1;
                if (2 * ph * rmax + 1 + rmax + 2 > smem_doubles) s_fp.flags |= YOND_PRM_FLAG_LUT_CAPACITY;
            }
            if (blockIdx.x == 0) frame_params_store(s_fp, prm, fc.t_out);
        }
        __syncthreads();
        K = s_fp.gain; sigma = s_fp.sigma;
        n = (s_fp.flags & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY)) ? 0 : s_fp.nk;
        if (n > 0) {
            pho = (int)sqrt(K);
            if (pho < 1) pho = 1;
            th = K < 1.0 ? 50.0 * K : 50.0 * sqrt(K);
            const int rmax = (int)(th * (1.0 / K) * 2.0 + sigma * 2.0 + th + 10.0) + 1;
            lmax = 2 * pho * rmax + 1;
        }
        // (the knots: what the last workgroup and the host read -- stored past this XCD's L2, like the ordinates below)
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < s_fp.nk; i += 256) __hip_atomic_store(&fc.lut_x[i], frame_knot(s_fp, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (prm) {
        // (n, K, sigma) from the frame's parameter block (frame_chain.hip); the derived constants as the host entry computes them
        const int fl = (int)prm[YOND_PRM_FLAGS];
        n = (int)prm[YOND_PRM_LUT_N];
        K = prm[YOND_PRM_GAIN];
        sigma = prm[YOND_PRM_SIGMA];
        if ((fl & (YOND_PRM_FLAG_BAD_ESTIMATE | YOND_PRM_FLAG_LUT_CAPACITY)) || (int)blockIdx.x >= n) return;
        pho = (int)sqrt(K);
        if (pho < 1) pho = 1;
        th = K < 1.0 ? 50.0 * K : 50.0 * sqrt(K);
        const int rmax = (int)(th * (1.0 / K) * 2.0 + sigma * 2.0 + th + 10.0) + 1;
        lmax = 2 * pho * rmax + 1;
        if (lmax + rmax + 2 > smem_doubles) {                                     // more LDS than this launch has: host path
            if (blockIdx.x == 0 && threadIdx.x == 0) prm[YOND_PRM_FLAGS] = (double)(fl | YOND_PRM_FLAG_LUT_CAPACITY);
            return;
        }
    }
    // Gaussian table on the grid, l entries: in LDS, or -- large K * sigma (14-bit sensors at a digital gain of 10 ...): hundreds of
    // KB -- in a per-workgroup slice of a global scratch buffer (yond_bias_lut_big_f64: the workgroups then stride over the knots)
    double* s_g = gbuf ? gbuf + (size_t)blockIdx.x * lmax : sm;
    double* s_p = gbuf ? sm : sm + lmax;   // Poisson mass at integer point m (x = m), r+1 entries; 0 if the grid misses it
    const int tid = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
    __syncthreads();                       // (the tables of the previous knot are no longer read)
    const double lam = chain ? frame_knot(s_fp, i) : lams[i];
    if (lam > th) {
        if (tid == 0) {
            // close_form_bias (utils/isp_algos.py:84-96)
            const double y = lam / K, sg = sigma / K;
            const double yh = y + 0.375 + sg * sg;
            const double yh2 = yh * yh;
            const double m1 = (y + sg * sg) / yh2;
            const double m2 = y / (yh2 * yh);
            const double m3 = (y + 3.0 * (y + sg * sg) * (y + sg * sg)) / (yh2 * yh2);
            const double cf = 2.0 * sqrt(yh) * (-1.0 / 8.0 * m1 + 1.0 / 16.0 * m2 - 5.0 / 128.0 * m3);
            if (bias64) bias64[i] = cf;
            else if (chain) __hip_atomic_store(&bias[i], (float)cf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else bias[i] = (float)cf;
        }
        continue;
    }
    // getGsP (utils/isp_algos.py:49-82), r as at :124
    const int r = (int)(lam * (1.0 / K) * 2.0 + sigma * 2.0 + lam + 10.0);
    const int l = 2 * pho * r + 1;
    const double start = -(double)r, stop = (double)r;
    const double step = (stop - start) / (double)(l - 1);
    const double mu = lam / K;
    const double sg = sigma / K;
    const int c = pho * r;            // index of x = 0
    // Poisson masses at the integer abscissae x = m (index c + pho*m), m = 0..r
    for (int m = tid; m <= r; m += 256) {
        const double x = linspace_pt(c + pho * m, l, start, stop, step);
        double pm = 0.0;
        if (x >= 0.0 && floor(x) == x) {
            // scipy: exp(xlogy(k, mu) - gammaln(k + 1) - mu)
            const double xl = (x == 0.0) ? 0.0 : x * log(mu);      // mu == 0: log -> -inf -> pmf 0 for k > 0
            pm = exp(xl - lgamma(x + 1.0) - mu);
        }
        s_p[m] = pm;
    }
    if (sigma > 0.0) {
        const double inv_c = 1.0 / sqrt(2.0 * M_PI);
        for (int j = tid; j < l; j += 256) {
            const double y = linspace_pt(j, l, start, stop, step) / sg;
            s_g[j] = exp(-(y * y) / 2.0) * inv_c / sg;               // norm.pdf(x, 0, sg)
        }
    }
    __syncthreads();
    // conv[j] = sum_m P[m] * G[j - pho*m]   ('same' window of the full convolution), then the two sums
    double acc_p = 0.0, acc_pv = 0.0;
    for (int j = tid; j < l; j += 256) {
        double cv;
        if (sigma > 0.0) {
            cv = 0.0;
            int mlo = (j - (l - 1) + pho - 1) / pho;                // need 0 <= j - pho*m <= l-1
            if (j - (l - 1) <= 0) mlo = 0;
            int mhi = j / pho;
            if (mhi > r) mhi = r;
            // eight terms per trip into four partial sums: a rolled loop exposes one LDS round trip and one dependent
            // float64 FMA per term (up to 174 of them per grid point, three grid points per thread)
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
            int m = mlo;
            for (; m + 7 <= mhi; m += 8) {
                const double p0 = s_p[m], p1 = s_p[m + 1], p2 = s_p[m + 2], p3 = s_p[m + 3];
                const double p4 = s_p[m + 4], p5 = s_p[m + 5], p6 = s_p[m + 6], p7 = s_p[m + 7];
                const double g0 = s_g[j - pho * m], g1 = s_g[j - pho * (m + 1)], g2 = s_g[j - pho * (m + 2)], g3 = s_g[j - pho * (m + 3)];
                const double g4 = s_g[j - pho * (m + 4)], g5 = s_g[j - pho * (m + 5)], g6 = s_g[j - pho * (m + 6)], g7 = s_g[j - pho * (m + 7)];
                c0 = fma(p0, g0, c0); c1 = fma(p1, g1, c1); c2 = fma(p2, g2, c2); c3 = fma(p3, g3, c3);
                c0 = fma(p4, g4, c0); c1 = fma(p5, g5, c1); c2 = fma(p6, g6, c2); c3 = fma(p7, g7, c3);
            }
            for (; m <= mhi; ++m) c0 = fma(s_p[m], s_g[j - pho * m], c0);
            cv = (c0 + c1) + (c2 + c3);
        } else {
            const int jm = j - c;
            cv = (jm >= 0 && jm % pho == 0) ? s_p[jm / pho] : 0.0;
        }
        if (cv < 0.0) cv = 0.0;
        const double x = linspace_pt(j, l, start, stop, step);
        acc_p += cv;
        acc_pv += cv * vst_d(K * x, sigma, K);
    }
    const double sp = block_sum(acc_p, s_red);
    const double spv = block_sum(acc_pv, s_red);
    if (tid == 0) {
        // p = conv / (sum/pho);  bias = sum(p * V / pho) - VST(lam)
        const double e = spv / (sp / (double)pho) / (double)pho;
        if (bias64) bias64[i] = e - vst_d(lam, sigma, K);
        else if (chain) __hip_atomic_store(&bias[i], (float)(e - vst_d(lam, sigma, K)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else bias[i] = (float)(e - vst_d(lam, sigma, K));
    }
    }
    if (!chain) return;
    // ---- the workgroup that arrives last: K1's prepared table (yond_lut_table_f64's kernel, vst.hip) ----
    if (!nf_arrive_last(fc.ticket, gridDim.x)) return;
    if (tid == 0) *fc.ticket = 0;                                  // (the workspace can take the next call)
    __shared__ LutLds L;
    LutHeader* hd = (LutHeader*)fc.lut_ws;
    if (n < 2 || (size_t)n * LUT_BYTES_PER_KNOT > (size_t)fc.smem_bytes) {
        if (tid == 0) {
            hd->n = 0; hd->nseg = 0; hd->nbreak = 0;
            if (n >= 2) prm[YOND_PRM_FLAGS] = (double)(s_fp.flags | YOND_PRM_FLAG_LUT_CAPACITY);
        }
        return;
    }
    if (tid == 0) {
        L.ab = (double2*)sm;
        L.x = (double*)(L.ab + n);
    }
    __syncthreads();
    lut_prepare(L, fc.lut_x, bias, n, 0);
    lut_table_store(L, fc.lut_ws, n);
    // K1 keeps only the coefficients in LDS and finds the interval from the run table: a grid that is not a few evenly spaced
    // runs (never the case for these grids) goes back to the host path
    if (tid == 0 && L.nseg == 0) prm[YOND_PRM_FLAGS] = (double)(s_fp.flags | YOND_PRM_FLAG_LUT_CAPACITY);
}

#define BIAS_LDS_MAX (160 * 1024 - 512)      // dynamic LDS the kernel may ask for (the CU has 160 KB; ~300 bytes are static)
static int bias_lut_attr() {                 // the kernel may use the whole LDS of a CU (set once for both entry points)
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)bias_lut_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BIAS_LDS_MAX);
        if (e != hipSuccess) return (int)e;
        done = true;
    }
    return 0;
}

extern "C" int yond_bias_lut_f64(const double* lams, int n, double gain, double sigma, float* bias, void* stream) {
    if (!lams || !bias || n <= 0 || !(gain > 0.0) || !(sigma >= 0.0)) return YOND_EINVAL;
    const double K = gain;
    int pho = (int)sqrt(K);
    if (pho < 1) pho = 1;
    const double th = K < 1.0 ? 50.0 * K : 50.0 * sqrt(K);
    // largest numerically integrated knot is <= th
    const int rmax = (int)(th * (1.0 / K) * 2.0 + sigma * 2.0 + th + 10.0) + 1;
    const int lmax = 2 * pho * rmax + 1;
    const size_t smem = ((size_t)lmax + rmax + 2) * sizeof(double);
    if (smem > BIAS_LDS_MAX) return YOND_EUNSUPPORTED;
    if (int e = bias_lut_attr()) return e;
    hipLaunchKernelGGL(bias_lut_kernel, dim3(n), dim3(256), smem, (hipStream_t)stream, lams, n, K, sigma, th, pho, lmax, bias,
                       (double*)nullptr, 0, (double*)nullptr, (double*)nullptr, FrameChain{});
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// (n, K, sigma) from the frame's parameter block: the grid is the knot CAPACITY, workgroups beyond the frame's knots exit;
// the LDS allotment is fixed at 48 KB (the integration tables of K ~ 0.5 ... 40, sigma ~ 0 ... 40 DN need 4-35 KB; a frame that
// needs more is flagged for the host path)
#define BIAS_DEV_SMEM_DOUBLES 6144
extern "C" int yond_bias_lut_dev_f64(const double* lams, int lut_cap, double* prm, float* bias, void* stream) {
    if (!lams || !prm || !bias || lut_cap < 2 || lut_cap > 4096) return YOND_EINVAL;
    const size_t smem = (size_t)BIAS_DEV_SMEM_DOUBLES * sizeof(double);
    if (int e = bias_lut_attr()) return e;
    hipLaunchKernelGGL(bias_lut_kernel, dim3(lut_cap), dim3(256), smem, (hipStream_t)stream, lams, 0, 1.0, 0.0, 0.0, 1, 0, bias, prm,
                       BIAS_DEV_SMEM_DOUBLES, (double*)nullptr, (double*)nullptr, FrameChain{});
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// The whole frame chain in one launch: parameters (yond_frame_params_f64) + knots + bias LUT (yond_bias_lut_dev_f64) + prepared
// table (yond_lut_table_f64); same results, same flags.  nle_ws: the estimator's workspace after yond_nle_moments_f32 (its ticket
// word 2 counts the arrivals and is left at zero).
extern "C" int yond_frame_chain_f64(void* nle_ws, const float* max_dev, int mode, double scale_est, double scale, double tfac, int lut_cap,
                                    double* prm, float* t_out, double* lut_x, float* lut_y, void* lut_ws, void* stream) {
    if (!nle_ws || !prm || !lut_x || !lut_y || !lut_ws || (mode != 0 && mode != 1) || !(scale > 0.0) || !(scale_est > 0.0)) return YOND_EINVAL;
    if (lut_cap < 2 || lut_cap > LUT_MAX) return YOND_EINVAL;
    size_t smem = (size_t)BIAS_DEV_SMEM_DOUBLES * sizeof(double);
    if (smem < (size_t)lut_cap * LUT_BYTES_PER_KNOT) smem = (size_t)lut_cap * LUT_BYTES_PER_KNOT;
    if (int e = bias_lut_attr()) return e;
    FrameChain fc;
    fc.st = (const NleState*)nle_ws; fc.max_dev = max_dev; fc.scale_est = scale_est; fc.scale = scale; fc.tfac = tfac;
    fc.lut_x = lut_x; fc.t_out = t_out; fc.lut_ws = lut_ws; fc.ticket = &((NleState*)nle_ws)->ticket[2];
    fc.mode = mode; fc.lut_cap = lut_cap; fc.smem_bytes = (int)smem;
    hipLaunchKernelGGL(bias_lut_kernel, dim3(lut_cap), dim3(256), smem, (hipStream_t)stream, (const double*)nullptr, 0, 1.0, 0.0, 0.0, 1, 0,
                       lut_y, prm, BIAS_DEV_SMEM_DOUBLES, (double*)nullptr, (double*)nullptr, fc);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// Large K * sigma: the Gaussian table does not fit the LDS (yond_bias_lut_f64 returns YOND_EUNSUPPORTED): the same
// integration with the table in a global scratch buffer, `nwg` workgroups striding over the knots.
// yond_bias_lut_big_scratch: doubles of scratch for (gain, sigma) at nwg workgroups (0: parameters out of range).
// get_bias_points (utils/isp_algos.py:142-160): the same integration for ARBITRARY abscissae, with the sampling rate
// pho = max(int(sqrt K), pho_min) (the reference's pointwise calls use pho_min = 100) and the closed form above th only when
// close_form is set (else every point is integrated: th = lam_max + 1).  Gaussian table in the global scratch buffer.
static void bias_points_consts(double gain, double sigma, int pho_min, int close_form, double lam_max, int& pho, double& th, int& rmax) {
    pho = (int)sqrt(gain);
    if (pho < pho_min) pho = pho_min;
    if (pho < 1) pho = 1;
    th = close_form ? (gain < 1.0 ? 50.0 * gain : 50.0 * sqrt(gain)) : lam_max + 1.0;
    const double top = close_form ? th : lam_max;                    // largest numerically integrated abscissa
    rmax = (int)(top * (1.0 / gain) * 2.0 + sigma * 2.0 + top + 10.0) + 1;
}

extern "C" size_t yond_bias_points_scratch(double gain, double sigma, int pho_min, int close_form, double lam_max, int nwg) {
    if (!(gain > 0.0) || !(sigma >= 0.0) || nwg <= 0 || pho_min < 1) return 0;
    int pho, rmax;
    double th;
    bias_points_consts(gain, sigma, pho_min, close_form, lam_max, pho, th, rmax);
    if (rmax > 15000 || (double)pho * rmax > 2.5e7) return 0;        // (LDS: the Poisson masses; scratch: 16 bytes x pho x r per workgroup)
    return (size_t)nwg * (size_t)(2 * pho * rmax + 1);
}

extern "C" int yond_bias_points_f64(const double* lams, int n, double gain, double sigma, int pho_min, int close_form, double lam_max,
                                    float* bias32, double* bias64, double* scratch, size_t scratch_doubles, int nwg, void* stream) {
    if (!lams || (!bias32 && !bias64) || !scratch || n <= 0 || nwg <= 0) return YOND_EINVAL;
    const size_t need = yond_bias_points_scratch(gain, sigma, pho_min, close_form, lam_max, nwg);
    if (need == 0) return YOND_EUNSUPPORTED;
    if (scratch_doubles < need) return YOND_EINVAL;
    int pho, rmax;
    double th;
    bias_points_consts(gain, sigma, pho_min, close_form, lam_max, pho, th, rmax);
    const int lmax = 2 * pho * rmax + 1;
    if (int e = bias_lut_attr()) return e;
    hipLaunchKernelGGL(bias_lut_kernel, dim3(nwg < n ? nwg : n), dim3(256), ((size_t)rmax + 2) * sizeof(double), (hipStream_t)stream, lams, n, gain,
                       sigma, th, pho, lmax, bias32, (double*)nullptr, 0, scratch, bias64, FrameChain{});
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

extern "C" size_t yond_bias_lut_big_scratch(double gain, double sigma, int nwg) {
    return yond_bias_points_scratch(gain, sigma, 1, 1, 0.0, nwg);
}

extern "C" int yond_bias_lut_big_f64(const double* lams, int n, double gain, double sigma, float* bias, double* scratch,
                                     size_t scratch_doubles, int nwg, void* stream) {
    return yond_bias_points_f64(lams, n, gain, sigma, 1, 1, 0.0, bias, nullptr, scratch, scratch_doubles, nwg, stream);
}
