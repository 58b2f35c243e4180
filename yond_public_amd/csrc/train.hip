// N4, first slice: the kernels a training step of the AWGN denoiser needs beyond the forward pass
// (trainer_AWGN.py:101-117: pred = net(lr, sigma); loss = F.l1_loss(pred, hr); loss.backward(); Adam.step()).
//   wgrad   dW[tap][co][ci] = sum over pixels of dY[p_out][co] * X[p_in(p_out, tap)][ci] for a 3x3 (stride 1 / 2, pad 1), a
//           2x2 stride-2 transposed or a 1x1 convolution, NHWC float32, on the fp32 matrix cores (v_mfma_f32_32x32x2_f32:
//           one MFMA = the outer product of two pixels' channel vectors; exact fp32 products, fp32 accumulation)
//   colsum  db[c] = sum over pixels of dY[p][c]
//   l1      loss = mean |pred - target|, dpred = sign(pred - target) / n            (losses/base_loss.py:81-113: F.l1_loss)
//   adam    torch.optim.Adam's update (defaults betas 0.9 / 0.999, eps 1e-8, no weight decay, no amsgrad)
// The data gradients (dgrad) need no kernels of their own: they are convolutions with re-indexed weights and run on the
// forward kernels (yond_public_amd/train.py).  Correctness first: operands stream from global memory / L2 straight into
// the MFMA (one 4-byte load per operand and lane), partial sums are combined with float atomics.
#include "common.h"

struct WgradGeom {
    int N, H, W, Cin;          // X: [N][H][W][Cin]
    int Ho, Wo, Cout;          // dY: [N][Ho][Wo][Cout]
    int mode;                  // 0: 3x3 pad 1 (stride s), 1: 2x2 stride-2 transposed (Ho = 2H, Wo = 2W), 2: 1x1
    int stride, taps;
    int chunk;                 // GEMM-K pixels per wave
};

// One wave per (tap, 32-wide co tile, 32-wide ci tile, pixel chunk).  MFMA 32x32x2: lane l supplies A[row l&31][k = l>>5]
// and B[k = l>>5][col l&31]: A = dY (row = output channel), B = X (column = input channel), k = two consecutive pixels of
// the chunk -- lanes 0..31 read 32 consecutive channels of one pixel, lanes 32..63 of the next (two 128-byte segments).
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, WgradGeom g,
                                                    float* __restrict__ dw /* [taps][Cout][Cin], zeroed by the caller */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int nci = g.Cin / 32, nco = g.Cout / 32;
    int b = blockIdx.x;
    const int cit = b % nci; b /= nci;
    const int cot = b % nco; b /= nco;
    const int tap = b;
    // GEMM-K index space: for mode 1 the pixels of X (each pairs with ONE dY pixel per tap), else the pixels of dY
    const long long npix = g.mode == 1 ? (long long)g.N * g.H * g.W : (long long)g.N * g.Ho * g.Wo;
    const long long p0 = ((long long)blockIdx.y * 4 + wave) * g.chunk;
    if (p0 >= npix) return;
    const long long p1 = p0 + g.chunk < npix ? p0 + g.chunk : npix;
    const int ky = g.mode == 0 ? tap / 3 : tap / 2, kx = g.mode == 0 ? tap % 3 : tap % 2;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (long long q = p0; q < p1; q += 2) {            // (wave-uniform trip count: the MFMA needs every lane)
        const long long p = q + lk;
        float a = 0.0f, bv = 0.0f;
        if (p < p1) {
            if (g.mode == 1) {
                const int xq = (int)(p % g.W);
                const long long t = p / g.W;
                const int yq = (int)(t % g.H), n = (int)(t / g.H);
                bv = x[p * g.Cin + cit * 32 + li];
                a = dy[(((long long)n * g.Ho + 2 * yq + ky) * g.Wo + 2 * xq + kx) * g.Cout + cot * 32 + li];
            } else {
                const int xo = (int)(p % g.Wo);
                const long long t = p / g.Wo;
                const int yo = (int)(t % g.Ho), n = (int)(t / g.Ho);
                a = dy[p * g.Cout + cot * 32 + li];
                const int yi = g.mode == 0 ? yo * g.stride + ky - 1 : yo, xi = g.mode == 0 ? xo * g.stride + kx - 1 : xo;
                if (yi >= 0 && yi < g.H && xi >= 0 && xi < g.W) bv = x[(((long long)n * g.H + yi) * g.W + xi) * g.Cin + cit * 32 + li];
            }
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
    }
    // D rows (output channel) (r&3) + 8 (r>>2) + 4 lk, column (input channel) li
    float* out = dw + ((size_t)tap * g.Cout + cot * 32) * g.Cin + cit * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(out + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lk) * g.Cin, acc[r]);
}

extern "C" int yond_conv_wgrad_f32(const float* x, const float* dy, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int mode,
                                   int stride, float* dw, void* stream) {
    if (!x || !dy || !dw || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return YOND_EINVAL;
    if (mode < 0 || mode > 2 || (mode == 0 && stride != 1 && stride != 2)) return YOND_EINVAL;
    if (mode == 0 && (Ho != (H + stride - 1) / stride || Wo != (W + stride - 1) / stride)) return YOND_EINVAL;
    if (mode == 1 && (Ho != 2 * H || Wo != 2 * W)) return YOND_EINVAL;
    if (mode == 2 && (Ho != H || Wo != W)) return YOND_EINVAL;
    WgradGeom g{N, H, W, Cin, Ho, Wo, Cout, mode, stride, mode == 0 ? 9 : (mode == 1 ? 4 : 1), 0};
    const long long npix = mode == 1 ? (long long)N * H * W : (long long)N * Ho * Wo;
    // ~1024 waves over the pixel axis at most; chunks of an even number of pixels
    long long chunk = (npix + 1023) / 1024;
    if (chunk < 64) chunk = 64;
    chunk += chunk & 1;
    g.chunk = (int)chunk;
    const long long waves = (npix + chunk - 1) / chunk;
    const unsigned gy = (unsigned)((waves + 3) / 4);
    const size_t bytes = (size_t)g.taps * Cout * Cin * sizeof(float);
    hipError_t e = hipMemsetAsync(dw, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(g.taps * (Cout / 32) * (Cin / 32)), gy), dim3(256), 0, (hipStream_t)stream, x, dy, g, dw);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// db[c] = sum over pixels of dY[p][c]  (C a multiple of 32; float64 partial sums per workgroup, one float atomic each)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, long long npix, int C, float* __restrict__ db) {
    __shared__ double s[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), row = threadIdx.x >> 5;
    double acc = 0.0;
    for (long long p = (long long)blockIdx.y * 8 + row; p < npix; p += (long long)gridDim.y * 8) acc += (double)dy[p * C + c];
    s[row][threadIdx.x & 31] = acc;
    __syncthreads();
    if (row == 0) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += s[r][threadIdx.x & 31];
        atomicAdd(db + c, (float)t);
    }
}

extern "C" int yond_colsum_f32(const float* dy, size_t npix, int C, float* db, void* stream) {
    if (!dy || !db || npix == 0 || C <= 0 || C % 32) return YOND_EINVAL;
    hipError_t e = hipMemsetAsync(db, 0, (size_t)C * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t gy = (npix + 8 * 64 - 1) / (8 * 64);
    if (gy > 256) gy = 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(C / 32, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dy, (long long)npix, C, db);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// F.l1_loss (mean reduction) and its gradient: loss_sum += sum |pred - target| (float64), grad = sign(pred - target) * gscale
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n, float gscale,
                                                 double* __restrict__ loss_sum, float* __restrict__ grad) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        acc += (double)fabsf(d);
        if (grad) grad[i] = d > 0.0f ? gscale : (d < 0.0f ? -gscale : 0.0f);          // torch: sign(0) = 0
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

extern "C" int yond_l1_loss_f32(const float* pred, const float* target, size_t n, double* loss_sum, float* grad, void* stream) {
    if (!pred || !target || !loss_sum || n == 0) return YOND_EINVAL;
    hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, n, 1.0f / (float)n, loss_sum, grad);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// L1_Charbonnier_loss (losses/base_loss.py:69-79): error = sqrt(diff * diff + eps), loss = mean(error); the gradient in the
// float32 steps of torch's backward: grad_u = (1 / n) / (2 error) through the sqrt, then grad_u * diff from EACH operand of
// diff * diff, summed
__global__ __launch_bounds__(256) void charbonnier_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n, float eps,
                                                          float gscale, double* __restrict__ loss_sum, float* __restrict__ grad) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = __fsub_rn(pred[i], target[i]);
        const float e = __fsqrt_rn(__fadd_rn(__fmul_rn(d, d), eps));
        acc += (double)e;
        if (grad) {
            const float gu = __fdiv_rn(gscale, __fmul_rn(2.0f, e));
            const float t = __fmul_rn(gu, d);
            grad[i] = __fadd_rn(t, t);
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

extern "C" int yond_charbonnier_loss_f32(const float* pred, const float* target, size_t n, double eps, double* loss_sum, float* grad,
                                         void* stream) {
    if (!pred || !target || !loss_sum || n == 0 || !(eps > 0.0)) return YOND_EINVAL;
    hipError_t e = hipMemsetAsync(loss_sum, 0, sizeof(double), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(charbonnier_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, target, n, (float)eps,
                       1.0f / (float)n, loss_sum, grad);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}

// torch.optim.Adam.step (single tensor, no weight decay, no amsgrad), float32 state as torch keeps it:
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float b1, float b2, float step_size, float inv_bc2_sqrt,
                                                   float eps) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        // torch: exp_avg.lerp_(grad, 1 - beta1) = m + (1 - b1) (g - m); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float mi = __fadd_rn(m[i], __fmul_rn(1.0f - b1, __fsub_rn(gi, m[i])));
        const float vi = __fadd_rn(__fmul_rn(v[i], b2), __fmul_rn(__fmul_rn(gi, gi), 1.0f - b2));
        m[i] = mi;
        v[i] = vi;
        const float denom = __fadd_rn(__fmul_rn(__fsqrt_rn(vi), inv_bc2_sqrt), eps);
        p[i] = __fsub_rn(p[i], __fmul_rn(step_size, __fdiv_rn(mi, denom)));
    }
}

extern "C" int yond_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps,
                                  int step, void* stream) {
    if (!p || !g || !m || !v || n == 0 || step < 1) return YOND_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    size_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)beta1, (float)beta2,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)eps);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
