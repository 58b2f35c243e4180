// N1: per-block PSNR / SSIM partial sums on the device (YOND_SIDD.py:649-652, 679-697).
// SSIM as the reference computes it: images scaled to [0,255] in float64, 11x11 Gaussian window
// (sigma 1.5, cv2.getGaussianKernel), 'valid' part only, C1=(0.01*255)^2, C2=(0.03*255)^2.  The window
// is separable, so each 32x32 tile of the valid map needs a 42x42 input patch: horizontal pass for the
// five moments into LDS, vertical pass + SSIM formula + block reduction.  Output: one (sum of squared
// error, sum of SSIM) pair per tile; the host adds the tiles of a block in a fixed order (deterministic).
#include "common.h"

#define MT 32
#define MH (MT + 10)

__global__ __launch_bounds__(256) void block_metrics_kernel(const float* __restrict__ dn, const float* __restrict__ hr,
                                                            int W, int bh, int bw, int nbx, int ntx,
                                                            double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* s_a = sm;                      // [MH][MH]
    double* s_b = s_a + MH * MH;
    double* s_q = s_b + MH * MH;           // [5][MH][MT]
    __shared__ double s_g[11];
    __shared__ double s_red[8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x, tile = blockIdx.y;
    const int by = blk / nbx, bx = blk % nbx;
    const int ty = tile / ntx, tx = tile % ntx;
    const int y0 = ty * MT, x0 = tx * MT;              // tile origin inside the block
    if (tid < 11) {
        double gs = 0.0;
        for (int i = 0; i < 11; ++i) gs += exp(-((i - 5.0) * (i - 5.0)) / (2.0 * 1.5 * 1.5));
        s_g[tid] = exp(-((tid - 5.0) * (tid - 5.0)) / (2.0 * 1.5 * 1.5)) / gs;
    }
    double se = 0.0;
    for (int it = tid; it < MH * MH; it += 256) {
        const int r = it / MH, c = it % MH;
        const int y = y0 + r, x = x0 + c;
        double a = 0.0, b = 0.0;
        if (y < bh && x < bw) {
            const size_t idx = (size_t)(by * bh + y) * W + (size_t)bx * bw + x;
            const float fa = dn[idx], fb = hr[idx];
            a = (double)__fmul_rn(fa, 255.0f);         // dn*255 is a float32 product (YOND_SIDD.py:652), then float64
            b = (double)__fmul_rn(fb, 255.0f);
            if (r < MT && c < MT) { const double d = (double)fa - (double)fb; se += d * d; }
        }
        s_a[it] = a;
        s_b[it] = b;
    }
    __syncthreads();
    for (int it = tid; it < MH * MT; it += 256) {
        const int r = it / MT, c = it % MT;
        double m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
        for (int d = 0; d < 11; ++d) {
            const double g = s_g[d], a = s_a[r * MH + c + d], b = s_b[r * MH + c + d];
            m1 += g * a; m2 += g * b; e11 += g * (a * a); e22 += g * (b * b); e12 += g * (a * b);
        }
        s_q[0 * MH * MT + it] = m1; s_q[1 * MH * MT + it] = m2; s_q[2 * MH * MT + it] = e11;
        s_q[3 * MH * MT + it] = e22; s_q[4 * MH * MT + it] = e12;
    }
    __syncthreads();
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double ss = 0.0;
    for (int it = tid; it < MT * MT; it += 256) {
        const int r = it / MT, c = it % MT;
        if (y0 + r + 10 < bh && x0 + c + 10 < bw) {    // inside the 'valid' map
            double m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
            for (int d = 0; d < 11; ++d) {
                const double g = s_g[d];
                const int o = (r + d) * MT + c;
                m1 += g * s_q[0 * MH * MT + o]; m2 += g * s_q[1 * MH * MT + o]; e11 += g * s_q[2 * MH * MT + o];
                e22 += g * s_q[3 * MH * MT + o]; e12 += g * s_q[4 * MH * MT + o];
            }
            const double s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
            ss += ((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2));
        }
    }
    se = wave_sum(se);
    ss = wave_sum(ss);
    if ((tid & 63) == 0) { s_red[tid >> 6] = se; s_red[4 + (tid >> 6)] = ss; }
    __syncthreads();
    if (tid == 0) {
        double* o = out + ((size_t)blk * gridDim.y + tile) * 2;
        o[0] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        o[1] = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    }
}

extern "C" int yond_block_metrics_tiles(int bh, int bw) {
    if (bh < 11 || bw < 11) return YOND_EINVAL;
    return ((bh + MT - 1) / MT) * ((bw + MT - 1) / MT);
}

extern "C" int yond_block_metrics_f32(const float* dn, const float* hr, int H, int W, int bh, int bw, double* out,
                                      void* stream) {
    if (!dn || !hr || !out || bh < 11 || bw < 11 || H < bh || W < bw || H % bh || W % bw) return YOND_EINVAL;
    const int nbx = W / bw, nby = H / bh;
    const int ntx = (bw + MT - 1) / MT, nty = (bh + MT - 1) / MT;
    const size_t smem = sizeof(double) * (2 * MH * MH + 5 * MH * MT);
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)block_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(block_metrics_kernel, dim3(nbx * nby, ntx * nty), dim3(256), smem, (hipStream_t)stream, dn, hr, W, bh, bw,
                       nbx, ntx, out);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}


// ------------------------------------------------------------------------------------------------------
// Measurement aid (bench.py): the shader clock the chip really holds while a workload runs.  One wave sleeps on one CU
// for `us` microseconds of wall time and returns the elapsed shader cycles (s_memtime) and reference ticks (s_memrealtime,
// 100 MHz): clock = cycles / ticks * 100 MHz (MI355X_MICROARCH.md 'DVFS give-back' item 6).  Launched on a side stream
// next to the timed region; it occupies one wave slot and issues no memory traffic but its final store.
// ------------------------------------------------------------------------------------------------------
__global__ void clock_probe_kernel(unsigned long long ticks, unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    for (unsigned long long guard = 0; r1 - r0 < ticks && guard < (1ull << 36); ++guard) {     // (bounded: every wave exits)
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

extern "C" int yond_clock_probe(double us, unsigned long long* out, void* stream) {
    if (!out || !(us > 0.0) || us > 5e6) return YOND_EINVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(us * 100.0), out);
    YOND_LAUNCH_CHECK();
    return YOND_OK;
}
