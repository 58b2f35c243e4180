// N1: per-block PSNR / SSIM partial sums on the device (YOND_SIDD.py:649-652, 679-697).
// SSIM as the reference computes it: images scaled to [0,255] in float64, 11x11 Gaussian window
// (sigma 1.5, cv2.getGaussianKernel), 'valid' part only, C1=(0.01*255)^2, C2=(0.03*255)^2.  The window
// is separable, so each 32x32 tile of the valid map needs a 42x42 input patch: horizontal pass for the
// five moments into LDS, vertical pass + SSIM formula + block reduction.  Output: one (sum of squared
// error, sum of SSIM) pair per tile; the host adds the tiles of a block in a fixed order (deterministic).
#include "common.h"

#define MT 32
#define MH (MT + 10)

struct MetricsGauss { double g[11]; };       // the normalised 11-tap window, computed once on the host (every workgroup used to evaluate 121 float64 exponentials)

__global__ __launch_bounds__(256) void block_metrics_kernel(const float* __restrict__ dn, const float* __restrict__ hr,
                                                            int W, int bh, int bw, int nbx, int ntx,
                                                            double* __restrict__ out, const MetricsGauss gw) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* s_q = sm;                      // [5][MH][MT]
    float* s_a = (float*)(s_q + 5 * MH * MT);   // [MH][MH]: dn * 255 and hr * 255 ARE float32 products (exact as floats; widened when read):
    float* s_b = s_a + MH * MH;                  // 68 KB of LDS instead of 82: two workgroups per CU
    __shared__ double s_g[11];
    __shared__ double s_red[8];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x, tile = blockIdx.y;
    const int by = blk / nbx, bx = blk % nbx;
    const int ty = tile / ntx, tx = tile % ntx;
    const int y0 = ty * MT, x0 = tx * MT;              // tile origin inside the block
    if (tid < 11) s_g[tid] = gw.g[tid];
    double se = 0.0;
    for (int it = tid; it < MH * MH; it += 256) {
        const int r = it / MH, c = it % MH;
        const int y = y0 + r, x = x0 + c;
        float a = 0.0f, b = 0.0f;
        if (y < bh && x < bw) {
            const size_t idx = (size_t)(by * bh + y) * W + (size_t)bx * bw + x;
            const float fa = dn[idx], fb = hr[idx];
            a = __fmul_rn(fa, 255.0f);                 // dn*255 is a float32 product (YOND_SIDD.py:652), then float64
            b = __fmul_rn(fb, 255.0f);
            if (r < MT && c < MT) { const double d = (double)fa - (double)fb; se += d * d; }
        }
        s_a[it] = a;
        s_b[it] = b;
    }
    __syncthreads();
    // Both passes reuse what they load: a thread owns four neighbouring outputs and walks the 14 inputs they share once, adding every
    // input into the outputs it belongs to -- each output still receives its eleven terms in the order d = 0 .. 10, so the sums are
    // bit-identical to the plain form (11 x 2 LDS reads and three recomputed products per output and tap before: the kernel was
    // bound by its LDS reads, 94 us for the 32 blocks of a SIDD image).
    // horizontal pass: task = (row r of the 42, group of four columns)
    for (int task = tid; task < MH * (MT / 4); task += 256) {
        const int r = task / (MT / 4), c0 = (task % (MT / 4)) * 4;
        double q[5][4];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[k][j] = 0.0;
#pragma unroll
        for (int e = 0; e < 14; ++e) {
            const double a = (double)s_a[r * MH + c0 + e], b = (double)s_b[r * MH + c0 + e];
            const double aa = a * a, bb = b * b, ab = a * b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = e - j;
                if (d >= 0 && d <= 10) {
                    const double g = s_g[d];
                    q[0][j] += g * a; q[1][j] += g * b; q[2][j] += g * aa; q[3][j] += g * bb; q[4][j] += g * ab;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_q[k * MH * MT + r * MT + c0 + j] = q[k][j];
    }
    __syncthreads();
    const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
    double ss = 0.0;
    {
        // vertical pass: thread = (column c, group of four rows)
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        double q[5][4];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) q[k][j] = 0.0;
#pragma unroll
        for (int e = 0; e < 14; ++e) {
            double v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = s_q[k * MH * MT + (r0 + e) * MT + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = e - j;
                if (d >= 0 && d <= 10) {
                    const double g = s_g[d];
#pragma unroll
                    for (int k = 0; k < 5; ++k) q[k][j] += g * v[k];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + j;
            if (y0 + r + 10 < bh && x0 + c + 10 < bw) {    // inside the 'valid' map
                const double m1 = q[0][j], m2 = q[1][j], e11 = q[2][j], e22 = q[3][j], e12 = q[4][j];
                const double s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
                ss += ((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s1 + s2 + C2));
            }
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
    const size_t smem = sizeof(double) * (5 * MH * MT) + sizeof(float) * (2 * MH * MH);
    MetricsGauss gw;
    {
        // cv2.getGaussianKernel(11, 1.5) as the oracle restates it: exp(-(i - 5)^2 / (2 sigma^2)) normalised by the sum in index order
        double gs = 0.0;
        for (int i = 0; i < 11; ++i) gs += exp(-((i - 5.0) * (i - 5.0)) / (2.0 * 1.5 * 1.5));
        for (int i = 0; i < 11; ++i) gw.g[i] = exp(-((i - 5.0) * (i - 5.0)) / (2.0 * 1.5 * 1.5)) / gs;
    }
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)block_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(block_metrics_kernel, dim3(nbx * nby, ntx * nty), dim3(256), smem, (hipStream_t)stream, dn, hr, W, bh, bw,
                       nbx, ntx, out, gw);
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
