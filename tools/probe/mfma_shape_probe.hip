// Which fp16 MFMA shape does more split-operand work per second UNDER THE POWER CAP?  (MI355X_MICROARCH.md, 'DVFS give-back' item 7: in bare
// bf16 loops on random data the 16x16x32 shape delivered 1.12-1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP -- the chip holds a
// higher clock.)  The dominant convolution kernel runs at the cap (1.85-2.0 GHz of 2.4), so the question decides whether a 16x16x32 rewrite
// could pay.  Both variants execute the split-operand pattern -- h_w h_x into one accumulator, h_w l_x + l_w h_x into a second -- with every
// operand re-read from LDS (random fp16 data: power depends on the toggling), 512-thread workgroups, two waves per SIMD, 256 workgroups,
// at the LDS traffic per MFMA cycle of conv_split_kernel<1,12,64,3> (0.59 KiB per 32 cycles).
//   32x32x16: per group 2 weight fragments (h, l) + 2 pixel fragments (h, l) feed 3 MFMAs for each of RB row blocks sharing the weights
//   16x16x32: the cross terms are ONE MFMA ([h_w | l_w] . [l_x ; h_x], K = 32), the main terms of TWO taps another (K = 2 x 16 channels)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip && ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k32(const uint4* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 8192; i += 512) ((uint4*)lds)[i] = src[i];          // 128 KB of random halves
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f16v a0[3], a1[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { a0[m][r] = 0.f; a1[m][r] = 0.f; }
    const char* base = lds + wave * 4096 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        // one "tap": weights (h, l) once, three rows' pixel fragments (h, l): 8 KB of fragments for 9 MFMAs = 0.89 KB per MFMA ... the real
        // kernel shares pixel fragments between taps: 15 + 9 fragment pairs per 27 product blocks = 0.59 KB per MFMA; here 5 pairs per 9:
        const int o = (it & 7) * 8192;
        const h8 wh = *(const h8*)(base + o), wl = *(const h8*)(base + o + 1024);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const h8 xh = *(const h8*)(base + o + 2048 + m * 2048 - (m ? 1024 * (m - 1) : 0)), xl = *(const h8*)(base + o + 3072 + m * 1024);
            a1[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, a1[m], 0, 0, 0);
            a0[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, a0[m], 0, 0, 0);
            a1[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, a1[m], 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m) r += a0[m][0] + a1[m][5];
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(clk, __builtin_amdgcn_s_memtime() - c0);
        atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - r0);
    }
}

__global__ __launch_bounds__(512) void k16(const uint4* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 8192; i += 512) ((uint4*)lds)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // the same output tile per wave: 3 rows x 32 pixels x 32 channels = 3 x (2 x 2) blocks of 16 x 16, two accumulators each
    f4 a0[3][4], a1[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) { a0[m][b] = f4{0.f, 0.f, 0.f, 0.f}; a1[m][b] = f4{0.f, 0.f, 0.f, 0.f}; }
    const char* base = lds + wave * 4096 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        // TWO taps per iteration (the main terms pair two taps in K): per tap and channel block a cross fragment [h_w | l_w] (1 KB), per tap
        // and pixel block a cross fragment [l_x ; h_x]; per tap pair a main fragment each.  18 KB... the same bytes per MAC as the 32x32 loop:
        const int o = (it & 3) * 16384;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            h8 wc[2], xc[3][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) wc[cb] = *(const h8*)(base + o + t * 8192 + cb * 1024);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) xc[m][pb] = *(const h8*)(base + o + t * 8192 + 2048 + (m * 2 + pb) * 1024);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        a1[m][pb * 2 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[cb], xc[m][pb], a1[m][pb * 2 + cb], 0, 0, 0);
        }
        {
            h8 wm[2], xm[3][2];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) wm[cb] = *(const h8*)(base + o + 512 + cb * 1024);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) xm[m][pb] = *(const h8*)(base + o + 8192 + 512 + (m * 2 + pb) * 1024);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        a0[m][pb * 2 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[cb], xm[m][pb], a0[m][pb * 2 + cb], 0, 0, 0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int b = 0; b < 4; ++b) r += a0[m][b][0] + a1[m][b][3];
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(clk, __builtin_amdgcn_s_memtime() - c0);
        atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - r0);
    }
}

int main() {
    uint4* src; float* out; unsigned long long* clk;
    hipMalloc(&src, 131072); hipMalloc(&out, 4096); hipMalloc(&clk, 16);
    unsigned short* h = (unsigned short*)malloc(131072);
    srand(1);
    for (int i = 0; i < 65536; ++i) {                        // random halves in [-2, 2): sign, exponent 12..15, random mantissa
        const unsigned m = rand() & 0x3ff, e = 12 + (rand() & 3), s = rand() & 1;
        h[i] = (unsigned short)((s << 15) | (e << 10) | m);
    }
    hipMemcpy(src, h, 131072, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    // the two loops do the same MACs per iteration pair: k32: 9 MFMAs x 16,384 = 147,456 MAC per wave and iteration (one tap);
    // k16: 36 MFMAs x 8,192 = 294,912 per iteration (two taps).  Run k32 with 2x the iterations.
    const int it16 = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 2; ++which) {
            hipMemset(clk, 0, 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k32, dim3(256), dim3(512), 131072, 0, src, out, 2 * it16, clk);
            else hipLaunchKernelGGL(k16, dim3(256), dim3(512), 131072, 0, src, out, it16, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double macs = 256.0 * 8 * 294912.0 * it16;
            printf("%s: %8.2f ms  %7.1f TFLOP/s issued (%5.3f of 2.5 PF)  in-kernel clock %6.1f MHz  cycles per 16,384-MAC %5.1f\n",
                   which == 0 ? "32x32x16" : "16x16x32", ms, 2 * macs / ms / 1e9, 2 * macs / ms / 1e9 / 2500.0, c[1] ? (double)c[0] / c[1] * 100.0 : 0.0,
                   c[0] / (2.0 * it16 * 9.0 * 2.0));
        }
    }
    return 0;
}
