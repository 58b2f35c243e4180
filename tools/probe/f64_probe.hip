// Issue cost of float64 vector instructions on gfx950 (one and two waves per SIMD): independent chains of v_add_f64,
// v_fma_f64, v_cvt_f64_f32, v_cvt_f32_f64, v_mul_f64 against v_add_f32.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
template <int OP>
__global__ void probe(double* out, unsigned long long* cyc, float seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7;
    const double c = 1.0000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N / 8; ++i) {
        if (OP == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
        if (OP == 1) { a0 = fma(a0, c, c); a1 = fma(a1, c, c); a2 = fma(a2, c, c); a3 = fma(a3, c, c); a4 = fma(a4, c, c); a5 = fma(a5, c, c); a6 = fma(a6, c, c); a7 = fma(a7, c, c); }
        if (OP == 2) { a0 += (double)f0; a1 += (double)f1; a2 += (double)f2; a3 += (double)f3; a4 += (double)f4; a5 += (double)f5; a6 += (double)f6; a7 += (double)f7;
                       f0 += 1.f; f1 += 1.f; f2 += 1.f; f3 += 1.f; f4 += 1.f; f5 += 1.f; f6 += 1.f; f7 += 1.f; }
        if (OP == 3) { f0 += 1.5f; f1 += 1.5f; f2 += 1.5f; f3 += 1.5f; f4 += 1.5f; f5 += 1.5f; f6 += 1.5f; f7 += 1.5f; }
        if (OP == 4) { a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c; }
        if (OP == 5) { f0 = (float)a0; f1 = (float)a1; f2 = (float)a2; f3 = (float)a3; a0 += f0; a1 += f1; a2 += f2; a3 += f3; a4 += c; a5 += c; a6 += c; a7 += c; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (threadIdx.x % 64 == 0 && blockIdx.x == 0) cyc[threadIdx.x / 64] = t1 - t0;
}
template <int OP>
void run(const char* name, int threads, int perit) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 64 * 8);
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(probe<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
    printf("%-34s %4d threads/WG: %.2f cycles per wave-instruction (wave 0), %.2f (last wave)\n", name, threads,
           (double)h[0] / (N / 8 * perit), (double)h[threads / 64 - 1] / (N / 8 * perit));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run<3>("v_add_f32", th, 8);
        run<0>("v_add_f64", th, 8);
        run<1>("v_fma_f64", th, 8);
        run<4>("v_mul_f64", th, 8);
        run<2>("v_cvt_f64_f32 + v_add_f64 + v_add_f32", th, 24);
        run<5>("4 v_cvt_f32_f64 + 4 cvt_f64_f32 + 8 add_f64", th, 16);
    }
    return 0;
}
