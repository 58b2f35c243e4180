// Do fp32 MFMAs of one wave and plain VALU work of ANOTHER wave on the same SIMD run concurrently on gfx950?
// 512-thread workgroups (two waves per SIMD): waves 0-3 issue MFMAs, waves 4-7 issue VALU FMAs.
//   hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip && ./coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>   // 0: 16x16x4 f32, 1: 32x32x2 f32
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int do_mfma, int do_valu) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            if (KIND == 0) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
                const float x = threadIdx.x * 1e-3f, y = 1.0f;
                for (int i = 0; i < n_mfma; i += 4) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                }
                r = a0[0] + a1[1] + a2[2] + a3[3];
            } else {
                f32x16 a0 = {0}, a1 = a0;
                const float x = threadIdx.x * 1e-3f, y = 1.0f;
                for (int i = 0; i < n_mfma; i += 2) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                }
                r = a0[0] + a1[1];
            }
        }
    } else if (do_valu) {
        float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
        const float c = 1.0001f, e = 0.5f;
        for (int i = 0; i < n_valu; i += 8) {
            v0 = __builtin_fmaf(v0, c, e); v1 = __builtin_fmaf(v1, c, e); v2 = __builtin_fmaf(v2, c, e); v3 = __builtin_fmaf(v3, c, e);
            v4 = __builtin_fmaf(v4, c, e); v5 = __builtin_fmaf(v5, c, e); v6 = __builtin_fmaf(v6, c, e); v7 = __builtin_fmaf(v7, c, e);
        }
        r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static float run(float* d, int nm, int nv, int dm, int dv) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, nm, nv, dm, dv);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, nm, nv, dm, dv);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int nm16 = 16384, nm32 = 8192, nv = 65536;
    printf("16x16x4 f32: mfma only %.1f us, valu only %.1f us, both %.1f us\n", run<0>(d, nm16, nv, 1, 0), run<0>(d, nm16, nv, 0, 1), run<0>(d, nm16, nv, 1, 1));
    printf("32x32x2 f32: mfma only %.1f us, valu only %.1f us, both %.1f us\n", run<1>(d, nm32, nv, 1, 0), run<1>(d, nm32, nv, 0, 1), run<1>(d, nm32, nv, 1, 1));
    for (int nvv : {16384, 32768, 131072})
        printf("16x16x4 f32 with %d valu: mfma only %.1f, valu only %.1f, both %.1f us\n", nvv, run<0>(d, nm16, nvv, 1, 0), run<0>(d, nm16, nvv, 0, 1), run<0>(d, nm16, nvv, 1, 1));
    return 0;
}
