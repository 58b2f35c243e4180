// Micro-probe: what does the fp32 MFMA pipe sustain on this box under the conv kernel's ingredients?
//   mode 0: 4 independent accumulators, operands in registers
//   mode 1: + 4 ds_read_b128 per 16 MFMAs (fragments re-read from LDS, one group ahead)
//   mode 2: mode 1 + one __syncthreads() per 288 MFMAs
//   mode 3: mode 2 + 6 global_load_dwordx4 + 6 ds_write_b128 per 288 MFMAs (register staging)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ src, float* __restrict__ out, int steps, int zero_data) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += 256) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (blockIdx.x * 40503u) ^ (unsigned)steps;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        smem[i] = zero_data ? (float)(i % 7) * 0.125f : ((float)(h & 0xffffff) / 8388608.0f - 1.0f);   // uniform [-1, 1)
    }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x4 fa[2], fb[2];
    fa[0] = *(f32x4*)(smem + tid * 4); fa[1] = *(f32x4*)(smem + 1024 + tid * 4);
    fb[0] = *(f32x4*)(smem + 2048 + tid * 4); fb[1] = *(f32x4*)(smem + 3072 + tid * 4);
    const float* gp = src + (size_t)blockIdx.x * 65536 + tid * 4;
    for (int s = 0; s < steps; ++s) {
        f32x4 vin[6];
        if (MODE >= 3) {
#pragma unroll
            for (int k = 0; k < 6; ++k) vin[k] = *(const f32x4*)(gp + ((s * 6 + k) & 31) * 1024);
        }
#pragma unroll
        for (int g = 0; g < 18; ++g) {
            f32x4 na[2], nb[2];
            if (MODE >= 1) {
                const int o = ((g + s) & 7) * 4096;
                na[0] = *(f32x4*)(smem + o + tid * 4); na[1] = *(f32x4*)(smem + o + 1024 + tid * 4);
                nb[0] = *(f32x4*)(smem + o + 2048 + tid * 4); nb[1] = *(f32x4*)(smem + o + 3072 + tid * 4);
            }
            if (MODE == 4) {
#pragma unroll
                for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[a4] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a4 >> 1][t], fb[a4 & 1][t], acc[a4], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][t], fb[0][t], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][t], fb[1][t], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][t], fb[0][t], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][t], fb[1][t], acc[3], 0, 0, 0);
            }
            }
            if (MODE >= 3 && g >= 10 && g < 16) *(f32x4*)(smem + 16384 + ((g - 10) * 256 + tid) * 4) = vin[g - 10];
            if (MODE >= 1) { fa[0] = na[0]; fa[1] = na[1]; fb[0] = nb[0]; fb[1] = nb[1]; }
            if (MODE >= 1 && MODE != 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            if (MODE != 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
static void run(const float* src, float* out, int steps, int blocks, int zero_data) {
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, 131072>>>(src, out, 4, zero_data);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, 131072>>>(src, out, steps, zero_data);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * steps * 288 * 4096.0;
    printf("zero=%d mode %d blocks %d: %.3f ms  %.1f TFLOP/s  err=%d\n", zero_data, MODE, blocks, ms, flops / ms / 1e9, (int)hipGetLastError());
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 200;
    float *src, *out;
    hipMalloc(&src, (size_t)512 * 65536 * 4 + (1 << 20));
    hipMemset(src, 0, (size_t)512 * 65536 * 4 + (1 << 20));
    hipMalloc(&out, 512 * 256 * 4);
    for (int z : {1, 0}) {
        run<0>(src, out, steps, 256, z);
        run<2>(src, out, steps, 256, z);
        run<3>(src, out, steps, 256, z);
        run<4>(src, out, steps, 256, z);
    }
    return 0;
}
