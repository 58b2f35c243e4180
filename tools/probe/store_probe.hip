// Micro-probe: cost of the conv epilogue's store patterns (388 MB written by 256 workgroups of 4 waves).
//   pattern 0: 16 B per lane, lane stride 128 B, lanes l / l+32 adjacent 16-B chunks (32 partial lines per instruction)
//   pattern 1: 16 B per lane, fully contiguous 1 KiB per wave instruction
//   pattern 2: 4 B per lane, 32 lanes contiguous (2 full 128-B lines per instruction)  [v1 epilogue]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256) void st(float* __restrict__ dst, int ntiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // tile = 8 rows x 32 px x 32 ch (C = 32): 32 KB; wave w owns rows 2w, 2w+1
        float* base = dst + (size_t)t * 8192 + wave * 2048;
        if (P == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) { f32x4 v = {1.f * t, 2.f, 3.f, 4.f}; *(f32x4*)(base + m * 1024 + li * 32 + 8 * g + 4 * lh) = v; }
        } else if (P == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { f32x4 v = {1.f * t, 2.f, 3.f, 4.f}; *(f32x4*)(base + k * 256 + lane * 4) = v; }
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) base[m * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = 1.f * t;
        }
    }
}

template <int P>
static void run(float* dst, int ntiles, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    st<P><<<grid, 256>>>(dst, ntiles); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) st<P><<<grid, 256>>>(dst, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("pattern %d grid %d: %.1f us  %.2f TB/s\n", P, grid, ms * 1e3, (double)ntiles * 32768 / ms / 1e9);
}

int main() {
    const int ntiles = 11844;
    float* dst; hipMalloc(&dst, (size_t)ntiles * 32768 + 4096);
    for (int grid : {256, 1024, 4096}) { run<0>(dst, ntiles, grid); run<1>(dst, ntiles, grid); run<2>(dst, ntiles, grid); }
    return 0;
}
