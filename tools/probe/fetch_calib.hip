// FETCH_SIZE calibration (MI355X_MICROARCH.md: "other access widths are uncalibrated"): stream the same 192 MiB buffer with 4-, 8-
// and 16-byte loads per lane; run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`.  Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T>
__global__ __launch_bounds__(256) void stream_read(const T* __restrict__ p, size_t n, float* out) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = p[i];
        if constexpr (sizeof(T) == 4) acc += v;
        else if constexpr (sizeof(T) == 8) acc += v[0] + v[1];
        else acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    const size_t bytes = 192ull << 20;
    float *buf, *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        stream_read<float><<<2048, 256>>>((const float*)buf, bytes / 4, out);
        stream_read<f2><<<2048, 256>>>((const f2*)buf, bytes / 8, out);
        stream_read<f4><<<2048, 256>>>((const f4*)buf, bytes / 16, out);
    }
    hipDeviceSynchronize();
    printf("streamed %zu MiB three times with 4 / 8 / 16 bytes per lane\n", bytes >> 20);
    return 0;
}
