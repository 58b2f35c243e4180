// ds_read_b64_tr_b16 semantics probe: LDS holds a [pixel][32 channels] fp16 tile with value = pixel * 64 + channel; every lane
// issues one transposed read with the address rule of MI355X guide T10 and prints what it received.
//   hipcc --offload-arch=gfx950 tools/probe/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(float* out) {
    __shared__ _Float16 t[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) t[i] = (_Float16)((i / 32) * 64 + (i % 32));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
    // group g: channels 16 (g & 1) .., pixels 8 (g >> 1) + q ; the lane supplies row q, column chunk p
    const int pixel = 8 * (g >> 1) + q, ch = 16 * (g & 1) + 4 * p;
    auto ptr = (__attribute__((address_space(3))) v4s*)(t + pixel * 32 + ch);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int e = 0; e < 4; ++e) {
        union { short s; _Float16 h; } u;
        u.s = r[e];
        out[lane * 4 + e] = (float)u.h;
    }
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  px %d ch %2d", (int)h[l * 4 + e] / 64, (int)h[l * 4 + e] % 64);
        printf("\n");
    }
    return 0;
}
