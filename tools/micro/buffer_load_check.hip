#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, float* y, int nbytes, int so) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, so, 0);
    for (int j = 0; j < 4; ++j) y[threadIdx.x * 4 + j] = __builtin_bit_cast(float, v[j]);
    float o = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0x80000000u + threadIdx.x * 4, so, 0));
    float o2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, nbytes - 4 + threadIdx.x * 4, 0, 0));
    y[64 + threadIdx.x] = o; y[128 + threadIdx.x] = o2;
}
int main() {
    float h[64], *dx, *dy, out[192];
    for (int i = 0; i < 64; ++i) h[i] = i + 1;
    hipMalloc(&dx, 256); hipMalloc(&dy, 192 * 4); hipMemcpy(dx, h, 256, hipMemcpyHostToDevice);
    k<<<1, 8>>>(dx, dy, 128, 16); hipMemcpy(out, dy, 192 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) printf("%g ", out[i]); printf("\n");
    for (int i = 0; i < 8; ++i) printf("%g ", out[64 + i]); printf("| ");
    for (int i = 0; i < 8; ++i) printf("%g ", out[128 + i]); printf("\n");
    return 0;
}
