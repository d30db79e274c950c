// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does each lane of a 16-lane group receive?
// LDS image: 16-bit value r*256 + c at row r, column c (row stride 72 elements).  Each group g reads the 4-row x
// 16-column block whose first row is 4g: lane 4q+p supplies the address of (row 4g+q, columns 4p..4p+3).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short sm[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) sm[i] = (short)((i / 72) * 256 + (i % 72));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, l = lane & 15, q = l >> 2, p = l & 3;
    __attribute__((address_space(3))) v4s* ptr = (__attribute__((address_space(3))) v4s*)(sm + (4 * g + q) * 72 + 4 * p);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    *(v4s*)(out + lane * 4) = v;
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 64; lane += 1) {
        if (lane % 16 < 3 || lane % 16 == 15) {
            printf("lane %2d:", lane);
            for (int e = 0; e < 4; ++e) printf("  (r%d,c%d)", h[lane * 4 + e] / 256, h[lane * 4 + e] % 256);
            printf("\n");
        }
    }
    return 0;
}
