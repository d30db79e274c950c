// Which XCD does workgroup b land on?  (speed-only knowledge for tile ordering)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
}
int main() {
    int* d; const int n = 64;
    hipMalloc(&d, n * 4);
    for (int threads : {64, 256}) {
        hipLaunchKernelGGL(k, dim3(n), dim3(threads), 0, 0, d);
        int h[n]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        printf("threads %3d:", threads);
        for (int i = 0; i < n; ++i) printf(" %d", h[i]);
        printf("\n");
    }
    return 0;
}
