#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x * 3 + 1;
    int a = __builtin_amdgcn_update_dpp(0, v, 0x150, 0xf, 0xf, false);   // row_newbcast:0
    int b = __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true);    // row_shl:1, bound_ctrl
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
}
int main() {
    int* d; hipMalloc(&d, 128 * 4); k<<<1, 64>>>(d); int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 128; ++i) printf("%d%c", h[i], (i % 16 == 15) ? '\n' : ' ');
    return 0;
}
