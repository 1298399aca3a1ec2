// What v_permlane16_swap_b32 does on gfx950 (used by the convolution epilogue, csrc/posepaf_conv_own.hip): prints, per lane, the two
// returned values for x = 100 + lane, y = 200 + lane.  hipcc --offload-arch=gfx950 -O2 -o permlane16_swap_probe permlane16_swap_probe.hip
// Observed: r0 = {even 16-lane rows: own x, odd rows: y of lane - 16}, r1 = {even rows: x of lane + 16, odd rows: own y}.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *a, unsigned *b) {
    unsigned x = 100 + threadIdx.x, y = 200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    a[threadIdx.x] = r[0];
    b[threadIdx.x] = r[1];
}
int main() {
    unsigned *a, *b, ha[64], hb[64];
    hipMalloc(&a, 256); hipMalloc(&b, 256);
    k<<<1, 64>>>(a, b);
    hipMemcpy(ha, a, 256, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 8) printf("lane %2d: r0=%u r1=%u\n", i, ha[i], hb[i]);
    return 0;
}
