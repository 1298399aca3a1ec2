// Does an out-of-range lane of buffer_load_dwordx4 ... lds write ZEROS to LDS or leave it alone?  (gfx950: zeros.)  The kernel fills LDS with
// 0xAAAAAAAA, loads 64 lanes x 16 B through a 512-byte buffer resource (lanes 32..63 out of range) and prints one dword per lane.
// hipcc --offload-arch=gfx950 -O3 -o buffer_lds_probe buffer_lds_probe.hip   ->  lanes 0..31: 11111111, lanes 32..63: 00000000
#include <hip/hip_runtime.h>
__global__ void k(void *p, int n, int *out) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ((int *)smem)[i] = 0xAAAAAAAA;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, n, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)smem, 16, threadIdx.x * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = ((int *)smem)[threadIdx.x * 4];
}
int main() {
    int *src, *out, h[256];
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    hipMemset(src, 0x11, 4096);
    hipMemset(out, 0xff, 1024);
    // LDS pre-filled? dynamic smem uninitialised: the kernel prints what landed; OOB lanes (offset >= n) show whether zero was written
    k<<<1, 64, 4096>>>(src, 512, out);   // 512 bytes in range = lanes 0..31; lanes 32..63 out of range
    hipMemcpy(h, out, 256 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 4) printf("lane %2d: %08x\n", i, h[i]);
    return 0;
}
