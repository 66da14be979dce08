// Where does global_load_lds with 12 bytes per lane put a lane's bytes in LDS?  Dumps the LDS image.
// Measured on MI355X: lane i's three dwords land at byte 16 i (not 12 i); the fourth dword of each 16 is left as it was.
//   hipcc --offload-arch=gfx950 -O2 lds_dma12.hip -o lds_dma12 && ./lds_dma12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
__global__ void k(const unsigned* src, unsigned* out) {
    __shared__ unsigned lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)((const char*)src + lane * 12), (LDS_AS void*)&lds[0], 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(256), o(512);
    for (int i = 0; i < 256; ++i) h[i] = i;
    unsigned *d, *dout;
    (void)hipMalloc(&d, 1024); (void)hipMalloc(&dout, 2048);
    (void)hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    (void)hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 272; ++i) printf("%s%x", i % 16 ? " " : "\n", o[i]);
    printf("\n");
    return 0;
}
