// Which XCD does workgroup b of a grid land on?  Prints HW_REG_XCC_ID per block for a plain and a cooperative launch.
// Build: hipcc -O2 --offload-arch=gfx950 xcc_probe.hip -o xcc_probe
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
}

int main() {
  const int nb = 64;
  unsigned* d;
  hipMalloc(&d, nb * 4);
  unsigned h[nb];
  for (int coop = 0; coop < 2; ++coop) {
    hipMemset(d, 0xff, nb * 4);
    if (coop) {
      void* args[] = {&d};
      hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k), dim3(nb), dim3(256), args, 0, 0);
      if (e != hipSuccess) printf("cooperative launch failed: %s\n", hipGetErrorString(e));
    } else {
      hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d);
    }
    hipDeviceSynchronize();
    hipMemcpy(h, d, nb * 4, hipMemcpyDeviceToHost);
    printf("%s:", coop ? "cooperative" : "plain");
    for (int b = 0; b < nb; ++b) printf(" %u", h[b]);
    printf("\n");
  }
  return 0;
}
