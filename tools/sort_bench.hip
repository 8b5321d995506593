// Developer tool: time of the in-workgroup ranking primitives of select.h, one workgroup per CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vqa_amd/csrc -I include tools/sort_bench.hip -o tools/bin/sort_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

#include "select.h"

template <int WHICH>
__global__ void __launch_bounds__(1024) sort_kernel(const unsigned long long* in, unsigned long long* out, int NP, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  unsigned long long acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i < NP; i += blockDim.x) keys[i] = in[(size_t)blockIdx.x * NP + i] ^ (unsigned long long)(r * 2654435761u);
    __syncthreads();
    if (WHICH == 0) bitonic_sort_desc_lds(keys, NP);
    if (WHICH == 1) bitonic_sort_desc(keys, NP);
    for (int i = threadIdx.x; i < NP; i += blockDim.x) acc += keys[i] * (unsigned long long)(i + 1);
    __syncthreads();
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int NP = argc > 1 ? atoi(argv[1]) : 1024, nt = argc > 2 ? atoi(argv[2]) : 512, reps = 50, wgs = 256;
  unsigned long long *in, *out;
  CK(hipMalloc((void**)&in, (size_t)wgs * NP * 8));
  CK(hipMalloc((void**)&out, (size_t)wgs * 1024 * 8));
  unsigned long long* h = (unsigned long long*)malloc((size_t)wgs * NP * 8);
  unsigned long long x = 88172645463325252ull;
  for (size_t i = 0; i < (size_t)wgs * NP; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x; }
  CK(hipMemcpy(in, h, (size_t)wgs * NP * 8, hipMemcpyHostToDevice));
  unsigned long long sums[2] = {0, 0};
  for (int which = 0; which < 2; ++which) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(sort_kernel<0>, dim3(wgs), dim3(nt), NP * 8, 0, in, out, NP, reps);
      else hipLaunchKernelGGL(sort_kernel<1>, dim3(wgs), dim3(nt), NP * 8, 0, in, out, NP, reps);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it == 2) printf("%s NP=%d threads=%d: %.2f us per sort (incl. fill and checksum)\n", which ? "registers" : "lds      ", NP, nt, ms * 1e3 / reps);
    }
    unsigned long long* ho = (unsigned long long*)malloc((size_t)wgs * nt * 8);
    CK(hipMemcpy(ho, out, (size_t)wgs * nt * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)wgs * nt; ++i) sums[which] += ho[i];
    free(ho);
  }
  printf("checksums %s\n", sums[0] == sums[1] ? "equal" : "DIFFER");
  return sums[0] == sums[1] ? 0 : 1;
}
