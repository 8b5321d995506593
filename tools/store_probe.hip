// Write-policy probe (developer tool): fills 238 MB with 16-byte stores under different cache policies
// and times the fill alone and the fill followed by a streaming read of 275 MB of other data (what the
// step does next), so that the cost a policy pushes onto the next kernel is seen too.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/store_probe.hip -o tools/bin/store_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ void __launch_bounds__(256) fill_kernel(u32x4* p, size_t n16, unsigned int val) {
  const u32x4 v = {val, val + 1, val + 2, val + 3};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    u32x4* d = p + i;
    if constexpr (V == 0) *d = v;
    else if constexpr (V == 1) __builtin_nontemporal_store(v, d);
    else if constexpr (V == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(v) : "memory");
    else if constexpr (V == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(d), "v"(v) : "memory");
    else if constexpr (V == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(d), "v"(v) : "memory");
    else if constexpr (V == 5) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" ::"v"(d), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt sc0 sc1" ::"v"(d), "v"(v) : "memory");
  }
}

__global__ void __launch_bounds__(256) read_kernel(const u32x4* p, size_t n16, unsigned int* out) {
  unsigned int acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 v = __builtin_nontemporal_load(p + i);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int V>
int run(const char* name, u32x4** w, u32x4** r, unsigned int* out, size_t nw, size_t nr, hipStream_t st) {
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  std::vector<float> tf, tb;
  for (int it = 0; it < 24; ++it) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(fill_kernel<V>, dim3(4096), dim3(256), 0, st, w[it % 4], nw, (unsigned)it);
    CK(hipEventRecord(e1, st));
    hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, st, r[it % 4], nr, out);
    CK(hipEventRecord(e2, st));
    CK(hipEventSynchronize(e2));
    float a, b;
    CK(hipEventElapsedTime(&a, e0, e1));
    CK(hipEventElapsedTime(&b, e0, e2));
    if (it >= 4) { tf.push_back(a * 1e3f); tb.push_back(b * 1e3f); }
  }
  std::sort(tf.begin(), tf.end());
  std::sort(tb.begin(), tb.end());
  printf("%-28s fill %6.1f us (%.2f TB/s)   fill + next read %6.1f us\n", name, tf[tf.size() / 2],
         nw * 16.0 / tf[tf.size() / 2] / 1e6, tb[tb.size() / 2]);
  return 0;
}

int main() {
  const size_t nw = (size_t)1024 * 227 * 512 * 2 / 16, nr = (size_t)275 * 1000 * 1000 / 16;
  u32x4 *w[4], *r[4];
  unsigned int* out;
  for (int i = 0; i < 4; ++i) {
    CK(hipMalloc((void**)&w[i], nw * 16));
    CK(hipMalloc((void**)&r[i], nr * 16));
    CK(hipMemset(r[i], 1, nr * 16));
  }
  CK(hipMalloc((void**)&out, 16));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (run<0>("plain", w, r, out, nw, nr, st)) return 1;
  if (run<1>("nontemporal builtin (nt)", w, r, out, nw, nr, st)) return 1;
  if (run<2>("sc0 sc1", w, r, out, nw, nr, st)) return 1;
  if (run<3>("sc1", w, r, out, nw, nr, st)) return 1;
  if (run<4>("sc0", w, r, out, nw, nr, st)) return 1;
  if (run<5>("nt sc1", w, r, out, nw, nr, st)) return 1;
  if (run<6>("nt sc0 sc1", w, r, out, nw, nr, st)) return 1;
  return 0;
}
