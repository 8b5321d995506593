// Timing of the two query-fold GEMMs outside Python (developer tool).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ivqa_amd/csrc -Iinclude tools/gemm_probe.hip -o tools/bin/gemm_probe
// Prints the median time of vqaf_fold_query_multi (both launches) over 50 calls for B = argv[1] (1024).
#include "../vqa_amd/csrc/gemm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, D = 512, H = 8, n = 2;
  void *q[2], *w[2], *bias[2];
  float *qp[2], *u[2];
  for (int i = 0; i < n; ++i) {
    CK(hipMalloc(&q[i], (size_t)B * D * 2));
    CK(hipMalloc(&w[i], (size_t)3 * D * D * 2));
    CK(hipMalloc(&bias[i], (size_t)3 * D * 2));
    CK(hipMalloc((void**)&qp[i], (size_t)B * D * 4));
    CK(hipMalloc((void**)&u[i], (size_t)B * H * D * 4));
    CK(hipMemset(q[i], 0x3c, (size_t)B * D * 2));
    CK(hipMemset(w[i], 0x3c, (size_t)3 * D * D * 2));
    CK(hipMemset(bias[i], 0x3c, (size_t)3 * D * 2));
  }
  vqaf_fold_desc descs[2];
  for (int i = 0; i < n; ++i) descs[i] = vqaf_fold_desc{q[i], w[i], bias[i], qp[i], u[i]};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int it = 0; it < 60; ++it) {
    CK(hipEventRecord(e0, st));
    const int rc = vqaf_fold_query_multi(descs, n, B, D, H, VQAF_BF16, VQAF_BF16, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    if (rc != 0) { fprintf(stderr, "rc %d\n", rc); return 1; }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 10) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  printf("B=%d fold (both launches): median %.1f us  min %.1f us\n", B, ts[ts.size() / 2], ts[0]);
  return 0;
}
