// placement_vmm.hip -- development aid (round 6): does the physical backing of the solver's vectors decide
// the slow / fast mode of a lockstep multi-stream sweep?  Sets of 11 buffers of 128 MiB, 8 read + 3 written in lockstep (the
// shape of E+A+B), allocated two ways in ONE process on one box:
//   A. hipMalloc, one per buffer (what the library does)
//   B. hipMemCreate + hipMemMap, one physical allocation per buffer, at the minimum and at the recommended granularity
// build: hipcc --offload-arch=gfx950 -O3 -o tools/placement_vmm.bin tools/placement_vmm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr size_t N = (size_t)16 << 20;  // doubles per buffer: 128 MiB
struct P11 { double2 *p[11]; };
__global__ void k_11(P11 q) {
  const size_t per = (N / 2 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < N / 2 ? lo + per : N / 2;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    double2 s = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = q.p[k][i]; s.x += v.x; s.y += v.y; }
    q.p[8][i] = s; q.p[9][i] = s; q.p[10][i] = s;
  }
}
static hipEvent_t e0, e1;
static double gbs(const P11 &q) {
  hipLaunchKernelGGL(k_11, dim3(2048), dim3(256), 0, 0, q);
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_11, dim3(2048), dim3(256), 0, 0, q);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return 11.0 * N * 8 / 1e6 / (ms / 5);
}
static void *vmm_alloc(size_t bytes, size_t gran, int dev, hipMemGenericAllocationHandle_t *hout) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  const size_t sz = (bytes + gran - 1) / gran * gran;
  hipMemGenericAllocationHandle_t h;
  CK(hipMemCreate(&h, sz, &prop, 0));
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, sz, gran, nullptr, 0));
  CK(hipMemMap(va, sz, 0, h, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, sz, &acc, 1));
  if (hout) *hout = h;
  return va;
}
int main(int argc, char **argv) {
  const int NSETS = argc > 1 ? atoi(argv[1]) : 6;
  int dev = 0; CK(hipGetDevice(&dev));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("allocation granularity: minimum %zu B, recommended %zu B\n", gmin, grec);
  const auto report = [&](const char *what, std::vector<double> &v) {
    double lo = 1e30, hi = 0; for (double x : v) { lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
    printf("%-70s", what); for (double x : v) printf(" %6.0f", x); printf("   GB/s  (%.0f .. %.0f)\n", lo, hi);
  };
  {  // A
    std::vector<double> r;
    for (int s = 0; s < NSETS; s++) { P11 q; for (auto &p : q.p) { CK(hipMalloc(&p, N * 8)); CK(hipMemset(p, 0, N * 8)); } r.push_back(gbs(q)); }
    report("A. hipMalloc per buffer (all sets alive)", r);
  }
  for (size_t gran : {gmin, grec, (size_t)1 << 30}) {  // B
    if (gran < gmin) continue;
    std::vector<double> r;
    for (int s = 0; s < NSETS; s++) { P11 q; for (auto &p : q.p) { p = (double2 *)vmm_alloc(N * 8, gran, dev, nullptr); CK(hipMemset(p, 0, N * 8)); } r.push_back(gbs(q)); }
    char w[128]; snprintf(w, sizeof w, "B. hipMemCreate + hipMemMap per buffer, granularity / alignment %zu", gran);
    report(w, r);
  }
  return 0;
}
