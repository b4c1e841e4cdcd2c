// placement_probe.hip -- development aid: is the slow / fast mode of a multi-stream sweep a property of single buffers or of
// their combination?  NB buffers of 128 MiB, each its own hipMalloc, all alive.
//   1. every buffer alone: a read-only streaming pass (sum) and a read+write pass (x += 1): GB/s per buffer
//   2. sets of 11 consecutive buffers: a lockstep pass over 8 read + 3 written streams (the shape of the solver's E+A+B): GB/s
//   3. the same 11-stream pass over sets chosen as the 11 fastest / the 11 slowest buffers of 1.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/placement_probe.bin tools/placement_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr size_t N = (size_t)16 << 20;  // doubles per buffer: 128 MiB
__global__ void k_read(const double2 *a, double *out) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N / 2; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) out[0] = s;
}
__global__ void k_rw(double2 *a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N / 2; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; v.x += 1; v.y += 1; a[i] = v; }
}
struct P11 { double2 *p[11]; };
__global__ void k_11(P11 q) {  // 8 read, 3 written: contiguous chunk per workgroup, like the solver's tile walk
  const size_t per = (N / 2 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < N / 2 ? lo + per : N / 2;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    double2 s = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) { const double2 v = q.p[k][i]; s.x += v.x; s.y += v.y; }
    q.p[8][i] = s; q.p[9][i] = s; q.p[10][i] = s;
  }
}
__global__ void k_2(P11 q) {  // read a, read + write b, contiguous chunk per workgroup
  const size_t per = (N / 2 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < N / 2 ? lo + per : N / 2;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double2 a = q.p[0][i]; double2 v = q.p[1][i];
    v.x += a.x; v.y += a.y;
    q.p[1][i] = v;
  }
}
int main(int argc, char **argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 44;
  std::vector<double *> b(NB);
  for (auto &p : b) { CK(hipMalloc(&p, N * 8)); CK(hipMemset(p, 0, N * 8)); }
  double *out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto launch) { launch(); CK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
  std::vector<double> rd(NB), rw(NB);
  for (int k = 0; k < NB; k++) {
    rd[k] = N * 8 / 1e6 / timeit([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const double2 *)b[k], out); });
    rw[k] = 2 * N * 8 / 1e6 / timeit([&] { hipLaunchKernelGGL(k_rw, dim3(2048), dim3(256), 0, 0, (double2 *)b[k]); });
  }
  printf("per buffer GB/s (read | read+write):\n");
  for (int k = 0; k < NB; k++) printf("  %2d %p  %7.1f  %7.1f\n", k, (void *)b[k], rd[k], rw[k]);
  auto set11 = [&](const std::vector<int> &idx) { P11 q; for (int k = 0; k < 11; k++) q.p[k] = (double2 *)b[idx[k]]; return 11.0 * N * 8 / 1e6 / timeit([&] { hipLaunchKernelGGL(k_11, dim3(2048), dim3(256), 0, 0, q); }); };
  printf("11 consecutive buffers, lockstep 8 read + 3 written, GB/s:\n");
  for (int s = 0; s + 11 <= NB; s += 11) { std::vector<int> idx(11); std::iota(idx.begin(), idx.end(), s); printf("  buffers %2d..%2d: %7.1f\n", s, s + 10, set11(idx)); }
  std::vector<int> order(NB); std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int c) { return rw[a] > rw[c]; });
  std::vector<int> fast(order.begin(), order.begin() + 11), slow(order.end() - 11, order.end());
  printf("the 11 buffers fastest alone: %7.1f GB/s    the 11 slowest alone: %7.1f GB/s\n", set11(fast), set11(slow));
  // 3b. pairs: a lockstep pass over TWO buffers (one read, one read + written) for every pair of the first NP buffers: is the
  //     interference pairwise?  Then: the 11 buffers picked greedily for the best worst-pair, against the consecutive sets
  {
    const int NP = NB < 33 ? NB : 33;
    std::vector<std::vector<double>> M(NP, std::vector<double>(NP, 0.0));
    double lo = 1e30, hi = 0;
    for (int i = 0; i < NP; i++)
      for (int j = i + 1; j < NP; j++) {
        P11 q; for (int k = 0; k < 11; k++) q.p[k] = nullptr;
        q.p[0] = (double2 *)b[i]; q.p[1] = (double2 *)b[j];
        const double gb = 3.0 * N * 8 / 1e6 / timeit([&] { hipLaunchKernelGGL(k_2, dim3(2048), dim3(256), 0, 0, q); });
        M[i][j] = M[j][i] = gb; lo = gb < lo ? gb : lo; hi = gb > hi ? gb : hi;
      }
    printf("pairs of the first %d buffers (read a, read + write b in lockstep): %.0f .. %.0f GB/s; rows = buffer, entries = GB/s / 100\n", NP, lo, hi);
    for (int i = 0; i < NP; i++) { printf("  %2d:", i); for (int j = 0; j < NP; j++) printf(" %2.0f", i == j ? 0.0 : M[i][j] / 100); printf("\n"); }
    // greedy: start from the best pair, add the buffer whose worst pair with the chosen ones is best
    std::vector<int> pick; int bi = 0, bj = 1;
    for (int i = 0; i < NP; i++) for (int j = i + 1; j < NP; j++) if (M[i][j] > M[bi][bj]) { bi = i; bj = j; }
    pick = {bi, bj};
    while ((int)pick.size() < 11) {
      int best = -1; double bestw = -1;
      for (int c = 0; c < NP; c++) {
        if (std::find(pick.begin(), pick.end(), c) != pick.end()) continue;
        double w = 1e30; for (int p : pick) w = M[c][p] < w ? M[c][p] : w;
        if (w > bestw) { bestw = w; best = c; }
      }
      pick.push_back(best);
    }
    std::vector<int> worstpick; { // and the opposite: greedily the worst pairs
      int wi = 0, wj = 1; for (int i = 0; i < NP; i++) for (int j = i + 1; j < NP; j++) if (M[i][j] < M[wi][wj]) { wi = i; wj = j; }
      worstpick = {wi, wj};
      while ((int)worstpick.size() < 11) {
        int best = -1; double bestw = 1e30;
        for (int c = 0; c < NP; c++) {
          if (std::find(worstpick.begin(), worstpick.end(), c) != worstpick.end()) continue;
          double w = 0; for (int p : worstpick) w += M[c][p];
          if (w < bestw) { bestw = w; best = c; }
        }
        worstpick.push_back(best);
      }
    }
    printf("11 streams: greedy best-pairs set %7.1f GB/s   greedy worst-pairs set %7.1f GB/s   (picked:", set11(pick), set11(worstpick));
    for (int p : pick) printf(" %d", p); printf(")\n");
  }
  // 4. ONE allocation, the 11 streams at base + k * (128 MiB + delta): does a spacing exist that is reliably fast?
  for (auto &p : b) CK(hipFree(p));
  char *big = nullptr;
  const size_t slab = (size_t)11 * ((N * 8) + ((size_t)64 << 20)) + ((size_t)1 << 30);
  CK(hipMalloc(&big, slab));
  CK(hipMemset(big, 0, slab));
  auto spaced = [&](size_t delta, size_t base_off) {
    P11 q;
    for (int k = 0; k < 11; k++) q.p[k] = (double2 *)(big + base_off + (size_t)k * (N * 8 + delta));
    return 11.0 * N * 8 / 1e6 / timeit([&] { hipLaunchKernelGGL(k_11, dim3(2048), dim3(256), 0, 0, q); });
  };
  printf("one allocation, stream k at k * (128 MiB + delta), GB/s for base offsets 0 / 1 MiB / 37 MiB:\n");
  const size_t deltas[] = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 65536 + 4096 + 256, 1 << 20, (1 << 20) + 4096 + 256, 2 << 20, (2 << 20) + 256, 3 << 20, 5 << 20,
                           8 << 20, (8 << 20) + 4096, 16 << 20, (16 << 20) + 256 * 7, 32 << 20, 48 << 20, (size_t)64 << 20};
  for (size_t d : deltas) printf("  delta %9zu: %7.1f  %7.1f  %7.1f\n", d, spaced(d, 0), spaced(d, (size_t)1 << 20), spaced(d, (size_t)37 << 20));
  // random offsets (multiples of 256 B) inside per-stream windows of 192 MiB
  srand(12345);
  printf("random offsets (stream k somewhere in its 192 MiB window), GB/s:\n ");
  for (int trial = 0; trial < 24; trial++) {
    P11 q;
    for (int k = 0; k < 11; k++) q.p[k] = (double2 *)(big + (size_t)k * ((size_t)192 << 20) + (size_t)(rand() % (1 << 18)) * 256);
    printf(" %6.0f", 11.0 * N * 8 / 1e6 / timeit([&] { hipLaunchKernelGGL(k_11, dim3(2048), dim3(256), 0, 0, q); }));
  }
  printf("\n");
  return 0;
}
