// ranges.h -- which groups of blocks (or quads) a workgroup walks, and which table entries a wave reads ahead while it
// does.  Host+device and free of HIP built-ins: the kernels call these with (gridDim.x, blockIdx.x), tests/walk_emul.cpp
// replays the very same functions for every workgroup of a launch on the CPU and checks that no index leaves the table
// (round 2 shipped a read-ahead that did: the review's finding on advect.hip:148-151).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RANGES_HD __host__ __device__ __forceinline__
#else
#define RANGES_HD inline
#endif

namespace cup2d {

struct GroupRange {
  int begin, end, stride;
};

// persistent grid of G workgroups, workgroup w: each XCD (w % 8, MI355X_MICROARCH.md "block b runs on XCD b % 8") takes a
// contiguous eighth of the groups, its workgroups interleave inside it
RANGES_HD GroupRange group_range_of(int count, int wpg, int G, int w) {
  const int groups = (count + wpg - 1) / wpg;
  GroupRange r;
  if (G >= 8 && (G % 8) == 0) {
    const int xcd = w & 7, slot = w >> 3, per = G >> 3;
    const long long lo = (long long)groups * xcd / 8, hi = (long long)groups * (xcd + 1) / 8;
    r.begin = (int)lo + slot;
    r.end = (int)hi;
    r.stride = per;
  } else {
    r.begin = w;
    r.end = groups;
    r.stride = G;
  }
  return r;
}
// chunked grid (chunked_grid() workgroups): workgroup w walks `chunk` consecutive groups of its XCD's eighth
RANGES_HD GroupRange group_range_chunked_of(int count, int wpg, int chunk, int w) {
  const int groups = (count + wpg - 1) / wpg;
  const int xcd = w & 7, slot = w >> 3;
  const long long lo = (long long)groups * xcd / 8, hi = (long long)groups * (xcd + 1) / 8;
  GroupRange r;
  r.begin = (int)lo + slot * chunk;
  const int e = r.begin + chunk;
  r.end = (int)hi < e ? (int)hi : e;
  r.stride = 1;
  return r;
}

// A wave's walk over the items (quads, blocks) `g * wpg + wave` of its workgroup's groups, with table reads issued AHEAD
// of the item they describe.  A read-ahead past the wave's last item is redirected to that last item (branch-free code
// downstream: the value is loaded and dropped), so `glast` only ever names a group whose item exists.
struct WaveCursor {
  int g, glast, end, stride, wave, wpg, n;
  bool have;
  RANGES_HD bool valid(int gg) const { return gg < end && gg * wpg + wave < n; }
  RANGES_HD void init(const GroupRange &r, int wave_, int wpg_, int n_) {
    g = glast = r.begin, end = r.end, stride = r.stride, wave = wave_, wpg = wpg_, n = n_;
    have = valid(g);
  }
  RANGES_HD void advance() {
    g += stride;
    have = valid(g);
    if (have) glast = g;
  }
  // the item `ahead` groups after the current position, or the wave's last one; only meaningful after init() found an item
  RANGES_HD int item(int ahead) const {
    const int gg = g + ahead * stride;
    return (valid(gg) ? gg : glast) * wpg + wave;
  }
};

}  // namespace cup2d
