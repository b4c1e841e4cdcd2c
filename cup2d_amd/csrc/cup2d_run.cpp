// cup2d_run.cpp -- a C++ host driver over the C ABI (include/cup2d_hip.h): the reference's time loop on a uniform grid
// with every block operator on the GPU.  What a CUP2D maintainer's main() looks like once the call sites of
// main.cpp:6606-6642, 7007-7027, 7115-7187 forward to the library (seam B2) -- here as a stand-alone
// program so that the sequence exists as compiled host code and not only as the Python mirror the tests use.
//
//   cup2d_run -n 256 [-ny 256] [-steps 10] [-nu 1e-3] [-cfl 0.5] [-poissonTol 1e-3] [-poissonTolRel 1e-2]
//             [-maxPoissonRestarts 0] [-maxiter 1000] [-init vel.f64] [-dump prefix] [-every k] [-device 0]
//
// Grid: nx x ny cells in 8 x 8 blocks, ordered along the Hilbert curve like the reference's (main.cpp:347-360,
// 1550-1562), walls on all four sides, h = extent / max(nx, ny) (main.cpp:6338, extent 1).  Initial velocity: the file
// given with -init (float64, row-major [ny][nx][2]) or the Taylor-Green vortex.  The first ten steps solve with zero
// tolerances and 100 restarts, later ones with the given tolerances (main.cpp:7028-7030).  -dump writes the
// reference's dump() files (main.cpp:3367-3466: <prefix>.<step>.xyz.raw / .attr.raw / .xdmf2) every k-th step.
// Only host code here: no kernels and no CPU fallback -- without the library's GPU path it fails.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cup2d_hip.h"

namespace {

constexpr int BS = CUP2D_BS, BC = BS * BS;

#define RUN(expr)                                                                                   \
  do {                                                                                              \
    const int rc_ = (expr);                                                                         \
    if (rc_ != CUP2D_OK) {                                                                          \
      std::fprintf(stderr, "cup2d_run: %s -> %d: %s\n", #expr, rc_, cup2d_last_error());            \
      std::exit(1);                                                                                 \
    }                                                                                               \
  } while (0)

uint64_t hilbert(int bits, uint64_t x, uint64_t y) {
  const uint64_t n = 1ull << bits;
  uint64_t d = 0;
  for (uint64_t s = n >> 1; s > 0; s >>= 1) {
    const uint64_t rx = (x & s) ? 1 : 0, ry = (y & s) ? 1 : 0;
    d += s * s * ((3 * rx) ^ ry);
    if (ry == 0) {
      if (rx == 1) {
        x = n - 1 - x;
        y = n - 1 - y;
      }
      std::swap(x, y);
    }
  }
  return d;
}

struct Grid {
  int nbx, nby, nblocks;
  std::vector<int> bx, by;    // block coordinates in device order
  std::vector<int32_t> nbr;   // [nblocks][4] W, E, S, N; CUP2D_WALL at a wall
  Grid(int nbx_, int nby_) : nbx(nbx_), nby(nby_), nblocks(nbx_ * nby_) {
    int bits = 1;
    while ((1 << bits) < std::max(std::max(nbx, nby), 2)) bits++;
    std::vector<std::pair<uint64_t, int>> key(nblocks);
    for (int y = 0; y < nby; y++)
      for (int x = 0; x < nbx; x++) key[y * nbx + x] = {hilbert(bits, x, y), y * nbx + x};
    std::stable_sort(key.begin(), key.end());
    bx.resize(nblocks);
    by.resize(nblocks);
    std::vector<int> index_of(nblocks);
    for (int b = 0; b < nblocks; b++) {
      bx[b] = key[b].second % nbx;
      by[b] = key[b].second / nbx;
      index_of[key[b].second] = b;
    }
    nbr.assign((size_t)4 * nblocks, CUP2D_WALL);
    for (int b = 0; b < nblocks; b++) {
      const int x = bx[b], y = by[b];
      if (x > 0) nbr[4 * b + 0] = index_of[y * nbx + x - 1];
      if (x < nbx - 1) nbr[4 * b + 1] = index_of[y * nbx + x + 1];
      if (y > 0) nbr[4 * b + 2] = index_of[(y - 1) * nbx + x];
      if (y < nby - 1) nbr[4 * b + 3] = index_of[(y + 1) * nbx + x];
    }
  }
  // row-major [ny][nx][dim] <-> block slab [nblocks][64][dim] (Info::block layout, main.cpp:510)
  void to_blocks(const double *a, int dim, double *slab) const {
    const int nx = nbx * BS;
    for (int b = 0; b < nblocks; b++)
      for (int c = 0; c < BC; c++)
        for (int d = 0; d < dim; d++)
          slab[((size_t)b * BC + c) * dim + d] = a[((size_t)(by[b] * BS + c / BS) * nx + bx[b] * BS + c % BS) * dim + d];
  }
};

// dump() of main.cpp:3367-3466 for a uniform level-0 grid: float32 corners and (u, v, 0) per cell, blocks in device
// order, in the reference's operation order (origin + h * index in double, then rounded)
void dump(const std::string &path, double time, const Grid &g, const double *vel_slab, double h) {
  const size_t ncell = (size_t)g.nblocks * BC;
  std::vector<float> xyz(ncell * 8), attr(ncell * 3);
  for (int b = 0; b < g.nblocks; b++) {
    const double ox = (double)g.bx[b] * BS * h, oy = (double)g.by[b] * BS * h;
    for (int c = 0; c < BC; c++) {
      const double u0 = ox + h * (double)(c % BS), v0 = oy + h * (double)(c / BS), u1 = u0 + h, v1 = v0 + h;
      float *p = &xyz[((size_t)b * BC + c) * 8];
      p[0] = (float)u0; p[1] = (float)v0; p[2] = (float)u0; p[3] = (float)v1;
      p[4] = (float)u1; p[5] = (float)v1; p[6] = (float)u1; p[7] = (float)v0;
      float *q = &attr[((size_t)b * BC + c) * 3];
      q[0] = (float)vel_slab[((size_t)b * BC + c) * 2];
      q[1] = (float)vel_slab[((size_t)b * BC + c) * 2 + 1];
      q[2] = 0.0f;
    }
  }
  const std::string xyz_path = path + ".xyz.raw", attr_path = path + ".attr.raw";
  const auto base = [](const std::string &s) { const size_t k = s.find_last_of('/'); return k == std::string::npos ? s : s.substr(k + 1); };
  const auto put = [](const std::string &file, const void *data, size_t bytes) {
    FILE *f = std::fopen(file.c_str(), "wb");
    if (!f || std::fwrite(data, 1, bytes, f) != bytes) { std::fprintf(stderr, "cup2d_run: cannot write %s\n", file.c_str()); std::exit(1); }
    std::fclose(f);
  };
  put(xyz_path, xyz.data(), xyz.size() * sizeof(float));
  put(attr_path, attr.data(), attr.size() * sizeof(float));
  FILE *f = std::fopen((path + ".xdmf2").c_str(), "w");
  if (!f) { std::fprintf(stderr, "cup2d_run: cannot write %s.xdmf2\n", path.c_str()); std::exit(1); }
  std::fprintf(f,
               "<Xdmf\n    Version=\"2.0\">\n  <Domain>\n    <Grid>\n      <Time Value=\"%.16e\"/>\n      <Topology\n"
               "          Dimensions=\"%zu\"\n          TopologyType=\"Quadrilateral\"/>\n     <Geometry\n"
               "         GeometryType=\"XY\">\n       <DataItem\n           Dimensions=\"%zu 2\"\n"
               "           Format=\"Binary\">\n         %s\n       </DataItem>\n     </Geometry>\n       <Attribute\n"
               "           AttributeType=\"Vector\"\n           Name=\"vort\"\n           Center=\"Cell\">\n"
               "         <DataItem\n             Dimensions=\"3 %zu\"\n             Format=\"Binary\">\n           %s\n"
               "         </DataItem>\n       </Attribute>\n    </Grid>\n  </Domain>\n</Xdmf>\n",
               time, ncell, 4 * ncell, base(xyz_path).c_str(), ncell, base(attr_path).c_str());
  std::fclose(f);
}

}  // namespace

int main(int argc, char **argv) {
  int nx = 256, ny = 0, steps = 10, max_restarts = 0, max_iter = 1000, every = 0, device = 0;
  double nu = 1e-3, cfl = 0.5, tol = 1e-3, tol_rel = 1e-2;
  std::string init, prefix;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char *v = argv[i + 1];
    if (k == "-n") nx = std::atoi(v);
    else if (k == "-ny") ny = std::atoi(v);
    else if (k == "-steps") steps = std::atoi(v);
    else if (k == "-nu") nu = std::atof(v);
    else if (k == "-cfl") cfl = std::atof(v);
    else if (k == "-poissonTol") tol = std::atof(v);
    else if (k == "-poissonTolRel") tol_rel = std::atof(v);
    else if (k == "-maxPoissonRestarts") max_restarts = std::atoi(v);
    else if (k == "-maxiter") max_iter = std::atoi(v);
    else if (k == "-init") init = v;
    else if (k == "-dump") prefix = v;
    else if (k == "-every") every = std::atoi(v);
    else if (k == "-device") device = std::atoi(v);
    else { std::fprintf(stderr, "cup2d_run: unknown option %s\n", k.c_str()); return 2; }
  }
  if (ny == 0) ny = nx;
  if (nx < BS || ny < BS || nx % BS || ny % BS || steps < 0) { std::fprintf(stderr, "cup2d_run: -n / -ny must be positive multiples of 8\n"); return 2; }
  const Grid g(nx / BS, ny / BS);
  const double h = 1.0 / std::max(nx, ny);
  const size_t ncell = (size_t)nx * ny;

  std::vector<double> vel(ncell * 2);
  if (!init.empty()) {
    FILE *f = std::fopen(init.c_str(), "rb");
    if (!f || std::fread(vel.data(), sizeof(double), vel.size(), f) != vel.size()) { std::fprintf(stderr, "cup2d_run: cannot read %zu doubles from %s\n", vel.size(), init.c_str()); return 1; }
    std::fclose(f);
  } else {
    const double pi2 = 2.0 * M_PI;
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        const double x = (i + 0.5) * h, y = (j + 0.5) * h;
        vel[((size_t)j * nx + i) * 2] = std::sin(pi2 * x) * std::cos(pi2 * y);
        vel[((size_t)j * nx + i) * 2 + 1] = -std::cos(pi2 * x) * std::sin(pi2 * y);
      }
  }
  std::vector<double> slab(ncell * 2);
  g.to_blocks(vel.data(), 2, slab.data());

  cup2d_ctx *ctx = nullptr;
  RUN(cup2d_create(&ctx, g.nblocks, 0, g.nblocks, g.nbr.data(), h, device));
  RUN(cup2d_upload_slab(ctx, CUP2D_VEL, slab.data()));
  double time = 0.0;
  const auto maybe_dump = [&](int step) {
    if (prefix.empty() || (every > 0 ? step % every != 0 : step != steps)) return;
    RUN(cup2d_download_slab(ctx, CUP2D_VEL, slab.data()));
    char tag[32];
    std::snprintf(tag, sizeof tag, ".%08d", step);
    dump(prefix + tag, time, g, slab.data(), h);
  };
  maybe_dump(0);
  for (int step = 0; step < steps; step++) {
    const bool early = step < 10;  // main.cpp:7028-7030
    double dt = 0, err = 0;
    int iters = 0;
    RUN(cup2d_step(ctx, nu, cfl, early ? 0.0 : tol, early ? 0.0 : tol_rel, early ? 100 : max_restarts, max_iter, &dt, &iters, &err));
    time += dt;
    std::printf("step %d time %.17g dt %.17g poisson_iters %d poisson_err %.6e\n", step + 1, time, dt, iters, err);
    maybe_dump(step + 1);
  }
  double umax = 0;
  RUN(cup2d_max_abs_vel(ctx, &umax));
  std::printf("done: %d steps, %zu cells, max|u| %.17g\n", steps, ncell, umax);
  cup2d_destroy(ctx);
  return 0;
}
