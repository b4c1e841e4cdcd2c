// cup2d_run.cpp -- a C++ host driver over the C ABI (include/cup2d_hip.h): the reference's time loop on a uniform grid
// with every block operator on the GPU.  What a CUP2D maintainer's main() looks like once the call sites of
// main.cpp:6606-6642, 7007-7027, 7115-7187 forward to the library (seam B2) -- here as a stand-alone
// program so that the sequence exists as compiled host code and not only as the Python mirror the tests use.
//
//   cup2d_run -n 256 [-ny 256] [-steps 10] [-nu 1e-3] [-cfl 0.5] [-poissonTol 1e-3] [-poissonTolRel 1e-2]
//             [-maxPoissonRestarts 0] [-maxiter 1000] [-init vel.f64] [-dump prefix] [-every k] [-device 0]
//             [-math fast|strict] [-state prefix]      (-state: <prefix>.vel.f64 [ny][nx][2], <prefix>.pres.f64 [ny][nx] at the end)
//   cup2d_run -levelMax 5 -levelStart 2 [-Rtol 2] [-Ctol 0.5] ... [-state prefix]      block-AMR (below)
//
// Grid: nx x ny cells in 8 x 8 blocks, ordered along the Hilbert curve like the reference's (main.cpp:347-360,
// 1550-1562), walls on all four sides, h = extent / max(nx, ny) (main.cpp:6338, extent 1).  Initial velocity: the file
// given with -init (float64, row-major [ny][nx][2]) or the Taylor-Green vortex.  The first ten steps solve with zero
// tolerances and 100 restarts, later ones with the given tolerances (main.cpp:7028-7030).  -dump writes the
// reference's dump() files (main.cpp:3367-3466: <prefix>.<step>.xyz.raw / .attr.raw / .xdmf2) every k-th step.
// With -levelMax the grid is the reference's block-AMR grid (one level-0 block as base, uniform at -levelStart to begin
// with): every step computes dt, regrids (adapt(), main.cpp:4657-5440: vorticity and its per-block max on the GPU,
// states validated by the library's host routines, fields prolonged / restricted by kernels between the old and the new
// context, the Poisson rows re-assembled on the new leaves) and advances (main.cpp:6579-7187).  -init is then a block-ordered slab
// [nblocks][64][2] of the start grid (blocks row-major), default two Gaussian vortices; -state writes
// <prefix>.blocks.i32, .vel.f64, .pres.f64 at the end.
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
  void from_blocks(const double *slab, int dim, double *a) const {
    const int nx = nbx * BS;
    for (int b = 0; b < nblocks; b++)
      for (int c = 0; c < BC; c++)
        for (int d = 0; d < dim; d++)
          a[((size_t)(by[b] * BS + c / BS) * nx + bx[b] * BS + c % BS) * dim + d] = slab[((size_t)b * BC + c) * dim + d];
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

// ---- block-AMR run -----------------------------------------------------------------------------------------------
struct AmrRun {
  std::vector<int32_t> blocks;  // [nb][3] level, i, j
  cup2d_ctx *ctx = nullptr;
  int device = 0, math = CUP2D_MATH_FAST;
  int nb() const { return (int)blocks.size() / 3; }
  // context + topology tables + Poisson operator of the current leaves (what the reference redoes after every regrid,
  // main.cpp:7034-7115)
  void build() {
    const int n = nb();
    std::vector<int32_t> kind(4 * n), nbr2(8 * n), half(4 * n), nbr(4 * n), level(n);
    RUN(cup2d_amr_tables(n, blocks.data(), 1, 1, kind.data(), nbr2.data(), half.data()));
    for (int q = 0; q < 4 * n; q++) nbr[q] = kind[q] == CUP2D_AMR_SAME ? nbr2[2 * q] : CUP2D_WALL;
    for (int b = 0; b < n; b++) level[b] = blocks[3 * b];
    const double h0 = 1.0 / BS;  // one level-0 block of 8 cells spans the unit square (main.cpp:6338)
    if (ctx) cup2d_destroy(ctx);
    ctx = nullptr;
    RUN(cup2d_create(&ctx, n, 0, n, nbr.data(), h0, device));
    RUN(cup2d_set_amr(ctx, h0, level.data(), kind.data(), nbr2.data(), half.data()));
    RUN(cup2d_set_math(ctx, math));
    RUN(cup2d_amr_install_poisson(ctx));  // the coarse-fine operator of main.cpp:7034-7113, from the tables just set
  }
  // adapt(): returns true if the grid changed
  bool adapt(double rtol, double ctol, int level_max) {
    const int n = nb();
    RUN(cup2d_vorticity(ctx, CUP2D_BLOCKS_ALL));
    std::vector<double> linf(n);
    RUN(cup2d_block_linf(ctx, CUP2D_TMP, linf.data()));
    std::vector<int32_t> st(n);
    bool any = false;
    for (int b = 0; b < n; b++) {  // main.cpp:4678-4690
      const int l = blocks[3 * b];
      st[b] = linf[b] > rtol ? 1 : linf[b] < ctol ? 2 : 0;
      if ((st[b] == 1 && l == level_max - 1) || (st[b] == 2 && l == 0)) st[b] = 0;
    }
    RUN(cup2d_amr_validate_states(n, blocks.data(), 1, 1, level_max, st.data()));
    for (int b = 0; b < n; b++) any = any || st[b] != 0;
    if (!any) return false;
    // the fields stay on the device: the plan gives the new leaf list, a context is built on it, and one kernel per field
    // copies / restricts / prolongs between the old and the new slabs (cup2d_amr_regrid_device; main.cpp:4981-5032, 5149-5166)
    static const int32_t fields[5] = {CUP2D_CHI, CUP2D_VEL, CUP2D_VOLD, CUP2D_PRES, CUP2D_POLD};
    const long long n2 = cup2d_amr_regrid_plan(n, blocks.data(), 1, 1, level_max, st.data(), 0, nullptr, nullptr, nullptr);
    if (n2 < 0) { std::fprintf(stderr, "cup2d_run: amr_regrid_plan: %s\n", cup2d_last_error()); std::exit(1); }
    std::vector<int32_t> nblocks2((size_t)3 * n2);
    if (cup2d_amr_regrid_plan(n, blocks.data(), 1, 1, level_max, st.data(), n2, nblocks2.data(), nullptr, nullptr) != n2) {
      std::fprintf(stderr, "cup2d_run: amr_regrid_plan: %s\n", cup2d_last_error());
      std::exit(1);
    }
    cup2d_ctx *old = ctx;
    std::vector<int32_t> old_blocks;
    old_blocks.swap(blocks);
    blocks.swap(nblocks2);
    ctx = nullptr;
    build();
    RUN(cup2d_amr_regrid_device(ctx, old, n, old_blocks.data(), 1, 1, level_max, st.data(), 5, fields));
    cup2d_destroy(old);
    return true;
  }
};

int run_amr(int level_start, int level_max, double rtol, double ctol, int steps, double nu, double cfl, double tol, double tol_rel,
            int max_restarts, int max_iter, const std::string &init, const std::string &state, int device, int math, int adapt_steps) {
  AmrRun R;
  R.device = device;
  R.math = math;
  const int n0 = 1 << level_start;
  for (int j = 0; j < n0; j++)
    for (int i = 0; i < n0; i++) {
      R.blocks.push_back(level_start);
      R.blocks.push_back(i);
      R.blocks.push_back(j);
    }
  R.build();
  std::vector<double> vel((size_t)R.nb() * BC * 2);
  if (!init.empty()) {
    FILE *f = std::fopen(init.c_str(), "rb");
    if (!f || std::fread(vel.data(), sizeof(double), vel.size(), f) != vel.size()) { std::fprintf(stderr, "cup2d_run: cannot read %zu doubles from %s\n", vel.size(), init.c_str()); return 1; }
    std::fclose(f);
  } else {
    const double h = 1.0 / (BS << level_start);
    for (int b = 0; b < R.nb(); b++)
      for (int c = 0; c < BC; c++) {
        const double x = (R.blocks[3 * b + 1] * BS + c % BS + 0.5) * h, y = (R.blocks[3 * b + 2] * BS + c / BS + 0.5) * h;
        double u = 0, v = 0;
        const double vort[2][3] = {{0.35, 0.5, 1.0}, {0.65, 0.5, -1.0}};
        for (const auto &w : vort) {
          const double dx = x - w[0], dy = y - w[1], fq = w[2] * std::exp(-(dx * dx + dy * dy) / (0.06 * 0.06)) / 0.06;
          u += -dy * fq;
          v += dx * fq;
        }
        vel[((size_t)b * BC + c) * 2] = u;
        vel[((size_t)b * BC + c) * 2 + 1] = v;
      }
  }
  RUN(cup2d_upload_slab(R.ctx, CUP2D_VEL, vel.data()));
  double time = 0.0;
  for (int step = 0; step < steps; step++) {
    double dt = 0, err = 0;
    int iters = 0;
    RUN(cup2d_compute_dt(R.ctx, nu, cfl, &dt));  // before the regrid, as main.cpp:6579-6603 orders them
    if (!(dt > 2e-16)) {  // main.cpp:6596 skips the body and would spin on the same state for ever: stop instead
      std::printf("step %d: dt %.3e <= 2e-16, nothing to advance\n", step + 1, dt);
      break;
    }
    if (step <= 10 || step % adapt_steps == 0) R.adapt(rtol, ctol, level_max);  // main.cpp:6603 (sim.AdaptSteps, default 20)
    const bool early = step < 10;
    RUN(cup2d_advect_diffuse_rk2(R.ctx, nu, dt));
    RUN(cup2d_poisson_rhs(R.ctx, dt, 0));
    RUN(cup2d_poisson_solve(R.ctx, early ? 0.0 : tol, early ? 0.0 : tol_rel, early ? 100 : max_restarts, max_iter, &iters, nullptr, &err, nullptr));
    RUN(cup2d_project(R.ctx, dt));
    time += dt;
    std::printf("step %d time %.17g dt %.17g poisson_iters %d poisson_err %.6e blocks %d\n", step + 1, time, dt, iters, err, R.nb());
  }
  if (!state.empty()) {
    const int n = R.nb();
    std::vector<double> v((size_t)n * BC * 2), p((size_t)n * BC);
    RUN(cup2d_download_slab(R.ctx, CUP2D_VEL, v.data()));
    RUN(cup2d_download_slab(R.ctx, CUP2D_PRES, p.data()));
    const auto put = [](const std::string &file, const void *data, size_t bytes) {
      FILE *f = std::fopen(file.c_str(), "wb");
      if (!f || std::fwrite(data, 1, bytes, f) != bytes) { std::fprintf(stderr, "cup2d_run: cannot write %s\n", file.c_str()); std::exit(1); }
      std::fclose(f);
    };
    put(state + ".blocks.i32", R.blocks.data(), R.blocks.size() * sizeof(int32_t));
    put(state + ".vel.f64", v.data(), v.size() * sizeof(double));
    put(state + ".pres.f64", p.data(), p.size() * sizeof(double));
  }
  std::printf("done: %d steps, %d blocks\n", steps, R.nb());
  cup2d_destroy(R.ctx);
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  int nx = 256, ny = 0, steps = 10, max_restarts = 0, max_iter = 1000, every = 0, device = 0, level_max = 0, level_start = 2;
  int math = CUP2D_MATH_FAST, adapt_steps = 20;  // -AdaptSteps: main.cpp:6603, run.sh passes 20
  double nu = 1e-3, cfl = 0.5, tol = 1e-3, tol_rel = 1e-2, rtol = 2.0, ctol = 0.5;
  std::string init, prefix, state;
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
    else if (k == "-levelMax") level_max = std::atoi(v);
    else if (k == "-levelStart") level_start = std::atoi(v);
    else if (k == "-AdaptSteps") adapt_steps = std::atoi(v);
    else if (k == "-Rtol") rtol = std::atof(v);
    else if (k == "-Ctol") ctol = std::atof(v);
    else if (k == "-state") state = v;
    else if (k == "-math") math = std::strcmp(v, "strict") == 0 ? CUP2D_MATH_STRICT : CUP2D_MATH_FAST;
    else { std::fprintf(stderr, "cup2d_run: unknown option %s\n", k.c_str()); return 2; }
  }
  if (level_max > 0) {
    if (level_start < 0 || level_start >= level_max || level_max > 16) { std::fprintf(stderr, "cup2d_run: 0 <= -levelStart < -levelMax <= 16 expected\n"); return 2; }
    if (adapt_steps < 1) { std::fprintf(stderr, "cup2d_run: -AdaptSteps >= 1 expected\n"); return 2; }
    return run_amr(level_start, level_max, rtol, ctol, steps, nu, cfl, tol, tol_rel, max_restarts, max_iter, init, state, device, math,
                   adapt_steps);
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
  RUN(cup2d_set_math(ctx, math));
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
  if (!state.empty()) {  // <state>.vel.f64 [ny][nx][2], <state>.pres.f64 [ny][nx], row-major (what cup2d_run_mpi writes per rank)
    const auto put = [](const std::string &file, const void *data, size_t bytes) {
      FILE *f = std::fopen(file.c_str(), "wb");
      if (!f || std::fwrite(data, 1, bytes, f) != bytes) { std::fprintf(stderr, "cup2d_run: cannot write %s\n", file.c_str()); std::exit(1); }
      std::fclose(f);
    };
    RUN(cup2d_download_slab(ctx, CUP2D_VEL, slab.data()));
    g.from_blocks(slab.data(), 2, vel.data());
    put(state + ".vel.f64", vel.data(), vel.size() * sizeof(double));
    std::vector<double> ps(ncell), pp(ncell);
    RUN(cup2d_download_slab(ctx, CUP2D_PRES, ps.data()));
    g.from_blocks(ps.data(), 1, pp.data());
    put(state + ".pres.f64", pp.data(), pp.size() * sizeof(double));
  }
  double umax = 0;
  RUN(cup2d_max_abs_vel(ctx, &umax));
  std::printf("done: %d steps, %zu cells, max|u| %.17g\n", steps, ncell, umax);
  cup2d_destroy(ctx);
  return 0;
}
