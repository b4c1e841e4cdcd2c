// cup2d_run_mpi.cpp -- the uniform-grid time loop on N ranks with the host side in C++: one MPI rank per GPU, a px x py
// Cartesian split of the block grid (BASELINE.json configs[3]: 8192^2 over 2 x 4), every block operator, halo exchange and
// reduction inside libcup2d_hip.so.  What replaces the reference's rank partitioning + Synchronizer + MPI_Allreduce sites
// (main.cpp:6494-6504, 1971-2142, 6583-6592, 7138, 7162; cuda.cu:365-380, 445-534) for same-level grids, as compiled host
// code (the Python mirror of the same plan is cup2d_amd/distributed.py, which bench.py and the tests drive).
//
//   mpiexec -n 8 cup2d_run_mpi -n 8192 [-ny 8192] [-px 2 -py 4] [-steps 10] [-nu 1e-3] [-cfl 0.5] [-maxiter 1000]
//           [-poissonTol 1e-3] [-poissonTolRel 1e-2] [-maxPoissonRestarts 0] [-comm rccl|mpi] [-math fast|strict]
//           [-state prefix]
//
// -comm rccl (default): the communicator inside the library (csrc/comm.hip: ncclSend / ncclRecv on a second HIP stream,
//            all-gather reductions); MPI only carries the 256-byte token.  One GPU per rank (device = node-local rank).
// -comm mpi: the callback interface (cup2d_set_comm) with host-staged MPI_Isend / MPI_Irecv / MPI_Allreduce: the
//            transport for boxes where RCCL cannot run, e.g. two ranks sharing one GPU -- how the tests run this program.
// Grid: nx x ny cells globally, rank (cx, cy) owns an (nx/px) x (ny/py) patch of 8 x 8 blocks -- blocks that touch no other
// rank first, each group along the Hilbert curve --, one ghost block per boundary block on every interior side, strips
// grouped by peer in side order W, E, S, N; both ends of a link enumerate a side's positions in the same order.
// Initial velocity: the Taylor-Green vortex of cup2d_run.  -state writes <prefix>.<rank>.vel.f64 / .pres.f64, the rank's patch
// row-major, and rank 0 <prefix>.meta (nx ny px py).  No kernels and no CPU fallback here.
#include <hip/hip_runtime_api.h>
#include <mpi.h>

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
int g_rank = 0;

#define RUN(expr)                                                                                              \
  do {                                                                                                         \
    const int rc_ = (expr);                                                                                    \
    if (rc_ != CUP2D_OK) {                                                                                     \
      std::fprintf(stderr, "cup2d_run_mpi[%d]: %s -> %d: %s\n", g_rank, #expr, rc_, cup2d_last_error());       \
      MPI_Abort(MPI_COMM_WORLD, 1);                                                                            \
    }                                                                                                          \
  } while (0)
#define HIP(expr)                                                                                              \
  do {                                                                                                         \
    const hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                                    \
      std::fprintf(stderr, "cup2d_run_mpi[%d]: %s -> %s\n", g_rank, #expr, hipGetErrorString(e_));             \
      MPI_Abort(MPI_COMM_WORLD, 1);                                                                            \
    }                                                                                                          \
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

// one rank's patch: device block order, neighbour table with ghost ids, the halo plan and its peers
struct Patch {
  int nbx, nby, nblocks, n_inner = 0, nghost = 0;
  std::vector<int> bx, by;
  std::vector<int32_t> nbr, send_block, send_face, recv_block, recv_face;
  std::vector<int32_t> peer, soff, roff, cnt;
  Patch(int nbx_, int nby_, int px, int py, int cx, int cy) : nbx(nbx_), nby(nby_), nblocks(nbx_ * nby_) {
    const bool side_on[4] = {cx > 0, cx < px - 1, cy > 0, cy < py - 1};
    int bits = 1;
    while ((1 << bits) < std::max(std::max(nbx, nby), 2)) bits++;
    struct Key { int halo; uint64_t h; int cell; };
    std::vector<Key> key(nblocks);
    for (int y = 0; y < nby; y++)
      for (int x = 0; x < nbx; x++) {
        const bool t = (x == 0 && side_on[0]) || (x == nbx - 1 && side_on[1]) || (y == 0 && side_on[2]) || (y == nby - 1 && side_on[3]);
        key[y * nbx + x] = {t ? 1 : 0, hilbert(bits, x, y), y * nbx + x};
        n_inner += t ? 0 : 1;
      }
    std::stable_sort(key.begin(), key.end(), [](const Key &a, const Key &b) { return a.halo != b.halo ? a.halo < b.halo : a.h < b.h; });
    bx.resize(nblocks);
    by.resize(nblocks);
    std::vector<int> index_of(nblocks);
    for (int b = 0; b < nblocks; b++) {
      bx[b] = key[b].cell % nbx;
      by[b] = key[b].cell / nbx;
      index_of[key[b].cell] = b;
    }
    // ghost blocks: one per boundary block on each interior side, numbered after the owned blocks in (side, position) order
    int ghost0[4] = {-1, -1, -1, -1};
    for (int s = 0; s < 4; s++)
      if (side_on[s]) {
        ghost0[s] = nblocks + nghost;
        nghost += s < 2 ? nby : nbx;
      }
    nbr.assign((size_t)4 * nblocks, CUP2D_WALL);
    for (int b = 0; b < nblocks; b++) {
      const int x = bx[b], y = by[b];
      nbr[4 * b + 0] = x > 0 ? index_of[y * nbx + x - 1] : side_on[0] ? ghost0[0] + y : CUP2D_WALL;
      nbr[4 * b + 1] = x < nbx - 1 ? index_of[y * nbx + x + 1] : side_on[1] ? ghost0[1] + y : CUP2D_WALL;
      nbr[4 * b + 2] = y > 0 ? index_of[(y - 1) * nbx + x] : side_on[2] ? ghost0[2] + x : CUP2D_WALL;
      nbr[4 * b + 3] = y < nby - 1 ? index_of[(y + 1) * nbx + x] : side_on[3] ? ghost0[3] + x : CUP2D_WALL;
    }
    const int rank = cy * px + cx;
    const int peer_of_side[4] = {rank - 1, rank + 1, rank - px, rank + px};
    const int opposite[4] = {1, 0, 3, 2};
    for (int s = 0; s < 4; s++) {
      if (!side_on[s]) continue;
      const int npos = s < 2 ? nby : nbx;
      peer.push_back(peer_of_side[s]);
      soff.push_back((int32_t)send_block.size());
      roff.push_back((int32_t)recv_block.size());
      cnt.push_back(npos);
      for (int pos = 0; pos < npos; pos++) {
        const int owned = s == 0 ? index_of[pos * nbx] : s == 1 ? index_of[pos * nbx + nbx - 1] : s == 2 ? index_of[pos] : index_of[(nby - 1) * nbx + pos];
        send_block.push_back(owned);
        send_face.push_back(s);  // my blocks' face on that side
        recv_block.push_back(ghost0[s] + pos);
        recv_face.push_back(opposite[s]);  // the peer's face that touches me
      }
    }
  }
  void to_blocks(const double *a, int dim, double *slab) const {  // patch row-major [ny][nx][dim] -> [nblocks][64][dim]
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

// -comm mpi: the three callbacks of cup2d_set_comm, host-staged
struct MpiTransport {
  const Patch *P = nullptr;
  double *d_send = nullptr, *d_recv = nullptr, *d_red = nullptr;  // device buffers the library packs into / unpacks from
  double *h_send = nullptr, *h_recv = nullptr, *h_red = nullptr;  // pinned
  std::vector<MPI_Request> req;
  double *pending_dst = nullptr;
  size_t pending_doubles = 0;
  static constexpr int MAX_STRIP = 128;  // include/cup2d_hip.h cup2d_set_comm
  void init(const Patch &p) {
    P = &p;
    const size_t ns = std::max<size_t>(1, p.send_block.size()) * MAX_STRIP, nr = std::max<size_t>(1, p.recv_block.size()) * MAX_STRIP;
    HIP(hipMalloc((void **)&d_send, ns * sizeof(double)));
    HIP(hipMalloc((void **)&d_recv, nr * sizeof(double)));
    HIP(hipMalloc((void **)&d_red, 8 * sizeof(double)));
    HIP(hipHostMalloc((void **)&h_send, ns * sizeof(double), hipHostMallocDefault));
    HIP(hipHostMalloc((void **)&h_recv, nr * sizeof(double), hipHostMallocDefault));
    HIP(hipHostMalloc((void **)&h_red, 8 * sizeof(double), hipHostMallocDefault));
  }
  static int exchange(void *user, double *dsend, double *drecv, int sd, void *stream) {
    MpiTransport &T = *static_cast<MpiTransport *>(user);
    const Patch &p = *T.P;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(T.h_send, dsend, p.send_block.size() * sd * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    T.req.clear();
    for (size_t k = 0; k < p.peer.size(); k++) {
      MPI_Request r;
      MPI_Irecv(T.h_recv + (size_t)p.roff[k] * sd, p.cnt[k] * sd, MPI_DOUBLE, p.peer[k], 4711, MPI_COMM_WORLD, &r);
      T.req.push_back(r);
    }
    for (size_t k = 0; k < p.peer.size(); k++) {
      MPI_Request r;
      MPI_Isend(T.h_send + (size_t)p.soff[k] * sd, p.cnt[k] * sd, MPI_DOUBLE, p.peer[k], 4711, MPI_COMM_WORLD, &r);
      T.req.push_back(r);
    }
    T.pending_dst = drecv;
    T.pending_doubles = p.recv_block.size() * (size_t)sd;
    return 0;
  }
  static int wait(void *user, void *stream) {
    MpiTransport &T = *static_cast<MpiTransport *>(user);
    if (!T.req.empty() && MPI_Waitall((int)T.req.size(), T.req.data(), MPI_STATUSES_IGNORE) != MPI_SUCCESS) return -1;
    T.req.clear();
    if (T.pending_dst &&
        hipMemcpyAsync(T.pending_dst, T.h_recv, T.pending_doubles * sizeof(double), hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)) != hipSuccess)
      return -1;
    T.pending_dst = nullptr;
    return 0;
  }
  static int allreduce(void *user, double *buf, int count, int op, void *stream) {
    MpiTransport &T = *static_cast<MpiTransport *>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(T.h_red, buf, count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (MPI_Allreduce(MPI_IN_PLACE, T.h_red, count, MPI_DOUBLE, op == 0 ? MPI_SUM : MPI_MAX, MPI_COMM_WORLD) != MPI_SUCCESS) return -1;
    return hipMemcpyAsync(buf, T.h_red, count * sizeof(double), hipMemcpyHostToDevice, st) == hipSuccess ? 0 : -1;
  }
};

void cartesian_dims(int world, int &px, int &py) {  // 8 -> 2 x 4, as cup2d_amd/distributed.py
  px = 1;
  while (px * px * 4 <= world && world % (px * 2) == 0) px *= 2;
  if (world % px) px = 1;
  py = world / px;
}

}  // namespace

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int world = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &g_rank);
  MPI_Comm_size(MPI_COMM_WORLD, &world);
  int nx = 256, ny = 0, px = 0, py = 0, steps = 10, max_restarts = 0, max_iter = 1000, math = CUP2D_MATH_FAST;
  double nu = 1e-3, cfl = 0.5, tol = 1e-3, tol_rel = 1e-2;
  std::string comm = "rccl", state;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char *v = argv[i + 1];
    if (k == "-n") nx = std::atoi(v);
    else if (k == "-ny") ny = std::atoi(v);
    else if (k == "-px") px = std::atoi(v);
    else if (k == "-py") py = std::atoi(v);
    else if (k == "-steps") steps = std::atoi(v);
    else if (k == "-nu") nu = std::atof(v);
    else if (k == "-cfl") cfl = std::atof(v);
    else if (k == "-poissonTol") tol = std::atof(v);
    else if (k == "-poissonTolRel") tol_rel = std::atof(v);
    else if (k == "-maxPoissonRestarts") max_restarts = std::atoi(v);
    else if (k == "-maxiter") max_iter = std::atoi(v);
    else if (k == "-comm") comm = v;
    else if (k == "-state") state = v;
    else if (k == "-math") math = std::strcmp(v, "strict") == 0 ? CUP2D_MATH_STRICT : CUP2D_MATH_FAST;
    else { if (g_rank == 0) std::fprintf(stderr, "cup2d_run_mpi: unknown option %s\n", k.c_str()); MPI_Finalize(); return 2; }
  }
  if (ny == 0) ny = nx;
  if (px == 0 || py == 0) cartesian_dims(world, px, py);
  if (px * py != world || nx % (BS * px) || ny % (BS * py) || nx < BS * px || ny < BS * py || (comm != "rccl" && comm != "mpi")) {
    if (g_rank == 0) std::fprintf(stderr, "cup2d_run_mpi: %d ranks need px * py = %d, -n / -ny multiples of 8 px / 8 py, -comm rccl|mpi\n", world, world);
    MPI_Finalize();
    return 2;
  }
  // one GPU per rank: the node-local rank picks the device (ranks share GPUs only under -comm mpi)
  MPI_Comm node;
  MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, g_rank, MPI_INFO_NULL, &node);
  int local = 0, ngpu = 0;
  MPI_Comm_rank(node, &local);
  HIP(hipGetDeviceCount(&ngpu));
  if (ngpu < 1 || (comm == "rccl" && local >= ngpu)) {
    std::fprintf(stderr, "cup2d_run_mpi[%d]: node-local rank %d but %d GPU(s) visible (-comm rccl needs one GPU per rank)\n", g_rank, local, ngpu);
    MPI_Abort(MPI_COMM_WORLD, 2);
  }
  const int device = local % ngpu;
  HIP(hipSetDevice(device));

  const int cx = g_rank % px, cy = g_rank / px;
  const int pnx = nx / px, pny = ny / py;
  const Patch P(pnx / BS, pny / BS, px, py, cx, cy);
  const double h = 1.0 / std::max(nx, ny);
  std::vector<double> vel((size_t)pnx * pny * 2), slab(vel.size());
  const double pi2 = 2.0 * M_PI;
  for (int j = 0; j < pny; j++)
    for (int i = 0; i < pnx; i++) {
      const double x = (cx * pnx + i + 0.5) * h, y = (cy * pny + j + 0.5) * h;
      vel[((size_t)j * pnx + i) * 2] = std::sin(pi2 * x) * std::cos(pi2 * y);
      vel[((size_t)j * pnx + i) * 2 + 1] = -std::cos(pi2 * x) * std::sin(pi2 * y);
    }
  P.to_blocks(vel.data(), 2, slab.data());

  cup2d_ctx *ctx = nullptr;
  RUN(cup2d_create(&ctx, P.nblocks, P.nghost, P.n_inner, P.nbr.data(), h, device));
  RUN(cup2d_halo_plan(ctx, (int)P.send_block.size(), P.send_block.data(), P.send_face.data(), (int)P.recv_block.size(),
                      P.recv_block.data(), P.recv_face.data()));
  MpiTransport T;
  if (comm == "rccl") {
    char token[CUP2D_COMM_ID_BYTES];
    if (g_rank == 0) RUN(cup2d_comm_unique_id(token));
    MPI_Bcast(token, sizeof token, MPI_BYTE, 0, MPI_COMM_WORLD);
    RUN(cup2d_comm_init(ctx, world, g_rank, token, (int)P.peer.size(), P.peer.data(), P.soff.data(), P.roff.data(), P.cnt.data(), nullptr));
    char report[512] = "";
    RUN(cup2d_comm_selftest(ctx, 20.0, report, (int)sizeof report));  // strips between all peers + reductions, checked, 20 s deadline
    if (g_rank == 0) fprintf(stderr, "cup2d_run_mpi: communicator ok: %s\n", report);
  } else {
    T.init(P);
    RUN(cup2d_set_comm(ctx, &MpiTransport::exchange, &MpiTransport::wait, &MpiTransport::allreduce, &T, T.d_send, T.d_recv, T.d_red));
  }
  RUN(cup2d_set_math(ctx, math));
  RUN(cup2d_upload_slab(ctx, CUP2D_VEL, slab.data()));
  double time = 0.0;
  MPI_Barrier(MPI_COMM_WORLD);
  const double t0 = MPI_Wtime();
  for (int step = 0; step < steps; step++) {
    const bool early = step < 10;  // main.cpp:7028-7030
    double dt = 0, err = 0;
    int iters = 0;
    RUN(cup2d_step(ctx, nu, cfl, early ? 0.0 : tol, early ? 0.0 : tol_rel, early ? 100 : max_restarts, max_iter, &dt, &iters, &err));
    time += dt;
    if (g_rank == 0) std::printf("step %d time %.17g dt %.17g poisson_iters %d poisson_err %.6e\n", step + 1, time, dt, iters, err);
  }
  RUN(cup2d_synchronize(ctx));
  MPI_Barrier(MPI_COMM_WORLD);
  const double wall = MPI_Wtime() - t0;
  double umax = 0;
  RUN(cup2d_max_abs_vel(ctx, &umax));  // reduced over the ranks by the library
  if (g_rank == 0)
    std::printf("done: %d steps on %d ranks (%d x %d, comm %s), %zu cells, max|u| %.17g, %.3f ms per step, %.1f Mcell-updates/s\n", steps, world,
                px, py, comm.c_str(), (size_t)nx * ny, umax, steps ? 1e3 * wall / steps : 0.0, steps ? (double)nx * ny * steps / wall / 1e6 : 0.0);
  if (!state.empty()) {
    const auto put = [&](const std::string &file, const void *data, size_t bytes) {
      FILE *f = std::fopen(file.c_str(), "wb");
      if (!f || std::fwrite(data, 1, bytes, f) != bytes) { std::fprintf(stderr, "cup2d_run_mpi: cannot write %s\n", file.c_str()); MPI_Abort(MPI_COMM_WORLD, 1); }
      std::fclose(f);
    };
    RUN(cup2d_download_slab(ctx, CUP2D_VEL, slab.data()));
    P.from_blocks(slab.data(), 2, vel.data());
    put(state + "." + std::to_string(g_rank) + ".vel.f64", vel.data(), vel.size() * sizeof(double));
    std::vector<double> ps((size_t)pnx * pny), pp(ps.size());
    RUN(cup2d_download_slab(ctx, CUP2D_PRES, ps.data()));
    P.from_blocks(ps.data(), 1, pp.data());
    put(state + "." + std::to_string(g_rank) + ".pres.f64", pp.data(), pp.size() * sizeof(double));
    if (g_rank == 0) {
      const std::string m = std::to_string(nx) + " " + std::to_string(ny) + " " + std::to_string(px) + " " + std::to_string(py) + "\n";
      put(state + ".meta", m.data(), m.size());
    }
  }
  if (comm == "rccl") RUN(cup2d_comm_finalize(ctx));
  cup2d_destroy(ctx);
  MPI_Finalize();
  return 0;
}
