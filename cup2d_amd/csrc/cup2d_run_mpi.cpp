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
//   mpiexec -n 8 cup2d_run_mpi -levelMax 8 -levelStart 4 [-Rtol 2] [-Ctol 0.5] [-AdaptSteps 20] ...      block-AMR on N ranks:
// BASELINE.json configs[4].  Every rank owns a contiguous range of the Hilbert-ordered leaf list (main.cpp:6494-6504) plus
// ghost copies of the remote blocks its kernels read (two rings, whole blocks through the halo plan); adapt() runs per
// rank: tags all-gathered (8 bytes per block), states validated on the replicated leaf list, the new list cut into ranges
// again, and every rank fetches from the old owners exactly what its new range is made of -- migrating blocks and the
// neighbourhoods of its refined / compressed ones (cup2d_amr_regrid_local) -- main.cpp:5055-5424.  The Python mirror is
// cup2d_amd/amr_dist.py.  -state then writes <prefix>.blocks.i32 (rank 0: the leaf list), <prefix>.<rank>.vel.f64 / .pres.f64
// (the rank's blocks, [n][64][dim]) and <prefix>.meta (the ranges).
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
    // the halo set is made of whole aligned patches of g x g blocks (cup2d_amd/grid.py BlockGrid has the reasons and the
    // measurement: the 16-block tiles of the Krylov sweeps stay 4 x 4 patches, the WENO walk finds its 2 x 2 quads)
    int g = 1;
    if (side_on[0] || side_on[1] || side_on[2] || side_on[3])
      for (int cand : {16, 4})
        if (nbx % cand == 0 && nby % cand == 0 && std::min(nbx, nby) >= 4 * cand) { g = cand; break; }
    for (int y = 0; y < nby; y++)
      for (int x = 0; x < nbx; x++) {
        const int cxg = x / g, cyg = y / g;
        const bool t = (cxg == 0 && side_on[0]) || (cxg == nbx / g - 1 && side_on[1]) || (cyg == 0 && side_on[2]) || (cyg == nby / g - 1 && side_on[3]);
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

// who exchanges which entries of the halo plan with whom (cup2d_comm_init's arguments)
struct Links {
  std::vector<int32_t> peer, soff, roff, scnt, rcnt;
  size_t nsend = 0, nrecv = 0;
};
static Links links_of(const Patch &p) {
  Links l;
  l.peer = p.peer; l.soff = p.soff; l.roff = p.roff; l.scnt = p.cnt; l.rcnt = p.cnt;
  l.nsend = p.send_block.size(); l.nrecv = p.recv_block.size();
  return l;
}

// -comm mpi: the three callbacks of cup2d_set_comm, host-staged
struct MpiTransport {
  const Links *P = nullptr;
  const Links *cells = nullptr;  // [3]: the cell plans of an adapted grid (strip_doubles < 0: CUP2D_CELL_STRIP), or none
  double *d_send = nullptr, *d_recv = nullptr, *d_red = nullptr;  // device buffers the library packs into / unpacks from
  double *h_send = nullptr, *h_recv = nullptr, *h_red = nullptr;  // pinned
  std::vector<MPI_Request> req;
  double *pending_dst = nullptr;
  size_t pending_doubles = 0;
  static constexpr int MAX_STRIP = CUP2D_MAX_STRIP_DOUBLES;  // include/cup2d_hip.h cup2d_set_comm
  void release() {
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    if (d_red) (void)hipFree(d_red);
    if (h_send) (void)hipHostFree(h_send);
    if (h_recv) (void)hipHostFree(h_recv);
    if (h_red) (void)hipHostFree(h_red);
    d_send = d_recv = d_red = h_send = h_recv = h_red = nullptr;
  }
  void init(const Links &p) {
    release();
    P = &p;
    const size_t ns = std::max<size_t>(1, p.nsend) * MAX_STRIP, nr = std::max<size_t>(1, p.nrecv) * MAX_STRIP;
    HIP(hipMalloc((void **)&d_send, ns * sizeof(double)));
    HIP(hipMalloc((void **)&d_recv, nr * sizeof(double)));
    HIP(hipMalloc((void **)&d_red, 8 * sizeof(double)));
    HIP(hipHostMalloc((void **)&h_send, ns * sizeof(double), hipHostMallocDefault));
    HIP(hipHostMalloc((void **)&h_recv, nr * sizeof(double), hipHostMallocDefault));
    HIP(hipHostMalloc((void **)&h_red, 8 * sizeof(double), hipHostMallocDefault));
  }
  static int exchange(void *user, double *dsend, double *drecv, int sd, void *stream) {
    MpiTransport &T = *static_cast<MpiTransport *>(user);
    if (sd < 0 && !T.cells) return -1;
    const Links &p = sd < 0 ? T.cells[CUP2D_CELL_STRIP_SET(sd)] : *T.P;  // a cell plan: its own offsets and counts, unit = one cell
    if (sd < 0) sd = CUP2D_CELL_STRIP_DIM(sd);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p.nsend && hipMemcpyAsync(T.h_send, dsend, p.nsend * sd * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    T.req.clear();
    for (size_t k = 0; k < p.peer.size(); k++) {
      MPI_Request r;
      if (!p.rcnt[k]) continue;
      MPI_Irecv(T.h_recv + (size_t)p.roff[k] * sd, p.rcnt[k] * sd, MPI_DOUBLE, p.peer[k], 4711, MPI_COMM_WORLD, &r);
      T.req.push_back(r);
    }
    for (size_t k = 0; k < p.peer.size(); k++) {
      MPI_Request r;
      if (!p.scnt[k]) continue;
      MPI_Isend(T.h_send + (size_t)p.soff[k] * sd, p.scnt[k] * sd, MPI_DOUBLE, p.peer[k], 4711, MPI_COMM_WORLD, &r);
      T.req.push_back(r);
    }
    T.pending_dst = drecv;
    T.pending_doubles = p.nrecv * (size_t)sd;
    return 0;
  }
  static int wait(void *user, void *stream) {
    MpiTransport &T = *static_cast<MpiTransport *>(user);
    if (!T.req.empty() && MPI_Waitall((int)T.req.size(), T.req.data(), MPI_STATUSES_IGNORE) != MPI_SUCCESS) return -1;
    T.req.clear();
    if (T.pending_dst && T.pending_doubles &&
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

// ---- block-AMR on N ranks ------------------------------------------------------------------------------------------
// One rank's share of an adapted grid (the C++ form of cup2d_amd/amr_dist.py AmrPartition): a contiguous range of the
// Hilbert-ordered leaf list, ghost copies of every remote block its kernels read -- the face neighbours of the owned
// blocks and THEIR face neighbours (the tangential neighbours across a coarser block; the coarse-fine Poisson rows stay
// inside the first ring) --, the topology tables of owned + ghost blocks in local numbering, and who sends which whole
// blocks to whom.  Every rank derives every rank's lists from the replicated leaf list: no negotiation round, as the
// reference's Setup() derives its plan from the tree (main.cpp:909-1380).
struct AmrPart {
  int nranks = 1, rank = 0, nb = 0, lo = 0, hi = 0, nowned = 0, nghost = 0;
  std::vector<long long> bounds;
  std::vector<int32_t> ghost_ids;                       // global ids, ascending
  std::vector<int32_t> level, kind, nbr2, half, nbr;    // local tables: [nowned + nghost] (nbr: owned blocks only)
  std::vector<int32_t> send_block, recv_block, zeros_s, zeros_r, gather;
  Links links;
  // cell plans (include/cup2d_hip.h cup2d_halo_plan_cells): of the blocks above, the cells the peer's kernels read -- per
  // operator family the send / receive cell lists and their per-peer offsets and counts (the unit of a message is one cell)
  bool strips = true;
  std::vector<int32_t> send_cell[3], recv_cell[3];
  Links cell_links[3];
  int owner_of(int g) const { return (int)(std::upper_bound(bounds.begin(), bounds.end(), (long long)g) - bounds.begin()) - 1; }
  static std::vector<long long> ranges(long long n, int nranks) {  // equally filled contiguous ranges (blocks per rank)
    std::vector<long long> b(nranks + 1);
    for (int r = 0; r <= nranks; r++) b[r] = n * r / nranks;
    return b;
  }
  void build(int nb_, const int32_t *blocks, const int32_t *gkind, const int32_t *gnbr2, const int32_t *ghalf, int nranks_, int rank_) {
    nb = nb_; nranks = nranks_; rank = rank_;
    if (const char *e = std::getenv("CUP2D_AMR_STRIPS")) strips = std::atoi(e) != 0;  // 0: whole ghost blocks (the block plan alone)
    bounds = ranges(nb, nranks);
    lo = (int)bounds[rank]; hi = (int)bounds[rank + 1]; nowned = hi - lo;
    std::vector<int> stamp(nb, -1);
    const auto neighbours = [&](int b, std::vector<int32_t> &out, int mark) {
      for (int s = 0; s < 4; s++) {
        const int k = gkind[4 * b + s];
        if (k == CUP2D_AMR_WALL) continue;
        const int n0 = gnbr2[(4 * b + s) * 2], n1 = k == CUP2D_AMR_FINER ? gnbr2[(4 * b + s) * 2 + 1] : -1;
        for (int n : {n0, n1})
          if (n >= 0 && stamp[n] != mark) { stamp[n] = mark; out.push_back(n); }
      }
    };
    const auto ghosts_of = [&](int r) {
      const int rlo = (int)bounds[r], rhi = (int)bounds[r + 1];
      std::vector<int32_t> ring1, ring2, g;
      for (int b = rlo; b < rhi; b++) neighbours(b, ring1, 2 * r);
      ring2 = ring1;                                       // (the stamp keeps ring1's members out of the second pass)
      for (int32_t b : ring1) neighbours(b, ring2, 2 * r);
      for (int32_t b : ring2)
        if (b < rlo || b >= rhi) g.push_back(b);
      std::sort(g.begin(), g.end());
      g.erase(std::unique(g.begin(), g.end()), g.end());
      return g;
    };
    std::vector<std::vector<int32_t>> ghosts(nranks);
    for (int r = 0; r < nranks; r++) ghosts[r] = ghosts_of(r);
    ghost_ids = ghosts[rank];
    nghost = (int)ghost_ids.size();
    const int nl = nowned + nghost;
    std::vector<int32_t> local_of(nb, -1);
    for (int i = 0; i < nowned; i++) local_of[lo + i] = i;
    for (int i = 0; i < nghost; i++) local_of[ghost_ids[i]] = nowned + i;
    level.resize(nl); kind.resize(4 * nl); nbr2.resize(8 * nl); half.resize(4 * nl); nbr.assign((size_t)4 * nowned, CUP2D_WALL);
    for (int i = 0; i < nl; i++) {
      const int g = i < nowned ? lo + i : ghost_ids[i - nowned];
      level[i] = blocks[3 * g];
      for (int s = 0; s < 4; s++) {
        int k = gkind[4 * g + s];
        const int n0 = gnbr2[(4 * g + s) * 2], n1 = gnbr2[(4 * g + s) * 2 + 1];
        int h0 = n0 >= 0 ? local_of[n0] : -1, h1 = n1 >= 0 ? local_of[n1] : -1, hf = ghalf[4 * g + s];
        const bool missing = k != CUP2D_AMR_WALL && (h0 < 0 || (k == CUP2D_AMR_FINER && h1 < 0));
        if (missing) {  // a side whose neighbour this rank does not hold is never read: a wall
          if (i < nowned) { std::fprintf(stderr, "cup2d_run_mpi[%d]: ghost closure does not cover block %d side %d\n", rank, g, s); MPI_Abort(MPI_COMM_WORLD, 1); }
          k = CUP2D_AMR_WALL; h0 = h1 = -1; hf = 0;
        }
        if (k != CUP2D_AMR_FINER) h1 = -1;
        kind[4 * i + s] = k; nbr2[(4 * i + s) * 2] = h0; nbr2[(4 * i + s) * 2 + 1] = h1; half[4 * i + s] = hf;
        if (i < nowned && k == CUP2D_AMR_SAME) nbr[4 * i + s] = h0;
      }
    }
    // peers ascending, blocks ascending on both ends of a link
    links = Links();
    send_block.clear(); recv_block.clear();
    for (int p = 0; p < nranks; p++) {
      if (p == rank) continue;
      std::vector<int32_t> out, inn;
      for (int32_t g : ghosts[p]) if (g >= lo && g < hi) out.push_back(local_of[g]);
      for (int32_t g : ghost_ids) if (g >= bounds[p] && g < bounds[p + 1]) inn.push_back(local_of[g]);
      if (out.empty() && inn.empty()) continue;
      links.peer.push_back(p); links.soff.push_back((int32_t)send_block.size()); links.roff.push_back((int32_t)recv_block.size());
      links.scnt.push_back((int32_t)out.size()); links.rcnt.push_back((int32_t)inn.size());
      send_block.insert(send_block.end(), out.begin(), out.end());
      recv_block.insert(recv_block.end(), inn.begin(), inn.end());
    }
    links.nsend = send_block.size(); links.nrecv = recv_block.size();
    for (int i = 0; i < nghost; i++)
      if (recv_block[i] != nowned + i) { std::fprintf(stderr, "cup2d_run_mpi[%d]: ghosts are not numbered in receive order\n", rank); MPI_Abort(MPI_COMM_WORLD, 1); }
    zeros_s.assign(std::max<size_t>(1, send_block.size()), 0);
    zeros_r.assign(std::max<size_t>(1, recv_block.size()), 0);
    gather.resize(send_block.size() * BC);
    for (size_t k = 0; k < send_block.size(); k++)
      for (int c = 0; c < BC; c++) gather[k * BC + c] = send_block[k] * BC + c;
    for (int set = 0; set < 3; set++) { send_cell[set].clear(); recv_cell[set].clear(); cell_links[set] = Links(); }
    if (!strips || links.peer.empty()) return;
    // Link (reader rank a, owner rank b): the cells of b's blocks that a's blocks read, in (global block, cell) order on both
    // ends -- the receiver traces its own blocks that some peer holds as ghosts, the sender its ghost copies of the receiver's
    // blocks (the same readers as far as the link goes): cup2d_amr_trace_reads runs the kernels' own ghost expressions
    std::vector<int32_t> readers_me;
    for (int p = 0; p < nranks; p++)
      if (p != rank)
        for (int32_t g : ghosts[p]) if (g >= lo && g < hi) readers_me.push_back(g);
    std::sort(readers_me.begin(), readers_me.end());
    readers_me.erase(std::unique(readers_me.begin(), readers_me.end()), readers_me.end());
    std::vector<uint64_t> mask_me(nb), mask_q(nb);
    for (int set = 0; set < 3; set++) {
      std::fill(mask_me.begin(), mask_me.end(), 0);
      RUN(cup2d_amr_trace_reads(nb, gkind, gnbr2, ghalf, (int)readers_me.size(), readers_me.data(), set, mask_me.data()));
      Links &cl = cell_links[set];
      for (size_t k = 0; k < links.peer.size(); k++) {
        const int q = links.peer[k];
        std::vector<int32_t> g_in;
        for (int32_t g : ghost_ids) if (g >= bounds[q] && g < bounds[q + 1]) g_in.push_back(g);
        cl.peer.push_back(q); cl.soff.push_back((int32_t)send_cell[set].size()); cl.roff.push_back((int32_t)recv_cell[set].size());
        for (int32_t g : g_in)
          for (int c = 0; c < BC; c++)
            if (mask_me[g] >> c & 1) recv_cell[set].push_back(local_of[g] * BC + c);
        std::fill(mask_q.begin(), mask_q.end(), 0);
        RUN(cup2d_amr_trace_reads(nb, gkind, gnbr2, ghalf, (int)g_in.size(), g_in.data(), set, mask_q.data()));
        for (int b = lo; b < hi; b++)
          for (int c = 0; c < BC; c++)
            if (mask_q[b] >> c & 1) send_cell[set].push_back((b - lo) * BC + c);
        cl.scnt.push_back((int32_t)send_cell[set].size() - cl.soff.back());
        cl.rcnt.push_back((int32_t)recv_cell[set].size() - cl.roff.back());
      }
      cl.nsend = send_cell[set].size(); cl.nrecv = recv_cell[set].size();
    }
    gather = send_cell[CUP2D_CELLS_MATRIX];
  }
};

struct AmrRunMpi {
  std::vector<int32_t> blocks;  // the GLOBAL leaf list [nb][3], on every rank (12 bytes per block)
  AmrPart P;
  cup2d_ctx *ctx = nullptr;
  MpiTransport T;
  int world = 1, device = 0, math = CUP2D_MATH_FAST;
  bool rccl = false;
  long long moved_blocks = 0, fetched_blocks = 0;  // regrid statistics, summed over the run (this rank)
  int nb() const { return (int)blocks.size() / 3; }
  static constexpr double H0 = 1.0 / BS;  // one level-0 block of 8 cells spans the unit square (main.cpp:6338)

  // this rank's context on the current leaf list; keep_old: the previous context is returned instead of destroyed (adapt()
  // copies the blocks that stay on the rank across on the device)
  cup2d_ctx *build(bool keep_old) {
    const int n = nb();
    std::vector<int32_t> gk(4 * n), gn(8 * n), gh(4 * n);
    RUN(cup2d_amr_tables(n, blocks.data(), 1, 1, gk.data(), gn.data(), gh.data()));
    cup2d_ctx *old = ctx;
    if (old && rccl) RUN(cup2d_comm_finalize(old));
    if (old && !keep_old) { cup2d_destroy(old); old = nullptr; }
    ctx = nullptr;
    P.build(n, blocks.data(), gk.data(), gn.data(), gh.data(), world, g_rank);
    RUN(cup2d_create(&ctx, P.nowned, P.nghost, P.nowned, P.nbr.data(), H0, device));
    RUN(cup2d_halo_plan(ctx, (int)P.send_block.size(), P.send_block.data(), P.zeros_s.data(), (int)P.recv_block.size(), P.recv_block.data(),
                        P.zeros_r.data()));
    const bool cells = P.strips && !P.links.peer.empty();
    if (cells)
      for (int set = 0; set < 3; set++)
        RUN(cup2d_halo_plan_cells(ctx, set, (int)P.send_cell[set].size(), P.send_cell[set].data(), (int)P.recv_cell[set].size(),
                                  P.recv_cell[set].data()));
    if (rccl) {
      char token[CUP2D_COMM_ID_BYTES];
      if (g_rank == 0) RUN(cup2d_comm_unique_id(token));
      MPI_Bcast(token, sizeof token, MPI_BYTE, 0, MPI_COMM_WORLD);
      RUN(cup2d_comm_init(ctx, world, g_rank, token, (int)P.links.peer.size(), P.links.peer.data(), P.links.soff.data(), P.links.roff.data(),
                          P.links.scnt.data(), P.links.rcnt.data()));
      if (cells)
        for (int set = 0; set < 3; set++) {
          const Links &cl = P.cell_links[set];
          RUN(cup2d_comm_set_cell_counts(ctx, set, (int)cl.peer.size(), cl.soff.data(), cl.scnt.data(), cl.roff.data(), cl.rcnt.data()));
        }
      RUN(cup2d_comm_selftest(ctx, 20.0, nullptr, 0));
    } else {
      T.init(P.links);
      T.cells = cells ? P.cell_links : nullptr;
      RUN(cup2d_set_comm(ctx, &MpiTransport::exchange, &MpiTransport::wait, &MpiTransport::allreduce, &T, T.d_send, T.d_recv, T.d_red));
      RUN(cup2d_set_comm_strip_capacity(ctx, MpiTransport::MAX_STRIP));
    }
    RUN(cup2d_set_amr(ctx, H0, P.level.data(), P.kind.data(), P.nbr2.data(), P.half.data()));
    int lmax = 0;
    for (int b = 0; b < n; b++) lmax = std::max(lmax, (int)blocks[3 * b]);
    RUN(cup2d_amr_set_finest_level(ctx, lmax));
    RUN(cup2d_set_math(ctx, math));
    RUN(cup2d_amr_install_poisson(ctx));  // this rank's rows of main.cpp:7034-7113, ghost cells as halo columns
    RUN(cup2d_set_gather(ctx, (int)P.gather.size(), P.gather.data()));
    return old;
  }

  static constexpr int NF = 5, UNIT = 64 * (1 + 2 + 2 + 1 + 1);  // a migrating block: chi, vel, vold, pres, pold = 448 doubles
  // [n][UNIT]: the five fields of the listed owned blocks (local indices), field after field
  void download_units(const std::vector<int32_t> &local, double *out) {
    static const int fields[NF] = {CUP2D_CHI, CUP2D_VEL, CUP2D_VOLD, CUP2D_PRES, CUP2D_POLD};
    static const int dims[NF] = {1, 2, 2, 1, 1};
    const int n = (int)local.size();
    std::vector<double> buf;
    int o = 0;
    for (int f = 0; f < NF; f++) {
      const int w = BC * dims[f];
      buf.resize((size_t)std::max(n, 1) * w);
      if (n) RUN(cup2d_download_blocks(ctx, fields[f], n, local.data(), buf.data()));
      for (int k = 0; k < n; k++) std::copy(buf.begin() + (size_t)k * w, buf.begin() + (size_t)(k + 1) * w, out + (size_t)k * UNIT + o);
      o += w;
    }
  }

  // adapt() of main.cpp:4657-5440 on N ranks (cup2d_amd/amr_dist.py adapt / fetch_new_range): true if the grid changed
  bool adapt(double rtol, double ctol, int level_max) {
    const int n = nb();
    RUN(cup2d_vorticity(ctx, CUP2D_BLOCKS_ALL));
    std::vector<double> mine(std::max(P.nowned, 1)), linf(n);
    RUN(cup2d_block_linf(ctx, CUP2D_TMP, mine.data()));
    std::vector<int> cnt(world), dsp(world);
    for (int r = 0; r < world; r++) { cnt[r] = (int)(P.bounds[r + 1] - P.bounds[r]); dsp[r] = (int)P.bounds[r]; }
    MPI_Allgatherv(mine.data(), P.nowned, MPI_DOUBLE, linf.data(), cnt.data(), dsp.data(), MPI_DOUBLE, MPI_COMM_WORLD);
    std::vector<int32_t> st(n);
    for (int b = 0; b < n; b++) {  // main.cpp:4678-4690
      const int l = blocks[3 * b];
      st[b] = linf[b] > rtol ? 1 : linf[b] < ctol ? 2 : 0;
      if ((st[b] == 1 && l == level_max - 1) || (st[b] == 2 && l == 0)) st[b] = 0;
    }
    RUN(cup2d_amr_validate_states(n, blocks.data(), 1, 1, level_max, st.data()));
    bool any = false;
    for (int b = 0; b < n; b++) any = any || st[b] != 0;
    if (!any) return false;
    // ---- the plan of this rank's new range ----
    const auto fail = [&](const char *what) { std::fprintf(stderr, "cup2d_run_mpi[%d]: %s: %s\n", g_rank, what, cup2d_last_error()); MPI_Abort(MPI_COMM_WORLD, 1); };
    const long long n2 = cup2d_amr_regrid_local(n, blocks.data(), 1, 1, level_max, st.data(), 0, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                                nullptr, nullptr, nullptr);
    if (n2 < 0) fail("amr_regrid_local (count)");
    const std::vector<long long> nbounds = AmrPart::ranges(n2, world);
    const long long nlo = nbounds[g_rank], nhi = nbounds[g_rank + 1];
    const int nmine = (int)(nhi - nlo);
    std::vector<int32_t> nblocks2((size_t)3 * n2), src(n2), needed(n);
    if (cup2d_amr_regrid_local(n, blocks.data(), 1, 1, level_max, st.data(), nlo, nhi, n2, nblocks2.data(), src.data(), needed.data(), 0, nullptr,
                               nullptr, nullptr, nullptr, nullptr) != n2)
      fail("amr_regrid_local (plan)");
    // old blocks (global ids) this rank reads: what its changed blocks are computed from + the unchanged copies of its range
    std::vector<char> want(n, 0);
    for (int b = 0; b < n; b++) want[b] = needed[b] != 0;
    for (long long p = nlo; p < nhi; p++)
      if (src[p] >= 0) want[src[p]] = 1;
    std::vector<int32_t> remote;
    for (int b = 0; b < n; b++)
      if (want[b] && (b < P.lo || b >= P.hi)) remote.push_back(b);
    // ---- who needs what from whom: the request lists travel (ids only), then whole blocks ----
    std::vector<int> rc(world), rd(world);
    const int nreq = (int)remote.size();
    MPI_Allgather(&nreq, 1, MPI_INT, rc.data(), 1, MPI_INT, MPI_COMM_WORLD);
    int tot = 0;
    for (int r = 0; r < world; r++) { rd[r] = tot; tot += rc[r]; }
    std::vector<int32_t> allreq(std::max(tot, 1));
    MPI_Allgatherv(remote.data(), nreq, MPI_INT32_T, allreq.data(), rc.data(), rd.data(), MPI_INT32_T, MPI_COMM_WORLD);
    std::vector<std::vector<int32_t>> send_local(world), recv_ids(world);
    for (int p = 0; p < world; p++) {
      if (p == g_rank) continue;
      for (int k = 0; k < rc[p]; k++) {
        const int g = allreq[rd[p] + k];
        if (g >= P.lo && g < P.hi) send_local[p].push_back(g - P.lo);
      }
    }
    for (int32_t g : remote) recv_ids[P.owner_of(g)].push_back(g);
    std::vector<std::vector<double>> sbuf(world), rbuf(world);
    std::vector<MPI_Request> req;
    for (int p = 0; p < world; p++) {
      if (!recv_ids[p].empty()) {
        rbuf[p].resize(recv_ids[p].size() * (size_t)UNIT);
        req.emplace_back();
        MPI_Irecv(rbuf[p].data(), (int)rbuf[p].size(), MPI_DOUBLE, p, 5120, MPI_COMM_WORLD, &req.back());
      }
    }
    for (int p = 0; p < world; p++) {
      if (send_local[p].empty()) continue;
      sbuf[p].resize(send_local[p].size() * (size_t)UNIT);
      download_units(send_local[p], sbuf[p].data());
      req.emplace_back();
      MPI_Isend(sbuf[p].data(), (int)sbuf[p].size(), MPI_DOUBLE, p, 5120, MPI_COMM_WORLD, &req.back());
      moved_blocks += (long long)send_local[p].size();
    }
    // own blocks the changed ones are computed from come to the host; own blocks that are only copied stay on the device
    std::vector<int32_t> local_need;
    for (int b = P.lo; b < P.hi; b++)
      if (needed[b]) local_need.push_back(b - P.lo);
    size_t nheld = local_need.size();
    for (int p = 0; p < world; p++) nheld += recv_ids[p].size();
    std::vector<double> units(std::max<size_t>(nheld, 1) * UNIT);
    download_units(local_need, units.data());
    if (!req.empty()) MPI_Waitall((int)req.size(), req.data(), MPI_STATUSES_IGNORE);
    std::vector<int32_t> slot(n, -1);
    size_t at = 0;
    for (int32_t l : local_need) slot[P.lo + l] = (int32_t)at++;
    for (int p = 0; p < world; p++) {
      std::copy(rbuf[p].begin(), rbuf[p].end(), units.begin() + at * UNIT);
      for (int32_t g : recv_ids[p]) slot[g] = (int32_t)at++;
    }
    fetched_blocks += (long long)nheld;
    // compact per-field arrays (block k of field f at comp[f] + slot[k] * 64 * dim)
    static const int fields[NF] = {CUP2D_CHI, CUP2D_VEL, CUP2D_VOLD, CUP2D_PRES, CUP2D_POLD};
    static const int32_t dims[NF] = {1, 2, 2, 1, 1}, vec[NF] = {0, 1, 1, 0, 0};
    std::vector<std::vector<double>> comp(NF), out(NF);
    const double *sp[NF];
    double *dp[NF];
    int o = 0;
    for (int f = 0; f < NF; f++) {
      const int w = BC * dims[f];
      comp[f].resize(std::max<size_t>(nheld, 1) * w);
      for (size_t k = 0; k < nheld; k++) std::copy(units.begin() + k * UNIT + o, units.begin() + k * UNIT + o + w, comp[f].begin() + k * w);
      out[f].assign((size_t)std::max(nmine, 1) * w, 0.0);
      sp[f] = comp[f].data();
      dp[f] = out[f].data();
      o += w;
    }
    if (cup2d_amr_regrid_local(n, blocks.data(), 1, 1, level_max, st.data(), nlo, nhi, n2, nblocks2.data(), nullptr, nullptr, NF, sp, slot.data(), dims, vec,
                               dp) != n2)
      fail("amr_regrid_local (compute)");
    // ---- the new context; what stays on this rank unchanged moves on the device ----
    const int old_lo = P.lo, old_hi = P.hi;
    blocks.swap(nblocks2);
    cup2d_ctx *old_ctx = build(true);
    std::vector<int32_t> dst_dev, src_dev, up_idx;
    for (int q = 0; q < nmine; q++) {
      const int sg = src[nlo + q];
      if (sg >= old_lo && sg < old_hi) { dst_dev.push_back(q); src_dev.push_back(sg - old_lo); }
      else up_idx.push_back(q);
    }
    std::vector<double> rows;
    for (int f = 0; f < NF; f++) {
      const int w = BC * dims[f];
      if (!dst_dev.empty()) RUN(cup2d_copy_blocks(ctx, old_ctx, fields[f], (int)dst_dev.size(), dst_dev.data(), src_dev.data()));
      if (up_idx.empty()) continue;
      rows.resize(up_idx.size() * (size_t)w);
      for (size_t k = 0; k < up_idx.size(); k++) {
        const int q = up_idx[k], sg = src[nlo + q];
        const double *from = sg >= 0 ? comp[f].data() + (size_t)slot[sg] * w   // an unchanged block that migrated here
                                     : out[f].data() + (size_t)q * w;          // prolonged / restricted here
        std::copy(from, from + w, rows.begin() + k * w);
      }
      RUN(cup2d_upload_blocks(ctx, fields[f], (int)up_idx.size(), up_idx.data(), rows.data()));
    }
    cup2d_destroy(old_ctx);
    return true;
  }
};

int run_amr_mpi(int world, int level_start, int level_max, double rtol, double ctol, int steps, double nu, double cfl, double tol, double tol_rel,
                int max_restarts, int max_iter, const std::string &state, int device, int math, int adapt_steps, bool rccl) {
  AmrRunMpi R;
  R.world = world; R.device = device; R.math = math; R.rccl = rccl;
  const int n0 = 1 << level_start;
  std::vector<std::pair<uint64_t, int>> key;
  for (int j = 0; j < n0; j++)
    for (int i = 0; i < n0; i++) key.push_back({hilbert(std::max(level_start, 1), i, j), j * n0 + i});
  std::sort(key.begin(), key.end());  // the start grid along the Hilbert curve: contiguous ranges are compact patches
  for (const auto &k : key) {
    R.blocks.push_back(level_start); R.blocks.push_back(k.second % n0); R.blocks.push_back(k.second / n0);
  }
  R.build(false);
  {  // two Gaussian vortices (cup2d_run's start field), this rank's blocks
    const AmrPart &P = R.P;
    std::vector<double> vel((size_t)std::max(P.nowned, 1) * BC * 2);
    const double h = 1.0 / (BS << level_start);
    for (int b = 0; b < P.nowned; b++)
      for (int c = 0; c < BC; c++) {
        const int g = P.lo + b;
        const double x = (R.blocks[3 * g + 1] * BS + c % BS + 0.5) * h, y = (R.blocks[3 * g + 2] * BS + c / BS + 0.5) * h;
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
    std::vector<int32_t> all(P.nowned);
    for (int b = 0; b < P.nowned; b++) all[b] = b;
    if (P.nowned) RUN(cup2d_upload_blocks(R.ctx, CUP2D_VEL, P.nowned, all.data(), vel.data()));
  }
  double time = 0.0;
  for (int step = 0; step < steps; step++) {
    double dt = 0, err = 0;
    int iters = 0;
    RUN(cup2d_compute_dt(R.ctx, nu, cfl, &dt));  // before the regrid, as main.cpp:6579-6603 orders them; reduced over the ranks
    if (!(dt > 2e-16)) {
      if (g_rank == 0) std::printf("step %d: dt %.3e <= 2e-16, nothing to advance\n", step + 1, dt);
      break;
    }
    if (step <= 10 || step % adapt_steps == 0) R.adapt(rtol, ctol, level_max);  // main.cpp:6603
    const bool early = step < 10;
    RUN(cup2d_advect_diffuse_rk2(R.ctx, nu, dt));
    RUN(cup2d_poisson_rhs(R.ctx, dt, 0));
    RUN(cup2d_poisson_solve(R.ctx, early ? 0.0 : tol, early ? 0.0 : tol_rel, early ? 100 : max_restarts, max_iter, &iters, nullptr, &err, nullptr));
    RUN(cup2d_project(R.ctx, dt));
    time += dt;
    if (g_rank == 0)
      std::printf("step %d time %.17g dt %.17g poisson_iters %d poisson_err %.6e blocks %d\n", step + 1, time, dt, iters, err, R.nb());
  }
  long long moved[2] = {R.moved_blocks, R.fetched_blocks}, tot[2] = {0, 0};
  MPI_Reduce(moved, tot, 2, MPI_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
  if (!state.empty()) {
    const auto put = [](const std::string &file, const void *data, size_t bytes) {
      FILE *f = std::fopen(file.c_str(), "wb");
      if (!f || std::fwrite(data, 1, bytes, f) != bytes) { std::fprintf(stderr, "cup2d_run_mpi: cannot write %s\n", file.c_str()); MPI_Abort(MPI_COMM_WORLD, 1); }
      std::fclose(f);
    };
    const AmrPart &P = R.P;
    std::vector<int32_t> all(P.nowned);
    for (int b = 0; b < P.nowned; b++) all[b] = b;
    std::vector<double> v((size_t)std::max(P.nowned, 1) * BC * 2), p((size_t)std::max(P.nowned, 1) * BC);
    if (P.nowned) {
      RUN(cup2d_download_blocks(R.ctx, CUP2D_VEL, P.nowned, all.data(), v.data()));
      RUN(cup2d_download_blocks(R.ctx, CUP2D_PRES, P.nowned, all.data(), p.data()));
    }
    put(state + "." + std::to_string(g_rank) + ".vel.f64", v.data(), (size_t)P.nowned * BC * 2 * sizeof(double));
    put(state + "." + std::to_string(g_rank) + ".pres.f64", p.data(), (size_t)P.nowned * BC * sizeof(double));
    if (g_rank == 0) {
      put(state + ".blocks.i32", R.blocks.data(), R.blocks.size() * sizeof(int32_t));
      std::string m;
      for (long long b : P.bounds) m += std::to_string(b) + " ";
      m += "\n";
      put(state + ".meta", m.data(), m.size());
    }
  }
  if (g_rank == 0)
    std::printf("done: %d steps, %d blocks on %d ranks; regrids moved %lld blocks between ranks, %lld blocks passed through host memory\n", steps,
                R.nb(), world, tot[0], tot[1]);
  if (rccl) RUN(cup2d_comm_finalize(R.ctx));
  cup2d_destroy(R.ctx);
  R.T.release();
  return 0;
}

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
  int level_max = 0, level_start = 2, adapt_steps = 20;
  double nu = 1e-3, cfl = 0.5, tol = 1e-3, tol_rel = 1e-2, rtol = 2.0, ctol = 0.5;
  std::string comm = "rccl", state, plan_blocks;
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
    else if (k == "-planOnly") plan_blocks = v;
    else if (k == "-levelMax") level_max = std::atoi(v);
    else if (k == "-levelStart") level_start = std::atoi(v);
    else if (k == "-AdaptSteps") adapt_steps = std::atoi(v);
    else if (k == "-Rtol") rtol = std::atof(v);
    else if (k == "-Ctol") ctol = std::atof(v);
    else { if (g_rank == 0) std::fprintf(stderr, "cup2d_run_mpi: unknown option %s\n", k.c_str()); MPI_Finalize(); return 2; }
  }
  if (ny == 0) ny = nx;
  if (px == 0 || py == 0) cartesian_dims(world, px, py);
  if ((comm != "rccl" && comm != "mpi") ||
      (level_max == 0 && plan_blocks.empty() && (px * py != world || nx % (BS * px) || ny % (BS * py) || nx < BS * px || ny < BS * py))) {
    if (g_rank == 0) std::fprintf(stderr, "cup2d_run_mpi: %d ranks need px * py = %d, -n / -ny multiples of 8 px / 8 py, -comm rccl|mpi\n", world, world);
    MPI_Finalize();
    return 2;
  }
  if (!plan_blocks.empty()) {
    // -planOnly <blocks.i32> -state <prefix>: this rank's share of the given leaf list (AmrPart) written out, no GPU touched --
    // how the CPU tests compare the C++ plan with cup2d_amd/amr_dist.py AmrPartition
    FILE *f = std::fopen(plan_blocks.c_str(), "rb");
    std::vector<int32_t> bl;
    int32_t v3[3];
    while (f && std::fread(v3, sizeof(int32_t), 3, f) == 3) bl.insert(bl.end(), v3, v3 + 3);
    if (f) std::fclose(f);
    const int n = (int)bl.size() / 3;
    if (n < world || state.empty()) { if (g_rank == 0) std::fprintf(stderr, "cup2d_run_mpi: -planOnly needs a leaf list with at least one block per rank and -state\n"); MPI_Finalize(); return 2; }
    std::vector<int32_t> gk(4 * n), gn(8 * n), gh(4 * n);
    RUN(cup2d_amr_tables(n, bl.data(), 1, 1, gk.data(), gn.data(), gh.data()));
    AmrPart P;
    P.build(n, bl.data(), gk.data(), gn.data(), gh.data(), world, g_rank);
    const std::string base = state + "." + std::to_string(g_rank) + ".";
    const auto put = [&](const char *name, const std::vector<int32_t> &a) {
      FILE *o = std::fopen((base + name).c_str(), "wb");
      if (!o || (a.size() && std::fwrite(a.data(), sizeof(int32_t), a.size(), o) != a.size())) { std::fprintf(stderr, "cup2d_run_mpi: cannot write %s%s\n", base.c_str(), name); MPI_Abort(MPI_COMM_WORLD, 1); }
      std::fclose(o);
    };
    put("range", {P.lo, P.hi, P.nghost});
    put("ghost_ids", P.ghost_ids); put("level", P.level); put("kind", P.kind); put("nbr2", P.nbr2); put("half", P.half); put("nbr", P.nbr);
    put("send_block", P.send_block); put("recv_block", P.recv_block); put("gather", P.gather);
    std::vector<int32_t> lk;
    for (size_t k = 0; k < P.links.peer.size(); k++)
      for (int32_t x : {P.links.peer[k], P.links.soff[k], P.links.roff[k], P.links.scnt[k], P.links.rcnt[k]}) lk.push_back(x);
    put("links", lk);
    for (int set = 0; set < 3; set++) {  // the cell plans (empty lists with CUP2D_AMR_STRIPS=0)
      const std::string k = std::to_string(set);
      put(("send_cell" + k).c_str(), P.send_cell[set]); put(("recv_cell" + k).c_str(), P.recv_cell[set]);
      const Links &cl = P.cell_links[set];
      std::vector<int32_t> ck;
      for (size_t i = 0; i < cl.peer.size(); i++)
        for (int32_t x : {cl.peer[i], cl.soff[i], cl.roff[i], cl.scnt[i], cl.rcnt[i]}) ck.push_back(x);
      put(("cell_links" + k).c_str(), ck);
    }
    MPI_Finalize();
    return 0;
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
  if (level_max > 0) {
    if (level_start < 0 || level_start >= level_max || level_max > 16 || adapt_steps < 1 || (1 << (2 * level_start)) < world) {
      if (g_rank == 0) std::fprintf(stderr, "cup2d_run_mpi: 0 <= -levelStart < -levelMax <= 16, -AdaptSteps >= 1 and at least one start block per rank expected\n");
      MPI_Finalize();
      return 2;
    }
    const int rc = run_amr_mpi(world, level_start, level_max, rtol, ctol, steps, nu, cfl, tol, tol_rel, max_restarts, max_iter, state, device, math,
                               adapt_steps, comm == "rccl");
    MPI_Finalize();
    return rc;
  }

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
  const Links L = links_of(P);
  if (comm == "rccl") {
    char token[CUP2D_COMM_ID_BYTES];
    if (g_rank == 0) RUN(cup2d_comm_unique_id(token));
    MPI_Bcast(token, sizeof token, MPI_BYTE, 0, MPI_COMM_WORLD);
    RUN(cup2d_comm_init(ctx, world, g_rank, token, (int)P.peer.size(), P.peer.data(), P.soff.data(), P.roff.data(), P.cnt.data(), nullptr));
    char report[512] = "";
    RUN(cup2d_comm_selftest(ctx, 20.0, report, (int)sizeof report));  // strips between all peers + reductions, checked, 20 s deadline
    if (g_rank == 0) fprintf(stderr, "cup2d_run_mpi: communicator ok: %s\n", report);
  } else {
    T.init(L);
    RUN(cup2d_set_comm(ctx, &MpiTransport::exchange, &MpiTransport::wait, &MpiTransport::allreduce, &T, T.d_send, T.d_recv, T.d_red));
      RUN(cup2d_set_comm_strip_capacity(ctx, MpiTransport::MAX_STRIP));
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
