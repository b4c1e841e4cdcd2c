// smoother.hip -- the Poisson smoother sweep and the Poisson residual as tile kernels.
//
// What they compute (§8d "one weighted-Jacobi sweep", "residual r = b - A x"):
//   A        the matrix-free Poisson operator: pressure_rhs1's 5-point sum (main.cpp:6209-6230) with the
//            homogeneous-Neumann walls of the assembled matrix (main.cpp:7034-7112: a wall contributes nothing,
//            so diag(A) = -(number of neighbours of the cell));
//   MODE 0   x' = x + r * c,  r = b - A x,  c = omega / diag(A)      one weighted-Jacobi sweep, 24 B/cell
//   MODE 1   out = r = b - A x                                        the residual, 24 B/cell
//   both     max|r| per workgroup (the reference's stopping norm is Linf, cuda.cu:303-311)
// The reference has no smoother (its Poisson solve is BiCGSTAB, krylov*.hip here); the sweep is the
// "50 Jacobi pressure iters/step" unit of BASELINE.json configs[1] and the streaming 5-point kernel of the path.
//
// Organisation.  The per-block kernels of pressure.hip fetch the four ghost edges of EVERY block from global
// memory; a W/E edge is 8 doubles 64 B apart, i.e. the whole neighbour block crosses L2 -> L1 for 64 useful
// bytes.  Here a wave owns a TILE of 16 consecutive blocks (a 4 x 4 patch in the Hilbert order): the 16 blocks
// are loaded once, coalesced, into LDS; the (block, side) slots -- one per lane -- whose neighbour lies in the
// tile take their ghost edge from LDS, only the <= 16 slots on the rim of the patch gather from global.  Tiles
// are dealt contiguously per XCD (workgroup w runs on XCD w % 8).
#include <type_traits>

#include <utility>

#include "block.h"
#include "krylov_common.h"

namespace cup2d {

constexpr int JT = 16;  // blocks per tile

// LDS record of one block of the tile: its 64 cells, then the 8 ghost cells across each of its 4 sides.  Cells and
// ghosts share the block stride, so a lane's five stencil operands are five fixed offsets into the record.
constexpr int REC = BC + 4 * BS + 1;  // odd stride: the 16 records start in different LDS banks
struct SmootherLds {
  double rec[JT * REC];
};

struct SmootherTile {
  double xv[JT], bv[JT], g[BS];
  int nb;  // neighbour of this lane's slot
};

template <int MODE, bool SMALL, int NT = 0>
__global__ __launch_bounds__(WG, 3) void k_smoother(const double *__restrict__ x, const double *__restrict__ b,
                                                 double *__restrict__ out, const int *__restrict__ nbr, int first,
                                                 int count, double c4, double c3, double c2,
                                                 double *partials, unsigned *ticket, double *linf_out) {
  __shared__ SmootherLds lds[WPG];
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  SmootherLds &L = lds[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const int si = lane >> 2, ss = lane & 3;  // this lane's (block, side) slot of a tile
  const int last = first + count;
  const int ntiles = (count + JT - 1) / JT;
  int t_begin, t_end, t_stride;
  {
    const int G = gridDim.x, w = blockIdx.x;
    if (G >= 8 && (G % 8) == 0) {
      const int xcd = w & 7, slot = w >> 3, per = G >> 3;
      const long long lo = (long long)ntiles * xcd / 8, hi = (long long)ntiles * (xcd + 1) / 8;
      t_begin = (int)lo + slot * WPG + wave;
      t_end = (int)hi;
      t_stride = per * WPG;
    } else {
      t_begin = w * WPG + wave;
      t_end = ntiles;
      t_stride = G * WPG;
    }
  }
  // Tile t = blocks [tb, tb + JT).  Every tile is FULL: the last, partial one is moved back to end at `last` (it
  // overlaps its predecessor and writes only its own tail, `skip` leading blocks are recomputed and dropped), so
  // that all 16 blocks are addressed as one wave-uniform base + compile-time offsets (scalar base, immediate
  // offsets: no per-load address registers).  Launches with fewer than JT blocks take the SMALL instantiation.
  const auto tile_base = [&](int t) -> int { return SMALL ? first : min(first + t * JT, last - JT); };
  const auto load_nb = [&](int t) -> int {
    const int bb = tile_base(t) + si;
    return (t < t_end && bb < last) ? nbr[4 * bb + ss] : CUP2D_WALL;
  };
  const int nvalid = SMALL ? count : JT;  // blocks of a tile that exist
  // cell q of the neighbour's edge that touches side ss of this lane's slot: side e = ss ^ 1 of the neighbour
  const int en = ss ^ 1, gc0 = en == 1 ? BS - 1 : en == 3 ? (BS - 1) * BS : 0, gstride = en < 2 ? BS : 1;
  const auto is_rim = [&](int nb, int tb) { return nb >= 0 && (nb < tb || nb >= tb + nvalid); };
  // The requests of a tile.  All of them are unconditional, straight-line code: the hardware counts outstanding
  // loads in issue order (s_waitcnt vmcnt), and a load the compiler cannot prove was issued makes every later
  // wait conservative.  Lanes whose slot is not on the rim all read one and the same cell instead (one line).
  const auto fetch_x = [&](SmootherTile &T, int tb) {
    const double *xb = x + (size_t)tb * BC + lane;
#pragma unroll
    for (int i = 0; i < JT; i++) {
      const double *px = xb + (SMALL ? min(i, nvalid - 1) : i) * BC;
      T.xv[i] = (NT & 4) ? __builtin_nontemporal_load(px) : *px;
    }
  };
  const auto fetch_rim = [&](SmootherTile &T, int tb, int nb) {
    const bool rim = is_rim(nb, tb);  // <= 16 lanes: a W/E edge is 8 cells 64 B apart, an S/N edge 64 contiguous bytes
    const double *src = x + (rim ? (size_t)nb * BC + gc0 : (size_t)tb * BC);
    const int st = rim ? gstride : 0;
#pragma unroll
    for (int q = 0; q < BS; q++) T.g[q] = src[q * st];
  };
  double amax = 0.0;
  // offsets of this lane's stencil operands in a block record
  const int o1 = ix > 0 ? lane - 1 : BC + 0 * BS + iy, o2 = ix < BS - 1 ? lane + 1 : BC + 1 * BS + iy;
  const int o3 = iy > 0 ? lane - BS : BC + 2 * BS + ix, o4 = iy < BS - 1 ? lane + BS : BC + 3 * BS + ix;
  const int fW = ix == 0, fE = ix == BS - 1, fS = iy == 0, fN = iy == BS - 1;

  // A wave walks its tiles with ONE register set (40 doubles per lane): the requests of tile t + 1 are issued
  // while tile t is processed, each into the registers tile t has just finished with -- x right after x(t) went
  // to LDS, the rim edges after the ghost fill, b(t + 1)[i] as soon as block i has consumed b(t)[i].  A second
  // register set would cost 80 VGPRs and a third of the occupancy (3 waves per SIMD by LDS).  The moved last
  // tile recomputes blocks of its predecessor and stores the same values again.
  if (t_begin < t_end) {
    SmootherTile T;
    int nb = load_nb(t_begin), nb_next = load_nb(t_begin + t_stride);
    {  // same issue order as the steady state: x, rim edges, b
      const int tb = uniform(tile_base(t_begin));
      fetch_x(T, tb);
      fetch_rim(T, tb, nb);
      const double *bb = b + (size_t)tb * BC + lane;
#pragma unroll
      for (int i = 0; i < JT; i++) {
        const double *pb = bb + (SMALL ? min(i, nvalid - 1) : i) * BC;
        T.bv[i] = (NT & 1) ? __builtin_nontemporal_load(pb) : *pb;
      }
    }
    const auto body = [&](auto has_next, int t) {
      constexpr bool HAS_NEXT = decltype(has_next)::value;
      const int tb = uniform(tile_base(t));
      const int tbn = uniform(HAS_NEXT ? tile_base(t + t_stride) : tb);
#pragma unroll
      for (int i = 0; i < JT; i++) L.rec[i * REC + lane] = T.xv[i];
      wave_lds_sync();
      if constexpr (HAS_NEXT) fetch_x(T, tbn);
      {
        // ghost edge of this lane's slot: across a wall the block's own edge cell (ScalarLab::Neumann2D,
        // main.cpp:3210-3255), inside the tile the neighbour's opposite edge from LDS, on the rim the gathered cells
        const bool rim = is_rim(nb, tb);
        const int sblk = (nb < 0 || rim) ? min(si, nvalid - 1) : nb - tb, e = nb < 0 ? ss : ss ^ 1;
        const double *sp = L.rec + sblk * REC + (e == 1 ? BS - 1 : e == 3 ? (BS - 1) * BS : 0);
        const int stride = e < 2 ? BS : 1;
        double v[BS];
#pragma unroll
        for (int q = 0; q < BS; q++) v[q] = sp[q * stride];
        double *gp = L.rec + si * REC + BC + ss * BS;
#pragma unroll
        for (int q = 0; q < BS; q++) gp[q] = rim ? T.g[q] : v[q];
      }
      const unsigned long long wallmask = __ballot(nb < 0);
      if constexpr (HAS_NEXT) fetch_rim(T, tbn, nb_next);
      wave_lds_sync();
      double *ob = out + (size_t)tb * BC + lane;
      const double *bn = b + (size_t)tbn * BC + lane;
#pragma unroll
      for (int i = 0; i < JT; i++) {
        const double bi = T.bv[i];
        if constexpr (HAS_NEXT) T.bv[i] = (NT & 1) ? __builtin_nontemporal_load(bn + i * BC) : bn[i * BC];
        if (!SMALL || i < nvalid) {
          const double *rb = L.rec + i * REC;
          const double l0 = rb[lane], l1 = rb[o1], l2 = rb[o2], l3 = rb[o3], l4 = rb[o4];
          const double r = bi - (l1 + l2 + l3 + l4 - 4 * l0);
          amax = fmax(amax, fabs(r));
          double o = r;
          if (MODE == 0) {
            const int walls = (int)(wallmask >> (4 * i)) & 15;  // wave-uniform: walls W, E, S, N of block i
            double c = c4;
            if (walls) {  // a block at the domain boundary: cells on a wall have fewer neighbours
              const int nw = ((walls & 1) ? fW : 0) + ((walls & 2) ? fE : 0) + ((walls & 4) ? fS : 0) + ((walls & 8) ? fN : 0);
              c = nw == 0 ? c4 : nw == 1 ? c3 : c2;
            }
            o = l0 + r * c;
          }
          if (NT & 2) __builtin_nontemporal_store(o, ob + i * BC);
          else ob[i * BC] = o;
        }
      }
      wave_lds_sync();  // the next tile overwrites the records
      nb = nb_next;
      if constexpr (HAS_NEXT) nb_next = load_nb(t + 2 * t_stride);
    };
    int t = t_begin;
    for (; t + t_stride < t_end; t += t_stride) body(std::true_type{}, t);
    body(std::false_type{}, t);
  }
  // max|r|: one partial per workgroup, finished by the last workgroup to arrive (krylov_common.h)
  double acc[1] = {amax};
  workgroup_reduce_store<1, true, true>(acc, partials, 0);
  if (arrive_last(ticket)) {
    __shared__ double red[WG];
    double a = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += WG)
      a = fmax(a, __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    red[threadIdx.x] = a;
    __syncthreads();
    for (int k = WG / 2; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + k]);
      __syncthreads();
    }
    if (threadIdx.x == 0) linf_out[0] = red[0];
  }
}

static int smoother_grid(cup2d_ctx *c, int count, const void *kernel) {
  const int ntiles = (count + JT - 1) / JT, wgs = (ntiles + WPG - 1) / WPG;
  int g = resident_grid(c, kernel, count);
  if (g > wgs) g = wgs;
  if (g >= 8) g -= g % 8;
  return g < 1 ? 1 : g;
}

// one launch over the owned blocks; max|r| lands in d_red[0]
template <int MODE>
static int launch_smoother(cup2d_ctx *c, const double *x, const double *b, double *out, double omega) {
  // cache policy (bit 0: b loads, bit 1: out stores, bit 2: x loads non-temporal).  x' of one sweep is x of the next
  // and 134 MB at 4096^2: it survives in the 256 MB memory-side cache only if b, read once per sweep, does not
  // allocate.  Measured at 4096^2, us per sweep: 0: 92   1: 73.5   2: 76.5   4: 84   5: 80   6: 79   3, 7: 90-99.
  constexpr int nt = 1;
  const bool small = c->nblocks < JT;
  using K = void (*)(const double *, const double *, double *, const int *, int, int, double, double, double, double *,
                     unsigned *, double *);
  static const K tab[8] = {(K)k_smoother<MODE, false, 0>, (K)k_smoother<MODE, false, 1>, (K)k_smoother<MODE, false, 2>,
                           (K)k_smoother<MODE, false, 3>, (K)k_smoother<MODE, false, 4>, (K)k_smoother<MODE, false, 5>,
                           (K)k_smoother<MODE, false, 6>, (K)k_smoother<MODE, false, 7>};
  const K k = small ? (K)k_smoother<MODE, true> : tab[nt & 7];
  const int g = smoother_grid(c, c->nblocks, reinterpret_cast<const void *>(k));
  hipLaunchKernelGGL(k, dim3(g), dim3(WG), 0, c->stream, x, b, out, c->d_nbr, 0, c->nblocks, omega / -4.0, omega / -3.0,
                     omega / -2.0, c->d_partials, c->d_ticket, c->d_red);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

__global__ __launch_bounds__(WG) void k_swap_contents(double2 *__restrict__ a, double2 *__restrict__ b, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    const double2 u = a[i], v = b[i];
    a[i] = v;
    b[i] = u;
  }
}

static int smoother_check(cup2d_ctx *c, const char *who) {
  if (c->amr.active || c->mat.active) {
    set_error("%s: built for the matrix-free operator of a same-level grid (no AMR tables, no installed matrix)", who);
    return CUP2D_ERR_UNSUPPORTED;
  }
  return CUP2D_OK;
}

static int smoother_norm(cup2d_ctx *c, double *linf) {
  if (c->allreduce && c->allreduce(c->comm_user, c->d_red, 1, 1, c->stream) != 0) return CUP2D_ERR_COMM;
  if (linf) {
    CUP2D_HIP_CHECK(hipMemcpyAsync(c->h_red, c->d_red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
    *linf = c->h_red[0];
  }
  return CUP2D_OK;
}

}  // namespace cup2d

using namespace cup2d;

extern "C" int cup2d_jacobi_sweeps(cup2d_ctx *c, double omega, int nsweeps, double *linf) {
  CUP2D_CHECK_CTX(c);
  CUP2D_TRY(smoother_check(c, "jacobi_sweeps"));
  if (nsweeps < 0 || !(omega > 0.0)) {
    set_error("jacobi_sweeps: nsweeps >= 0 and omega > 0 expected");
    return CUP2D_ERR_ARG;
  }
  // the sweeps ping-pong between the two slabs; the public slab pointers (cup2d_field_ptr) never change, so after an odd
  // number of sweeps the CONTENTS are exchanged once (the new iterate belongs in PRES, the one before it in POLD)
  double *x = c->d_field[CUP2D_PRES], *xn = c->d_field[CUP2D_POLD];
  for (int s = 0; s < nsweeps; s++) {
    CUP2D_TRY(exchange_halo(c, x, 1, 1));
    {
      ProfScope ps(c, CUP2D_T_SMOOTHER);
      CUP2D_TRY(launch_smoother<0>(c, x, c->d_field[CUP2D_TMP], xn, omega));
    }
    std::swap(x, xn);
  }
  if (nsweeps & 1) {
    const size_t n2 = (size_t)c->ntotal * BC / 2;
    int g = (int)((n2 + WG - 1) / WG);
    if (g > c->grid) g = c->grid;
    hipLaunchKernelGGL(k_swap_contents, dim3(g), dim3(WG), 0, c->stream, (double2 *)c->d_field[CUP2D_PRES],
                       (double2 *)c->d_field[CUP2D_POLD], n2);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  return nsweeps > 0 ? smoother_norm(c, linf) : CUP2D_OK;
}

extern "C" int cup2d_poisson_residual(cup2d_ctx *c, double *linf) {
  CUP2D_CHECK_CTX(c);
  CUP2D_TRY(smoother_check(c, "poisson_residual"));
  CUP2D_TRY(exchange_halo(c, c->d_field[CUP2D_PRES], 1, 1));
  {
    ProfScope ps(c, CUP2D_T_SMOOTHER);
    CUP2D_TRY(launch_smoother<1>(c, c->d_field[CUP2D_PRES], c->d_field[CUP2D_TMP], c->d_field[CUP2D_POLD], 1.0));
  }
  return smoother_norm(c, linf);
}
