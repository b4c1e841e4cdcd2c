// halo.hip -- face-halo pack / unpack kernels for a domain-decomposed block grid
// (sync1 / pack / unpack_subregion, main.cpp:1971-2142, 58-110, same-level faces).
//
// A strip is the `width` cell layers of one 8x8 block nearest one of its faces:
//   face 0 (W): ix in [0,width)      face 1 (E): ix in [8-width,8)
//   face 2 (S): iy in [0,width)      face 3 (N): iy in [8-width,8)
// stored as [8 positions along the face][width layers][dim] for W/E and [width][8][dim] for S/N, i.e.
// in the source block's own (iy, ix) order.  The sender packs strips of owned blocks; the receiver
// unpacks the same cells into its ghost copy of that block, so every stencil kernel reads owned
// and ghost neighbours through the one neighbour table.  One thread per double.
#include "block.h"

namespace cup2d {

static __device__ __forceinline__ int strip_cell(int face, int width, int e) {
  // e in [0, 8*width): cell index (iy*8+ix) of the e-th strip cell in source-block order
  if (face < 2) {
    const int iy = e / width, k = e - iy * width;
    return iy * BS + (face == 0 ? k : BS - width + k);
  }
  const int j = e >> 3, ix = e & 7;
  return (face == 2 ? j : BS - width + j) * BS + ix;
}

template <bool PACK>
__global__ __launch_bounds__(WG) void k_halo(double *__restrict__ field, double *__restrict__ buf,
                                             const int32_t *__restrict__ blocks, const int32_t *__restrict__ faces,
                                             int nstrips, int dim, int width) {
  const int per = BS * width * dim;
  const size_t total = (size_t)nstrips * per;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
    const int s = (int)(i / per), q = (int)(i - (size_t)s * per);
    const int e = q / dim, comp = q - e * dim;
    const size_t cell = (size_t)blocks[s] * BC + strip_cell(faces[s], width, e);
    if (PACK) buf[i] = field[cell * dim + comp];
    else field[cell * dim + comp] = buf[i];
  }
}

int halo_pack_impl(cup2d_ctx *c, const double *src, int dim, int width, double *buf) {
  if (c->plan.nsend == 0) return CUP2D_OK;
  const size_t total = (size_t)c->plan.nsend * BS * width * dim;
  int grid = (int)((total + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  ProfScope prof(c, CUP2D_T_HALO);
  hipLaunchKernelGGL(k_halo<true>, dim3(grid), dim3(WG), 0, c->stream, const_cast<double *>(src), buf,
                     c->plan.d_send_block, c->plan.d_send_face, c->plan.nsend, dim, width);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
int halo_unpack_impl(cup2d_ctx *c, double *dst, int dim, int width, const double *buf) {
  if (c->plan.nrecv == 0) return CUP2D_OK;
  const size_t total = (size_t)c->plan.nrecv * BS * width * dim;
  int grid = (int)((total + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  ProfScope prof(c, CUP2D_T_HALO);
  hipLaunchKernelGGL(k_halo<false>, dim3(grid), dim3(WG), 0, c->stream, dst, const_cast<double *>(buf),
                     c->plan.d_recv_block, c->plan.d_recv_face, c->plan.nrecv, dim, width);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// Whole blocks of TWO scalar vectors in one message: strip s = [64 cells of f0 | 64 cells of f1] of block blocks[s]
// (cell order of a width-8 strip of that face).  The Krylov ghost-block exchange of nu' and p' (krylov_fused.hip).
template <bool PACK>
__global__ __launch_bounds__(WG) void k_halo_blocks2(double *__restrict__ f0, double *__restrict__ f1, double *__restrict__ buf,
                                                     const int32_t *__restrict__ blocks, const int32_t *__restrict__ faces,
                                                     int nstrips) {
  const size_t total = (size_t)nstrips * 2 * BC;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
    const int s = (int)(i / (2 * BC)), q = (int)(i - (size_t)s * 2 * BC);
    double *f = q < BC ? f0 : f1;
    const size_t cell = (size_t)blocks[s] * BC + strip_cell(faces[s], BS, q & (BC - 1));
    if (PACK) buf[i] = f[cell];
    else f[cell] = buf[i];
  }
}
// ... and of THREE (r', p'', nu'' behind the launch that holds sweep E and the next A+B, krylov_fused.hip)
template <bool PACK>
__global__ __launch_bounds__(WG) void k_halo_blocks3(double *__restrict__ f0, double *__restrict__ f1, double *__restrict__ f2,
                                                     double *__restrict__ buf, const int32_t *__restrict__ blocks,
                                                     const int32_t *__restrict__ faces, int nstrips) {
  const size_t total = (size_t)nstrips * 3 * BC;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
    const int s = (int)(i / (3 * BC)), q = (int)(i - (size_t)s * 3 * BC), v = q / BC;
    double *f = v == 0 ? f0 : (v == 1 ? f1 : f2);
    const size_t cell = (size_t)blocks[s] * BC + strip_cell(faces[s], BS, q & (BC - 1));
    if (PACK) buf[i] = f[cell];
    else f[cell] = buf[i];
  }
}
int exchange_begin_blocks3(cup2d_ctx *c, const double *v0, const double *v1, const double *v2) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  if (c->plan.nsend > 0) {
    const size_t total = (size_t)c->plan.nsend * 3 * BC;
    int grid = (int)((total + WG - 1) / WG);
    if (grid > c->grid) grid = c->grid;
    ProfScope prof(c, CUP2D_T_HALO);
    hipLaunchKernelGGL(k_halo_blocks3<true>, dim3(grid), dim3(WG), 0, c->stream, const_cast<double *>(v0), const_cast<double *>(v1),
                       const_cast<double *>(v2), c->d_send, c->plan.d_send_block, c->plan.d_send_face, c->plan.nsend);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  if (c->exchange(c->comm_user, c->d_send, c->d_recv, 3 * BC, c->stream) != 0) {
    set_error("exchange callback failed");
    return CUP2D_ERR_COMM;
  }
  return CUP2D_OK;
}
int exchange_end_blocks3(cup2d_ctx *c, double *v0, double *v1, double *v2) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  if (c->wait && c->wait(c->comm_user, c->stream) != 0) {
    set_error("wait callback failed");
    return CUP2D_ERR_COMM;
  }
  if (c->plan.nrecv == 0) return CUP2D_OK;
  const size_t total = (size_t)c->plan.nrecv * 3 * BC;
  int grid = (int)((total + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  ProfScope prof(c, CUP2D_T_HALO);
  hipLaunchKernelGGL(k_halo_blocks3<false>, dim3(grid), dim3(WG), 0, c->stream, v0, v1, v2, c->d_recv, c->plan.d_recv_block,
                     c->plan.d_recv_face, c->plan.nrecv);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}
int exchange_begin_blocks2(cup2d_ctx *c, const double *v0, const double *v1) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  if (c->plan.nsend > 0) {
    const size_t total = (size_t)c->plan.nsend * 2 * BC;
    int grid = (int)((total + WG - 1) / WG);
    if (grid > c->grid) grid = c->grid;
    ProfScope prof(c, CUP2D_T_HALO);
    hipLaunchKernelGGL(k_halo_blocks2<true>, dim3(grid), dim3(WG), 0, c->stream, const_cast<double *>(v0), const_cast<double *>(v1),
                       c->d_send, c->plan.d_send_block, c->plan.d_send_face, c->plan.nsend);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  if (c->exchange(c->comm_user, c->d_send, c->d_recv, 2 * BC, c->stream) != 0) {
    set_error("exchange callback failed");
    return CUP2D_ERR_COMM;
  }
  return CUP2D_OK;
}
int exchange_end_blocks2(cup2d_ctx *c, double *v0, double *v1) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  if (c->wait && c->wait(c->comm_user, c->stream) != 0) {
    set_error("wait callback failed");
    return CUP2D_ERR_COMM;
  }
  if (c->plan.nrecv == 0) return CUP2D_OK;
  const size_t total = (size_t)c->plan.nrecv * 2 * BC;
  int grid = (int)((total + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  ProfScope prof(c, CUP2D_T_HALO);
  hipLaunchKernelGGL(k_halo_blocks2<false>, dim3(grid), dim3(WG), 0, c->stream, v0, v1, c->d_recv, c->plan.d_recv_block,
                     c->plan.d_recv_face, c->plan.nrecv);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// pack + start the transfer (exchange callback)
int exchange_begin(cup2d_ctx *c, const double *vec, int dim, int width) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  CUP2D_TRY(halo_pack_impl(c, vec, dim, width, c->d_send));
  if (c->exchange(c->comm_user, c->d_send, c->d_recv, BS * width * dim, c->stream) != 0) {
    set_error("exchange callback failed");
    return CUP2D_ERR_COMM;
  }
  return CUP2D_OK;
}
// wait for the transfer + unpack into the ghost blocks
int exchange_end(cup2d_ctx *c, double *vec, int dim, int width) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  if (c->wait && c->wait(c->comm_user, c->stream) != 0) {
    set_error("wait callback failed");
    return CUP2D_ERR_COMM;
  }
  return halo_unpack_impl(c, vec, dim, width, c->d_recv);
}
// ---- cell plans (adapted grids on N ranks): what travels is the list of cells the receiver's kernels read -------------
template <bool PACK>
__global__ __launch_bounds__(WG) void k_cells(double *__restrict__ field, double *__restrict__ buf, const int32_t *__restrict__ cells,
                                              int n, int dim) {
  const size_t total = (size_t)n * dim;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < total; i += (size_t)gridDim.x * WG) {
    const size_t e = i / dim;
    const int comp = (int)(i - e * dim);
    const size_t at = (size_t)cells[e] * dim + comp;
    if (PACK) buf[i] = field[at];
    else field[at] = buf[i];
  }
}
int exchange_cells_begin(cup2d_ctx *c, int set, double *vec, int dim) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  const CellPlan &P = c->cells[set];
  if (!P.active) { set_error("exchange_cells: no cell plan %d", set); return CUP2D_ERR_ARG; }
  if (!c->d_send || !c->d_recv) { set_error("exchange_cells: a cell plan needs both buffers of cup2d_set_comm"); return CUP2D_ERR_ARG; }
  {
    ProfScope prof(c, CUP2D_T_HALO);
    if (P.nsend > 0) {
      int g = (int)(((size_t)P.nsend * dim + WG - 1) / WG);
      hipLaunchKernelGGL(k_cells<true>, dim3(g > c->grid ? c->grid : (g < 1 ? 1 : g)), dim3(WG), 0, c->stream, vec, c->d_send, P.d_send, P.nsend, dim);
      CUP2D_HIP_CHECK(hipGetLastError());
    }
  }
  if (c->exchange(c->comm_user, c->d_send, c->d_recv, CUP2D_CELL_STRIP(set, dim), c->stream) != 0) {
    set_error("exchange callback failed (cell plan %d)", set);
    return CUP2D_ERR_COMM;
  }
  return CUP2D_OK;
}
int exchange_cells_end(cup2d_ctx *c, int set, double *vec, int dim) {
  if (c->nghost == 0 || !c->exchange) return CUP2D_OK;
  const CellPlan &P = c->cells[set];
  if (c->wait && c->wait(c->comm_user, c->stream) != 0) {
    set_error("wait callback failed");
    return CUP2D_ERR_COMM;
  }
  if (P.nrecv > 0) {
    ProfScope prof(c, CUP2D_T_HALO);
    int g = (int)(((size_t)P.nrecv * dim + WG - 1) / WG);
    hipLaunchKernelGGL(k_cells<false>, dim3(g > c->grid ? c->grid : (g < 1 ? 1 : g)), dim3(WG), 0, c->stream, vec, c->d_recv, P.d_recv, P.nrecv, dim);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  return CUP2D_OK;
}
int exchange_cells(cup2d_ctx *c, int set, double *vec, int dim) {
  CUP2D_TRY(exchange_cells_begin(c, set, vec, dim));
  return exchange_cells_end(c, set, vec, dim);
}
int exchange_halo(cup2d_ctx *c, double *vec, int dim, int width) {
  CUP2D_TRY(exchange_begin(c, vec, dim, width));
  return exchange_end(c, vec, dim, width);
}

}  // namespace cup2d
