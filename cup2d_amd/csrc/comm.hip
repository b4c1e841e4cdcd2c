// comm.hip -- the communicator inside the library: RCCL over xGMI, no host language in the data path.
//
// Replaces, for a domain-decomposed block grid (one process per GPU):
//   * the face exchange of the reference's synchroniser -- MPI_Irecv / MPI_Isend of the packed strips and the
//     MPI_Waitall before the halo blocks are swept (main.cpp:2040-2047, 2133-2139, computeA's overlap 3035-3057;
//     for the assembled operator cuda.cu:365-380): ncclRecv / ncclSend pairs of one ncclGroup on a SECOND HIP stream,
//     ordered after the pack kernel by an event; the compute stream waits for the arrival event only where the
//     library unpacks, so the inner blocks are swept while the strips are on the links;
//   * the scalar reductions -- MPI_Allreduce of max|u| and of the pressure means (main.cpp:6583-6592, 7138, 7162)
//     and the solver's dot products and norms (cuda.cu:445-449, 491-493, 513-515, 533-534): ncclAllReduce of <= 3
//     doubles on the compute stream, or ONE ncclAllGather of {sum, sum, max} per rank followed by a fixed-order
//     finish on the device where a sum and a max are due together (a sum and a max cannot share an all-reduce).
// Two communicators, one per stream: RCCL serialises the operations of ONE communicator in issue order, and a
// reduction must not queue behind an exchange that is deliberately left in flight.
//
// librccl.so.1 is opened on first use (dlopen), not linked: a process that never decomposes its grid never loads it.
// The callback interface of cup2d_set_comm stays for callers that bring their own transport (the gloo tests).
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "krylov_common.h"

namespace cup2d {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string path;  // the file the symbols came from (dladdr), for cup2d_comm_selftest's report
};

static RcclApi *rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  // The RCCL that is opened must sit on the SAME HIP runtime this library is linked to (the streams, events and
  // buffers handed to it are that runtime's): the librccl next to it.  A process may have mapped another pair already --
  // PyTorch ships libamdhip64.so / librccl.so of its own, without sonames, so the loader treats them as different
  // libraries -- and RCCL on that other runtime would see none of this library's streams.
  std::string sib1, sib2;
  {
    Dl_info di;
    if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t cut = dir.rfind('/');
      if (cut != std::string::npos) {
        dir.resize(cut + 1);
        sib1 = dir + "librccl.so.1";
        sib2 = dir + "librccl.so";
      }
    }
  }
  // RCCL itself looks the HSA runtime up BY NAME (dlopen("libhsa-runtime64.so"), rocmwrap.cc) to ask it about dmabuf
  // support; in a process that has PyTorch imported but not yet on the GPU that name is PyTorch's own, uninitialised,
  // copy and the query fails with HSA_STATUS_ERROR_NOT_INITIALIZED.  hsa_init() is reference-counted: bring that copy up.
  if (void *hsa = dlopen("libhsa-runtime64.so", RTLD_NOW | RTLD_NOLOAD)) {
    auto get = reinterpret_cast<int (*)(int, void *)>(dlsym(hsa, "hsa_system_get_info"));
    auto init = reinterpret_cast<int (*)()>(dlsym(hsa, "hsa_init"));
    unsigned short major = 0;
    if (get && init && get(0 /* HSA_SYSTEM_INFO_VERSION_MAJOR */, &major) == 0x100B /* not initialised */) init();
  }
  const char *names[] = {getenv("CUP2D_RCCL_LIB"), sib1.c_str(), sib2.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)
    if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) {
    set_error("comm: cannot open librccl.so.1 (%s)", dlerror());
    return nullptr;
  }
  bool ok = true;
  const auto sym = [&](const char *name) -> void * {
    void *p = dlsym(h, name);
    if (!p) {
      set_error("comm: librccl lacks %s", name);
      ok = false;
    }
    return p;
  };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {
    dlclose(h);
    return nullptr;
  }
  {
    Dl_info di;
    if (dladdr(reinterpret_cast<const void *>(api.Send), &di) && di.dli_fname) api.path = di.dli_fname;
  }
  api.handle = h;
  return &api;
}

struct RcclComm {
  RcclApi *api = nullptr;
  ncclComm_t p2p = nullptr, red = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_arrived = nullptr;
  int nranks = 1, rank = 0;
  std::vector<int> peer, soff, roff, cnt, rcnt;   // per peer: rank, first strip in the send / receive list, strips sent / received
  std::vector<int> rblock0;  // per peer: the first of its ghost blocks
  std::vector<int> csoff[CELL_SETS], croff[CELL_SETS], ccnt[CELL_SETS], crcnt[CELL_SETS];  // the same for the cell plans (cup2d_comm_set_cell_counts)
  bool direct = false;       // every peer's ghost blocks are consecutive: whole blocks can be received in place
  bool defer_ok = false;     // ... on every rank, and every rank has ghost blocks: the reduction records may ride in the send/recv group
  bool split_ok = false;     // every rank's inner / halo cut falls on a tile boundary: split sweeps are possible on ALL ranks
  double *d_send = nullptr, *d_recv = nullptr, *d_red = nullptr, *d_gather = nullptr;
  long long n_exchange = 0, n_allreduce = 0, n_allgather = 0;  // calls issued (diagnostics)
};

#define CUP2D_NCCL(rc, expr)                                                                     \
  do {                                                                                           \
    ncclResult_t _r = (expr);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, (rc)->api->GetErrorString(_r));     \
      return -1;                                                                                 \
    }                                                                                            \
  } while (0)
#define CUP2D_HIP_CB(expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));             \
      return -1;                                                                                 \
    }                                                                                            \
  } while (0)

// exchange callback: the strips packed by work already enqueued on `stream` go out on the communication stream
static int rccl_exchange(void *user, double *send, double *recv, int strip_doubles, void *stream) {
  RcclComm *rc = static_cast<RcclComm *>(user);
  rc->n_exchange++;
  if (rc->peer.empty()) return 0;
  CUP2D_HIP_CB(hipEventRecord(rc->ev_packed, (hipStream_t)stream));
  CUP2D_HIP_CB(hipStreamWaitEvent(rc->comm_stream, rc->ev_packed, 0));
  // strips of the block plan, or (strip_doubles < 0: CUP2D_CELL_STRIP) cells of a cell plan with its own offsets and counts
  const std::vector<int> *soff = &rc->soff, *roff = &rc->roff, *cnt = &rc->cnt, *rcnt = &rc->rcnt;
  size_t sd = (size_t)(strip_doubles > 0 ? strip_doubles : 0);
  if (strip_doubles < 0) {
    const int set = CUP2D_CELL_STRIP_SET(strip_doubles);
    sd = (size_t)CUP2D_CELL_STRIP_DIM(strip_doubles);
    if (set < 0 || set >= CELL_SETS || rc->ccnt[set].size() != rc->peer.size()) {
      set_error("rccl exchange: no cell counts for set %d (cup2d_comm_set_cell_counts)", set);
      return -1;
    }
    soff = &rc->csoff[set]; roff = &rc->croff[set]; cnt = &rc->ccnt[set]; rcnt = &rc->crcnt[set];
  }
  CUP2D_NCCL(rc, rc->api->GroupStart());
  for (size_t i = 0; i < rc->peer.size(); i++)  // receives first, as main.cpp:2040-2047 posts them
    if ((*rcnt)[i] > 0)
      CUP2D_NCCL(rc, rc->api->Recv(recv + (*roff)[i] * sd, (*rcnt)[i] * sd, ncclDouble, rc->peer[i], rc->p2p, rc->comm_stream));
  for (size_t i = 0; i < rc->peer.size(); i++)
    if ((*cnt)[i] > 0)
      CUP2D_NCCL(rc, rc->api->Send(send + (*soff)[i] * sd, (*cnt)[i] * sd, ncclDouble, rc->peer[i], rc->p2p, rc->comm_stream));
  CUP2D_NCCL(rc, rc->api->GroupEnd());
  CUP2D_HIP_CB(hipEventRecord(rc->ev_arrived, rc->comm_stream));
  return 0;
}
// Whole ghost blocks of the Krylov vectors (krylov_fused.hip, the ghost-block form of the sweeps), the short way.  Measured on
// a patch that is its own W and E neighbour (tools/gpu_selfperiodic_step.py, tools/kernel_timeline.py; 4096^2 cells): between
// a sweep and the next one the generic path spends 57 us -- pack 5, 16 idle until the send/recv kernel starts on the
// communication stream (event record -> wait on the other stream -> launch), the kernel 13, 17 idle until the unpack kernel
// starts on the compute stream (the same hand-over back), unpack 6 -- to overlap the transfer with 11 us of all-gather and
// scalar kernel.  There is no inner sweep to hide these exchanges behind (the sweep that follows needs both the ghost blocks
// and the scalars), so they run on the COMPUTE stream, back to back with the reduction, and the blocks of a peer -- which
// cup2d_halo_plan numbers consecutively -- are received straight into the vectors' ghost regions: no second stream, no event,
// no unpack kernel.  The send buffer is vector-major ([vector][strip][64 cells], a whole block in its own cell order) so that
// what goes to one peer for one vector is one contiguous piece.
// G (optional, krylov_fused.hip GhostRP): the same launch forms r' and p'' of the ghost blocks from what the rank holds of
// them -- the elements behind the packed ones; nothing of it touches what is packed (nu'')
__global__ __launch_bounds__(WG) void k_pack_blocks(const double *__restrict__ v0, const double *__restrict__ v1,
                                                   const double *__restrict__ v2, double *__restrict__ buf,
                                                   const int32_t *__restrict__ blocks, int nstrips, int nv, GhostRP G,
                                                   const double *__restrict__ rec, double *__restrict__ rec_slot) {
  // (records in the group: this rank's record into its own slot of the gathered records -- what the other ranks receive)
  if (rec_slot && blockIdx.x == 0 && threadIdx.x < RED_REC) rec_slot[threadIdx.x] = rec[threadIdx.x];
  const size_t per = (size_t)nstrips * BC, total = per * nv;
  const bool ghosts = G.count > 0 && G.sc->status == 0;
  const size_t all = total + (ghosts ? G.count : 0);
  double malpha = 0, c1 = 0, beta = 0;
  bool restart = false;
  if (ghosts) { malpha = -G.sc->alpha; c1 = -G.sc->omega; beta = G.sc->beta; restart = G.sc->restart_flag != 0; }
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < all; i += (size_t)gridDim.x * WG) {
    if (i < total) {
      const int v = (int)(i / per);
      const size_t q = i - (size_t)v * per;
      const double *f = v == 0 ? v0 : (v == 1 ? v1 : v2);
      buf[i] = f[(size_t)blocks[q >> 6] * BC + (q & 63)];
    } else {  // form_v of krylov_edge.h MODE 2, operation for operation
      const size_t k = G.first + (i - total);
      const double sv = G.r[k] + malpha * G.nu[k];
      const double rn = sv + c1 * G.t[k];
      G.rout[k] = rn;
      if (restart) G.pout[k] = rn;
      else {
        double v = G.p[k] + c1 * G.nu[k];
        v = v * beta;
        G.pout[k] = v + rn;
      }
    }
  }
}
// (rc->direct is agreed over all ranks in cup2d_comm_init -- the two paths have different wire formats, so a rank must not
// choose alone; CUP2D_COMM_DIRECT=0 on any rank turns the in-place path off on every rank)
bool comm_blocks_direct(const cup2d_ctx *c) {
  return c->rccl && c->comm_user == (void *)c->rccl && c->rccl->direct && c->nghost > 0;
}
int comm_blocks_wait(cup2d_ctx *c) {
  RcclComm *rc = c->rccl;
  if (rc->peer.empty()) return CUP2D_OK;
  CUP2D_HIP_CHECK(hipStreamWaitEvent(c->stream, rc->ev_arrived, 0));
  return CUP2D_OK;
}
const double *comm_gathered(const cup2d_ctx *c) { return c->rccl ? c->rccl->d_gather : nullptr; }
int comm_nranks(const cup2d_ctx *c) { return c->rccl ? c->rccl->nranks : 1; }
bool comm_defer_ok(const cup2d_ctx *c) { return c->rccl && c->comm_user == (void *)c->rccl && c->rccl->defer_ok; }
bool comm_split_ok(const cup2d_ctx *c) { return c->rccl && c->comm_user == (void *)c->rccl && c->rccl->split_ok; }
int comm_exchange_blocks(cup2d_ctx *c, int nv, double *v0, double *v1, double *v2, bool on_comm_stream, const GhostRP *ghosts, bool records) {
  RcclComm *rc = c->rccl;
  rc->n_exchange++;
  if (rc->peer.empty() || nv < 1 || nv > 3) {
    if (records) { set_error("comm_exchange_blocks: records without a block exchange"); return CUP2D_ERR_ARG; }
    return CUP2D_OK;
  }
  hipStream_t xs = on_comm_stream ? rc->comm_stream : c->stream;
  double *vecs[3] = {v0, v1, v2};
  const int ns = c->plan.nsend;
  GhostRP G = ghosts ? *ghosts : GhostRP();
  if (ns > 0 || G.count > 0 || records) {
    const size_t total = (size_t)ns * BC * nv + G.count;
    int grid = (int)((total + WG - 1) / WG);
    if (grid > c->grid) grid = c->grid;
    if (grid < 1) grid = 1;
    ProfScope prof(c, CUP2D_T_HALO);
    hipLaunchKernelGGL(k_pack_blocks, dim3(grid), dim3(WG), 0, c->stream, (const double *)v0, (const double *)v1, (const double *)v2,
                       rc->d_send, (const int32_t *)c->plan.d_send_block, ns, nv, G, (const double *)c->d_red,
                       records ? rc->d_gather + (size_t)RED_REC * rc->rank : (double *)nullptr);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  const auto fail = [&](ncclResult_t r, const char *what) {
    set_error("comm_exchange_blocks: %s -> %s", what, rc->api->GetErrorString(r));
    return CUP2D_ERR_COMM;
  };
  if (on_comm_stream) {
    CUP2D_HIP_CHECK(hipEventRecord(rc->ev_packed, c->stream));
    CUP2D_HIP_CHECK(hipStreamWaitEvent(rc->comm_stream, rc->ev_packed, 0));
  }
  ncclResult_t r = rc->api->GroupStart();
  if (r != ncclSuccess) return fail(r, "ncclGroupStart");
  if (records) {  // rank by rank, in front of the blocks (RCCL pairs the operations of two ranks in issue order: every rank issues this order)
    for (int q = 0; q < rc->nranks; q++)
      if (q != rc->rank) {
        r = rc->api->Recv(rc->d_gather + (size_t)RED_REC * q, RED_REC, ncclDouble, q, rc->p2p, xs);
        if (r != ncclSuccess) return fail(r, "ncclRecv(record)");
      }
    for (int q = 0; q < rc->nranks; q++)
      if (q != rc->rank) {
        r = rc->api->Send(c->d_red, RED_REC, ncclDouble, q, rc->p2p, xs);
        if (r != ncclSuccess) return fail(r, "ncclSend(record)");
      }
  }
  for (int v = 0; v < nv; v++)  // receives first, vector by vector and peer by peer: the order the peers send in
    for (size_t i = 0; i < rc->peer.size(); i++)
      if (rc->rcnt[i] > 0) {
        r = rc->api->Recv(vecs[v] + (size_t)rc->rblock0[i] * BC, (size_t)rc->rcnt[i] * BC, ncclDouble, rc->peer[i], rc->p2p, xs);
        if (r != ncclSuccess) return fail(r, "ncclRecv");
      }
  for (int v = 0; v < nv; v++)
    for (size_t i = 0; i < rc->peer.size(); i++)
      if (rc->cnt[i] > 0) {
        r = rc->api->Send(rc->d_send + ((size_t)v * ns + rc->soff[i]) * BC, (size_t)rc->cnt[i] * BC, ncclDouble, rc->peer[i], rc->p2p, xs);
        if (r != ncclSuccess) return fail(r, "ncclSend");
      }
  r = rc->api->GroupEnd();
  if (r != ncclSuccess) return fail(r, "ncclGroupEnd");
  if (on_comm_stream) CUP2D_HIP_CHECK(hipEventRecord(rc->ev_arrived, rc->comm_stream));
  return CUP2D_OK;
}
// wait callback: the compute stream goes on once the strips of the last exchange have arrived
static int rccl_wait(void *user, void *stream) {
  RcclComm *rc = static_cast<RcclComm *>(user);
  if (rc->peer.empty()) return 0;
  CUP2D_HIP_CB(hipStreamWaitEvent((hipStream_t)stream, rc->ev_arrived, 0));
  return 0;
}
static int rccl_allreduce(void *user, double *buf, int count, int op, void *stream) {
  RcclComm *rc = static_cast<RcclComm *>(user);
  rc->n_allreduce++;
  CUP2D_NCCL(rc, rc->api->AllReduce(buf, buf, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, rc->red, (hipStream_t)stream));
  return 0;
}

// The solver's reductions with the in-library communicator: {sum, sum, max} of every rank side by side (ONE all-gather of
// three doubles per rank), then ONE single-wave kernel that adds / maximises them in rank order -- a summation order that
// does not depend on the algorithm RCCL picks, and a sum and a max in one collective -- and runs the scalar update of the
// stage on the result (krylov_common.h scalars_update; stage < 0: only the reduced values, into red).
__global__ void k_gather_scalars(const double *__restrict__ g, int nranks, int nsum, int with_max, double *__restrict__ red,
                                 KrylovScalars *sc, int stage, int *host_status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (stage > 0 && sc->status != 0) {  // the solve is over: the sweeps before this were no-ops
    // (a group's last iteration still tells the host, solve_fused_impl: one look per group of iterations)
    if ((stage == 3 || stage == 4) && host_status) __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // (a record per rank: RED_REC doubles -- up to five sums, krylov_common.h stage 5; with a maximum it is entry 2 and the
  // sums are at most two)
  double v[RED_REC];
  sum_records(g, nranks, nsum, with_max, v);
  for (int k = 0; k < RED_REC; k++) red[k] = v[k];
  if (stage >= 0) {
    scalars_update(sc, v, stage);
    if ((stage == 3 || stage == 4) && host_status) __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int comm_reduce_scalars(cup2d_ctx *c, int nsum, int with_max, int stage, int *host_status) {
  RcclComm *rc = c->rccl;
  rc->n_allgather++;
  CUP2D_NCCL(rc, rc->api->AllGather(c->d_red, rc->d_gather, RED_REC, ncclDouble, rc->red, c->stream));
  hipLaunchKernelGGL(k_gather_scalars, dim3(1), dim3(64), 0, c->stream, rc->d_gather, rc->nranks, nsum, with_max, c->d_red, c->d_sc,
                     stage, host_status);
  CUP2D_HIP_CB(hipGetLastError());
  return 0;
}

// this rank's record (d_red) to slot `rank` of every rank's gathered records: one all-gather on the compute stream, no kernel
// behind it (the consumer sweep sums the records in its prologue)
int comm_gather_records(cup2d_ctx *c) {
  RcclComm *rc = c->rccl;
  rc->n_allgather++;
  CUP2D_NCCL(rc, rc->api->AllGather(c->d_red, rc->d_gather, RED_REC, ncclDouble, rc->red, c->stream));
  return 0;
}
int comm_apply_gathered(cup2d_ctx *c, int nsum, int with_max, int stage) {
  RcclComm *rc = c->rccl;
  hipLaunchKernelGGL(k_gather_scalars, dim3(1), dim3(64), 0, c->stream, rc->d_gather, rc->nranks, nsum, with_max, c->d_red, c->d_sc,
                     stage, (int *)nullptr);
  CUP2D_HIP_CB(hipGetLastError());
  return 0;
}

int comm_finalize_impl(cup2d_ctx *c) {
  RcclComm *rc = c->rccl;
  if (!rc) return CUP2D_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (rc->comm_stream) (void)hipStreamSynchronize(rc->comm_stream);
  if (c->comm_user == rc) {
    c->exchange = nullptr; c->wait = nullptr; c->allreduce = nullptr; c->comm_user = nullptr;
    c->d_send = c->d_recv = nullptr;
    c->d_red = c->d_red_own;
  }
  if (rc->p2p) (void)rc->api->CommDestroy(rc->p2p);
  if (rc->red) (void)rc->api->CommDestroy(rc->red);
  if (rc->ev_packed) (void)hipEventDestroy(rc->ev_packed);
  if (rc->ev_arrived) (void)hipEventDestroy(rc->ev_arrived);
  if (rc->comm_stream) (void)hipStreamDestroy(rc->comm_stream);
  (void)hipFree(rc->d_send); (void)hipFree(rc->d_recv); (void)hipFree(rc->d_red); (void)hipFree(rc->d_gather);
  delete rc;
  c->rccl = nullptr;
  return CUP2D_OK;
}

}  // namespace cup2d

using namespace cup2d;

extern "C" {

int cup2d_comm_unique_id(void *id_bytes) {
  if (!id_bytes) { set_error("comm_unique_id: null"); return CUP2D_ERR_ARG; }
  RcclApi *api = rccl_api();
  if (!api) return CUP2D_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) * 2 == CUP2D_COMM_ID_BYTES, "CUP2D_COMM_ID_BYTES = two ncclUniqueIds");
  ncclUniqueId ids[2];
  for (auto &id : ids) {
    const ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { set_error("ncclGetUniqueId -> %s", api->GetErrorString(r)); return CUP2D_ERR_COMM; }
  }
  memcpy(id_bytes, ids, sizeof ids);
  return CUP2D_OK;
}

// COLLECTIVE over the nranks processes: ncclCommInitRank below blocks until every rank has arrived.  The argument and plan
// checks in front of it are functions of the decomposition, which every rank derives from the same numbers -- ranks must
// validate identically (a rank that returns early leaves the others in the rendezvous; callers bound that wait with their
// own watchdog, as bench.py and cup2d_run_mpi do, and run cup2d_comm_selftest right after a successful init).  The two
// communicators are used from two streams with no ordering between ranks; RCCL documents concurrent communicators as safe
// only while every rank issues the operations of EACH communicator in the same order -- which holds here: the exchange
// stream carries only the plan's send/recv groups, the compute stream only the solver's reductions, both in program order.
int cup2d_comm_init(cup2d_ctx *c, int nranks, int rank, const void *id_bytes, int npeers, const int32_t *peer_rank,
                    const int32_t *send_offset, const int32_t *recv_offset, const int32_t *nstrips, const int32_t *nstrips_recv) {
  CUP2D_CHECK_CTX(c);
  if (nranks < 1 || rank < 0 || rank >= nranks || !id_bytes || npeers < 0 ||
      (npeers && (!peer_rank || !send_offset || !recv_offset || !nstrips))) {
    set_error("comm_init: bad argument");
    return CUP2D_ERR_ARG;
  }
  if (!nstrips_recv) nstrips_recv = nstrips;  // same-level faces: as many strips come in as go out
  for (int i = 0; i < npeers; i++)
    if (peer_rank[i] < 0 || peer_rank[i] >= nranks || nstrips[i] < 0 || nstrips_recv[i] < 0 || send_offset[i] < 0 || recv_offset[i] < 0 ||
        send_offset[i] + nstrips[i] > c->plan.nsend || recv_offset[i] + nstrips_recv[i] > c->plan.nrecv) {
      set_error("comm_init: peer %d (rank %d, strips %d at %d out, %d at %d in) does not fit the halo plan (%d sent, %d received): "
                "call cup2d_halo_plan first", i, peer_rank[i], nstrips[i], send_offset[i], nstrips_recv[i], recv_offset[i], c->plan.nsend, c->plan.nrecv);
      return CUP2D_ERR_ARG;
    }
  if (c->rccl) CUP2D_TRY(comm_finalize_impl(c));
  RcclApi *api = rccl_api();
  if (!api) return CUP2D_ERR_COMM;
  RcclComm *rc = new RcclComm;
  c->rccl = rc;  // owned by the context from here on: cup2d_destroy / comm_finalize release whatever exists
  rc->api = api;
  rc->nranks = nranks;
  rc->rank = rank;
  for (int i = 0; i < npeers; i++) {
    if (nstrips[i] == 0 && nstrips_recv[i] == 0) continue;
    rc->peer.push_back(peer_rank[i]); rc->soff.push_back(send_offset[i]);
    rc->roff.push_back(recv_offset[i]); rc->cnt.push_back(nstrips[i]); rc->rcnt.push_back(nstrips_recv[i]);
  }
  // whole ghost blocks can be received in place where every peer's ghost blocks are consecutive (cup2d_amd/grid.py and the
  // C++ plans number them so; a plan that does not keeps the generic path)
  rc->direct = (int)c->plan.h_recv_block.size() == c->plan.nrecv;
  for (size_t i = 0; i < rc->peer.size() && rc->direct; i++) {
    const int r0 = rc->rcnt[i] > 0 ? c->plan.h_recv_block[(size_t)rc->roff[i]] : c->nblocks;
    rc->rblock0.push_back(r0);
    for (int k = 0; k < rc->rcnt[i]; k++)
      if (c->plan.h_recv_block[(size_t)rc->roff[i] + k] != r0 + k) rc->direct = false;
  }
  // widest message: whole blocks of three Krylov vectors = 192 doubles per strip (the WENO halo is 3 x 8 x 2 = 48)
  const size_t strip = 3 * BC;
  CUP2D_HIP_CHECK(hipMalloc(&rc->d_send, sizeof(double) * strip * (size_t)(c->plan.nsend > 0 ? c->plan.nsend : 1)));
  CUP2D_HIP_CHECK(hipMalloc(&rc->d_recv, sizeof(double) * strip * (size_t)(c->plan.nrecv > 0 ? c->plan.nrecv : 1)));
  CUP2D_HIP_CHECK(hipMalloc(&rc->d_red, sizeof(double) * 8));
  CUP2D_HIP_CHECK(hipMalloc(&rc->d_gather, sizeof(double) * RED_REC * (size_t)nranks));
  CUP2D_HIP_CHECK(hipMemset(rc->d_red, 0, sizeof(double) * 8));
  CUP2D_HIP_CHECK(hipStreamCreateWithFlags(&rc->comm_stream, hipStreamNonBlocking));
  CUP2D_HIP_CHECK(hipEventCreateWithFlags(&rc->ev_packed, hipEventDisableTiming));
  CUP2D_HIP_CHECK(hipEventCreateWithFlags(&rc->ev_arrived, hipEventDisableTiming));
  ncclUniqueId ids[2];
  memcpy(ids, id_bytes, sizeof ids);
  ncclResult_t r = api->CommInitRank(&rc->p2p, nranks, ids[0], rank);
  if (r == ncclSuccess) r = api->CommInitRank(&rc->red, nranks, ids[1], rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(%d of %d) -> %s", rank, nranks, api->GetErrorString(r));
    return CUP2D_ERR_COMM;
  }
  // The in-place receive of whole ghost blocks and the generic exchange differ on the wire (nv messages of cnt x 64 doubles per
  // peer, vector-major, against one strip-major message): the choice is made by ALL ranks together -- a minimum over the ranks
  // of "my ghost blocks are consecutive per peer and CUP2D_COMM_DIRECT is not 0 here".
  {
    static const bool env_on = [] { const char *e = getenv("CUP2D_COMM_DIRECT"); return !e || atoi(e) != 0; }();
    // [1]: the reduction records in the send/recv group and the scalar updates in the consumer sweeps (krylov_fused.hip
    // "deferred"): every rank must have ghost blocks (a block exchange per reduction point to ride on); CUP2D_DEFER_SCALARS=0
    static const bool defer_on = [] { const char *e = getenv("CUP2D_DEFER_SCALARS"); return !e || atoi(e) != 0; }();
    // [2]: split sweeps (halo set first).  With the deferred update a split sweep and an unsplit one differ on the wire (records
    // in the send/recv group against an all-gather behind the inner launch), so a rank must not decide from its own n_inner
    // alone: on small grids the ranks' cuts differ (8 x 8 blocks per rank in 1 x 3: 56 / 48 / 56 inner blocks)
    const bool split_here = c->n_inner > 0 && c->n_inner < c->nblocks && c->n_inner % FUSED_TILE == 0;
    const double mine[3] = {(rc->direct && env_on) ? 1.0 : 0.0, (rc->direct && env_on && defer_on && c->nghost > 0 && !rc->peer.empty()) ? 1.0 : 0.0,
                            split_here ? 1.0 : 0.0};
    double all[3] = {0.0, 0.0, 0.0};
    CUP2D_HIP_CHECK(hipMemcpy(rc->d_red, mine, sizeof mine, hipMemcpyHostToDevice));
    r = api->AllReduce(rc->d_red, rc->d_red, 3, ncclDouble, ncclMin, rc->red, c->stream);
    if (r != ncclSuccess) {
      set_error("comm_init: ncclAllReduce(direct) -> %s", api->GetErrorString(r));
      return CUP2D_ERR_COMM;
    }
    CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
    CUP2D_HIP_CHECK(hipMemcpy(all, rc->d_red, sizeof all, hipMemcpyDeviceToHost));
    CUP2D_HIP_CHECK(hipMemset(rc->d_red, 0, sizeof(double) * 8));
    rc->direct = all[0] == 1.0;
    rc->defer_ok = all[1] == 1.0;
    rc->split_ok = all[2] == 1.0;
  }
  CUP2D_TRY(cup2d_set_comm(c, rccl_exchange, rccl_wait, rccl_allreduce, rc, rc->d_send, rc->d_recv, rc->d_red));
  return cup2d_set_comm_strip_capacity(c, (int)strip);  // three whole blocks per strip: allocated above
}

// One round of everything the time loop will ask of the communicator, with known values and a deadline: the strips of the
// halo plan between every pair of peers (ncclSend / ncclRecv on the communication stream, as cup2d_halo_exchange issues them),
// an all-reduce and an all-gather on the compute stream's communicator.  Collective: every rank of the communicator calls
// it, right after cup2d_comm_init.  A rank whose operations do not complete within timeout_s returns CUP2D_ERR_COMM with
// the stage it was waiting for (the caller should then end the process: the collective cannot be cancelled); wrong values
// are reported the same way.  info (may be NULL): "rccl=<library file> ranks=N rank=r peers=a,b,.. exchange_us=.. reduce_us=..".
int cup2d_comm_selftest(cup2d_ctx *c, double timeout_s, char *info, int info_bytes) {
  CUP2D_CHECK_CTX(c);
  RcclComm *rc = c->rccl;
  if (!rc || c->comm_user != rc) { set_error("comm_selftest: no in-library communicator (cup2d_comm_init)"); return CUP2D_ERR_ARG; }
  if (timeout_s <= 0) timeout_s = 20.0;
  const int ns = c->plan.nsend, nr = c->plan.nrecv;
  std::vector<double> hs((size_t)(ns > 0 ? ns : 1), (double)(rc->rank + 1)), hr((size_t)(nr > 0 ? nr : 1), -1.0);
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  CUP2D_HIP_CHECK(hipMemcpy(rc->d_send, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(rc->d_recv, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice));
  const auto t0 = std::chrono::steady_clock::now();
  const auto wait_stream = [&](const char *stage) -> int {  // bounded wait: poll, never block inside the runtime
    for (;;) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) return CUP2D_OK;
      if (q != hipErrorNotReady) { set_error("comm_selftest: %s -> %s", stage, hipGetErrorString(q)); return CUP2D_ERR_HIP; }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
        set_error("comm_selftest: rank %d of %d still waiting for '%s' after %.0f s (librccl %s) -- a peer is missing or the "
                  "links are down", rc->rank, rc->nranks, stage, timeout_s, rc->api->path.c_str());
        return CUP2D_ERR_COMM;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
  };
  // 1. the strips of the plan, one double each
  if (rccl_exchange(rc, rc->d_send, rc->d_recv, 1, c->stream) != 0 || rccl_wait(rc, c->stream) != 0) return CUP2D_ERR_COMM;
  CUP2D_TRY(wait_stream("send/recv of the halo plan's strips"));
  const double us_x = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
  CUP2D_HIP_CHECK(hipMemcpy(hr.data(), rc->d_recv, hr.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < rc->peer.size(); i++)
    for (int k = 0; k < rc->rcnt[i]; k++)
      if (hr[(size_t)rc->roff[i] + k] != (double)(rc->peer[i] + 1)) {
        set_error("comm_selftest: rank %d received %g in strip %d from rank %d (expected %d)", rc->rank, hr[(size_t)rc->roff[i] + k],
                  rc->roff[i] + k, rc->peer[i], rc->peer[i] + 1);
        return CUP2D_ERR_COMM;
      }
  // 2. all-reduce (sum) and all-gather on the compute stream's communicator
  const auto t1 = std::chrono::steady_clock::now();
  double three[3] = {(double)(rc->rank + 1), 2.0 * (rc->rank + 1), (double)rc->rank};
  CUP2D_HIP_CHECK(hipMemcpy(c->d_red, three, sizeof three, hipMemcpyHostToDevice));
  CUP2D_NCCL(rc, rc->api->AllGather(c->d_red, rc->d_gather, 3, ncclDouble, rc->red, c->stream));
  if (rccl_allreduce(rc, c->d_red, 1, 0, c->stream) != 0) return CUP2D_ERR_COMM;
  CUP2D_TRY(wait_stream("all-gather / all-reduce"));
  const double us_r = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() * 1e6;
  std::vector<double> g((size_t)3 * rc->nranks);
  double sum = 0;
  CUP2D_HIP_CHECK(hipMemcpy(g.data(), rc->d_gather, g.size() * sizeof(double), hipMemcpyDeviceToHost));
  CUP2D_HIP_CHECK(hipMemcpy(&sum, c->d_red, sizeof sum, hipMemcpyDeviceToHost));
  for (int r = 0; r < rc->nranks; r++)
    if (g[3 * r] != r + 1 || g[3 * r + 1] != 2.0 * (r + 1) || g[3 * r + 2] != r) {
      set_error("comm_selftest: all-gather slot %d holds (%g, %g, %g)", r, g[3 * r], g[3 * r + 1], g[3 * r + 2]);
      return CUP2D_ERR_COMM;
    }
  if (sum != 0.5 * rc->nranks * (rc->nranks + 1)) {
    set_error("comm_selftest: all-reduce gave %g, expected %g", sum, 0.5 * rc->nranks * (rc->nranks + 1));
    return CUP2D_ERR_COMM;
  }
  CUP2D_HIP_CHECK(hipMemset(c->d_red, 0, sizeof(double) * 8));
  rc->n_exchange = rc->n_allreduce = rc->n_allgather = 0;  // the counters describe the time loop
  if (info && info_bytes > 0) {
    std::string peers;
    for (size_t i = 0; i < rc->peer.size(); i++) peers += (i ? "," : "") + std::to_string(rc->peer[i]);
    snprintf(info, (size_t)info_bytes, "rccl=%s ranks=%d rank=%d peers=%s exchange_us=%.0f reduce_us=%.0f", rc->api->path.c_str(),
             rc->nranks, rc->rank, peers.empty() ? "-" : peers.c_str(), us_x, us_r);
  }
  return CUP2D_OK;
}

int cup2d_comm_finalize(cup2d_ctx *c) {
  CUP2D_CHECK_CTX(c);
  return comm_finalize_impl(c);
}

int cup2d_comm_set_cell_counts(cup2d_ctx *c, int set, int npeers, const int32_t *so, const int32_t *sn, const int32_t *ro,
                               const int32_t *rn) {
  CUP2D_CHECK_CTX(c);
  if (!c->rccl) { set_error("comm_set_cell_counts: no in-library communicator (cup2d_comm_init)"); return CUP2D_ERR_COMM; }
  RcclComm *rc = c->rccl;
  if (set < 0 || set >= CELL_SETS || npeers != (int)rc->peer.size() || (npeers && (!so || !sn || !ro || !rn))) {
    set_error("comm_set_cell_counts: set %d, %d peers (the communicator has %d)", set, npeers, (int)rc->peer.size());
    return CUP2D_ERR_ARG;
  }
  const CellPlan &P = c->cells[set];
  for (int i = 0; i < npeers; i++)
    if (so[i] < 0 || sn[i] < 0 || ro[i] < 0 || rn[i] < 0 || so[i] + sn[i] > P.nsend || ro[i] + rn[i] > P.nrecv) {
      set_error("comm_set_cell_counts: peer %d (%d at %d out, %d at %d in) does not fit cell plan %d (%d sent, %d received): call "
                "cup2d_halo_plan_cells first", i, sn[i], so[i], rn[i], ro[i], set, P.nsend, P.nrecv);
      return CUP2D_ERR_ARG;
    }
  rc->csoff[set].assign(so, so + npeers); rc->ccnt[set].assign(sn, sn + npeers);
  rc->croff[set].assign(ro, ro + npeers); rc->crcnt[set].assign(rn, rn + npeers);
  return CUP2D_OK;
}
int cup2d_comm_stats(cup2d_ctx *c, int *nranks, int *npeers, long long *exchanges, long long *allreduces, long long *allgathers) {
  CUP2D_CHECK_CTX(c);
  if (!c->rccl) { set_error("comm_stats: no communicator"); return CUP2D_ERR_ARG; }
  if (nranks) *nranks = c->rccl->nranks;
  if (npeers) *npeers = (int)c->rccl->peer.size();
  if (exchanges) *exchanges = c->rccl->n_exchange;
  if (allreduces) *allreduces = c->rccl->n_allreduce;
  if (allgathers) *allgathers = c->rccl->n_allgather;
  return CUP2D_OK;
}

int cup2d_halo_exchange(cup2d_ctx *c, int field, int width) {
  CUP2D_CHECK_CTX(c);
  if (!field_ok(field) || width < 1 || width > BS) { set_error("halo_exchange: field %d width %d", field, width); return CUP2D_ERR_ARG; }
  if (c->nghost > 0 && !c->exchange) { set_error("halo_exchange: no communicator (cup2d_comm_init / cup2d_set_comm)"); return CUP2D_ERR_COMM; }
  return exchange_halo(c, c->d_field[field], dim_of(field), width);
}

}  // extern "C"
