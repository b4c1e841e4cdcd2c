// api.hip -- the extern "C" surface declared in include/cup2d_hip.h
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "ctx.h"

namespace cup2d {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// Block-Jacobi preconditioner P_inv = -(A_loc)^-1, A_loc = 64x64 in-block Dirichlet 5-point
// Laplacian (4 on the diagonal, -1 towards in-block neighbours): what main.cpp:46-57 + 6451-6488
// build with a Cholesky factorisation.  Here: A = L L^T, then columns of A^-1 by two triangular
// solves.
static void build_P_inv(std::vector<double> &P) {
  const int N = BC;
  std::vector<double> A(N * N, 0.0), L(N * N, 0.0);
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      const int dx = abs(i % BS - j % BS), dy = abs(i / BS - j / BS);
      A[i * N + j] = (dx + dy == 0) ? 4.0 : (dx + dy == 1 ? -1.0 : 0.0);
    }
  for (int j = 0; j < N; j++) {
    double d = A[j * N + j];
    for (int k = 0; k < j; k++) d -= L[j * N + k] * L[j * N + k];
    L[j * N + j] = sqrt(d);
    for (int i = j + 1; i < N; i++) {
      double s = A[i * N + j];
      for (int k = 0; k < j; k++) s -= L[i * N + k] * L[j * N + k];
      L[i * N + j] = s / L[j * N + j];
    }
  }
  P.assign(N * N, 0.0);
  std::vector<double> y(N), x(N);
  for (int col = 0; col < N; col++) {
    for (int i = 0; i < N; i++) {  // L y = e_col
      double s = (i == col) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= L[i * N + k] * y[k];
      y[i] = s / L[i * N + i];
    }
    for (int i = N - 1; i >= 0; i--) {  // L^T x = y
      double s = y[i];
      for (int k = i + 1; k < N; k++) s -= L[k * N + i] * x[k];
      x[i] = s / L[i * N + i];
    }
    for (int i = 0; i < N; i++) P[i * N + col] = -x[i];
  }
  // symmetrise exactly: kernels read P[j][i] for row i
  for (int i = 0; i < N; i++)
    for (int j = i + 1; j < N; j++) {
      const double m = 0.5 * (P[i * N + j] + P[j * N + i]);
      P[i * N + j] = P[j * N + i] = m;
    }
}

static size_t slab_doubles(const cup2d_ctx *c, int dim) { return (size_t)c->ntotal * BC * dim; }

int resident_grid(cup2d_ctx *c, const void *kernel, int count) {
  int per_cu = 0;
  for (auto &e : c->resident)
    if (e.first == kernel) per_cu = e.second;
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WG, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    c->resident.emplace_back(kernel, per_cu);
  }
  const int groups = (count + WPG - 1) / WPG;
  int g = per_cu * (c->num_cus > 0 ? c->num_cus : 256);
  if (g > c->grid) g = c->grid;
  if (g > groups) g = groups;
  if (g >= 8) g -= g % 8;  // equal share per XCD
  return g < 1 ? 1 : g;
}

int prof_resolve(cup2d_ctx *c) {
  if (c->prof_used == 0) return CUP2D_OK;
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < c->prof_used; k++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->prof_ev[2 * k], c->prof_ev[2 * k + 1]) == hipSuccess) {
      c->t_ms[c->prof_id[k]] += ms;
      c->t_calls[c->prof_id[k]]++;
    }
  }
  c->prof_used = 0;
  return CUP2D_OK;
}

// ---- device-memory pool (ctx.h dev_malloc / dev_free) ---------------------------------------------
namespace {
struct DevPool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void *>> idle;  // (device, bucket) -> buffers not in use
  std::map<void *, std::pair<int, size_t>> owner;              // every buffer the pool has handed out or holds
  size_t idle_bytes = 0;
  const bool on = [] { const char *e = getenv("CUP2D_POOL"); return !e || atoi(e) != 0; }();
  const size_t cap = [] { const char *e = getenv("CUP2D_POOL_MAX_GB"); return (size_t)(e ? atof(e) : 64.0) << 30; }();
  static size_t bucket(size_t bytes) {
    if (bytes < 4096) return 4096;
    int k = 0;
    while ((bytes >> k) >= 16) k++;  // bytes >> k in [8, 16)
    const size_t m = (bytes + ((size_t)1 << k) - 1) >> k;
    return m << k;
  }
};
DevPool &pool() {
  static DevPool P;
  return P;
}
}  // namespace

// Streams, events and pinned words of a context, recycled like the device buffers: a regrid destroys one context and
// creates another (adapt(), main.cpp:4657-5440), and hipHostMalloc / hipHostFree / stream and event create / destroy cost
// 3 ms each way (measured, 63 k-block grid: cup2d_destroy "events, stream, delete" 3.2 ms)
namespace {
struct HostRes {
  hipStream_t stream = nullptr;
  hipEvent_t ev[cup2d_ctx::SOLVE_AHEAD] = {nullptr};
  KrylovScalars *h_sc = nullptr;
  double *h_red = nullptr;
  int *h_status = nullptr;
};
struct HostResPool {
  std::mutex mu;
  std::map<int, std::vector<HostRes>> idle;  // per device
  std::map<int, int> num_cus;                // hipGetDeviceProperties is slow: once per device
};
HostResPool &host_pool() {
  static HostResPool P;
  return P;
}
void host_res_destroy(HostRes &r) {
  for (auto &e : r.ev)
    if (e) (void)hipEventDestroy(e);
  if (r.stream) (void)hipStreamDestroy(r.stream);
  if (r.h_sc) (void)hipHostFree(r.h_sc);
  if (r.h_red) (void)hipHostFree(r.h_red);
  if (r.h_status) (void)hipHostFree(r.h_status);
  r = HostRes();
}
}  // namespace

HostStage &host_stage() {
  static HostStage S;
  return S;
}
void *host_stage_reserve(HostStage &S, size_t bytes) {
  if (bytes <= S.cap) return S.p;
  if (S.p) (void)hipHostFree(S.p);
  S.p = nullptr;
  S.cap = 0;
  size_t want = bytes + bytes / 4;  // (a grid that grows a little finds its buffer)
  if (hipHostMalloc(&S.p, want) != hipSuccess) {
    S.p = nullptr;
    return nullptr;
  }
  S.cap = want;
  return S.p;
}

hipError_t dev_malloc_raw(void **p, size_t bytes) {
  DevPool &P = pool();
  if (!P.on) return hipMalloc(p, bytes);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const size_t b = DevPool::bucket(bytes ? bytes : 1);
  void *q = nullptr;
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.idle.find({dev, b});
    if (it != P.idle.end() && !it->second.empty()) {
      q = it->second.back();
      it->second.pop_back();
      P.idle_bytes -= b;
    }
  }
  if (!q) {
    e = hipMalloc(&q, b);
    if (e != hipSuccess) {  // out of memory with buffers idling in the pool: give them back and try once more
      (void)hipGetLastError();
      cup2d_trim_pool();
      e = hipMalloc(&q, b);
      if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.owner[q] = {dev, b};
  }
  // zero-filled, always: a recycled buffer must not look different from a fresh one
  e = hipMemsetAsync(q, 0, b, nullptr);
  if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) return e;
  *p = q;
  return hipSuccess;
}

void dev_free(void *q) {
  if (!q) return;
  DevPool &P = pool();
  std::unique_lock<std::mutex> g(P.mu);
  auto it = P.owner.find(q);
  if (it == P.owner.end()) {  // not ours (pool off)
    g.unlock();
    (void)hipFree(q);
    return;
  }
  if (P.idle_bytes + it->second.second > P.cap) {
    P.owner.erase(it);
    g.unlock();
    (void)hipFree(q);
    return;
  }
  P.idle[it->second].push_back(q);
  P.idle_bytes += it->second.second;
}

// back to the driver at once, past the pool: what a search allocated only to look at (krylov_fused.hip tune_placement) must not
// idle in this process for its lifetime -- other ranks sharing the GPU, torch and RCCL in the same process see it as used
void dev_release(void *q) {
  if (!q) return;
  DevPool &P = pool();
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.owner.find(q);
    if (it != P.owner.end()) P.owner.erase(it);
  }
  (void)hipFree(q);
}

// A pooled temporary of one call.  The pool hands a freed buffer to the next dev_malloc at once and fills it with zeros on
// the null stream, which the (non-blocking) context streams do not wait for: whatever the context has enqueued on the
// buffer must be done before it goes back -- on the error returns as well.
struct DevTmp {
  cup2d_ctx *c;
  void *p = nullptr;
  explicit DevTmp(cup2d_ctx *c_) : c(c_) {}
  DevTmp(const DevTmp &) = delete;
  DevTmp &operator=(const DevTmp &) = delete;
  ~DevTmp() {
    if (!p) return;
    (void)hipStreamSynchronize(c->stream);
    dev_free(p);
  }
  hipError_t alloc(size_t bytes) { return dev_malloc(&p, bytes); }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

}  // namespace cup2d

using namespace cup2d;

extern "C" int cup2d_trim_pool(void) {
  {
    HostResPool &HP = host_pool();
    std::lock_guard<std::mutex> g(HP.mu);
    int cur = -1;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;  // the caller's current device is the caller's (dev_malloc_raw retries on it)
    for (auto &kv : HP.idle) {
      if (kv.second.empty()) continue;
      (void)hipSetDevice(kv.first);
      for (HostRes &r : kv.second) host_res_destroy(r);
      kv.second.clear();
    }
    if (have_cur) (void)hipSetDevice(cur);
  }
  auto &P = cup2d::pool();
  std::vector<void *> drop;
  {
    std::lock_guard<std::mutex> g(P.mu);
    for (auto &kv : P.idle)
      for (void *q : kv.second) {
        P.owner.erase(q);
        drop.push_back(q);
      }
    P.idle.clear();
    P.idle_bytes = 0;
  }
  for (void *q : drop) (void)hipFree(q);
  return CUP2D_OK;
}

extern "C" {

static int create_impl(cup2d_ctx *c, int nblocks, int nghost, int n_inner, const int32_t *nbr, double h, int device);

const char *cup2d_last_error(void) { return g_err; }
const char *cup2d_version(void) { return "cup2d_hip 0.1 (gfx950)"; }

int cup2d_create(cup2d_ctx **out, int nblocks, int nghost, int n_inner, const int32_t *nbr, double h, int device) {
  if (!out || nblocks <= 0 || nghost < 0 || !nbr || !(h > 0) || n_inner < 0 || n_inner > nblocks) {
    set_error("cup2d_create: bad argument");
    return CUP2D_ERR_ARG;
  }
  for (int i = 0; i < 4 * nblocks; i++)
    if (nbr[i] < CUP2D_WALL || nbr[i] >= nblocks + nghost) {
      set_error("cup2d_create: nbr[%d] = %d out of range", i, nbr[i]);
      return CUP2D_ERR_ARG;
    }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("cup2d_create: no HIP device visible");
    return CUP2D_ERR_NODEVICE;
  }
  if (device < 0 || device >= ndev) {
    set_error("cup2d_create: device %d of %d", device, ndev);
    return CUP2D_ERR_ARG;
  }
  CUP2D_HIP_CHECK(hipSetDevice(device));
  cup2d_ctx *c = new cup2d_ctx;
  StageClock clk("cup2d_create");
  const int st = create_impl(c, nblocks, nghost, n_inner, nbr, h, device);
  clk.lap("create_impl");
  if (st != CUP2D_OK) {  // nothing of a half-built context leaks: cup2d_destroy frees whatever was allocated
    cup2d_destroy(c);
    return st;
  }
  *out = c;
  return CUP2D_OK;
}
static int create_impl(cup2d_ctx *c, int nblocks, int nghost, int n_inner, const int32_t *nbr, double h, int device) {
  c->device = device;
  c->nblocks = nblocks;
  c->nghost = nghost;
  c->ntotal = nblocks + nghost;
  c->n_inner = n_inner;
  c->h = h;
  c->grid = MAX_GRID;
  HostRes res;
  bool recycled = false;
  {
    HostResPool &HP = host_pool();
    std::lock_guard<std::mutex> g(HP.mu);
    auto it = HP.num_cus.find(device);
    if (it != HP.num_cus.end()) c->num_cus = it->second;
    auto &v = HP.idle[device];
    if (!v.empty()) {
      res = v.back();
      v.pop_back();
      recycled = true;
    }
  }
  if (c->num_cus <= 0) {
    hipDeviceProp_t prop;
    CUP2D_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    HostResPool &HP = host_pool();
    std::lock_guard<std::mutex> g(HP.mu);
    HP.num_cus[device] = c->num_cus;
  }
  if (recycled) {  // handed over whole: cup2d_destroy returns (or frees) whatever the context holds
    c->own_stream = res.stream;
    c->h_sc = res.h_sc;
    c->h_red = res.h_red;
    c->h_status = res.h_status;
    for (int i = 0; i < cup2d_ctx::SOLVE_AHEAD; i++) c->solve_ev[i] = res.ev[i];
  } else {
    CUP2D_HIP_CHECK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  }
  c->stream = c->own_stream;
  CUP2D_HIP_CHECK(dev_malloc(&c->d_nbr, sizeof(int32_t) * 4 * nblocks));
  CUP2D_HIP_CHECK(hipMemcpy(c->d_nbr, nbr, sizeof(int32_t) * 4 * nblocks, hipMemcpyHostToDevice));
  c->h_nbr.assign(nbr, nbr + 4 * (size_t)nblocks);
  for (int f = 0; f < CUP2D_NFIELDS; f++) {
    const size_t bytes = slab_doubles(c, dim_of(f)) * sizeof(double);
    CUP2D_HIP_CHECK(dev_malloc(&c->d_field[f], bytes));
    CUP2D_HIP_CHECK(hipMemset(c->d_field[f], 0, bytes));  // calloc, main.cpp:6517
  }
  CUP2D_HIP_CHECK(dev_malloc(&c->d_vscratch, slab_doubles(c, 2) * sizeof(double)));
  CUP2D_HIP_CHECK(hipMemset(c->d_vscratch, 0, slab_doubles(c, 2) * sizeof(double)));
  double **kv[] = {&c->d_r, &c->d_rhat, &c->d_p, &c->d_nu, &c->d_t, &c->d_z, &c->d_z2, &c->d_xopt};
  for (double **p : kv) {
    CUP2D_HIP_CHECK(dev_malloc(p, slab_doubles(c, 1) * sizeof(double)));
    CUP2D_HIP_CHECK(hipMemset(*p, 0, slab_doubles(c, 1) * sizeof(double)));
  }
  // CUP2D_POISON_GHOSTS=1 (tests): the ghost blocks of every field and Krylov vector start as NaN instead of zero, so that a
  // kernel reading a ghost cell no exchange delivered (cup2d_halo_plan_cells lists too little) cannot pass a parity test
  static const bool poison = [] { const char *e = getenv("CUP2D_POISON_GHOSTS"); return e && atoi(e) != 0; }();
  if (poison && nghost > 0) {
    const auto ghosts_nan = [&](double *q, int dim) {
      return hipMemset(q + (size_t)nblocks * BC * dim, 0xFF, (size_t)nghost * BC * dim * sizeof(double));
    };
    for (int f = 0; f < CUP2D_NFIELDS; f++) CUP2D_HIP_CHECK(ghosts_nan(c->d_field[f], dim_of(f)));
    for (double **q : kv) CUP2D_HIP_CHECK(ghosts_nan(*q, 1));
  }
  build_P_inv(c->h_Pinv);
  CUP2D_HIP_CHECK(dev_malloc(&c->d_Pinv, BC * BC * sizeof(double)));
  // the dense kernels read d_Pinv[k][n] as the coefficient of input k in output n, i.e. the transpose of
  // the row-major matrix the caller means (cuda.cu:484-486 Dgemm(T,N) on a column-major view); the
  // built-in matrix is exactly symmetric
  CUP2D_HIP_CHECK(hipMemcpy(c->d_Pinv, c->h_Pinv.data(), BC * BC * sizeof(double), hipMemcpyHostToDevice));
  {  // T = tridiag(-1, 2, -1) (8x8) = Q diag(lam) Q^T: the factors of A_loc = T (x) I + I (x) T
    double fd[BC + BS];
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < BS; i++) {
      for (int k = 0; k < BS; k++) fd[i * BS + k] = sqrt(2.0 / (BS + 1)) * sin((i + 1) * (k + 1) * pi / (BS + 1));
      fd[BC + i] = 2.0 - 2.0 * cos((i + 1) * pi / (BS + 1));
    }
    for (int i = 0; i < BS; i++)  // exact symmetry
      for (int k = i + 1; k < BS; k++) fd[k * BS + i] = fd[i * BS + k];
    CUP2D_HIP_CHECK(dev_malloc(&c->d_fd, sizeof fd));
    CUP2D_HIP_CHECK(hipMemcpy(c->d_fd, fd, sizeof fd, hipMemcpyHostToDevice));
  }
  CUP2D_HIP_CHECK(dev_malloc(&c->d_partials, sizeof(double) * NSLOT * PSTRIDE));
  CUP2D_HIP_CHECK(dev_malloc(&c->d_red_own, sizeof(double) * 8));
  c->d_red = c->d_red_own;
  CUP2D_HIP_CHECK(dev_malloc(&c->d_ticket, sizeof(unsigned)));
  CUP2D_HIP_CHECK(hipMemset(c->d_ticket, 0, sizeof(unsigned)));
  CUP2D_HIP_CHECK(dev_malloc(&c->d_sc, sizeof(KrylovScalars)));
  CUP2D_HIP_CHECK(dev_malloc(&c->d_sc2, sizeof(KrylovScalars)));
  if (!recycled) {
    CUP2D_HIP_CHECK(hipHostMalloc(&c->h_sc, sizeof(KrylovScalars)));
    CUP2D_HIP_CHECK(hipHostMalloc(&c->h_red, sizeof(double) * 8));
    CUP2D_HIP_CHECK(hipHostMalloc(&c->h_status, sizeof(int) * cup2d_ctx::SOLVE_AHEAD));
    for (auto &e : c->solve_ev) CUP2D_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  ::memset(c->h_sc, 0, sizeof(KrylovScalars));
  ::memset(c->h_red, 0, sizeof(double) * 8);
  ::memset(c->h_status, 0, sizeof(int) * cup2d_ctx::SOLVE_AHEAD);
  return CUP2D_OK;
}

void cup2d_destroy(cup2d_ctx *c) {
  if (!c) return;
  StageClock clk("cup2d_destroy");
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  clk.lap("device sync");
  (void)comm_finalize_impl(c);
  bodies_release(c);
  walk_plans_release(c);
  dev_free(c->d_nbr);
  for (int f = 0; f < CUP2D_NFIELDS; f++) dev_free(c->d_field[f]);
  dev_free(c->d_vscratch);
  dev_free(c->d_ko_scratch);
  if (c->vec_arena) {  // the solver's eleven vectors are pieces of one allocation (krylov_fused.hip tune_placement)
    double **piece[] = {&c->d_r, &c->d_s, &c->d_p, &c->d_p2, &c->d_nu, &c->d_nu2, &c->d_t, &c->d_y, &c->d_yopt, &c->d_xopt, &c->d_rhat};
    for (double **q : piece) *q = nullptr;
    (void)hipFree(c->vec_arena);
    c->vec_arena = nullptr;
  }
  double *kv[] = {c->d_r, c->d_rhat, c->d_p, c->d_nu, c->d_t, c->d_z, c->d_z2, c->d_xopt, c->d_Pinv, c->d_fd, c->d_partials, c->d_red_own};
  for (double *p : kv) dev_free(p);
  double *fv[] = {c->d_p2, c->d_nu2, c->d_s, c->d_y, c->d_yopt};
  for (double *p : fv) dev_free(p);
  dev_free(c->d_fault);
  dev_free(c->d_ticket);
  dev_free(c->d_sc);
  dev_free(c->d_sc2);
  {  // stream, events and pinned words go back to the per-device free list (complete sets only; a few are kept)
    HostRes res;
    res.stream = c->own_stream;
    res.h_sc = c->h_sc;
    res.h_red = c->h_red;
    res.h_status = c->h_status;
    bool complete = res.stream && res.h_sc && res.h_red && res.h_status;
    for (int i = 0; i < cup2d_ctx::SOLVE_AHEAD; i++) {
      res.ev[i] = c->solve_ev[i];
      complete = complete && res.ev[i];
    }
    bool kept = false;
    if (complete) {
      HostResPool &HP = host_pool();
      std::lock_guard<std::mutex> g(HP.mu);
      auto &v = HP.idle[c->device];
      if (v.size() < 8) {
        v.push_back(res);
        kept = true;
      }
    }
    if (!kept) host_res_destroy(res);
    c->own_stream = nullptr;
  }
  dev_free(c->mat.d_ptr); dev_free(c->mat.d_col); dev_free(c->mat.d_val); dev_free(c->mat.d_gather);
  dev_free(c->mat.d_reg);
  dev_free(c->amr.d_level); dev_free(c->amr.d_kind); dev_free(c->amr.d_nbr2); dev_free(c->amr.d_half);
  dev_free(c->amr.d_faces); dev_free(c->amr.d_faces2);
  amr_phase_release(c);
  for (auto &cp : c->cells) { dev_free(cp.d_send); dev_free(cp.d_recv); cp = cup2d::CellPlan(); }
  dev_free(c->plan.d_send_block); dev_free(c->plan.d_send_face);
  dev_free(c->plan.d_recv_block); dev_free(c->plan.d_recv_face);
  clk.lap("frees");
  for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  clk.lap("events, stream, delete");
}

int cup2d_set_stream(cup2d_ctx *c, void *s) {
  CUP2D_CHECK_CTX(c);
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return CUP2D_OK;
}
int cup2d_get_stream(cup2d_ctx *c, void **s) {
  CUP2D_CHECK_CTX(c);
  if (!s) return CUP2D_ERR_ARG;
  *s = (void *)c->stream;
  return CUP2D_OK;
}
int cup2d_synchronize(cup2d_ctx *c) {
  CUP2D_CHECK_CTX(c);
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_set_math(cup2d_ctx *c, int math) {
  CUP2D_CHECK_CTX(c);
  if (math != CUP2D_MATH_FAST && math != CUP2D_MATH_STRICT) { set_error("bad math mode %d", math); return CUP2D_ERR_ARG; }
  c->math = math;
  return CUP2D_OK;
}

#define CHECK_FIELD(f)                                                   \
  if (!field_ok(f)) { set_error("%s: bad field %d", __func__, f); return CUP2D_ERR_ARG; }

int cup2d_upload(cup2d_ctx *c, int field, const double *const *blocks) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!blocks) return CUP2D_ERR_ARG;
  const size_t per = (size_t)BC * dim_of(field);
  std::vector<double> stage((size_t)c->nblocks * per);
  for (int b = 0; b < c->nblocks; b++) memcpy(&stage[b * per], blocks[b], per * sizeof(double));
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_field[field], stage.data(), stage.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_download(cup2d_ctx *c, int field, double *const *blocks) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!blocks) return CUP2D_ERR_ARG;
  const size_t per = (size_t)BC * dim_of(field);
  std::vector<double> stage((size_t)c->nblocks * per);
  CUP2D_HIP_CHECK(hipMemcpyAsync(stage.data(), c->d_field[field], stage.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  for (int b = 0; b < c->nblocks; b++) memcpy(blocks[b], &stage[b * per], per * sizeof(double));
  return CUP2D_OK;
}
int cup2d_upload_slab(cup2d_ctx *c, int field, const double *slab) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!slab) return CUP2D_ERR_ARG;
  const size_t bytes = (size_t)c->nblocks * BC * dim_of(field) * sizeof(double);
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_field[field], slab, bytes, hipMemcpyHostToDevice, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_download_slab(cup2d_ctx *c, int field, double *slab) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!slab) return CUP2D_ERR_ARG;
  const size_t bytes = (size_t)c->nblocks * BC * dim_of(field) * sizeof(double);
  CUP2D_HIP_CHECK(hipMemcpyAsync(slab, c->d_field[field], bytes, hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
// ---- block-wise transfers (a regridding host moves only the blocks that change, the rest stays on the device) ----
// one wave per block, `per` doubles per lane
__global__ void k_blocks_gather(const double *__restrict__ f, const int32_t *__restrict__ idx, double *__restrict__ out, int n, int per) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < n; k += gridDim.x * (blockDim.x >> 6)) {
    const size_t src = (size_t)idx[k] * 64 * per, dst = (size_t)k * 64 * per;
    for (int j = 0; j < per; j++) out[dst + j * 64 + lane] = f[src + j * 64 + lane];
  }
}
__global__ void k_blocks_scatter(double *__restrict__ f, const int32_t *__restrict__ didx, const double *__restrict__ in,
                                 const int32_t *__restrict__ sidx, int n, int per) {
  const int lane = threadIdx.x & 63;
  for (int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < n; k += gridDim.x * (blockDim.x >> 6)) {
    const size_t dst = (size_t)didx[k] * 64 * per, src = (size_t)(sidx ? sidx[k] : k) * 64 * per;
    for (int j = 0; j < per; j++) f[dst + j * 64 + lane] = in[src + j * 64 + lane];
  }
}
static int blocks_grid(int n) { return n < 4 ? 1 : (n / 4 > 2048 ? 2048 : n / 4); }
static int check_block_list(const cup2d_ctx *c, int n, const int32_t *blocks, const char *what) {
  if (n < 0 || (n && !blocks)) { set_error("%s: block list", what); return CUP2D_ERR_ARG; }
  for (int k = 0; k < n; k++)
    if (blocks[k] < 0 || blocks[k] >= c->nblocks) { set_error("%s: block %d of %d", what, blocks[k], c->nblocks); return CUP2D_ERR_ARG; }
  return CUP2D_OK;
}
int cup2d_download_blocks(cup2d_ctx *c, int field, int n, const int32_t *blocks, double *host) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  CUP2D_TRY(check_block_list(c, n, blocks, "download_blocks"));
  if (n == 0) return CUP2D_OK;
  if (!host) return CUP2D_ERR_ARG;
  const int per = dim_of(field);
  const size_t bytes = (size_t)n * BC * per * sizeof(double);
  DevTmp t_idx(c), t_buf(c);
  CUP2D_HIP_CHECK(t_idx.alloc((size_t)n * sizeof(int32_t)));
  CUP2D_HIP_CHECK(t_buf.alloc(bytes));
  int32_t *d_idx = t_idx.as<int32_t>();
  double *d_buf = t_buf.as<double>();
  CUP2D_HIP_CHECK(hipMemcpyAsync(d_idx, blocks, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_blocks_gather, dim3(blocks_grid(n)), dim3(WG), 0, c->stream, c->d_field[field], d_idx, d_buf, n, per);
  CUP2D_HIP_CHECK(hipGetLastError());
  CUP2D_HIP_CHECK(hipMemcpyAsync(host, d_buf, bytes, hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_upload_blocks(cup2d_ctx *c, int field, int n, const int32_t *blocks, const double *host) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  CUP2D_TRY(check_block_list(c, n, blocks, "upload_blocks"));
  if (n == 0) return CUP2D_OK;
  if (!host) return CUP2D_ERR_ARG;
  const int per = dim_of(field);
  const size_t bytes = (size_t)n * BC * per * sizeof(double);
  DevTmp t_idx(c), t_buf(c);
  CUP2D_HIP_CHECK(t_idx.alloc((size_t)n * sizeof(int32_t)));
  CUP2D_HIP_CHECK(t_buf.alloc(bytes));
  int32_t *d_idx = t_idx.as<int32_t>();
  double *d_buf = t_buf.as<double>();
  CUP2D_HIP_CHECK(hipMemcpyAsync(d_idx, blocks, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  CUP2D_HIP_CHECK(hipMemcpyAsync(d_buf, host, bytes, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_blocks_scatter, dim3(blocks_grid(n)), dim3(WG), 0, c->stream, c->d_field[field], d_idx, d_buf,
                     (const int32_t *)nullptr, n, per);
  CUP2D_HIP_CHECK(hipGetLastError());
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_copy_blocks(cup2d_ctx *c, cup2d_ctx *src, int field, int n, const int32_t *dst_blocks, const int32_t *src_blocks) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!src || src->device != c->device) { set_error("copy_blocks: the source context must live on the same device"); return CUP2D_ERR_ARG; }
  CUP2D_TRY(check_block_list(c, n, dst_blocks, "copy_blocks (destination)"));
  CUP2D_TRY(check_block_list(src, n, src_blocks, "copy_blocks (source)"));
  if (n == 0) return CUP2D_OK;
  DevTmp t_idx(c);
  CUP2D_HIP_CHECK(t_idx.alloc((size_t)2 * n * sizeof(int32_t)));
  int32_t *d_idx = t_idx.as<int32_t>();
  CUP2D_HIP_CHECK(hipStreamSynchronize(src->stream));  // what the source context has enqueued is done
  CUP2D_HIP_CHECK(hipMemcpyAsync(d_idx, dst_blocks, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  CUP2D_HIP_CHECK(hipMemcpyAsync(d_idx + n, src_blocks, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_blocks_scatter, dim3(blocks_grid(n)), dim3(WG), 0, c->stream, c->d_field[field], d_idx,
                     (const double *)src->d_field[field], (const int32_t *)(d_idx + n), n, dim_of(field));
  CUP2D_HIP_CHECK(hipGetLastError());
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_field_ptr(cup2d_ctx *c, int field, void **p) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!p) return CUP2D_ERR_ARG;
  *p = c->d_field[field];
  if (field == CUP2D_VEL) c->vel_ptr_exposed = true;  // the caller may write the velocity behind the library's back
  return CUP2D_OK;
}
__global__ void k_fill(double *p, double v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int cup2d_fill(cup2d_ctx *c, int field, double value) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  const size_t n = slab_doubles(c, dim_of(field));
  if (value == 0.0) {
    CUP2D_HIP_CHECK(hipMemsetAsync(c->d_field[field], 0, n * sizeof(double), c->stream));
  } else {
    hipLaunchKernelGGL(k_fill, dim3(c->grid), dim3(WG), 0, c->stream, c->d_field[field], value, n);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  return CUP2D_OK;
}
int cup2d_copy_field(cup2d_ctx *c, int dst, int src) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(dst);
  CHECK_FIELD(src);
  if (dim_of(dst) != dim_of(src)) { set_error("copy_field: dim mismatch"); return CUP2D_ERR_ARG; }
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_field[dst], c->d_field[src], slab_doubles(c, dim_of(src)) * sizeof(double),
                                 hipMemcpyDeviceToDevice, c->stream));
  return CUP2D_OK;
}

// ---- block-AMR topology -------------------------------------------------------------------------
int cup2d_set_amr(cup2d_ctx *c, double h0, const int32_t *level, const int32_t *kind, const int32_t *nbr2,
                  const int32_t *half) {
  CUP2D_CHECK_CTX(c);
  if (!(h0 > 0) || !level || !kind || !nbr2 || !half) { set_error("set_amr: bad argument"); return CUP2D_ERR_ARG; }
  // with ghost blocks (N ranks) the tables cover owned + ghost blocks: a ghost block's entries are read where a kernel
  // looks across a COARSER neighbour's tangential side (halo-3 tile, amr.hip); sides of ghost blocks whose neighbour this
  // rank does not hold are passed as CUP2D_AMR_WALL and never read
  const int nb = c->ntotal;
  for (int b = 0; b < nb; b++) {
    if (level[b] < 0 || level[b] > 30) { set_error("set_amr: level[%d] = %d", b, level[b]); return CUP2D_ERR_ARG; }
    for (int s = 0; s < 4; s++) {
      const int k = kind[4 * b + s], n0 = nbr2[(4 * b + s) * 2], n1 = nbr2[(4 * b + s) * 2 + 1];
      bool ok = k >= CUP2D_AMR_WALL && k <= CUP2D_AMR_FINER;
      if (ok && k != CUP2D_AMR_WALL) ok = n0 >= 0 && n0 < nb;
      if (ok && k == CUP2D_AMR_SAME) ok = level[n0] == level[b];
      if (ok && k == CUP2D_AMR_COARSER) ok = level[n0] == level[b] - 1 && (half[4 * b + s] == 0 || half[4 * b + s] == 1);
      if (ok && k == CUP2D_AMR_FINER) ok = n1 >= 0 && n1 < nb && level[n0] == level[b] + 1 && level[n1] == level[b] + 1;
      if (!ok) { set_error("set_amr: block %d side %d: kind %d neighbours %d %d", b, s, k, n0, n1); return CUP2D_ERR_ARG; }
    }
  }
  StageClock clk("cup2d_set_amr");
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  cup2d::AmrTopo &A = c->amr;
  dev_free(A.d_level); dev_free(A.d_kind); dev_free(A.d_nbr2); dev_free(A.d_half); dev_free(A.d_faces); dev_free(A.d_faces2);
  amr_phase_release(c);
  A = cup2d::AmrTopo();
  CUP2D_HIP_CHECK(dev_malloc(&A.d_level, sizeof(int32_t) * nb));
  CUP2D_HIP_CHECK(dev_malloc(&A.d_kind, sizeof(int32_t) * nb * 4));
  CUP2D_HIP_CHECK(dev_malloc(&A.d_nbr2, sizeof(int32_t) * nb * 8));
  CUP2D_HIP_CHECK(dev_malloc(&A.d_half, sizeof(int32_t) * nb * 4));
  CUP2D_HIP_CHECK(dev_malloc(&A.d_faces, sizeof(double) * nb * 4 * BS));
  CUP2D_HIP_CHECK(hipMemcpy(A.d_level, level, sizeof(int32_t) * nb, hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(A.d_kind, kind, sizeof(int32_t) * nb * 4, hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(A.d_nbr2, nbr2, sizeof(int32_t) * nb * 8, hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemcpy(A.d_half, half, sizeof(int32_t) * nb * 4, hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(hipMemset(A.d_faces, 0, sizeof(double) * nb * 4 * BS));
  CUP2D_HIP_CHECK(dev_malloc(&A.d_faces2, sizeof(double) * nb * 4 * BS * 2));
  CUP2D_HIP_CHECK(hipMemset(A.d_faces2, 0, sizeof(double) * nb * 4 * BS * 2));
  A.h_kind.assign(kind, kind + (size_t)nb * 4);
  A.h_nbr2.assign(nbr2, nbr2 + (size_t)nb * 8);
  A.h_half.assign(half, half + (size_t)nb * 4);
  A.h_level.assign(level, level + (size_t)nb);
  A.h0 = h0;
  int lmax = 0;
  for (int b = 0; b < nb; b++) lmax = level[b] > lmax ? level[b] : lmax;
  A.h_min = h0 / (double)(1 << lmax);  // N ranks: the finest level of the WHOLE grid, cup2d_amr_set_finest_level
  A.active = true;
  clk.lap("tables to the device");
  return CUP2D_OK;
}
int cup2d_amr_set_finest_level(cup2d_ctx *c, int level_finest) {
  CUP2D_CHECK_CTX(c);
  if (!c->amr.active || level_finest < 0 || level_finest > 30) { set_error("amr_set_finest_level: cup2d_set_amr first; 0 <= level <= 30"); return CUP2D_ERR_ARG; }
  c->amr.h_min = c->amr.h0 / (double)(1 << level_finest);
  return CUP2D_OK;
}
// Adapted grids take the three phases as well (computeA's split works on any grid, main.cpp:3035-3057): _ALL refreshes the
// ghost copies itself -- the blocks that read no ghost block are swept while they travel --, _INNER / _HALO are its two halves
// for a caller that exchanges in between (cup2d_halo_exchange); the flux correction of ALL blocks rides on _HALO (amr.hip)
#define AMR_PHASE_OK(c, phase)                                                                  \
  if ((phase) != CUP2D_BLOCKS_ALL && (phase) != CUP2D_BLOCKS_INNER && (phase) != CUP2D_BLOCKS_HALO) { \
    set_error("%s: phase %d", __func__, (int)(phase));                                          \
    return CUP2D_ERR_ARG;                                                                       \
  }

// ---- block operators --------------------------------------------------------------------------
int cup2d_advect_diffuse_rhs(cup2d_ctx *c, double nu, double dt, int phase) {
  CUP2D_CHECK_CTX(c);
  AMR_PHASE_OK(c, phase);
  if (c->amr.active) return amr_advect_diffuse_rhs(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMPV], nu, dt, phase);
  int first, count;
  CUP2D_TRY(phase_range(c, phase, &first, &count));
  return launch_advect(c, c->d_field[CUP2D_VEL], nullptr, c->d_field[CUP2D_TMPV], 0, nu, dt, 0.0, first, count);
}
int cup2d_advect_diffuse_stage(cup2d_ctx *c, double nu, double dt, int stage, int phase) {
  CUP2D_CHECK_CTX(c);
  int first, count;
  CUP2D_TRY(phase_range(c, phase, &first, &count));
  if (c->amr.active) {  // the reference's un-fused stage (flux correction between the functor and the update)
    if (phase != CUP2D_BLOCKS_ALL) { set_error("advect_diffuse_stage: a whole stage on an adapted grid takes CUP2D_BLOCKS_ALL (cup2d_advect_diffuse_rhs has the phases)"); return CUP2D_ERR_ARG; }
    if (stage != 1 && stage != 2) { set_error("advect_diffuse_stage: stage %d", stage); return CUP2D_ERR_ARG; }
    return amr_advect_diffuse_stage(c, nu, dt, stage);
  }
  const double ih2 = 1.0 / (c->h * c->h);
  if (stage == 1)  // main.cpp:6616-6626: mid = vel + (0.5/h^2) rhs(vel)
    return launch_advect(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_VEL], c->d_vscratch, 1, nu, dt, 0.5 / (c->h * c->h),
                         first, count);
  if (stage == 2)  // main.cpp:6632-6642: vel = vold + (1/h^2) rhs(mid); vold is the unmodified vel
    return launch_advect(c, c->d_vscratch, c->d_field[CUP2D_VEL], c->d_field[CUP2D_VEL], 1, nu, dt, ih2, first, count);
  set_error("advect_diffuse_stage: stage %d", stage);
  return CUP2D_ERR_ARG;
}
int cup2d_advect_diffuse_rk2(cup2d_ctx *c, double nu, double dt) {
  CUP2D_CHECK_CTX(c);
  if (c->amr.active) return amr_advect_diffuse_rk2(c, nu, dt);
  for (int stage = 1; stage <= 2; stage++) {
    double *src = stage == 1 ? c->d_field[CUP2D_VEL] : c->d_vscratch;
    if (overlapped(c)) {  // inner blocks while the face strips are in flight (main.cpp:3035-3057)
      CUP2D_TRY(exchange_begin(c, src, 2, 3));
      CUP2D_TRY(cup2d_advect_diffuse_stage(c, nu, dt, stage, CUP2D_BLOCKS_INNER));
      CUP2D_TRY(exchange_end(c, src, 2, 3));
      CUP2D_TRY(cup2d_advect_diffuse_stage(c, nu, dt, stage, CUP2D_BLOCKS_HALO));
    } else {
      CUP2D_TRY(exchange_halo(c, src, 2, 3));
      CUP2D_TRY(cup2d_advect_diffuse_stage(c, nu, dt, stage, CUP2D_BLOCKS_ALL));
    }
  }
  return CUP2D_OK;
}
int cup2d_vorticity(cup2d_ctx *c, int phase) {
  CUP2D_CHECK_CTX(c);
  int first, count;
  CUP2D_TRY(phase_range(c, phase, &first, &count));
  if (c->amr.active) return amr_vorticity(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMP], phase);
  return launch_vorticity(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMP], first, count);
}
int cup2d_pressure_rhs(cup2d_ctx *c, double dt, int use_bodies, int phase) {
  CUP2D_CHECK_CTX(c);
  int first, count;
  CUP2D_TRY(phase_range(c, phase, &first, &count));
  if (!(dt > 0)) { set_error("pressure_rhs: dt"); return CUP2D_ERR_ARG; }
  if (c->amr.active)  // the chi / udef terms are always evaluated (chi = 0 without bodies), as the reference does
    return amr_pressure_rhs(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMPV], c->d_field[CUP2D_CHI], c->d_field[CUP2D_TMP], dt, phase);
  return launch_pressure_rhs(c, c->d_field[CUP2D_VEL], use_bodies ? c->d_field[CUP2D_TMPV] : nullptr,
                             use_bodies ? c->d_field[CUP2D_CHI] : nullptr, nullptr, c->d_field[CUP2D_TMP], dt, first, count);
}
int cup2d_laplacian_sub(cup2d_ctx *c, int phase) {
  CUP2D_CHECK_CTX(c);
  int first, count;
  CUP2D_TRY(phase_range(c, phase, &first, &count));
  if (c->amr.active) return amr_laplacian(c, c->d_field[CUP2D_POLD], c->d_field[CUP2D_TMP], 1, phase);
  return launch_laplacian(c, c->d_field[CUP2D_POLD], c->d_field[CUP2D_TMP], 1, first, count);
}
// zero_pres = false (cup2d_step only): the solve that follows is told that its initial guess is zero (ctx.h x0_is_zero)
// and neither reads nor needs the fill -- pres receives the solution
static int poisson_rhs_uniform(cup2d_ctx *c, double dt, int use_bodies, bool zero_pres) {
  double *pres = c->d_field[CUP2D_PRES], *pold = c->d_field[CUP2D_POLD];
  if (!zero_pres) {
    // cup2d_step: the rhs kernel reads pres AS pold and writes the block's own cells to POLD on its way (pold = pres
    // without a pass of its own); pres is left as it is -- the solve does not read it and overwrites it
    CUP2D_TRY(exchange_halo(c, c->d_field[CUP2D_VEL], 2, 1));
    CUP2D_TRY(exchange_halo(c, pres, 1, 1));
    if (use_bodies) CUP2D_TRY(exchange_halo(c, c->d_field[CUP2D_TMPV], 2, 1));
    return launch_pressure_rhs(c, c->d_field[CUP2D_VEL], use_bodies ? c->d_field[CUP2D_TMPV] : nullptr,
                               use_bodies ? c->d_field[CUP2D_CHI] : nullptr, pres, c->d_field[CUP2D_TMP], dt, 0, c->nblocks, pold);
  }
  // pold = pres; pres = 0 (main.cpp:7016-7021): a device copy, not a pointer swap -- the slab pointers a caller got
  // from cup2d_field_ptr stay valid for the life of the context
  CUP2D_HIP_CHECK(hipMemcpyAsync(pold, pres, (size_t)c->nblocks * BC * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  CUP2D_TRY(launch_zero(c, pres, slab_doubles(c, 1)));
  CUP2D_TRY(exchange_halo(c, c->d_field[CUP2D_VEL], 2, 1));
  CUP2D_TRY(exchange_halo(c, pold, 1, 1));
  if (use_bodies) CUP2D_TRY(exchange_halo(c, c->d_field[CUP2D_TMPV], 2, 1));
  return launch_pressure_rhs(c, c->d_field[CUP2D_VEL], use_bodies ? c->d_field[CUP2D_TMPV] : nullptr,
                             use_bodies ? c->d_field[CUP2D_CHI] : nullptr, pold, c->d_field[CUP2D_TMP], dt, 0, c->nblocks);
}
int cup2d_poisson_rhs(cup2d_ctx *c, double dt, int use_bodies) {
  CUP2D_CHECK_CTX(c);
  if (!(dt > 0)) { set_error("poisson_rhs: dt"); return CUP2D_ERR_ARG; }
  if (c->amr.active) return amr_poisson_rhs(c, dt);
  return poisson_rhs_uniform(c, dt, use_bodies, true);
}
int cup2d_pressure_correction(cup2d_ctx *c, double dt, int phase) {
  CUP2D_CHECK_CTX(c);
  int first, count;
  CUP2D_TRY(phase_range(c, phase, &first, &count));
  if (c->amr.active) return amr_pressure_correction(c, c->d_field[CUP2D_PRES], c->d_field[CUP2D_TMPV], dt, phase);
  return launch_pressure_correction(c, c->d_field[CUP2D_PRES], c->d_field[CUP2D_TMPV], nullptr, dt, 0, first, count);
}
int cup2d_add_correction(cup2d_ctx *c) {
  CUP2D_CHECK_CTX(c);
  return launch_axpy_field(c, c->d_field[CUP2D_VEL], c->d_field[CUP2D_TMPV], 1.0 / c->h / c->h,
                           (size_t)c->nblocks * BC * 2);
}
int cup2d_project(cup2d_ctx *c, double dt) {
  CUP2D_CHECK_CTX(c);
  if (c->amr.active) return amr_project(c, dt);
  ProfScope t(c, CUP2D_T_PROJECT);
  return project_impl(c, dt);
}

// ---- scalars ----------------------------------------------------------------------------------
int cup2d_max_abs_vel(cup2d_ctx *c, double *umax) {
  CUP2D_CHECK_CTX(c);
  if (!umax) return CUP2D_ERR_ARG;
  if (c->use_cached_umax && c->umax_on_host) {  // ... already reduced and copied behind that projection (cup2d_step)
    *umax = c->h_red[6];
    return CUP2D_OK;
  }
  if (c->use_cached_umax)  // the maxima the previous step's projection left (ctx.h umax_partials)
    CUP2D_TRY(launch_max_from_partials(c, c->d_partials + (size_t)3 * PSTRIDE, c->umax_partials, c->d_red));
  else
    CUP2D_TRY(launch_max_abs(c, c->d_field[CUP2D_VEL], (size_t)c->nblocks * BC * 2, c->d_red));
  if (c->allreduce && c->allreduce(c->comm_user, c->d_red, 1, 1, c->stream) != 0) return CUP2D_ERR_COMM;
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->h_red, c->d_red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  *umax = c->h_red[0];
  return CUP2D_OK;
}
int cup2d_block_linf(cup2d_ctx *c, int field, double *linf) {
  CUP2D_CHECK_CTX(c);
  if (!linf || field < 0 || field >= CUP2D_NFIELDS || dim_of(field) != 1) {
    set_error("block_linf: a scalar field and an output array expected");
    return CUP2D_ERR_ARG;
  }
  double *d_out = c->d_t;  // a solver work vector: at least one double per block, free between solves
  CUP2D_TRY(launch_block_linf(c, c->d_field[field], d_out));
  CUP2D_HIP_CHECK(hipMemcpyAsync(linf, d_out, (size_t)c->nblocks * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  return CUP2D_OK;
}
int cup2d_compute_dt(cup2d_ctx *c, double nu, double cfl, double *dt) {
  CUP2D_CHECK_CTX(c);
  if (!dt) return CUP2D_ERR_ARG;
  double umax = 0;
  CUP2D_TRY(cup2d_max_abs_vel(c, &umax));
  const double h = c->amr.active ? c->amr.h_min : c->h;  // the finest cell size (main.cpp:6580-6583)
  const double dtDiffusion = 0.25 * h * h / (nu + 0.25 * h * umax);
  const double dtAdvection = h / (umax + 1e-8);
  *dt = fmin(dtDiffusion, cfl * dtAdvection);
  return CUP2D_OK;
}

// ---- Poisson ----------------------------------------------------------------------------------
// An adapted grid solves with the operator of main.cpp:7034-7112.  A caller that has not installed one gets the library's own
// assembly from the topology tables (cup2d_amr_install_poisson: stored rows only where the grid is irregular) -- on one rank;
// on N ranks the gather list of the exchange is the caller's to give (cup2d_set_gather), so nothing can be installed for it.
static int amr_operator_ready(cup2d_ctx *c, const char *who) {
  if (!c->amr.active || c->mat.active) return CUP2D_OK;
  if (c->nghost > 0) {
    set_error("%s: adapted grid on N ranks without an operator: cup2d_amr_install_poisson (or cup2d_set_matrix_coo) and cup2d_set_gather first", who);
    return CUP2D_ERR_UNSUPPORTED;
  }
  return cup2d_amr_install_poisson(c);
}
int cup2d_poisson_solve(cup2d_ctx *c, double max_error, double max_rel_error, int max_restarts, int max_iter, int *iters,
                        int *restarts, double *linf, double *linf_init) {
  CUP2D_CHECK_CTX(c);
  if (max_iter < 0) { set_error("poisson_solve: max_iter"); return CUP2D_ERR_ARG; }
  CUP2D_TRY(amr_operator_ready(c, "poisson_solve"));
  c->last_solver = (c->solver == CUP2D_SOLVER_FUSED && fused_supported(c)) ? CUP2D_SOLVER_FUSED : CUP2D_SOLVER_SWEEPS;
  if (c->last_solver == CUP2D_SOLVER_FUSED)
    return solve_fused_impl(c, max_error, max_rel_error, max_restarts, max_iter, iters, restarts, linf, linf_init);
  return solve_impl(c, max_error, max_rel_error, max_restarts, max_iter, iters, restarts, linf, linf_init);
}
int cup2d_get_last_solver(cup2d_ctx *c, int *kind) {
  CUP2D_CHECK_CTX(c);
  if (!kind) return CUP2D_ERR_ARG;
  *kind = c->last_solver;
  return CUP2D_OK;
}
int cup2d_get_last_solver_form(cup2d_ctx *c, int *form, int *merge, int *handover) {
  CUP2D_CHECK_CTX(c);
  const bool fused = c->last_solver == CUP2D_SOLVER_FUSED;
  if (form) *form = fused ? c->last_form : 0;
  if (merge) *merge = fused ? c->last_merge : 0;
  if (handover) *handover = fused ? c->last_handover : 0;
  return CUP2D_OK;
}
int cup2d_get_placement(cup2d_ctx *c, int *candidates, double *kept_us, double *slowest_us, double *first_us) {
  CUP2D_CHECK_CTX(c);
  if (candidates) *candidates = c->placement_candidates;
  if (kept_us) *kept_us = c->placement_best_us;
  if (slowest_us) *slowest_us = c->placement_worst_us;
  if (first_us) *first_us = c->placement_first_us;
  return CUP2D_OK;
}
int cup2d_set_solver(cup2d_ctx *c, int kind, int finish_in_kernel) {
  CUP2D_CHECK_CTX(c);
  if (kind != CUP2D_SOLVER_SWEEPS && kind != CUP2D_SOLVER_FUSED) { set_error("set_solver: kind %d", kind); return CUP2D_ERR_ARG; }
  c->solver = kind;
  c->finish_in_kernel = finish_in_kernel != 0;
  return CUP2D_OK;
}
int cup2d_set_nrank_organisation(cup2d_ctx *c, int deferred, int split) {
  CUP2D_CHECK_CTX(c);
  if (deferred < -1 || deferred > 1 || split < -1 || split > 1) { set_error("set_nrank_organisation: -1 (default), 0 or 1"); return CUP2D_ERR_ARG; }
  c->org_defer = deferred;
  c->org_split = split;
  return CUP2D_OK;
}
int cup2d_set_solver_form(cup2d_ctx *c, int form) {
  CUP2D_CHECK_CTX(c);
  if (form < CUP2D_FORM_AUTO || form > CUP2D_FORM_EAB) { set_error("set_solver_form: form %d", form); return CUP2D_ERR_ARG; }
  c->solver_form = form;
  return CUP2D_OK;
}
static bool scalar_field(int f) { return field_ok(f) && dim_of(f) == 1; }
int cup2d_solver_keep_last(cup2d_ctx *c, int on) {
  CUP2D_CHECK_CTX(c);
  c->keep_last = on != 0;
  if (!on) c->have_last = false;
  return CUP2D_OK;
}
int cup2d_solver_last_iterate(cup2d_ctx *c, int dst, double *linf_recurrence) {
  CUP2D_CHECK_CTX(c);
  if (!scalar_field(dst)) { set_error("solver_last_iterate: field %d is not a scalar field", dst); return CUP2D_ERR_ARG; }
  if (!c->have_last) { set_error("solver_last_iterate: no iterate kept (cup2d_solver_keep_last before the solve)"); return CUP2D_ERR_ARG; }
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_field[dst], c->d_z, (size_t)c->nblocks * BC * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (linf_recurrence) *linf_recurrence = c->h_sc->err;  // max|r| of the recurrence after the last iteration (cuda.cu:525-534)
  return CUP2D_OK;
}
int cup2d_apply_A(cup2d_ctx *c, int dst, int src) {
  CUP2D_CHECK_CTX(c);
  if (!scalar_field(dst) || !scalar_field(src) || dst == src) { set_error("apply_A: fields"); return CUP2D_ERR_ARG; }
  if (c->mat.active) return launch_matvec(c, c->d_field[src], c->d_field[dst]);
  if (c->amr.active) return amr_laplacian(c, c->d_field[src], c->d_field[dst], 0);
  CUP2D_TRY(exchange_halo(c, c->d_field[src], 1, 1));
  return launch_laplacian(c, c->d_field[src], c->d_field[dst], 0, 0, c->nblocks);
}
int cup2d_precond(cup2d_ctx *c, int dst, int src) {
  CUP2D_CHECK_CTX(c);
  if (!scalar_field(dst) || !scalar_field(src)) { set_error("precond: fields"); return CUP2D_ERR_ARG; }
  return launch_precond(c, c->d_field[src], c->d_field[dst], 0, c->nblocks);
}
int cup2d_get_P_inv(cup2d_ctx *c, double *P) {
  CUP2D_CHECK_CTX(c);
  if (!P) return CUP2D_ERR_ARG;
  memcpy(P, c->h_Pinv.data(), BC * BC * sizeof(double));
  return CUP2D_OK;
}

int cup2d_set_precond(cup2d_ctx *c, int kind) {
  CUP2D_CHECK_CTX(c);
  if (kind != PRECOND_LDS && kind != PRECOND_MFMA && kind != PRECOND_FD) { set_error("set_precond: kind %d", kind); return CUP2D_ERR_ARG; }
  if (kind == PRECOND_FD && c->custom_Pinv) {
    set_error("set_precond: fast diagonalisation applies only the built-in -(A_loc)^-1");
    return CUP2D_ERR_ARG;
  }
  c->precond = kind;
  return CUP2D_OK;
}
int cup2d_set_P_inv(cup2d_ctx *c, const double *P) {
  CUP2D_CHECK_CTX(c);
  if (!P) return CUP2D_ERR_ARG;
  std::vector<double> builtin;
  build_P_inv(builtin);
  double dmax = 0, amax = 0;
  for (int i = 0; i < BC * BC; i++) {
    dmax = fmax(dmax, fabs(P[i] - builtin[i]));
    amax = fmax(amax, fabs(builtin[i]));
  }
  // main.cpp:6451-6488 always passes -(A_loc)^-1 (its Cholesky differs from ours by round-off): keep the
  // fast-diagonalisation kernels for it; anything else goes through the dense 64x64 product
  c->custom_Pinv = !(dmax <= 1e-12 * amax);
  if (c->custom_Pinv && c->precond == PRECOND_FD) c->precond = PRECOND_MFMA;
  c->h_Pinv.assign(P, P + BC * BC);
  std::vector<double> T(BC * BC);
  for (int i = 0; i < BC; i++)
    for (int j = 0; j < BC; j++) T[j * BC + i] = P[i * BC + j];
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  CUP2D_HIP_CHECK(hipMemcpy(c->d_Pinv, T.data(), BC * BC * sizeof(double), hipMemcpyHostToDevice));
  return CUP2D_OK;
}

// ---- assembled operator ---------------------------------------------------------------------------
int cup2d_clear_matrix(cup2d_ctx *c) {
  CUP2D_CHECK_CTX(c);
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  dev_free(c->mat.d_ptr); dev_free(c->mat.d_col); dev_free(c->mat.d_val); dev_free(c->mat.d_gather);
  dev_free(c->mat.d_reg); dev_free(c->mat.d_fnbr); dev_free(c->mat.d_zmask); dev_free(c->mat.d_tile0); dev_free(c->mat.d_gen);
  dev_free(c->mat.d_rrec);
  c->mat = SellMatrix();
  return CUP2D_OK;
}
// upload of an operator in (hybrid) sliced-ELL form + the tables of the tile-fused sweeps
// (ecol / eval: `entries` stored entries; pinned: they lie in the process's pinned staging buffer, whose lock the caller holds --
// the two big copies are then asynchronous and overlap the tiling below; they have been waited for when this returns)
static int install_sell(cup2d_ctx *c, int halo, bool hybrid, const std::vector<int32_t> &reg, const std::vector<long long> &ptr,
                        const int32_t *ecol, const double *eval, size_t entries, bool pinned, int nregular) {
  const auto stored = [&](int s) { return reg[(size_t)4 * s] == SELL_STORED; };
  StageClock clk("install_sell");
  CUP2D_TRY(cup2d_clear_matrix(c));
  SellMatrix &M = c->mat;
  CUP2D_HIP_CHECK(dev_malloc(&M.d_ptr, ptr.size() * sizeof(long long)));
  if (entries) {
    CUP2D_HIP_CHECK(dev_malloc(&M.d_col, entries * sizeof(int32_t)));
    CUP2D_HIP_CHECK(dev_malloc(&M.d_val, entries * sizeof(double)));
    if (pinned) {
      CUP2D_HIP_CHECK(hipMemcpyAsync(M.d_col, ecol, entries * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
      CUP2D_HIP_CHECK(hipMemcpyAsync(M.d_val, eval, entries * sizeof(double), hipMemcpyHostToDevice, c->stream));
    } else {
      CUP2D_HIP_CHECK(hipMemcpy(M.d_col, ecol, entries * sizeof(int32_t), hipMemcpyHostToDevice));
      CUP2D_HIP_CHECK(hipMemcpy(M.d_val, eval, entries * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  CUP2D_HIP_CHECK(hipMemcpy(M.d_ptr, ptr.data(), ptr.size() * sizeof(long long), hipMemcpyHostToDevice));
  CUP2D_HIP_CHECK(dev_malloc(&M.d_reg, reg.size() * sizeof(int32_t)));
  CUP2D_HIP_CHECK(hipMemcpy(M.d_reg, reg.data(), reg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  M.nregular = nregular;
  M.entries = entries;
  M.halo = halo;
  clk.lap("upload ptr / reg (col / val in flight)");
  if (hybrid) {
    // tables of the tile-fused sweeps (ctx.h SellMatrix): the tiling, which tiles are all plain, whose z the rows of the
    // other tiles read
    const int nbk = c->nblocks;
    std::vector<int> plain_run((size_t)nbk + 1, 0);  // number of consecutive plain slices from s on
    for (int s = nbk - 1; s >= 0; s--) plain_run[s] = stored(s) ? 0 : plain_run[s + 1] + 1;
    const auto good_at = [&](int s) {  // 16 plain slices from s on with <= 16 neighbour slots outside the set
      if (plain_run[s] < FUSED_TILE) return false;
      int ring = 0;
      for (int b = s; b < s + FUSED_TILE; b++)
        for (int side = 0; side < 4; side++) {
          const int32_t nb_ = reg[(size_t)4 * b + side];
          ring += nb_ >= 0 && (nb_ < s || nb_ >= s + FUSED_TILE);
        }
      return ring <= FUSED_TILE;
    };
    // (evaluated for every start in parallel: the serial walk below asks for most of them where tiles do not line up)
    std::vector<unsigned char> good_tab((size_t)nbk, 0);
    parallel_chunks(nbk, 4096, [&](long long lo, long long hi, int) {
      for (long long q = lo; q < hi; q++) good_tab[(size_t)q] = good_at((int)q) ? 1 : 0;
    });
    const auto good = [&](int s) { return good_tab[(size_t)s] != 0; };
    std::vector<int32_t> tile0;
    for (int s = 0; s < nbk;) {
      tile0.push_back(s);
      if (good(s)) { s += FUSED_TILE; continue; }
      int e = s + 1;  // a chunk: up to the next good start, at most 16
      while (e < nbk && e - s < FUSED_TILE && !good(e)) e++;
      s = e;
    }
    tile0.push_back(nbk);
    const int ntiles = (int)tile0.size() - 1;
    std::vector<int32_t> fnbr(reg), gen;
    M.h_zmask.assign((size_t)ntiles, 0);
    M.h_slot.assign((size_t)nbk, 0);
    for (int t = 0; t < ntiles; t++)
      for (int s = tile0[t]; s < tile0[t + 1]; s++) M.h_slot[s] = t * FUSED_TILE + (s - tile0[t]);
    // which z the rows of the general tiles read: the tiles in parallel (a few thousand of them hold two million column
    // indices), every thread into its own mask, the masks OR-ed at the end
    std::vector<unsigned char> is_general((size_t)ntiles, 0);
    for (int t = 0; t < ntiles; t++) {
      bool general = false;
      for (int s = tile0[t]; s < tile0[t + 1]; s++) general = general || stored(s);
      is_general[(size_t)t] = general;
      if (!general) continue;
      // (every block of a general tile, the plain ones too.  Round 6 left only the slices with stored rows to k_hyb_rows and swept
      // the plain blocks of these tiles with the tile: the rows launches 14.0 -> 12.6 us -- they are a chain of a dozen memory
      // round trips, not work --, the sweeps 42.9 -> 45.7 and 29.3 -> 31.3 us: general tiles then have ring entries; not kept)
      for (int s = tile0[t]; s < tile0[t + 1]; s++) {
        gen.push_back(s);
        for (int side = 0; side < 4; side++) fnbr[(size_t)4 * s + side] = FUSED_GENERAL;
      }
    }
    const int nmask = chunk_count(ntiles, 64);
    std::vector<std::vector<int32_t>> masks((size_t)nmask);
    parallel_chunks(ntiles, 64, [&](long long lo, long long hi, int th) {
      std::vector<int32_t> &zm = masks[(size_t)th];
      zm.assign((size_t)ntiles, 0);
      const auto flag = [&](long long b) {
        if (b >= 0 && b < nbk) zm[(size_t)(M.h_slot[b] / FUSED_TILE)] |= 1 << (M.h_slot[b] % FUSED_TILE);
      };
      for (long long t = lo; t < hi; t++) {
        if (!is_general[(size_t)t]) continue;
        for (int s = tile0[t]; s < tile0[t + 1]; s++) {
          flag(s);
          if (stored(s)) {
            for (long long e = ptr[s]; e < ptr[s + 1]; e++) flag(ecol[e] >> 6);
          } else {
            for (int side = 0; side < 4; side++) flag(reg[(size_t)4 * s + side]);
          }
        }
      }
    });
    for (const auto &zm : masks)
      if (!zm.empty())
        for (int t = 0; t < ntiles; t++) M.h_zmask[(size_t)t] |= zm[(size_t)t];
    M.ntiles = ntiles;
    clk.lap("tiling");
    if (clk.on) {
      int ngt = 0, nzt = 0, nshort = 0, ring_max = 0;
      long long ring_sum = 0;
      for (int t = 0; t < ntiles; t++) {
        ngt += is_general[(size_t)t];
        nzt += M.h_zmask[(size_t)t] != 0;
        nshort += tile0[t + 1] - tile0[t] < FUSED_TILE;
        int ring = 0;
        for (int s = tile0[t]; s < tile0[t + 1]; s++)
          for (int side = 0; side < 4; side++) {
            const int32_t nb_ = fnbr[(size_t)4 * s + side];
            ring += nb_ >= 0 && (nb_ < tile0[t] || nb_ >= tile0[t + 1]);
          }
        ring_sum += ring;
        ring_max = ring > ring_max ? ring : ring_max;
      }
      fprintf(stderr, "[cup2d timing] install_sell: %d tiles: %d general, %d with z to store, %d shorter than %d blocks; ring entries %.1f on average, %d at most\n",
              ntiles, ngt, nzt, nshort, FUSED_TILE, (double)ring_sum / ntiles, ring_max);
    }
    CUP2D_HIP_CHECK(dev_malloc(&M.d_tile0, tile0.size() * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(M.d_tile0, tile0.data(), tile0.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    CUP2D_HIP_CHECK(dev_malloc(&M.d_fnbr, fnbr.size() * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(M.d_fnbr, fnbr.data(), fnbr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    CUP2D_HIP_CHECK(dev_malloc(&M.d_zmask, M.h_zmask.size() * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(M.d_zmask, M.h_zmask.data(), M.h_zmask.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    M.ngen = (int)gen.size();
    if (M.ngen) {
      CUP2D_HIP_CHECK(dev_malloc(&M.d_gen, gen.size() * sizeof(int32_t)));
      CUP2D_HIP_CHECK(hipMemcpy(M.d_gen, gen.data(), gen.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      std::vector<RowsRec> rr(gen.size());
      for (size_t k = 0; k < gen.size(); k++) {
        const int s = gen[k];
        rr[k].s = s;
        rr[k].base = ptr[(size_t)s];
        rr[k].width = (int)((ptr[(size_t)s + 1] - ptr[(size_t)s]) >> 6);
        for (int q = 0; q < 4; q++) rr[k].reg[q] = reg[(size_t)4 * s + q];
      }
      CUP2D_HIP_CHECK(dev_malloc(&M.d_rrec, rr.size() * sizeof(RowsRec)));
      CUP2D_HIP_CHECK(hipMemcpy(M.d_rrec, rr.data(), rr.size() * sizeof(RowsRec), hipMemcpyHostToDevice));
    }
    clk.lap("upload tile tables");
  }
  if (pinned && entries) CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));  // col / val have left the staging buffer
  M.active = true;
  return CUP2D_OK;
}
int cup2d_set_matrix_coo(cup2d_ctx *c, int halo, long long nnz, const int32_t *row, const int32_t *col, const double *val) {
  CUP2D_CHECK_CTX(c);
  const long long m = (long long)c->nblocks * BC;
  if (halo < 0 || nnz < 0 || (nnz && (!row || !col || !val))) { set_error("set_matrix_coo: bad argument"); return CUP2D_ERR_ARG; }
  if ((long long)halo > (long long)c->nghost * BC) {
    set_error("set_matrix_coo: halo %d does not fit %d ghost blocks", halo, c->nghost);
    return CUP2D_ERR_ARG;
  }
  std::vector<int> cnt((size_t)m, 0);
  for (long long k = 0; k < nnz; k++) {
    if (row[k] < 0 || row[k] >= m || col[k] < 0 || col[k] >= m + halo) {
      set_error("set_matrix_coo: entry %lld (%d, %d) outside %lld x %lld", k, row[k], col[k], m, m + halo);
      return CUP2D_ERR_ARG;
    }
    cnt[row[k]]++;
  }
  // Hybrid form: a slice (= block) whose 64 rows are exactly the same-level rows of main.cpp:7034-7112 -- in-block
  // neighbours and, per side, the mirrored edge cells of ONE other block or nothing (wall), all with coefficient 1,
  // diagonal = -(number of neighbours) -- is applied matrix-free from its four neighbour ids; only the other slices
  // (coarse-fine rows, rows with halo columns, anything else the caller assembled) keep stored entries.  On an adapted
  // grid that is ~5 % of the blocks; the stored form costs 12 B per entry and bounds the product otherwise.
  constexpr bool hybrid = true;
  std::vector<int32_t> reg((size_t)4 * c->nblocks, CUP2D_WALL);
  int nregular = 0;
  if (hybrid) {
    const int32_t UNSET = -3;
    std::fill(reg.begin(), reg.end(), UNSET);
    std::vector<unsigned char> bad((size_t)c->nblocks, 0), seen((size_t)m, 0), ndiag((size_t)m, 0), face_rows((size_t)4 * c->nblocks, 0);
    std::vector<double> diag((size_t)m, 0.0);
    for (long long k = 0; k < nnz; k++) {
      const int r = row[k], cc_ = col[k], b = r >> 6, cell = r & 63, ix = cell & 7, iy = cell >> 3;
      if (cc_ >= m) { bad[b] = 1; continue; }
      if (cc_ == r) {
        diag[r] += val[k];
        if (++ndiag[r] > 1) bad[b] = 1;
        continue;
      }
      if (val[k] != 1.0) { bad[b] = 1; continue; }
      const int cb = cc_ >> 6, cc = cc_ & 63;
      int bit = -1;
      if (cb == b) {
        if (cc == cell - 1 && ix > 0) bit = 0;
        else if (cc == cell + 1 && ix < BS - 1) bit = 1;
        else if (cc == cell - BS && iy > 0) bit = 2;
        else if (cc == cell + BS && iy < BS - 1) bit = 3;
      } else {
        int side = -1;
        if (ix == 0 && cc == iy * BS + (BS - 1)) side = 0;
        else if (ix == BS - 1 && cc == iy * BS) side = 1;
        else if (iy == 0 && cc == (BS - 1) * BS + ix) side = 2;
        else if (iy == BS - 1 && cc == ix) side = 3;
        if (side >= 0) {
          int32_t &slot = reg[(size_t)4 * b + side];
          if (slot == UNSET) slot = cb;
          if (slot == cb) {
            bit = 4 + side;
            face_rows[(size_t)4 * b + side]++;
          }
        }
      }
      if (bit < 0 || ((seen[r] >> bit) & 1)) { bad[b] = 1; continue; }
      seen[r] |= (unsigned char)(1u << bit);
    }
    for (int s = 0; s < c->nblocks; s++) {
      for (int l = 0; l < BC && !bad[s]; l++) {
        const size_t r = (size_t)s * BC + l;
        const int ix = l & 7, iy = l >> 3;
        const unsigned inblock = (ix > 0 ? 1u : 0u) | (ix < BS - 1 ? 2u : 0u) | (iy > 0 ? 4u : 0u) | (iy < BS - 1 ? 8u : 0u);
        int n = 0;
        for (int bit = 0; bit < 8; bit++) n += (seen[r] >> bit) & 1;
        if ((seen[r] & 15u) != inblock || ndiag[r] != 1 || diag[r] != -(double)n) bad[s] = 1;
      }
      for (int side = 0; side < 4; side++) {
        const unsigned char fr = face_rows[(size_t)4 * s + side];
        if (fr != 0 && fr != BS) bad[s] = 1;  // a face is shared by all 8 edge cells or by none
        if (reg[(size_t)4 * s + side] == UNSET) reg[(size_t)4 * s + side] = CUP2D_WALL;
      }
      if (bad[s]) reg[(size_t)4 * s] = SELL_STORED;
      else nregular++;
    }
  } else {
    for (int s = 0; s < c->nblocks; s++) reg[(size_t)4 * s] = SELL_STORED;
  }
  const auto stored = [&](int s) { return reg[(size_t)4 * s] == SELL_STORED; };
  std::vector<long long> ptr((size_t)c->nblocks + 1, 0);
  for (int s = 0; s < c->nblocks; s++) {
    int w = 0;
    if (stored(s))
      for (int l = 0; l < BC; l++) w = cnt[(size_t)s * BC + l] > w ? cnt[(size_t)s * BC + l] : w;
    ptr[s + 1] = ptr[s] + (long long)w * BC;
  }
  const size_t entries = (size_t)ptr[c->nblocks];
  std::vector<int32_t> ecol(entries);
  std::vector<double> eval(entries, 0.0);
  for (int s = 0; s < c->nblocks; s++)  // padding: own row, coefficient 0
    for (long long e = ptr[s]; e < ptr[s + 1]; e++) ecol[e] = s * BC + (int)((e - ptr[s]) & 63);
  std::fill(cnt.begin(), cnt.end(), 0);
  for (long long k = 0; k < nnz; k++) {  // list order within a row is kept
    const int r = row[k], s = r >> 6, l = r & 63;
    if (!stored(s)) continue;
    const size_t e = (size_t)ptr[s] + (size_t)cnt[r]++ * BC + l;
    ecol[e] = col[k];
    eval[e] = val[k];
  }
  return install_sell(c, halo, hybrid, reg, ptr, ecol.data(), eval.data(), entries, false, nregular);
}
int cup2d_matrix_stats(cup2d_ctx *c, int *plain_blocks, int *general_tile_blocks, long long *stored_entries) {
  CUP2D_CHECK_CTX(c);
  if (!c->mat.active) { set_error("matrix_stats: no operator installed"); return CUP2D_ERR_ARG; }
  if (plain_blocks) *plain_blocks = c->mat.nregular;
  if (general_tile_blocks) *general_tile_blocks = c->mat.d_fnbr ? c->mat.ngen : c->nblocks;
  if (stored_entries) *stored_entries = (long long)c->mat.entries;
  return CUP2D_OK;
}
int cup2d_amr_install_poisson(cup2d_ctx *c) {
  CUP2D_CHECK_CTX(c);
  if (!c->amr.active) { set_error("amr_install_poisson: cup2d_set_amr first"); return CUP2D_ERR_ARG; }
  std::vector<int32_t> reg;
  std::vector<long long> ptr;
  int32_t *ecol = nullptr;
  double *eval = nullptr;
  int nregular = 0;
  StageClock clk("amr_install_poisson");
  HostStage &stage = host_stage();
  std::lock_guard<std::mutex> hold(stage.mu);  // the staging buffer is this call's until its upload has been waited for
  amr_assemble_hybrid(c->nblocks, c->amr.h_kind.data(), c->amr.h_nbr2.data(), c->amr.h_half.data(), reg, ptr, stage, &ecol, &eval, &nregular);
  clk.lap("assemble rows");
  const size_t entries = (size_t)ptr[(size_t)c->nblocks];
  if (entries && !ecol) { set_error("amr_install_poisson: no pinned staging buffer for %zu entries", entries); return CUP2D_ERR_HIP; }
  const int rc = install_sell(c, c->nghost * BC, true, reg, ptr, ecol, eval, entries, true, nregular);
  if (rc != CUP2D_OK) (void)hipStreamSynchronize(c->stream);  // (a copy out of the staging buffer may be in flight: not while the lock goes)
  clk.lap("install_sell");
  return rc;
}
int cup2d_set_gather(cup2d_ctx *c, int nsend, const int32_t *idx) {
  CUP2D_CHECK_CTX(c);
  if (nsend < 0 || (nsend && !idx)) return CUP2D_ERR_ARG;
  for (int i = 0; i < nsend; i++)
    if (idx[i] < 0 || idx[i] >= c->nblocks * BC) { set_error("set_gather: idx[%d] = %d", i, idx[i]); return CUP2D_ERR_ARG; }
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  dev_free(c->mat.d_gather);
  c->mat.d_gather = nullptr;
  c->mat.ngather = nsend;
  if (nsend) {
    CUP2D_HIP_CHECK(dev_malloc(&c->mat.d_gather, nsend * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(c->mat.d_gather, idx, nsend * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (c->mat.d_zmask) {  // the fused sweeps store z where it is read from memory: the gathered entries are
    for (int i = 0; i < nsend; i++) {
      const int32_t slot = c->mat.h_slot[(size_t)(idx[i] >> 6)];
      c->mat.h_zmask[(size_t)(slot / FUSED_TILE)] |= 1 << (slot % FUSED_TILE);
    }
    CUP2D_HIP_CHECK(hipMemcpy(c->mat.d_zmask, c->mat.h_zmask.data(), c->mat.h_zmask.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return CUP2D_OK;
}

// ---- whole step (main.cpp:6576-7187, body-free) -------------------------------------------------
static int step_impl(cup2d_ctx *c, double nu, double cfl, double max_error, double max_rel_error, int max_restarts,
                     int max_iter, double *dt_out, int *iters, double *linf);
int cup2d_step(cup2d_ctx *c, double nu, double cfl, double max_error, double max_rel_error, int max_restarts,
               int max_iter, double *dt_out, int *iters, double *linf) {
  CUP2D_CHECK_CTX(c);
  CUP2D_TRY(amr_operator_ready(c, "step"));
  if (c->timing == 2) {  // sampled timing: the launches outside the solver in every 4th step (ctx.h prof_outer; mode 3: in every step)
    c->prof_outer = c->prof_every_step || (c->prof_step++ % 4) == 0;
    c->prof_sample = c->prof_outer;
  }
  const int rc = step_impl(c, nu, cfl, max_error, max_rel_error, max_restarts, max_iter, dt_out, iters, linf);
  // the sampling of the launches outside the solver is a property of cup2d_step: operators called on their own (halo, AMR
  // operators, rhs) between two steps are always sampled
  c->prof_outer = c->prof_sample = true;
  return rc;
}
static int step_impl(cup2d_ctx *c, double nu, double cfl, double max_error, double max_rel_error, int max_restarts,
                     int max_iter, double *dt_out, int *iters, double *linf) {
  // the previous call on this context was a cup2d_step that left max|u| of its result behind (ctx.h)
  c->use_cached_umax = !c->amr.active && !c->vel_ptr_exposed && c->umax_partials > 0 && c->api_calls == c->umax_valid_at + 1;
  double dt = 0;
  const int rc_dt = cup2d_compute_dt(c, nu, cfl, &dt);
  c->use_cached_umax = false;
  c->umax_partials = 0;
  CUP2D_TRY(rc_dt);
  if (!(dt > 2e-16)) {  // main.cpp:6596
    if (dt_out) *dt_out = dt;
    return CUP2D_OK;
  }
  CUP2D_TRY(cup2d_advect_diffuse_rk2(c, nu, dt));
  // the same-level stencil with the tile-fused solver: pres = 0 is the solve's initial guess (main.cpp:7016-7021) and
  // that solver works on the correction alone (x = x0 + P_inv y) -- told that x0 = 0 it forms r = b without reading x and
  // writes x = P_inv y at the end: no fill of pres, no stencil pass over zeros, one read less in the last pass.  The
  // numbers are the ones the unfused sequence produces (b - A 0 = b and 0 + v = v exactly).
  const bool lazy_zero = !c->amr.active && !c->mat.active && c->solver == CUP2D_SOLVER_FUSED && fused_supported(c);
  if (lazy_zero) {
    CUP2D_TRY(poisson_rhs_uniform(c, dt, 0, false));
    c->x0_is_zero = true;
  } else {
    CUP2D_TRY(cup2d_poisson_rhs(c, dt, 0));
  }
  // ... and behind the projection the maximum of its per-workgroup max|u| (what the NEXT step's dt is made of) on its way to
  // the host: the solve's final wait covers it, and a step that follows at once starts without a launch or a wait of its own
  // (one rank; on N ranks the all-reduce of the maximum stays where it is)
  c->solve_tail = [](cup2d_ctx *cc, double dt_) -> int {
    const int rc_p = cup2d_project(cc, dt_);
    if (rc_p != CUP2D_OK) return rc_p;
    if (!cc->amr.active && !cc->allreduce && cc->umax_partials > 0) {
      CUP2D_TRY(launch_max_from_partials(cc, cc->d_partials + (size_t)3 * PSTRIDE, cc->umax_partials, cc->d_partials + (size_t)5 * PSTRIDE));
      CUP2D_HIP_CHECK(hipMemcpyAsync(cc->h_red + 6, cc->d_partials + (size_t)5 * PSTRIDE, sizeof(double), hipMemcpyDeviceToHost, cc->stream));
      cc->umax_on_host = true;
    }
    return CUP2D_OK;
  };
  c->solve_tail_arg = dt;
  c->solve_tail_ran = false;
  c->umax_on_host = false;
  const int rc_solve = cup2d_poisson_solve(c, max_error, max_rel_error, max_restarts, max_iter, iters, nullptr, linf, nullptr);
  c->x0_is_zero = false;
  c->solve_tail = nullptr;
  CUP2D_TRY(rc_solve);
  if (!c->solve_tail_ran) {  // (solvers that wait for their end before the last pass: nothing waits behind this projection)
    c->umax_on_host = false;
    CUP2D_TRY(cup2d_project(c, dt));
  }
  c->umax_valid_at = c->api_calls;  // (umax_partials > 0 only if the same-level projection kernel wrote them)
  if (dt_out) *dt_out = dt;
  return CUP2D_OK;
}

// ---- halos -------------------------------------------------------------------------------------
int cup2d_halo_plan(cup2d_ctx *c, int nsend, const int32_t *sb, const int32_t *sf, int nrecv, const int32_t *rb,
                    const int32_t *rf) {
  CUP2D_CHECK_CTX(c);
  if (nsend < 0 || nrecv < 0 || (nsend && (!sb || !sf)) || (nrecv && (!rb || !rf))) return CUP2D_ERR_ARG;
  for (int i = 0; i < nsend; i++)
    if (sb[i] < 0 || sb[i] >= c->nblocks || sf[i] < 0 || sf[i] > 3) { set_error("halo_plan: send entry %d", i); return CUP2D_ERR_ARG; }
  for (int i = 0; i < nrecv; i++)
    if (rb[i] < c->nblocks || rb[i] >= c->ntotal || rf[i] < 0 || rf[i] > 3) { set_error("halo_plan: recv entry %d", i); return CUP2D_ERR_ARG; }
  if (c->rccl) {  // its per-peer offsets, cell counts and in-place receive targets were derived from the plan it was given
    set_error("halo_plan: the in-library communicator was initialised on the previous plan: cup2d_comm_finalize first, then "
              "cup2d_halo_plan, cup2d_comm_init");
    return CUP2D_ERR_ARG;
  }
  HaloPlan &p = c->plan;
  // cup2d_halo_exchange is asynchronous: pack / unpack kernels of the OLD plan may still read these tables, and the pool
  // would hand them to the next dev_malloc (zero fill on the null stream) at once.  (The communicator's second stream hands
  // back to c->stream through an event before unpack: waiting for c->stream covers it.)
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  dev_free(p.d_send_block); dev_free(p.d_send_face); dev_free(p.d_recv_block); dev_free(p.d_recv_face);
  for (auto &cp : c->cells) { dev_free(cp.d_send); dev_free(cp.d_recv); cp = cup2d::CellPlan(); }  // (subsets of the old plan)
  p = HaloPlan();
  p.nsend = nsend; p.nrecv = nrecv;
  if (nsend) {
    CUP2D_HIP_CHECK(dev_malloc(&p.d_send_block, nsend * sizeof(int32_t)));
    CUP2D_HIP_CHECK(dev_malloc(&p.d_send_face, nsend * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(p.d_send_block, sb, nsend * sizeof(int32_t), hipMemcpyHostToDevice));
    CUP2D_HIP_CHECK(hipMemcpy(p.d_send_face, sf, nsend * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (nrecv) {
    CUP2D_HIP_CHECK(dev_malloc(&p.d_recv_block, nrecv * sizeof(int32_t)));
    CUP2D_HIP_CHECK(dev_malloc(&p.d_recv_face, nrecv * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(p.d_recv_block, rb, nrecv * sizeof(int32_t), hipMemcpyHostToDevice));
    CUP2D_HIP_CHECK(hipMemcpy(p.d_recv_face, rf, nrecv * sizeof(int32_t), hipMemcpyHostToDevice));
    p.h_recv_block.assign(rb, rb + nrecv);
  }
  return CUP2D_OK;
}
int cup2d_halo_plan_cells(cup2d_ctx *c, int set, int nsend, const int32_t *sc, int nrecv, const int32_t *rc) {
  CUP2D_CHECK_CTX(c);
  const bool remove = nsend == -1 && nrecv == -1;
  if (set < 0 || set >= cup2d::CELL_SETS || (!remove && (nsend < 0 || nrecv < 0 || (nsend && !sc) || (nrecv && !rc)))) return CUP2D_ERR_ARG;
  if (remove) nsend = nrecv = 0;
  // a cell list is a subset of the block plan's blocks: the communicator's buffers are sized by that plan
  if ((long long)nsend > (long long)c->plan.nsend * BC || (long long)nrecv > (long long)c->plan.nrecv * BC) {
    set_error("halo_plan_cells: %d / %d cells exceed the block plan (%d / %d blocks)", nsend, nrecv, c->plan.nsend, c->plan.nrecv);
    return CUP2D_ERR_ARG;
  }
  for (int i = 0; i < nsend; i++)
    if (sc[i] < 0 || sc[i] >= c->nblocks * BC) { set_error("halo_plan_cells: send cell %d = %d", i, sc[i]); return CUP2D_ERR_ARG; }
  for (int i = 0; i < nrecv; i++)
    if (rc[i] < c->nblocks * BC || rc[i] >= c->ntotal * BC) { set_error("halo_plan_cells: receive cell %d = %d", i, rc[i]); return CUP2D_ERR_ARG; }
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));  // kernels of the old plan may still read its lists (cup2d_halo_plan)
  cup2d::CellPlan &P = c->cells[set];
  dev_free(P.d_send); dev_free(P.d_recv);
  P = cup2d::CellPlan();
  if (remove) return CUP2D_OK;
  // (an EMPTY plan is a plan: this rank then sends and receives nothing for the set, while its peers exchange cells among
  // themselves -- falling back to whole blocks here would post messages nobody waits for)
  P.nsend = nsend; P.nrecv = nrecv;
  if (nsend) {
    CUP2D_HIP_CHECK(dev_malloc(&P.d_send, nsend * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(P.d_send, sc, nsend * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (nrecv) {
    CUP2D_HIP_CHECK(dev_malloc(&P.d_recv, nrecv * sizeof(int32_t)));
    CUP2D_HIP_CHECK(hipMemcpy(P.d_recv, rc, nrecv * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  P.active = true;
  return CUP2D_OK;
}
static bool width_ok(int w) { return w >= 1 && w <= 4; }
int cup2d_halo_pack(cup2d_ctx *c, int field, int width, double *buf) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!width_ok(width) || (!buf && c->plan.nsend)) return CUP2D_ERR_ARG;
  return halo_pack_impl(c, c->d_field[field], dim_of(field), width, buf);
}
int cup2d_halo_unpack(cup2d_ctx *c, int field, int width, const double *buf) {
  CUP2D_CHECK_CTX(c);
  CHECK_FIELD(field);
  if (!width_ok(width) || (!buf && c->plan.nrecv)) return CUP2D_ERR_ARG;
  return halo_unpack_impl(c, c->d_field[field], dim_of(field), width, buf);
}
int cup2d_halo_pack_vec(cup2d_ctx *c, const double *vec, int dim, int width, double *buf) {
  CUP2D_CHECK_CTX(c);
  if (!vec || (dim != 1 && dim != 2) || !width_ok(width)) return CUP2D_ERR_ARG;
  return halo_pack_impl(c, vec, dim, width, buf);
}
int cup2d_halo_unpack_vec(cup2d_ctx *c, double *vec, int dim, int width, const double *buf) {
  CUP2D_CHECK_CTX(c);
  if (!vec || (dim != 1 && dim != 2) || !width_ok(width)) return CUP2D_ERR_ARG;
  return halo_unpack_impl(c, vec, dim, width, buf);
}
int cup2d_set_comm(cup2d_ctx *c, cup2d_exchange_fn ex, cup2d_wait_fn wt, cup2d_allreduce_fn ar, void *user, double *send,
                   double *recv, double *red) {
  CUP2D_CHECK_CTX(c);
  // stencil mode unpacks from `recv`; in matrix mode the received entries land in the vector itself
  if (ex && c->nghost > 0 && (!send || (!recv && !c->mat.active))) { set_error("set_comm: buffers"); return CUP2D_ERR_ARG; }
  c->exchange = ex;
  c->wait = wt;
  c->allreduce = ar;
  c->comm_user = user;
  c->d_send = send;
  c->d_recv = recv;
  c->d_red = red ? red : c->d_red_own;
  c->strip_cap = CUP2D_MIN_STRIP_DOUBLES;  // what this call alone promises; cup2d_set_comm_strip_capacity says more
  return CUP2D_OK;
}
int cup2d_set_comm_strip_capacity(cup2d_ctx *c, int doubles) {
  CUP2D_CHECK_CTX(c);
  if (doubles < CUP2D_MIN_STRIP_DOUBLES) { set_error("set_comm_strip_capacity: %d < %d doubles per strip", doubles, CUP2D_MIN_STRIP_DOUBLES); return CUP2D_ERR_ARG; }
  c->strip_cap = doubles;
  return CUP2D_OK;
}

// ---- instrumentation -----------------------------------------------------------------------------
int cup2d_debug_walk_knockout(cup2d_ctx *c, int knockout) {
  CUP2D_CHECK_CTX(c);
  if (knockout < 0 || knockout > 2) { set_error("debug_walk_knockout: 0, 1 or 2"); return CUP2D_ERR_ARG; }
  c->walk_knockout = knockout;
  return CUP2D_OK;
}
int cup2d_set_timing(cup2d_ctx *c, int enabled) {
  CUP2D_CHECK_CTX(c);
  if (enabled && c->prof_ev.empty()) {
    const int pairs = 4096;
    c->prof_ev.resize(2 * pairs);
    c->prof_id.resize(pairs);
    for (auto &e : c->prof_ev) CUP2D_HIP_CHECK(hipEventCreate(&e));
  }
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->prof_used = 0;
  c->timing = (enabled == 2 || enabled == 3) ? 2 : (enabled ? 1 : 0);
  c->prof_every_step = enabled == 3;  // 3: sampled inside the solver, the launches outside it in EVERY step
  c->prof_sample = true;
  c->prof_outer = true;
  c->prof_step = 0;
  for (int i = 0; i < CUP2D_T_NTIMERS; i++) { c->t_ms[i] = 0; c->t_calls[i] = 0; }
  return CUP2D_OK;
}
int cup2d_get_timing(cup2d_ctx *c, int timer, double *ms, int *calls) {
  CUP2D_CHECK_CTX(c);
  if (timer < 0 || timer >= CUP2D_T_NTIMERS) return CUP2D_ERR_ARG;
  CUP2D_TRY(prof_resolve(c));
  if (ms) *ms = c->t_ms[timer];
  if (calls) *calls = c->t_calls[timer];
  return CUP2D_OK;
}

}  // extern "C"
