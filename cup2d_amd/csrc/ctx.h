// ctx.h -- context object and launch helpers shared by the translation units of libcup2d_hip.so
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <system_error>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/cup2d_hip.h"
#include "krylov_scalars.h"

namespace cup2d {

constexpr int BS = CUP2D_BS;
constexpr int BC = BS * BS;        // cells per block = one wavefront
constexpr int WG = 256;            // threads per workgroup = 4 waves = 4 blocks in flight
constexpr int WPG = WG / 64;       // waves (= blocks) per workgroup pass
constexpr int MAX_GRID = 2048;     // persistent grid: 256 CUs x 8 workgroups, multiple of the 8 XCDs
constexpr int NSLOT = 6;           // reduction slots per launch (k_edge MODE 3: five sums)
constexpr int PSTRIDE = 2 * MAX_GRID;  // partials per slot: an inner-block and a halo-block launch

void set_error(const char *fmt, ...);
#define CUP2D_HIP_CHECK(expr)                                                                   \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      cup2d::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
      return CUP2D_ERR_HIP;                                                                     \
    }                                                                                           \
  } while (0)
#define CUP2D_CHECK_CTX(c)                                   \
  do {                                                       \
    if (!(c)) {                                              \
      cup2d::set_error("%s: null context", __func__);        \
      return CUP2D_ERR_ARG;                                  \
    }                                                        \
    CUP2D_HIP_CHECK(hipSetDevice((c)->device));              \
    (c)->api_calls++;                                        \
  } while (0)
#define CUP2D_TRY(expr)        \
  do {                         \
    int _s = (expr);           \
    if (_s != CUP2D_OK)        \
      return _s;               \
  } while (0)

// (struct KrylovScalars and the scalar recurrences: krylov_scalars.h, host+device)

struct HaloPlan {
  int nsend = 0, nrecv = 0;
  int32_t *d_send_block = nullptr, *d_send_face = nullptr;
  int32_t *d_recv_block = nullptr, *d_recv_face = nullptr;
  std::vector<int32_t> h_recv_block;  // host copy: the communicator checks whether a peer's ghost blocks are consecutive
};

// cup2d_halo_plan_cells: the ghost cells one family of operators reads, cell by cell (adapted grids on N ranks).  A cell is
// 64 * block + cell of the context's numbering: owned blocks on the send side, ghost blocks on the receive side; entry i of a
// peer's send list lands in entry i of what the receiver lists for that peer.
constexpr int CELL_SETS = 3;  // CUP2D_CELLS_HALO1, _HALO3, _MATRIX
struct CellPlan {
  bool active = false;
  int nsend = 0, nrecv = 0;
  int32_t *d_send = nullptr, *d_recv = nullptr;
};

// General sparse Poisson operator (the matrix of main.cpp:7034-7112 as LocalSpMatDnVec hands it over,
// cuda.cu:206-296) in sliced-ELL form with slices of 64 rows: one slice = the 64 rows of one 8x8 block
// = one wavefront.  Entry k of row (slice s, lane l) sits at ptr[s] + 64*k + l, so a wave reads
// columns and values with unit stride; every slice is as wide as its longest row (5 on a same-level
// block, more where coarse-fine interpolation rows exist), padded with (col = own row, val = 0).
// Columns >= 64*nblocks address halo entries appended to the Krylov vector (cuda.cu:344-402).
constexpr int32_t SELL_STORED = -2;
struct alignas(32) RowsRec {  // k_hyb_rows: slice, its width and first entry (d_ptr), its neighbour record (d_reg)
  int32_t s, width;
  long long base;
  int32_t reg[4];
};
struct SellMatrix {
  bool active = false;
  int halo = 0;
  size_t entries = 0;
  long long *d_ptr = nullptr;  // [nblocks + 1]
  int32_t *d_col = nullptr;
  double *d_val = nullptr;
  // hybrid form: reg[4 s .. 4 s + 3] = the W, E, S, N neighbour blocks (CUP2D_WALL at a wall) of a slice whose 64 rows
  // are exactly the same-level 5-point rows (no stored entries: its width is 0); reg[4 s] = SELL_STORED otherwise
  int32_t *d_reg = nullptr;
  int nregular = 0;
  int ngather = 0;             // send_buff_pack (cuda.cu:338-343): d_send[i] = vec[gather[i]]
  int32_t *d_gather = nullptr;
  // The tile-fused sweeps on the hybrid form (krylov_fused.hip).  The slices are cut into TILES of <= 16 consecutive
  // slices (tile t = slices [tile0[t], tile0[t+1])): 16 plain slices whose neighbours outside the set number <= 16 (an
  // aligned 4x4 patch of a Hilbert-ordered level) are a tile wherever they start, what lies between such runs is cut
  // into chunks.  A tile of plain slices is FUSED -- its z never leaves the chip --, one that holds a stored slice is
  // GENERAL: the sweep forms v and z = P_inv v of its blocks, stores both, and a second launch (k_hyb_rows) applies the
  // rows of the general tiles from z in memory.
  int ntiles = 0;
  int32_t *d_tile0 = nullptr;  // [ntiles + 1]
  int32_t *d_fnbr = nullptr;   // [4 nblocks] k_fused's neighbour table: reg of a plain block, FUSED_GENERAL in a general tile
  int32_t *d_zmask = nullptr;  // [ntiles] bit b: z of block b of the tile is read from memory by someone (stored to d_z)
  int32_t *d_gen = nullptr;    // [ngen] the blocks of the general tiles
  struct RowsRec *d_rrec = nullptr;  // [ngen] ... with what k_hyb_rows needs of each in ONE record (one round trip, not three)
  int ngen = 0;
  std::vector<int32_t> h_zmask, h_slot;  // slot[b] = 16 tile + position; cup2d_set_gather adds the blocks other ranks read
};
constexpr int32_t FUSED_GENERAL = -3;
constexpr int FUSED_TILE = 16;  // blocks per tile of the fused sweeps = N of v_mfma_f64_16x16x4_f64

enum { PRECOND_LDS = 0, PRECOND_MFMA = 1, PRECOND_FD = 2 };

// advect.hip: the quads (advect_walk.h) of one block range of a context, and the blocks that found no partners
struct WalkPlan {
  int first = 0, count = 0, nquads = 0, nsingles = 0;
  int32_t *d_quads = nullptr, *d_singles = nullptr;
};

struct RcclComm;  // comm.hip: the in-library RCCL communicator (cup2d_comm_init)
struct Bodies;    // penalize.hip: host-supplied bodies (cup2d_body_set)

// Block-AMR topology (amr.hip): per block its level and, per side W,E,S,N, what lies across it
struct AmrTopo {
  bool active = false;
  double h0 = 0;                 // cell size of level 0 (main.cpp:6338)
  double h_min = 0;              // cell size of the finest level present (dt, main.cpp:6580-6583)
  int32_t *d_level = nullptr;    // [nblocks]
  int32_t *d_kind = nullptr;     // [nblocks][4]  0 wall, 1 same level, 2 coarser, 3 finer
  int32_t *d_nbr2 = nullptr;     // [nblocks][4][2] neighbour block(s): one, or the two finer ones along the face
  int32_t *d_half = nullptr;     // [nblocks][4]  coarser neighbour: which half of its face this block touches
  double *d_faces = nullptr;     // [nblocks][4][8] fluxes recorded by the functors (BlockCase::d, main.cpp:513-517)
  double *d_faces2 = nullptr;    // [nblocks][4][8][2] the same for vector functors (KernelAdvectDiffuse)
  std::vector<int32_t> h_kind, h_nbr2, h_half;  // host copies: cup2d_amr_install_poisson assembles from them
  std::vector<int32_t> h_level;
  // One rank, FAST arithmetic: the same-level 2 x 2 groups of blocks whose eight outer sides are walls or same-level blocks take
  // the quad form of KernelAdvectDiffuse (advect_walk.h; one list per level: h differs), everything else the per-block kernel
  // with the interpolated tile (amr.hip amr_quads; built on first use)
  struct Quads {
    bool built = false;
    std::vector<int> level, nq;
    std::vector<int32_t *> d_quads;
    int32_t *d_left = nullptr;
    int nleft = 0;
  } quads;
  // N ranks: the owned blocks by whether their operators read a ghost block (computeA's inner / halo split on an adapted
  // grid, main.cpp:3035-3057), per operator family: 0 the halo-1 operators, 1 the halo-3 tile (amr.hip amr_phase_lists; built
  // on first use from the kernels' own ghost expressions)
  struct Phase {
    bool built = false;
    int n_inner = 0, n_halo = 0;
    int32_t *d_inner = nullptr, *d_halo = nullptr;
  } phase[2];
};

}  // namespace cup2d

struct cup2d_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  int nblocks = 0, nghost = 0, ntotal = 0, n_inner = 0;
  double h = 0;
  int math = CUP2D_MATH_FAST;
  int grid = 0;  // persistent grid size (upper bound)
  int num_cus = 0;
  int walk_knockout = 0;  // cup2d_debug_walk_knockout: timing-only instantiations of the quad WENO5 stage kernel (advect.hip KO)
  int spare_cus = 0;  // N ranks, overlap organisation: CUs the persistent Krylov sweeps leave to the communication stream's kernel
  std::vector<std::pair<const void *, int>> resident;  // kernel -> workgroups per CU (occupancy query, cached)
  int32_t *d_nbr = nullptr;
  std::vector<int32_t> h_nbr;                 // host copy (advect.hip builds its quad plans from it)
  std::vector<cup2d::WalkPlan> walk_plans;    // one per block range asked for
  double *d_field[CUP2D_NFIELDS] = {nullptr};
  double *d_vscratch = nullptr;  // RK2 mid-point velocity (vector slab)
  double *d_ko_scratch = nullptr;  // cup2d_debug_walk_knockout: where a knocked-out stage launch writes (allocated on first use)
  // Krylov vectors (scalar slabs; z/z2 carry ghost blocks)
  double *d_r = nullptr, *d_rhat = nullptr, *d_p = nullptr, *d_nu = nullptr, *d_t = nullptr;
  double *d_z = nullptr, *d_z2 = nullptr, *d_xopt = nullptr;
  double *d_Pinv = nullptr;
  double *d_fd = nullptr;  // eigenvectors Q[64] + eigenvalues lam[8] of the 8x8 second-difference matrix
  std::vector<double> h_Pinv;
  // tile-fused solver (krylov_fused.hip): ping-pong copies of p and nu, s = r - alpha nu, and the
  // preconditioned-space accumulator y (x = x0 + P_inv y) with its best-iterate copy; allocated on first use
  double *d_p2 = nullptr, *d_nu2 = nullptr, *d_s = nullptr, *d_y = nullptr, *d_yopt = nullptr;
  int *d_fault = nullptr;    // k_edge's fault word (krylov_edge.h)
  int solver_form = 0;       // cup2d_fused_form (0: the process default, CUP2D_FUSED_FORM)
  int edge_share = -1;       // k_edge: sibling waves share z edges (-1: not yet decided from the neighbour table)
  int solver = 1;            // cup2d_solver_kind: 0 five sweeps (krylov.hip), 1 tile-fused (krylov_fused.hip)
  int last_solver = 0;       // what the last solve ran
  int last_form = 0, last_merge = 0, last_handover = 0;  // cup2d_get_last_solver_form
  // placement search of the solver's vectors (krylov_fused.hip tune_placement): done once per context
  bool placement_tuned = false;
  int placement_candidates = 0;
  double placement_best_us = 0, placement_worst_us = 0, placement_first_us = 0;
  int finish_in_kernel = 1;  // the last workgroup of a reducing sweep finishes the reduction (krylov_common.h)
  unsigned *d_ticket = nullptr;  // arrival counter of arrive_last, zero between launches
  double *d_partials = nullptr;  // [NSLOT][grid]
  double *d_red = nullptr;       // [8] local sums handed to the allreduce callback
  cup2d::KrylovScalars *d_sc = nullptr;
  int org_defer = -1, org_split = -1;  // cup2d_set_nrank_organisation: -1 = the process default (environment), 0 / 1
  void *vec_arena = nullptr;  // the eleven vectors of the two-launch solver are pieces of this one allocation (tune_placement), or null
  cup2d::KrylovScalars *d_sc2 = nullptr;  // N ranks, deferred scalar updates (krylov_fused.hip): the state alternates between d_sc and this
  cup2d::KrylovScalars *h_sc = nullptr;  // pinned
  static constexpr int SOLVE_AHEAD = 16;  // upper bound of the iterations the host may run ahead of the GPU (default 4)
  bool keep_last = false, have_last = false;  // cup2d_solver_keep_last: the solve leaves its LAST iterate in d_z
  bool x0_is_zero = false;               // set by cup2d_step around its solve: the initial guess is zero, PRES is not read
  // set by cup2d_step around its solve: what follows the solve in the step (the projection).  The fused solver enqueues it
  // right behind its last pass, BEFORE it waits for the solve to end and reads the iteration count -- the stream has the
  // projection's launches by the time the solve's last kernel retires (solve_tail_ran tells the step that it happened)
  int (*solve_tail)(cup2d_ctx *, double) = nullptr;
  double solve_tail_arg = 0.0;
  bool solve_tail_ran = false;
  // max|u| of the velocity a cup2d_step leaves behind: its projection kernel writes per-workgroup maxima (slot 3 of
  // d_partials) and the NEXT cup2d_step takes its dt from them instead of reading the field again -- valid only if that
  // step is the very next call on the context (every entry point counts itself in api_calls) and nobody holds a raw
  // pointer to VEL (cup2d_field_ptr).  A maximum does not depend on the order it is taken in: the same number.
  unsigned long long api_calls = 0, umax_valid_at = ~0ull;
  int umax_partials = 0;
  bool vel_ptr_exposed = false, use_cached_umax = false;
  bool umax_on_host = false;  // cup2d_step reduced those maxima behind its projection and h_red[6] holds the result (valid with use_cached_umax)
  int *h_status = nullptr;               // pinned [SOLVE_AHEAD], written by the last scalar kernel of an iteration
  hipEvent_t solve_ev[SOLVE_AHEAD] = {nullptr};
  double *h_red = nullptr;               // pinned [8]
  cup2d::HaloPlan plan;
  cup2d::CellPlan cells[cup2d::CELL_SETS];
  cup2d::SellMatrix mat;
  cup2d::AmrTopo amr;
  int precond = cup2d::PRECOND_FD;  // block-Jacobi implementation (krylov.hip); FD needs the built-in P_inv
  bool custom_Pinv = false;         // cup2d_set_P_inv installed something else than -(A_loc)^-1
  // communication callbacks
  cup2d_exchange_fn exchange = nullptr;
  cup2d_wait_fn wait = nullptr;
  cup2d_allreduce_fn allreduce = nullptr;
  double *d_red_own = nullptr;
  void *comm_user = nullptr;
  double *d_send = nullptr, *d_recv = nullptr;
  int strip_cap = CUP2D_MIN_STRIP_DOUBLES;  // doubles per strip the send / receive buffers hold (cup2d_set_comm_strip_capacity)
  cup2d::RcclComm *rccl = nullptr;  // set by cup2d_comm_init; the callbacks above then point into comm.hip
  cup2d::Bodies *bodies = nullptr;  // cup2d_body_set
  bool fused_lds_opt_in = false;    // k_fused's > 64 KiB of dynamic LDS opted in on THIS context's device
  // timing: pool of event pairs, resolved lazily
  int timing = 0;          // 0 off, 1 every launch, 2 sampled (every launch outside the solver, every 8th iteration inside)
  bool prof_sample = true; // sampled mode: record the launches issued now
  // sampled mode (cup2d_set_timing 2): an event pair is a barrier packet on the stream -- the kernels on either side of it do
  // not overlap their tail and head (~12 us per pair, 0.3 ms of a 4096^2 step with every launch outside the solver and every
  // 8th iteration sampled: round 5 measured 18.17 against 17.90 ms).  So: every 32nd BiCGSTAB iteration (16th until round 6), and the launches
  // outside the solver in every 4th cup2d_step (prof_outer says whether this step is one of them)
  int prof_step = 0;
  bool prof_outer = true;
  bool prof_every_step = false;  // cup2d_set_timing 3
  std::vector<hipEvent_t> prof_ev;
  std::vector<int> prof_id;
  int prof_used = 0;
  double t_ms[CUP2D_T_NTIMERS] = {0};
  int t_calls[CUP2D_T_NTIMERS] = {0};
};

namespace cup2d {

static inline int dim_of(int field) {
  return (field == CUP2D_VEL || field == CUP2D_VOLD || field == CUP2D_TMPV) ? 2 : 1;
}
static inline bool field_ok(int f) { return f >= 0 && f < CUP2D_NFIELDS; }

// block range of a phase (computeA's inner/halo split, main.cpp:3035-3057)
static inline int phase_range(const cup2d_ctx *c, int phase, int *first, int *count) {
  switch (phase) {
  case CUP2D_BLOCKS_ALL: *first = 0; *count = c->nblocks; return CUP2D_OK;
  case CUP2D_BLOCKS_INNER: *first = 0; *count = c->n_inner; return CUP2D_OK;
  case CUP2D_BLOCKS_HALO: *first = c->n_inner; *count = c->nblocks - c->n_inner; return CUP2D_OK;
  }
  set_error("bad phase %d", phase);
  return CUP2D_ERR_ARG;
}
// persistent grid for `count` blocks: every workgroup loops over groups of WPG blocks
static inline int grid_for(const cup2d_ctx *c, int count) {
  int groups = (count + WPG - 1) / WPG;
  int g = groups < c->grid ? groups : c->grid;
  if (g >= 8) g -= g % 8;  // equal share per XCD
  return g < 1 ? 1 : g;
}

// Persistent grid of a kernel whose workgroups loop over blocks: exactly as many workgroups as are
// resident at once (occupancy query x CUs, multiple of the 8 XCDs), so that every workgroup gets the
// same share and there is no second, partially filled round of workgroups -- on the FP64-bound WENO5
// kernel (4 workgroups per CU by registers) a 2048-workgroup grid ran as two rounds with a ragged
// tail (VALU busy 71 % of the kernel time).  Only speed depends on this number.
int resident_grid(cup2d_ctx *c, const void *kernel, int count);

// grid of a chunked launch (group_range_chunked): 8 x the slots the largest XCD share needs
static inline int chunked_grid(int count, int chunk) {
  const int groups = (count + WPG - 1) / WPG;
  const int per_xcd = (groups + 7) / 8;  // >= every (hi - lo)
  return 8 * ((per_xcd + chunk - 1) / chunk);
}

int prof_resolve(cup2d_ctx *c);
// records an event pair around the launches issued during its lifetime
struct ProfScope {
  cup2d_ctx *c;
  int slot;
  ProfScope(cup2d_ctx *c_, int id) : c(c_), slot(-1) {  // id < 0: nothing is timed
    if (id < 0 || !c->timing || (c->timing == 2 && !c->prof_sample)) return;
    if ((size_t)(2 * c->prof_used + 2) > c->prof_ev.size()) (void)prof_resolve(c);
    slot = c->prof_used++;
    c->prof_id[slot] = id;
    (void)hipEventRecord(c->prof_ev[2 * slot], c->stream);
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(c->prof_ev[2 * slot + 1], c->stream);
  }
};

// implemented in the kernel translation units
int launch_advect(cup2d_ctx *c, const double *vel, const double *vold, double *out, int mode, double nu,
                  double dt, double coef, int first, int count);
// MODE 0 of the quad kernel (out = rhs) on a caller's quad list with the caller's coefficients (adapted grids: amr.hip)
int launch_advect_walk_rhs(cup2d_ctx *c, const double *vel, double *out, const int32_t *d_quads, int nq, double afac, double dfac);
int launch_vorticity(cup2d_ctx *c, const double *vel, double *out, int first, int count);
int launch_pressure_rhs(cup2d_ctx *c, const double *vel, const double *udef, const double *chi,
                        const double *pold, double *out, double dt, int first, int count, double *pold_copy = nullptr);
int launch_laplacian(cup2d_ctx *c, const double *x, double *y, int subtract, int first, int count);
int launch_max_from_partials(cup2d_ctx *c, const double *partials, int n, double *d_out);
int launch_pressure_correction(cup2d_ctx *c, const double *pres, double *tmpV, double *vel, double dt,
                               int fused_update, int first, int count, double *umax_partials = nullptr);
int launch_axpy_field(cup2d_ctx *c, double *y, const double *x, double a, size_t n);
int launch_zero(cup2d_ctx *c, double *v, size_t n);
int launch_max_abs(cup2d_ctx *c, const double *v, size_t n, double *d_out);
int launch_block_linf(cup2d_ctx *c, const double *f, double *d_out);
int launch_precond(cup2d_ctx *c, const double *in, double *out, int first, int count);
int launch_precond_add(cup2d_ctx *c, const double *y, double *x, double *tmp);
bool launch_final_x_on_device(cup2d_ctx *c, const double *y0, const double *y1, const double *y2, double *x, bool x0_zero);
int launch_matvec(cup2d_ctx *c, double *x, double *y);  // y = A x through the installed SellMatrix
// Device memory of a context comes from a per-process pool (api.hip): cup2d_destroy / cup2d_clear_matrix / a new
// cup2d_set_amr hand their buffers back, the next context -- after every regrid a host builds one -- takes them again
// instead of paying hipFree + hipMalloc (28 ms of a 89 ms regrid on a 63 k-block grid).  Sizes are rounded up to
// m * 2^k, m in 8..15 (<= 12.5 % slack) so that a grid that grew or shrank a little still finds its buffers; a buffer
// is ALWAYS handed out zero-filled.  CUP2D_POOL=0 turns the pool off; cup2d_trim_pool() returns the cache to the driver.
// regrid-time host loops over independent blocks run on a few threads (CUP2D_HOST_THREADS, default min(16, cores)):
// n items in chunk_count(n, grain) contiguous chunks, fn(lo, hi, chunk)
inline int host_threads() {
  static const int want = [] {
    const char *e = getenv("CUP2D_HOST_THREADS");
    const unsigned hw = std::thread::hardware_concurrency();
    const int v = e ? atoi(e) : (int)(hw ? (hw < 16u ? hw : 16u) : 1u);
    return v < 1 ? 1 : v;
  }();
  return want;
}
inline int chunk_count(long long n, long long grain) {
  long long nt = (n + grain - 1) / grain;
  if (nt > host_threads()) nt = host_threads();
  return (int)(nt < 1 ? 1 : nt);
}
// The threads are a POOL of the process (round 6): a regrid runs a dozen of these regions, and sixteen std::thread
// constructions per region cost it 2-3 ms on a 256-thread host.  Workers are created on first use, detached, and wait on a
// condition variable; a region is an object of its own (workers that come late for one see its task counter exhausted), one
// region runs at a time -- a second caller, or a region started from inside a task, runs its chunks serially.
struct HostRegion {
  std::function<void(int)> job;
  int ntasks = 0;
  std::atomic<int> next{0}, done{0};
};
struct HostThreads {
  std::mutex region_mu;  // one parallel region at a time
  std::mutex mu;
  std::condition_variable cv;
  std::shared_ptr<HostRegion> cur;
  unsigned long long gen = 0;
  int nworkers = 0;
};
inline HostThreads &host_thread_pool() {
  static HostThreads *P = new HostThreads;  // (never destroyed: detached workers may be waiting on it when the process ends)
  return *P;
}
inline void host_thread_worker(HostThreads *P) {
  unsigned long long seen = 0;
  for (;;) {
    std::shared_ptr<HostRegion> R;
    {
      std::unique_lock<std::mutex> lk(P->mu);
      P->cv.wait(lk, [&] { return P->gen != seen; });
      seen = P->gen;
      R = P->cur;
    }
    if (!R) continue;
    for (;;) {
      const int t = R->next.fetch_add(1);
      if (t >= R->ntasks) break;
      R->job(t);
      R->done.fetch_add(1);
    }
  }
}
template <class F>
inline void parallel_chunks(long long n, long long grain, F fn) {
  const long long nt = chunk_count(n, grain);
  if (nt <= 1) {
    fn(0LL, n, 0);
    return;
  }
  HostThreads &P = host_thread_pool();
  std::unique_lock<std::mutex> region(P.region_mu, std::try_to_lock);
  if (!region.owns_lock()) {  // another region is running (or this is a task of one): the chunks here, one after the other
    for (long long t = 0; t < nt; t++) fn(n * t / nt, n * (t + 1) / nt, (int)t);
    return;
  }
  auto R = std::make_shared<HostRegion>();
  R->ntasks = (int)nt;
  R->job = [&fn, n, nt](int t) { fn(n * t / nt, n * (t + 1) / nt, t); };
  {
    std::lock_guard<std::mutex> lk(P.mu);
    // a thread that cannot be created (std::system_error: the process is out of threads) must not take the process down from
    // inside a C ABI call: the chunks that find no worker run on the caller
    try {
      while (P.nworkers < (int)nt - 1) {
        std::thread(host_thread_worker, &P).detach();
        P.nworkers++;
      }
    } catch (const std::system_error &) {
    }
    P.cur = R;
    P.gen++;
  }
  P.cv.notify_all();
  for (;;) {  // the caller takes chunks too
    const int t = R->next.fetch_add(1);
    if (t >= R->ntasks) break;
    R->job(t);
    R->done.fetch_add(1);
  }
  while (R->done.load() < R->ntasks) std::this_thread::yield();
  {
    std::lock_guard<std::mutex> lk(P.mu);
    P.cur.reset();  // (fn dies with this call: nobody may find the region any more; late workers hold their own reference)
  }
}

// regrid-time host work, stage by stage, on stderr when CUP2D_HOST_TIMING is set (development aid)
struct StageClock {
  const char *what;
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit StageClock(const char *w) : what(w), on(getenv("CUP2D_HOST_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void lap(const char *stage) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[cup2d timing] %s: %s %.3f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
hipError_t dev_malloc_raw(void **p, size_t bytes);
template <class T>
inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc_raw(reinterpret_cast<void **>(p), bytes); }
void dev_free(void *p);
void dev_release(void *p);  // hipFree past the pool (api.hip)
// amr_host.hip: the Poisson operator of an adapted grid straight in the hybrid sliced-ELL form
// The stored entries land in ONE pinned staging buffer of the process (host_stage: grow-only, handed out under a lock that the
// caller holds until its upload has been waited for): [entries] int32 columns, then [entries] double values -- the upload is
// then one asynchronous copy each at the link's rate instead of the runtime's staging of pageable memory.
struct HostStage {
  std::mutex mu;
  void *p = nullptr;
  size_t cap = 0;
};
HostStage &host_stage();
void *host_stage_reserve(HostStage &S, size_t bytes);  // (call with S.mu held) nullptr: out of pinned memory
void amr_assemble_hybrid(int nowned, const int32_t *kind, const int32_t *nbr2, const int32_t *half, std::vector<int32_t> &reg,
                         std::vector<long long> &ptr, HostStage &stage, int32_t **ecol, double **eval, int *nregular);
int matrix_exchange(cup2d_ctx *c, double *vec);         // fill vec[m .. m+halo) from the neighbour ranks
int project_impl(cup2d_ctx *c, double dt);
int solve_impl(cup2d_ctx *c, double max_error, double max_rel_error, int max_restarts, int max_iter,
               int *iters, int *restarts, double *linf, double *linf_init);
// the tile-fused variant (krylov_fused.hip); fused_supported: same-level stencil, no ghost blocks
bool fused_supported(const cup2d_ctx *c);
int solve_fused_impl(cup2d_ctx *c, double max_error, double max_rel_error, int max_restarts, int max_iter,
                     int *iters, int *restarts, double *linf, double *linf_init);
// reduction finish + scalar update of `stage` as its own launch(es) (+ all-reduce callback with N GPUs)
int finish(cup2d_ctx *c, int G, int nsum, int with_max, int stage, bool guarded, int *host_status = nullptr);
int finish_local(cup2d_ctx *c, int nsum, int with_max, int stage, int *host_status = nullptr);
// halo-1 block operators on a block-AMR grid (amr.hip)
// blocks: CUP2D_BLOCKS_ALL = refresh the ghost copies (inner blocks swept while they travel), the functor on every block, the
// flux correction; _INNER = the functor on the blocks that read no ghost block, no exchange; _HALO = the functor on the
// others + the flux correction of all blocks (the caller has refreshed the ghost copies in between: cup2d_halo_exchange)
int amr_laplacian(cup2d_ctx *c, const double *x, double *y, int subtract, int blocks = CUP2D_BLOCKS_ALL);
int amr_pressure_correction(cup2d_ctx *c, const double *pres, double *tmpV, double dt, int blocks = CUP2D_BLOCKS_ALL);
int amr_vorticity(cup2d_ctx *c, const double *vel, double *out, int blocks = CUP2D_BLOCKS_ALL);
int amr_advect_diffuse_rhs(cup2d_ctx *c, const double *vel, double *tmpV, double nu, double dt, int blocks = CUP2D_BLOCKS_ALL);
void amr_phase_release(cup2d_ctx *c);
// amr_host.hip: owned blocks [0, nowned) by whether the operators of family `set` (0 halo 1, 1 halo 3) read a block >= nowned
void amr_blocks_reading_ghosts(int nowned, int ntotal, const int32_t *kind, const int32_t *nbr2, const int32_t *half, int set,
                               std::vector<int32_t> &inner, std::vector<int32_t> &halo);
int amr_advect_diffuse_rk2(cup2d_ctx *c, double nu, double dt);
int amr_advect_diffuse_stage(cup2d_ctx *c, double nu, double dt, int stage);
int amr_poisson_rhs(cup2d_ctx *c, double dt);
int amr_project(cup2d_ctx *c, double dt);
int amr_pressure_rhs(cup2d_ctx *c, const double *vel, const double *udef, const double *chi, double *out, double dt,
                     int blocks = CUP2D_BLOCKS_ALL);
int halo_pack_impl(cup2d_ctx *c, const double *src, int dim, int width, double *buf);
int halo_unpack_impl(cup2d_ctx *c, double *dst, int dim, int width, const double *buf);
// ghost-strip exchange of a device vector through the comm callbacks (no-ops without ghosts):
// begin = pack + start transfer; end = wait + unpack.  exchange_halo = begin + end.
int exchange_begin(cup2d_ctx *c, const double *vec, int dim, int width);
int exchange_end(cup2d_ctx *c, double *vec, int dim, int width);
int exchange_halo(cup2d_ctx *c, double *vec, int dim, int width);
// the same through a cell plan (cup2d_halo_plan_cells), blocking like exchange_halo: gather the listed cells, transport with
// strip_doubles = CUP2D_CELL_STRIP(set, dim), scatter into the ghost blocks.  dst == nullptr: into vec's own ghost blocks
int exchange_cells(cup2d_ctx *c, int set, double *vec, int dim);
int exchange_cells_begin(cup2d_ctx *c, int set, double *vec, int dim);  // pack + the transfer under way
int exchange_cells_end(cup2d_ctx *c, int set, double *vec, int dim);    // arrival + unpack
// whole blocks of two scalar vectors in one message (128 doubles per strip)
int exchange_begin_blocks2(cup2d_ctx *c, const double *v0, const double *v1);
int exchange_end_blocks2(cup2d_ctx *c, double *v0, double *v1);
// whole ghost blocks of up to three Krylov vectors straight into the vectors' ghost regions on the compute stream (in-library
// communicator, comm.hip); false: the caller takes the generic begin / end pair
bool comm_blocks_direct(const cup2d_ctx *c);
// on_comm_stream: the transfer runs on the communication stream behind the pack (the caller sweeps on meanwhile) and
// comm_blocks_wait makes the compute stream wait for its arrival
// r' and p'' of the ghost blocks formed by the rank that holds them (krylov_fused.hip k_ghost_rp); passed to
// comm_exchange_blocks, the pack launch does it on its way
struct GhostRP {
  const double *p = nullptr, *nu = nullptr, *r = nullptr, *t = nullptr;
  double *rout = nullptr, *pout = nullptr;
  const KrylovScalars *sc = nullptr;
  size_t first = 0, count = 0;
};
// records: this rank's reduction record (d_red, RED_REC doubles) travels in the SAME ncclGroup -- to every other rank, into
// slot `rank` of their gathered records (comm_gathered) -- and the pack launch copies it into this rank's own slot: one RCCL
// kernel per reduction point, and no all-gather (the consumer sweep sums the records itself, krylov_edge.h MERGE 3)
int comm_exchange_blocks(cup2d_ctx *c, int nv, double *v0, double *v1, double *v2, bool on_comm_stream = false,
                         const GhostRP *ghosts = nullptr, bool records = false);
const double *comm_gathered(const cup2d_ctx *c);  // [nranks][RED_REC]
int comm_nranks(const cup2d_ctx *c);
bool comm_defer_ok(const cup2d_ctx *c);           // agreed over all ranks at cup2d_comm_init
bool comm_split_ok(const cup2d_ctx *c);           // every rank can split its sweeps (its inner / halo cut is a tile boundary): agreed likewise
// the standalone form of what a MERGE 3 sweep does with the gathered records (the last pending stage of a solve)
int comm_apply_gathered(cup2d_ctx *c, int nsum, int with_max, int stage);
int comm_gather_records(cup2d_ctx *c);  // d_red of every rank -> the gathered records (one all-gather, compute stream)
int comm_blocks_wait(cup2d_ctx *c);
int exchange_begin_blocks3(cup2d_ctx *c, const double *v0, const double *v1, const double *v2);
int exchange_end_blocks3(cup2d_ctx *c, double *v0, double *v1, double *v2);
void bodies_release(cup2d_ctx *c);  // penalize.hip
void walk_plans_release(cup2d_ctx *c);  // advect.hip
// comm.hip
int comm_finalize_impl(cup2d_ctx *c);
// d_red[0 .. nsum) summed and d_red[2] maximised over the ranks in ONE all-gather, finished in rank order by the kernel that
// also runs the scalar update of `stage` (-1: none)
int comm_reduce_scalars(cup2d_ctx *c, int nsum, int with_max, int stage, int *host_status);
static inline bool overlapped(const cup2d_ctx *c) { return c->nghost > 0 && c->exchange && c->n_inner < c->nblocks; }

}  // namespace cup2d
