/* include/cup2d_hip.h -- C-ABI of libcup2d_hip.so, the MI355X (gfx950) backend for CUP2D's
 * per-block stencil hot path.  Plain C, plain pointers and sizes; no C++/torch types.
 *
 * Every entry point names the reference interface it replaces (file:line under
 * /root/reference).  The reference crosses to an accelerator through exactly one seam,
 * cuda.h's C++ class LocalSpMatDnVec (cuda.h:26-79) -- that seam is served by
 * libcup2d_spmat.so (cup2d_amd/csrc/local_spmat_hip.cpp, same mangled symbols as cuda.cu).
 * The block-operator call sites of main.cpp (computeA/computeB, main.cpp:3024-3125) are
 * `static` templates with no linkable symbol, so for them this header is the interface a
 * maintainer binds instead (INTEGRATION.md shows the edit).
 *
 * Data model.  One context owns device-resident slabs for the reference's seven fields
 * (main.cpp:3264-3278: tmp, chi, vel, vold, pres, pold, tmpV).  A slab is
 * [nblocks + nghost][8*8*dim] float64 in the reference's own per-block layout
 * (row-major iy*8+ix, vector components interleaved, main.cpp:510, 5497-5502), blocks in the
 * caller's order (the reference's Hilbert id order works unchanged).  Topology is a
 * neighbour table: nbr[b][0..3] = index of the same-level block to the W, E, S, N of block
 * b, or CUP2D_WALL at a domain wall (free-slip / Neumann, main.cpp:3131-3255).  Indices
 * >= nblocks address ghost blocks: copies of blocks owned by another rank, filled by
 * cup2d_halo_unpack.  All arithmetic is FP64 (main.cpp:24).
 *
 * Every function returns CUP2D_OK (0) or a negative cup2d_status; nothing aborts and
 * nothing falls back to the CPU.  Calls are asynchronous on the context's HIP stream
 * unless they return data to the host.
 */
#ifndef CUP2D_HIP_H
#define CUP2D_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CUP2D_BS 8           /* -D_BS_=8, Makefile:11 */
#define CUP2D_WALL (-1)

typedef struct cup2d_ctx cup2d_ctx;

typedef enum {
  CUP2D_OK = 0,
  CUP2D_ERR_ARG = -1,          /* bad argument */
  CUP2D_ERR_HIP = -2,          /* a HIP runtime call failed: cup2d_last_error() has the text */
  CUP2D_ERR_NODEVICE = -3,     /* no gfx950 device visible */
  CUP2D_ERR_UNSUPPORTED = -4,  /* valid in the reference, not built yet (AMR, bodies in solver) */
  CUP2D_ERR_COMM = -5          /* a communication callback failed */
} cup2d_status;

/* field ids in the order of var.F[] (main.cpp:3264-3278) */
typedef enum {
  CUP2D_TMP = 0,   /* scalar: Poisson rhs / vorticity */
  CUP2D_CHI = 1,   /* scalar: body indicator */
  CUP2D_VEL = 2,   /* vector */
  CUP2D_VOLD = 3,  /* vector */
  CUP2D_PRES = 4,  /* scalar */
  CUP2D_POLD = 5,  /* scalar */
  CUP2D_TMPV = 6,  /* vector: advect-diffuse rhs / udef / pressure-gradient increment */
  CUP2D_NFIELDS = 7
} cup2d_field;

/* arithmetic policy of the WENO5 kernel.
 * STRICT: operation-for-operation the reference expression (main.cpp:162-208, 5493-5502), IEEE
 *         division, no FMA contraction: bit-identical to the reference CPU functor.
 * FAST:   same algorithm, weights normalised with one division per reconstruction and FMAs;
 *         differs from STRICT by round-off only (tolerance stated in tests/test_advect_gpu.py). */
typedef enum { CUP2D_MATH_FAST = 0, CUP2D_MATH_STRICT = 1 } cup2d_math;

/* which blocks an operator sweeps -- the inner/halo split of computeA (main.cpp:3035-3057) */
typedef enum { CUP2D_BLOCKS_ALL = 0, CUP2D_BLOCKS_INNER = 1, CUP2D_BLOCKS_HALO = 2 } cup2d_phase;

/* ---------------------------------------------------------------- lifetime --------------- */
/* Replaces grid construction main.cpp:6508-6541 (per-block calloc) with device slabs.
 * nblocks owned blocks, nghost ghost blocks, nbr[nblocks][4] (W,E,S,N), h = cell size of the
 * (uniform) level (main.cpp:693).  n_inner: blocks [0,n_inner) touch no ghost block; pass
 * nblocks when nghost == 0.  device: HIP device ordinal. */
int cup2d_create(cup2d_ctx **ctx, int nblocks, int nghost, int n_inner, const int32_t *nbr, double h,
                 int device);
void cup2d_destroy(cup2d_ctx *ctx);
/* A context's device buffers come from a per-process pool: cup2d_destroy (and cup2d_clear_matrix, a new cup2d_set_amr)
 * hand them back to it, the next context takes them again -- a host that regrids builds a new context every few steps.
 * Buffers are always handed out zero-filled.  cup2d_trim_pool returns the idle buffers to the driver; CUP2D_POOL=0 turns
 * the pool off, CUP2D_POOL_MAX_GB (default 64) bounds what it keeps. */
int cup2d_trim_pool(void);
const char *cup2d_last_error(void);
const char *cup2d_version(void);
/* run every later call on this hipStream_t (NULL = the context's own stream) */
int cup2d_set_stream(cup2d_ctx *ctx, void *hip_stream);
int cup2d_get_stream(cup2d_ctx *ctx, void **hip_stream);
int cup2d_synchronize(cup2d_ctx *ctx);
int cup2d_set_math(cup2d_ctx *ctx, int math /* cup2d_math */);

/* ---------------------------------------------------------------- data movement ---------- */
/* blocks[i] = Info::block of block i (main.cpp:510): nblocks separately allocated 64*dim arrays */
int cup2d_upload(cup2d_ctx *ctx, int field, const double *const *blocks);
int cup2d_download(cup2d_ctx *ctx, int field, double *const *blocks);
/* same, one contiguous [nblocks][64*dim] host array */
int cup2d_upload_slab(cup2d_ctx *ctx, int field, const double *slab);
int cup2d_download_slab(cup2d_ctx *ctx, int field, double *slab);
/* raw device pointer of a slab ([nblocks+nghost][64*dim]) for callers that keep data on the GPU.  The pointer is valid
 * and names the same field until cup2d_destroy: no call of this library re-seats a slab (pold = pres of
 * main.cpp:7016-7021 is a device copy, the smoother's ping-pong ends with the contents in place). */
int cup2d_field_ptr(cup2d_ctx *ctx, int field, void **device_ptr);
int cup2d_fill(cup2d_ctx *ctx, int field, double value);
int cup2d_copy_field(cup2d_ctx *ctx, int dst_field, int src_field);

/* ---------------------------------------------------------------- block operators -------- */
/* computeA<VectorLab>(KernelAdvectDiffuse(), var.vel, 2)  main.cpp:6616 (functor 5441-5503):
 * tmpV = -dt*h*(u.D_weno5)u + nu*dt*Lap5(u) */
int cup2d_advect_diffuse_rhs(cup2d_ctx *ctx, double nu, double dt, int phase);
/* the whole RK2 block main.cpp:6607-6642 fused into two kernels:
 * vold = vel; vel = vold + 0.5*rhs(vel)/h^2; vel = vold + rhs(vel)/h^2.  tmpV is not written. */
int cup2d_advect_diffuse_rk2(cup2d_ctx *ctx, double nu, double dt);
/* the two halves of the above for callers that exchange halos in between (multi-GPU):
 * stage 1: VOLD <- vel + 0.5*rhs(vel)/h^2 is written to the scratch slab; stage 2 reads it. */
int cup2d_advect_diffuse_stage(cup2d_ctx *ctx, double nu, double dt, int stage /*1|2*/, int phase);
/* computeA<VectorLab>(KernelVorticity(), var.vel, 2) main.cpp:4659 (functor 3343-3366): tmp = curl(vel) */
int cup2d_vorticity(cup2d_ctx *ctx, int phase);
/* computeB<pressure_rhs,..>(.., var.vel, var.tmpV) main.cpp:7011 (functor 6105-6139):
 * tmp = 0.5h/dt * (div vel - chi * div tmpV).  use_bodies = 0 skips the chi/udef reads (tmpV=0). */
int cup2d_pressure_rhs(cup2d_ctx *ctx, double dt, int use_bodies, int phase);
/* computeA<ScalarLab>(pressure_rhs1(), var.pold, 1) main.cpp:7026 (functor 6209-6230): tmp -= Lap5(pold) */
int cup2d_laplacian_sub(cup2d_ctx *ctx, int phase);
/* main.cpp:7007-7026 in one call: tmp = pressure_rhs; pold = pres; pres = 0; tmp -= Lap5(pold) */
int cup2d_poisson_rhs(cup2d_ctx *ctx, double dt, int use_bodies);
/* computeA<ScalarLab>(pressureCorrectionKernel(), var.pres, 1) main.cpp:7178 (functor 6021-6043):
 * tmpV = -0.5*dt*h*grad(pres) */
int cup2d_pressure_correction(cup2d_ctx *ctx, double dt, int phase);
/* main.cpp:7180-7187: vel += tmpV/h^2 */
int cup2d_add_correction(cup2d_ctx *ctx);
/* main.cpp:7120-7187 in one call: pres = x - mean(x); pres += pold - mean(pres);
 * vel += (-0.5*dt*h*grad pres)/h^2, where x is the solver result (held in PRES). */
int cup2d_project(cup2d_ctx *ctx, double dt);

/* ---------------------------------------------------------------- scalars ---------------- */
/* main.cpp:6585-6591: max |vel| over all owned blocks */
int cup2d_max_abs_vel(cup2d_ctx *ctx, double *umax);
/* linf[b] = max |field| over the cells of block b, b < nblocks: the norm adapt() tags blocks by (main.cpp:4671-4690,
 * after cup2d_vorticity: field = CUP2D_TMP).  linf is a host array; scalar fields only. */
int cup2d_block_linf(cup2d_ctx *ctx, int field, double *linf);
/* main.cpp:6593-6595 */
int cup2d_compute_dt(cup2d_ctx *ctx, double nu, double cfl, double *dt);

/* ---------------------------------------------------------------- Poisson solve ---------- */
/* Replaces getVec + LocalSpMatDnVec::solveWithUpdate/solveNoUpdate (main.cpp:7114-7118,
 * cuda.cu:403-548) on a same-level block grid, matrix-free: b = TMP, x0 = PRES, result -> PRES.
 * Preconditioned BiCGSTAB with the reference's update order, breakdown restarts, best-iterate
 * tracking and Linf stopping rule; block-Jacobi preconditioner P_inv = -(A_loc)^-1
 * (main.cpp:6451-6488).  max_iter: the reference hard-codes 1000 (cuda.cu:438).
 * Outputs (may be NULL): iterations run, restarts, Linf residual of the returned iterate,
 * initial Linf residual. */
int cup2d_poisson_solve(cup2d_ctx *ctx, double max_error, double max_rel_error, int max_restarts,
                        int max_iter, int *iters, int *restarts, double *linf, double *linf_init);
/* y = A x on two scalar fields (A = 5-point graph Laplacian with Neumann walls: the matrix of
 * main.cpp:7034-7112 on a same-level grid, equal to pressure_rhs1's stencil) */
int cup2d_apply_A(cup2d_ctx *ctx, int dst_field, int src_field);
/* z = P_inv p per block (cuda.cu:484-486) on two scalar fields */
int cup2d_precond(cup2d_ctx *ctx, int dst_field, int src_field);
/* copy of the 64x64 preconditioner the context uses (row-major) */
int cup2d_get_P_inv(cup2d_ctx *ctx, double *P_inv_4096);
/* Install the caller's block preconditioner: the P_inv argument of LocalSpMatDnVec's constructor
 * (cuda.h:28-29, uploaded at cuda.cu:136-139), row-major 64x64, applied as z_b = P_inv p_b.  The matrix
 * main.cpp:6451-6488 builds (-(A_loc)^-1, recognised up to round-off) keeps the fast-diagonalisation
 * kernels; any other matrix is applied as a dense 64x64 product on the FP64 matrix cores. */
int cup2d_set_P_inv(cup2d_ctx *ctx, const double *P_inv_4096);
/* implementation of the block-Jacobi product: all three apply the same operator (round-off apart) */
typedef enum { CUP2D_PRECOND_LDS = 0, CUP2D_PRECOND_MFMA = 1, CUP2D_PRECOND_FD = 2 } cup2d_precond_kind;
int cup2d_set_precond(cup2d_ctx *ctx, int kind);

/* Organisation of one BiCGSTAB iteration (all variants run the recurrences of cuda.cu:403-548; they differ in
 * how many times a vector crosses HBM and by round-off):
 *   CUP2D_SOLVER_SWEEPS five fused sweeps, z and z2 stored (184 B/cell/iteration); every configuration.
 *   CUP2D_SOLVER_FUSED  tile-fused: P_inv on the FP64 matrix cores recomputed on tile edges, sweeps A+B and
 *                       C+D one launch each, x accumulated as x0 + P_inv y (136 B/cell/iteration; 112 in the
 *                       two-launch form, cup2d_set_solver_form); same-level stencil on one GPU or N ranks and the
 *                       assembled operator in its hybrid form -- elsewhere SWEEPS is used.
 * finish_in_kernel != 0: the last workgroup of a reducing sweep finishes the reduction and updates the device
 * scalars instead of a separate single-workgroup launch (ignored when an all-reduce callback is installed).
 * Defaults: FUSED where it applies, finish in the kernel. */
typedef enum { CUP2D_SOLVER_SWEEPS = 0, CUP2D_SOLVER_FUSED = 1 } cup2d_solver_kind;
int cup2d_set_solver(cup2d_ctx *ctx, int kind, int finish_in_kernel);
/* Form of the FUSED organisation (same recurrences; they differ by round-off and in bytes per iteration):
 *   CUP2D_FORM_AUTO  the default: EAB where it applies, FULL elsewhere (the process-wide default can be set with
 *                    CUP2D_FUSED_FORM = full | edge | eab)
 *   CUP2D_FORM_FULL  three launches per iteration (A+B, C+D, E), dense 64 x 64 block product: every configuration of FUSED
 *   CUP2D_FORM_EDGE  three launches, A P_inv v = v + ghost edges of P_inv v (built-in preconditioner, same-level stencil)
 *   CUP2D_FORM_EAB   two launches: C+D with the sums of the next beginning, and E together with the next iteration's A+B
 *                    (112 B/cell/iteration); finish in the kernel, built-in preconditioner, same-level stencil; one GPU, or
 *                    N ranks in the ghost-block form (whole boundary blocks of t, and of r', p'', nu'' in one message of
 *                    192 doubles per strip: cup2d_set_comm_strip_capacity) with two reductions over the ranks per iteration.
 *                    rho = rhat.r comes from the sums of C+D (rhat.s - omega rhat.t) instead of its own pass over r.
 *                    On the hybrid assembled operator of an adapted grid (one rank, finish in the kernel) EAB is available ON
 *                    REQUEST only -- two sweeps + two rows launches per iteration (k_edge HYB + k_hyb_rows); AUTO keeps FULL
 *                    there (three sweeps + two rows launches: measured faster on the 63 k-block grid). */
typedef enum { CUP2D_FORM_AUTO = 0, CUP2D_FORM_FULL = 1, CUP2D_FORM_EDGE = 2, CUP2D_FORM_EAB = 3 } cup2d_fused_form;
int cup2d_set_solver_form(cup2d_ctx *ctx, int form);
/* N ranks, two launches per iteration: which organisation of a reduction point the next solves take (every rank must ask for
 * the same one).  deferred: 1 the reduction records ride in the ghost blocks' send/recv group and the scalar update happens in
 * the consumer sweep (in-library communicator with in-place ghost blocks on every rank, else ignored), 0 an all-gather and a
 * one-wave scalar kernel per reduction point; split: 1 a sweep as halo-set-first + inner launches with the ghost blocks
 * travelling in between (computeA's split, main.cpp:3035-3057), 0 one launch with the exchange behind it.  -1 = the process
 * default (CUP2D_DEFER_SCALARS, CUP2D_SWEEP_SPLIT; deferred, unsplit).  (1, 1) = "overlap": split sweeps whose halo-set launch
 * runs the pending scalar update in its prologue; the ghost blocks travel on the communication stream while the inner launch
 * runs and the records go round in one all-gather behind it.  With the same `split` the deferred and the undeferred forms are
 * the same numbers bit for bit; split changes the order of the partial sums (round-off).  bench.py --gpus N times all four so
 * that one N-GPU run says which one the links favour.  cup2d_get_last_solver_form reports merge = 2 (undeferred), 3 (deferred,
 * unsplit) or 4 (overlap). */
int cup2d_set_nrank_organisation(cup2d_ctx *ctx, int deferred, int split);
/* the organisation the last cup2d_poisson_solve actually ran (FUSED falls back to SWEEPS where it does not apply) */
int cup2d_get_last_solver(cup2d_ctx *ctx, int *kind);
/* what the last FUSED solve ran in detail (diagnostic, for tests that must know which organisation they pinned): form =
 * CUP2D_FORM_FULL | _EDGE | _EAB as resolved for that solve; merge = 0 finish launches, 1 finish in the kernel on one GPU,
 * 2 finish in the kernel + reductions over the ranks (an all-gather and a one-wave scalar kernel per reduction point), 3 the
 * same on the in-library communicator with the reduction records inside the ghost blocks' send/recv group and the scalar
 * update in the consumer sweep (one pack launch + one RCCL kernel per reduction point); handover = bit mask by kind of sweep (bit 0 A+B, 1 C+D, 2 E+A+B,
 * 3 C+D') of the sweeps whose sibling waves handed z edges through LDS.  All zero after a five-sweep solve. */
int cup2d_get_last_solver_form(cup2d_ctx *ctx, int *form, int *merge, int *handover);
/* The placement search of the solver's vectors (krylov_fused.hip tune_placement: the durations of the two launches of an
 * iteration come in two modes that follow where the eleven vectors they stream lie in device memory; the first two-launch solve
 * of a context on a grid of 2048^2 cells and more tries CUP2D_PLACEMENT_TRIES = 8 complete sets -- up to three times as many until a fast one has shown or been made (10.5 % below the median) -- and keeps the fastest).
 * candidates = sets timed (0: no search ran), the microseconds per iteration of the kept set, of the slowest set seen and of
 * the set the context was created with.  Diagnostic (bench.py "placement"). */
int cup2d_get_placement(cup2d_ctx *ctx, int *candidates, double *kept_us, double *slowest_us, double *first_us);
/* Diagnostic: the reference returns the BEST iterate in the max norm (cuda.cu:535-547), which within a capped number of
 * iterations may still be the initial guess -- nothing of the iterations is then visible in PRES.  With keep_last on, a
 * solve also keeps its LAST iterate (x0 + P_inv y for the fused organisation) in a solver scratch vector;
 * cup2d_solver_last_iterate copies it to a scalar field and reports the max norm of the RECURRENCE residual r after the
 * last iteration (may be NULL): recomputing max|b - A x_last| from the field and comparing is a check of every iteration
 * the solve ran (bench.py "verified"). */
int cup2d_solver_keep_last(cup2d_ctx *ctx, int on);
int cup2d_solver_last_iterate(cup2d_ctx *ctx, int dst_field, double *linf_recurrence);

/* ---------------------------------------------------------------- assembled operator ----- */
/* The seam the reference itself crosses (cuda.h LocalSpMatDnVec): instead of the 5-point stencil on
 * the neighbour table, cup2d_poisson_solve / cup2d_apply_A use the COO matrix the caller assembled
 * (main.cpp:7034-7112) -- needed for coarse-fine rows.  Replaces BiCGSTABSolver::updateAll's upload +
 * cuSPARSE descriptors (cuda.cu:206-296) and the SpMV pair of cuda.cu:344-402.
 * row/col are LOCAL int32 indices as produced by LocalSpMatDnVec::make (cuda.cu:661-688): rows in
 * [0, 64*nblocks), columns in [0, 64*nblocks + halo); columns >= 64*nblocks address halo entries
 * that the exchange callback (cup2d_set_comm) delivers behind the vector.  halo <= 64*nghost.
 * Local and boundary triplets (cuda.cu's loc_/bd_ lists) go in one list.  The matrix is converted
 * to sliced-ELL on the host; order of a row's entries is kept. */
int cup2d_set_matrix_coo(cup2d_ctx *ctx, int halo, long long nnz, const int32_t *row, const int32_t *col,
                         const double *val);
/* back to the matrix-free stencil; also drops the gather list */
int cup2d_clear_matrix(cup2d_ctx *ctx);
/* send_pack_idx_ of cuda.h:76 / send_buff_pack cuda.cu:338-343: before every operator application
 * device_send_buffer[i] = vec[idx[i]], i < nsend, is handed to the exchange callback with
 * strip_doubles = 1 and device_recv = &vec[64*nblocks].  Call after cup2d_set_matrix_coo. */
int cup2d_set_gather(cup2d_ctx *ctx, int nsend, const int32_t *idx);
/* How the installed operator is applied (any pointer may be NULL).  plain_blocks: blocks whose 64 rows are exactly the
 * same-level 5-point rows -- applied matrix-free from the four neighbour ids, no stored entries; stored_entries: the
 * sliced-ELL entries kept for the other blocks (coarse-fine rows, halo columns); general_tile_blocks: blocks that sit in
 * a tile of 16 consecutive blocks holding at least one such block -- the tile-fused sweeps (CUP2D_SOLVER_FUSED) keep
 * z = P_inv v of every other tile on the chip and apply the rows of these from memory in a second launch per sweep. */
int cup2d_matrix_stats(cup2d_ctx *ctx, int *plain_blocks, int *general_tile_blocks, long long *stored_entries);

/* ---------------------------------------------------------------- block-AMR --------------- */
/* Adapted grids (BASELINE.json configs[4]; the reference's Info::level / tree, main.cpp:504-517, 2197-2198).
 * level[b] = refinement level of block b (cell size h0 / 2^level, main.cpp:693); for every side W,E,S,N of b:
 *   kind[b][s]   CUP2D_AMR_WALL, _SAME (neighbour nbr2[b][s][0] on the same level), _COARSER (nbr2[b][s][0] is one
 *                level coarser; half[b][s] = 0|1 = which half of its face b touches, in increasing y for W/E and
 *                increasing x for S/N), _FINER (the two blocks nbr2[b][s][0..1] one level finer, ordered along the face)
 * 2:1 balance is the caller's invariant (the reference's adapt() enforces it, main.cpp:4734-4861).
 * Once set, the block operators run their AMR form -- ghost cells across coarse-fine faces as BlockLab::load /
 * post_load builds them (main.cpp:2270-2933) and the flux correction of prepare0/fillcases (main.cpp:1564-1849):
 * cup2d_advect_diffuse_rhs, cup2d_advect_diffuse_rk2 (the reference's un-fused stage sequence, per-block h),
 * cup2d_laplacian_sub, cup2d_apply_A, cup2d_pressure_rhs, cup2d_poisson_rhs, cup2d_pressure_correction,
 * cup2d_project (volume-weighted means), cup2d_vorticity, cup2d_compute_dt (finest h), and cup2d_poisson_solve /
 * cup2d_step once the caller's coarse-fine matrix rows are installed with cup2d_set_matrix_coo (main.cpp:7034-7113).
 * phase must be CUP2D_BLOCKS_ALL; the nbr table of cup2d_create is ignored.  cup2d_advect_diffuse_stage runs one stage of
 * that un-fused sequence (stage 1: VOLD = vel, then the mid-point velocity is left in VEL -- not in a scratch slab as on a
 * uniform grid; stage 2: vel = VOLD + rhs(vel) / h^2; TMPV holds the stage's rhs; stages 1 + 2 = cup2d_advect_diffuse_rk2 to
 * the bit).  cup2d_poisson_solve / cup2d_step without an installed operator: on one rank the library assembles the rows from
 * these tables itself (cup2d_amr_install_poisson: plain same-level blocks stay matrix-free, stored rows only where the grid is
 * irregular); on N ranks they return CUP2D_ERR_UNSUPPORTED until the operator and the gather list are installed. */
typedef enum { CUP2D_AMR_WALL = 0, CUP2D_AMR_SAME = 1, CUP2D_AMR_COARSER = 2, CUP2D_AMR_FINER = 3 } cup2d_amr_kind;
int cup2d_set_amr(cup2d_ctx *ctx, double h0, const int32_t *level, const int32_t *kind, const int32_t *nbr2,
                  const int32_t *half);
/* Adapted grids on N ranks (BASELINE.json configs[4]; the reference: contiguous Hilbert ranges per rank main.cpp:6494-6504,
 * remote blocks through the synchroniser 1971-2142, 2582-2684, flux faces 1819-1825).  A rank's context holds its owned
 * blocks plus GHOST blocks (nghost of cup2d_create) = copies of every remote block its kernels read: the face neighbours
 * of its blocks and, across a coarser neighbour, that neighbour's tangential neighbours (two rings).  The tables of
 * cup2d_set_amr then cover owned + ghost blocks (level[nblocks + nghost], ...; sides of a ghost block whose neighbour the
 * rank does not hold: CUP2D_AMR_WALL).  With a halo plan that lists whole blocks (cup2d_halo_plan: owned blocks to send,
 * ghost blocks to receive, face 0) and a communicator (cup2d_comm_init / cup2d_set_comm), every block operator first
 * refreshes the ghost copies of the field it reads (whole blocks), the flux-correction face arrays of fine blocks travel
 * to the rank that owns the coarse side, reductions go over the ranks, and cup2d_poisson_solve takes an assembled operator
 * whose columns >= 64 * nblocks address the ghost blocks' cells in ghost order (cup2d_set_matrix_coo with
 * halo = 64 * nghost; cup2d_set_gather listing the 64 cells of every sent block in plan order).
 * cup2d_amr_set_finest_level: the finest level present on ANY rank (dt uses the finest cell size, main.cpp:6580-6583). */
int cup2d_amr_set_finest_level(cup2d_ctx *ctx, int level_finest);

/* The Poisson matrix the reference assembles on an adapted grid (the serial host loop main.cpp:7034-7112 with
 * Solver::makeFlux / interpolate / D1 / D2, main.cpp:5915-5997), from the topology tables of cup2d_set_amr.  Host-side,
 * regrid-time, no context and no GPU needed.  Returns the number of triplets; when row, col, val are non-NULL (capacity
 * cap) they are filled: rows/columns numbered 64 * block + 8 * iy + ix, sorted by (row, column), duplicate columns of a
 * row summed in arrival order like SpRowInfo::mapColVal (cuda.h:1-24).  Negative cup2d_status on error.  Feed the result
 * to cup2d_set_matrix_coo. */
long long cup2d_amr_poisson_coo(int nblocks, const int32_t *kind, const int32_t *nbr2, const int32_t *half, long long cap,
                                int32_t *row, int32_t *col, double *val);
/* The same operator installed directly (what a host does after every regrid, main.cpp:7034-7115): assembled from the
 * tables cup2d_set_amr was given, in the form cup2d_set_matrix_coo would arrive at -- but no row is built for a block whose
 * four sides are walls or same-level blocks of this rank (94 % of a typical grid), only for the blocks with coarse-fine
 * sides or, on N ranks, ghost neighbours (their cells are halo columns 64 * (nblocks + g) + cell; cup2d_set_gather
 * afterwards as with cup2d_set_matrix_coo).  Bit-identical in its action to the triplet route; 0.01 s instead of 0.14 s on
 * a 63 k-block grid. */
int cup2d_amr_install_poisson(cup2d_ctx *ctx);

/* ---- Poisson smoother sweep and residual (the streaming 5-point kernels of the path) ----
 * A = the matrix-free Poisson operator: pressure_rhs1's 5-point sum (main.cpp:6209-6230) with the homogeneous-Neumann
 * walls of the assembled matrix (main.cpp:7034-7112), x = PRES, b = TMP.
 * cup2d_jacobi_sweeps: nsweeps times x <- x + (b - A x) * (omega / diag(A)), diag(A) = -(number of neighbours of the
 *   cell).  The reference has no smoother (its solve is BiCGSTAB, cup2d_poisson_solve); this is the "50 Jacobi
 *   pressure iters/step" unit of BASELINE.json configs[1] (24 B/cell).  POLD is the second buffer: the
 *   iterate before the last sweep is left there.  linf (may be NULL) = max|b - A x| of that previous iterate.
 * cup2d_poisson_residual: POLD = b - A x, linf = max|b - A x| (the reference's stopping norm, cuda.cu:303-311).
 * Same-level grids with the matrix-free operator only (CUP2D_ERR_UNSUPPORTED otherwise); with a communicator the
 * width-1 halo of x is exchanged before every sweep and linf is reduced over the ranks. */
int cup2d_jacobi_sweeps(cup2d_ctx *ctx, double omega, int nsweeps, double *linf);
int cup2d_poisson_residual(cup2d_ctx *ctx, double *linf);

/* ---- regridding on the host (no context, no GPU): the reference's adapt() (main.cpp:4657-5440) on dense tables ----
 * blocks = [nblocks][3] leaves (level, i, j) of a bpdx x bpdy base grid.  Fields are per-block arrays
 * [nblocks][64 * dim], components interleaved, in the order of `blocks` (what cup2d_download_slab returns).
 *
 * cup2d_amr_tables: the topology tables cup2d_set_amr takes (replaces the tree / Znei / Zchild look-ups, main.cpp:672-738);
 * CUP2D_ERR_ARG when a side is neither wall, leaf, coarser leaf nor two finer leaves (grid not 2:1 balanced).
 *
 * cup2d_amr_validate_states: states[] (0 leave, 1 refine, 2 compress -- the tags of main.cpp:4671-4703) made
 * consistent in place exactly as main.cpp:4718-4861 does: 2:1 balance across faces and corners, finest level first;
 * four siblings compress together or not at all.
 *
 * cup2d_amr_regrid: applies validated states.  A refined block becomes its four children, prolonged from its
 * tensorial halo-1 tile (BlockLab Stencil{-1,-1,2,2,true}, main.cpp:4906-4913, 4981-5032) of the OLD grid; four
 * compressing siblings become their parent (2 x 2 means, main.cpp:5149-5166).  is_vector selects the wall condition
 * of the tile (VectorLab / ScalarLab).  Returns the new block count; with new_blocks == NULL nothing else is done;
 * otherwise new_blocks[cap][3] and new_fields[f][cap][64 * dims[f]] are filled, ordered along the Hilbert curve of
 * the finest level (main.cpp:1550-1562).  Bit-identical to the reference's adapt() (tests/test_amr.py). */
int cup2d_amr_tables(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int32_t *kind, int32_t *nbr2, int32_t *half);
int cup2d_amr_validate_states(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, int32_t *states);
long long cup2d_amr_regrid(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *states,
                           int nfields, const double *const *fields, const int32_t *dims, const int32_t *is_vector,
                           long long cap, int32_t *new_blocks, double *const *new_fields);
/* The same for a host that keeps the fields on the device (only the blocks that change cross PCIe):
 * cup2d_amr_regrid_plan: the new leaves, per new block the old block it is an unchanged copy of (src_of_new[cap], -1 for a
 *   prolonged or restricted block) and needed_old[nblocks] = 1 for every old block such a block is computed from (refined
 *   parents and every leaf overlapping the 3 x 3 block neighbourhood of one -- the tensorial halo-1 tile --, compressing
 *   siblings).  new_blocks == NULL: only the count.
 * cup2d_amr_regrid_changed: cup2d_amr_regrid that reads only the needed blocks of `fields` (full-size arrays, the other
 *   blocks may hold anything) and writes only the prolonged / restricted blocks of new_fields.
 * cup2d_download_blocks / cup2d_upload_blocks: field data of a list of blocks, host[n][64 * dim].
 * cup2d_copy_blocks: block dst_blocks[k] of `ctx` = block src_blocks[k] of `src` (the context of the old grid, same device). */
long long cup2d_amr_regrid_plan(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *states,
                                long long cap, int32_t *new_blocks, int32_t *src_of_new, int32_t *needed_old);
long long cup2d_amr_regrid_changed(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *states,
                                   int nfields, const double *const *fields, const int32_t *dims, const int32_t *is_vector,
                                   long long cap, int32_t *new_blocks, double *const *new_fields);
/* The same for ONE RANK of a partitioned leaf list (a contiguous range of the new list per rank, main.cpp:6494-6504): only the
 * new blocks at positions [new_lo, new_hi) count.  Plan outputs (each may be NULL): new_blocks[cap][3] -- the whole new leaf
 * list, every rank derives the same one; with new_blocks == NULL only the count is returned --, src_of_new[cap] as above,
 * needed_old[nblocks] = 1 for the old blocks the prolonged /
 * restricted blocks OF THE RANGE are computed from.  With nfields > 0 the changed blocks of the range are computed:
 * fields[f] are COMPACT arrays holding just the blocks this rank fetched -- old block k at fields[f] + slot_of_old[k] * 64 *
 * dims[f] -- and new_fields[f][new_hi - new_lo][64 * dims[f]] receives the prolonged / restricted blocks at (position -
 * new_lo); unchanged copies are not written (src_of_new says where they come from: the caller moves them, on the device or
 * between ranks).  Replaces the per-rank refine / compress of main.cpp:5055-5130 and the data side of its block migration
 * (5198-5424): no rank reads a block outside its needed_old and src_of_new sets.  slot_of_old[k] < 0 for a block the range
 * reads is CUP2D_ERR_ARG (the caller did not fetch what the plan names); the rows' upper bound is the caller's to keep. */
long long cup2d_amr_regrid_local(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *states,
                                 long long new_lo, long long new_hi, long long cap, int32_t *new_blocks, int32_t *src_of_new,
                                 int32_t *needed_old, int nfields, const double *const *fields, const int32_t *slot_of_old,
                                 const int32_t *dims, const int32_t *is_vector, double *const *new_fields);
/* The regrid with the fields ON THE DEVICE (prolongation main.cpp:4981-5032, restriction 5149-5166 as
 * kernels).  src: the context on the old leaf list blocks[nblocks][3] with its tables set (cup2d_set_amr); dst: a context
 * created on the NEW leaf list in cup2d_amr_regrid_plan's order (same device).  For every field listed in fields[nfields]
 * (CUP2D_VEL, ...) the new slab is written by one launch: unchanged blocks copied, compressed sibling groups averaged, the
 * four children of every refined block prolonged from the parent's tensorial halo-1 tile (side cells by the closed forms
 * of the block operators from src's device tables, corner cells from descriptors the host derives from the leaf list).
 * No field crosses PCIe; bit-identical to cup2d_amr_regrid (same expressions, -ffp-contract=off).  One rank (no ghost
 * blocks); the N-rank regrid is cup2d_amr_regrid_local.  The caller destroys src afterwards. */
int cup2d_amr_regrid_device(cup2d_ctx *dst, cup2d_ctx *src, int nblocks, const int32_t *blocks, int bpdx, int bpdy,
                            int level_max, const int32_t *states, int nfields, const int32_t *fields);
/* The tables cup2d_amr_regrid_device gives its kernel, without a GPU (for inspection and for the CPU replay of the kernel in
 * tests/test_amr.py): jobs[njobs][8] = {type, a0..a4, i0, j0} -- type 0 COPY {dst, src}, 1 RESTRICT {dst, s00, s10, s01, s11},
 * 2 PROLONG {src, d0, d1, d2, d3, -, i0, j0} (the prolong jobs come first; job k's corner record is number k) --,
 * corners[nprolong][4][12] = {kind, flags, nine coarse cells (block << 1 | averaged, -1: none), pad}, kind 0 wall (flags bit 1:
 * y wall) / 1 same-level cell of block c[2] / 2 TestInterp on the nine coarse cells / 3 2 x 2 mean in finer block c[2] / 4 none.
 * jobs == NULL: counts only.  Returns njobs. */
long long cup2d_amr_regrid_jobs(int nblocks, const int32_t *blocks, int bpdx, int bpdy, int level_max, const int32_t *states,
                                long long cap_jobs, int32_t *jobs, long long cap_prolong, int32_t *corners, long long *nprolong);
/* N ranks, computeA's inner / halo split (main.cpp:3035-3057) on an adapted grid: which owned blocks [0, nowned) of tables that
 * cover ntotal = owned + ghost blocks read a cell of a ghost block in the operators of family `set` (CUP2D_CELLS_HALO1 |
 * CUP2D_CELLS_HALO3) -- by the kernels' own ghost expressions, as cup2d_amr_trace_reads.  reads_ghost[nowned] = 0 | 1.  These are
 * the blocks CUP2D_BLOCKS_HALO sweeps (the others: CUP2D_BLOCKS_INNER).  Host routine, no context. */
int cup2d_amr_blocks_reading_ghosts(int nowned, int ntotal, const int32_t *kind, const int32_t *nbr2, const int32_t *half, int set,
                                    int32_t *reads_ghost);
int cup2d_download_blocks(cup2d_ctx *ctx, int field, int n, const int32_t *blocks, double *host);
int cup2d_upload_blocks(cup2d_ctx *ctx, int field, int n, const int32_t *blocks, const double *host);
int cup2d_copy_blocks(cup2d_ctx *ctx, cup2d_ctx *src, int field, int n, const int32_t *dst_blocks, const int32_t *src_blocks);

/* ---------------------------------------------------------------- whole step ------------- */
/* One pass of the body-free time-loop body main.cpp:6576-7187:
 * dt -> RK2 advect-diffuse -> Poisson rhs -> solve -> projection.  Outputs may be NULL.
 * The projection is enqueued behind the solve's last pass before the host has seen the solve end (one host wait per step):
 * when the step returns an error (a failed solve, a lost hand-over, a communicator error) VEL, PRES and the cached max|u|
 * partials are UNDEFINED -- restore the fields (cup2d_upload) before the next call. */
int cup2d_step(cup2d_ctx *ctx, double nu, double cfl, double max_error, double max_rel_error,
               int max_restarts, int max_iter, double *dt, int *iters, double *linf);

/* ---------------------------------------------------------------- multi-GPU halos -------- */
/* Replaces sync1/pack/unpack_subregion (main.cpp:1971-2142, 58-110) for same-level faces.
 * A plan lists, in send-buffer order, (owned block, face) strips to pack and, in
 * recv-buffer order, (ghost block, face) strips to unpack; face = 0..3 = W,E,S,N edge of the
 * SOURCE block (a ghost block holds a copy of the remote block's cells at the same positions).
 * A strip is width x 8 cells x dim doubles, rows in iy order. */
int cup2d_halo_plan(cup2d_ctx *ctx, int nsend, const int32_t *send_block, const int32_t *send_face,
                    int nrecv, const int32_t *recv_block, const int32_t *recv_face);
int cup2d_halo_pack(cup2d_ctx *ctx, int field, int width, double *device_send_buffer);
int cup2d_halo_unpack(cup2d_ctx *ctx, int field, int width, const double *device_recv_buffer);
/* same for the solver's internal Krylov vector (which = 0: z, 1: z2) */
int cup2d_halo_pack_vec(cup2d_ctx *ctx, const double *device_vec, int dim, int width, double *device_send_buffer);
int cup2d_halo_unpack_vec(cup2d_ctx *ctx, double *device_vec, int dim, int width, const double *device_recv_buffer);

/* Communication callbacks so that one code path serves 1 and N GPUs (MPI_Allreduce /
 * Isend/Irecv sites main.cpp:6583-6592, 7138, 7162, cuda.cu:371-375, 448-534).
 * exchange : send_buffer holds nsend strips of `strip_doubles` doubles, packed by work already
 *            enqueued on `stream`.  The callback starts the transfer (RCCL send/recv, typically on
 *            its own communication stream ordered after `stream`) and returns without blocking.
 * wait     : makes `stream` wait until recv_buffer holds the nrecv strips of the last exchange.
 *            May be NULL when exchange itself orders its completion on `stream`.
 *            Between the two calls the library sweeps the inner blocks -- the overlap of
 *            computeA (main.cpp:3035-3057) -- and after wait + unpack the halo blocks.
 * allreduce: in place on `count` doubles of the device reduction buffer, op 0 = sum, 1 = max,
 *            ordered on `stream`. */
#define CUP2D_MAX_STRIP_DOUBLES 192
typedef int (*cup2d_exchange_fn)(void *user, double *device_send, double *device_recv, int strip_doubles,
                                 void *hip_stream);
typedef int (*cup2d_wait_fn)(void *user, void *hip_stream);
typedef int (*cup2d_allreduce_fn)(void *user, double *device_buf, int count, int op, void *hip_stream);
/* device_send_buffer / device_recv_buffer: caller-owned device memory for nsend / nrecv strips of
 * the widest exchange: CUP2D_MAX_STRIP_DOUBLES = 192 doubles per strip (whole blocks of three Krylov vectors in one
 * message; the WENO halo is 3 layers x 8 cells x 2 components = 48).
 * device_reduce_buffer: caller-owned 8 doubles used for every reduction handed to allreduce
 * (NULL keeps the context's own). */
int cup2d_set_comm(cup2d_ctx *ctx, cup2d_exchange_fn exchange, cup2d_wait_fn wait, cup2d_allreduce_fn allreduce,
                   void *user, double *device_send_buffer, double *device_recv_buffer, double *device_reduce_buffer);
/* What the caller's buffers really hold, in doubles per strip.  cup2d_set_comm alone promises CUP2D_MIN_STRIP_DOUBLES = 128
 * (the contract of the first two rounds: whole blocks of two Krylov vectors in one message), which is what every
 * organisation of the solver needs by default (the two-launch organisation sends nu' and p' in one message once per solve and
 * single vectors afterwards: r' and p'' of the ghost blocks are formed by the receiving rank); with CUP2D_GHOST_LOCAL=0 it sends
 * whole blocks of THREE vectors in one message (192 doubles per strip) and is then chosen only when the capacity says the
 * buffers take it -- otherwise the three-launch form runs.  Call it
 * after cup2d_set_comm; values below CUP2D_MIN_STRIP_DOUBLES are an error.  The in-library communicator (cup2d_comm_init)
 * owns its buffers and sets CUP2D_MAX_STRIP_DOUBLES itself. */
#define CUP2D_MIN_STRIP_DOUBLES 128
int cup2d_set_comm_strip_capacity(cup2d_ctx *ctx, int doubles_per_strip);

/* ---- strips on adapted grids: cell plans (main.cpp:2053-2125 pack with fine->coarse extent, 2582-2684 remote unpack) ----
 * The ghost blocks of an adapted grid on N ranks travel whole through the plan of cup2d_halo_plan unless the caller also
 * installs CELL PLANS: for one family of operators, the cells of the sent blocks the receiving rank's kernels actually read.
 *   CUP2D_CELLS_HALO1   cup2d_laplacian_sub / apply_A / pressure_rhs / pressure_correction / vorticity (one ghost layer: the
 *                       edge row of a same-level block, two rows of a finer one, four cells of a coarser one)
 *   CUP2D_CELLS_HALO3   the WENO tile of cup2d_advect_diffuse_rhs / _rk2 (three layers; 3 x 3 coarse cells around the third
 *                       layer across a coarser side, which reach into that block's tangential neighbour)
 *   CUP2D_CELLS_MATRIX  the columns of the rank's rows of the Poisson matrix (main.cpp:7034-7112) in ghost blocks: the
 *                       Krylov vector's exchange, twice per BiCGSTAB iteration
 * cup2d_amr_trace_reads (host only, no context): which cells those are.  kind / nbr2 / half = tables of cup2d_set_amr over
 *   `nblocks` blocks in ANY consistent numbering (every rank holds the global leaf list, so: the global one); readers = the
 *   blocks whose operators are meant; mask[nblocks] receives, OR-ed in, bit (8 iy + ix) of every cell of every OTHER block
 *   that the readers' kernels (set 0, 1) or matrix rows (set 2) touch.  It runs the kernels' own ghost expressions
 *   (csrc/amr_ghost.h) and the library's own row assembly with a recording accessor: nothing to keep in step by hand.
 *   Both ends of a link derive the same list from it -- the receiver with its blocks as readers, the sender with its ghost
 *   copies of the receiver's blocks as readers -- in (global block, cell) order; no negotiation round.
 * cup2d_halo_plan_cells: send_cell[nsend] = 64 * owned block + cell, receive_cell[nrecv] = 64 * ghost block + cell (local
 *   numbering of the context), peers in the order of the block plan.  nsend = nrecv = -1 removes the plan of that set; so
 *   does a new cup2d_halo_plan (the cell lists are subsets of its blocks: install them after it).  Empty lists (0, 0) are a
 *   plan like any other: every rank of a run installs the set or none does.
 *   With a plan for a set the exchange callback is called with strip_doubles = CUP2D_CELL_STRIP(set, dim) < 0: the message
 *   unit is one cell of dim doubles, and the per-peer offsets and counts are those of the set's cell lists
 *   (cup2d_comm_set_cell_counts tells them to the in-library communicator); device_send / device_recv are the buffers of
 *   cup2d_set_comm (a cell list never outgrows them: it is a subset of the plan's blocks).  For CUP2D_CELLS_MATRIX the
 *   gather list of cup2d_set_gather must be the same send list; the received cells are scattered into the vector's ghost
 *   blocks (columns keep their meaning: 64 * ghost block + cell).
 * Every operator's result is unchanged to the bit: the cells not delivered are cells no kernel reads. */
#define CUP2D_CELLS_HALO1 0
#define CUP2D_CELLS_HALO3 1
#define CUP2D_CELLS_MATRIX 2
#define CUP2D_CELL_STRIP(set, dim) (-(16 * (set) + (dim)))
#define CUP2D_CELL_STRIP_SET(strip_doubles) ((-(strip_doubles)) >> 4)
#define CUP2D_CELL_STRIP_DIM(strip_doubles) ((-(strip_doubles)) & 15)
int cup2d_amr_trace_reads(int nblocks, const int32_t *kind, const int32_t *nbr2, const int32_t *half, int nreaders,
                          const int32_t *readers, int set, uint64_t *mask);
int cup2d_halo_plan_cells(cup2d_ctx *ctx, int set, int nsend, const int32_t *send_cell, int nrecv, const int32_t *receive_cell);
/* per peer of cup2d_comm_init (same order): first cell and number of cells in the set's send and receive lists */
int cup2d_comm_set_cell_counts(cup2d_ctx *ctx, int set, int npeers, const int32_t *send_offset, const int32_t *send_count,
                               const int32_t *recv_offset, const int32_t *recv_count);

/* ---- penalisation with host-supplied bodies (main.cpp:6643-7006) ----
 * A body is what the reference keeps per shape: for every block the shape touches an Obstacle with the shape's own
 * indicator chi[8][8] and deformation velocity udef[8][8][2] (main.cpp:3283-3286; produced on the host by the shape
 * model, which stays with the caller), and the shape's centre of mass.  CUP2D_CHI holds the field chi (var.chi).
 * cup2d_body_set: body number `body` (0, 1, ...): blocks[nblk] = indices of the touched blocks, ascending (the order of the
 *   reference's block loop), origin[nblk][2] = Info::origin of each, chi[nblk][64], udef[nblk][64][2], centre of mass.
 * cup2d_body_momentum: main.cpp:6643-6702 -- the seven penalisation moments PM, PJ, PX, PY, UM, VM, AM of the body over
 *   CUP2D_VEL (the integrands on the GPU, added up on the host in the reference's order; all-reduced over the ranks) and
 *   the 3 x 3 LU solve for uvw = (u, v, omega); integrals (may be NULL) receives the seven sums.
 * cup2d_penalize: main.cpp:6944-7006 with the bodies' velocities uvw[nbodies][3] (the caller may have passed them
 *   through its collision model, main.cpp:6703-6943): CUP2D_VEL blended towards the body velocity where a body's chi
 *   dominates, then CUP2D_TMPV = sum of the dominating bodies' udef (the u_def of cup2d_pressure_rhs).  Without bodies
 *   it only clears CUP2D_TMPV.  Bit-identical to the reference's single-threaded loops; works on adapted grids
 *   (cell size per block from cup2d_set_amr). */
int cup2d_body_set(cup2d_ctx *ctx, int body, int nblk, const int32_t *blocks, const double *origin, const double *chi,
                   const double *udef, double cx, double cy);
int cup2d_body_clear(cup2d_ctx *ctx);
int cup2d_body_momentum(cup2d_ctx *ctx, int body, double lambda, double dt, double *uvw /* [3] */, double *integrals /* [7] */);
int cup2d_penalize(cup2d_ctx *ctx, double lambda, double dt, const double *uvw);

/* ---- the communicator inside the library: RCCL over xGMI (one process per GPU) ----
 * Replaces the MPI of the reference on this path: Irecv / Isend / Waitall of the synchroniser (main.cpp:2040-2047,
 * 2133-2139; cuda.cu:365-380) by ncclRecv / ncclSend pairs of one ncclGroup on a second HIP stream, ordered behind the
 * pack kernel by an event and awaited only where the library unpacks (the inner blocks are swept meanwhile, computeA's
 * overlap main.cpp:3035-3057); MPI_Allreduce of max|u|, the pressure means and the solver's dot products and norms
 * (main.cpp:6583-6592, 7138, 7162; cuda.cu:445-449, 491-493, 513-515, 533-534) by ncclAllReduce of <= 3 doubles -- or one
 * ncclAllGather where two sums and a max are due together: three collectives per BiCGSTAB iteration, the reference has four.
 * No host language between a cup2d_* call and its return.
 *
 * cup2d_comm_unique_id: one rank creates the rendezvous token (two ncclUniqueIds: exchange and reduction communicators)
 *   and hands the bytes to every rank by whatever it has (MPI_Bcast, a file, torch.distributed's store).
 * cup2d_comm_init: collective over all ranks, after cup2d_halo_plan.  Peer p = rank peer_rank[p]: entries
 *   [send_offset[p], +nstrips[p]) of the plan's send list go out to it, entries [recv_offset[p], +nstrips_recv[p]) of its
 *   receive list come in; both ends enumerate a link's strips in the same order.  The library owns the message buffers
 *   and the communication stream; it installs itself where cup2d_set_comm installs callbacks.
 * cup2d_halo_exchange: pack, exchange, unpack of the ghost strips of a field (width cell layers, 1..8) -- sync1 of
 *   main.cpp:1971-2142 for callers that run single block operators; cup2d_step and the solver do their own, overlapped.
 * cup2d_comm_stats: ranks, peers of this rank, and how many exchanges / all-reduces / all-gathers were issued so far. */
#define CUP2D_COMM_ID_BYTES 256
int cup2d_comm_unique_id(void *id_bytes /* [CUP2D_COMM_ID_BYTES] */);
int cup2d_comm_init(cup2d_ctx *ctx, int nranks, int rank, const void *id_bytes, int npeers, const int32_t *peer_rank,
                    const int32_t *send_offset, const int32_t *recv_offset, const int32_t *nstrips,
                    const int32_t *nstrips_recv /* NULL: as many come in as go out (same-level faces); adapted grids differ */);
/* One round of everything the time loop asks of the communicator, with known values and a deadline (default 20 s when
 * timeout_s <= 0): the plan's strips between every pair of peers (ncclSend / ncclRecv on the communication stream), an
 * all-gather and an all-reduce on the compute stream.  Collective; call it right after cup2d_comm_init.  Returns
 * CUP2D_ERR_COMM with the stage and the RCCL library in cup2d_last_error() when an operation does not complete in time
 * or delivers wrong values.  info (may be NULL) receives "rccl=<file> ranks=N rank=r peers=.. exchange_us=.. reduce_us=..".
 * Replaces nothing in the reference (its MPI calls are unchecked, main.cpp:2040-2047, cuda.cu:445-449). */
int cup2d_comm_selftest(cup2d_ctx *ctx, double timeout_s, char *info, int info_bytes);
int cup2d_comm_finalize(cup2d_ctx *ctx);
int cup2d_comm_stats(cup2d_ctx *ctx, int *nranks, int *npeers, long long *exchanges, long long *allreduces,
                     long long *allgathers);
int cup2d_halo_exchange(cup2d_ctx *ctx, int field, int width);

/* ---------------------------------------------------------------- instrumentation -------- */
/* HIP-event timing per kernel family, recorded on the context stream around the launches:
 * cup2d_set_timing(ctx, 1) every launch; (ctx, 2) sampled -- every 32nd BiCGSTAB iteration, and the launches outside the
 * solver in every 4th cup2d_step (an event pair is a barrier packet between two kernels, ~12 us of stream time: every launch
 * timed costs 10 % of a 4096^2 step, sampled < 1 %); (ctx, 3) as 2 with the launches outside the solver sampled in EVERY
 * cup2d_step; (ctx, 0) off.  Operators called on their own between two steps are always sampled.  cup2d_get_timing returns accumulated GPU
 * milliseconds and the number of timed launches since timing was enabled.  Events are resolved lazily
 * (no extra synchronisation per launch). */
typedef enum {
  CUP2D_T_ADVECT_STAGE = 0, /* fused WENO5 advect-diffuse RK stage (k_advect_diffuse) */
  CUP2D_T_POISSON_RHS = 1,  /* pressure_rhs + pressure_rhs1 fused */
  CUP2D_T_SWEEP_A = 2,      /* p update + block-Jacobi preconditioner */
  CUP2D_T_SWEEP_B = 3,      /* nu = A z + dot */
  CUP2D_T_SWEEP_C = 4,      /* r update + preconditioner */
  CUP2D_T_SWEEP_D = 5,      /* t = A z2 + 2 dots */
  CUP2D_T_SWEEP_E = 6,      /* x, r update + 3 reductions */
  CUP2D_T_SCALARS = 7,      /* reduction finish + device-side scalar algebra */
  CUP2D_T_PROJECT = 8,      /* mean removal + pressure-gradient update */
  CUP2D_T_REDUCE = 9,       /* max|u| (dt) */
  CUP2D_T_HALO = 10,        /* pack / unpack */
  CUP2D_T_INIT_RESIDUAL = 11, /* r = b - A x0 + its reductions (once per solve) */
  CUP2D_T_SMOOTHER = 12,    /* weighted-Jacobi sweep / Poisson residual (k_smoother) */
  CUP2D_T_SWEEP_EA = 13,    /* sweep E + the next iteration's sweeps A, B in one launch (CUP2D_FUSED_FORM=eab) */
  CUP2D_T_ADVECT_STAGE2 = 14, /* RK stage 2 of the fused WENO5 stage (stage 1 stays under CUP2D_T_ADVECT_STAGE when the two are told
                                 apart: cup2d_advect_diffuse_stage with its own old values, 48 B/cell against 32) */
  CUP2D_T_FINAL_X = 15,     /* the solve's last pass x = x0 + P_inv y (once per solve) */
  CUP2D_T_NTIMERS = 16
} cup2d_timer;
int cup2d_set_timing(cup2d_ctx *ctx, int enabled);
int cup2d_get_timing(cup2d_ctx *ctx, int timer, double *ms_total, int *calls);
/* Timing aid for the roofline of the fused WENO5 stage (bench.py "roofline_north_star"): knockout = 1 launches the quad kernel
 * without its arithmetic (the memory skeleton of its loop), 2 without the loads and stores inside its loop (the arithmetic
 * alone, on each wave's first quad), 0 the product.  While it is not 0 every fused stage launches the knocked-out kernel FIRST
 * -- under the stage's timer, on the stage's inputs, writing to a scratch slab -- and the product kernel behind it, untimed:
 * the step's results are unchanged, its stage timers hold the knocked-out launches.  Nothing in the library sets it; there is
 * no environment switch for it. */
int cup2d_debug_walk_knockout(cup2d_ctx *ctx, int knockout);

#ifdef __cplusplus
}
#endif
#endif /* CUP2D_HIP_H */
