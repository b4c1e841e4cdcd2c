// krylov_fused.hip -- tile-fused BiCGSTAB sweeps: the block-Jacobi preconditioner on the FP64 matrix
// cores, fused with the 5-point operator that consumes its result.
//
// Reference: BiCGSTABSolver::main cuda.cu:403-548 -- the same recurrences as krylov.hip, which runs them as
// five sweeps moving 184 B/cell per iteration.  A sweep pair (A,B) or (C,D) has no reduction between its two
// halves; it is split into two launches only because nu = A z needs the edge cells of z = P_inv p of the four
// NEIGHBOUR blocks.  P_inv is block-local and dense (64x64), i.e. 4096 FMA per block on the matrix cores,
// which idle on this bandwidth-bound path: so a wave RECOMPUTES the z edges it needs from the neighbours'
// p instead of reading z back from HBM.  One wave owns a tile of 16 consecutive blocks (the N of a
// v_mfma_f64_16x16x4_f64; in the reference's Hilbert order an aligned run of 16 blocks is a 4x4 patch):
//
//   ring     the (block, side) pairs of the tile whose neighbour lies outside it are collected (ballot +
//            prefix popcount); for every 16 of them: load the neighbour blocks, form v (p_new or s), stage
//            it in LDS, Z = V P_inv on the MFMA pipe, keep the 8 edge cells the tile touches
//   tile     the same for the 16 blocks of the tile; v is also written out (p or s)
//   stencil  y = A z from the LDS tile + edges, the dot products of sweeps B / D fused in
//
// z and z2 are never stored.  The x update moves into the preconditioned space -- y += alpha p + omega s,
// x = x0 + P_inv y once at the end (P_inv is linear, z = P_inv p, z2 = P_inv s) -- so sweep E reads p
// instead of z and z2.  Per iteration and cell:
//   AB  reads p, nu, r, rhat (32) writes p', nu' (16)           = 48 B   (+ ring re-reads, served by L2)
//   CD  reads r, nu' (16)         writes t (8)                  = 24 B   (s = r - alpha nu' is not stored)
//   E   reads y, p', r, nu', t, rhat (48) writes y, r (16)      = 64 B        total 136 B (five sweeps: 184 B)
// p and nu are double-buffered (a ring recomputation must see the OLD p, nu of a block another wave may
// already have advanced) and s gets its own vector for the same reason.
#include <stdlib.h>
#include <string.h>

#include "block.h"
#include "krylov_common.h"
#include "precond_mfma.h"

namespace cup2d {

constexpr int TB = FUSED_TILE;   // blocks per tile
constexpr int XS = 66;   // LDS stride of one block in the staging tile: the A-operand reads of a half-wave
                         // (block = lane%16, k = 4ks + lane/16) then fall on 32 distinct 8-byte banks
// Waves per workgroup (ONE workgroup per CU).  8: two waves per SIMD, 256 registers each.  4: one wave per SIMD with the
// whole 512-entry register file -- room for deeper load schedules without spills, and half as many tiles in flight per
// XCD (their footprint then fits the 4 MiB L2 the ring re-reads have to hit).  CUP2D_FUSED_PREG 1: P_inv fragments in
// registers (128 VGPRs; sensible only with 4 waves) instead of 32 KiB of LDS.
#ifndef CUP2D_FUSED_WAVES
#define CUP2D_FUSED_WAVES 8
#endif
#ifndef CUP2D_FUSED_PREG
#define CUP2D_FUSED_PREG 0
#endif
constexpr int FWAVES = CUP2D_FUSED_WAVES;
constexpr int FWG = 64 * FWAVES;  // threads per workgroup of the fused sweeps
constexpr bool PREG = CUP2D_FUSED_PREG != 0;
// Cache policy of the individual streams of an iteration (bit set = non-temporal access).  An iteration moves ~20
// vectors of 134 MB (4096^2) through a 256 MB memory-side cache and 8 x 4 MiB of L2: left alone, everything evicts
// everything.  Policy: a stream that is consumed once, or whose consumer is a whole iteration away, bypasses
// (p', t, y' stores; y, p, t, rhat loads of sweep E; rhat of sweep AB); the streams the NEXT launch consumes first
// allocate (nu' for CD, s for E, r for AB) -- and so do the tile and ring loads of AB / CD, whose second touch by the
// neighbouring tile must find them in L2.  Measured at 4096^2 (tools/build_policy_variants.sh, tools/gpu_calls/gpu_call6.sh;
// A / C / E in us, step in ms):
//   0x000 nothing non-temporal             200 / 153 / 187   27.9
//   0x00F stores of AB and CD              201 / 153 / 175   27.4    (the previous default)
//   0xED9 this policy                      202 / 153 / 147   26.1
//   0xED9 | 0x4000 (all AB loads)          224 / 160 / 156   27.9    ring re-reads miss L2
//   0xED9 & ~0x208 (keep t as well as s)   202 / 152 / 169   27.1    two kept vectors overflow the cache
//   0xED9 | 0x020 (r store)                198 / 154 / 158   26.3
//   0xED9 | 0x20000 (ring loads)           +2 % / -2 % / =   (same box as a 186 / 144 / 144 run of 0xED9: noise)
// Sweep E (pure streaming, 56 B/cell) gains 16 %: s comes out of the cache; AB and CD do not react -- they are bound
// by their own load -> LDS -> MFMA pipeline, not by what the memory side delivers.
//   0x001 AB store p'   0x002 AB store nu'   0x004 CD store s   0x008 CD store t
//   0x010 E store y'    0x020 E store r      0x040 E load y     0x080 E load p    0x100 E load s   0x200 E load t
//   0x400 E load rhat   0x800 AB load rhat   0x1000 AB tile load p   0x2000 AB tile load nu
//   0x4000 AB every load of p, nu, r (tile and ring)   0x8000 CD loads of r   0x10000 CD loads of nu'
//   0x20000 ring loads of AB and CD
#ifndef CUP2D_POLICY
#define CUP2D_POLICY 0xED9
#endif
constexpr unsigned POL = CUP2D_POLICY;
// Load schedule of a job's two half-batches (8 blocks each).  0: the second half is requested when the first is being
// staged -- it has one staging (< 1 us) of lead and the wave then waits out the memory latency once per job.  1: both
// halves of the NEXT job are requested as soon as both halves of this one are staged, i.e. before this job's MFMA,
// epilogue and stencil; same registers.
// Bit MODE of CUP2D_FUSED_DEEP selects it per sweep.  Measured at 4096^2: CD 153 -> 149 us; AB has the rhat operand in
// registers at the same time and spills under it (204 -> 467 us), so AB keeps schedule 0.
#ifndef CUP2D_FUSED_DEEP
#define CUP2D_FUSED_DEEP 2
#endif
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool NT>
static __device__ __forceinline__ double2 ld2(const double2 *p) {
  if (NT) {
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
    return make_double2(v.x, v.y);
  }
  return *p;
}
template <bool NT>
static __device__ __forceinline__ void st2(double2 *p, double2 v) {
  if (NT) {
    v2d w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<v2d *>(p));
  } else {
    *p = v;
  }
}  // 
constexpr int PL_DOUBLES = 16 * 4 * 64;  // P_inv as MFMA B fragments: [k-step][n-tile][lane]

constexpr int GS = 10;   // doubles per (block, side) slot of GE: 8 used; with stride 8 the 64 lanes that each fill one slot hit
                         // 4 banks (8-way conflicts, 7 M conflict cycles per launch); 10 is even (the stencil reads pairs with
                         // ds_read_b128) and leaves 2-way
struct alignas(16) FusedLds {
  double S[TB * XS];       // v of 16 blocks, then (same storage) z of those blocks
  double GE[TB * 4 * GS];  // z on the ghost edges of the tile's blocks: [block][W,E,S,N][position], slot stride GS
  int ring_nb[TB * 4];     // neighbour block of ring entry e ...
  int ring_dst[TB * 4];    // ... and the slot (block * 4 + side) it feeds
};
// The RING job needs one edge (8 cells) of z per entry, not the block: it multiplies with the 32 columns of P_inv that
// produce edge cells -- [W | E] and [S | N], two 16-column tiles instead of four -- and is half a job.  The FP64 matrix
// core is what bounds these sweeps (64 cycles per v_mfma_f64_16x16x4_f64); same k order, same numbers.
#ifndef CUP2D_FUSED_EDGEOP
#define CUP2D_FUSED_EDGEOP 1
#endif
constexpr bool EDGEOP = CUP2D_FUSED_EDGEOP != 0 && !PREG;
constexpr int PE_DOUBLES = EDGEOP ? 16 * 2 * 64 : 0;  // the edge columns as B fragments: [k-step][n-tile][lane]
constexpr int PL_LDS_DOUBLES = (PREG ? 0 : PL_DOUBLES) + PE_DOUBLES;
constexpr size_t FUSED_LDS_BYTES = PL_LDS_DOUBLES * sizeof(double) + FWAVES * sizeof(FusedLds);

// cell (iy*8+ix) at position q of the edge on side s (W, E, S, N) of a block
static __device__ __forceinline__ int edge_cell(int s, int q) {
  return s == 0 ? q * BS : s == 1 ? q * BS + (BS - 1) : s == 2 ? q : (BS - 1) * BS + q;
}

// S (v, block-major) -> Z = V P_inv -> S (z, block-major) on v_mfma_f64_16x16x4_f64 with the fragment maps
// of precond_mfma.h.  The B operands (P_inv) come from LDS, one conflict-free ds_read_b64 per MFMA: in
// registers they cost every wave 128 VGPRs -- the first version of this kernel had nothing left to keep
// loads in flight with (2 waves per SIMD, 234 VGPRs) and ran latency-bound at 3.3 TB/s.  Same k order and
// operands as precond_tile, so z is bit-identical to the unfused MFMA preconditioner.
// CUP2D_FUSED_MFMA_PRIO: the two waves of a SIMD share its matrix core; per-phase clocks (-DFUSED_PHASES) show 5.2-6.5 k
// cycles per 64-MFMA job, and half of a CD tile's time goes into its two jobs.  With the priority raised for the duration of
// a job the first wave to arrive would finish at the full rate and the two waves drift apart: one wave's job beside the
// other's loads and staging.
// Measured at 4096^2: AB 162.4 -> 163.3 us, CD 121.6 -> 123.8 us -- nothing: the matrix core itself is the limit (a
// v_mfma_f64_16x16x4_f64 takes 64 cycles on this part, tools/fp64_peak.hip; two waves x two jobs x 64 MFMAs = 16 k cycles of
// a 23.5 k-cycle CD tile).  Off.
#ifndef CUP2D_FUSED_MFMA_PRIO
#define CUP2D_FUSED_MFMA_PRIO 0
#endif
template <bool DB = false>
static __device__ __forceinline__ void tile_precond(double *S, const double *PL, const PinvFragments &PR, int lane, bool skip) {
  const int ablk = lane & 15, akk = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; nt++) acc[nt] = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (!skip) {
    double xa[16];
#pragma unroll
    for (int ks = 0; ks < 16; ks++) xa[ks] = S[ablk * XS + 4 * ks + akk];
#if CUP2D_FUSED_MFMA_PRIO
    __builtin_amdgcn_s_setprio(1);  // the job of the wave that gets here first runs at the full MFMA rate (below)
#endif
    if constexpr (PREG || !DB) {
#pragma unroll
      for (int ks = 0; ks < 16; ks++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], PREG ? PR.b[ks][nt] : PL[(ks * 4 + nt) * 64 + lane], acc[nt], 0, 0, 0);
    } else {
      // The B fragments of k-step ks + 1 are requested BEFORE the four MFMAs of k-step ks (two register sets, the order
      // pinned with scheduling barriers); left to the scheduler every k-step reads its fragments right in front of its
      // MFMAs.  No gain (CUP2D_FUSED_DB above): the four MFMAs of a k-step take 256 cycles, the read is hidden either way.
      double bf[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; nt++) bf[0][nt] = PL[nt * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < 16; ks++) {
        if (ks + 1 < 16) {
#pragma unroll
          for (int nt = 0; nt < 4; nt++) bf[(ks + 1) & 1][nt] = PL[((ks + 1) * 4 + nt) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], bf[ks & 1][nt], acc[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if CUP2D_FUSED_MFMA_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  }
  wave_lds_sync();  // every lane has read its operands before the tile is overwritten
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) S[(akk + 4 * v) * XS + 16 * nt + ablk] = acc[nt][v];
  wave_lds_sync();
}

// the ring job's product: S (v of 16 ring entries, block-major) -> S[e][8 side + q] = z of entry e on the edge cells of side
// W, E, S, N.  PE = the 32 edge columns of P_inv as B fragments.
static __device__ __forceinline__ void ring_precond(double *S, const double *PE, int lane, bool skip) {
  double xa[16];
  const int ablk = lane & 15, akk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 16; ks++) xa[ks] = S[ablk * XS + 4 * ks + akk];
  v4f64 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; nt++) acc[nt] = (v4f64){0.0, 0.0, 0.0, 0.0};
  if (!skip) {
#pragma unroll
    for (int ks = 0; ks < 16; ks++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], PE[(ks * 2 + nt) * 64 + lane], acc[nt], 0, 0, 0);
  }
  wave_lds_sync();  // every lane has read its operands before the tile is overwritten
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) S[(akk + 4 * v) * XS + 16 * nt + ablk] = acc[nt][v];
  wave_lds_sync();
}

// -DFUSED_PHASES (timing aid, tools/gpu_fused_phases.sh): every wave adds up the shader-clock cycles it spends in the
// phases of a tile; waves 0 of the first workgroups print their sums for iteration 5
#ifdef FUSED_PHASES
#define PH(k) { const long long now_ = clock64(); ph[k] += now_ - tprev; tprev = now_; }
#else
#define PH(k) {}
#endif
struct FusedArgs {
  const double *in0, *in1, *in2;  // AB: p, nu, r     CD: r, nu, -
  double *w;                      // AB: rhat (written on a restart)
  double *vout, *yout;            // AB: p', nu'      CD: s, t
  // k_edge MODE 2 (sweep E + the next A+B): t, the three buffers of the accumulated correction (k_sweepE_y), where r' goes,
  // and the host's status word when this launch is the one of its group that reports
  const double *t;
  double *y0, *y1, *y2, *rout;
  int *host_status;
  int rev;  // k_edge: the tiles in descending order (krylov_edge.h "Direction")
  // k_edge MERGE 3 (N ranks, deferred scalar update): the records of the previous reduction point gathered from all ranks
  // ([pn][RED_REC]), the stage they belong to (-1: none pending), and where the updated state goes (the kernel's sc is read only)
  const double *pg;
  int pn, pstage, pnsum, pmax;
  KrylovScalars *sc_out;
  // k_edge HYB (MODE 2 / 3 on the hybrid assembled operator, one rank): where z of the blocks the stored rows read goes
  // (k_hyb_rows takes it from there), which blocks of a tile those are, the tiling (ctx.h SellMatrix d_zmask, d_tile0)
  double *zg;
  const int32_t *zmask, *tile0;
};

// one partial per workgroup and slot for a workgroup of NW waves (block.h's version is for WPG waves)
template <int NW, int N, bool COHERENT>
static __device__ __forceinline__ void fused_reduce_store(double (&v)[N], double *partials) {
  __shared__ double red[N][NW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double w = wave_sum(v[i]);
    if (lane == 0) red[i][wave] = w;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double a = red[threadIdx.x][0];
    for (int k = 1; k < NW; k++) a += red[threadIdx.x][k];
    double *dst = partials + (size_t)threadIdx.x * PSTRIDE + blockIdx.x;
    if (COHERENT) __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = a;
  }
}

// the maximum of one value per lane over a workgroup of NW waves, to this workgroup's partial
template <int NW, bool COHERENT>
static __device__ __forceinline__ void fused_reduce_store_max(double v, double *partials) {
  __shared__ double redm[NW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double w = wave_max(v);
  if (lane == 0) redm[wave] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = redm[0];
    for (int k = 1; k < NW; k++) a = fmax(a, redm[k]);
    double *dst = partials + blockIdx.x;
    if (COHERENT) __hip_atomic_store(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *dst = a;
  }
}

// MODE 0 (sweeps A+B): v = p' = beta (p - omega nu) + r   (cuda.cu:478-483; restart: p' = rhat = r, 461-476)
//                      y = nu' = A P_inv p' ; partial(rhat . nu')                       (484-488)
// MODE 1 (sweeps C+D): v = s  = r - alpha nu'                                            (499-502)
//                      y = t  = A P_inv s  ; partial(t . s, t . t)                      (503-509)
// A tile is a chain of JOBS of 16 blocks each -- ceil(nring/16) ring passes, then the tile itself -- and a job
// is two batches of 8 blocks.  The global loads of a batch are issued one batch ahead of their use, across
// job boundaries, so a wave always has 8 blocks x {3|2} vectors in flight while it stages, multiplies,
// gathers; the w operand of the dot product (rhat) is requested before the tile's MFMA.
//
// Every global access is 16 bytes per lane: a wave instruction moves TWO blocks (lanes 0-31 the cell pairs of one,
// lanes 32-63 those of the next), staging writes and the stencil's reads are ds_*_b128 on the same pairs.  With 8-byte
// accesses (one cell per lane) both sweeps sat at the rate the vector memory path sustains for 8-byte requests
// (5.4-6.1 TB/s of L2-side traffic, ring re-reads included; SQ_WAIT_INST_ANY 44 %: issue stalled behind a full memory
// pipeline) although HBM had room; 16-byte requests halve the instructions per byte.
// HYB: the hybrid assembled operator (ctx.h SellMatrix): nbr = its d_fnbr.  A tile whose slots read FUSED_GENERAL holds a
// slice with stored rows: the wave forms v and z = P_inv v of its blocks and stores them (v as always, z to zg), the rows
// themselves are applied by k_hyb_rows from z in memory; a fused tile also stores the z of the blocks zmask names (the
// ones those rows read).  Ring entries work as ever: the neighbour of a plain block is a block with p, nu, r in memory.
template <int MODE, int MERGE, bool HYB = false>
__global__ __launch_bounds__(FWG, 1) void k_fused(FusedArgs A, const double *__restrict__ Pinv,
                                                  const int *__restrict__ nbr, KrylovScalars *sc, double *partials,
                                                  int first, int count, int poff, int nowned,
                                                  double *__restrict__ zg, double *red, unsigned *ticket,
                                                  int dbg, const int32_t *__restrict__ zmask, const int32_t *__restrict__ tile0) {
  // blocks [first, first + count); neighbour ids < nowned name blocks whose p, nu, r are in memory (owned blocks, and on N
  // ranks the ghost blocks: solve_fused_impl passes ntotal)
  extern __shared__ __attribute__((aligned(16))) double fsm[];
  if (sc->status != 0) return;
  constexpr bool DEEP = ((CUP2D_FUSED_DEEP >> MODE) & 1) != 0;
  // double-buffered B fragments in the MFMA jobs (tile_precond): bit MODE of CUP2D_FUSED_DB.  Measured at 4096^2: CD
  // 121.0 -> 122.8 / 120.3 us, AB (which spills under the eight registers more) 163 -> 168 us: the 64-MFMA job is not
  // waiting for its fragments.  Off.
#ifndef CUP2D_FUSED_DB
#define CUP2D_FUSED_DB 0
#endif
  constexpr bool PRECOND_DB = ((CUP2D_FUSED_DB >> MODE) & 1) != 0;
  double *PL = fsm;
  PinvFragments PR;
  if constexpr (PREG) {
    PR.load(Pinv, threadIdx.x & 63);
  } else {
    for (int idx = threadIdx.x; idx < PL_DOUBLES; idx += FWG) {
      const int l = idx & 63, nt = (idx >> 6) & 3, ks = idx >> 8;
      PL[idx] = Pinv[(4 * ks + (l >> 4)) * BC + 16 * nt + (l & 15)];
    }
    if constexpr (EDGEOP) {
      for (int idx = threadIdx.x; idx < PE_DOUBLES; idx += FWG) {
        const int l = idx & 63, nt = (idx >> 6) & 1, ks = idx >> 7, col = 16 * nt + (l & 15);
        PL[PL_DOUBLES + idx] = Pinv[(4 * ks + (l >> 4)) * BC + edge_cell(col >> 3, col & 7)];
      }
    }
    __syncthreads();
  }
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  FusedLds &L = reinterpret_cast<FusedLds *>(fsm + PL_LDS_DOUBLES)[wave];
  // pair layout of the 16-byte accesses: this lane holds cells c0 = 2 hl, c0 + 1 of block 2 i + hf
  const int hf = lane >> 5, hl = lane & 31, c0 = 2 * hl, px = c0 & 7, py = hl >> 2;
  const double c1 = MODE == 0 ? -sc->omega : -sc->alpha;  // momega | malpha
  const double beta = sc->beta;
  // restart: p' = rhat = r (cuda.cu:461-476).  The first iteration of a solve has p = nu = 0 (cuda.cu:436-437), so
  // p' = beta (0 - omega 0) + r = r as well: the two vectors are not read as zeros, they are not looked at -- the solve
  // does not fill them
  const bool fresh = MODE == 0 && sc->iter == 0;
  const bool restart = MODE == 0 && sc->restart_flag != 0;
  constexpr int NDOT = MODE == 0 ? 1 : 2;
  double acc[NDOT];
#pragma unroll
  for (int i = 0; i < NDOT; i++) acc[i] = 0.0;

  // v at one cell -- the arithmetic of k_sweepA_fd / k_sweepC_fd, operation for operation
  const auto form_v = [&](double a, double b, double c) -> double {
    if (MODE == 0) {
      if (restart || fresh) return c;
      double v = a + c1 * b;
      v = v * beta;
      return v + c;
    }
    return a + c1 * b;
  };
  struct Raw {  // one batch = 8 blocks = 4 block pairs
    double2 a[4], b[4], c[4];
  };

  // tiles of 16 blocks over the waves of the persistent grid, contiguous per XCD (workgroup w runs on XCD w % 8)
  const int ntiles = HYB ? count : (count + TB - 1) / TB;  // HYB: count = number of tiles of the table tile0
  int t_begin, t_end, t_stride;
  {
    const int G = gridDim.x, w = blockIdx.x;
    if (G >= 8 && (G % 8) == 0) {
      const int xcd = w & 7, slot = w >> 3, per = G >> 3;
      const long long lo = (long long)ntiles * xcd / 8, hi = (long long)ntiles * (xcd + 1) / 8;
      t_begin = (int)lo + slot * FWAVES + wave;
      t_end = (int)hi;
      t_stride = per * FWAVES;
    } else {
      t_begin = w * FWAVES + wave;
      t_end = ntiles;
      t_stride = G * FWAVES;
    }
  }
  const int si = lane >> 2, ss = lane & 3;  // this lane's (block, side) slot of a tile
  const int last = first + count;
  const auto load_nb = [&](int t) -> int {
    if constexpr (HYB) {
      if (t >= t_end) return CUP2D_WALL;
      const int b0 = uniform(tile0[t]), nv = uniform(tile0[t + 1]) - b0;
      return si < nv ? nbr[4 * (b0 + si) + ss] : CUP2D_WALL;
    }
    const int b = first + t * TB + si;
    return (t < t_end && b < last) ? nbr[4 * b + ss] : CUP2D_WALL;
  };
  // state of a tile: its blocks, this lane's neighbour slot, the ring list (in LDS) it was classified into
  struct Tile {
    int b0, nvalid, nb, nring, npass, zm;
    bool is_ring, gen;
  };
  // classify the 64 (block, side) neighbour slots of tile t and write its ring list (overwrites the list
  // of the previous tile: call only when that one is dead)
  const auto classify = [&](int t, int nb) -> Tile {
    Tile T;
    T.b0 = HYB ? uniform(tile0[t]) : first + t * TB;
    T.nvalid = HYB ? uniform(tile0[t + 1]) - T.b0 : min(TB, last - T.b0);
    T.nb = nb;
    T.is_ring = si < T.nvalid && nb >= 0 && nb < nowned && (nb < T.b0 || nb >= T.b0 + T.nvalid);
    const unsigned long long rmask = __ballot(T.is_ring);
    T.nring = (dbg & 1) ? 0 : __popcll(rmask);  // dbg 1: timing experiment without the ring -- WRONG results
    T.npass = (T.nring + TB - 1) / TB;
    T.gen = HYB && __ballot(si < T.nvalid && nb == FUSED_GENERAL) != 0ull;
    T.zm = HYB ? uniform(zmask[t]) : 0;
    if (T.is_ring) {
      const int slot = __popcll(rmask & ((1ull << lane) - 1ull));
      L.ring_nb[slot] = nb;
      L.ring_dst[slot] = lane;
    }
    wave_lds_sync();
    return T;
  };
  // the 4 block pairs of batch `half` of job j of tile T -- ring entries, or for the last job blocks of the tile --
  // branch-free (the ring list is read even when it is not used), so that the 4 x {3|2} loads issue back to back
  const auto issue = [&](Raw &R, const Tile &T, int j, int half) {
    const bool tile_job = j >= T.npass;
    const int ne = max(1, min(TB, T.nring - j * TB));
    size_t off[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int idx = 8 * half + 2 * p + hf;  // per half-wave
      const int rb = L.ring_nb[min(j * TB + min(idx, ne - 1), TB * 4 - 1)];
      const int blk = tile_job ? T.b0 + min(idx, T.nvalid - 1) : rb;
      off[p] = ((size_t)blk * BC + c0) >> 1;  // in double2 units
    }
    const double2 *in0 = reinterpret_cast<const double2 *>(A.in0), *in1 = reinterpret_cast<const double2 *>(A.in1);
    const double2 *in2 = reinterpret_cast<const double2 *>(A.in2);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const size_t o = off[p];
      if (!tile_job && (POL & 0x20000)) {  // ring loads only
        R.a[p] = ld2<true>(in0 + o);
        R.b[p] = ld2<true>(in1 + o);
        if (MODE == 0) R.c[p] = ld2<true>(in2 + o);
      } else if (MODE == 0 && (POL & 0x4000)) {
        R.a[p] = ld2<true>(in0 + o);
        R.b[p] = ld2<true>(in1 + o);
        R.c[p] = ld2<true>(in2 + o);
      } else if (MODE == 1 && (POL & 0x18000)) {
        R.a[p] = ld2<(POL & 0x8000) != 0>(in0 + o);
        R.b[p] = ld2<(POL & 0x10000) != 0>(in1 + o);
      } else if (MODE == 0 && tile_job && (POL & 0x3000)) {
        R.a[p] = ld2<(POL & 0x1000) != 0>(in0 + o);
        R.b[p] = ld2<(POL & 0x2000) != 0>(in1 + o);
        R.c[p] = in2[o];
      } else {
        R.a[p] = in0[o];
        R.b[p] = in1[o];
        if (MODE == 0) R.c[p] = in2[o];
      }
    }
  };

  Raw Ra, Rb;
  Tile T;
  if (t_begin < t_end) {
    T = classify(t_begin, load_nb(t_begin));
    issue(Ra, T, 0, 0);
    if (DEEP) issue(Rb, T, 0, 1);
  }
  int nb_next = load_nb(t_begin + t_stride);
#ifdef FUSED_PHASES
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
  int ntile = 0;
#endif
  for (int t = t_begin; t < t_end; t += t_stride) {
    // invariant: T describes tile t, its ring list is in LDS, the first batch of its first job is in flight in Ra
    const int b0 = T.b0, nvalid = T.nvalid;
#ifdef FUSED_PHASES
    ntile++;
#endif
    if (dbg & 1) {
#pragma unroll
      for (int q = 0; q < BS; q++) L.GE[lane * GS + q] = 0.0;
    }
    // dot-product operand of the tile's cells in pair layout: s itself (CD); rhat (AB; r on a restart)
    double2 W[TB / 2];
    const auto stage = [&](const Raw &R, bool is_tile, int half) {
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int idx = 8 * half + 2 * p + hf;
        double2 v;
        v.x = form_v(R.a[p].x, R.b[p].x, MODE == 0 ? R.c[p].x : 0.0);
        v.y = form_v(R.a[p].y, R.b[p].y, MODE == 0 ? R.c[p].y : 0.0);
        *reinterpret_cast<double2 *>(L.S + idx * XS + c0) = v;
        if (is_tile) {
          if (MODE == 1) W[4 * half + p] = v;
          if (MODE == 0 && restart) W[4 * half + p] = R.c[p];
          if (MODE == 0 && idx < nvalid) {  // CD does not store s: its only reader, sweep E, forms it again from r and nu'
            const size_t o = ((size_t)(b0 + idx) * BC + c0) >> 1;
            st2<(POL & 0x001) != 0>(reinterpret_cast<double2 *>(A.vout) + o, v);
            if (restart) reinterpret_cast<double2 *>(A.w)[o] = R.c[p];  // rhat = r
          }
        }
      }
    };
    Tile N = T;  // the next tile, once classified
    for (int j = 0; j <= T.npass; j++) {
      const bool is_tile = j == T.npass;
      if (!DEEP) issue(Rb, T, j, 1);
      stage(Ra, is_tile, 0);
      if (DEEP) stage(Rb, is_tile, 1);
      if (!is_tile) {
        issue(Ra, T, j + 1, 0);
        if (DEEP) issue(Rb, T, j + 1, 1);
      } else if (MODE == 0 && !restart) {
#pragma unroll
        for (int i = 0; i < TB / 2; i++) {
          const double2 *pw = reinterpret_cast<const double2 *>(A.w) + (((size_t)(b0 + min(2 * i + hf, nvalid - 1)) * BC + c0) >> 1);
          W[i] = ld2<(POL & 0x800) != 0>(pw);
        }
      }
      if (!DEEP) stage(Rb, is_tile, 1);
      if (is_tile) PH(2) else PH(0)
      if (is_tile && t + t_stride < t_end) {
        // this tile's ring list is dead: classify the NEXT tile into it and put its first job in flight
        // behind this tile's MFMA, edge fill and stencil
        N = classify(t + t_stride, nb_next);
        nb_next = load_nb(t + 2 * t_stride);
        issue(Ra, N, 0, 0);
        if (DEEP) issue(Rb, N, 0, 1);
      } else {
        wave_lds_sync();
      }
      if (is_tile) PH(3)
      if (EDGEOP && !is_tile) ring_precond(L.S, PL + PL_DOUBLES, lane, (dbg & 2) != 0);
      else tile_precond<PRECOND_DB>(L.S, PL, PR, lane, (dbg & (is_tile ? 4 : 2)) != 0);
      if (is_tile) PH(4)
      if constexpr (HYB) {
        if (is_tile && T.zm != 0) {  // z of the blocks somebody reads from memory
#pragma unroll
          for (int i = 0; i < TB / 2; i++) {
            const int blk = 2 * i + hf;
            if (blk < nvalid && ((T.zm >> blk) & 1))
              reinterpret_cast<double2 *>(zg)[((size_t)(b0 + blk) * BC + c0) >> 1] = *reinterpret_cast<const double2 *>(L.S + blk * XS + c0);
          }
        }
      }
      if (!is_tile) {
        // entry e feeds slot dst = block*4 + side with the OPPOSITE edge of the neighbour block
        const int ne = min(TB, T.nring - j * TB);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int idx = lane + 64 * h, e = idx >> 3, q = idx & 7;
          if (e < ne) {
            const int dst = L.ring_dst[j * TB + e];
            L.GE[dst * GS + q] = L.S[e * XS + (EDGEOP ? 8 * ((dst & 3) ^ 1) + q : edge_cell((dst & 3) ^ 1, q))];
          }
        }
        wave_lds_sync();
        PH(1)
      }
    }
    // ---- edges inside the tile and at domain walls (ScalarLab::Neumann2D, main.cpp:3210-3255: ghost =
    //      edge cell), from the z tile: this lane's (block, side) slot ----
    if (HYB && T.gen) {  // rows with stored entries in this tile: k_hyb_rows applies them
      wave_lds_sync();
      T = N;
      continue;
    }
    if (si < nvalid && !T.is_ring) {
      const int sblk = T.nb < 0 ? si : T.nb - b0, sside = T.nb < 0 ? ss : ss ^ 1;
#pragma unroll
      for (int q = 0; q < BS; q++) L.GE[lane * GS + q] = L.S[sblk * XS + edge_cell(sside, q)];
    }
    wave_lds_sync();
    PH(5)
    // ---- y = A z (operand order of k_sweepBD / pressure_rhs1 main.cpp:6228) + the fused dot products: two cells per
    //      lane, eight block pairs ----
#pragma unroll
    for (int i = 0; i < TB / 2; i++) {
      const int blk = 2 * i + hf;
      if (blk < nvalid) {
        const double *zb = L.S + blk * XS + c0;
        const double *ge = L.GE + blk * 4 * GS;
        const double2 zc = *reinterpret_cast<const double2 *>(zb);
        const double lw = *(px > 0 ? zb - 1 : ge + 0 * GS + py);           // west of cell c0
        const double le = *(px < BS - 2 ? zb + 2 : ge + 1 * GS + py);      // east of cell c0 + 1
        const double2 ls = *reinterpret_cast<const double2 *>(py > 0 ? zb - BS : ge + 2 * GS + px);
        const double2 ln = *reinterpret_cast<const double2 *>(py < BS - 1 ? zb + BS : ge + 3 * GS + px);
        double2 yv;
        yv.x = lw + zc.y + ls.x + ln.x - 4 * zc.x;
        yv.y = zc.x + le + ls.y + ln.y - 4 * zc.y;
        st2<(POL & (MODE == 0 ? 0x002 : 0x008)) != 0>(reinterpret_cast<double2 *>(A.yout) + (((size_t)(b0 + blk) * BC + c0) >> 1), yv);
        acc[0] = __builtin_fma(yv.x, W[i].x, acc[0]);
        acc[0] = __builtin_fma(yv.y, W[i].y, acc[0]);
        if constexpr (NDOT == 2) {
          acc[1] = __builtin_fma(yv.x, yv.x, acc[1]);
          acc[1] = __builtin_fma(yv.y, yv.y, acc[1]);
        }
      }
    }
    wave_lds_sync();  // the next tile overwrites S and GE
    PH(6)
    T = N;
  }
#ifdef FUSED_PHASES
  if (sc->iter == 5 && lane == 0 && blockIdx.x < 2 && (wave == 0 || wave == 5) && ntile > 0)
    printf("PHASES mode %d wg %d wave %d tiles %d cycles/tile: ring stage+wait %lld  ring mfma+edges %lld  tile stage+wait %lld  classify+issue %lld  "
           "tile mfma %lld  edges %lld  stencil+stores(+2nd job) %lld\n", MODE, (int)blockIdx.x, wave, ntile, ph[0] / ntile, ph[1] / ntile,
           ph[2] / ntile, ph[3] / ntile, ph[4] / ntile, ph[5] / ntile, ph[6] / ntile);
#endif
  // MERGE 1: the last workgroup finishes the reduction and runs the scalar update (one GPU).  MERGE 2: it only sums
  // this rank's partials -- those of an earlier launch of the same sweep included, poff of them -- into red; the
  // all-reduce and the scalar update follow on the stream (N GPUs).
  fused_reduce_store<FWAVES, NDOT, MERGE != 0>(acc, partials + poff);
  if (MERGE && arrive_last(ticket))  // (arrive_last's barrier: every wave of this workgroup is done with the dynamic LDS)
    finish_reduce<true, true>(partials, poff + (int)gridDim.x, NDOT, 0, red, sc, MERGE == 1 ? MODE + 1 : -1, nullptr, fsm);
}

#include "krylov_edge.h"

// ---- the rows of the general tiles of the hybrid operator (k_fused HYB) -----------------------------
// y = A z for the listed blocks, z from memory (k_fused stored it for exactly the blocks these rows read; the halo entries
// of other ranks arrive in between), with the dot products of the sweep:  MODE 0: y = nu', w = rhat;  MODE 1: y = t,
// w = s = r - alpha nu' formed again.  One wave per block, lane = row (krylov_common.h sell_row).  The partials go behind
// the poff partials of k_fused; MERGE as there: the last workgroup finishes the reduction of BOTH launches.
// MODE 2 / 3: behind k_edge HYB (the two-launch organisation on the hybrid operator).  MODE 2: y = nu'', w = rhat, into the
// slots {rhat.nu'', r'.r', max|r'|} of that launch (the last two are complete there: zeros here), finish = stage 4 and the
// report to the host.  MODE 3: y = t, w = s, w2 = rhat, into {t.s, t.t, rhat.s, rhat.t, s.s} (rhat.s and s.s complete there),
// finish = stage 5.
constexpr int RWAVES = 16, RWG = RWAVES * 64;  // one wave per block; wide workgroups: few tickets (arrive_last), many waves
template <int MODE, int MERGE>
__global__ __launch_bounds__(RWG) void k_hyb_rows(const double *__restrict__ z, double *__restrict__ y,
                                                 const double *__restrict__ w0, const double *__restrict__ w1,
                                                 const RowsRec *__restrict__ rrec, const int32_t *__restrict__ col,
                                                 const double *__restrict__ val, int nlist, KrylovScalars *sc,
                                                 double *partials, int poff, double *red, unsigned *ticket,
                                                 const double *__restrict__ w2, int *host_status) {
  // The launch is a chain of memory round trips, not work (3.6 blocks per wave; 12.5 us with half the list, api.hip): the
  // record of the wave's first block is requested BEFORE the look at the solve's status, and one record holds what the list,
  // d_ptr and d_reg held in three dependent steps
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int e0 = blockIdx.x * RWAVES + wave;
  RowsRec R = rrec[e0 < nlist ? e0 : 0];
  if (sc->status != 0) {
    // (as k_sweepE_y: the last launch of a group of iterations reports to the host, also behind a solve that has ended)
    if (MODE == 2 && MERGE == 1 && host_status && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  constexpr int NDOT = MODE == 0 ? 1 : MODE == 1 ? 2 : MODE == 2 ? 3 : 5;  // partial slots of the sweep in front
  constexpr bool WS = MODE == 1 || MODE == 3;                             // w = s = r - alpha nu', formed again
  const double c1 = -sc->alpha;
  double acc[NDOT];
#pragma unroll
  for (int i = 0; i < NDOT; i++) acc[i] = 0.0;
  for (int e = e0; e < nlist; e += gridDim.x * RWAVES) {
    if (e != e0) R = rrec[e];
    const int s = uniform(R.s);
    const size_t o = (size_t)s * BC + lane;
    // (the operands of the dot products do not depend on the row: requested beside its entries)
    const double wa = w0[o], wb = WS ? w1[o] : 0.0, wc = MODE == 3 ? w2[o] : 0.0;
    const double a = sell_row_at(z, s, lane, R.base, uniform(R.width), make_int4(R.reg[0], R.reg[1], R.reg[2], R.reg[3]), col, val);
    y[o] = a;
    const double w = WS ? wa + c1 * wb : wa;
    acc[0] = __builtin_fma(a, w, acc[0]);
    if constexpr (WS) acc[1] = __builtin_fma(a, a, acc[1]);
    if constexpr (MODE == 3) acc[3] = __builtin_fma(wc, a, acc[3]);
  }
  fused_reduce_store<RWAVES, NDOT, MERGE != 0>(acc, partials + poff);
  if constexpr (MODE == 3) {
    __shared__ double ext5[5 * WG];
    if (MERGE && arrive_last(ticket)) finish_reduce_n<5>(partials, poff + (int)gridDim.x, red, sc, MERGE == 1 ? 5 : -1, ext5);
  } else if constexpr (MODE == 2) {
    if (MERGE && arrive_last(ticket))
      finish_reduce<true>(partials, poff + (int)gridDim.x, 2, 1, red, sc, MERGE == 1 ? 4 : -1, MERGE == 1 ? host_status : nullptr);
  } else {
    if (MERGE && arrive_last(ticket))
      finish_reduce<true>(partials, poff + (int)gridDim.x, NDOT, 0, red, sc, MERGE == 1 ? MODE + 1 : -1, nullptr);
  }
}

// ---- sweep E in the preconditioned space ---------------------------------------------------------
// y' = y + alpha p + omega s ; r = s - omega t ; partial(rhat.r, r.r), max|r|   (cuda.cu:498, 520-525, 440-442
// with x = x0 + P_inv y).  y lives in three buffers: y' goes to the one that holds neither y nor the best
// iterate so far, so the reference's copy of the best iterate (cuda.cu:535-538) is a change of index
// (krylov_common.h y_out_buffer) and the sweep moves a flat 56 B/cell.
// s = r - alpha nu' (cuda.cu:499-502) is formed here again from r and nu' -- the same two operands and operation CD used,
// bit for bit -- instead of being stored by CD and read back: a stored double costs the L2 one 64-byte write request per 8
// cells plus the read, the recomputation one more 128-byte read request per 16 (the sweeps are bound by L2 requests).
template <int MERGE>
__global__ __launch_bounds__(WG) void k_sweepE_y(double2 *y0, double2 *y1, double2 *y2, const double2 *__restrict__ p,
                                                 double2 *__restrict__ r, const double2 *__restrict__ nu,
                                                 const double2 *__restrict__ t, const double2 *__restrict__ rhat,
                                                 KrylovScalars *sc, double *partials, size_t n2, double *red,
                                                 unsigned *ticket, int *host_status) {
  if (sc->status != 0) {
    // the solve ended in an iteration that did not report to the host (solve_fused_impl: one report per GROUP of
    // iterations): the group's last sweep E does, also when there is nothing left for it to compute
    if (MERGE == 1 && host_status && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  const double alpha = sc->alpha, omega = sc->omega, momega = -sc->omega, malpha = -sc->alpha;
  const int cur = sc->ycur, out = y_out_buffer(cur, sc->ybest);
  const double2 *__restrict__ yin = cur == 0 ? y0 : (cur == 1 ? y1 : y2);
  double2 *__restrict__ yout = out == 0 ? y0 : (out == 1 ? y1 : y2);
  double sm[2] = {0.0, 0.0}, m[1] = {0.0};
  // the accumulated correction starts at zero (cuda.cu:436-437 in the preconditioned space): the first sweep does not
  // read it -- its buffer is never filled
  const bool first = sc->iter == 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (size_t)gridDim.x * WG) {
    double2 yv = {0.0, 0.0};
    if (!first) yv = ld2<(POL & 0x040) != 0>(yin + i);
    const double2 pv = ld2<(POL & 0x080) != 0>(p + i), ro = r[i], nv = ld2<(POL & 0x100) != 0>(nu + i);
    const double2 tv = ld2<(POL & 0x200) != 0>(t + i), hv = ld2<(POL & 0x400) != 0>(rhat + i);
    double2 sv;
    sv.x = ro.x + malpha * nv.x;
    sv.y = ro.y + malpha * nv.y;
    yv.x = yv.x + alpha * pv.x;
    yv.y = yv.y + alpha * pv.y;
    yv.x = yv.x + omega * sv.x;
    yv.y = yv.y + omega * sv.y;
    st2<(POL & 0x010) != 0>(yout + i, yv);
    double2 rv;
    rv.x = sv.x + momega * tv.x;
    rv.y = sv.y + momega * tv.y;
    st2<(POL & 0x020) != 0>(r + i, rv);
    sm[0] = __builtin_fma(hv.x, rv.x, sm[0]);
    sm[0] = __builtin_fma(hv.y, rv.y, sm[0]);
    sm[1] = __builtin_fma(rv.x, rv.x, sm[1]);
    sm[1] = __builtin_fma(rv.y, rv.y, sm[1]);
    m[0] = fmax(m[0], fmax(fabs(rv.x), fabs(rv.y)));
  }
  workgroup_reduce_store<2, false, MERGE != 0>(sm, partials, 0);
  workgroup_reduce_store<1, true, MERGE != 0>(m, partials, 2);
  if (MERGE && arrive_last(ticket))
    finish_reduce<true>(partials, gridDim.x, 2, 1, red, sc, MERGE == 1 ? 3 : -1, MERGE == 1 ? host_status : nullptr);
}

// the same-level stencil on one or N ranks, or the assembled operator in its hybrid form (one or N ranks: the halo
// columns of its rows are z entries, exchanged between the two launches of a sweep)
bool fused_supported(const cup2d_ctx *c) {
  if (c->mat.active) return c->mat.d_fnbr != nullptr && (c->mat.halo == 0 || c->exchange != nullptr);
  return c->nghost == 0 || c->exchange != nullptr;
}

static int ensure_fused_buffers(cup2d_ctx *c) {
  if (!c->d_fault) CUP2D_HIP_CHECK(dev_malloc(&c->d_fault, sizeof(int)));  // (zero-filled)
  const size_t bytes = (size_t)c->ntotal * BC * sizeof(double);
  double **v[] = {&c->d_p2, &c->d_nu2, &c->d_s, &c->d_y, &c->d_yopt, &c->d_z};
  for (double **p : v)
    if (!*p) {
      CUP2D_HIP_CHECK(dev_malloc(p, bytes));
      CUP2D_HIP_CHECK(hipMemsetAsync(*p, 0, bytes, c->stream));
    }
  return CUP2D_OK;
}

int launch_init_residual(cup2d_ctx *c, double *x, const double *b, int *GP, bool x0_zero);  // krylov.hip

// one wave per 16-block tile; 8 waves = ONE 512-thread workgroup per CU (P_inv fragments + 8 staging tiles
// are 134 KiB of its 160 KiB LDS)
static int fused_grid(const cup2d_ctx *c, int count) {
  const int ntiles = (count + TB - 1) / TB;
  int g = (ntiles + FWAVES - 1) / FWAVES;
  int cap = c->num_cus > 0 ? c->num_cus : 256;
  // N ranks: CUs the persistent sweeps leave to the communication stream's kernels (a multiple of 8: one per XCD and step)
  if (c->spare_cus > 0 && cap > 8 + c->spare_cus) cap -= c->spare_cus;
  if (g > cap) g = cap;
  if (g >= 8) g -= g % 8;
  return g < 1 ? 1 : g;
}

// One fused sweep (MODE 0: A+B, MODE 1: C+D) over all owned blocks; *GP = number of per-workgroup partials written.  (N ranks:
// whole ghost blocks of the sweep's input vectors are in place, solve_fused_impl -- the ghost-EDGE form of round 2, a pre-pass
// for z on the send-list faces + a width-1 exchange + a second launch per sweep, is gone: DESIGN.md 7.)
// The edge form of these two sweeps (krylov_edge.h MODE 0 / 1; CUP2D_FORM_EDGE) applies with the built-in preconditioner on
// the same-level stencil: one rank, or N ranks in the ghost-block form.  Measured at 4096^2 in the first half of round 3
// (tools/gpu_edge_check.py, tools/gpu_calls/gpu_r03_call3.sh; AB / CD in us, L2-miss traffic per launch from FETCH_SIZE /
// WRITE_SIZE):
//   full form (k_fused)            175 / 105     998 MB / 505 MB
//   edge form, no sharing          178 / 109    1032 MB / 525 MB
//   edge form, sharing             175 / 116     896 MB / 456 MB   (reads 763 -> 627 MB: the sibling ring is gone)
//   edge form, loads a job ahead without spills (second half, gpu_r03_call11.sh): 161-166 / 97
// Half the MFMAs, 13 % less traffic and loads requested earlier buy a few per cent at most: the launches move their actual
// traffic at the rate the memory system gives streaming kernels.  What pays is fewer bytes per ITERATION: the organisation
// below (eab_form: two launches, MODE 3 and MODE 2), the default wherever this form applies.  DESIGN.md 4.5.
static int form_of(const cup2d_ctx *c) {  // cup2d_set_solver_form, else the process default
  static const int env = [] {
    const char *e = getenv("CUP2D_FUSED_FORM");
    return !e ? CUP2D_FORM_AUTO : !strcmp(e, "full") ? CUP2D_FORM_FULL : !strcmp(e, "edge") ? CUP2D_FORM_EDGE : !strcmp(e, "eab") ? CUP2D_FORM_EAB : CUP2D_FORM_AUTO;
  }();
  return c->solver_form != CUP2D_FORM_AUTO ? c->solver_form : env;
}
static bool edge_form(const cup2d_ctx *c, bool ghost_blocks, int dbg) {
  const bool on = form_of(c) == CUP2D_FORM_EDGE;
  const bool ghosts = c->nghost > 0 && c->exchange;
  return on && !c->custom_Pinv && !c->mat.active && dbg == 0 && (!ghosts || ghost_blocks);
}
// the organisation with sweep E and the next A+B in one launch (k_edge MODE 2 / 3): the default wherever the edge form
// applies -- built-in preconditioner, same-level stencil -- on one GPU with the finish in the kernel.  CUP2D_FUSED_FORM =
// full | edge selects the three-launch organisation (k_fused | k_edge MODE 0 / 1), eab (or unset) this one.
// N ranks (merge 2): in the ghost-block form -- whole boundary blocks of t behind the reduction of C+D, of r', p'', nu'' in one
// message behind the reduction of the other launch; two reductions over the ranks per iteration instead of three.
static bool ghost_local_enabled() {
  static const bool on = [] { const char *e = getenv("CUP2D_GHOST_LOCAL"); return !e || atoi(e) != 0; }();
  return on;
}
// ... and, one rank, on the hybrid assembled operator of an adapted grid (k_edge HYB + k_hyb_rows: hyb_eab_sweep)
static bool hyb_eab_ok(const cup2d_ctx *c, int merge) {
  return c->mat.active && c->mat.d_fnbr != nullptr && c->mat.halo == 0 && c->nghost == 0 && merge == 1;
}
static bool eab_form(const cup2d_ctx *c, int merge, int dbg, bool ghost_blocks) {
  const bool on = form_of(c) == CUP2D_FORM_AUTO || form_of(c) == CUP2D_FORM_EAB;
  // (on request only -- cup2d_set_solver_form(CUP2D_FORM_EAB) / CUP2D_FUSED_FORM=eab: measured on the 63 k-block grid it does not
  // beat three sweeps + two rows launches, krylov_edge.h HYB)
  if (c->mat.active) return form_of(c) == CUP2D_FORM_EAB && hyb_eab_ok(c, merge) && !c->custom_Pinv && dbg == 0;
  const bool ghosts = c->nghost > 0 && c->exchange;
  // (N ranks: the widest message is two whole blocks per strip -- nu' and p' behind the A+B of iteration 0 -- when r' and p'' of
  // the ghost blocks are formed locally (k_ghost_rp, the default), three with CUP2D_GHOST_LOCAL=0: the caller's buffers must
  // be that wide, cup2d_set_comm_strip_capacity)
  const bool wide = !ghosts || c->strip_cap >= (ghost_local_enabled() ? 2 : 3) * BC;
  return on && (merge == 1 || merge == 2) && !c->custom_Pinv && !c->mat.active && (!ghosts || (merge == 2 && ghost_blocks)) && dbg == 0 && wide;
}
static int edge_share_of(cup2d_ctx *c) {  // the grid allows it: every tile has <= 16 perimeter sides
  if (c->edge_share < 0) c->edge_share = edge_share_ok(c->h_nbr.data(), 0, c->nblocks) ? 1 : 0;
  return c->edge_share;
}
// sharing between sibling waves per kind of sweep: CUP2D_EDGE_SHARE = bit mask, bit MODE; default 0b1101 (A+B, MODE 2, MODE 3).
// It pays where the ring is four vectors wide (MODE 2); in the short C+D sweeps the waiting eats what it saves -- MODE 1 off;
// MODE 3 (round 6, gpu_r06_call22.sh, alternating processes): 4096^2 120.2 / 119.6 us without, 120.6 / 121.2 with (nothing);
// 2048^2 39.2 / 39.0 -> 38.1 / 38.1 us, 755 -> 766 Mcell-updates/s: on, for the small grids and the ranks of configs[3]
static int edge_share_mode(cup2d_ctx *c, int mode) {
  static const int mask = [] { const char *e = getenv("CUP2D_EDGE_SHARE"); return e ? atoi(e) : 13; }();
  if (!edge_share_of(c)) return 0;
  return (mask >> mode) & 1;
}
// N ranks, two-launch organisation: of the three vectors the launch that holds sweep E and the next A+B leaves behind -- r', p'',
// nu'' -- only nu'' = A P_inv p'' is not a function of the same cells.  A rank holds p', nu', r and t of its ghost blocks (the
// ring entries of that very launch), so it forms r' = (r - alpha nu') - omega t and p'' = beta' (p' - omega nu') + r' of the
// ghost blocks ITSELF, with the scalars and the operations the owner used (form_v of krylov_edge.h MODE 2, operation for
// operation: the same bits), and one vector travels instead of three (a third of the bytes on the link: 262 KB instead of
// 786 KB per neighbour on a 4096-cell side).  Runs behind the launch and BEFORE its reduction is finished (stage 4 replaces
// alpha).  CUP2D_GHOST_LOCAL=0: the three vectors travel (the results are the same bit for bit, tests/test_comm.py).
__global__ __launch_bounds__(WG) void k_ghost_rp(const double *__restrict__ p, const double *__restrict__ nu, const double *__restrict__ r,
                                                 const double *__restrict__ t, double *__restrict__ rout, double *__restrict__ pout,
                                                 const KrylovScalars *sc, size_t first, size_t count) {
  if (sc->status != 0) return;
  const double malpha = -sc->alpha, c1 = -sc->omega, beta = sc->beta;
  const bool restart = sc->restart_flag != 0;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < count; i += (size_t)gridDim.x * WG) {
    const size_t k = first + i;
    const double sv = r[k] + malpha * nu[k];
    const double rn = sv + c1 * t[k];
    rout[k] = rn;
    if (restart) pout[k] = rn;
    else {
      double v = p[k] + c1 * nu[k];
      v = v * beta;
      pout[k] = v + rn;
    }
  }
}
static int ghost_rp(cup2d_ctx *c, const double *p, const double *nu, const double *r, const double *t, double *rout, double *pout) {
  const size_t first = (size_t)c->nblocks * BC, count = (size_t)c->nghost * BC;
  if (count == 0) return CUP2D_OK;
  int grid = (int)((count + WG - 1) / WG);
  if (grid > c->grid) grid = c->grid;
  ProfScope prof(c, CUP2D_T_HALO);
  hipLaunchKernelGGL(k_ghost_rp, dim3(grid), dim3(WG), 0, c->stream, p, nu, r, t, rout, pout, (const KrylovScalars *)c->d_sc, first, count);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// blocks [first, first + count) (a multiple of 16 blocks in front of it: the tiling is the whole range's); merge 0: the launch
// leaves its partials at [poff, poff + grid) and a later launch of the same sweep finishes over all of them.  *G: its grid.
template <int MODE>
static int eab_sweep(cup2d_ctx *c, const FusedArgs &a, int merge, int first, int count, int poff, int *G, KrylovScalars *sc_in = nullptr) {
  const int g = fused_grid(c, count);
  const int share = edge_share_mode(c, MODE);
  const FusedArgs &a2 = a;
  const auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(g), dim3(FWG), EDGE_LDS_BYTES, c->stream, a2, c->d_Pinv, c->d_nbr, sc_in ? sc_in : c->d_sc, c->d_partials, first, count,
                       poff, share, c->d_red, c->d_ticket, c->d_fault);
  };
  if (merge == 1) go(k_edge<MODE, 1>);
  else if (merge == 2) go(k_edge<MODE, 2>);
  else if (merge == 3 || merge == 4) {  // N ranks, deferred scalar update: MODE 2 / 3 only (the A+B of iteration 0 keeps the scalar kernel behind it)
    if constexpr (MODE >= 2) {
      if (merge == 3) go(k_edge<MODE, 3>);
      else go(k_edge<MODE, 4>);
    } else { set_error("eab_sweep: MERGE 3 / 4 are for MODE 2 / 3"); return CUP2D_ERR_ARG; }
  } else go(k_edge<MODE, 0>);
  CUP2D_HIP_CHECK(hipGetLastError());
  if (G) *G = g;
  return CUP2D_OK;
}
template <int MODE>
static int eab_sweep(cup2d_ctx *c, const FusedArgs &a, int merge) { return eab_sweep<MODE>(c, a, merge, 0, c->nblocks, 0, nullptr); }

// One sweep of the two-launch organisation on the hybrid operator (one rank, finish in the kernel): k_edge HYB over the tiles,
// then the rows of the general tiles, whose last workgroup finishes the reduction of both launches (and, MODE 2, reports)
template <int MODE>
static int hyb_eab_sweep(cup2d_ctx *c, FusedArgs a, int *host_status) {
  static_assert(MODE == 2 || MODE == 3, "hyb_eab_sweep: the A+B of iteration 0 is fused_sweep<0>");
  const SellMatrix &M = c->mat;
  a.zg = c->d_z; a.zmask = M.d_zmask; a.tile0 = M.d_tile0;
  int g = (M.ntiles + FWAVES - 1) / FWAVES;
  const int cus = c->num_cus > 0 ? c->num_cus : 256;
  if (g > cus) g = cus;
  if (g >= 8) g -= g % 8;
  if (g < 1) g = 1;
  const auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(g), dim3(FWG), EDGE_LDS_BYTES, c->stream, a, c->d_Pinv, M.d_fnbr, c->d_sc, c->d_partials, 0, M.ntiles,
                       0, 0, c->d_red, c->d_ticket, c->d_fault);
  };
  if (M.ngen == 0) {
    a.host_status = host_status;
    go(k_edge<MODE, 1, true>);
    CUP2D_HIP_CHECK(hipGetLastError());
    return CUP2D_OK;
  }
  a.host_status = nullptr;
  go(k_edge<MODE, 0, true>);
  CUP2D_HIP_CHECK(hipGetLastError());
  int g2 = (M.ngen + RWAVES - 1) / RWAVES;
  if (g2 > cus) g2 = cus;
  const double *w0 = MODE == 2 ? a.w : a.in0, *w1 = MODE == 2 ? nullptr : a.in1, *w2 = MODE == 2 ? nullptr : a.w;
  hipLaunchKernelGGL((k_hyb_rows<MODE, 1>), dim3(g2), dim3(RWG), 0, c->stream, (const double *)c->d_z, a.yout, w0, w1, (const RowsRec *)M.d_rrec,
                     M.d_col, M.d_val, M.ngen, c->d_sc, c->d_partials, g, c->d_red, c->d_ticket, w2, host_status);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

template <int MODE>
static int fused_sweep(cup2d_ctx *c, const FusedArgs &a, int merge, int dbg, int *GP, bool ghost_blocks) {
  const int nb = c->nblocks;
  if (edge_form(c, ghost_blocks, dbg)) {
    const int share = edge_share_mode(c, MODE);
    const int g = fused_grid(c, nb);
    const auto go = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(g), dim3(FWG), EDGE_LDS_BYTES, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc, c->d_partials, 0, nb,
                         0, share, c->d_red, c->d_ticket, c->d_fault);
    };
    if (merge == 1) go(k_edge<MODE, 1>);
    else if (merge == 2) go(k_edge<MODE, 2>);
    else go(k_edge<MODE, 0>);
    CUP2D_HIP_CHECK(hipGetLastError());
    *GP = g;
    return CUP2D_OK;
  }
  if (c->mat.active) {
    // hybrid assembled operator: the fused tiles in one launch; then the halo entries of z, then the rows of the general
    // tiles, whose last workgroup finishes the reduction of both launches
    const SellMatrix &M = c->mat;
    int g = (M.ntiles + FWAVES - 1) / FWAVES;
    const int cus = c->num_cus > 0 ? c->num_cus : 256;
    if (g > cus) g = cus;
    if (g >= 8) g -= g % 8;
    if (g < 1) g = 1;
    const int m1 = M.ngen > 0 ? 0 : merge;
    const auto go = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(g), dim3(FWG), FUSED_LDS_BYTES, c->stream, a, c->d_Pinv, M.d_fnbr, c->d_sc, c->d_partials,
                         0, M.ntiles, 0, c->ntotal, c->d_z, c->d_red, c->d_ticket, dbg, M.d_zmask, M.d_tile0);
    };
    if (m1 == 1) go(k_fused<MODE, 1, true>);
    else if (m1 == 2) go(k_fused<MODE, 2, true>);
    else go(k_fused<MODE, 0, true>);
    CUP2D_HIP_CHECK(hipGetLastError());
    *GP = g;
    if (M.ngen == 0) return CUP2D_OK;
    CUP2D_TRY(matrix_exchange(c, c->d_z));
    int g2 = (M.ngen + RWAVES - 1) / RWAVES;
    if (g2 > cus) g2 = cus;  // (2 or 4 workgroups per CU change nothing: 7.89 / 8.03 / 8.06 ms per adapted-grid step, round 4)
    const double *w0 = MODE == 0 ? a.w : a.in0, *w1 = MODE == 0 ? nullptr : a.in1;
    const auto rows = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(g2), dim3(RWG), 0, c->stream, (const double *)c->d_z, a.yout, w0, w1, (const RowsRec *)M.d_rrec, M.d_col,
                         M.d_val, M.ngen, c->d_sc, c->d_partials, g, c->d_red, c->d_ticket, (const double *)nullptr, (int *)nullptr);
    };
    if (merge == 1) rows(k_hyb_rows<MODE, 1>);
    else if (merge == 2) rows(k_hyb_rows<MODE, 2>);
    else rows(k_hyb_rows<MODE, 0>);
    CUP2D_HIP_CHECK(hipGetLastError());
    *GP = g + g2;
    return CUP2D_OK;
  }
  if (ghost_blocks) {
    // N ranks, ghost-block form: the ghost copies of the sweep's input vectors are complete (solve_fused_impl
    // exchanges whole boundary blocks behind the reductions), so a ghost block is a ring entry like any other
    // neighbour outside the tile -- its z edge is recomputed here -- and the sweep is ONE launch over the owned blocks
    const int g = fused_grid(c, nb);
    const auto go = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(g), dim3(FWG), FUSED_LDS_BYTES, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc, c->d_partials,
                         0, nb, 0, c->ntotal, c->d_z, c->d_red, c->d_ticket, dbg, nullptr, nullptr);
    };
    if (merge == 1) go(k_fused<MODE, 1>);
    else if (merge == 2) go(k_fused<MODE, 2>);
    else go(k_fused<MODE, 0>);
    CUP2D_HIP_CHECK(hipGetLastError());
    *GP = g;
    return CUP2D_OK;
  }
  // one rank, same-level stencil: one launch over all blocks
  const int g = fused_grid(c, nb);
  const auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(g), dim3(FWG), FUSED_LDS_BYTES, c->stream, a, c->d_Pinv, c->d_nbr, c->d_sc, c->d_partials,
                       0, nb, 0, nb, c->d_z, c->d_red, c->d_ticket, dbg, nullptr, nullptr);
  };
  if (merge == 1) go(k_fused<MODE, 1>);
  else if (merge == 2) go(k_fused<MODE, 2>);
  else go(k_fused<MODE, 0>);
  CUP2D_HIP_CHECK(hipGetLastError());
  *GP = g;
  return CUP2D_OK;
}

// ---- where the solver's vectors lie -----------------------------------------------------------------------------------------
// The two launches of an iteration stream eleven vectors of ntotal blocks each (p, nu, r in two buffers, t, y in three, rhat).
// Their durations come in modes -- 324-330, 341-347 and 357-371 us per iteration at 4096^2 (1 : 1.05 : 1.12) -- and which one a
// solve gets is a property of WHERE those eleven buffers landed in device memory, nothing else: contexts created one after the
// other in one process and kept alive each run the same again on every later solve (tools/gpu_placement_modes.py).  What round 6
// found out about it (tools/gpu_calls/gpu_r06_call23.sh ... call31.sh, profiles/r06_placement_repair.txt):
//   * every vector of every set, read ALONE, streams at the same rate (25.3 us for 134 MB): no allocation is slow by itself;
//   * a slow set is slow because of a CONFLICT between two or three of its vectors, streams the launches write among them every
//     time: the slowest set of a process with ONE vector taken from the fastest set runs 347 (s), 349 (p), 349 (nu) instead of
//     371 us, with any of the other eight as before; in another process t 348, xopt 351.  The fastest set with any one vector of
//     the slowest stays fast;
//   * the conflict is not one of addresses below 64 MiB (pseudo-random offsets of the vectors inside ONE arena, physically
//     contiguous or not, granularity 256 B to 1 MiB: 80 candidates, all slow; arenas at any stride: slow) -- the vectors of a set
//     are neighbours in physical memory, and it takes a vector from somewhere else to end it;
//   * a slow set IS repairable with vectors of other slow sets: the slowest of 12 sets, its s, p, t (and one more) exchanged
//     with the same slots of other slow sets: 371.7 -> 323.8, 365.4 -> 322.0, 367.8 -> 321.2 us in three processes -- faster than
//     any complete set the allocator handed out (324-330).
// So the first fused solve of a context on a large grid (1) allocates CUP2D_PLACEMENT_TRIES (default 8) complete sets of the
// eleven vectors -- all held while it looks, so that every set is other memory --, (2) times three iterations' worth of the two
// launches on each (the MERGE 0 instances on zero-filled vectors with scratch scalar records -- one per rotation of the three y
// buffers --: no reduction finish, nothing of the context's state touched), (3) REPAIRS the fastest: slot by slot, written
// streams first, its vector is exchanged with the same slot's of up to four other sets and the exchange kept where it gains
// 1.5 % (and, if that made no fast set, once more with every other set as a donor and 1 % as the bar), (4) if fastest and slowest then
// the fastest still lies less than 10.5 % below the median (no fast set seen or made) goes on with more sets, up
// to six times as many (three on N ranks) within CUP2D_PLACEMENT_MAX_GB (64 GB on one rank, 40 GB on N) and a quarter of the
// free memory, and repairs once more, (5) keeps the fastest set and gives the others back.  5 ms per set, 2 ms per exchange
// tried: 0.1 s, once per context.  CUP2D_PLACEMENT_ARENA="pad,pad,..." adds arenas (one allocation carved at 2^27 + pad) as
// candidates: the experiment's switch.
static int tune_placement(cup2d_ctx *c) {
  static const int tries_env = [] { const char *e = getenv("CUP2D_PLACEMENT_TRIES"); return e ? atoi(e) : 8; }();
  // what the search may hold beyond the context's own set while it looks (per process: ranks that share a GPU each search)
  static const double budget_env = [] { const char *e = getenv("CUP2D_PLACEMENT_MAX_GB"); return e ? atof(e) : -1.0; }();
  // (a context with ghost blocks is one of several ranks, which may share a GPU in the tests: the smaller bound)
  const bool one_rank = c->nghost == 0 && !c->exchange;
  const double budget_gb = budget_env >= 0.0 ? budget_env : one_rank ? 64.0 : 40.0;
  if (c->placement_tuned) return CUP2D_OK;
  c->placement_tuned = true;
  const size_t bytes = (size_t)c->ntotal * BC * sizeof(double);
  if (tries_env <= 1 || bytes < ((size_t)32 << 20)) return CUP2D_OK;  // below 2048^2 a launch is a few rounds: nothing to choose
  constexpr int NV = 11;
  double **slot[NV] = {&c->d_r, &c->d_s, &c->d_p, &c->d_p2, &c->d_nu, &c->d_nu2, &c->d_t, &c->d_y, &c->d_yopt, &c->d_xopt, &c->d_rhat};
  size_t free_b = 0, total_b = 0;
  CUP2D_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  int tries = tries_env > 48 ? 48 : tries_env;
  const size_t budget = std::min((size_t)(budget_gb * (double)((size_t)1 << 30)), free_b / 4);  // never more than a quarter of what is free
  while (tries > 1 && (size_t)(tries - 1) * NV * bytes > budget) tries--;
  if (tries <= 1) return CUP2D_OK;
  // (how rare fast sets are depends on the box: four of 16 in one, one of 25 in two processes of another)
  int tries_more = (one_rank ? 6 : 3) * tries > 48 ? 48 : (one_rank ? 6 : 3) * tries;
  while (tries_more > tries && (size_t)(tries_more - 1) * NV * bytes > budget) tries_more--;
  StageClock clk("tune_placement");
  static const std::vector<size_t> arena_pads = [] {
    std::vector<size_t> v;
    if (const char *e = getenv("CUP2D_PLACEMENT_ARENA"))
      for (const char *q = e; *q;) {
        char *end = nullptr;
        const long long x = strtoll(q, &end, 10);
        if (end == q) break;
        if (x >= 0) v.push_back((size_t)x & ~(size_t)255);
        q = *end == ',' ? end + 1 : end;
      }
    return v;
  }();
  struct Cand { double *v[NV]; float ms; void *arena; long long pad; };
  std::vector<Cand> cand((size_t)tries_more);
  for (auto &C : cand) { C.arena = nullptr; C.pad = -1; C.ms = 0.f; }
  for (int k = 0; k < NV; k++) cand[0].v[k] = *slot[k];
  cand[0].arena = c->vec_arena;
  // three scratch records: sweep E writes the y buffer that is neither the current nor the best one -- every rotation is timed
  KrylovScalars hs[3];
  for (int q = 0; q < 3; q++) {
    ::memset(&hs[q], 0, sizeof hs[q]);
    hs[q].alpha = hs[q].beta = hs[q].omega = hs[q].omega_r = hs[q].rho_prev = hs[q].rho_curr = 1.0;
    hs[q].eps = 1e-21; hs[q].err = hs[q].err_init = hs[q].err_opt = 1.0; hs[q].max_error = -1.0; hs[q].max_rel_error = -1.0;
    hs[q].max_iter = 1 << 30; hs[q].iter = 1; hs[q].ycur = q; hs[q].ybest = (q + 1) % 3;
  }
  KrylovScalars *d_scratch = nullptr;
  CUP2D_HIP_CHECK(dev_malloc(&d_scratch, sizeof hs));
  CUP2D_HIP_CHECK(hipMemcpyAsync(d_scratch, hs, sizeof hs, hipMemcpyHostToDevice, c->stream));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  CUP2D_HIP_CHECK(hipEventCreate(&e0));
  CUP2D_HIP_CHECK(hipEventCreate(&e1));
  const int nb = c->nblocks, g = fused_grid(c, nb);
  int rc = CUP2D_OK, made = 1, probes = 0;
  const auto probe = [&](Cand &C) -> int {
    double *r = C.v[0], *sv = C.v[1], *p = C.v[2], *p2 = C.v[3], *nu = C.v[4], *nu2 = C.v[5], *t = C.v[6], *y = C.v[7], *yo = C.v[8],
           *xo = C.v[9], *rh = C.v[10];
    FusedArgs a3 = {}, a2 = {};
    a3.in0 = r; a3.in1 = nu2; a3.w = rh; a3.yout = t; a3.rev = 1;
    a2.in0 = p2; a2.in1 = nu2; a2.in2 = r; a2.w = rh; a2.vout = p; a2.yout = nu; a2.t = t; a2.y0 = y; a2.y1 = yo; a2.y2 = xo; a2.rout = sv;
    const auto pair = [&](int rot) {
      hipLaunchKernelGGL((k_edge<3, 0>), dim3(g), dim3(FWG), EDGE_LDS_BYTES, c->stream, a3, c->d_Pinv, c->d_nbr, d_scratch + rot, c->d_partials, 0, nb, 0,
                         edge_share_mode(c, 3) & 1, c->d_red, c->d_ticket, c->d_fault);
      hipLaunchKernelGGL((k_edge<2, 0>), dim3(g), dim3(FWG), EDGE_LDS_BYTES, c->stream, a2, c->d_Pinv, c->d_nbr, d_scratch + rot, c->d_partials, 0, nb, 0,
                         edge_share_mode(c, 2) & 1, c->d_red, c->d_ticket, c->d_fault);
    };
    pair(0);  // (warm: the first touch of fresh memory is not what a solve sees)
    CUP2D_HIP_CHECK(hipEventRecord(e0, c->stream));
    for (int rep = 0; rep < 3; rep++) pair(rep);
    CUP2D_HIP_CHECK(hipEventRecord(e1, c->stream));
    CUP2D_HIP_CHECK(hipEventSynchronize(e1));
    CUP2D_HIP_CHECK(hipGetLastError());
    CUP2D_HIP_CHECK(hipEventElapsedTime(&C.ms, e0, e1));
    C.ms /= 3.0f;
    probes++;
    return CUP2D_OK;
  };
  // (the context's own vectors may hold anything: the probe computes on what is there; values do not change a duration)
  rc = probe(cand[0]);
  float lo_ms = cand[0].ms, hi_ms = cand[0].ms;
  const float first_ms = cand[0].ms;
  // "a fast set has shown": the fastest 10.5 % below the median.  (Fastest against slowest let middle sets through beside a slow one of
  // the upper end -- 337 kept beside 372, tools/gpu_calls/gpu_r06_call44.sh --: the modes are ranges, 321-330 | 337-347 | 357-381)
  // ... measured against the MEDIAN of the complete sets as the allocator handed them out, which is a slow one on every box seen
  // (the slowest varies, 365-381: against it 333 passed beside 381 on a box where fast sets are rare, gpu_r06_call57.sh)
  std::vector<float> natural(1, cand[0].ms);
  const auto fast_seen = [&]() {
    std::vector<float> v(natural);
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    return v.size() >= 3 && lo_ms < 0.895f * v[v.size() / 2] && hi_ms - lo_ms > 0.095f * lo_ms;
  };
  std::string repair_log;
  // step (3) over the sets [0, made): separate allocations only (an arena's vectors are not its own to give away)
  int repaired_at = 0;  // the number of sets the last repair looked at
  // wide: up to twelve donors per slot and 1 % as the bar (the second pass over a batch whose best set the first did not make fast).
  // Not every box's slow sets can be repaired by single exchanges: on one (tools/gpu_calls/gpu_r06_call32.sh, second run) the best
  // of a first batch without a fast set went 361.3 -> 355.3 and no further in three of six processes, and a complete fast set turned
  // up only among 41-42 sets (kept 322 / 332 / 329): the search for more sets stays
  const auto repair = [&](bool wide) -> int {
    if (made < 3 || (made == repaired_at && !wide)) return CUP2D_OK;
    repaired_at = made;
    int b = 0;
    for (int q = 1; q < made; q++) if (cand[q].ms < cand[b].ms) b = q;
    if (cand[b].arena || cand[b].pad != -1) return CUP2D_OK;
    int ndonors = 0;
    for (int q = 0; q < made; q++) ndonors += q != b && !cand[q].arena && cand[q].pad == -1;
    if (ndonors < 2) return CUP2D_OK;
    static const int order[NV] = {1, 2, 4, 9, 6, 7, 8, 0, 3, 5, 10};  // s, p, nu, xopt, t, y, yopt, r, p2, nu2, rhat
    const int p0 = probes;
    // (the whole set at once -- every slot from ANOTHER set, no two vectors neighbours any more -- was tried first and dropped: 326.9 ->
    // 336.3, 325.5 -> 331.4, 330.5 -> 328.4 ... in six processes: far apart is not the criterion; one exchange at a time is what works)
    for (int oi = 0; oi < NV; oi++) {
      const int k = order[oi];
      int tried = 0;
      for (int j = 1; j < made && tried < (wide ? 12 : 4); j++) {
        const int q = (b + j + 2 * oi) % made;
        if (q == b || cand[q].arena || cand[q].pad != -1) continue;
        tried++;
        const float before = cand[b].ms;
        std::swap(cand[b].v[k], cand[q].v[k]);
        CUP2D_TRY(probe(cand[b]));
        if (cand[b].ms < (wide ? 0.99f : 0.985f) * before) {
          char buf[64];
          snprintf(buf, sizeof buf, " slot %d from set %d: %.1f -> %.1f;", k, q, 1e3 * before, 1e3 * cand[b].ms);
          repair_log += buf;
          lo_ms = cand[b].ms < lo_ms ? cand[b].ms : lo_ms;
          break;
        }
        std::swap(cand[b].v[k], cand[q].v[k]);
        cand[b].ms = before;
      }
    }
    // ... and, in the wide pass, TWO written streams at once from one donor set (where no single exchange helps, the conflict may
    // need two vectors moved: a box whose slow sets single exchanges did not repair, profiles/r06_placement_repair.txt item 11.
    // Tried alone on the slowest set of a batch, tools/gpu_calls/gpu_r06_call54.sh: 370.3 -> 335.5 -> 320.8, 368.1 -> 333.7 -> 323.6,
    // 373.2 -> 356.4 -> 336.3 -> 324.6 us in two or three pair exchanges)
    if (wide && !fast_seen()) {
      static const int wr[5] = {1, 2, 4, 6, 9};  // s, p, nu, t, xopt
      for (int a = 0; a < 5 && !fast_seen(); a++)
        for (int e = a + 1; e < 5 && !fast_seen(); e++) {
          int tried = 0;
          for (int j = 1; j < made && tried < 4; j++) {
            const int q = (b + j + a + 2 * e) % made;
            if (q == b || cand[q].arena || cand[q].pad != -1) continue;
            tried++;
            const float before = cand[b].ms;
            std::swap(cand[b].v[wr[a]], cand[q].v[wr[a]]);
            std::swap(cand[b].v[wr[e]], cand[q].v[wr[e]]);
            CUP2D_TRY(probe(cand[b]));
            if (cand[b].ms < 0.98f * before) {
              char buf2[80];
              snprintf(buf2, sizeof buf2, " slots %d + %d from set %d: %.1f -> %.1f;", wr[a], wr[e], q, 1e3 * before, 1e3 * cand[b].ms);
              repair_log += buf2;
              lo_ms = cand[b].ms < lo_ms ? cand[b].ms : lo_ms;
              break;
            }
            std::swap(cand[b].v[wr[a]], cand[q].v[wr[a]]);
            std::swap(cand[b].v[wr[e]], cand[q].v[wr[e]]);
            cand[b].ms = before;
          }
        }
    }
    char buf[64];
    snprintf(buf, sizeof buf, " (set %d, %d probes%s)", b, probes - p0, wide ? ", wide" : "");
    repair_log += buf;
    return CUP2D_OK;
  };
  for (int q = 1; q < tries_more && rc == CUP2D_OK; q++) {
    if (q == tries) {  // the first batch is in: its best set repaired before more memory is asked for
      rc = repair(false);
      if (rc == CUP2D_OK && !fast_seen()) rc = repair(true);
    }
    if (q >= tries && (rc != CUP2D_OK || fast_seen())) break;  // a fast and a slow set seen (or made): decided
    bool ok = true;
    for (int k = 0; k < NV; k++) cand[q].v[k] = nullptr;
    if ((size_t)(q - 1) < arena_pads.size()) {
      const size_t stride = bytes + arena_pads[(size_t)(q - 1)];
      void *A = nullptr;
      ok = hipMalloc(&A, stride * NV) == hipSuccess && hipMemsetAsync(A, 0, stride * NV, c->stream) == hipSuccess;
      if (ok) {
        cand[q].arena = A;
        cand[q].pad = (long long)arena_pads[(size_t)(q - 1)];
        for (int k = 0; k < NV; k++) cand[q].v[k] = reinterpret_cast<double *>(static_cast<char *>(A) + (size_t)k * stride);
      } else if (A) (void)hipFree(A);
    } else {
      for (int k = 0; k < NV && ok; k++) ok = dev_malloc(&cand[q].v[k], bytes) == hipSuccess;  // (zero-filled)
    }
    made = q + 1;
    if (!ok) {  // out of memory: what exists is enough
      (void)hipGetLastError();
      if (!cand[q].arena)
        for (int k = 0; k < NV; k++) dev_release(cand[q].v[k]);
      made = q;
      break;
    }
    rc = probe(cand[q]);
    if (rc == CUP2D_OK) {
      natural.push_back(cand[q].ms);
      const bool new_best = cand[q].ms < lo_ms;
      lo_ms = cand[q].ms < lo_ms ? cand[q].ms : lo_ms;
      hi_ms = cand[q].ms > hi_ms ? cand[q].ms : hi_ms;
      // a set past the first batch that is nearly fast (327.6 against a median of 364: 47 sets were allocated behind it for nothing,
      // 2 s, before the last repair made it 320.5 -- tools/gpu_calls/gpu_r06_call58.sh): repaired at once
      if (q >= tries && new_best && !fast_seen()) {
        std::vector<float> v(natural);
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        if (lo_ms < 0.93f * v[v.size() / 2]) rc = repair(false);
      }
    }
  }
  // (fewer sets than a first batch -- the budget --, or every set the budget allows timed and none fast: on all of them)
  if (rc == CUP2D_OK && (repaired_at == 0 || !fast_seen())) {
    rc = repair(false);
    if (rc == CUP2D_OK && !fast_seen()) rc = repair(true);
  }
  int best = 0;
  if (rc == CUP2D_OK)
    for (int q = 1; q < made; q++)
      if (cand[q].ms < cand[best].ms) best = q;
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  for (int q = 0; q < made; q++) {  // the sets that lost go back to the driver, not into the process pool (dev_release)
    if (q == best) continue;
    if (cand[q].arena) (void)hipFree(cand[q].arena);
    else
      for (int k = 0; k < NV; k++) dev_release(cand[q].v[k]);
  }
  for (int k = 0; k < NV; k++) *slot[k] = cand[best].v[k];
  c->vec_arena = cand[best].arena;  // (interior pointers then: cup2d_destroy frees the arena, not the vectors)
  // the survivors start a solve as every solver vector does: zero
  for (int k = 0; k < NV; k++) CUP2D_HIP_CHECK(hipMemsetAsync(cand[best].v[k], 0, bytes, c->stream));
  CUP2D_HIP_CHECK(hipMemsetAsync(c->d_fault, 0, sizeof(int), c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dev_free(d_scratch);
  c->placement_candidates = made;
  c->placement_best_us = 1e3 * cand[best].ms;
  c->placement_worst_us = 1e3 * hi_ms;   // the slowest complete set as the allocator handed it out
  c->placement_first_us = 1e3 * first_ms;
  if (clk.on) {
    // (sets that gave a vector to the repaired one are listed with the time they had before)
    for (int q = 0; q < made; q++)
      fprintf(stderr, "[cup2d timing] tune_placement: set %d (%s, pad %lld, first vector at %p): %.1f us per iteration%s\n", q,
              cand[q].arena ? "arena" : "separate", cand[q].pad, (void *)cand[q].v[0], 1e3 * cand[q].ms, q == best ? "  <- kept" : "");
    if (!repair_log.empty()) fprintf(stderr, "[cup2d timing] tune_placement: repair:%s\n", repair_log.c_str());
    clk.lap("search");
  }
  return rc;
}

// b = TMP, x0 = PRES, result -> PRES (same contract as solve_impl)
int solve_fused_impl(cup2d_ctx *c, double max_error, double max_rel_error, int max_restarts, int max_iter, int *iters,
                     int *restarts, double *linf, double *linf_init) {
  CUP2D_TRY(ensure_fused_buffers(c));
  c->spare_cus = 0;
  c->have_last = false;  // whatever happens below, the previous solve's last iterate is no longer this solve's
  // cup2d_step's solve on the same-level stencil: the initial guess is zero and PRES need not hold it (api.hip)
  const bool x0_zero = c->x0_is_zero && !c->mat.active;
  const int nb = c->nblocks;
  const size_t n = (size_t)nb * BC;
  double *x = c->d_field[CUP2D_PRES];
  const double *b = c->d_field[CUP2D_TMP];
  KrylovScalars init;
  ::memset(&init, 0, sizeof init);
  init.alpha = init.beta = init.omega = init.omega_r = init.rho_prev = init.rho_curr = 1.0;
  init.eps = 1e-21;  // cuda.cu:409
  init.err = init.err_init = init.err_opt = 1e50;
  init.max_error = max_error; init.max_rel_error = max_rel_error;
  init.max_restarts = max_restarts; init.max_iter = max_iter;
  *c->h_sc = init;
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_sc, c->h_sc, sizeof init, hipMemcpyHostToDevice, c->stream));
  // sweep E: two workgroups per CU.  Its last workgroup finishes the reduction (arrive_last), and every workgroup pays an
  // agent-scope ticket for that: 2048 workgroups cost the sweep 8 us more than 512 (measured 161.5 / 161.7 / 152.9 / 152.9 us
  // at 2048 / 1024 / 512 / 256 workgroups, 4096^2)
  int gridE = (int)((n / 2 + WG - 1) / WG);
  const int cap = 2 * (c->num_cus > 0 ? c->num_cus : 256);
  if (gridE > cap) gridE = cap;
  if (gridE > c->grid) gridE = c->grid;
  if (!c->fused_lds_opt_in) {  // > 64 KiB of LDS is an opt-in per kernel AND device: remembered per context
    const void *ks[] = {reinterpret_cast<const void *>(&k_fused<0, 0>), reinterpret_cast<const void *>(&k_fused<0, 1>),
                        reinterpret_cast<const void *>(&k_fused<0, 2>), reinterpret_cast<const void *>(&k_fused<1, 0>),
                        reinterpret_cast<const void *>(&k_fused<1, 1>), reinterpret_cast<const void *>(&k_fused<1, 2>),
                        reinterpret_cast<const void *>(&k_fused<0, 0, true>), reinterpret_cast<const void *>(&k_fused<0, 1, true>),
                        reinterpret_cast<const void *>(&k_fused<0, 2, true>), reinterpret_cast<const void *>(&k_fused<1, 0, true>),
                        reinterpret_cast<const void *>(&k_fused<1, 1, true>), reinterpret_cast<const void *>(&k_fused<1, 2, true>)};
    for (const void *k : ks)
      CUP2D_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS_BYTES));
    const void *ke[] = {reinterpret_cast<const void *>(&k_edge<0, 0>), reinterpret_cast<const void *>(&k_edge<0, 1>),
                        reinterpret_cast<const void *>(&k_edge<0, 2>), reinterpret_cast<const void *>(&k_edge<1, 0>),
                        reinterpret_cast<const void *>(&k_edge<1, 1>), reinterpret_cast<const void *>(&k_edge<1, 2>),
                        reinterpret_cast<const void *>(&k_edge<2, 1>), reinterpret_cast<const void *>(&k_edge<3, 1>),
                        reinterpret_cast<const void *>(&k_edge<2, 2>), reinterpret_cast<const void *>(&k_edge<3, 2>),
                        reinterpret_cast<const void *>(&k_edge<2, 0>), reinterpret_cast<const void *>(&k_edge<3, 0>),
                        reinterpret_cast<const void *>(&k_edge<2, 3>), reinterpret_cast<const void *>(&k_edge<3, 3>),
                        reinterpret_cast<const void *>(&k_edge<2, 4>), reinterpret_cast<const void *>(&k_edge<3, 4>),
                        reinterpret_cast<const void *>(&k_edge<2, 0, true>), reinterpret_cast<const void *>(&k_edge<3, 0, true>),
                        reinterpret_cast<const void *>(&k_edge<2, 1, true>), reinterpret_cast<const void *>(&k_edge<3, 1, true>)};
    for (const void *k : ke)
      CUP2D_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EDGE_LDS_BYTES));
    c->fused_lds_opt_in = true;
  }
  // in-kernel finish.  1: one GPU -- one launch per sweep, the last workgroup also runs the scalar update.
  // 2: N GPUs -- the last workgroup of the sweep's last launch sums this rank's partials, then all-reduce + k_scalars
  const int merge = !c->finish_in_kernel ? 0 : (c->allreduce || (c->nghost > 0 && c->exchange)) ? 2 : 1;
  // (restart in the hybrid form: rhat = r is written by k_fused for every block before k_hyb_rows reads it)
  // N ranks: how a sweep learns z on the other side of a rank boundary -- whole boundary blocks of nu', p', r are exchanged as
  // they are produced (nu' and p' in one message behind the reduction of AB, r behind the reduction of E) and the sweeps
  // recompute the z edge of a ghost block like that of any other block outside the tile: one launch per sweep, no exchange on
  // its critical path
  const bool gb = c->nghost > 0 && c->exchange && !c->mat.active;
  const int dbg = 0;  // (k_fused's timing-only knock-outs: development builds pass a mask here)

  // the first solve of a context in the two-launch organisation chooses where its vectors lie (tune_placement) -- before
  // anything of this solve is in them; the sc record of this solve was uploaded above and the probe uses its own
  if (eab_form(c, merge, dbg, gb) && !c->mat.active) CUP2D_TRY(tune_placement(c));
  int GP = 0;
  {
    ProfScope prof(c, CUP2D_T_INIT_RESIDUAL);
    CUP2D_TRY(launch_init_residual(c, x, b, &GP, x0_zero));
  }
  CUP2D_TRY(finish(c, GP, 1, 1, 0, false));
  // p, nu and the accumulated correction y start at zero (cuda.cu:436-437): the first AB sweep and the first sweep E
  // know (k_fused: fresh, k_sweepE_y: first) and do not read them; y's first buffer stays "the best iterate" only as
  // long as no iterate has beaten the initial guess (KrylovScalars::best_is_x0) and is not read then either -- no fills
  // in-library communicator: the ghost blocks travel on the compute stream and land in place (comm.hip comm_exchange_blocks)
  const bool direct = gb && comm_blocks_direct(c);
  if (gb) {
    if (direct) CUP2D_TRY(comm_exchange_blocks(c, 1, c->d_r, nullptr, nullptr));
    else CUP2D_TRY(exchange_halo(c, c->d_r, 1, BS));  // width 8 = the whole block
  }

  constexpr int AHEAD = 4;
  static_assert(AHEAD <= cup2d_ctx::SOLVE_AHEAD, "event ring of the context");
  // Finish in the kernel (merge 1 | 2): the host looks at the solve once per GROUP of iterations -- the event record and
  // the system-scope store of the status word cost the stream 6 us of idle time between sweep E and the next AB
  // (tools/kernel_gaps.py), per iteration in round 2.  The last sweep E of a group reports (also when the
  // solve ended earlier in the group and its kernels returned at once); the host stays at most AHEAD groups in front, so at
  // most AHEAD * GROUP iterations of early-returning kernels are wasted behind a solve that has ended.
  constexpr int GROUP_ENV = 4;
  // merge 2 (N ranks): the status word is written by the one-wave kernel behind the last all-gather of an iteration
  // (comm.hip k_gather_scalars; callbacks: k_scalars), which reports for a finished solve as well -- same grouping.  Every
  // rank sees the same scalars and looks at the same group boundaries, so all ranks enqueue the same collectives.
  // N ranks (merge 2): every iteration enqueued behind a finished solve still packs, sends and unpacks its ghost blocks and
  // issues its all-gathers (the exchange path is host-enqueued and does not read the status word), so there the host looks
  // once per iteration (CUP2D_SOLVE_GROUP_N, default 1): at most AHEAD - 1 dead iterations, against GROUP * AHEAD_G - 1 -- the
  // 6 us of idle stream the grouping saves are nothing next to a collective.
  static const int GROUP_N_ENV = [] { const char *e = getenv("CUP2D_SOLVE_GROUP_N"); return e ? atoi(e) : 1; }();
  const int GROUP = merge == 1 ? (GROUP_ENV < 1 ? 1 : GROUP_ENV) : merge == 2 ? (GROUP_N_ENV < 1 ? 1 : GROUP_N_ENV) : 1;
  const int AHEAD_G = GROUP > 1 ? (AHEAD + 1) / 2 : AHEAD;  // groups the host may run ahead
  for (int i = 0; i < AHEAD; i++) c->h_status[i] = 0;
  const bool eab = eab_form(c, merge, dbg, gb);
  {  // cup2d_get_last_solver_form
    const bool edge = !eab && edge_form(c, gb, dbg);
    c->last_form = eab ? CUP2D_FORM_EAB : edge ? CUP2D_FORM_EDGE : CUP2D_FORM_FULL;
    c->last_merge = merge;
    c->last_handover = 0;
    if (eab && c->mat.active) c->last_handover = 0;
    else if (eab) c->last_handover = ((edge_share_mode(c, 0) & 1) ? 1 : 0) | ((edge_share_mode(c, 2) & 1) ? 4 : 0) | ((edge_share_mode(c, 3) & 1) ? 8 : 0);
    else if (edge) c->last_handover = ((edge_share_mode(c, 0) & 1) ? 1 : 0) | ((edge_share_mode(c, 1) & 1) ? 2 : 0);
  }
  if (eab) {
    // A+B of iteration 0, then per iteration TWO launches: C+D with the sums the next beginning needs (MODE 3), and sweep E
    // with the next A+B (MODE 2).  p, nu and r alternate between two buffers (r': ring entries re-read r of other tiles; the
    // second one is the s vector this organisation never stores)
    double *P[2] = {c->d_p2, c->d_p}, *N[2] = {c->d_nu2, c->d_nu}, *R[2] = {c->d_r, c->d_s};
    constexpr int zigzag = 1;  // C+D' walks the tiles in descending order (krylov_edge.h "Direction": -1.5 % of a step, round 3)
    // N ranks, opt-in (CUP2D_SWEEP_SPLIT=1): sweep the halo set first and the inner blocks while its ghost blocks travel on
    // the communication stream -- computeA's split (main.cpp:3035-3057) for the Krylov sweeps.  The halo set is a whole number
    // of tiles when the host orders it in patches (grid.py); the inner blocks never read a ghost block, so the blocks arriving
    // in place disturb nothing they touch.  Measured on a patch that is its own W and E neighbour (4096^2 cells, bytes through
    // RCCL on one GPU; tools/gpu_calls/gpu_r04_call8.sh, profiles/r04_nrank_timeline_split.txt): the transfer disappears
    // behind the inner sweep (13 / 29 us), but the two launches of a sweep take 127 / 247 us where one takes 123 / 234 (a
    // short first launch on part of the chip, the transfer's kernel beside the second) and two more launch boundaries cost
    // 12 us: 23.3 ms per step against 21.9 in one launch with the exchange behind it, 14.1 against 13.5 on a 4096 x 2048
    // patch.  Off by default; what it would gain with a slower link than a copy on one GPU is what an N-GPU run has to show.
    static const bool split_on = [] { const char *e = getenv("CUP2D_SWEEP_SPLIT"); return e && atoi(e) != 0; }();
    const bool split_asked = (c->org_split < 0 ? split_on : c->org_split != 0) && merge == 2 && gb;
    const bool split_here = split_asked && c->n_inner > 0 && c->n_inner < nb && c->n_inner % TB == 0;
    const bool ghost_local = ghost_local_enabled() && merge == 2 && gb;  // r' and p'' of the ghost blocks formed here, nu'' travels (k_ghost_rp)
    // N ranks with the in-library communicator, "deferred" (the default there): per reduction point ONE pack launch and ONE RCCL
    // kernel -- the rank's reduction record travels to every rank inside the ncclGroup that carries the ghost blocks (no
    // all-gather), and the scalar update happens in the prologue of the sweep that consumes it (k_edge MERGE 3: no one-wave
    // kernel) -- where round 4 had pack, send/recv, all-gather and the scalar kernel (profiles/r04_nrank_timeline.txt: 77 us of
    // an iteration outside the sweeps).  The state alternates between two records (a launch reads one and writes the other).
    // The host learns of the end of the solve one launch later than before (C+D' of iteration k reports the state after
    // iteration k - 1).  Agreed over all ranks at cup2d_comm_init (comm_defer_ok); CUP2D_DEFER_SCALARS=0 keeps round 4's form.
    const bool defer_any = merge == 2 && direct && ghost_local && comm_defer_ok(c) && c->org_defer != 0;
    // (with the deferred update the split and the unsplit sweep are two wire protocols: the choice is the one all ranks agreed
    // on at cup2d_comm_init, not this rank's own cut; without it both forms look the same to the peers)
    const bool split = defer_any ? (split_asked && comm_split_ok(c)) : split_here;
    const bool defer = defer_any && !split;
    // ... and with SPLIT sweeps ("overlap": cup2d_set_nrank_organisation(ctx, 1, 1)): the halo-set launch of the consumer sweep
    // runs the pending update in its prologue (MERGE 4: it finishes nothing), its ghost blocks travel on the communication
    // stream while the inner launch runs (MERGE 3 with nothing pending: it sums the partials of both launches), and the records
    // go round in ONE all-gather behind it -- per reduction point one small collective is exposed, the block transfer is not
    const bool overlap = defer_any && split;
    // (split sweeps: they leave one CU per XCD free, so that the send/recv kernel on the communication stream starts beside the
    // inner launch at once instead of waiting for a workgroup of the persistent grid to retire.  Measured to self on the 512 x 512
    // four-sided patch, tools/gpu_calls/gpu_r06_call1.sh: 24.8 -> 23.8 ms per step; the serial organisation: 22.5)
    c->spare_cus = split ? 8 : 0;  // (round 4's split sweeps too: the two split organisations take their partial sums over the same grids)
    KrylovScalars *S[2] = {c->d_sc, c->d_sc2};
    int sq = 0, enqueued = 0;
    if (defer) c->last_merge = 3;
    if (overlap) c->last_merge = 4;
    {
      ProfScope prof(c, CUP2D_T_SWEEP_A);
      c->prof_sample = c->prof_outer;
      FusedArgs a = {};
      a.in0 = c->d_p; a.in1 = c->d_nu; a.in2 = c->d_r; a.w = c->d_rhat; a.vout = P[0]; a.yout = N[0];
      if (c->mat.active) CUP2D_TRY(fused_sweep<0>(c, a, merge, dbg, &GP, gb));  // (hybrid operator: k_fused HYB + its rows)
      else CUP2D_TRY(eab_sweep<0>(c, a, merge));
    }
    if (merge == 2) {  // N ranks: the ghost blocks of nu' and p' in flight behind the reduction
      if (direct) CUP2D_TRY(comm_exchange_blocks(c, 2, N[0], P[0], nullptr));
      else if (gb) CUP2D_TRY(exchange_begin_blocks2(c, N[0], P[0]));
      CUP2D_TRY(finish_local(c, 1, 0, 1));
      if (gb && !direct) CUP2D_TRY(exchange_end_blocks2(c, N[0], P[0]));
    }
    // (exactly max_iter iterations are enqueued: the scalars of the last one set the status to 'cap reached', and what the
    // host would enqueue behind it could only return at once -- on N ranks with its exchanges and all-gathers still issued)
    for (int k = 0; k < max_iter; k++) {
      const int grp = k / GROUP, slot = grp % AHEAD_G;
      const bool first_of_group = k % GROUP == 0, last_of_group = k % GROUP == GROUP - 1;
      if (first_of_group && grp >= AHEAD_G) {
        CUP2D_HIP_CHECK(hipEventSynchronize(c->solve_ev[slot]));
        if (*(volatile int *)&c->h_status[slot] != 0) break;
      }
      c->prof_sample = (k % 32 == 0) && k < max_iter;
      const int o = k & 1, n = o ^ 1;
      enqueued = k + 1;
      if (overlap) {
        const int nh = nb - c->n_inner;
        int gh = 0;
        {  // C+D': halo set (pending: stage 4 of the previous iteration), t of the ghost blocks on its way, inner blocks, records
          FusedArgs a = {};
          a.in0 = R[o]; a.in1 = N[o]; a.w = c->d_rhat; a.yout = c->d_t; a.rev = zigzag;
          a.pg = comm_gathered(c); a.pn = comm_nranks(c); a.pstage = k == 0 ? -1 : 4; a.pnsum = 2; a.pmax = 1;
          a.sc_out = S[sq ^ 1];
          a.host_status = last_of_group ? &c->h_status[slot] : nullptr;
          { ProfScope prof(c, CUP2D_T_SWEEP_C); CUP2D_TRY(eab_sweep<3>(c, a, 4, c->n_inner, nh, 0, &gh, S[sq])); }
          sq ^= 1;
          CUP2D_TRY(comm_exchange_blocks(c, 1, c->d_t, nullptr, nullptr, true));
          a.pstage = -1; a.sc_out = nullptr; a.host_status = nullptr;
          { ProfScope prof(c, CUP2D_T_SWEEP_C); CUP2D_TRY(eab_sweep<3>(c, a, 3, 0, c->n_inner, gh, nullptr, S[sq])); }
          if (comm_gather_records(c) != 0) return CUP2D_ERR_COMM;
          CUP2D_TRY(comm_blocks_wait(c));
        }
        {  // E+A+B: halo set (pending: stage 5), nu'' of the ghost blocks on its way (r', p'' of them formed here), inner blocks
          FusedArgs a = {};
          a.in0 = P[o]; a.in1 = N[o]; a.in2 = R[o]; a.w = c->d_rhat; a.vout = P[n]; a.yout = N[n];
          a.t = c->d_t; a.y0 = c->d_y; a.y1 = c->d_yopt; a.y2 = c->d_xopt; a.rout = R[n];
          a.pg = comm_gathered(c); a.pn = comm_nranks(c); a.pstage = 5; a.pnsum = 5; a.pmax = 0;
          a.sc_out = S[sq ^ 1];
          { ProfScope prof(c, CUP2D_T_SWEEP_EA); CUP2D_TRY(eab_sweep<2>(c, a, 4, c->n_inner, nh, 0, &gh, S[sq])); }
          sq ^= 1;
          const GhostRP G = {P[o], N[o], R[o], c->d_t, R[n], P[n], S[sq], (size_t)nb * BC, (size_t)c->nghost * BC};
          CUP2D_TRY(comm_exchange_blocks(c, 1, N[n], nullptr, nullptr, true, &G));
          a.pstage = -1; a.sc_out = nullptr;
          { ProfScope prof(c, CUP2D_T_SWEEP_EA); CUP2D_TRY(eab_sweep<2>(c, a, 3, 0, c->n_inner, gh, nullptr, S[sq])); }
          if (comm_gather_records(c) != 0) return CUP2D_ERR_COMM;
          CUP2D_TRY(comm_blocks_wait(c));
        }
        if (last_of_group) CUP2D_HIP_CHECK(hipEventRecord(c->solve_ev[slot], c->stream));
        continue;
      }
      if (defer) {
        {
          FusedArgs a = {};
          a.in0 = R[o]; a.in1 = N[o]; a.w = c->d_rhat; a.yout = c->d_t; a.rev = zigzag;
          a.pg = comm_gathered(c); a.pn = comm_nranks(c); a.pstage = k == 0 ? -1 : 4; a.pnsum = 2; a.pmax = 1;  // {rhat.nu'', r'.r', max|r'|}
          a.sc_out = S[sq ^ 1];
          a.host_status = last_of_group ? &c->h_status[slot] : nullptr;
          ProfScope prof(c, CUP2D_T_SWEEP_C);
          CUP2D_TRY(eab_sweep<3>(c, a, 3, 0, nb, 0, nullptr, S[sq]));
          sq ^= 1;
        }
        CUP2D_TRY(comm_exchange_blocks(c, 1, c->d_t, nullptr, nullptr, false, nullptr, true));  // t of the ghost blocks + the five sums
        {
          FusedArgs a = {};
          a.in0 = P[o]; a.in1 = N[o]; a.in2 = R[o]; a.w = c->d_rhat; a.vout = P[n]; a.yout = N[n];
          a.t = c->d_t; a.y0 = c->d_y; a.y1 = c->d_yopt; a.y2 = c->d_xopt; a.rout = R[n];
          a.pg = comm_gathered(c); a.pn = comm_nranks(c); a.pstage = 5; a.pnsum = 5; a.pmax = 0;
          a.sc_out = S[sq ^ 1];
          ProfScope prof(c, CUP2D_T_SWEEP_EA);
          CUP2D_TRY(eab_sweep<2>(c, a, 3, 0, nb, 0, nullptr, S[sq]));
          sq ^= 1;
        }
        // nu'' of the ghost blocks + {rhat.nu'', r'.r', max|r'|}; r' and p'' of the ghost blocks formed here with the scalars that
        // launch used (the record it wrote)
        const GhostRP G = {P[o], N[o], R[o], c->d_t, R[n], P[n], S[sq], (size_t)nb * BC, (size_t)c->nghost * BC};
        CUP2D_TRY(comm_exchange_blocks(c, 1, N[n], nullptr, nullptr, false, &G, true));
        if (last_of_group) CUP2D_HIP_CHECK(hipEventRecord(c->solve_ev[slot], c->stream));
        continue;
      }
      {
        FusedArgs a = {};
        a.in0 = R[o]; a.in1 = N[o]; a.w = c->d_rhat; a.yout = c->d_t;
        a.rev = zigzag;  // (A+B of iteration 0 and MODE 2 ascend: this one starts where they end, and ends where MODE 2 starts)
        if (split) {
          // N ranks, computeA's split (main.cpp:3035-3057) for the sweep: the halo set first, its t on the way to the
          // neighbours while the inner blocks are swept; the second launch finishes the reduction of both
          int gh = 0;
          { ProfScope prof(c, CUP2D_T_SWEEP_C); CUP2D_TRY(eab_sweep<3>(c, a, 0, c->n_inner, nb - c->n_inner, 0, &gh)); }
          if (direct) CUP2D_TRY(comm_exchange_blocks(c, 1, c->d_t, nullptr, nullptr, true));
          else CUP2D_TRY(exchange_begin(c, c->d_t, 1, BS));
          { ProfScope prof(c, CUP2D_T_SWEEP_C); CUP2D_TRY(eab_sweep<3>(c, a, 2, 0, c->n_inner, gh, nullptr)); }
        } else {
          ProfScope prof(c, CUP2D_T_SWEEP_C);
          if (c->mat.active) CUP2D_TRY(hyb_eab_sweep<3>(c, a, nullptr));
          else CUP2D_TRY(eab_sweep<3>(c, a, merge));
        }
      }
      if (merge == 2) {  // MODE 2 recomputes r' of the blocks around a tile: it needs t in the ghost blocks
        if (!split) {
          if (direct) CUP2D_TRY(comm_exchange_blocks(c, 1, c->d_t, nullptr, nullptr));
          else if (gb) CUP2D_TRY(exchange_begin(c, c->d_t, 1, BS));
        }
        CUP2D_TRY(finish_local(c, 5, 0, 5));
        if (split && direct) CUP2D_TRY(comm_blocks_wait(c));
        else if (gb && !direct) CUP2D_TRY(exchange_end(c, c->d_t, 1, BS));
      }
      int *const report = last_of_group ? &c->h_status[slot] : nullptr;
      {
        FusedArgs a = {};
        a.in0 = P[o]; a.in1 = N[o]; a.in2 = R[o]; a.w = c->d_rhat; a.vout = P[n]; a.yout = N[n];
        a.t = c->d_t; a.y0 = c->d_y; a.y1 = c->d_yopt; a.y2 = c->d_xopt; a.rout = R[n];
        a.host_status = merge == 1 ? report : nullptr;
        if (split) {
          int gh = 0;
          { ProfScope prof(c, CUP2D_T_SWEEP_EA); CUP2D_TRY(eab_sweep<2>(c, a, 0, c->n_inner, nb - c->n_inner, 0, &gh)); }
          if (ghost_local) {
            const GhostRP G = {P[o], N[o], R[o], c->d_t, R[n], P[n], c->d_sc, (size_t)nb * BC, (size_t)c->nghost * BC};
            if (direct) CUP2D_TRY(comm_exchange_blocks(c, 1, N[n], nullptr, nullptr, true, &G));
            else {
              CUP2D_TRY(ghost_rp(c, P[o], N[o], R[o], c->d_t, R[n], P[n]));
              CUP2D_TRY(exchange_begin(c, N[n], 1, BS));
            }
          } else if (direct) CUP2D_TRY(comm_exchange_blocks(c, 3, R[n], P[n], N[n], true));
          else CUP2D_TRY(exchange_begin_blocks3(c, R[n], P[n], N[n]));
          { ProfScope prof(c, CUP2D_T_SWEEP_EA); CUP2D_TRY(eab_sweep<2>(c, a, 2, 0, c->n_inner, gh, nullptr)); }
        } else {
          ProfScope prof(c, CUP2D_T_SWEEP_EA);
          if (c->mat.active) CUP2D_TRY(hyb_eab_sweep<2>(c, a, report));
          else CUP2D_TRY(eab_sweep<2>(c, a, merge));
        }
      }
      if (merge == 2) {
        if (!split && gb) {
          if (ghost_local) {
            const GhostRP G = {P[o], N[o], R[o], c->d_t, R[n], P[n], c->d_sc, (size_t)nb * BC, (size_t)c->nghost * BC};
            if (direct) CUP2D_TRY(comm_exchange_blocks(c, 1, N[n], nullptr, nullptr, false, &G));
            else {
              CUP2D_TRY(ghost_rp(c, P[o], N[o], R[o], c->d_t, R[n], P[n]));
              CUP2D_TRY(exchange_begin(c, N[n], 1, BS));
            }
          } else if (direct) CUP2D_TRY(comm_exchange_blocks(c, 3, R[n], P[n], N[n]));
          else CUP2D_TRY(exchange_begin_blocks3(c, R[n], P[n], N[n]));
        }
        CUP2D_TRY(finish_local(c, 2, 1, 4, report));
        if (split && direct) CUP2D_TRY(comm_blocks_wait(c));
        else if (gb && !direct) {
          if (ghost_local) CUP2D_TRY(exchange_end(c, N[n], 1, BS));
          else CUP2D_TRY(exchange_end_blocks3(c, R[n], P[n], N[n]));
        }
      }
      if (last_of_group) CUP2D_HIP_CHECK(hipEventRecord(c->solve_ev[slot], c->stream));
    }
    if ((defer || overlap) && enqueued > 0) {
      // the last pending update (stage 4 of the last iteration enqueued; a no-op behind a solve that has ended) in a launch of its
      // own, on the context's first record: what follows (the last pass over x, the host's copy of the scalars) reads it there
      if (sq == 1) CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_sc, c->d_sc2, sizeof(KrylovScalars), hipMemcpyDeviceToDevice, c->stream));
      ProfScope prof(c, CUP2D_T_SCALARS);
      if (comm_apply_gathered(c, 2, 1, 4) != 0) return CUP2D_ERR_COMM;
    }
  }
  for (int k = 0; !eab && k < max_iter; k++) {
    const int grp = k / GROUP, slot = grp % AHEAD_G;
    const bool first_of_group = k % GROUP == 0, last_of_group = k % GROUP == GROUP - 1;
    if (first_of_group && grp >= AHEAD_G) {
      CUP2D_HIP_CHECK(hipEventSynchronize(c->solve_ev[slot]));
      if (*(volatile int *)&c->h_status[slot] != 0) break;
    }
    int *const report = last_of_group ? &c->h_status[slot] : nullptr;
    c->prof_sample = (k % 32 == 0) && k < max_iter;
    double *p_in = (k & 1) ? c->d_p2 : c->d_p, *p_out = (k & 1) ? c->d_p : c->d_p2;
    double *nu_in = (k & 1) ? c->d_nu2 : c->d_nu, *nu_out = (k & 1) ? c->d_nu : c->d_nu2;
    {
      ProfScope prof(c, CUP2D_T_SWEEP_A);
      FusedArgs a = {};
      a.in0 = p_in; a.in1 = nu_in; a.in2 = c->d_r; a.w = c->d_rhat; a.vout = p_out; a.yout = nu_out;
      CUP2D_TRY(fused_sweep<0>(c, a, merge, dbg, &GP, gb));
    }
    // CD needs the ghost nu', the next AB the ghost p': one message, in flight behind the reduction
    if (direct) CUP2D_TRY(comm_exchange_blocks(c, 2, nu_out, p_out, nullptr));
    else if (gb) CUP2D_TRY(exchange_begin_blocks2(c, nu_out, p_out));
    if (merge == 0) CUP2D_TRY(finish(c, GP, 1, 0, 1, true));
    if (merge == 2) CUP2D_TRY(finish_local(c, 1, 0, 1));
    if (gb && !direct) CUP2D_TRY(exchange_end_blocks2(c, nu_out, p_out));
    {
      ProfScope prof(c, CUP2D_T_SWEEP_C);
      FusedArgs a = {};
      a.in0 = c->d_r; a.in1 = nu_out; a.vout = c->d_s; a.yout = c->d_t;
      CUP2D_TRY(fused_sweep<1>(c, a, merge, dbg, &GP, gb));
    }
    if (merge == 0) CUP2D_TRY(finish(c, GP, 2, 0, 2, true));
    if (merge == 2) CUP2D_TRY(finish_local(c, 2, 0, 2));
    {
      ProfScope prof(c, CUP2D_T_SWEEP_E);
      const auto launchE = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(gridE), dim3(WG), 0, c->stream, (double2 *)c->d_y, (double2 *)c->d_yopt,
                           (double2 *)c->d_xopt, (const double2 *)p_out, (double2 *)c->d_r, (const double2 *)nu_out, (const double2 *)c->d_t,
                           (const double2 *)c->d_rhat, c->d_sc, c->d_partials, n / 2, c->d_red, c->d_ticket, report);
      };
      if (merge == 1) launchE(k_sweepE_y<1>);
      else if (merge == 2) launchE(k_sweepE_y<2>);
      else launchE(k_sweepE_y<0>);
    }
    CUP2D_HIP_CHECK(hipGetLastError());
    if (direct) CUP2D_TRY(comm_exchange_blocks(c, 1, c->d_r, nullptr, nullptr));  // the new r
    else if (gb) CUP2D_TRY(exchange_begin(c, c->d_r, 1, BS));  // ... in flight behind the reduction of E
    if (merge == 0) CUP2D_TRY(finish(c, gridE, 2, 1, 3, true, report));
    if (merge == 2) CUP2D_TRY(finish_local(c, 2, 1, 3, report));
    if (gb && !direct) CUP2D_TRY(exchange_end(c, c->d_r, 1, BS));
    if (last_of_group) CUP2D_HIP_CHECK(hipEventRecord(c->solve_ev[slot], c->stream));
  }
  c->prof_sample = c->prof_outer;
  c->spare_cus = 0;
  // cuda.cu:546-547: return x_opt = x0 + P_inv y_opt.  Which buffer holds y_opt is in the scalars: the launch reads it there,
  // so it is enqueued behind the last iteration without the host having seen the solve end (one wait per solve instead of two;
  // a caller that asked for the last iterate takes the host's route below).  Enqueued BEFORE the copies of the scalars: a
  // copy into pageable memory (edge_fault) may hold the host until it has happened
  const double *ybuf[3] = {c->d_y, c->d_yopt, c->d_xopt};  // k_sweepE_y's three buffers
  bool x_done = false;
  if (!c->keep_last) {
    ProfScope prof(c, CUP2D_T_FINAL_X);
    x_done = launch_final_x_on_device(c, ybuf[0], ybuf[1], ybuf[2], x, x0_zero);
    CUP2D_HIP_CHECK(hipGetLastError());
  }
  if (x_done && c->solve_tail) {  // cup2d_step: the projection, enqueued while the solve's last launches still run
    c->solve_tail_ran = true;
    CUP2D_TRY(c->solve_tail(c, c->solve_tail_arg));
  }
  CUP2D_HIP_CHECK(hipMemcpyAsync(c->h_sc, c->d_sc, sizeof init, hipMemcpyDeviceToHost, c->stream));
  int edge_fault = 0;
  CUP2D_HIP_CHECK(hipMemcpyAsync(&edge_fault, c->d_fault, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (edge_fault) {  // k_edge: a wave gave up waiting for a sibling's export (never seen; the result would be wrong)
    CUP2D_HIP_CHECK(hipMemsetAsync(c->d_fault, 0, sizeof(int), c->stream));
    set_error("poisson_solve: the edge-form sweep lost a hand-over between sibling waves (CUP2D_EDGE_SHARE=0 avoids the path)");
    return CUP2D_ERR_HIP;
  }
  const double *ybest = ybuf[c->h_sc->ybest];
  c->have_last = false;
  if (c->keep_last) {  // the last iterate x0 + P_inv y, for cup2d_solver_last_iterate (before x0 in PRES is overwritten)
    const double *ylast = ybuf[c->h_sc->ycur];
    if (c->h_sc->iter == 0) {
      if (x0_zero) CUP2D_TRY(launch_zero(c, c->d_z, n));
      else CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_z, x, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else if (x0_zero) {
      CUP2D_TRY(launch_precond(c, ylast, c->d_z, 0, nb));
    } else {
      CUP2D_HIP_CHECK(hipMemcpyAsync(c->d_z, x, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      CUP2D_TRY(launch_precond_add(c, ylast, c->d_z, c->d_s));
    }
    c->have_last = true;
  }
  if (!x_done) {
    ProfScope prof(c, CUP2D_T_FINAL_X);
    if (c->h_sc->best_is_x0) {  // the initial guess is the answer: x = x0
      if (x0_zero) CUP2D_TRY(launch_zero(c, x, n));
    } else if (x0_zero) {
      CUP2D_TRY(launch_precond(c, ybest, x, 0, nb));  // x = 0 + P_inv y_opt, written, not added
    } else {
      CUP2D_TRY(launch_precond_add(c, ybest, x, c->d_s));
    }
    CUP2D_HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  if (iters) *iters = c->h_sc->iter;
  if (restarts) *restarts = c->h_sc->restarts;
  if (linf) *linf = c->h_sc->err_opt;
  if (linf_init) *linf_init = c->h_sc->err_init;
  return CUP2D_OK;
}

}  // namespace cup2d
