// krylov_edge.h -- the EDGE form of the tile-fused BiCGSTAB sweeps (included by krylov_fused.hip, which holds the full form).
//
// Reference: BiCGSTABSolver::main cuda.cu:478-509 -- nu = A (P_inv p), t = A (P_inv s) -- with P_inv the block-Jacobi
// matrix main.cpp:6451-6488 builds: P_inv = -(A_loc)^-1, A_loc = 4 on the diagonal, -1 between in-block neighbours.
//
// Identity.  Write the Poisson matrix of a same-level grid (main.cpp:7034-7112) as A = L + C + W: L the block-diagonal part
// with -4 on EVERY diagonal (L = -A_loc, blockwise), C the +1 couplings between edge cells of neighbouring blocks, W the +1
// on the diagonal of a cell per domain wall it touches (Neumann: the ghost is the edge cell, main.cpp:3210-3255).  L P_inv =
// I, so for any v
//     A P_inv v = v + (C + W) z,   z = P_inv v:
// the operator is the IDENTITY on the 36 interior cells of a block, and on its 28 edge cells it adds ghost values that are
// edge cells of z -- the own block's at a wall, the neighbour block's otherwise.  Only the 32 edge entries (4 sides x 8) of
// z are ever needed: the dense 64 x 64 product per block shrinks to 64 x 32 (half the MFMAs), the 5-point stencil over the
// tile disappears, and the result differs from the full form by the round-off of L (P_inv v) - v, a few 1e-15 |v|
// (tests/test_solver_variants_gpu.py pins both forms to the CPU restatement of the reference and to each other).  Needs the built-in P_inv
// (!custom_Pinv); any other preconditioner takes the full form.
//
// Sharing.  A tile still needs the z edges of the blocks around it.  The full form recomputes every one of them from the
// neighbour's inputs (16 re-read blocks + a job on the matrix cores per 16-block tile; the re-reads are what keeps its HBM
// traffic at 1.3x the algorithmic bytes).  Here the 8 waves of a workgroup work on 8 CONSECUTIVE tiles in the same round
// (in the reference's Hilbert order an aligned run of 128 blocks: a 16 x 8 patch), every wave EXPORTS the z edges of its
// tile's perimeter sides to LDS, and a wave whose neighbour block belongs to a sibling's tile takes the edge from there:
// 48 of the 128 perimeter sides of the 8 tiles are left to recompute (-62 % of the re-reads).  No workgroup barrier: per-wave
// round counters in LDS (pub[w] = rounds wave w has published), a consumer waits for the one sibling it needs, a producer
// for every sibling to be at most NXB - 1 rounds behind before it reuses an export buffer (NXB = 2 buffers, round mod NXB).
#pragma once
// (a fragment of krylov_fused.hip: included INSIDE its namespace cup2d, after FusedArgs, ld2 / st2, ring_precond and
// fused_reduce_store)

#ifndef CUP2D_EDGE_NXB
#define CUP2D_EDGE_NXB 3
#endif
constexpr int NXB = CUP2D_EDGE_NXB;  // export buffers per wave: a wave may run NXB - 1 rounds ahead of its slowest sibling
constexpr int EXP_SLOTS = 16;  // perimeter sides a tile may export (host check edge_share_ok: every tile has <= 16)
constexpr int PE2_DOUBLES = 16 * 2 * 64;  // the 32 edge columns of P_inv as MFMA B fragments: [k-step][n-tile][lane]
constexpr int EDGE_HDR_DOUBLES = PE2_DOUBLES + 16;  // + the 8 round counters (padded to 128 bytes)
struct alignas(16) EdgeLds {
  double S[TB * XS];            // v of 16 blocks (A operand of the job), then the z edges of those blocks: S[b*XS + 8*side + q]
  double GE[TB * 4 * GS];       // ghost edges of the tile's blocks: [block][W,E,S,N][q]
  double X[NXB][EXP_SLOTS * BS];  // exported z edges of this tile's perimeter sides, in slot order; [round mod NXB]
  unsigned long long xmask[NXB];  // which (block, side) slots X holds: slot of bit i = popcount of the bits below i
  int ring_nb[TB * 4];          // neighbour block of ring entry e (to recompute) ...
  int ring_dst[TB * 4];         // ... and the slot (block * 4 + side) it feeds
};
constexpr size_t EDGE_LDS_BYTES = EDGE_HDR_DOUBLES * sizeof(double) + FWAVES * sizeof(EdgeLds);
static_assert(EDGE_LDS_BYTES <= 156 * 1024, "k_edge: LDS budget (160 KiB per workgroup, some of it static)");
constexpr int EDGE_SPIN_LIMIT = 1 << 22;  // a wave that waits longer than this for a sibling reports a fault instead of hanging

// timing-only builds (wrong results): bit 0: no MFMAs in the jobs; bit 1: no ring (no ring loads, no ring jobs); bit 2: no LDS
// phases at all (no staging, classification, jobs, gathers: y = v -- the memory skeleton of the tile loads and stores);
// bit 3: no stores
#ifndef CUP2D_EDGE_KNOCK
#define CUP2D_EDGE_KNOCK 0
#endif
constexpr int KNOCK = CUP2D_EDGE_KNOCK;
// cache policy of the streams MODE 2 / 3 add (non-temporal where the bit is set; CUP2D_POLICY in krylov_fused.hip has the
// others): 0x01 / 0x02 / 0x04 MODE 2's tile loads of p', nu', r; 0x08 MODE 3's rhat; 0x10 / 0x20 the stores of r', nu'';
// 0x40 / 0x80 MODE 3's tile loads of r, nu'
// Measured at 4096^2 (tools/gpu_calls/gpu_r03_call19.sh; step of 50 iterations, MODE 3 / MODE 2 in us):
//   0x00  19.9 ms  125 / 247      0x01  19.65  125 / 243      0x07  19.18  125 / 231  <- default
//   0x08  19.9     117 / 256      0x30  20.0   123 / 252      0xC0  20.3   125 / 256
#ifndef CUP2D_POLICY2
#define CUP2D_POLICY2 0x07
#endif
constexpr unsigned POL2 = CUP2D_POLICY2;
// C+D (MODE 1 / 3): halves of the NEXT tile's own batches requested behind this tile's job, next to its ring pass 0 -- they are
// then in flight across the hand-over, the gathers, the epilogue and the next ring job instead of across the ring job alone
// (MODE 3 holds 173 registers of 256: one half = 48 more).  0: as before (requested behind the ring's staging)
#ifndef CUP2D_CD_AHEAD
#define CUP2D_CD_AHEAD 1
#endif
// set bits of a wave mask below this lane (v_mbcnt: no 64-bit lane mask held in registers)
static __device__ __forceinline__ int bits_below_lane(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// S (v of 16 blocks, block-major) -> S[b * XS + 8 * side + q] = z of block b on the edge cells of side W, E, S, N: the product
// with the 32 edge columns of P_inv (krylov_fused.hip ring_precond; the A operands in two halves of eight k-steps: 16
// registers instead of 32 -- S is not written before the last MFMA has its operands)
static __device__ __forceinline__ void edge_precond(double *S, const double *PE, int lane) {
  const int ablk = lane & 15, akk = lane >> 4;
  v4f64 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; nt++) acc[nt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int h = 0; h < 2 && !(KNOCK & 1); h++) {
    double xa[8];
#pragma unroll
    for (int k = 0; k < 8; k++) xa[k] = S[ablk * XS + 4 * (8 * h + k) + akk];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
        acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[k], PE[((8 * h + k) * 2 + nt) * 64 + lane], acc[nt], 0, 0, 0);
  }
  wave_lds_sync();  // every lane has read its operands before the tile is overwritten
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) S[(akk + 4 * v) * XS + 16 * nt + ablk] = acc[nt][v];
  wave_lds_sync();
}

// HYB: S (v of 16 blocks, block-major) -> S (z = P_inv v of those blocks, block-major): the FULL 64 x 64 product of k_fused's
// tile_precond -- same fragment maps, same k order: the same bits -- for the tiles of the hybrid operator whose z somebody reads
// from memory (the general tiles and the blocks their rows reach: a quarter of the tiles of an adapted grid).  This kernel's LDS
// holds the edge columns only (the staging tiles fill the rest), so the B fragments come from P_inv in memory: 32 KB that every
// such tile of every workgroup reads, i.e. from the L2.
static __device__ __forceinline__ void full_precond_mem(double *S, const double *__restrict__ Pinv, int lane) {
  const int ablk = lane & 15, akk = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; nt++) acc[nt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int ks = 0; ks < 16; ks++) {
    const double xa = S[ablk * XS + 4 * ks + akk];
    const double *pr = Pinv + (4 * ks + akk) * BC + ablk;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, pr[16 * nt], acc[nt], 0, 0, 0);
  }
  wave_lds_sync();  // every lane has read its operands before the tile is overwritten
#pragma unroll
  for (int v = 0; v < 4; v++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) S[(akk + 4 * v) * XS + 16 * nt + ablk] = acc[nt][v];
  wave_lds_sync();
}

// bit `lane` of a wave mask (no 1 << lane held in two registers across the loop)
static __device__ __forceinline__ bool mask_bit(unsigned long long m, int lane) {
  const unsigned half = lane < 32 ? (unsigned)m : (unsigned)(m >> 32);
  return ((half >> (lane & 31)) & 1u) != 0u;
}
// a copy of x the optimiser cannot see through: per-lane addresses formed from it stay inside the loop iteration (LICM would
// hoist dozens of loop-invariant LDS addresses out of the tile loop, where they are spilled; one add each to form them again)
static __device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// wave-uniform wait until sibling u has published `need` rounds
static __device__ __forceinline__ void edge_wait(int *pub, int u, int need, int *fault) {
  int spins = 0;
  while (__hip_atomic_load(pub + u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > EDGE_SPIN_LIMIT) {
      if ((threadIdx.x & 63) == 0) __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

// -DEDGE_PHASES (timing aid, tools/gpu_calls/gpu_r06_call43.sh): every wave adds up the shader-clock cycles it spends in the phases of a
// tile; waves 0 and 5 of the first two workgroups print their sums for iteration 5.  Round 6 (hand-over on in both launches, 4096^2),
// cycles per tile, wave 0 / wave 5 of a workgroup:
//   E+A+B  ring staging 2.5 k / 2.9 k, requests + ring job 8.8 k / 14.1 k, tile staging (+ wait for its batches) 12.2 k / 14.5 k,
//          tile job 3.5 k / 3.9 k, requests of the next tile 5.9 k / 6.6 k, export 2.4 k / 2.6 k, own edges + imports (+ wait for the
//          siblings) 13.9 k / 4.0 k, epilogue 6.4 k / 7.7 k: 55-56 k = 26 us -- the siblings pace each other, the slack of the
//          fastest shows up as its wait at the imports; the launch moves its bytes at the copy ceiling
//   C+D'   2.3 k, 3.7 k, 0.9 k (its first half is requested a tile ahead), 3.0 k, 2.5 k, 2.3 k, 5.8 k (wait), 5.7 k: 26 k = 12.5 us
//   (the export in FRONT of the next tile's requests -- the siblings get the edges earlier, the requests fill the wait -- measured: 937.5 /
//   944.3 against 945.3 / 943.6 Mcell-updates/s, nothing; not kept)
#ifdef EDGE_PHASES
#define EPH(k) { const long long now_ = clock64(); eph[k] += now_ - etprev; etprev = now_; }
#else
#define EPH(k) {}
#endif
// MODE 0 (sweeps A+B): v = p' = beta (p - omega nu) + r   (cuda.cu:478-483; restart: p' = rhat = r, 461-476)
//                      y = nu' = p' + ghosts(P_inv p') ; partial(rhat . nu')             (484-488)
// MODE 1 (sweeps C+D): v = s  = r - alpha nu'                                            (499-502)
//                      y = t  = s + ghosts(P_inv s)   ; partial(t . s, t . t)           (503-509)
// MODE 2 (sweep E of iteration k + sweeps A+B of iteration k + 1 in one pass over the cells; finish in the kernel):
//                      s = r - alpha nu' ; y' = y + alpha p' + omega s ; r' = s - omega t          (498-502, 520-525)
//                      v = p'' = beta' (p' - omega nu') + r'  (restart: p'' = rhat = r') ; y = nu'' = p'' + ghosts(P_inv p'')
//                      partial(rhat . nu'', r' . r', max|r'|)
//                      beta' and the restart decision need rho' = rhat . r' and ||r'||^2 BEFORE this launch: MODE 3 supplies
//                      them (krylov_common.h stage 5); r' goes to a second buffer (ring entries re-read r of other tiles)
// MODE 3 (MODE 1 + the dot products of the NEXT iteration's beginning): partial(t.s, t.t, rhat.s, rhat.t, s.s), so that
//                      rho' = rhat.(s - omega t) = rhat.s - omega rhat.t and ||r'||^2 = s.s - 2 omega t.s + omega^2 t.t
// Jobs, batches, load schedule and cache policy as in k_fused (krylov_fused.hip); what differs is the job's product (edge
// columns only, for the tile's own blocks too), where a ghost edge comes from, and the epilogue.
// HYB (MODE 2 / 3, one rank): the hybrid assembled operator of an adapted grid (ctx.h SellMatrix; nbr = its d_fnbr, count = its
// number of tiles, A.tile0 / A.zmask / A.zg).  The identity holds on every PLAIN block (same-level or wall sides: its rows are
// the 5-point rows), so a tile of plain blocks is swept exactly as on a uniform grid.  A block marked FUSED_GENERAL (the table
// marks every block of a tile that holds a slice with stored coarse-fine rows) gets everything that is a function of its own
// cells here -- y', r', p'' and their sums (MODE 2), the sums of s (MODE 3) --; its rows (nu'' or t and the dot products with
// them) are k_hyb_rows' in the launch behind this one, from z in memory: a tile with blocks those rows read (zmask) forms
// z = P_inv v of its blocks in full and stores theirs.  Two sweeps + two short rows launches per iteration where the full form (k_fused HYB) has three
// sweeps + two: 112 instead of 136 B/cell.  No sharing between sibling waves (the tiles are not aligned runs of 16).
template <int MODE, int MERGE, bool HYB = false>
__global__ __launch_bounds__(FWG, 1) void k_edge(FusedArgs A, const double *__restrict__ Pinv,
                                                 const int *__restrict__ nbr, KrylovScalars *sc, double *partials,
                                                 int first, int count, int poff, int share, double *red, unsigned *ticket,
                                                 int *fault) {
  extern __shared__ __attribute__((aligned(16))) double fsm[];
  constexpr bool AB = MODE == 0, CD = MODE == 1 || MODE == 3, CDX = MODE == 3, EAB = MODE == 2;
  // MERGE 3 (N ranks): the scalar update of the PREVIOUS reduction point happens here -- every thread of every workgroup (and
  // every rank) forms the same few dozen flops from the records gathered from all ranks, summed in rank order (what the
  // one-wave kernel behind an all-gather did, comm.hip k_gather_scalars: two launches per reduction point fewer).  The state
  // is read from sc and written -- by one thread of the launch -- to A.sc_out, ANOTHER record: nobody of this launch reads what
  // it writes.  A launch behind a finished solve hands the state on and returns.
  // MERGE 4: the same prologue on a launch that finishes nothing (the halo-set launch of a split sweep: the inner launch
  // behind it -- MERGE 3 with no stage pending and no sc_out -- sums the partials of both)
  constexpr bool PRO = MERGE >= 3;            // deferred scalar update in the prologue
  constexpr int FIN = MERGE == 4 ? 0 : MERGE; // what the epilogue does: 0 partials only, 1 finish + scalars, 2 / 3 this rank's sums
  KrylovScalars ST;
  if constexpr (PRO) {
    ST = *sc;
    if (ST.status == 0 && A.pstage >= 0) {
      double v[RED_REC];
      sum_records(A.pg, A.pn, A.pnsum, A.pmax, v);
      scalars_update(&ST, v, A.pstage);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (A.sc_out) *A.sc_out = ST;
      if (A.host_status) __hip_atomic_store(A.host_status, ST.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    sc = &ST;
  }
  if (sc->status != 0) {
    // (as k_sweepE_y: the last launch of a group of iterations reports to the host, also behind a solve that has ended)
    if (!PRO && EAB && A.host_status && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(A.host_status, sc->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  double *PE = fsm;
  int *pub = reinterpret_cast<int *>(fsm + PE2_DOUBLES);
  for (int idx = threadIdx.x; idx < PE2_DOUBLES; idx += FWG) {
    const int l = idx & 63, nt = (idx >> 6) & 1, ks = idx >> 7, col = 16 * nt + (l & 15);
    PE[idx] = Pinv[(4 * ks + (l >> 4)) * BC + edge_cell(col >> 3, col & 7)];
  }
  if (threadIdx.x < FWAVES) pub[threadIdx.x] = 0;
  __syncthreads();
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  EdgeLds *LL = reinterpret_cast<EdgeLds *>(fsm + EDGE_HDR_DOUBLES);
  EdgeLds &L = LL[wave];
  // pair layout of the 16-byte accesses: this lane holds cells c0 = 2 hl, c0 + 1 of block 2 i + hf
  const int hf = lane >> 5, hl = lane & 31, c0 = 2 * hl, px = c0 & 7, py = hl >> 2;
  const double c1 = CD ? -sc->alpha : -sc->omega;  // malpha | momega
  const double beta = sc->beta;
  const bool fresh = AB && sc->iter == 0;                     // p = nu = 0 (cuda.cu:436-437): not read
  const bool restart = (AB || EAB) && sc->restart_flag != 0;  // p' = rhat = r (cuda.cu:461-476)
  // sweep E's scalars and buffers (k_sweepE_y)
  const double alpha = sc->alpha, omega = sc->omega, malpha = -sc->alpha;
  const bool yfirst = EAB && sc->iter == 0;  // the accumulated correction starts at zero: not read
  const double2 *yin = nullptr;
  double2 *yout2 = nullptr;
  if (EAB) {
    const int cur = sc->ycur, out = y_out_buffer(cur, sc->ybest);
    yin = reinterpret_cast<const double2 *>(cur == 0 ? A.y0 : (cur == 1 ? A.y1 : A.y2));
    yout2 = reinterpret_cast<double2 *>(out == 0 ? A.y0 : (out == 1 ? A.y1 : A.y2));
  }
  constexpr int NDOT = AB ? 1 : CDX ? 5 : 2;
  double acc[NDOT], rmax[1] = {0.0};
#pragma unroll
  for (int i = 0; i < NDOT; i++) acc[i] = 0.0;

  // v at one cell -- the arithmetic of k_sweepA_fd / k_sweepC_fd (and, MODE 2, of k_sweepE_y in front of it), operation for
  // operation.  d: t (MODE 2); rn: where MODE 2 leaves r'
  const auto form_v = [&](double a, double b, double c, double d, double &rn) -> double {
    if (EAB) {
      const double sv = c + malpha * b;
      rn = sv + c1 * d;
      if (restart) return rn;
      double v = a + c1 * b;
      v = v * beta;
      return v + rn;
    }
    if (AB) {
      if (restart || fresh) return c;
      double v = a + c1 * b;
      v = v * beta;
      return v + c;
    }
    return a + c1 * b;
  };

  // tiles of 16 blocks; the 8 waves of a workgroup take 8 consecutive tiles per round, contiguous ranges per XCD
  const int ntiles = HYB ? count : (count + TB - 1) / TB;
  int t_begin, t_end, t_stride;
  {
    const int G = gridDim.x, w = blockIdx.x;
    if (G >= 8 && (G % 8) == 0) {
      // strided walk inside the XCD's contiguous range: the 32 workgroups of an XCD sweep 32 ADJACENT 16 x 8 patches at the same
      // time and meet in its L2 and in the memory-side cache (a contiguous piece per workgroup -- consecutive rounds as
      // neighbouring patches, their shared side taken from the previous round's exports -- was built in round 4 and is
      // slower: the walk costs 6-15 us per launch, the exports return 3-5)
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
  // Direction.  A.rev: the rounds in descending tile order -- a launch that STARTS where the previous one on the stream
  // ended finds the tail of what that one read and wrote in the memory-side cache (256 MB: two of the 134 MB vectors at
  // 4096^2; walking every launch in the same direction, each line is evicted long before the next launch comes back to it).
  // The rounds are the WORKGROUP's (sibling waves must be in the same one): a wave without a tile in the partial round --
  // the last one ascending, the first one descending -- skips it.
  const int t_wg = t_begin - wave;
  const int nrounds = t_wg < t_end ? (t_end - 1 - t_wg) / t_stride + 1 : 0;
  const bool rev = A.rev != 0;
  const auto tile_at = [&](int j) -> int { return j < nrounds ? t_begin + (rev ? nrounds - 1 - j : j) * t_stride : t_end; };
  const int si = lane >> 2, ss = lane & 3;  // this lane's (block, side) slot of a tile
  const int last = first + count;
  const auto load_nb = [&](int t) -> int {
    if constexpr (HYB) {
      if (t >= t_end) return CUP2D_WALL;
      const int b0 = uniform(A.tile0[t]), nv = uniform(A.tile0[t + 1]) - b0;
      return si < nv ? nbr[4 * (b0 + si) + ss] : CUP2D_WALL;
    }
    const int b = first + t * TB + si;
    return (t < t_end && b < last) ? nbr[4 * b + ss] : CUP2D_WALL;
  };
  struct Tile {
    unsigned long long pmask;               // the tile's perimeter slots (neighbour = a block outside the tile)
    int b0, nvalid, nb, nring, npass, sib;  // sib: the sibling wave whose tile holds this slot's neighbour, or -1
    int is_ring, pad;                       // ... of which the ones to recompute (no tail padding: copies stay in registers)
                                            // pad, HYB: zmask of the tile | (bit b: block b has stored rows) << 16
  };
  // classify the 64 (block, side) neighbour slots of tile t and write its ring list (overwrites the previous tile's)
  // share: edges of the siblings of the same round are taken from their exports (a wait per sibling)
  const bool share_now = (share & 1) != 0;
  const auto classify = [&](int t, int nb) -> Tile {
    Tile T;
    T.b0 = HYB ? uniform(A.tile0[t]) : first + t * TB;
    T.nvalid = HYB ? uniform(A.tile0[t + 1]) - T.b0 : min(TB, last - T.b0);
    T.nb = nb;
    const bool outside = si < T.nvalid && nb >= 0 && (nb < T.b0 || nb >= T.b0 + T.nvalid);
    // a sibling: another tile this workgroup holds in the same round (tiles t - wave .. t - wave + 7 below t_end)
    const int nt = (nb - first) / TB, t0 = t - wave;
    const bool sibling = !HYB && share_now && outside && nb >= first && nb < last && nt >= t0 && nt < t0 + FWAVES && nt < t_end;
    T.sib = sibling ? nt - t0 : -1;
    T.pmask = __ballot(outside);
    T.is_ring = outside && !sibling;
    T.pad = 0;
    if constexpr (HYB) {
      const unsigned long long gm = __ballot(si < T.nvalid && nb == FUSED_GENERAL);  // bit 4 b: block b
      int sm = 0;
#pragma unroll
      for (int b = 0; b < TB; b++) sm |= (int)((gm >> (4 * b)) & 1ull) << b;
      T.pad = uniform(A.zmask[t]) | (sm << 16);
    }
    const unsigned long long rmask = __ballot(T.is_ring);
    T.nring = (KNOCK & 6) ? 0 : __popcll(rmask);
    T.npass = (T.nring + TB - 1) / TB;
    if (T.is_ring && !(KNOCK & 6)) {
      const int slot = bits_below_lane(rmask);
      L.ring_nb[slot] = nb;
      L.ring_dst[slot] = lane;
    }
    wave_lds_sync();
    return T;
  };
  // Batches: 8 blocks = 4 block pairs per half-wave, one 16-byte load per lane, pair and vector; all issue functions are
  // branch-free so that the loads go out back to back.  Ring and tile batches live in SEPARATE register sets (a ring batch
  // has no fourth vector, and two sets with clear live ranges are what the register allocator handles without spills).
  struct RawR {
    double2 a[4], b[4], c[4], d[4];  // AB: p, nu, r   CD: r, nu'   MODE 2: p', nu', r, t
  };
  struct RawT {
    double2 a[4], b[4], c[4], d[4], e[4];  // as RawR; e: MODE 2: y, MODE 3: rhat
  };
  const double2 *in0 = reinterpret_cast<const double2 *>(A.in0), *in1 = reinterpret_cast<const double2 *>(A.in1);
  const double2 *in2 = reinterpret_cast<const double2 *>(A.in2), *inw = reinterpret_cast<const double2 *>(A.w);
  const double2 *int_ = reinterpret_cast<const double2 *>(A.t);
  const auto issue_ring = [&](RawR &R, const Tile &X, int pass, int half) {  // entries 16 pass + 8 half .. + 7 of X's ring list
    if (KNOCK & 6) return;
    const int ne = max(1, min(TB, X.nring - pass * TB));
    const int lo = opaque(lane), hfo = lo >> 5, hlo = lo & 31;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int idx = 8 * half + 2 * p + hfo;
      const int rb = L.ring_nb[min(pass * TB + min(idx, ne - 1), TB * 4 - 1)];
      const size_t o = (size_t)(X.nring > 0 ? rb : X.b0) * (BC / 2) + hlo;  // (no ring: any block of the tile, loaded and dropped)
      if (!fresh) {  // (iteration 0: p = nu = 0 are not read)
        R.a[p] = in0[o];
        R.b[p] = in1[o];
      }
      if (AB || EAB) R.c[p] = in2[o];
      if (EAB) R.d[p] = int_[o];
    }
  };
  const auto issue_tile = [&](RawT &R, const Tile &X, int half) {
    const int lo = opaque(lane), hfo = lo >> 5, hlo = lo & 31;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const size_t o = (size_t)(X.b0 + min(8 * half + 2 * p + hfo, X.nvalid - 1)) * (BC / 2) + hlo;
      if (AB) {
        if (!fresh) {
          R.a[p] = ld2<(POL & 0x1000) != 0>(in0 + o);
          R.b[p] = ld2<(POL & 0x2000) != 0>(in1 + o);
        }
        R.c[p] = in2[o];
      } else if (EAB) {
        R.a[p] = ld2<(POL2 & 0x01) != 0>(in0 + o);
        R.b[p] = ld2<(POL2 & 0x02) != 0>(in1 + o);
        R.c[p] = ld2<(POL2 & 0x04) != 0>(in2 + o);
        R.d[p] = ld2<(POL & 0x200) != 0>(int_ + o);
        if (!yfirst) R.e[p] = ld2<(POL & 0x040) != 0>(yin + o);
      } else {
        R.a[p] = ld2<(POL2 & 0x40) != 0 && CDX>(in0 + o);
        R.b[p] = ld2<(POL2 & 0x80) != 0 && CDX>(in1 + o);
        if (CDX) R.e[p] = ld2<(POL2 & 0x08) != 0>(inw + o);
      }
    }
  };

  // Schedule of a tile (both halves of a job requested a job ahead; the set that is in flight across a job on the matrix
  // cores is never the one being consumed):
  //   top             Qa, Qb = ring pass 0 of this tile in flight (requested behind the previous tile's classification)
  //   ring pass 0     stage Qa, Qb; (further ring passes -- more than 16 entries: other block orders -- requested and waited
  //                   for in place;) request Ta, Tb (the tile's halves); ring job; scatter
  //   tile            stage Ta; stage Tb; classify the next tile, request its ring pass 0 into Qa, Qb;
  //                   tile job; hand-over; ghost gathers; epilogue
  RawR Qa, Qb;
  RawT Ta, Tb;
  Tile T;
  int j = tile_at(0) < t_end ? 0 : 1;
  if (tile_at(j) < t_end) {
    T = classify(tile_at(j), load_nb(tile_at(j)));
    issue_ring(Qa, T, 0, 0);  // (always: a batch set that is assigned on SOME paths only is live around the whole loop)
    issue_ring(Qb, T, 0, 1);
    if constexpr (CD && !HYB && CUP2D_CD_AHEAD >= 1) issue_tile(Ta, T, 0);
    if constexpr (CD && !HYB && CUP2D_CD_AHEAD >= 2) issue_tile(Tb, T, 1);
  }
  int nb_next = load_nb(tile_at(j + 1));
#ifdef EDGE_PHASES
  long long eph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, etprev = clock64();
  int entile = 0;
#endif
  for (int round = j; tile_at(round) < t_end; round++) {
    const int t = tile_at(round);
#ifdef EDGE_PHASES
    entile++;
#endif
    const int b0 = T.b0, nvalid = T.nvalid, par = round % NXB;
    const bool more = tile_at(round + 1) < t_end;
    double2 V[TB / 2];  // v of the tile's cells in pair layout
    double2 W[TB / 2];  // AB, MODE 2, MODE 3: rhat (on a restart: the new one)
    const auto stage_ring = [&](const RawR &R, int half) {
      const int lo = opaque(lane);
      double *Sst = L.S + (lo >> 5) * XS + 2 * (lo & 31);  // one base, the block pair in the instruction's offset field
#pragma unroll
      for (int p = 0; p < 4; p++) {
        double2 v, rn;
        v.x = form_v(R.a[p].x, R.b[p].x, CD ? 0.0 : R.c[p].x, EAB ? R.d[p].x : 0.0, rn.x);
        v.y = form_v(R.a[p].y, R.b[p].y, CD ? 0.0 : R.c[p].y, EAB ? R.d[p].y : 0.0, rn.y);
        *reinterpret_cast<double2 *>(Sst + (8 * half + 2 * p) * XS) = v;
      }
    };
    const auto ring_job = [&](int pass) {
      wave_lds_sync();
      edge_precond(L.S, PE, lane);  // S[e * XS + 8 * side + q] = z of entry (block) e on its four edges
      const int ne = min(TB, T.nring - pass * TB);
      const int lo = opaque(lane);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int idx = lo + 64 * h, e = idx >> 3, q = idx & 7;
        if (e < ne) {
          const int dst = L.ring_dst[pass * TB + e];  // entry e feeds slot dst = block * 4 + side with the OPPOSITE edge
          L.GE[dst * GS + q] = L.S[e * XS + 8 * ((dst & 3) ^ 1) + q];
        }
      }
      wave_lds_sync();
    };
    const auto stage_tile = [&](const RawT &R, int half) {
      const int lo = opaque(lane), hfo = lo >> 5, hlo = lo & 31;
      double *Sst = L.S + hfo * XS + 2 * hlo;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int idx = 8 * half + 2 * p + hfo;
        double2 v, rn;
        v.x = form_v(R.a[p].x, R.b[p].x, CD ? 0.0 : R.c[p].x, EAB ? R.d[p].x : 0.0, rn.x);
        v.y = form_v(R.a[p].y, R.b[p].y, CD ? 0.0 : R.c[p].y, EAB ? R.d[p].y : 0.0, rn.y);
        if (!(KNOCK & 4)) *reinterpret_cast<double2 *>(Sst + (8 * half + 2 * p) * XS) = v;
        V[4 * half + p] = v;
        if (AB && restart) W[4 * half + p] = R.c[p];
        if (EAB && restart) W[4 * half + p] = rn;
        if (CDX) W[4 * half + p] = R.e[p];
        if ((AB || EAB) && idx < nvalid) {  // CD does not store s: sweep E forms it again from r and nu'
          const size_t o = (size_t)(b0 + idx) * (BC / 2) + hlo;
          if (!(KNOCK & 8)) st2<(POL & 0x001) != 0>(reinterpret_cast<double2 *>(A.vout) + o, v);
          if (AB && restart) reinterpret_cast<double2 *>(A.w)[o] = R.c[p];  // rhat = r
          if (EAB) {  // sweep E (k_sweepE_y, operation for operation): y' = y + alpha p' + omega s, r', its norms
            double2 sv, yv = {0.0, 0.0};
            if (!yfirst) yv = R.e[p];
            sv.x = R.c[p].x + malpha * R.b[p].x;
            sv.y = R.c[p].y + malpha * R.b[p].y;
            yv.x = yv.x + alpha * R.a[p].x;
            yv.y = yv.y + alpha * R.a[p].y;
            yv.x = yv.x + omega * sv.x;
            yv.y = yv.y + omega * sv.y;
            if (!(KNOCK & 8)) {
              st2<(POL & 0x010) != 0>(yout2 + o, yv);
              st2<(POL2 & 0x10) != 0>(reinterpret_cast<double2 *>(A.rout) + o, rn);
              if (restart) reinterpret_cast<double2 *>(A.w)[o] = rn;  // rhat = r'
            }
            acc[1] = __builtin_fma(rn.x, rn.x, acc[1]);
            acc[1] = __builtin_fma(rn.y, rn.y, acc[1]);
            rmax[0] = fmax(rmax[0], fmax(fabs(rn.x), fabs(rn.y)));
          }
        }
        if (CDX && idx < nvalid) {  // rhat . s, s . s
          acc[2] = __builtin_fma(R.e[p].x, v.x, acc[2]);
          acc[2] = __builtin_fma(R.e[p].y, v.y, acc[2]);
          acc[4] = __builtin_fma(v.x, v.x, acc[4]);
          acc[4] = __builtin_fma(v.y, v.y, acc[4]);
        }
      }
    };
    // ---- ring passes ----
    // (scheduling barriers: left alone, the scheduler hoists every request to the top of the tile -- four batches in
    // flight at once, 224 registers, and the wave spills)
    if (T.npass > 0) {
      stage_ring(Qa, 0);
      stage_ring(Qb, 1);
    }
    EPH(0)
    // (more than 16 ring entries -- block orders other than the reference's: the further passes are requested and waited
    // for in place, BEFORE the tile's batches go out: a ring set and both tile sets at once do not fit the register file)
    for (int pass = 1; pass < T.npass; pass++) {
      ring_job(pass - 1);
      issue_ring(Qa, T, pass, 0);
      issue_ring(Qb, T, pass, 1);
      stage_ring(Qa, 0);
      stage_ring(Qb, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(CD && !HYB && CUP2D_CD_AHEAD >= 1)) issue_tile(Ta, T, 0);  // (else: in flight since the previous tile's job)
    if constexpr (!(CD && !HYB && CUP2D_CD_AHEAD >= 2)) issue_tile(Tb, T, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (T.npass > 0) ring_job(T.npass - 1);  // the tile's batches are in flight across the job
    __builtin_amdgcn_sched_barrier(0);
    EPH(1)
    // ---- the tile ----
    Tile N = T;
    stage_tile(Ta, 0);
    __builtin_amdgcn_sched_barrier(0);
    if ((AB || EAB) && !restart) {
#pragma unroll
      for (int i = 0; i < TB / 2; i++) {
        const double2 *pw = inw + (((size_t)(b0 + min(2 * i + hf, nvalid - 1)) * BC + c0) >> 1);
        W[i] = ld2<(POL & 0x800) != 0>(pw);
      }
    }
    stage_tile(Tb, 1);
    __builtin_amdgcn_sched_barrier(0);
    EPH(2)
    // this tile's ring list is dead: the NEXT tile is classified into it (past the wave's last tile: this tile once more --
    // the batch sets must be assigned on EVERY path around the loop, or their old contents stay live through all of it)
    N = classify(more ? tile_at(round + 1) : t, more ? nb_next : T.nb);
    nb_next = load_nb(tile_at(round + 2));
    int stored_mask = 0;  // HYB: the blocks of this tile whose rows are k_hyb_rows'
    if constexpr (HYB) {
      const int tp = uniform(T.pad);
      stored_mask = (tp >> 16) & 0xffff;
      if ((tp & 0xffff) != 0) {
        // somebody's stored rows read z of this tile's blocks from memory: the full product, z of those blocks stored
        full_precond_mem(L.S, Pinv, lane);
#pragma unroll
        for (int i = 0; i < TB / 2; i++) {
          const int blk = 2 * i + hf;
          if (blk < nvalid && ((tp >> blk) & 1))
            reinterpret_cast<double2 *>(A.zg)[(size_t)(b0 + blk) * (BC / 2) + hl] = *reinterpret_cast<const double2 *>(L.S + blk * XS + c0);
        }
        if (__popc((unsigned)stored_mask) < nvalid) {  // blocks to finish here: on in the edge layout, S[b * XS + 8 * side + q]
          double ez[BS];
#pragma unroll
          for (int q = 0; q < BS; q++) ez[q] = L.S[si * XS + edge_cell(ss, q)];
          wave_lds_sync();
#pragma unroll
          for (int q = 0; q < BS; q++) L.S[si * XS + 8 * ss + q] = ez[q];
          wave_lds_sync();
        }
      } else {
        edge_precond(L.S, PE, lane);
      }
    } else {
      if (!(KNOCK & 4)) edge_precond(L.S, PE, lane);  // the same product for the tile's own blocks: S[b * XS + 8 * side + q]
    }
    EPH(3)
    // ... and its ring pass 0 requested BEHIND the job: no batch is live across it
    __builtin_amdgcn_sched_barrier(0);
    issue_ring(Qa, N, 0, 0);
    issue_ring(Qb, N, 0, 1);
    if constexpr (CD && !HYB && CUP2D_CD_AHEAD >= 1) issue_tile(Ta, N, 0);
    if constexpr (CD && !HYB && CUP2D_CD_AHEAD >= 2) issue_tile(Tb, N, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (HYB && __popc((unsigned)stored_mask) >= nvalid) {  // every block's rows are k_hyb_rows'
      wave_lds_sync();
      T = N;
      continue;
    }
    EPH(4)
    // ---- export the z edges of the perimeter sides for the siblings ----
    if (share && !(KNOCK & 4)) {
      // this buffer held round - NXB: a sibling has read it by the END of that round of its own, i.e. once it has published
      // the round after it
      const int need = round - NXB + 2;
      if (need > 0) {
        for (int u = 0; u < FWAVES; u++)
          if (u != wave) edge_wait(pub, u, need, fault);
      }
      if (mask_bit(T.pmask, opaque(lane))) {
        const int slot = bits_below_lane(T.pmask);
        if (slot < EXP_SLOTS) {
#pragma unroll
          for (int q = 0; q < BS; q++) L.X[par][slot * BS + q] = L.S[si * XS + 8 * ss + q];
        }
      }
      if (lane == 0) L.xmask[par] = T.pmask;
      wave_lds_sync();
      if (lane == 0) __hip_atomic_store(pub + wave, round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    EPH(5)
    // ---- ghost edges from the own tile and at domain walls (ScalarLab::Neumann2D, main.cpp:3210-3255: ghost = edge cell) ----
    if (si < nvalid && !mask_bit(T.pmask, opaque(lane)) && !(KNOCK & 4)) {
      const int sblk = T.nb < 0 ? si : T.nb - b0, sside = T.nb < 0 ? ss : ss ^ 1;
#pragma unroll
      for (int q = 0; q < BS; q++) L.GE[lane * GS + q] = L.S[sblk * XS + 8 * sside + q];
    }
    // ---- ghost edges a sibling wave exported ----
    if (share_now && !(KNOCK & 4)) {
      for (int u = 0; u < FWAVES; u++) {
        const bool mine = T.sib == u;
        if (__ballot(mine) == 0ull) continue;
        edge_wait(pub, u, round + 1, fault);
        if (mine) {
          const EdgeLds &O = LL[u];
          const unsigned long long m = O.xmask[par];
          const int bit = (T.nb - (b0 + (u - wave) * TB)) * 4 + (ss ^ 1);  // the slot of the neighbour block that faces this one
          const int slot = min((int)__popcll(m & ((1ull << bit) - 1ull)), EXP_SLOTS - 1);  // (int: min(unsigned, int) resolves to the double overload)
#pragma unroll
          for (int q = 0; q < BS; q++) L.GE[lane * GS + q] = O.X[par][slot * BS + q];
        }
      }
    }
    wave_lds_sync();
    EPH(6)
    // ---- y = v + ghosts on the block edges (A P_inv v = v + (C + W) z), the fused dot products: two cells per lane, eight
    //      block pairs ----
    const int lo = opaque(lane), hfo = lo >> 5, hlo = lo & 31;
    const double *gex = L.GE + hfo * 4 * GS + ((2 * hlo) & 7), *gey = L.GE + hfo * 4 * GS + (hlo >> 2);
#pragma unroll
    for (int i = 0; i < TB / 2; i++) {
      const int blk = 2 * i + hfo;
      if (blk < nvalid && !(HYB && ((stored_mask >> blk) & 1))) {  // (stored rows: nu'' or t and the sums with them are k_hyb_rows')
        const double *gx = gex + 2 * i * 4 * GS;
        double2 yv = V[i];
        if (!(KNOCK & 4)) {
#ifndef CUP2D_EDGE_BRANCHY_EPILOGUE
#define CUP2D_EDGE_BRANCHY_EPILOGUE 0
#endif
        if constexpr (AB || CUP2D_EDGE_BRANCHY_EPILOGUE) {  // (MODE 0, once per solve, spills 24 registers under the other form)
        if (px == 0) yv.x += gey[2 * i * 4 * GS + 0 * GS];        // west of cell c0
        if (px == BS - 2) yv.y += gey[2 * i * 4 * GS + 1 * GS];   // east of cell c0 + 1
        if (py == 0) {
          const double2 g = *reinterpret_cast<const double2 *>(gx + 2 * GS);
          yv.x += g.x;
          yv.y += g.y;
        }
        if (py == BS - 1) {
          const double2 g = *reinterpret_cast<const double2 *>(gx + 3 * GS);
          yv.x += g.x;
          yv.y += g.y;
        }
        } else {
        // Branch-free: every lane reads the four ghost operands of its cell pair (the slots exist for every lane; what an interior
        // lane reads is dropped by the select) and the additions -- the same ones in the same order -- are selected per lane.  Written
        // with `if`, every one of the four became a branch with its own LDS round trip behind it (s_and_saveexec ... ds_read ...
        // s_waitcnt lgkmcnt(0)), 32 per tile: the epilogue took 5.7 k of a C+D' tile's 26 k cycles (-DEDGE_PHASES)
        const double gw = gey[2 * i * 4 * GS + 0 * GS], ge = gey[2 * i * 4 * GS + 1 * GS];
        const double2 gs = *reinterpret_cast<const double2 *>(gx + 2 * GS), gn = *reinterpret_cast<const double2 *>(gx + 3 * GS);
        const double xw = yv.x + gw;
        yv.x = px == 0 ? xw : yv.x;        // west of cell c0
        const double ye = yv.y + ge;
        yv.y = px == BS - 2 ? ye : yv.y;   // east of cell c0 + 1
        const double xs = yv.x + gs.x, ys = yv.y + gs.y;
        yv.x = py == 0 ? xs : yv.x;
        yv.y = py == 0 ? ys : yv.y;
        const double xn = yv.x + gn.x, yn = yv.y + gn.y;
        yv.x = py == BS - 1 ? xn : yv.x;
        yv.y = py == BS - 1 ? yn : yv.y;
        }
        }
        if (!(KNOCK & 8)) st2<EAB ? (POL2 & 0x20) != 0 : (POL & (CD ? 0x008 : 0x002)) != 0>(reinterpret_cast<double2 *>(A.yout) + (size_t)(b0 + blk) * (BC / 2) + hlo, yv);
        const double2 wv = CD ? V[i] : W[i];
        acc[0] = __builtin_fma(yv.x, wv.x, acc[0]);
        acc[0] = __builtin_fma(yv.y, wv.y, acc[0]);
        if constexpr (CD) {
          acc[1] = __builtin_fma(yv.x, yv.x, acc[1]);
          acc[1] = __builtin_fma(yv.y, yv.y, acc[1]);
        }
        if constexpr (CDX) {  // rhat . t
          acc[3] = __builtin_fma(W[i].x, yv.x, acc[3]);
          acc[3] = __builtin_fma(W[i].y, yv.y, acc[3]);
        }
      }
    }
    wave_lds_sync();  // the next tile overwrites S and GE
    EPH(7)
    T = N;
  }
#ifdef EDGE_PHASES
  if (sc->iter == 5 && lane == 0 && blockIdx.x < 2 && (wave == 0 || wave == 5) && entile > 0)
    printf("EPHASES mode %d wg %d wave %d tiles %d cycles/tile: ring stage(+wait) %lld  issue+ring job %lld  tile stage(+wait) %lld  classify+tile job %lld  "
           "issue next %lld  export(+wait producers) %lld  own edges + imports(+wait) %lld  epilogue %lld\n", MODE, (int)blockIdx.x, wave, entile, eph[0] / entile,
           eph[1] / entile, eph[2] / entile, eph[3] / entile, eph[4] / entile, eph[5] / entile, eph[6] / entile, eph[7] / entile);
#endif
  // this wave publishes nothing more: nobody may wait for it
  if (lane == 0) __hip_atomic_store(pub + wave, 1 << 30, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  fused_reduce_store<FWAVES, NDOT, FIN != 0>(acc, partials + poff);
  if constexpr (EAB) fused_reduce_store_max<FWAVES, FIN != 0>(rmax[0], partials + 2 * PSTRIDE + poff);
  if (FIN && arrive_last(ticket)) {  // (arrive_last's barrier: every wave of this workgroup is done with the dynamic LDS)
    // MERGE 1: the scalar update too; MERGE 2 (N ranks): this rank's sums into red, the reduction over the ranks and the
    // scalar kernel follow on the stream
    if constexpr (EAB) {
      finish_reduce<true, true>(partials, poff + (int)gridDim.x, 2, 1, red, sc, MERGE == 1 ? 4 : -1, MERGE == 1 ? A.host_status : nullptr, fsm);
    } else if constexpr (CDX) {
      finish_reduce_n<5>(partials, poff + (int)gridDim.x, red, sc, MERGE == 1 ? 5 : -1, fsm);
    } else {
      finish_reduce<true, true>(partials, poff + (int)gridDim.x, NDOT, 0, red, sc, MERGE == 1 ? MODE + 1 : -1, nullptr, fsm);
    }
  }
}

// every tile of [first, first + count) has at most EXP_SLOTS perimeter sides: the export buffers hold them all
static bool edge_share_ok(const int32_t *nbr, int first, int count) {
  for (int b0 = first; b0 < first + count; b0 += TB) {
    const int nv = std::min(TB, first + count - b0);
    int n = 0;
    for (int b = b0; b < b0 + nv; b++)
      for (int s = 0; s < 4; s++) {
        const int nb = nbr[4 * b + s];
        n += nb >= 0 && (nb < b0 || nb >= b0 + nv);
      }
    if (n > EXP_SLOTS) return false;
  }
  return true;
}

