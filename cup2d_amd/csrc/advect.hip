// advect.hip -- WENO5 advection + 5-point diffusion on the block grid
// and the vorticity functor (a19).
//
// Reference: KernelAdvectDiffuse::operator() main.cpp:5441-5503 driven by computeA<VectorLab>
// (main.cpp:3024-3061) and the RK2 glue main.cpp:6607-6642.
//
// Mapping.  One wavefront owns one 8x8 block (64 cells = 64 lanes).  The 14x14 ghosted tile is
// staged in LDS straight from the device-resident slabs through the neighbour table (no host-side
// BlockLab, no hash maps); the loads of the NEXT block are issued before the current one is
// computed.  The kernel is FP64-issue bound, not HBM bound, so the work is cut,
// not the bytes:
//   * a WENO face value is a function of five cells only (weno.h), so each lane evaluates the
//     `plus` and `minus` face values about ITS OWN cell once per direction and component (shared
//     smoothness indicators), the 64 rim centres of the block (centre -1 / centre 8 of every row and
//     column, both components) are spread over the 64 lanes as one extra reconstruction each, and
//     the upwind neighbour's value is picked up through LDS: 9 reconstructions per cell, no
//     divergent upwind branch, where the reference's expression costs 8 (uniform sign) to 16;
//   * in the fused modes the Runge-Kutta update is applied in the same pass, so the tmpV field and
//     the separate axpy sweep (48 B/cell) of the reference disappear.
#include <stdlib.h>

#include "advect_tile.h"
#include "advect_walk.h"
#include "block.h"
#include "weno.h"

namespace cup2d {

// registers holding one block's tile while it is in flight from HBM/L2.  Nothing here may touch
// a loaded value (not even a copy): the first use would put the s_waitcnt right behind the loads and
// turn the prefetch back into a blocking load.  The wall sign flips are applied in lab3_store.
struct LabRegs {
  double2 own, we, sn, old;
  bool flip_we, flip_sn;
};

// Branch-free on purpose: lanes 48..63 (no strip cell of their own) re-read their own cell, and a
// wave past its last block re-reads its current one, so that the loaded registers are never merged
// with other values at a control-flow join (the compiler would wait for the loads there).
template <int MODE>
static __device__ __forceinline__ void lab3_fetch(LabRegs &R, const double2 *__restrict__ f,
                                                  const double2 *__restrict__ vold, const int4 *__restrict__ nbr4,
                                                  int b, int lane) {
  const int4 nb4 = nbr4[b];  // b is wave-uniform: one scalar load
  const double2 *own = f + (size_t)b * BC;
  R.own = own[lane];
  if (MODE == 1) R.old = vold[(size_t)b * BC + lane];
  const bool strip = lane < 48;
  const int side = lane >= 24 && strip, t = strip ? lane - 24 * side : 0;
  {  // W / E strips: 8 rows x 3 columns
    const int r = t / 3, k = t - 3 * r;
    const int nb = side ? nb4.y : nb4.x;
    const int cell_nb = r * BS + (side ? k : 5 + k), cell_own = r * BS + (side ? 7 : 0);
    const double2 *src = nb >= 0 ? f + (size_t)nb * BC + cell_nb : own + cell_own;
    R.we = *src;
    R.flip_we = nb < 0;  // VectorLab::applyBCface, main.cpp:3131-3204: wall-normal component negated
  }
  {  // S / N strips: 3 rows x 8 columns
    const int j = t >> 3, x = t & 7;
    const int nb = side ? nb4.w : nb4.z;
    const int cell_nb = (side ? j : 5 + j) * BS + x, cell_own = (side ? 7 : 0) * BS + x;
    const double2 *src = nb >= 0 ? f + (size_t)nb * BC + cell_nb : own + cell_own;
    R.sn = *src;
    R.flip_sn = nb < 0;
  }
}
static __device__ __forceinline__ void lab3_store(const LabRegs &R, int lane, double2 *lab) {
  const int ix = lane & 7, iy = lane >> 3;
  lab[(iy + 3) * LABS + ix + 3] = R.own;
  if (lane < 48) {
    const int side = lane >= 24, t = lane - 24 * side;
    const int r = t / 3, k = t - 3 * r;
    double2 we = R.we, sn = R.sn;
    if (R.flip_we) we.x = -we.x;
    if (R.flip_sn) sn.y = -sn.y;
    lab[(r + 3) * LABS + (side ? 11 + k : k)] = we;
    const int j = t >> 3, x = t & 7;
    lab[(side ? 11 + j : j) * LABS + x + 3] = sn;
  }
}

// MODE 0: out = rhs                      (the functor alone: tmpV)
// MODE 1: out = vold + coef * rhs        (RK stage: coef = 0.5/h^2 or 1/h^2, main.cpp:6623, 6639)
template <class W, int MODE>
__global__ __launch_bounds__(WG, 4) void k_advect_diffuse(const double2 *__restrict__ vel,
                                                       const double2 *__restrict__ vold,
                                                       double2 *__restrict__ out, const int *__restrict__ nbr,
                                                       const int *__restrict__ list, int first, int count, int chunk,
                                                       double afac, double dfac, double coef) {
  __shared__ AdvectLds lds[WPG];
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  AdvectLds &L = lds[wave];
  const int4 *nbr4 = (const int4 *)nbr;
  RimSlot rim;
  rim.init(L, lane);

  const GroupRange gr = chunk > 0 ? group_range_chunked(count, chunk) : group_range(count);
  int g = gr.begin;
  bool have = g < gr.end && g * WPG + wave < count;
  if (!have) return;
  // list: the blocks of a plan that found no partners for a quad (advect_walk.h); otherwise the range [first, first + count)
  auto block_of = [&](int i) { return list ? list[first + i] : first + i; };
  LabRegs R;
  int b = block_of(g * WPG + wave);
  lab3_fetch<MODE>(R, vel, vold, nbr4, b, lane);
  while (have) {
    lab3_store(R, lane, L.lab);
    const double2 old = R.old;
    wave_lds_sync();
    // prefetch the next block of this wave (its last block is simply fetched twice)
    g += gr.stride;
    have = g < gr.end && g * WPG + wave < count;
    const int bnext = have ? block_of(g * WPG + wave) : b;
    lab3_fetch<MODE>(R, vel, vold, nbr4, bnext, lane);
    double2 r = advect_cell<W>(L, rim, lane, afac, dfac);
    if (MODE == 1) {
      r.x = old.x + r.x * coef;
      r.y = old.y + r.y * coef;
    }
    out[(size_t)b * BC + lane] = r;
    b = bnext;
    wave_lds_sync();  // tile and face values are overwritten by the next block
  }
}

// ---- the quad form (advect_walk.h): one wave = 2 x 2 blocks, the reconstruction walks along the grid lines ----------
// 13 KB of LDS per wave (ghosted 22 x 22 tile + hand-over buffer) -> three workgroups of four waves per CU; the next
// quad's 11 loads per lane (own cells, old values, three ghost cells) are in flight while the current one is computed.
// KO (timing aid, WRONG results, reachable through cup2d_debug_walk_knockout only: bench.py's floors of this kernel, measured in
// the process, on the data and behind the launches of the product): 1 = no arithmetic -- the memory skeleton of the loop;
// 2 = no loads / stores inside the loop -- the arithmetic alone, on the wave's first quad
template <int MODE, bool OLDLAB, int KO = 0>
__global__ __launch_bounds__(WG, 3) void k_advect_walk(const walk::V2 *__restrict__ vel, const walk::V2 *__restrict__ vold,
                                                       double *__restrict__ out, const int *__restrict__ quads, int nq,
                                                       int chunk, double afc, double dfc, int prio_mode) {
  constexpr bool NEED_OLD = MODE == 1 && !OLDLAB;
  __shared__ walk::Lds lds[WPG];
  const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
  walk::Lds &L = lds[wave];
  int gp[3];
#pragma unroll
  for (int i = 0; i < 3; i++) gp[i] = walk::ghost_pack(lane, i);
  const GroupRange gr = chunk > 0 ? group_range_chunked(nq, chunk) : group_range(nq);
  // cur: the wave's position in its workgroup's groups (ranges.h).  A read-ahead past the wave's last quad is redirected
  // to that last quad -- never to `cur.g`, which runs one stride past the end on the final iteration.
  WaveCursor cur;
  cur.init(gr, wave, WPG, nq);
  if (!cur.have) return;
  // The plan entry of a quad (12 ints) is read by lanes 0..11 with ONE vector load, two quads ahead of its use, and
  // handed out with v_readlane: the block ids are wave-uniform values by the time the loads that depend on them are
  // issued (a per-lane load of the table in front of the ghost loads stalled every quad for a memory round trip).
  const int tl = lane < walk::QINTS ? lane : 0;
  auto entry_ahead = [&](int ahead) { return quads + (size_t)cur.item(ahead) * walk::QINTS; };
  // (no copy of a loaded register anywhere: a v_mov of `vnext` is a use, and the wait it brings covers every load and
  // store issued before it)
#define WALK_READ(E, v)                                                                                             \
  E.b0 = __builtin_amdgcn_readlane(v, 0), E.b1 = __builtin_amdgcn_readlane(v, 1), E.b2 = __builtin_amdgcn_readlane(v, 2),    \
  E.b3 = __builtin_amdgcn_readlane(v, 3), E.n0 = __builtin_amdgcn_readlane(v, 4), E.n1 = __builtin_amdgcn_readlane(v, 5),    \
  E.n2 = __builtin_amdgcn_readlane(v, 6), E.n3 = __builtin_amdgcn_readlane(v, 7), E.n4 = __builtin_amdgcn_readlane(v, 8),    \
  E.n5 = __builtin_amdgcn_readlane(v, 9), E.n6 = __builtin_amdgcn_readlane(v, 10), E.n7 = __builtin_amdgcn_readlane(v, 11)
  walk::Entry E;
  {
    const int v0 = entry_ahead(0)[tl];
    WALK_READ(E, v0);
  }
  walk::Regs R;
  walk::fetch<NEED_OLD>(R, vel, vold, E, lane, gp);
  int vnext = entry_ahead(1)[tl];
  walk::V2 *out2 = reinterpret_cast<walk::V2 *>(out);
  int pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;  // the blocks of the quad whose results wait in L.T
  bool pending = false;
  // VALU issue goes to the OLDEST of the three waves that share a SIMD; with equal shares of quads the oldest wave of a
  // SIMD finishes early and the youngest computes alone at the end (average wave lifetime 74 % of the kernel).  Priority
  // outranks age, so every wave takes the top priority for one quad in three: equal progress, a common finish.
  const int wid = __builtin_amdgcn_s_getreg((3 << 11) | 4) % 3;  // HW_ID.wave_id: the slot in the SIMD's wave buffer
  int turn = wid;
  while (cur.have) {
    if (prio_mode) {
      if (turn == 0) __builtin_amdgcn_s_setprio(2);
      else if (turn == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
      turn = turn == 2 ? 0 : turn + 1;
    }
    walk::stage_lab(R, L, lane, gp);
    // which upwind sides anybody in the tile asks for: eight compares whose results are lane masks, combined in scalar registers
    unsigned long long anyu = 0, allu = ~0ull, anyv = 0, allv = ~0ull;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const unsigned long long mu = __ballot(R.own[p][0] > 0), mv = __ballot(R.own[p][1] > 0);
      anyu |= mu, allu &= mu, anyv |= mv, allv &= mv;
    }
    const bool nPx = anyu != 0ull, nMx = allu != ~0ull, nPy = anyv != 0ull, nMy = allv != ~0ull;
    const int cb0 = E.b0, cb1 = E.b1, cb2 = E.b2, cb3 = E.b3;
    // the next quad of this wave (its last one is simply fetched twice)
    cur.advance();
    WALK_READ(E, vnext);  // (waits for everything issued so far: the loads of this quad, the stores of the one before last)
    if (pending && KO != 2) walk::flush(L, lane, out2, pb0, pb1, pb2, pb3);
    if (NEED_OLD) walk::stage_old(R, L, lane);
    wave_lds_sync();
    if (KO != 2) {
      walk::fetch<NEED_OLD>(R, vel, vold, E, lane, gp);
      vnext = entry_ahead(1)[tl];
    }
    if (KO == 1) {
      pb0 = cb0, pb1 = cb1, pb2 = cb2, pb3 = cb3;
      pending = true;
      continue;
    }
    if (!nMx) walk::xwalk<true, false, MODE, OLDLAB>(L, lane, afc, dfc);
    else if (!nPx) walk::xwalk<false, true, MODE, OLDLAB>(L, lane, afc, dfc);
    else walk::xwalk<true, true, MODE, OLDLAB>(L, lane, afc, dfc);
    wave_lds_sync();
    if (!nMy) walk::ywalk<true, false, true>(L, lane, afc, dfc);
    else if (!nPy) walk::ywalk<false, true, true>(L, lane, afc, dfc);
    else walk::ywalk<true, true, true>(L, lane, afc, dfc);
    wave_lds_sync();  // the tile is overwritten by the next quad; its results stay in L.T
    pb0 = cb0, pb1 = cb1, pb2 = cb2, pb3 = cb3;
    pending = true;
  }
  walk::flush(L, lane, out2, pb0, pb1, pb2, pb3);
#undef WALK_READ
}

// the plan of a block range: built once per (first, count) of a context (the neighbour table never changes)
static const WalkPlan *walk_plan(cup2d_ctx *c, int first, int count) {
  for (const WalkPlan &p : c->walk_plans)
    if (p.first == first && p.count == count) return &p;
  std::vector<int32_t> quads, singles;
  walk::build_plan(c->h_nbr.data(), first, count, quads, singles);
  WalkPlan p;
  p.first = first;
  p.count = count;
  p.nquads = (int)(quads.size() / walk::QINTS);
  p.nsingles = (int)singles.size();
  const auto fail = [&]() -> const WalkPlan * {  // nothing has been launched on these tables: they go straight back
    dev_free(p.d_quads);
    dev_free(p.d_singles);
    return nullptr;
  };
  if (p.nquads) {
    if (dev_malloc(&p.d_quads, quads.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(p.d_quads, quads.data(), quads.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
      return fail();
  }
  if (p.nsingles) {
    if (dev_malloc(&p.d_singles, singles.size() * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(p.d_singles, singles.data(), singles.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
      return fail();
  }
  c->walk_plans.push_back(p);
  return &c->walk_plans.back();
}
void walk_plans_release(cup2d_ctx *c) {
  for (WalkPlan &p : c->walk_plans) {
    dev_free(p.d_quads);
    dev_free(p.d_singles);
  }
  c->walk_plans.clear();
}

int launch_advect(cup2d_ctx *c, const double *vel, const double *vold, double *out, int mode, double nu, double dt,
                  double coef, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  const double afac = -dt * c->h, dfac = nu * dt;  // main.cpp:5446-5447
  const int timer = (mode == 1 && vold != vel) ? CUP2D_T_ADVECT_STAGE2 : CUP2D_T_ADVECT_STAGE;  // stage 2 reads its own old values: 48 B/cell
  // cup2d_debug_walk_knockout (the floors of bench.py's roofline): the knocked-out instantiation is launched -- and timed --
  // FIRST, on the stage's own inputs with the launches of the step in front of it, and writes to a scratch slab; the product
  // launch follows untimed, so that the step stays a step and the next one meets the same conditions
  const int ko = (c->math != CUP2D_MATH_STRICT && mode == 1) ? c->walk_knockout : 0;
  if (ko && !c->d_ko_scratch) {
    CUP2D_HIP_CHECK(dev_malloc(&c->d_ko_scratch, (size_t)c->ntotal * BC * 2 * sizeof(double)));
  }
  ProfScope prof(c, ko ? -1 : timer);
  const double2 *v = (const double2 *)vel, *vo = (const double2 *)vold;
  double2 *o = (double2 *)out;
  constexpr int chunk = 16;  // groups (of 4 blocks) per workgroup of the per-block kernel (0 would select the persistent grid)
  // FAST policy: quads of 2 x 2 blocks take the register-walk kernel, blocks without partners the per-block one
  // (CUP2D_ADVECT_WALK=0: everything per block, the round-1 kernel -- A/B timing aid)
  static const bool use_walk = [] { const char *e = getenv("CUP2D_ADVECT_WALK"); return !e || atoi(e) != 0; }();
  constexpr int wchunk = 0, wprio = 1;  // the quad kernel: persistent grid, s_setprio around the arithmetic phases (measured: round 2)
  const int *list = nullptr;
  if (c->math != CUP2D_MATH_STRICT && use_walk && ((size_t)c->ntotal << 10) < (1ull << 32)) {  // 32-bit byte offsets into a vector slab
    const WalkPlan *p = walk_plan(c, first, count);
    if (!p) {
      set_error("launch_advect: plan of blocks [%d, %d) could not be built", first, first + count);
      return CUP2D_ERR_HIP;
    }
    if (p->nquads) {
      const double k = mode == 0 ? 1.0 : coef;
      const walk::V2 *wv = (const walk::V2 *)vel, *wo = (const walk::V2 *)vold;
#define LAUNCHW(M, OL, KO)                                                                                         \
  hipLaunchKernelGGL((k_advect_walk<M, OL, KO>),                                                                   \
                     dim3(wchunk > 0 ? chunked_grid(p->nquads, wchunk)                                             \
                                     : resident_grid(c, reinterpret_cast<const void *>(&k_advect_walk<M, OL, KO>), p->nquads)), \
                     dim3(WG), 0, c->stream, wv, wo, out, p->d_quads, p->nquads, wchunk, k * afac, k * dfac, wprio)
      if (ko) {
        double *real_out = out;
        out = c->d_ko_scratch;
        {
          ProfScope pko(c, timer);
          if (ko == 1) { if (vold == vel) LAUNCHW(1, true, 1); else LAUNCHW(1, false, 1); }
          else { if (vold == vel) LAUNCHW(1, true, 2); else LAUNCHW(1, false, 2); }
        }
        out = real_out;
      }
      if (mode == 0) LAUNCHW(0, false, 0);
      else if (vold == vel) LAUNCHW(1, true, 0);
      else LAUNCHW(1, false, 0);
#undef LAUNCHW
      CUP2D_HIP_CHECK(hipGetLastError());
    }
    if (!p->nsingles) return CUP2D_OK;
    list = p->d_singles;
    first = 0;
    count = p->nsingles;
  }
#define LAUNCH(Wt, M)                                                                                              \
  hipLaunchKernelGGL((k_advect_diffuse<Wt, M>),                                                                    \
                     dim3(chunk > 0 ? chunked_grid(count, chunk)                                                   \
                                    : resident_grid(c, reinterpret_cast<const void *>(&k_advect_diffuse<Wt, M>), count)), \
                     dim3(WG), 0, c->stream, v, vo, o, c->d_nbr, list, first, count, chunk, afac, dfac, coef)
  if (c->math == CUP2D_MATH_STRICT) {
    if (mode == 0) LAUNCH(WenoStrict, 0); else LAUNCH(WenoStrict, 1);
  } else {
    if (mode == 0) LAUNCH(WenoFast, 0); else LAUNCH(WenoFast, 1);
  }
#undef LAUNCH
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

int launch_advect_walk_rhs(cup2d_ctx *c, const double *vel, double *out, const int32_t *d_quads, int nq, double afac, double dfac) {
  if (nq <= 0) return CUP2D_OK;
  const walk::V2 *wv = (const walk::V2 *)vel;
  hipLaunchKernelGGL((k_advect_walk<0, false, 0>), dim3(resident_grid(c, reinterpret_cast<const void *>(&k_advect_walk<0, false, 0>), nq)),
                     dim3(WG), 0, c->stream, wv, wv, out, d_quads, nq, 0, afac, dfac, 1);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

// KernelVorticity main.cpp:3343-3366: tmp = (0.5/h) * (u_S - u_N + v_E - v_W)
__global__ __launch_bounds__(WG) void k_vorticity(const double2 *__restrict__ vel, double *__restrict__ out,
                                                  const int *__restrict__ nbr, int first, int count, double i2h) {
  __shared__ double2 labs[WPG][LAB1 * LAB1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double2 *lab = labs[wave];
  const int ix = lane & 7, iy = lane >> 3;
  const GroupRange gr = group_range(count);
  for (int g = gr.begin; g < gr.end; g += gr.stride) {
    const int rel = g * WPG + wave;
    if (rel < count) {
      const int b = first + rel;
      load_vector_lab1(vel, nbr, b, lane, lab);
      wave_lds_sync();
      const int c0 = (iy + 1) * LAB1 + ix + 1;
      const double e0 = lab[c0 - LAB1].x, e1 = lab[c0 + LAB1].x, e2 = lab[c0 + 1].y, e3 = lab[c0 - 1].y;
      out[(size_t)b * BC + lane] = i2h * (e0 - e1 + e2 - e3);
      wave_lds_sync();
    }
  }
}

int launch_vorticity(cup2d_ctx *c, const double *vel, double *out, int first, int count) {
  if (count <= 0) return CUP2D_OK;
  hipLaunchKernelGGL(k_vorticity, dim3(grid_for(c, count)), dim3(WG), 0, c->stream, (const double2 *)vel, out,
                     c->d_nbr, first, count, 0.5 / c->h);
  CUP2D_HIP_CHECK(hipGetLastError());
  return CUP2D_OK;
}

}  // namespace cup2d
