// walk_emul.cpp -- TEST INFRASTRUCTURE.  Runs the per-lane functions of cup2d_amd/csrc/advect_walk.h (the very code
// k_advect_walk executes on the GPU) lane by lane on the CPU: one "wave" = a loop over 64 lanes per phase, the LDS image
// a plain struct.  tests/test_walk_host.py compares the result with the oracle, so that indexing and arithmetic of
// the quad kernel are checked without a GPU.  Built on demand: g++ -O1 -ffp-contract=off -shared -fPIC.
#include <string.h>

#include "../cup2d_amd/csrc/advect_walk.h"
#include "../cup2d_amd/csrc/ranges.h"

using namespace cup2d::walk;

template <int MODE, bool OLDLAB>
static void run_quad(const V2 *vel, const V2 *vold, double *out, const int *q, double afc, double dfc) {
  constexpr bool NEED_OLD = MODE == 1 && !OLDLAB;
  static Lds L;
  static Regs R[64];
  int gp[64][3];
  const Entry t = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]};
  memset(&L, 0x7f, sizeof L);  // poison: a cell nobody staged shows up as a huge number
  for (int lane = 0; lane < 64; lane++) {
    for (int i = 0; i < 3; i++) gp[lane][i] = ghost_pack(lane, i);
    fetch<NEED_OLD>(R[lane], vel, vold, t, lane, gp[lane]);
  }
  int signs = 0;
  for (int lane = 0; lane < 64; lane++) {
    stage_lab(R[lane], L, lane, gp[lane]);
    if (NEED_OLD) stage_old(R[lane], L, lane);
    signs |= lane_signs(R[lane]);
  }
  const bool nPx = signs & 1, nMx = signs & 2, nPy = signs & 4, nMy = signs & 8;
  for (int lane = 0; lane < 64; lane++) {
    if (!nMx) xwalk<true, false, MODE, OLDLAB>(L, lane, afc, dfc);
    else if (!nPx) xwalk<false, true, MODE, OLDLAB>(L, lane, afc, dfc);
    else xwalk<true, true, MODE, OLDLAB>(L, lane, afc, dfc);
  }
  for (int lane = 0; lane < 64; lane++) {
    if (!nMy) ywalk<true, false, true>(L, lane, afc, dfc);
    else if (!nPy) ywalk<false, true, true>(L, lane, afc, dfc);
    else ywalk<true, true, true>(L, lane, afc, dfc);
  }
  for (int lane = 0; lane < 64; lane++) flush(L, lane, (V2 *)out, q[0], q[1], q[2], q[3]);
}

extern "C" {
// plan of [first, first + count): returns the number of quads; quads (12 ints each) and singles are copied out
int walk_emul_plan(const int32_t *nbr, int first, int count, int32_t *quads, int32_t *singles, int *nsingles) {
  std::vector<int32_t> q, s;
  build_plan(nbr, first, count, q, s);
  if (quads) memcpy(quads, q.data(), q.size() * sizeof(int32_t));
  if (singles) memcpy(singles, s.data(), s.size() * sizeof(int32_t));
  *nsingles = (int)s.size();
  return (int)(q.size() / QINTS);
}
// the kernel on every quad of the plan; mode / oldlab / coefficients as launch_advect passes them
void walk_emul_run(const int32_t *quads, int nquads, const double *vel, const double *vold, double *out, int mode,
                   int oldlab, double afc, double dfc) {
  const V2 *v = (const V2 *)vel, *vo = (const V2 *)vold;
  for (int i = 0; i < nquads; i++) {
    const int *q = quads + (size_t)i * QINTS;
    if (mode == 0) run_quad<0, false>(v, vo, out, q, afc, dfc);
    else if (oldlab) run_quad<1, true>(v, vo, out, q, afc, dfc);
    else run_quad<1, false>(v, vo, out, q, afc, dfc);
  }
}

// The loop of k_advect_walk over a wave's quads, replayed for EVERY workgroup of a launch with the kernel's own cursor
// (ranges.h): init, the plan entry of the first quad and the read-ahead of the second, then per iteration advance() and the
// read-ahead of the quad after next.  G workgroups of wpg waves; chunk > 0 = the chunked grid (G must be
// chunked_grid(nq, chunk)), 0 = the persistent one.  Returns the largest table index any wave touched (must be < nq), or
// -1 - (a quad that was not visited exactly once).  visits: scratch of nq ints.
long long walk_emul_cursor(int nq, int wpg, int G, int chunk, int *visits) {
  long long worst = -1;
  for (int i = 0; i < nq; i++) visits[i] = 0;
  for (int w = 0; w < G; w++) {
    const cup2d::GroupRange gr = chunk > 0 ? cup2d::group_range_chunked_of(nq, wpg, chunk, w) : cup2d::group_range_of(nq, wpg, G, w);
    for (int wave = 0; wave < wpg; wave++) {
      cup2d::WaveCursor cur;
      cur.init(gr, wave, wpg, nq);
      if (!cur.have) continue;
      long long i0 = cur.item(0), i1 = cur.item(1);
      if (i0 > worst) worst = i0;
      if (i1 > worst) worst = i1;
      if (i0 < 0 || i1 < 0) return -1 - (long long)nq;
      int current = (int)i0, next = (int)i1;  // the entry in E and the one in flight (vnext)
      while (cur.have) {
        if (current >= 0 && current < nq) visits[current]++;
        cur.advance();
        current = next;  // WALK_READ(E, vnext): valid also past the end (the last quad again, fetched and dropped)
        const long long ia = cur.item(1);
        if (ia > worst) worst = ia;
        if (ia < 0) return -1 - (long long)nq;
        next = (int)ia;
      }
    }
  }
  for (int i = 0; i < nq; i++)
    if (visits[i] != 1) return -1 - (long long)i;
  return worst;
}
}
