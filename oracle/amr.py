"""CPU restatement of the reference's block-AMR ghost fill and flux correction for HALO-1 labs (numpy, small cases)
-- TEST INFRASTRUCTURE, NOT PRODUCT (only tests/ may import it).

What it restates, line by line, for Stencil{-1,-1,2,2, tensorial=false} (pressure_rhs1, pressureCorrectionKernel,
pressure_rhs, KernelVorticity):
  * BlockLab::load      main.cpp:2270-2687  same-level copy (2398-2440), finer neighbour: 2x2 averages (2441-2566,
                        INCLUDING the reference's unrolled W/E branch whose first row averages fine rows y and y+2,
                        lines 2528-2531), coarser neighbour: copy of the coarse block's cells into `c` (2312-2390)
  * BlockLab::post_load main.cpp:2689-2933  coarse -> fine ghosts: quadratic interpolation ALONG the face from the
                        coarse column (2771-2846), then LI() ACROSS the face with the two fine interior cells (2848-2927,
                        LI main.cpp:2203-2210); use_averages is false for this stencil (2265-2267)
  * ScalarLab/VectorLab::_apply_bc (3131-3255): wall ghost = edge cell, wall-normal vector component negated
  * pressure_rhs1 (6209-6285) with its face arrays, prepare0 / fillcases / fillcase0 / fillcase1 (1564-1849):
    a coarse block's edge cell gets  + (its own flux through the coarse-fine face) + (sum of the two fine fluxes)
Pinned against the reference itself: tests/test_amr_oracle.py compares with oracle/_ref/ref_harness 'amr' (reps=-1) and the
committed golden tests/golden/amr_functors.npz.  Domain: bpdx = bpdy = 1, extent 1 (the harness configuration)."""
import numpy as np

BS = 8
W, E, S, N = 0, 1, 2, 3


class AmrGrid:
    def __init__(self, blocks):
        """blocks: (nb, 3) int array of leaf blocks (level, i, j)"""
        self.blocks = np.asarray(blocks, dtype=np.int64)
        self.index = {tuple(b): k for k, b in enumerate(self.blocks)}

    def h(self, level):
        return 1.0 / BS / (1 << level)

    def neighbour(self, b, side):
        """('wall',) | ('same', k) | ('coarse', k) | ('fine', k0, k1)  (k0, k1 ordered along the face)"""
        l, i, j = (int(v) for v in self.blocks[b])
        di, dj = ((-1, 0), (1, 0), (0, -1), (0, 1))[side]
        ni, nj = i + di, j + dj
        n = 1 << l
        if ni < 0 or nj < 0 or ni >= n or nj >= n:
            return ("wall",)
        if (l, ni, nj) in self.index:
            return ("same", self.index[(l, ni, nj)])
        if l > 0 and (l - 1, ni // 2, nj // 2) in self.index:
            return ("coarse", self.index[(l - 1, ni // 2, nj // 2)])
        # refined: the two children of (l, ni, nj) that touch our face
        if side == W:
            kids = [(l + 1, 2 * ni + 1, 2 * nj + a) for a in (0, 1)]
        elif side == E:
            kids = [(l + 1, 2 * ni, 2 * nj + a) for a in (0, 1)]
        elif side == S:
            kids = [(l + 1, 2 * ni + a, 2 * nj + 1) for a in (0, 1)]
        else:
            kids = [(l + 1, 2 * ni + a, 2 * nj) for a in (0, 1)]
        return ("fine", self.index[kids[0]], self.index[kids[1]])


def LI(a, b, c):
    """main.cpp:2203-2210"""
    kappa = ((4.0 / 15.0) * a + (6.0 / 15.0) * c) + (-10.0 / 15.0) * b
    lam = (b - c) - kappa
    return (4.0 * kappa + 2.0 * lam) + c


def _tangential(cc):
    """cc[0..3]: the four coarse cells along the face that span the fine block; returns the 8 fine ghost values
    before the normal correction (main.cpp:2797-2846: centred / one-sided quadratic at dy = -+1/4)."""
    out = np.empty(8)
    for q in range(4):
        c1 = cc[q]
        if q == 0:
            d1 = (-0.5 * cc[2] - 1.5 * cc[0]) + 2.0 * cc[1]
            d2 = (cc[2] + cc[0]) - 2.0 * cc[1]
        elif q == 3:
            d1 = (0.5 * cc[1] + 1.5 * cc[3]) - 2.0 * cc[2]
            d2 = (cc[1] + cc[3]) - 2.0 * cc[2]
        else:
            d1 = 0.5 * (cc[q + 1] - cc[q - 1])
            d2 = (cc[q + 1] + cc[q - 1]) - 2.0 * cc[q]
        dy = -0.25
        out[2 * q] = c1 + dy * d1 + (0.5 * dy * dy) * d2
        out[2 * q + 1] = c1 - dy * d1 + (0.5 * dy * dy) * d2
    return out


def lab1(grid, field, b, comp_sign=None):
    """10x10 ghosted tile (cross only) of scalar `field[nb][64]` for block b.  comp_sign: None for a scalar with
    Neumann walls; for one component of a vector, (sx, sy) = factor applied to the wall ghost on W/E resp. S/N
    walls (VectorLab: the wall-normal component is negated)."""
    own = field[b].reshape(BS, BS)
    m = np.full((BS + 2, BS + 2), np.nan)
    m[1:-1, 1:-1] = own
    l, bi, bj = (int(v) for v in grid.blocks[b])
    for side in (W, E, S, N):
        nb = grid.neighbour(b, side)
        g = np.empty(8)
        if nb[0] == "wall":
            edge = own[:, 0] if side == W else own[:, 7] if side == E else own[0, :] if side == S else own[7, :]
            sgn = 1.0 if comp_sign is None else (comp_sign[0] if side < 2 else comp_sign[1])
            g[:] = sgn * edge
        elif nb[0] == "same":
            o = field[nb[1]].reshape(BS, BS)
            g[:] = o[:, 7] if side == W else o[:, 0] if side == E else o[7, :] if side == S else o[0, :]
        elif nb[0] == "fine":
            for a in (0, 1):
                f = field[nb[1 + a]].reshape(BS, BS)  # [row][col]
                if side in (S, N):
                    r0 = 6 if side == S else 0  # rows y, y+1 of the fine block (main.cpp:2548-2551)
                    for ee in range(4):
                        g[4 * a + ee] = (f[r0, 2 * ee] + f[r0 + 1, 2 * ee] + f[r0, 2 * ee + 1] + f[r0 + 1, 2 * ee + 1]) / 4
                else:
                    x = 6 if side == W else 0
                    # unrolled branch (main.cpp:2476-2534): rows y0..y3 = 0, 2, 4, 6; p0 pairs rows y0 and y1 (!)
                    g[4 * a + 0] = (f[0, x] + f[2, x] + f[0, x + 1] + f[2, x + 1]) / 4
                    for r in (1, 2, 3):
                        y = 2 * r
                        g[4 * a + r] = (f[y, x] + f[y + 1, x] + f[y, x + 1] + f[y + 1, x + 1]) / 4
        else:  # coarser neighbour
            cb = field[nb[1]].reshape(BS, BS)
            if side in (W, E):
                half = bj % 2
                col = 7 if side == W else 0
                t = _tangential(cb[4 * half:4 * half + 4, col])
                b1 = own[:, 0] if side == W else own[:, 7]
                b2 = own[:, 1] if side == W else own[:, 6]
            else:
                half = bi % 2
                row = 7 if side == S else 0
                t = _tangential(cb[row, 4 * half:4 * half + 4])
                b1 = own[0, :] if side == S else own[7, :]
                b2 = own[1, :] if side == S else own[6, :]
            for q in range(8):
                g[q] = LI(t[q], b1[q], b2[q])
        if side == W:
            m[1:-1, 0] = g
        elif side == E:
            m[1:-1, -1] = g
        elif side == S:
            m[0, 1:-1] = g
        else:
            m[-1, 1:-1] = g
    return m


def laplacian_sub_amr(grid, pold, tmp):
    """tmp -= Lap5(pold) on the adapted grid with the reference's flux correction (main.cpp:7022-7027)."""
    nb = len(grid.blocks)
    out = tmp.copy().reshape(nb, BS, BS)
    faces = {}
    for b in range(nb):
        m = lab1(grid, pold, b)
        l0 = m[1:-1, 1:-1]
        out[b] -= m[1:-1, :-2] + m[1:-1, 2:] + m[:-2, 1:-1] + m[2:, 1:-1] - 4 * l0
        for side in (W, E, S, N):
            if grid.neighbour(b, side)[0] in ("coarse", "fine"):
                gh = m[1:-1, 0] if side == W else m[1:-1, -1] if side == E else m[0, 1:-1] if side == S else m[-1, 1:-1]
                ed = l0[:, 0] if side == W else l0[:, 7] if side == E else l0[0, :] if side == S else l0[7, :]
                faces[(b, side)] = gh - ed
    # fillcases: fine -> coarse (fillcase0), then coarse face -> edge cells, x faces before y faces (fillcase1)
    for (b, side), fl in list(faces.items()):
        nbh = grid.neighbour(b, side)
        if nbh[0] != "coarse":
            continue
        cb = nbh[1]
        l, bi, bj = (int(v) for v in grid.blocks[b])
        half = (bj % 2) if side < 2 else (bi % 2)
        cf = faces[(cb, side ^ 1)]
        for q in range(4):
            cf[4 * half + q] += fl[2 * q] + fl[2 * q + 1]
    for pass_sides in ((W, E), (S, N)):
        for (b, side), fl in faces.items():
            if side not in pass_sides or grid.neighbour(b, side)[0] != "fine":
                continue
            if side == W:
                out[b][:, 0] += fl
            elif side == E:
                out[b][:, 7] += fl
            elif side == S:
                out[b][0, :] += fl
            else:
                out[b][7, :] += fl
    return out.reshape(nb, BS * BS)


def vorticity_amr(grid, vel):
    """KernelVorticity (main.cpp:3343-3366) on the adapted grid: (1/2h)(u_S - u_N + v_E - v_W)"""
    nb = len(grid.blocks)
    u = np.ascontiguousarray(vel[..., 0])
    v = np.ascontiguousarray(vel[..., 1])
    out = np.empty((nb, BS, BS))
    for b in range(nb):
        mu = lab1(grid, u, b, comp_sign=(-1.0, 1.0))
        mv = lab1(grid, v, b, comp_sign=(1.0, -1.0))
        i2h = 0.5 / grid.h(int(grid.blocks[b][0]))
        out[b] = i2h * (mu[:-2, 1:-1] - mu[2:, 1:-1] + mv[1:-1, 2:] - mv[1:-1, :-2])
    return out.reshape(nb, BS * BS)


def _flux_correct(grid, out, faces):
    """fillcases for a scalar field (main.cpp:1767-1849): fillcase0 (fine pair sums into the coarse face), then
    fillcase1 (coarse face into the edge cells), W/E before S/N"""
    for (b, side), fl in list(faces.items()):
        nbh = grid.neighbour(b, side)
        if nbh[0] != "coarse":
            continue
        l, bi, bj = (int(v) for v in grid.blocks[b])
        half = (bj % 2) if side < 2 else (bi % 2)
        cf = faces[(nbh[1], side ^ 1)]
        for q in range(4):
            cf[4 * half + q] += fl[2 * q] + fl[2 * q + 1]
    for pass_sides in ((W, E), (S, N)):
        for (b, side), fl in faces.items():
            if side not in pass_sides or grid.neighbour(b, side)[0] != "fine":
                continue
            if side == W:
                out[b][:, 0] += fl
            elif side == E:
                out[b][:, 7] += fl
            elif side == S:
                out[b][0, :] += fl
            else:
                out[b][7, :] += fl


def pressure_rhs_amr(grid, vel, udef, chi, dt):
    """pressure_rhs (main.cpp:6105-6206) + fillcases on the adapted grid"""
    nb = len(grid.blocks)
    out = np.empty((nb, BS, BS))
    faces = {}
    comps = [np.ascontiguousarray(a[..., k]) for a in (vel, udef) for k in (0, 1)]
    for b in range(nb):
        h = grid.h(int(grid.blocks[b][0]))
        facDiv = 0.5 * h / dt
        vx = lab1(grid, comps[0], b, comp_sign=(-1.0, 1.0))
        vy = lab1(grid, comps[1], b, comp_sign=(1.0, -1.0))
        ux = lab1(grid, comps[2], b, comp_sign=(-1.0, 1.0))
        uy = lab1(grid, comps[3], b, comp_sign=(1.0, -1.0))
        ch = chi[b].reshape(BS, BS)
        out[b] = facDiv * (vx[1:-1, 2:] - vx[1:-1, :-2] + vy[2:, 1:-1] - vy[:-2, 1:-1]) - \
            facDiv * ch * (ux[1:-1, 2:] - ux[1:-1, :-2] + uy[2:, 1:-1] - uy[:-2, 1:-1])
        for side in (W, E, S, N):
            if grid.neighbour(b, side)[0] not in ("coarse", "fine"):
                continue
            v, u = (vx, ux) if side < 2 else (vy, uy)
            if side == W:
                g_v, e_v, g_u, e_u, che = v[1:-1, 0], v[1:-1, 1], u[1:-1, 0], u[1:-1, 1], ch[:, 0]
            elif side == E:
                g_v, e_v, g_u, e_u, che = v[1:-1, -1], v[1:-1, -2], u[1:-1, -1], u[1:-1, -2], ch[:, 7]
            elif side == S:
                g_v, e_v, g_u, e_u, che = v[0, 1:-1], v[1, 1:-1], u[0, 1:-1], u[1, 1:-1], ch[0, :]
            else:
                g_v, e_v, g_u, e_u, che = v[-1, 1:-1], v[-2, 1:-1], u[-1, 1:-1], u[-2, 1:-1], ch[7, :]
            if side % 2 == 0:
                faces[(b, side)] = facDiv * (g_v + e_v) - (facDiv * che) * (g_u + e_u)
            else:
                faces[(b, side)] = -facDiv * (g_v + e_v) + (facDiv * che) * (g_u + e_u)
    _flux_correct(grid, out, faces)
    return out.reshape(nb, BS * BS)


def pressure_correction_amr(grid, pres, dt):
    """pressureCorrectionKernel (main.cpp:6021-6043) on the adapted grid; its face arrays are not consumed by the
    reference's fillcases call (main.cpp:7174-7179 passes tmp's buffers), so there is no correction to restate"""
    nb = len(grid.blocks)
    out = np.empty((nb, BS, BS, 2))
    for b in range(nb):
        h = grid.h(int(grid.blocks[b][0]))
        pFac = -0.5 * dt * h
        m = lab1(grid, pres, b)
        out[b, ..., 0] = pFac * (m[1:-1, 2:] - m[1:-1, :-2])
        out[b, ..., 1] = pFac * (m[2:, 1:-1] - m[:-2, 1:-1])
    return out.reshape(nb, BS * BS, 2)


# ---- halo-3 vector lab (KernelAdvectDiffuse, Stencil{-3,-3,4,4,true}): closed forms of the cross ghosts -------------
def LE(a, b, c):
    """main.cpp:2211-2218"""
    kappa = ((4.0 / 15.0) * a + (6.0 / 15.0) * c) + (-10.0 / 15.0) * b
    lam = (b - c) - kappa
    return (9.0 * kappa + 3.0 * lam) + c


def _test_interp(C, dx, dy):
    """TestInterp main.cpp:2219-2230 on a 3x3 stencil C[i][j] (i along x)"""
    dudx = 0.5 * (C[2][1] - C[0][1])
    dudy = 0.5 * (C[1][2] - C[1][0])
    dudxdy = 0.25 * ((C[0][0] + C[2][2]) - (C[2][0] + C[0][2]))
    dudx2 = (C[0][1] + C[2][1]) - 2.0 * C[1][1]
    dudy2 = (C[1][0] + C[1][2]) - 2.0 * C[1][1]
    return (C[1][1] + (dx * dudx + dy * dudy)) + (((0.5 * dx * dx) * dudx2 + (0.5 * dy * dy) * dudy2) + (dx * dy) * dudxdy)


def lab3_cross(grid, vel, b):
    """The 14x14x2 tile of block b with the CROSS ghosts (3 layers per side) as BlockLab::load/post_load leaves them
    (corners are not read by KernelAdvectDiffuse and stay NaN here).  Closed forms derived from the literal
    transcription oracle/amr_lab.py, pinned against the reference's own tiles (tests/test_amr.py):
      wall    edge cell, wall-normal component negated                                    (main.cpp:3131-3204)
      same    the neighbour's cells
      finer   2x2 means; on W/E faces the first of every four rows pairs fine rows 0 and 2 (main.cpp:2528-2531)
      coarser layers 1, 2: quadratic ALONG the face through the coarse cells of the block's span, then LI / LE
              ACROSS it with the two interior cells; layer 3: TestInterp on the 3x3 coarse cells around it -- taken
              from component 0 for BOTH components (main.cpp:2753-2763 passes Test without the component offset)
    The coarse cell one step beyond the span's outer end comes from the block across the coarse neighbour's
    tangential side: wall (mirrored, x-walls negate component 0), one level coarser (its cell), or this level (2x2 mean)."""
    own = vel[b].reshape(BS, BS, 2)
    m = np.full((BS + 6, BS + 6, 2), np.nan)
    m[3:11, 3:11] = own
    l, bi, bj = (int(v) for v in grid.blocks[b])

    def setg(side, k, q, val):  # layer k = 0 nearest
        if side == W:
            m[3 + q, 2 - k] = val
        elif side == E:
            m[3 + q, 11 + k] = val
        elif side == S:
            m[2 - k, 3 + q] = val
        else:
            m[11 + k, 3 + q] = val

    for side in (W, E, S, N):
        nb = grid.neighbour(b, side)
        for q in range(8):
            e1 = own[q, 0] if side == W else own[q, 7] if side == E else own[0, q] if side == S else own[7, q]
            e2 = own[q, 1] if side == W else own[q, 6] if side == E else own[1, q] if side == S else own[6, q]
            if nb[0] == "wall":
                v = e1 * (np.array([-1.0, 1.0]) if side < 2 else np.array([1.0, -1.0]))
                for k in range(3):
                    setg(side, k, q, v)
            elif nb[0] == "same":
                o = vel[nb[1]].reshape(BS, BS, 2)
                for k in range(3):
                    setg(side, k, q, o[q, 7 - k] if side == W else o[q, k] if side == E else o[7 - k, q] if side == S else o[k, q])
            elif nb[0] == "fine":
                a, t = q >> 2, q & 3
                f = vel[nb[1 + a]].reshape(BS, BS, 2)
                for k in range(3):
                    if side < 2:
                        c0 = 2 * k if side == E else 6 - 2 * k
                        y0, y1 = (0, 2) if t == 0 else (2 * t, 2 * t + 1)
                        v = (f[y0, c0] + f[y1, c0] + f[y0, c0 + 1] + f[y1, c0 + 1]) / 4
                    else:
                        y = 2 * k if side == N else 6 - 2 * k
                        v = (f[y, 2 * t] + f[y + 1, 2 * t] + f[y, 2 * t + 1] + f[y + 1, 2 * t + 1]) / 4
                    setg(side, k, q, v)
            else:
                cb = vel[nb[1]].reshape(BS, BS, 2)
                half = (bj % 2) if side < 2 else (bi % 2)
                # layers 1, 2
                if side < 2:
                    col = 7 if side == W else 0
                    cc = cb[4 * half:4 * half + 4, col]
                else:
                    row = 7 if side == S else 0
                    cc = cb[row, 4 * half:4 * half + 4]
                tq = np.array([_tangential(cc[:, d])[q] for d in (0, 1)])
                setg(side, 0, q, np.array([LI(tq[d], e1[d], e2[d]) for d in (0, 1)]))
                setg(side, 1, q, np.array([LE(tq[d], e1[d], e2[d]) for d in (0, 1)]))
                # layer 3: 3x3 coarse stencil, component 0
                tside = (S if half == 0 else N) if side < 2 else (W if half == 0 else E)  # outer end of the span
                ext = grid.neighbour(nb[1], tside)

                def coarse0(X, Y):
                    """component 0 of the coarse cell at (X, Y) in the coarse neighbour's own cell coordinates"""
                    if 0 <= X < BS and 0 <= Y < BS:
                        return cb[Y, X, 0]
                    if ext[0] == "wall":
                        if side < 2:
                            return cb[min(max(Y, 0), 7), X, 0]       # y-wall: component 0 copied
                        return -cb[Y, min(max(X, 0), 7), 0]          # x-wall: component 0 negated
                    if ext[0] == "same":                              # same level as the coarse neighbour
                        eb = vel[ext[1]].reshape(BS, BS, 2)
                        return eb[Y % BS, X % BS, 0]
                    if ext[0] == "fine":                              # this block's level: 2x2 mean
                        if side < 2:
                            eb = vel[ext[1 + (1 if side == W else 0)]].reshape(BS, BS, 2)
                            Xc = X - (4 if side == W else 0)
                            y0 = 6 if Y < 0 else 0
                            return (eb[y0, 2 * Xc, 0] + eb[y0 + 1, 2 * Xc, 0] + eb[y0, 2 * Xc + 1, 0] + eb[y0 + 1, 2 * Xc + 1, 0]) / 4
                        eb = vel[ext[1 + (1 if side == S else 0)]].reshape(BS, BS, 2)
                        Yc = Y - (4 if side == S else 0)
                        x0 = 6 if X < 0 else 0
                        return (eb[2 * Yc, x0, 0] + eb[2 * Yc + 1, x0, 0] + eb[2 * Yc, x0 + 1, 0] + eb[2 * Yc + 1, x0 + 1, 0]) / 4
                    return np.nan

                if side < 2:
                    Xc = 1 if side == E else 6
                    Yc = 4 * half + (q >> 1)
                    dx = -0.25 if side == E else 0.25
                    dy = 0.25 * (2 * (q & 1) - 1)
                else:
                    Yc = 1 if side == N else 6
                    Xc = 4 * half + (q >> 1)
                    dy = -0.25 if side == N else 0.25
                    dx = 0.25 * (2 * (q & 1) - 1)
                C = [[coarse0(Xc - 1 + i, Yc - 1 + j) for j in range(3)] for i in range(3)]
                v3 = _test_interp(C, dx, dy)
                setg(side, 2, q, np.array([v3, v3]))
    return m


def advect_diffuse_amr(grid, vel, nu, dt):
    """KernelAdvectDiffuse (main.cpp:5441-5572) on the adapted grid with its flux correction (prepare0 / fillcases with
    dim = 2, main.cpp:6611-6617): per block the functor on the interpolated tile, face arrays dfac (edge - ghost) on
    every coarse-fine face, then the coarse side's edge cells += own face + the two fine faces."""
    import ctypes
    from . import oracle as O
    lib = O.lib()
    nb = len(grid.blocks)
    out = np.empty((nb, BS, BS, 2))
    faces = {}
    dfac = nu * dt
    dp = ctypes.POINTER(ctypes.c_double)
    for b in range(nb):
        m = np.ascontiguousarray(np.nan_to_num(lab3_cross(grid, vel, b), nan=0.0))  # corners are never read
        o = np.empty((BS, BS, 2))
        lib.oracle_advect_diffuse_lab(m.ctypes.data_as(dp), ctypes.c_double(grid.h(int(grid.blocks[b][0]))), ctypes.c_double(nu),
                                      ctypes.c_double(dt), o.ctypes.data_as(dp))
        out[b] = o
        for side in (W, E, S, N):
            if grid.neighbour(b, side)[0] in ("coarse", "fine"):
                if side == W:
                    ed, gh = m[3:11, 3], m[3:11, 2]
                elif side == E:
                    ed, gh = m[3:11, 10], m[3:11, 11]
                elif side == S:
                    ed, gh = m[3, 3:11], m[2, 3:11]
                else:
                    ed, gh = m[10, 3:11], m[11, 3:11]
                faces[(b, side)] = dfac * (ed - gh)  # (8, 2)
    for (b, side), fl in list(faces.items()):
        nbh = grid.neighbour(b, side)
        if nbh[0] != "coarse":
            continue
        l, bi, bj = (int(v) for v in grid.blocks[b])
        half = (bj % 2) if side < 2 else (bi % 2)
        cf = faces[(nbh[1], side ^ 1)]
        for q in range(4):
            cf[4 * half + q] += fl[2 * q] + fl[2 * q + 1]
    # fillcase1 runs once per RECEIVED face, i.e. twice per coarse face (one per fine child), and for dim = 2 its
    # memset(&CoarseFace[i2], 0, dim * sizeof(Real)) (main.cpp:1660, 1668) clears entries 0..8 only: the second run adds
    # entries 9..15 (position 4 component 1 and positions 5..7) AGAIN.  Parity is with the reference as it is.
    again = (2 * np.arange(8)[:, None] + np.arange(2)[None, :]) >= 9
    for pass_sides in ((W, E), (S, N)):
        for (b, side), fl in faces.items():
            if side not in pass_sides or grid.neighbour(b, side)[0] != "fine":
                continue
            edge = out[b][:, 0] if side == W else out[b][:, 7] if side == E else out[b][0, :] if side == S else out[b][7, :]
            edge += fl
            edge[again] += fl[again]
    return out.reshape(nb, BS * BS, 2)
