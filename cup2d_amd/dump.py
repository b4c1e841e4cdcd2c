"""The reference's field output (dump(), main.cpp:3367-3466) for the block grid of this package, so that
the reference's post.py renders both implementations and field diffs are a byte comparison
(SURVEY.md 8f item 1).  Three files per dump, `<path>.xyz.raw`, `<path>.attr.raw`, `<path>.xdmf2`:

  xyz.raw   float32 [ncell][4 corners][2]: (x0,y0) (x0,y1) (x1,y1) (x1,y0) of every cell, blocks in grid order
            (the reference's Hilbert order), cells row-major inside a block, x0 = origin + h*ix in double then
            rounded to float, x1 = x0 + h                                        (main.cpp:3431-3449)
  attr.raw  float32 [ncell][3]: (u, v, 0)                                         (main.cpp:3450-3452)
  xdmf2     the XDMF 2.0 sidecar, character for character                          (main.cpp:3390-3425)

Host-side I/O on a field already downloaded from the device: a data format at the edge of the path, not a
kernel.  One rank writes one file set (the reference's MPI-IO offsets are the block order of the ranks'
Hilbert ranges; a decomposed run gathers patches into global row-major first, cup2d_amd/distributed.py)."""
import os

import numpy as np

BS = 8

_XDMF = (
    "<Xdmf\n"
    "    Version=\"2.0\">\n"
    "  <Domain>\n"
    "    <Grid>\n"
    "      <Time Value=\"%.16e\"/>\n"
    "      <Topology\n"
    "          Dimensions=\"%d\"\n"
    "          TopologyType=\"Quadrilateral\"/>\n"
    "     <Geometry\n"
    "         GeometryType=\"XY\">\n"
    "       <DataItem\n"
    "           Dimensions=\"%d 2\"\n"
    "           Format=\"Binary\">\n"
    "         %s\n"
    "       </DataItem>\n"
    "     </Geometry>\n"
    "       <Attribute\n"
    "           AttributeType=\"Vector\"\n"
    "           Name=\"vort\"\n"
    "           Center=\"Cell\">\n"
    "         <DataItem\n"
    "             Dimensions=\"3 %d\"\n"
    "             Format=\"Binary\">\n"
    "           %s\n"
    "         </DataItem>\n"
    "       </Attribute>\n"
    "    </Grid>\n"
    "  </Domain>\n"
    "</Xdmf>\n"
)


def dump_arrays(grid, vel_slab, h0, level=0):
    """(xyz float32 [ncell, 8], attr float32 [ncell, 3]) for a velocity slab [nblocks][64*2] in grid order.
    h0 = cell size of level 0 (main.cpp:6338), level = refinement level of the (uniform) grid:
    h = h0 / 2^level, block origin = index * 8 * h0 / 2^level (main.cpp:693-696), in the reference's own
    operation order so that the float32 values are bit-identical."""
    nb = grid.nblocks
    h = h0 / (1 << level)
    vel = np.asarray(vel_slab, dtype=np.float64).reshape(nb, BS * BS, 2)
    ox = grid.coords[:, 0].astype(np.float64) * BS * h0 / (1 << level)
    oy = grid.coords[:, 1].astype(np.float64) * BS * h0 / (1 << level)
    ix = np.tile(np.arange(BS), BS).astype(np.float64)
    iy = np.repeat(np.arange(BS), BS).astype(np.float64)
    u0 = ox[:, None] + h * ix[None, :]
    v0 = oy[:, None] + h * iy[None, :]
    u1, v1 = u0 + h, v0 + h
    xyz = np.stack([u0, v0, u0, v1, u1, v1, u1, v0], axis=-1).astype(np.float32).reshape(nb * BS * BS, 8)
    attr = np.zeros((nb, BS * BS, 3), dtype=np.float32)
    attr[..., 0] = vel[..., 0]
    attr[..., 1] = vel[..., 1]
    return xyz, attr.reshape(nb * BS * BS, 3)


def dump(path, time, grid, vel_slab, h0, level=0):
    """write <path>.xyz.raw, <path>.attr.raw, <path>.xdmf2 exactly as the reference's dump() does"""
    xyz, attr = dump_arrays(grid, vel_slab, h0, level)
    ncell = xyz.shape[0]
    xyz_path, attr_path = path + ".xyz.raw", path + ".attr.raw"
    xyz.tofile(xyz_path)
    attr.tofile(attr_path)
    with open(path + ".xdmf2", "w") as f:
        f.write(_XDMF % (time, ncell, 4 * ncell, os.path.basename(xyz_path), ncell, os.path.basename(attr_path)))


def read_dump(path):
    """inverse, as post.py reads it: (time, xyz [ncell, 4, 2], attr [ncell, 3])"""
    import re
    import xml.etree.ElementTree
    path = re.sub(r"\.(xdmf2|attr\.raw|xyz\.raw)$", "", path)
    time = float(xml.etree.ElementTree.parse(path + ".xdmf2").find("Domain/Grid/Time").get("Value"))
    xyz = np.fromfile(path + ".xyz.raw", dtype=np.float32)
    ncell = xyz.size // 8
    attr = np.fromfile(path + ".attr.raw", dtype=np.float32).reshape(ncell, -1)
    return time, xyz.reshape(ncell, 4, 2), attr
