"""Multi-process tests of the domain-decomposed path.
CPU (gloo, world_size 2 and 4): plan + exchange + all-reduce callbacks.
GPU (-m gpu; 2 and 4 ranks sharing the one GPU of the box, gloo host-staged): full HIP path vs the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def launch(mode, world, px, py, nbx, nby, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, mode, str(px), str(py), str(nbx), str(nby)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=timeout, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "DIST_OK" in out, out[-4000:]


@pytest.mark.parametrize("world,px,py,nbx,nby", [(2, 2, 1, 4, 6), (2, 1, 2, 5, 3), (4, 2, 2, 4, 4)])
def test_plan_and_exchange_cpu_gloo(world, px, py, nbx, nby):
    launch("cpu", world, px, py, nbx, nby, 29611 + world + px)


def test_cartesian_dims():
    from cup2d_amd.distributed import cartesian_dims
    assert cartesian_dims(1) == (1, 1) and cartesian_dims(2) == (1, 2)
    assert cartesian_dims(4) == (2, 2) and cartesian_dims(8) == (2, 4)
    for w in (3, 6, 12):
        px, py = cartesian_dims(w)
        assert px * py == w


@pytest.mark.gpu
@pytest.mark.parametrize("world,px,py,nbx,nby", [(2, 2, 1, 8, 16), (2, 1, 2, 16, 8), (4, 2, 2, 8, 8)])
def test_decomposed_step_matches_global_oracle_gpu(world, px, py, nbx, nby):
    launch("gpu", world, px, py, nbx, nby, 29711 + world + px, timeout=900)
