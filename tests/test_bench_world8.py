"""bench.py at world = 8 before the driver's one 8-GPU run does (-m gpu): `python bench.py --gpus 8` with the eight ranks
sharing GPU 0 of the box (CUP2D_BENCH_SHARE_GPU=1: torch.distributed's gloo, host-staged, behind cup2d_set_comm -- RCCL
cannot connect two ranks on one device).  Everything of the file an 8-rank run executes is executed: the self-spawn through
torch.distributed.run, the 2 x 4 Cartesian layout of BASELINE.json configs[3] (main.cpp:6494-6504), both layouts in one
command, the verification on N ranks, the per-kernel sampling outside the timed region, the JSON merge on rank 0, the
watchdog.  The numbers of the line are not a measurement (it says "shared_gpu": true)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_at_world_8_sharing_one_gpu():
    env = dict(os.environ, CUP2D_BENCH_SHARE_GPU="1", CUP2D_BENCH_WATCHDOG_S="200", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):  # (a test that ran a one-rank process group in this
        env.pop(k, None)                                                           # process leaves them behind: bench.py would take itself for a rank)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--n", "256", "--steps", "2", "--warmup", "1",
           "--layout", "configs3", "--configs3-n", "1024", "--iters", "20", "--no-cpu-baseline"]
    env["CUP2D_BENCH_DETAIL"] = os.path.join(ROOT, "gpurun_out", "bench_detail_world8_test.json")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=840, cwd=ROOT)
    err = r.stderr.decode("utf-8", "replace")
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout.decode()[-2000:], err[-4000:])
    line = json.loads(lines[0])
    assert len(lines[0]) < 8000, len(lines[0])   # the whole line fits a record that keeps the last 8 KB of stdout
    assert line["n_gpus"] == 8 and line["summary"]["verified"]["ok"] is True and line["verified_ok"] is True
    assert line["config"]["comm"]["cartesian"] == "2x4" and len(line["config"]["comm"]["organisations_ms_per_step"]) == 4, line["config"]
    assert line["summary"]["second_layout"]["layout"] == "weak", line["summary"]
    with open(os.path.join(ROOT, line["detail"])) as f:   # everything measured: the side file the line names
        J = json.load(f)
    assert J["value"] == line["value"] and J["ms_per_step"] == line["ms_per_step"]
    assert J["n_gpus"] == 8 and J["steps"] == 2 and J["warmup"] == 1 and J["unit"] == "Mcell-updates/s" and J["value"] > 0
    assert J["scaling"] == "strong" and J["config"]["layout"] == "configs3" and J["config"]["parallelism"] == "cart2x4"
    assert J["config"]["global_grid"] == "1024x1024" and J["config"]["global_cells"] == 1024 * 1024
    comm = J["config"]["comm"]
    assert comm["shared_gpu"] is True and comm["cartesian"] == "2x4" and comm["peers_of_rank0"] == 2, comm
    org = comm["organisations"]   # the four organisations of a reduction point timed back to back (over callbacks: MERGE 2 in all)
    assert len(org) == 4 and all("error" not in v and v["ms_per_step"] > 0 and v["solver_form"][0] == "eab" for v in org.values()), org
    second = J["second_layout"]
    assert second["layout"] == "weak" and second["scaling"] == "weak" and "error" not in second, second
    assert second["cells_per_rank"] == "256x256" and second["global_cells"] == 8 * 256 * 256 and second["value"] > 0
    v = J["verified"]
    assert v["ok"] is True and v["iters"] == 20 and v["residual_reported"] <= v["residual_initial"], v
    assert J["verified_summary"]["ok"] is True
    assert J["roofline"] and J["kernels"]["sweep_EA"]["launches"] > 0   # the two-launch organisation ran on every rank
    assert "made no progress" not in err, err[-3000:]                  # the watchdog stayed silent
