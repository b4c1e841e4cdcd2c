import os
import subprocess
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))  # see oracle/oracle.py

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# GPU tests of these files run one per child process (see pytest_runtest_protocol below): they hold the big grids
# (4096^2, 8192^2, the 38 692- and 63 k-block adapted grids) and the multi-process runs.  A GPU memory fault is a SIGABRT
# of the process that owns the context; in a child it costs that one test (reported as failed, with the child's output),
# not the whole session -- round 2's driver run lost all 104 tests to one abort.
ISOLATED_FILES = ("test_amr.py", "test_baseline_sizes_gpu.py", "test_distributed.py")
CHILD_ENV = "CUP2D_TEST_CHILD"
CHILD_TIMEOUT_S = 900


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _say(config, text):
    """to the REAL stdout and stderr (capture suspended), flushed: the tail of an aborted run names the test"""
    capman = config.pluginmanager.getplugin("capturemanager")
    def out():
        for stream in (sys.stdout, sys.stderr):
            try:
                stream.write(text)
                stream.flush()
            except Exception:
                pass
    if capman is not None:
        with capman.global_and_fixture_disabled():
            out()
    else:
        out()


def pytest_runtest_logstart(nodeid, location):
    cfg = _CONFIG[0]
    if cfg is None:
        return
    expr = (cfg.getoption("markexpr", "") or "").strip()
    if (("gpu" in expr and "not gpu" not in expr) or os.environ.get(CHILD_ENV)) and not os.environ.get("CUP2D_TEST_QUIET"):
        _say(cfg, "\n[cup2d] start %s\n" % nodeid)


_CONFIG = [None]


def pytest_sessionstart(session):
    _CONFIG[0] = session.config


def _isolated(item):
    return (not os.environ.get(CHILD_ENV) and item.get_closest_marker("gpu") is not None
            and os.path.basename(str(item.fspath)) in ISOLATED_FILES)


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_protocol(item, nextitem):
    """GPU tests of ISOLATED_FILES: `python -m pytest <node id>` in a child; its return code is the verdict."""
    if not _isolated(item):
        return None
    from _pytest.reports import TestReport
    ihook = item.ihook
    ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    env = dict(os.environ)
    env[CHILD_ENV] = "1"
    # the parent's output / warning options travel with the node id (-s, -v, -r*, --tb=, -W, -o name=value, --durations=);
    # selection (-k, -m, --deselect, paths, node ids) was applied when the parent collected and is not repeated
    passed = [a for a in getattr(item.config.invocation_params, "args", ()) if isinstance(a, str)]
    opts, k = [], 0
    while k < len(passed):
        a = passed[k]
        if a in ("-o", "-W") and k + 1 < len(passed):
            opts += [a, passed[k + 1]]
            k += 2
            continue
        if a == "-s" or a.startswith(("-v", "-r", "--tb=", "-W", "--durations=", "--showlocals", "--capture=", "-o")) and a != "-o":
            opts.append(a)
        k += 1
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "--rootdir", str(item.config.rootpath)] + opts + [item.nodeid]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, cwd=str(item.config.rootpath), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=CHILD_TIMEOUT_S)
        rc, text = r.returncode, r.stdout.decode("utf-8", "replace")
    except subprocess.TimeoutExpired as e:
        rc, text = -999, (e.stdout or b"").decode("utf-8", "replace") + "\n[cup2d] child timed out after %d s" % CHILD_TIMEOUT_S
    dt = time.time() - t0
    tail = text[-8000:]
    last = [l for l in text.strip().splitlines() if l.strip()][-1:] or [""]
    if rc == 0 and " passed" in last[0]:
        outcome, longrepr = "passed", None
        if "-s" in opts:  # the parent was asked not to capture: what the test printed in its child is shown here
            shown = [l for l in text.splitlines() if l.strip() and not l.startswith("[cup2d] start") and l.strip() != "."
                     and " passed" not in l]
            if shown:
                _say(item.config, "\n".join(shown) + "\n")
    elif rc in (0, 5) and " skipped" in last[0] and " passed" not in last[0] and " failed" not in last[0]:
        outcome, longrepr = "skipped", (str(item.fspath), 0, "skipped in the child process:\n" + tail[-1500:])
    else:
        outcome = "failed"
        longrepr = "child process `%s` ended with rc %s%s\n%s" % (
            " ".join(cmd[1:]), rc, " (killed by signal %d)" % -rc if -64 < rc < 0 else "", tail)
        _say(item.config, "\n[cup2d] FAILED in its child process (rc %s): %s\n" % (rc, item.nodeid))
    for when in ("setup", "call", "teardown"):
        rep = TestReport(item.nodeid, item.location, dict(item.keywords), outcome if when == "call" else "passed",
                         longrepr if when == "call" else None, when, (), dt if when == "call" else 0.0)
        ihook.pytest_runtest_logreport(report=rep)
    ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()  # builds liboracle.so on first use
    return O


@pytest.fixture(scope="session")
def gpu_lib():
    """libcup2d_hip.so on a machine with a GPU; fails (not skips) when the library is missing."""
    import cup2d_amd
    return cup2d_amd.load_library()
