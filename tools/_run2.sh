mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu_all.log
CUP2D_PRECOND=lds timeout 600 python -m pytest tests/test_distributed.py -m gpu -q -x > gpurun_out/pytest_dist_lds.log 2>&1; echo "dist lds rc=$?"
tail -5 gpurun_out/pytest_dist_lds.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
