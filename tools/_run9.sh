mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"
tail -45 gpurun_out/pytest_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
