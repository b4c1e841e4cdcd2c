mkdir -p gpurun_out; export TMPDIR=/tmp
./tools/fp64_peak.bin 2>&1 | head -2
CUP2D_PRECOND=fd timeout 300 python tools/gpu_quick.py --time > gpurun_out/quick_fd2.log 2>&1; echo "quick fd rc=$?"; tail -9 gpurun_out/quick_fd2.log
python tools/gpu_advect_only.py 4096 5
python tools/gpu_advect_only.py 4096 3 strict
P="python tools/gpu_advect_only.py 4096 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_adv4 -o pmc -- $P > gpurun_out/pmc_adv4.log 2>&1; echo "pmc rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_adv5 -o pmc -- $P > gpurun_out/pmc_adv5.log 2>&1; echo "pmc rc=$?"
