mkdir -p gpurun_out; export TMPDIR=/tmp
./tools/fp64_peak.bin > gpurun_out/fp64_peak.log 2>&1; cat gpurun_out/fp64_peak.log
python tools/gpu_advect_only.py 4096 5
P="python tools/gpu_advect_only.py 4096 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_adv1 -o pmc -- $P > gpurun_out/pmc_adv1.log 2>&1; echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmc_adv2 -o pmc -- $P > gpurun_out/pmc_adv2.log 2>&1; echo "pmc2 rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d gpurun_out/pmc_adv3 -o pmc -- $P > gpurun_out/pmc_adv3.log 2>&1; echo "pmc3 rc=$?"
tail -3 gpurun_out/pmc_adv1.log gpurun_out/pmc_adv2.log gpurun_out/pmc_adv3.log
