mkdir -p gpurun_out
for k in fd mfma lds; do
  CUP2D_PRECOND=$k timeout 300 python tools/gpu_quick.py --time > gpurun_out/quick_$k.log 2>&1; echo "quick $k rc=$?"; cat gpurun_out/quick_$k.log | tail -12
done
