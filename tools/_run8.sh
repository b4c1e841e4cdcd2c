for w in 0 2 4 8 16 32; do echo "CHUNK=$w"; CUP2D_ADVECT_CHUNK=$w python tools/gpu_advect_only.py 4096 5; done
CUP2D_ADVECT_CHUNK=4 python tools/gpu_advect_only.py 4096 3 strict
CUP2D_PRECOND=fd timeout 300 python tools/gpu_quick.py 2>&1 | tail -8
