for w in 1 2 3 4; do echo "WGS_PER_CU=$w"; CUP2D_WGS_PER_CU=$w python tools/gpu_advect_only.py 4096 5; done
