import sys, time
sys.path.insert(0, '.')
from oracle import oracle as O
for thr in (16, 32, 64, 128):
    t = time.time()
    try:
        r = O.ref_step_time(512, steps=2, max_iter=50, threads=thr, timeout=120)
        print(thr, r, round(time.time() - t, 1), flush=True)
    except Exception as e:
        print(thr, 'ERR', str(e)[:100], flush=True)
