import os, sys, time
from scratch_potrf import *   # noqa
for nb in (256, 384, 512, 640, 768, 1024):
    ctx.set_option("nb", nb); ctx.set_option("lookahead", 1); ctx.set_option("nb_big", 0)
    ts = []
    for rep in range(3):
        mat = build(); t0 = time.perf_counter(); info = mat.potrf(); ctx.sync(); ts.append(time.perf_counter() - t0); del mat
    print(f"nb={nb}: potrf {min(ts)*1e3:.2f} ms ({1.608e12/min(ts)/1e12:.1f} TF) info={info}")
