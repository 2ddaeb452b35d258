import os, sys, time

import numpy as np
from scratch_potrf import *   # noqa
for nb_big, mt in ((0, 96), (1024, 112), (1024, 96), (1024, 80), (1024, 64), (2048, 96), (1536, 96)):
    ctx.set_option("nb", 512); ctx.set_option("lookahead", 1); ctx.set_option("nb_big", nb_big); ctx.set_option("nb_big_min_tiles", mt)
    ts = []
    for rep in range(3):
        mat = build(); t0 = time.perf_counter(); info = mat.potrf(); ctx.sync(); ts.append(time.perf_counter() - t0); del mat
    print(f"nb_big={nb_big} min_tiles={mt}: potrf {min(ts)*1e3:.2f} ms ({1.608e12/min(ts)/1e12:.1f} TF) info={info}")
