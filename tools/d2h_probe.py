import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracers_amd import api
ctx = api.Context(0)
sc = ctx.scene("rgbbox")
ps = api.prepare_scene(1000, 1000, sc)
buf = ctx.alloc_i32(1000 * 1000)
api.render_into(buf.ptr, 1000, 1000, ps); ctx.sync()
for i in range(5):
    t0 = time.perf_counter(); a = buf.to_host((1000, 1000)); t1 = time.perf_counter()
    print(f"to_host 4 MB: {1e6*(t1-t0):.0f} us")
