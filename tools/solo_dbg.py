import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import raytracers_amd as R
import oracle_lib as O
for scene, h, w in (("rgbbox", 200, 200), ("irreg", 200, 200), ("floor:300:1800", 160, 160), ("floor:37:222", 90, 120)):
    if scene.startswith("floor"):
        _, n, k = scene.split(":"); orc = O.OracleScene("floor", n=int(n), k=float(k))
    else:
        orc = O.OracleScene(scene)
    want, cnt = orc.render(h, w)
    for opts in (dict(), dict(gpu_build=0), dict(deep_class=5, deep_split=6, deep_cap_log2=0), dict(gpu_build=0, deep_class=5, deep_split=6, deep_cap_log2=0), dict(deep_class=8, deep_split=6, deep_cap_log2=0, grid_div=16)):
        c = R.Context(); c.set_variant(3)
        for k, v in opts.items(): c.set_option(k, v)
        sc = c.floor(int(scene.split(":")[1]), float(scene.split(":")[2])) if scene.startswith("floor") else c.scene(scene)
        ps = R.prepare_scene(h, w, sc)
        for f in range(3):
            got = R.render(h, w, ps)
            bad = np.argwhere(got != want)
            print(scene, opts, "frame", f, "bad", len(bad), [(int(y), int(x), hex(int(got[y, x])), hex(int(want[y, x]))) for y, x in bad[:4]])
        c.close()
