#!/bin/bash
# (-> profiles/r04/exp/e10; build/lib_look* = render_kernels.hip with the look restricted at compile time)
# Round 4: VERDICT r3 item 5's "skip the look at finished folds while the box stack still holds >= 32 items": variants of the
# library with the look restricted to nbox < 16 / 32 / 48 (build/lib_look*), against the product, through the native bench
cd "$(dirname "$0")/.."
bash tools/gpu_ab.sh r04i/ab <<'AB'
new|rgbbox|1000|-r 20|
look32|rgbbox|1000|-r 20|
look48|rgbbox|1000|-r 20|
look16|rgbbox|1000|-r 20|
new|irreg|1000|-r 20|
look32|irreg|1000|-r 20|
look48|irreg|1000|-r 20|
look16|irreg|1000|-r 20|
new|rgbbox|1000|-r 0 -B 20|
look32|rgbbox|1000|-r 0 -B 20|
look48|rgbbox|1000|-r 0 -B 20|
new|irreg|1000|-r 0 -B 20|
look32|irreg|1000|-r 0 -B 20|
look48|irreg|1000|-r 0 -B 20|
new|irreg|4000|-r 5|
look32|irreg|4000|-r 5|
new|big|2000|-r 4|
look32|big|2000|-r 4|
new|rgbbox|1000|-r 20|
look32|rgbbox|1000|-r 20|
AB
echo r04i done
