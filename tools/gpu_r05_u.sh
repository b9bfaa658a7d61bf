#!/bin/bash
# Round 5: the model's constants once more, with the bulk zipped.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05u; mkdir -p $OUT
export AB_TIMEOUT=60
{
for rep in 1 2; do
echo "new|rgbbox|1000|-r 20|"
for v in 180 300 360; do echo "new|rgbbox|1000|-r 20|px_g64=$v"; done
for v in 200 300; do echo "new|rgbbox|1000|-r 20|px_ray_ns=$v"; done
echo "new|rgbbox|1000|-r 20|px_g32=80"
echo "new|rgbbox|1000|-r 20|px_g32=130"
echo "new|irreg|1000|-r 20|"
for v in 250 420; do echo "new|irreg|1000|-r 20|px_g64=$v"; done
echo "new|irreg|1000|-r 20|px_g32=180"
echo "new|irreg|1000|-r 20|px_g32=290"
echo "new|irreg|1000|-r 20|px_g1=38"
echo "new|irreg|1000|-r 20|px_g1=52"
done
} | bash tools/gpu_ab.sh r05u/ab > /dev/null
echo done
