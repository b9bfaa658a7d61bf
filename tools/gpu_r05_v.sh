#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05v; mkdir -p $OUT
export AB_TIMEOUT=60
{
for n in 300 500 700 1000 1400; do for rep in 1 2; do
echo "new|rgbbox|$n|-r 20|"
echo "new|rgbbox|$n|-r 20|px_g64=180"
echo "new|rgbbox|$n|-r 20|px_g64=180 px_g32=120"
echo "new|rgbbox|$n|-r 20|px_g64=150"
done; done
} | bash tools/gpu_ab.sh r05v/ab > /dev/null
echo done
