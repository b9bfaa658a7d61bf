#!/bin/bash
# One-call experiment: the tile-queue layouts (xcd_queues / tpt_log2 / static_first) -- parity under both layouts, then A/B
# timings through the native bench, L2 hit rates of the 10^6-sphere scene under both layouts.  usage: gpu_exp.sh <tag>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-exp}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
# 1. the whole GPU suite on the library defaults (the generalised queue code with one shard)
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
# 2. the render-path tests again with eight shards as the context default
RT_XCD_QUEUES=1 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q \
  -k "golden_500 or pixels_bit_exact or adaptive_tile_order or parts_assemble or stacked_parts or irreg_4000 or big_2000 or multi_device_context or many_views or bounce_limit or tall_trees or batch_of_frames or repeated_launches" \
  > $OUT/pytest_xcd.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_xcd.log
tail -3 $OUT/pytest_xcd.log
# 3. A/B through the native bench
ab() {  # scene size runs label opts...
  local s=$1 n=$2 r=$3; shift 3
  local o=""; for kv in "$@"; do o="$o -o $kv"; done
  local res=$(timeout 120 ./build/rtbench -s $s -n $n -m $n -r $r $o 2>&1 | grep -E "HIP-event|Checksum|failed" | tr '\n' ' ')
  echo "$s $n [$*] : $res"
}
{
for cfg in "xcd_queues=0 tpt_log2=0 static_first=0" "xcd_queues=0 tpt_log2=0 static_first=1" "xcd_queues=0 tpt_log2=2 static_first=1" \
           "xcd_queues=0 tpt_log2=3 static_first=1" "xcd_queues=1 tpt_log2=0 static_first=1" "xcd_queues=1 tpt_log2=1 static_first=1" \
           "xcd_queues=1 tpt_log2=2 static_first=1" "xcd_queues=1 tpt_log2=2 static_first=0"; do
  ab irreg 4000 8 $cfg
  ab big 2000 5 $cfg
done
for cfg in "xcd_queues=0 tpt_log2=0 static_first=0" "xcd_queues=0 tpt_log2=0 static_first=1" "xcd_queues=1 tpt_log2=0 static_first=1" \
           "xcd_queues=1 tpt_log2=0 static_first=0" "xcd_queues=0 tpt_log2=1 static_first=1"; do
  ab rgbbox 1000 20 $cfg
  ab irreg 1000 20 $cfg
done
for cfg in "waves_per_wg=8" "waves_per_wg=12" "grid_div=2" "xcd_queues=1 waves_per_wg=8" "lds_sph_first=1"; do
  ab big 2000 5 $cfg
done
for cfg in "tpt_log2=2 static_first=0" "tpt_log2=2 static_first=1" "tpt_log2=1 static_first=1" "tpt_log2=3 static_first=1"; do
  o=""; for kv in $cfg; do o="$o -o $kv"; done
  for s in rgbbox irreg; do
    echo "$s 1000 batch20 [$cfg] : $(timeout 120 ./build/rtbench -s $s -n 1000 -m 1000 -r 0 -B 20 $o 2>&1 | grep -E "Batch|failed" | tr '\n' ' ')"
  done
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
# 4. one rank's share at world size 8 under both layouts (no exchange)
for x in 0 1; do
  echo "== RT_XCD_QUEUES=$x" >> $OUT/rank_share.txt
  RT_XCD_QUEUES=$x timeout 200 python tools/rank_share_probe.py 20 1,8 1 2 2s >> $OUT/rank_share.txt 2>&1
done
cat $OUT/rank_share.txt
# 5. L2 hit rate and memory-side traffic of the 10^6-sphere frame under both layouts (counters in their own runs)
cd /tmp
for x in 0 1; do
  i=0
  for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE SQ_INSTS_VMEM" "WRITE_SIZE SQ_WAVES"; do
    d=$OUT/pmc_big_x${x}_p$i
    timeout 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- $OLDPWD/build/rtbench -s big -n 2000 -m 2000 -r 2 -o xcd_queues=$x > $d.log 2>&1
    i=$((i+1))
  done
done
cd $OLDPWD
python - "$OUT" <<'PY' > $OUT/pmc_big.txt 2>&1
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_big_x*_p[0-9]"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            if "pooled" in k:
                print(os.path.basename(d), k, c, len(v), "mean %.0f" % (sum(v) / len(v)), "last %.0f" % v[-1])
PY
cat $OUT/pmc_big.txt
rm -rf $OUT/pmc_big_x*_p[0-9]/
echo exp done
