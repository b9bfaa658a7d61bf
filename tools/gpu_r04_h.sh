#!/bin/bash
# (-> profiles/r04/exp/e2_wide_fetch_pmc_mem.txt)
# Round 4: the vector-memory counters of the quad-coalesced record fetch (the library of commit c263347, build/lib_wide) on the
# 10^6-sphere frame and on irreg's batch launch, against the same library with wide=0 -- the counters behind the dead end e2.
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=$PWD/build/lib_wide:$LD_LIBRARY_PATH
bash profiles/r03/exp/pmc_mem.sh r04h "-s big -n 2000 -m 2000 -r 4" "-s big -n 2000 -m 2000 -r 4 -o wide=1" "-s irreg -n 1000 -m 1000 -r 0 -B 20" "-s irreg -n 1000 -m 1000 -r 0 -B 20 -o wide=1"
echo r04h done
