#!/bin/bash
# Runs tools/issue_peak op by op (a faulting op then costs only its own lines), 40 untimed launches ahead of every timed one:
# a process that starts on an idle GPU runs its first ~20 ms at a lower clock (with one warm-up launch the
# v_add-class peak read 1041 instead of 1129 G/s, profiles/r03/peak_warm.txt).  usage: gpu_issue_peak.sh <outfile>
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/issue_peak.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
first=1
for k in k_add k_mul k_fma k_max3 k_min k_add_dep k_pk_mul k_pk_add k_cndmask_vcc k_cndmask_vcc_e64 k_cmp_cnd_vcc k_cmp_cnd_sgpr k_sub k_bfe k_and k_add_u32v k_cndmask_sgpr k_cmp_vcc k_cmp_sgpr k_cmp_salu_cnd \
         k_rcp k_sqrt k_div_scale k_div_fmas k_div_fixup k_mbcnt k_lshl_add k_mov k_readfirstlane k_salu k_bcnt k_snop k_mix_v1s1 k_mix_v2s1 \
         k_bperm k_bperm_same_sel k_read_b32 k_read2st64 k_read_b64 k_read_b128 k_read_b128_b96 k_write_b32 k_add_u32 k_min_u64 k_boxmix; do
  if [ $first = 1 ]; then timeout 120 ./build/issue_peak -W 40 -k $k >> "$OUT" 2>&1; first=0
  else timeout 120 ./build/issue_peak -W 40 -k $k 2>&1 | grep -v '^#\|^op ' >> "$OUT"; fi
done
echo issue_peak done
