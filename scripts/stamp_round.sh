#!/bin/bash
# copy the round's profile sets from gpurun_out/ into profiles/ and stamp profiles/traffic.json with the current source hash
# usage: bash scripts/stamp_round.sh r03         after `scripts/profile_set.sh r03 counters` (summaries + traffic stamp)
#        bash scripts/stamp_round.sh r03 lines   after `scripts/profile_set.sh r03 lines`    (only the bench lines, now with traffic)
T=${1:-r05}
for t in c2_stored c2_stored_3k c3 c4 c2otf c2otf_kron c1 c5_76 cx87 reortho; do
  [ -d gpurun_out/profile_${T}_$t ] || continue
  for f in bench.json kernel_stats.csv pmc_summary.csv; do
    [ "$2" = lines ] && [ $f != bench.json ] && continue
    [ "$2" = lines ] && [ $t = reortho ] && continue
    [ -f gpurun_out/profile_${T}_$t/$f ] && cp gpurun_out/profile_${T}_$t/$f profiles/${T}_${t}_$f
  done
done
[ "$2" = lines ] && { grep -l '"stale' profiles/${T}_*_bench.json && echo "STALE LINES ABOVE"; exit 0; }
python scripts/traffic_stamp.py stored hubbard_4x4_half_filling_pbc_U4 profiles/${T}_c2_stored_pmc_summary.csv "k_pb_up,k_pb_down<" &&
python scripts/traffic_stamp.py onthefly hubbard_4x4_half_filling_pbc_U4 profiles/${T}_c2otf_pmc_summary.csv "k_pb_up,k_pb_down<" &&
python scripts/traffic_stamp.py stored heisenberg_chain_L28_sz0_obc profiles/${T}_c3_pmc_summary.csv "k_pb_up_seg<true,k_pb_combine" &&
python scripts/traffic_stamp.py stored tj_4x5_9up9down_complex profiles/${T}_c4_pmc_summary.csv "true>(lpp::TjArgs)" &&
python scripts/traffic_stamp.py stored hubbard_chain_L12_half_filling_U4 profiles/${T}_c1_pmc_summary.csv k_spmv_window &&
python scripts/traffic_stamp.py onthefly hubbard_4x5_7up6down_pbc_U4 profiles/${T}_c5_76_pmc_summary.csv "k_pb_up_seg,k_pb_down<,k_pb_combine" &&
{ [ ! -f profiles/${T}_cx87_pmc_summary.csv ] || python scripts/traffic_stamp.py stored hubbard_4x4_8up7down_complex_U4 profiles/${T}_cx87_pmc_summary.csv "k_pb_up_big,k_pb_down<,k_pb_combine"; } &&
{ [ ! -f profiles/${T}_c2otf_kron_pmc_summary.csv ] || python scripts/traffic_stamp.py onthefly_kron hubbard_4x4_half_filling_pbc_U4 profiles/${T}_c2otf_kron_pmc_summary.csv k_spmv_kron; }
