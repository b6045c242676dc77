#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/final into profiles/ (newest file of each pass) and folds the traffic
# passes: bash tools/collect_profiles.sh   (run in the repo root, in the container, after the gpurun call has returned)
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
P=profiles/${ROUND:-r03}_bench_c5_nsv4096
newest() { ls -t $1 | head -1; }
cp $F/bench_default.json ${P}_default.json
for m in "" _f16x3 _f32; do
  n=${m:-_f16s}
  cp "$(newest "$F/kt$m/*/*kernel_stats.csv")" ${P}${n}_kernel_stats.csv
  cp "$(newest "$F/fetch$m/*/*counter_collection.csv")" ${P}${n}_pmc_FETCH_SIZE.csv
  cp "$(newest "$F/write$m/*/*counter_collection.csv")" ${P}${n}_pmc_WRITE_SIZE.csv
done
cp "$(newest "$F/sq/*/*counter_collection.csv")" ${P}_f16s_pmc_SQ.csv
cp "$(newest "$F/lds/*/*counter_collection.csv")" ${P}_f16s_pmc_LDS_TCC.csv
cp "$(newest "$F/ta/*/*counter_collection.csv")" ${P}_f16s_pmc_TA.csv
cp "$(newest "$F/tcp/*/*counter_collection.csv")" ${P}_f16s_pmc_TCP.csv
cp "$(newest "$F/sqf/*/*counter_collection.csv")" ${P}_f16s_pmc_SQ_insts.csv
python tools/pmc_summary.py --fetch ${P}_f16s_pmc_FETCH_SIZE.csv --write ${P}_f16s_pmc_WRITE_SIZE.csv \
  --fetch-f32 ${P}_f32_pmc_FETCH_SIZE.csv --write-f32 ${P}_f32_pmc_WRITE_SIZE.csv \
  --fetch-f16x3 ${P}_f16x3_pmc_FETCH_SIZE.csv --write-f16x3 ${P}_f16x3_pmc_WRITE_SIZE.csv \
  --grid 512 --rolls 36 --nsv 4096 -o profiles/pmc_traffic.json
