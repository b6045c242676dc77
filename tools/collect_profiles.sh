#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/final into profiles/ (newest file of each pass) and folds the traffic
# passes: bash tools/collect_profiles.sh   (run in the repo root, in the container, after the gpurun call has returned)
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
R=${ROUND:-r05}
P=profiles/${R}_bench_c5_nsv4096
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
# round 4: full-size kernel averages (bench.py: roofline.frac_rocprof), seed 11 / trained model steps, tier-kernel counters, latency traces
python - <<PY
import json
d = json.load(open("$F/kernel_avg.json"))
d["workload"] = {"grid": 512, "rolls": 36, "n_sv": 4096, "seed": 42}
json.dump(d, open("profiles/${R}_kernel_avg.json", "w"), indent=1, sort_keys=True)
PY
for S in 11 trained hard; do
  cp "$(newest "$F/kt_s$S/*/*kernel_stats.csv")" profiles/${R}_seed${S}_kernel_stats.csv
  cp $F/kernel_avg_s$S.json profiles/${R}_seed${S}_kernel_avg.json
done
for M in f16x3 f32; do cp $F/kernel_avg_$M.json profiles/${R}_${M}_kernel_avg.json; done
# round 5: the trained model's own counter passes
cp "$(newest "$F/FETCH_SIZE_strained/*/*counter_collection.csv")" profiles/${R}_seedtrained_pmc_FETCH_SIZE.csv
cp "$(newest "$F/WRITE_SIZE_strained/*/*counter_collection.csv")" profiles/${R}_seedtrained_pmc_WRITE_SIZE.csv
cp "$(newest "$F/sq_strained/*/*counter_collection.csv")" profiles/${R}_seedtrained_pmc_SQ.csv
cp "$(newest "$F/sq_s11/*/*counter_collection.csv")" profiles/${R}_seed11_pmc_SQ.csv
cp "$(newest "$F/lds_s11/*/*counter_collection.csv")" profiles/${R}_seed11_pmc_LDS.csv
for C in C2 C3; do
  c=$(echo $C | tr A-Z a-z)
  cp "$(newest "$F/lat_$C/*/*kernel_stats.csv")" profiles/${R}_latency_${c}_kernel_stats.csv
done
python tools/pmc_summary.py --fetch ${P}_f16s_pmc_FETCH_SIZE.csv --write ${P}_f16s_pmc_WRITE_SIZE.csv \
  --fetch-f32 ${P}_f32_pmc_FETCH_SIZE.csv --write-f32 ${P}_f32_pmc_WRITE_SIZE.csv \
  --fetch-f16x3 ${P}_f16x3_pmc_FETCH_SIZE.csv --write-f16x3 ${P}_f16x3_pmc_WRITE_SIZE.csv \
  --grid 512 --rolls 36 --nsv 4096 -o profiles/pmc_traffic.json
# the provenance-checked index bench.py reads (roofline.kernel_ms_rocprof / frac_rocprof / mfma_busy / traffic)
python tools/profile_index.py --round ${R} --commit "$(git rev-parse --short HEAD)"
