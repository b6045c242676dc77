# Reproduces profiles/ for one round on the GPU box: bash tools/profile_round.sh   (about 4 GPU-minutes)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
# (the profiled runs take ONE model seed -- the median seed of the default run above, 42 -- so that the kernel statistics are those
# of the headline line; the calibration request of haf_create adds one small launch of every kernel to the call counts)
B="bench.py --steps 3 --warmup 1 --seeds ${SEED:-42} --no-label-stats --no-cpu-baseline --no-latency --no-f32-side --no-hard-side --no-trained-side --no-cabi-side"
B1="bench.py --steps 1 --warmup 0 --seeds ${SEED:-42} --no-label-stats --no-cpu-baseline --no-latency --no-f32-side --no-hard-side --no-trained-side --no-cabi-side"
# default (screened) mode: kernel trace + PMC passes (counters in their own runs)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $B1 > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $B1 > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 $B1 > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/lds -- python3 $B1 > $O/lds.log 2>&1
# vector-L1 side of the feature kernels (two counters per pass: more do not fit the TA / TCP blocks)
timeout 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum --output-format csv -d $O/ta -- python3 $B1 > $O/ta.log 2>&1
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/tcp -- python3 $B1 > $O/tcp.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sqf -- python3 $B1 > $O/sqf.log 2>&1
# the other two contraction modes: kernel trace + traffic
for M in f16x3 f32; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$M -- python3 $B --precision $M > $O/kt_$M.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$M -- python3 $B1 --precision $M > $O/fetch_$M.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$M -- python3 $B1 --precision $M > $O/write_$M.log 2>&1
  python3 tools/kernel_avg.py "$(ls -t $O/kt_$M/*/*kernel_trace.csv | head -1)" -o $O/kernel_avg_$M.json > $O/kernel_avg_$M.txt 2>&1
done
# ---- round 4 ----
# full-size averages per kernel of the default-mode trace (bench.py reads them back: roofline.frac_rocprof)
python3 tools/kernel_avg.py "$(ls -t $O/kt/*/*kernel_trace.csv | head -1)" -o $O/kernel_avg.json > $O/kernel_avg.txt 2>&1
# the hardest seed of the five (11) and the trained 8964-SV model: kernel statistics of the C5 step (tools/seed_sweep.py, 4 steps)
for S in 11 trained hard; do
  case "$S" in trained) ARGS="--seeds , --trained" ;; hard) ARGS="--seeds , --hard" ;; *) ARGS="--seeds $S" ;; esac
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_s$S -- python3 tools/seed_sweep.py $ARGS --no-ab --steps 4 > $O/kt_s$S.log 2>&1
  python3 tools/kernel_avg.py "$(ls -t $O/kt_s$S/*/*kernel_trace.csv | head -1)" -o $O/kernel_avg_s$S.json > $O/kernel_avg_s$S.txt 2>&1
done
# seed 11: SQ / LDS counters of the tier kernels (k_svm_screen<2>, k_recheck_i8, k_features_serial<2>, k_features<2, 16>)
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_s11 -- python3 tools/seed_sweep.py --seeds 11 --no-ab --steps 1 > $O/sq_s11.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/lds_s11 -- python3 tools/seed_sweep.py --seeds 11 --no-ab --steps 1 > $O/lds_s11.log 2>&1
# round 5: the trained model's own counter passes (bench.py hands out trained_model.roofline.traffic / mfma_busy only from THESE: profiles/index.json)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${C}_strained -- python3 tools/seed_sweep.py --seeds , --trained --no-ab --steps 1 > $O/${C}_strained.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq_strained -- python3 tools/seed_sweep.py --seeds , --trained --no-ab --steps 1 > $O/sq_strained.log 2>&1
# the latency half of the metric: BASELINE C2 and C3 requests under the kernel trace (33 requests each)
for C in C2 C3; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat_$C -- python3 tools/latency_breakdown.py --only $C --no-profile > $O/lat_$C.log 2>&1
done
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
cut -c1-600 $O/bench_default.json
