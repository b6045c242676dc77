cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-f32-side"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/kt.log 2>&1
B1="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-f32-side"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $B1 > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $B1 > $O/write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 $B1 > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/lds -- python3 $B1 > $O/lds.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -- python3 $B --precision f32 > $O/kt32.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch32 -- python3 $B1 --precision f32 > $O/fetch32.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write32 -- python3 $B1 --precision f32 > $O/write32.log 2>&1
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
cat $O/bench_default.json | cut -c1-600
