# SQ / TA / TCP counters of the screening feature kernel on the default bench workload (GPU box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/feat
rm -rf $O; mkdir -p $O
B1="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-f32-side"
[ -n "$ONLY_TC" ] || timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/p1 -- python3 $B1 > $O/p1.log 2>&1
[ -n "$ONLY_TC" ] || timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/p2 -- python3 $B1 > $O/p2.log 2>&1
timeout 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum --output-format csv -d $O/p3 -- python3 $B1 > $O/p3.log 2>&1
timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/p5 -- python3 $B1 > $O/p5.log 2>&1
timeout 150 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $O/p6 -- python3 $B1 > $O/p6.log 2>&1
[ -n "$ONLY_TC" ] || timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p4 -- python3 $B1 > $O/p4.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ("p1","p2","p3","p4","p5","p6"):
    fs=glob.glob("/root/repo/gpurun_out/feat/%s/**/*counter_collection.csv"%d, recursive=True)
    if not fs: print(d, "no output"); continue
    for f in fs:
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "k_features_serial" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])
        print(d, dict(acc))
PY
