# quick per-kernel profile of the default bench workload (GPU box): kernel trace + SQ counters of one kernel
# (KERNEL=substring of its name, default k_svm_screen): KERNEL=k_features_serial bash tools/profile_quick.sh
export KERNEL="${KERNEL:-k_svm_screen}"
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/quick
rm -rf $O; mkdir -p $O
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-f32-side --no-hard-side --no-trained-side --no-cabi-side"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/kt.log 2>&1
B1="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-f32-side --no-hard-side --no-trained-side --no-cabi-side"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 $B1 > $O/sq.log 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/sq2 -- python3 $B1 > $O/sq2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq3 -- python3 $B1 > $O/sq3.log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-8 | head -20
python3 - <<'PY'
import csv,glob,collections,os,sys
for d in ("sq","sq2","sq3"):
    fs=glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "quick", d, "**", "*counter_collection.csv"), recursive=True)
    if not fs: print("no counter CSV for pass %s under $GRAFT_REPO_ROOT/gpurun_out/quick (see %s.log)" % (d, d), file=sys.stderr)
    for f in fs:
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if os.environ["KERNEL"] in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])
        print(d, dict(acc))
PY
