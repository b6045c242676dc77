# per-kernel average durations of the default bench workload (GPU box): bash tools/kernel_times.sh [bench args]
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ktimes
rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-f32-side --no-hard-side --no-cabi-side "$@" > $O/kt.log 2>&1
python3 - <<'PY'
import csv,glob,os,sys
fs=glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ktimes/kt/**/*kernel_stats.csv"), recursive=True)
if not fs: sys.exit("no rocprofv3 CSV under $GRAFT_REPO_ROOT/gpurun_out: did the profiled run fail? see the .log next to it")
f=fs[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0].replace("void ","").replace("haf::","")
    if float(r["Percentage"])>0.3: print("%-28s calls %3s avg %10.3f ms  %6s%%" % (n, r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"]))
PY
