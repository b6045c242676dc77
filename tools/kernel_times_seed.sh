# per-kernel times (rocprofv3 --kernel-trace --stats) of the C5 step for one model seed: bash tools/kernel_times_seed.sh 11
# (or: bash tools/kernel_times_seed.sh trained | hard)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-11}
O=$GRAFT_REPO_ROOT/gpurun_out/kt_seed$S
rm -rf $O; mkdir -p $O
case "$S" in
  trained) ARGS="--seeds , --trained" ;;
  hard) ARGS="--seeds , --hard" ;;
  *) ARGS="--seeds $S" ;;
esac
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/seed_sweep.py $ARGS --no-ab --steps 4 > $O/run.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-60s calls %5s  avg %9.1f us  total %8.2f ms  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
tail -2 $O/run.log
