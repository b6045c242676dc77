# full-size time of k_features_serial<2> for ablation variants whose garbage attributes make the engine drop the screening pass after
# the first request (tools/ablate_h.sh build first): bash tools/ablate_feat.sh "0 5"
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for n in ${1:-0 5}; do
    O=$GRAFT_REPO_ROOT/gpurun_out/ablf_$n
    rm -rf $O; mkdir -p $O
    HAF_NO_CALIBRATE=1 HAF_TESTLIB=$GRAFT_REPO_ROOT/haf_grasping_amd/variants/libhafgrasp_testing_abl$n.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/seed_sweep.py --seeds 42 --no-ab --steps 1 > $O/run.log 2>&1
    python3 - $O $n <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_features_serial<2>" in r["Name"]:
        print("variant %s  k_features_serial<2>  calls %s  max %.1f us  avg %.1f us" % (sys.argv[2], r["Calls"], float(r["MaxNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
done
