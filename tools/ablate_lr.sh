# Timing ablation of the low-rank sweep k_svm_screen_lr<0, true> (screen.hip: HAF_LR_ABL bits: 1 no in-loop LDS-DMA, 2 no epilogue VALU,
# 4 no B-fragment reads, 8 no tile barrier / wait, 16 no projection MFMAs).  Results of ablated variants are garbage: only kernel times count.
#   here (no GPU):   for n in 0 1 2 4 8 16 31; do python -m haf_grasping_amd.build --variant lrabl$n [--no-checks] -DHAF_LR_ABL=$n; done
#   on the GPU box:  bash tools/ablate_lr.sh "0 1 2 4 8 16 31"   -> gpurun_out/ablate_lr.txt
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ablate_lr.txt
: > $OUT
for n in ${1:-0 1 2 4 8 16 31}; do
    O=$GRAFT_REPO_ROOT/gpurun_out/abl_lr_$n
    rm -rf $O; mkdir -p $O
    HAF_ITERS=3 HAF_LIB=$GRAFT_REPO_ROOT/haf_grasping_amd/variants/libhafgrasp_${PREFIX:-lrabl}$n.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/time_svm_stage.py > $O/run.log 2>&1
    python3 - $O $n >> $OUT <<'PY'
import csv, glob, sys, collections, re
fs = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)
if not fs:
    print("variant %s: no trace (see run.log)" % sys.argv[2]); sys.exit(0)
d = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").split("::")[-1]
    if k.startswith("k_svm_screen") or k.startswith("k_features_serial") or k.startswith("k_project"):
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
print("variant %-4s " % sys.argv[2] + "  ".join("%s max %.3f ms (n %d)" % (k, max(v), len(v)) for k, v in sorted(d.items(), key=lambda kv: -max(kv[1]))[:4]))
PY
    tail -1 $OUT
done
