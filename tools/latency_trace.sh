# kernel-by-kernel timeline of one small request (C2) on the GPU box: bash tools/latency_trace.sh
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lat_${1:-C2}
rm -rf $O; mkdir -p $O
# C3T / C3R: the C3 request against the trained 8 964-SV model / the seeded 4096-SV one (tools/c3_big_model.py)
case "${1:-C2}" in
  C3T) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/c3_big_model.py trained --requests 6 > $O/kt.log 2>&1 ;;
  C3R) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/c3_big_model.py rand4096 --requests 6 > $O/kt.log 2>&1 ;;
  *) timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/latency_breakdown.py --only ${1:-C2} > $O/kt.log 2>&1 ;;
esac
python3 - ${1:-C2} <<'PY'
import csv,glob,os,sys
fs=glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "lat_%s/kt/**/*kernel_trace.csv" % (sys.argv[1] if len(sys.argv) > 1 else "C2")), recursive=True)
if not fs: sys.exit("no rocprofv3 CSV under $GRAFT_REPO_ROOT/gpurun_out: did the profiled run fail? see the .log next to it")
f=fs[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last request = from the last launch of a request's FIRST kernel (k_small_pre, or k_fill / k_bkt_count / k_bin* when the separate
# pre-stage kernels run) to the end
names=[r["Kernel_Name"].split("(")[0].replace("void ","").replace("haf::","") for r in rows]
firsts=("k_small_pre","k_fill","k_bkt_count")
starts=[i for i,n in enumerate(names) if n.startswith(firsts) and (i==0 or not names[i-1].startswith(("k_fill","k_bin")))]
i=starts[-1]
t0=int(rows[i]["Start_Timestamp"]); prev=t0
for r,n in zip(rows[i:],names[i:]):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-26s start %8.1f us  dur %7.1f us  gap %6.1f us  grid %s" % (n[:26], (s-t0)/1e3, (e-s)/1e3, (s-prev)/1e3, r.get("Grid_Size_X", r.get("Grid_Size",""))))
    prev=e
print("total %.1f us" % ((prev-t0)/1e3))
PY
