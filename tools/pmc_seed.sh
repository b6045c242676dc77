# SQ counters per kernel of one C5 step for one model seed: bash tools/pmc_seed.sh 11
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-11}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_seed$S
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 tools/seed_sweep.py --seeds $S --no-ab --steps 1 > $O/run.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 tools/seed_sweep.py --seeds $S --no-ab --steps 1 > $O/run2.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
f2 = glob.glob(sys.argv[1] + "/sq2/**/*counter_collection.csv", recursive=True)
if f2:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f2[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("haf::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
    for k in ("k_recheck_i8", "k_svm_rbf_h<true>", "k_recheck_mfma", "k_svm_screen<false>", "k_features<1, 16>", "k_features_serial<2>"):
        a = acc.get(k)
        if not a: continue
        n = max(1, cnt[k])
        gui = a["GRBM_GUI_ACTIVE"] / 8 / n
        # SQ_ACTIVE_INST_* count per-wave busy cycles (x4 quad cycles); per SIMD share = value * 4 / (1024 SIMDs * cycles)
        print("%-22s VALU inst %.3g LDS inst %.3g | active_valu/wave-cycles %.1f %% active_lds %.1f %% | LDS idx active/cycles/CU %.1f %% bank conflict %.1f %%" % (
            k, a["SQ_INSTS_VALU"] / n, a["SQ_INSTS_LDS"] / n, 100 * a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_ACTIVE_INST_LDS"] / a["SQ_WAVE_CYCLES"],
            100 * a["SQ_LDS_IDX_ACTIVE"] / n / 256 / gui, 100 * a["SQ_LDS_BANK_CONFLICT"] / max(1.0, a["SQ_LDS_IDX_ACTIVE"])))
f = glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("haf::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
for k in ("k_recheck_i8", "k_svm_rbf_h<true>", "k_recheck_mfma", "k_svm_screen<false>", "k_features<1, 16>", "k_features_serial<2>"):
    a = acc.get(k)
    if not a: continue
    n = max(1, cnt[k])
    gui = a["GRBM_GUI_ACTIVE"] / 8 / n
    print("%-22s launches %d  cycles/XCD %.3g  MFMA busy %.1f %%  wave-cycles: wait_any %.1f %% wait_inst_any %.1f %% (LDS %.1f %%) active %.1f %%" % (
        k, n, gui, 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / n / 1024 / gui, 100 * a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"],
        100 * a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_WAIT_INST_LDS"] / a["SQ_WAVE_CYCLES"], 100 * a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
PY
