# SQ / LDS counters per kernel of one C5 step for one model seed (largest launch of every kernel): bash tools/pmc_seed.sh 11
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=${1:-11}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_seed$S
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -- python3 tools/seed_sweep.py --seeds $S --no-ab --steps 1 > $O/run.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/lds -- python3 tools/seed_sweep.py --seeds $S --no-ab --steps 1 > $O/run2.log 2>&1
python3 tools/pmc_sq_summary.py "$(find $O/sq -name '*counter_collection.csv' | head -1)"
python3 tools/pmc_sq_summary.py "$(find $O/lds -name '*counter_collection.csv' | head -1)" --lds
