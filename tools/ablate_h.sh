# Timing experiments on the three-pass list kernel k_svm_rbf_h<true> (contraction.hip: HAF_ABL): what does its time hang on?
#   here (no GPU):   bash tools/ablate_h.sh build      -> haf_grasping_amd/abl/libhafgrasp_testing_abl{0..4}.so, then restores the real build
#   on the GPU box:  bash tools/ablate_h.sh run [seed] -> per-variant kernel times (rocprofv3 --kernel-trace --stats)
# Variants: 0 as shipped; 1 without the VALU adds of sweep 2; 2 without the sixteen v_exp_f32; 3 without sweep 1 (80 of 132 MFMAs);
# 4 without the LDS-DMA of the SV tiles (stale LDS); 5 (feature kernel, decq.h): an 8-byte instead of a 16-byte entry of the decimal pair
# table.  Variants 1-5 compute garbage: only their kernel times mean anything.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
if [ "$1" = build ]; then
    mkdir -p haf_grasping_amd/abl
    # (the variants are linked under the product names on the way; whatever happens -- an error, an interrupt -- the real
    # build is put back before this script ends, and a library built with experiment flags never stays under those names)
    restore() { unset HAF_EXPERIMENT_FLAGS; python3 -c "from haf_grasping_amd import build; build.build(verbose=False, force=True)"; }
    # an interrupt restores the real build ONCE and ends the script (a bare `trap restore INT` would run the handler and then go on
    # building variants under the product names); EXIT covers the normal end and `exit 1`
    trap 'trap - EXIT; restore; exit 130' INT TERM
    trap restore EXIT
    for n in ${VARIANTS:-0 1 2 3 4 5}; do
        HAF_EXPERIMENT_FLAGS="-DHAF_ABL=$n" python3 -c "from haf_grasping_amd import build; build.build(verbose=False, force=True)" || exit 1
        cp haf_grasping_amd/libhafgrasp_testing.so haf_grasping_amd/abl/libhafgrasp_testing_abl$n.so
    done
    exit 0          # (the trap restores the real build)
fi
S=${2:-11}
for n in ${VARIANTS:-0 1 2 3 4 5}; do
    echo "== variant $n"
    HAF_TESTLIB=$GRAFT_REPO_ROOT/haf_grasping_amd/abl/libhafgrasp_testing_abl$n.so bash tools/kernel_times_seed.sh $S 2>&1 | grep -E "k_svm_rbf_h|k_features<1|k_svm_screen|k_features_serial" 
done
