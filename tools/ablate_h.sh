# Timing experiments on the three-pass list kernel k_svm_rbf_h<true> (contraction.hip: HAF_ABL): what does its time hang on?
#   here (no GPU):   bash tools/ablate_h.sh build      -> haf_grasping_amd/variants/libhafgrasp_testing_abl{0..5}.so (the product build is not touched)
#   on the GPU box:  bash tools/ablate_h.sh run [seed] -> per-variant kernel times (rocprofv3 --kernel-trace --stats)
# Variants: 0 as shipped; 1 without the VALU adds of sweep 2; 2 without the sixteen v_exp_f32; 3 without sweep 1 (80 of 132 MFMAs);
# 4 without the LDS-DMA of the SV tiles (stale LDS); 5 (feature kernel, decq.h): an 8-byte instead of a 16-byte entry of the decimal pair
# table.  Variants 1-5 compute garbage: only their kernel times mean anything.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"; export GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
if [ "$1" = build ]; then
    # every variant is built NEXT TO the product (haf_grasping_amd/build.py: build_variant -> haf_grasping_amd/variants/): the
    # product's library names are never overwritten, so there is nothing to restore whatever interrupts this loop
    for n in ${VARIANTS:-0 1 2 3 4 5}; do
        python3 -m haf_grasping_amd.build --variant abl$n -DHAF_ABL=$n > /dev/null || exit 1
    done
    exit 0
fi
S=${2:-11}
for n in ${VARIANTS:-0 1 2 3 4 5}; do
    echo "== variant $n"
    HAF_TESTLIB=$GRAFT_REPO_ROOT/haf_grasping_amd/variants/libhafgrasp_testing_abl$n.so bash tools/kernel_times_seed.sh $S 2>&1 | grep -E "k_svm_rbf_h|k_features<1|k_svm_screen|k_features_serial" 
done
