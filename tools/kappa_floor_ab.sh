for m in "" "HAF_KAPPA_T1_MEASURED=1"; do
  echo "== $m"
  env $m python tools/seed_sweep.py --seeds 42,11 --trained --no-ab --steps 3
done
