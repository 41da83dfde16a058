cd $GRAFT_REPO_ROOT; L=$PWD/reface_amd/lib/alt
for rep in 1 2 3; do for v in attnold attnnew; do
  echo "$v: $(REFACE_HIP_LIB=$L/$v.so python tools/bench_gemm.py --only 'attn d40' --reps 30 2>/dev/null | tail -1)"
done; done
bash tools/ab_libs.sh $L/attnold.so $L/attnnew.so $L/attnold.so $L/attnnew.so
