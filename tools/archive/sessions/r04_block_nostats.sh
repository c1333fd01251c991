cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04f; mkdir -p $OUT
for v in libq2048_b64.so product libq2048_b64.so; do
  if [ "$v" = product ]; then unset Q2048_LIB_PATH; else export Q2048_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/$v; fi
  INTERCEPT_ONLY="learning, row cache, NO statistics (stats" timeout -k 10 300 python3 tools/archive/exp_intercept.py 2>>$OUT/err.log | head -1 | sed "s/^/$v /" | tee -a $OUT/block_intercept_nostats.txt
done
