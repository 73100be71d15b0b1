# Kernel-level times of the exact convergence chain's kernels (rmse_*), for the library in the tree and -- when it is there -- for
# wgsassign_amd/libwgsassign_old.so (an A/B on one box): rocprofv3 --kernel-trace --stats of tools/check_codes.py.  On the GPU box:
#   bash tools/prof_walk.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
for l in old hip; do
  [ -f $R/wgsassign_amd/libwgsassign_$l.so ] || continue
  export WGSASSIGN_LIB_PATH=$R/wgsassign_amd/libwgsassign_$l.so
  rm -rf $R/gpurun_out/walk_$l
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/walk_$l -- python3 $R/tools/check_codes.py > $R/gpurun_out/walk_$l.log 2>&1 || exit 1
  echo "== $l"; grep "rmse_" $R/gpurun_out/walk_$l/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
done
