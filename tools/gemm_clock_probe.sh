cd /tmp && export TMPDIR=/tmp
for f in 0 -1; do
  rm -rf /tmp/pc$f
  ( cd $GRAFT_REPO_ROOT && GEMM_FILL=$f GEMM_ONLY="square 4096^3 NN" rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/pc$f -o t -- tools/bin/gemm_probe > /tmp/pc$f.log 2>&1 )
  echo "GEMM_FILL=$f"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$(find /tmp/pc$f -name '*.db' | head -1)" --pmc | grep -E "k_dgemm|KERNEL" | cut -c1-40,95-170
done
