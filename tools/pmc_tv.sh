set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r02; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TV="python bench.py --workload tv --steps 2 --warmup 1 --no-cpu-baseline --tv-utterances 2048"
for c in FETCH_SIZE WRITE_SIZE; do
  tag=bench_tv_pmc_$(echo $c | tr A-Z a-z)
  ( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$tag -o t -- $TV > $OUT/pmc_$tag.log 2>&1 )
  python $ROOT/tools/rocpd_summary.py "$(find $OUT/pmc_$tag -name '*.db' | head -1)" --pmc > $OUT/$tag.txt 2>&1
  rm -rf $OUT/pmc_$tag
done
head -14 $OUT/bench_tv_pmc_fetch_size.txt | cut -c1-200; head -14 $OUT/bench_tv_pmc_write_size.txt | cut -c1-200
