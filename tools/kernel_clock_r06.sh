cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
run() { # run <tag> <cmd...>
  local tag=$1; shift
  rm -rf /tmp/g_$tag
  ( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d /tmp/g_$tag -o t -- "$@" > /tmp/g_$tag.log 2>&1 )
  python $ROOT/tools/rocpd_summary.py "$(find /tmp/g_$tag -name '*.db' | head -1)" --pmc > $ROOT/gpurun_out/grbm_$tag.txt 2>&1
}
run em python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary
run iv python tools/pmc_blocks.py iv 2560 2
run tv python bench.py --workload tv --steps 2 --warmup 1 --no-cpu-baseline --tv-utterances 2048
run score python tools/pmc_blocks.py score 100000 2
run topc python tools/topc_bw.py
python - <<'PY'
import re, os, glob
root = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(root + "/gpurun_out/grbm_*.txt")):
    print("==", os.path.basename(f))
    for line in open(f):
        m = re.match(r"^(.*?)\s+GRBM_GUI_ACTIVE\s+(\d+)\s+([\d.]+)\s+(\d+)\s", line)
        if m and float(m.group(4)) > 200000:
            cyc, ns = float(m.group(3)), float(m.group(4))
            print("  %-60s calls %4s  %8.3f ms  %.3f GHz" % (m.group(1)[:60], m.group(2), ns * 1e-6, cyc / 8 / ns))
PY
