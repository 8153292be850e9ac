#!/bin/bash
# Run ON THE GPU BOX: quick probes of the sparse leading levels (kernel stats + forced last sparse cut).
export GENPHI_ENV_HOOKS=1      # the library reads GENPHI_* hooks only under this gate
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r5probe}; mkdir -p "$OUT"
for WL in cfg2 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$WL" -- python3 bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline --no-d2h > "$OUT/kt_$WL.log" 2>&1
  f=$(ls $OUT/kt_$WL/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$WL.csv"
  rm -rf "$OUT/kt_$WL"
done
for spec in "cfg2 10 200" "cfg2 9 200" "cfg2 8 200" "cfg4 6 200" "cfg4 4 200"; do
  set -- $spec
  GENPHI_SPARSE_K=$2 GENPHI_SPARSE_PERMILLE=$3 timeout -k 10 200 python3 bench.py --workload $1 --steps 5 --no-cpu-baseline --no-d2h > "$OUT/bench_$1_k$2.json" 2> "$OUT/bench_$1_k$2.err"
  echo "$spec rc=$?"
done
