#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: kernel trace + HBM counters of bench.py.
# PMC counters go in their own passes (never combined with trace domains other than
# --kernel-trace/--stats).  Output lands in gpurun_out/prof_<tag>/; summarise with
# profiles/summarize.py and commit the summaries it writes into profiles/.
set -u
TAG=${1:-r01}; WL=${2:-cfg4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${WL}; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --workload "$WL" --steps 5 --warmup 1 --no-cpu-baseline --no-d2h --no-others --no-call-wall > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --workload "$WL" --steps 2 --warmup 0 --no-cpu-baseline --no-d2h --no-others --no-call-wall > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --workload "$WL" --steps 2 --warmup 0 --no-cpu-baseline --no-d2h --no-others --no-call-wall > "$OUT/pmc_write.log" 2>&1
# requests from the CUs into L2 (TCP -> TCC): what the staged source rows cost on the CU side whether they hit L2 or not
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d "$OUT/pmc_tcp" -- python3 bench.py --workload "$WL" --steps 2 --warmup 0 --no-cpu-baseline --no-d2h --no-others --no-call-wall > "$OUT/pmc_tcp.log" 2>&1
grep '^{' "$OUT/kt.log" | tail -1 | cut -c1-400
