#!/bin/bash
# Same-box A/B of environment switches on ONE build, printing the first level steps' times (the sparse leading levels).
# usage: ab_env_levels.sh workload n_levels "ENV=1 ..." "ENV=2 ..." ...   ("" = default); three alternating repetitions.
export GENPHI_ENV_HOOKS=1      # the library reads GENPHI_* hooks only under this gate
WL=$1; NL=$2; shift; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $v timeout -k 10 300 python bench.py --workload "$WL" --no-cpu-baseline --no-d2h > /tmp/ab.json 2> /tmp/ab.err || { echo "fail [$v]"; tail -3 /tmp/ab.err; }
    python - "$v" "$NL" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); l = d["config"]["level_ms"]; n = int(sys.argv[2])
print("[%s]" % sys.argv[1], round(d["ms_per_step"], 3), "replay", round(d["roofline"]["graph_replay_ms_per_step"], 3), "first", n, "levels", round(sum(l[:n]), 4), [round(x, 3) for x in l[:n]])
PY
  done
done
