#!/bin/bash
# Same-box A/B of environment switches on ONE build (boxes differ by several per cent, so only
# numbers from one gpurun call are comparable).  usage: ab_env.sh workload "ENV=1 ..." "ENV=2 ..." ...
# An empty string "" is the default configuration.  Three alternating repetitions.
export GENPHI_ENV_HOOKS=1      # the library reads GENPHI_* hooks only under this gate
WL=$1; shift
for rep in 1 2 3; do
  for v in "$@"; do
    env $v timeout -k 10 300 python bench.py --workload "$WL" --no-cpu-baseline --no-d2h > /tmp/ab.json 2> /tmp/ab.err || { echo "fail [$v]"; tail -3 /tmp/ab.err; }
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); l = d["config"]["level_ms"]
print("[%s]" % sys.argv[1], round(d["ms_per_step"], 2), "upper", round(sum(l[1:24]) / 23, 4) if len(l) > 25 else "", "final", round(l[-1], 3), "frac", round(d["roofline"]["frac"], 4))
PY
  done
done
