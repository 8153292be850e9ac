#!/bin/bash
# Same-box A/B of any number of builds of libgenphi.so, three alternating repetitions.  usage: ab_libs_n.sh workload a.so b.so ...
WL=$1; shift
cp genlib.jl_amd/lib/libgenphi.so /tmp/keep.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp "$v" genlib.jl_amd/lib/libgenphi.so
    timeout -k 10 300 python bench.py --workload "$WL" --no-cpu-baseline --no-d2h > /tmp/ab.json 2> /tmp/ab.err || { echo "fail $v"; tail -3 /tmp/ab.err; }
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); l = d["config"]["level_ms"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"], 4), "upper", round(sum(l[1:24]) / 23, 4) if len(l) > 25 else "", "final", round(l[-1], 3),
      "replay", round(d["roofline"].get("graph_replay_ms_per_step") or 0, 4))
PY
  done
done
cp /tmp/keep.so genlib.jl_amd/lib/libgenphi.so
