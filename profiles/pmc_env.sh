#!/bin/bash
# Run ON THE GPU BOX: HBM bytes of the certified-rows kernel under an environment setting (one FETCH_SIZE and one WRITE_SIZE pass
# of bench.py cfg4, one sweep).  usage: pmc_env.sh tag "ENV=1 ENV2=2"
export GENPHI_ENV_HOOKS=1      # the library reads GENPHI_* hooks only under this gate
TAG=$1; ENVS=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_env/$TAG
for pass in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  for kv in $ENVS; do export "$kv"; done
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_env/$TAG/$n -- python3 bench.py --workload cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-d2h > gpurun_out/pmc_env/$TAG/$n.log 2>&1
done
python3 - $TAG <<'PY'
import csv, glob, sys, collections, re
tag = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/pmc_env/{tag}/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\(anonymous namespace\)::([A-Za-z_0-9]+(?:<[^>]*>)?)\(", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:30]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[k]["ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, c in out.items():
    if "fast" in k:
        n = len(c["FETCH_SIZE"]) or 1
        print(tag, k, "launches", n, "ms", round(sum(c["ms"]) / max(len(c["ms"]), 1), 4),
              "read GB (x2 corr.)", round(sum(c["FETCH_SIZE"]) * 1024 * 2 / n / 1e9, 3), "write GB", round(sum(c["WRITE_SIZE"]) * 1024 / max(len(c["WRITE_SIZE"]), 1) / 1e9, 3),
              "L2 hit %", round(100 * sum(c["TCC_HIT_sum"]) / max(sum(c["TCC_HIT_sum"]) + sum(c["TCC_MISS_sum"]), 1), 1))
PY
rm -rf gpurun_out/pmc_env/$TAG/*/
