for c in 8 12 16 20 24 28; do
  GENPHI_MAX_CPT=$c timeout -k 10 120 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/sw_$c.json 2> gpurun_out/sw_$c.err || echo fail $c
  python - <<PY
import json
d=json.load(open("gpurun_out/sw_$c.json"))
l=d["config"]["level_ms"]
print($c, round(d["ms_per_step"],2), "upper", round(sum(l[1:24])/23,4), "final", l[-1], "lvl0", l[0], "late", l[24:28])
PY
done
