"""Table of a profiles/scale.sh run: per rank count the whole-job value, its ratio to N = 1, kernel-only and end-to-end times, and for
the column-panel runs the exchange volume with the xGMI prediction.  usage: scale_summary.py OUT_DIR N [N ...]"""
import json
import os
import sys


def last_json(path):
    try:
        lines = [l for l in open(path) if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except OSError:
        return None


def main():
    out, ns = sys.argv[1], [int(x) for x in sys.argv[2:]]
    rows = {n: last_json(os.path.join(out, f"rows_n{n}.json")) for n in ns}
    base = rows.get(1)
    print("row shards (upper levels replicated, last level sharded, no collective):")
    for n in ns:
        d = rows[n]
        if d is None:
            print(f"  N={n}: no line"); continue
        if d.get("dry_run"):
            print(f"  N={n}: dry run ok, shards {d['shards']}, wall is max over ranks: {d['wall_is_max']}"); continue
        e = d.get("end_to_end", {})
        print(f"  N={n}: {d['value']:.4g} pairs/s, {d['ms_per_step']:.2f} ms/sweep"
              + (f", x{d['value'] / base['value']:.2f} of N=1" if base and not base.get("dry_run") else "")
              + f"; kernels {d['config']['kernel_ms_per_step']:.2f} ms; end to end plan {e.get('plan_ms', 0):.0f} + sweep {e.get('sweep_ms', 0):.1f} + copy {e.get('d2h_ms', float('nan')):.0f} ms")
    print("column panels (cfg3s, one all-to-all of parent columns per level step):")
    for n in ns:
        d = last_json(os.path.join(out, f"panels_cfg3s_n{n}.json"))
        if d is None:
            continue
        c = d["config"]
        print(f"  N={n}: {d['ms_per_step']:.2f} ms/sweep, {c['exchange_bytes_sent_per_rank_max'] / 1e6:.1f} MB sent per rank and sweep, "
              f"xGMI prediction {c.get('exchange_ms_predicted_xgmi', 0):.2f} ms, kernels per rank {d['roofline']['per_rank_kernels']['device_ms_per_sweep']:.2f} ms")


if __name__ == "__main__":
    main()
