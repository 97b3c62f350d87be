#!/usr/bin/env python3
"""One line per bench.py JSON file: value, kernel ms, roofline fraction, ceiling, placement (for before/after comparisons)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    r, p = d["roofline"], d.get("placement", {})
    print("%-32s %.1f Grows/s  kernel %.4f ms  frac %.3f  ceiling %.0f GB/s (%.3f of it)  probe %.0f ratio %.3f tried %d/%d held %.0f GiB" % (
        f.split("/")[-1], d["value"] / 1e9, r["kernel_ms"], r["frac"], r["read_only_ceiling_gb_per_s"], r["frac_of_read_only_ceiling"],
        p.get("probe_gb_per_s", 0), p.get("probe_mix_over_read_only", 0), p.get("candidates_tried", 0), p.get("bitset_candidates_tried", 0), p.get("held_gib", 0)))
