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
    print("%-32s %.1f Grows/s  kernel %.4f ms  frac %.3f  ceiling %.0f GB/s (%.3f of it)  arena %.0f/%.0f GiB mapped/budget, bitset pair check %s, spacers %.0f GiB, setup %.1f s" % (
        f.split("/")[-1], d["value"] / 1e9, r["kernel_ms"], r["frac"], r["read_only_ceiling_gb_per_s"], r["frac_of_read_only_ceiling"],
        p.get("arena_mapped_gib", 0), p.get("arena_budget_gib", 0), p.get("bitset_pair_check_at_allocation"), p.get("mixed_zone_gib", 0), p.get("setup_s", 0)))
