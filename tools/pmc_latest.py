#!/usr/bin/env python
"""Derive profiles/pmc_latest.json (what bench.py reports as roofline.traffic / valu_busy_frac_pmc) from the
per-kernel counter means of tools/pmc_summary.py --json.
  HBM bytes per launch = FETCH_SIZE [KiB] * 1024 * 2  (gfx950: FETCH_SIZE counts 64-B requests as 32 B, MI355X guide)
                         + WRITE_SIZE [KiB] * 1024
  kernel cycles        = GRBM_GUI_ACTIVE / 8            (summed over the 8 XCDs)
  VALU busy fraction   = min(1, SQ_ACTIVE_INST_VALU [quad-cycles] * 4 / (1024 SIMDs * kernel cycles)); the unclamped ratio
                         (valu_cycles_over_kernel_cycles) exceeds 1 on the composite kernels once they are balanced:
                         the 4-cycles-per-wave64-instruction model over-counts the cheaper VALU forms, so read it as
                         "the VALU issue port is saturated", not as a calibrated percentage
usage: pmc_latest.py <pmc_summary.json> <source label> [out.json]"""
import json, sys
src, label = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_latest.json"
res = {}
for kn, c in json.load(open(src)).items():
    short = kn.split("::")[-1].split("<")[0]
    short = short[2:] if short.startswith("k_") else short
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    f2, w = c.get("FETCH_SIZE", 0.0) * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    res[short] = {"hbm_bytes_per_launch": int(f2 + w), "fetch_bytes_x2": int(f2), "write_bytes": int(w), "kernel_cycles": int(cyc),
                  "valu_busy_frac": round(min(1.0, c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024 * cyc)), 3) if cyc else None,
                  "valu_cycles_over_kernel_cycles": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024 * cyc), 3) if cyc else None,
                  "valu_wave_insts": int(c.get("SQ_INSTS_VALU", 0)), "source": label}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"], v["valu_busy_frac"]) for k, v in res.items()}))
