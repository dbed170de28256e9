#!/usr/bin/env python
"""Derive profiles/pmc_latest.json (what bench.py reports as roofline.traffic / roofline.valu) from the per-kernel counter
means of tools/pmc_summary.py --json.  The file is STAMPED with the digest of the kernel sources + flags of the library that
was measured (styl3r_amd/_lib.built_digest()); bench.py reports the figures only when its own library carries the same
digest, otherwise `traffic: null` and a "stale" note.
  HBM bytes per launch = FETCH_SIZE [KiB] * 1024 * 2  (gfx950: FETCH_SIZE counts 128-B requests as 64 B, MI355X guide)
                         + WRITE_SIZE [KiB] * 1024
  kernel cycles        = GRBM_GUI_ACTIVE / 8            (summed over the 8 XCDs)
  valu_insts_per_pair  = SQ_INSTS_VALU / pairs (the (tile, Gaussian) pairs of the launch: what the composite kernels' VALU
                         work scales with); salu / lds likewise
  valu_active_frac     = SQ_ACTIVE_INST_VALU [quad-cycles] * 4 / (1024 SIMDs * kernel cycles); > 1 is possible: the counter
                         charges every instruction ~4 cycles while the common fp32 forms issue in 2 (profiles/r02_issue_cost.md):
                         read >= 1 as "the VALU port is the limiter"
usage: pmc_latest.py <pmc_summary.json> <source label> <pairs per launch> [out.json]"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from styl3r_amd import _lib
src, label, pairs = sys.argv[1], sys.argv[2], float(sys.argv[3])
out = sys.argv[4] if len(sys.argv) > 4 else "profiles/pmc_latest.json"
res = {}
for kn, c in json.load(open(src)).items():
    short = kn.split("::")[-1].split("<")[0]
    short = short[2:] if short.startswith("k_") else short
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    f2, w = c.get("FETCH_SIZE", 0.0) * 1024 * 2, c.get("WRITE_SIZE", 0.0) * 1024
    res[short] = {"hbm_bytes_per_launch": int(f2 + w), "fetch_bytes_x2": int(f2), "write_bytes": int(w), "kernel_cycles": int(cyc),
                  "valu_insts_per_launch": int(c.get("SQ_INSTS_VALU", 0)),
                  "valu_insts_per_pair": round(c.get("SQ_INSTS_VALU", 0) / pairs, 2),
                  "salu_insts_per_pair": round(c.get("SQ_INSTS_SALU", 0) / pairs, 2),
                  "lds_insts_per_pair": round(c.get("SQ_INSTS_LDS", 0) / pairs, 2),
                  "valu_active_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024 * cyc), 3) if cyc else None,
                  "lds_bank_conflict_frac": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3) if c.get("SQ_LDS_IDX_ACTIVE") else None}
doc = {"build_digest": _lib.built_digest(), "source": label, "pairs_per_launch": int(pairs), "kernels": res}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"], v["valu_insts_per_pair"]) for k, v in res.items()}))
