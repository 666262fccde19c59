#!/usr/bin/env python3
"""SQ counters of passes 1 and 3 (tools/pmc_profile.py's two counter sets, per launch totals over the 32 shader engines)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import pmc_profile as P
sets = [["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
        ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_VMEM"]]
res = {}
for i, cs in enumerate(sets):
    db, _ = P.run_pass(cs, f"p13_{i}", sys.argv[1:])
    for k, d in P.per_kernel(db).items():
        if "encode_bwd" in k or "encode_fwd" in k:
            res.setdefault(k[:40], {}).update({n: round(v[1]) for n, v in d.items()})
for k, d in res.items():
    busy = d["SQ_BUSY_CYCLES"] / 32.0
    print(k, json.dumps(d))
    print("   cycles per SE %.0f; per SIMD (1024): VALU active %.1f %%  MFMA busy %.1f %%  VALU insts %.0f  MFMA insts %.0f  LDS active %.1f %%  bank conflict cycles / LDS active quad-cycles %.2f  VMEM insts %.0f" % (
        busy, 100 * 4 * d["SQ_ACTIVE_INST_VALU"] / 1024 / busy, 100 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / busy, d["SQ_INSTS_VALU"] / 1024, d["SQ_INSTS_MFMA"] / 1024,
        100 * 4 * d["SQ_ACTIVE_INST_LDS"] / 1024 / busy, d["SQ_LDS_BANK_CONFLICT"] / max(1, d["SQ_ACTIVE_INST_LDS"]), d["SQ_INSTS_VMEM"] / 1024))
    print("   wave quad-cycles %.3g  waiting on any instruction %.1f %%  waiting (any) %.1f %%" % (d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], 100 * d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]))
