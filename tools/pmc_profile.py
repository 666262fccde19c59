#!/usr/bin/env python3
"""Counter passes of one bench step with rocprofv3 (--pmc only, kernel dispatch records; never combined with tracing
domains) -> gpurun_out/<round>_pmc_hbm.{json,txt} and gpurun_out/<round>_pmc_sq.{json,txt}; copy them to profiles/.

    python tools/pmc_profile.py [hbm] [sq] [-- bench flags]

hbm: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass on gfx950).  Corrected as
     MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE x2 for wide coalesced streams, WRITE_SIZE as counted.
sq : issue counters of the pass-2 kernel in two passes.  SQ counters are recorded per shader engine (32 records per
     dispatch); the JSON holds per-launch TOTALS (sum over the shader engines).
Both JSON files carry ``src_hash`` (sha256 of the kernel sources, bench.source_hash) and ``workload_key``: bench.py reports
their numbers only while the sources and the workload are the ones profiled."""
import collections
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
RND = bench.PROFILE_ROUND


def run_pass(counters, tag, bench_flags):
    d = f"/tmp/pmc_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", *counters, "-d", d, "-o", "run", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--steps", "5", "--warmup", "2", "--ramp-ms", "0", "--no-cpu-baseline", *bench_flags]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    dbs = [os.path.join(p, f) for p, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    if not dbs or not line:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
        raise SystemExit(f"rocprofv3 pass {tag} failed")
    return dbs[0], json.loads(line[-1])


def per_kernel(db):
    """{kernel: {counter: (n_dispatches, total per dispatch)}}: a counter's records of one dispatch are summed."""
    c = sqlite3.connect(db)
    rows = c.execute("""select s.kernel_name, p.name, e.value, d.id from rocpd_pmc_event e
      join rocpd_info_pmc p on e.pmc_id = p.id
      join rocpd_kernel_dispatch d on e.event_id = d.event_id
      join rocpd_info_kernel_symbol s on d.kernel_id = s.id""").fetchall()
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for k, n, v, did in rows:
        agg[re.sub(r"\(.*", "", k)][n][did] += v
    return {k: {n: (len(dd), sum(dd.values()) / len(dd)) for n, dd in d.items()} for k, d in agg.items()}


def workload_key(line):
    return line["roofline"]["workload_key"]


def main():
    argv = sys.argv[1:]
    flags = []
    if "--" in argv:
        flags = argv[argv.index("--") + 1:]
        argv = argv[: argv.index("--")]
    what = argv or ["hbm", "sq"]
    os.makedirs(OUT, exist_ok=True)
    src = bench.source_hash()
    if "hbm" in what:
        txt, vals, line = [], {}, None
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            db, line = run_pass([c], c, flags)
            for k, d in per_kernel(db).items():
                if "nadm" in k and "synth" not in k:
                    vals.setdefault(k, {}).update(d)
        for k, d in vals.items():
            txt.append(k[:100])
            for n, (cnt, v) in sorted(d.items()):
                txt.append(f"    {n:14s} dispatches={cnt:3d}  KiB per launch={v:14.1f}")
        dec = [k for k in vals if "decode_bce" in k]
        f = sum(vals[k]["FETCH_SIZE"][1] for k in dec)
        w = sum(vals[k]["WRITE_SIZE"][1] for k in dec)
        nl = max(1, len(dec))
        out = {"round": RND, "src_hash": src, "workload_key": workload_key(line), "kernels": [k[:80] for k in dec],
               "fetch_size_kib_raw": f, "write_size_kib_raw": w,
               "correction": "FETCH_SIZE x2 (gfx950 tallies the 128 B requests of wide coalesced streams at 64 B, MI355X_MICROARCH.md HBM section; "
                             "calibrated in round 1 on the stand-alone Adam launch: 4 x 32 MiB read, 62.6 MiB counted); WRITE_SIZE as counted; "
                             "all pass-2 launches of one step summed",
               "traffic_bytes_per_launch": (2 * f + w) * 1024.0, "launches_per_step": nl,
               "alg_bytes_per_launch_8d": line["roofline"]["alg_bytes_per_launch"], "alg_bytes_min_per_launch": line["roofline"]["alg_bytes_min_per_launch"]}
        json.dump(out, open(os.path.join(OUT, f"{RND}_pmc_hbm.json"), "w"), indent=1)
        open(os.path.join(OUT, f"{RND}_pmc_hbm.txt"), "w").write(
            f"# {RND}: FETCH_SIZE / WRITE_SIZE per launch (KiB), separate --pmc passes, tools/pmc_profile.py hbm {' '.join(flags)}\n" + "\n".join(txt) + "\n")
        print(json.dumps(out))
    if "sq" in what:
        sets = [["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
                 "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
                ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY",
                 "SQ_INST_CYCLES_VMEM", "SQ_INSTS_VMEM"]]
        vals, line = {}, None
        for i, s in enumerate(sets):
            db, line = run_pass(s, f"sq{i}", flags)
            for k, d in per_kernel(db).items():
                if "nadm" in k and "synth" not in k:
                    vals.setdefault(k, {}).update(d)
        txt = []
        for k, d in vals.items():
            txt.append(k[:100])
            for n, (cnt, v) in sorted(d.items()):
                txt.append(f"    {n:28s} dispatches={cnt:3d}  per launch (sum over shader engines)={v:16.1f}")
        dec = [k for k in vals if "decode_bce" in k]
        tot = lambda n: sum(vals[k][n][1] for k in dec)
        out = {"round": RND, "src_hash": src, "workload_key": workload_key(line), "kernels": [k[:80] for k in dec],
               "valu_insts_per_launch": tot("SQ_INSTS_VALU"), "mfma_insts_per_launch": tot("SQ_INSTS_MFMA"),
               "lds_insts_per_launch": tot("SQ_INSTS_LDS"), "salu_insts_per_launch": tot("SQ_INSTS_SALU"),
               "busy_cycles_sum_over_se": tot("SQ_BUSY_CYCLES"), "wave_quad_cycles": tot("SQ_WAVE_CYCLES"), "active_inst_valu_quad_cycles": tot("SQ_ACTIVE_INST_VALU"),
               "valu_mfma_busy_cycles": tot("SQ_VALU_MFMA_BUSY_CYCLES"), "wait_inst_any_quad_cycles": tot("SQ_WAIT_INST_ANY"),
               "wait_any_quad_cycles": tot("SQ_WAIT_ANY"), "lds_bank_conflict_cycles": tot("SQ_LDS_BANK_CONFLICT"),
               "active_inst_lds_quad_cycles": tot("SQ_ACTIVE_INST_LDS"), "sclk_ghz": 2.4,
               "note": "per launch = all pass-2 launches of one step summed, counters summed over the 32 shader engines"}
        json.dump(out, open(os.path.join(OUT, f"{RND}_pmc_sq.json"), "w"), indent=1)
        open(os.path.join(OUT, f"{RND}_pmc_sq.txt"), "w").write(
            f"# {RND}: SQ counters per launch, two --pmc passes, tools/pmc_profile.py sq {' '.join(flags)}\n"
            "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES cycles\n" + "\n".join(txt) + "\n")
        print(json.dumps(out))


if __name__ == "__main__":
    main()
