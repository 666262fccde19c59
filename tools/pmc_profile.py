#!/usr/bin/env python3
"""Counter passes of one bench step with rocprofv3 (--pmc only, kernel dispatch records; never combined with tracing
domains) -> gpurun_out/<round>_pmc_hbm.{json,txt} and gpurun_out/<round>_pmc_sq.{json,txt}; copy them to profiles/.

    python tools/pmc_profile.py [calib] [req] [hbm] [sq] [-- bench flags]

req: (r04) the L2's memory-side REQUEST counters by size -- TCC_EA0_RDREQ / _32B / _64B / _128B, TCC_BUBBLE, TCC_EA0_WRREQ / _64B
     -- for the calibration kernels and for pass 2: physical bytes = sum over sizes of requests x size, no calibration ratio in
     between -> gpurun_out/<round>_pmc_req.json.  `hbm` then reports three figures side by side: raw FETCH_SIZE + WRITE_SIZE, the
     guide's rule (FETCH_SIZE x 2), and the request-size bytes.

calib: what FETCH_SIZE / WRITE_SIZE count for the access patterns of pass 2 (tools/calib_fetch.hip: kernels with a known byte
     count each) -> gpurun_out/<round>_pmc_calib.json = counted / known per pattern.  MI355X_MICROARCH.md "HBM": a wide coalesced
     stream is tallied at half its bytes on gfx950, "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
     byte count in your own access pattern".
hbm: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass on gfx950), converted to bytes with the calibrated
     ratios, weighted by the known composition of the launch's reads (X pieces / P, m, v streams) and writes (batch copy / dQ
     slab / P, m, v streams); without a calibration file: FETCH_SIZE x2, WRITE_SIZE as counted (the guide's rule, r02).
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
           "--steps", "5", "--warmup", "2", "--ramp-ms", "0", "--no-cpu-baseline", "--no-epoch-loop", "--no-clock-probe", *bench_flags]
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


def calibrate():
    """counted / known bytes per access pattern -> <round>_pmc_calib.json"""
    exe = os.path.join(ROOT, "tools", "abl", "calib")
    if not os.path.exists(exe):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "calib_fetch.hip"), "-o", exe], check=True)
    out = {"round": RND, "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; ratio = counted KiB * 1024 / known bytes of the launch"}
    known = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = f"/tmp/pmc_calib_{c}"
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--pmc", c, "-d", d, "-o", "run", "--", exe], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                           capture_output=True, text=True)
        m = re.search(r"known_bytes (.*)", r.stdout)
        toks = m.group(1).split()
        known = {toks[i]: int(toks[i + 1]) for i in range(0, len(toks), 2)}
        db = [os.path.join(p_, f) for p_, _, fs in os.walk(d) for f in fs if f.endswith(".db")][0]
        for k, dd in per_kernel(db).items():
            name = next((n for n in known if n in k), None)
            if name and c in dd:
                out.setdefault(name, {"known_bytes": known[name]})[c.lower() + "_kib"] = dd[c][1]
    for name, d in out.items():
        if isinstance(d, dict) and "known_bytes" in d:
            key = "fetch_size_kib" if name.endswith("read") else "write_size_kib"
            d["ratio"] = d.get(key, 0.0) * 1024.0 / d["known_bytes"]
    json.dump(out, open(os.path.join(OUT, f"{RND}_pmc_calib.json"), "w"), indent=1)
    print(json.dumps(out))
    return out


REQ_SETS = [["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
            ["TCC_BUBBLE_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]]


def req_bytes(d):
    """{counter: (n, per dispatch)} of one kernel -> request counts by size and the bytes they move."""
    g = lambda n: d[n][1] if n in d else None
    rd, r32, r64, r128, bub = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum"), g("TCC_EA0_RDREQ_64B_sum"), g("TCC_EA0_RDREQ_128B_sum"), g("TCC_BUBBLE_sum")
    wr, w64 = g("TCC_EA0_WRREQ_sum"), g("TCC_EA0_WRREQ_64B_sum")
    out = {"rdreq": rd, "rdreq_32b": r32, "rdreq_64b": r64, "rdreq_128b": r128, "tcc_bubble": bub, "wrreq": wr, "wrreq_64b": w64}
    if None not in (rd, r32, r64, r128):
        other = rd - r32 - r64 - r128                                  # requests no size counter claims (0 if the three partition RDREQ)
        out["rdreq_unsized"] = other
        out["read_bytes_by_request_size"] = 32.0 * r32 + 64.0 * r64 + 128.0 * r128 + 64.0 * max(other, 0.0)
    if None not in (wr, w64):
        out["write_bytes_by_request_size"] = 64.0 * w64 + 32.0 * (wr - w64)
    return out


def requests():
    """Request-size counters of the calibration kernels (known payload per launch) -> physical bytes per payload byte per access
    pattern; and of the bench step's pass 2."""
    exe = os.path.join(ROOT, "tools", "abl", "calib")
    if not os.path.exists(exe):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "calib_fetch.hip"), "-o", exe], check=True)
    out = {"round": RND, "note": "TCC_EA0_* = the L2's requests to the fabric (Infinity Cache / HBM side), summed over the TCC instances; bytes = sum "
                                 "over request sizes of count x size.  physical_per_payload = those bytes / the bytes the kernel asked for"}
    pats = {}
    for i, cs in enumerate(REQ_SETS):
        d = f"/tmp/pmc_calibreq_{i}"
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["rocprofv3", "--pmc", *cs, "-d", d, "-o", "run", "--", exe], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                           capture_output=True, text=True)
        m = re.search(r"known_bytes (.*)", r.stdout)
        if not m:
            sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
            raise SystemExit("calibration run under the request counters failed")
        toks = m.group(1).split()
        known = {toks[j]: int(toks[j + 1]) for j in range(0, len(toks), 2)}
        db = [os.path.join(p_, f) for p_, _, fs in os.walk(d) for f in fs if f.endswith(".db")][0]
        for k, dd in per_kernel(db).items():
            name = next((n for n in known if n in k), None)
            if name:
                pats.setdefault(name, {"known_bytes": known[name], "_c": {}})["_c"].update(dd)
    for name, e in pats.items():
        rb = req_bytes(e.pop("_c"))
        e.update(rb)
        key = "read_bytes_by_request_size" if name.endswith("read") else "write_bytes_by_request_size"
        if rb.get(key) is not None:
            e["physical_per_payload"] = rb[key] / e["known_bytes"]
    out["patterns"] = pats
    return out


def load_calib():
    for d in (OUT, os.path.join(ROOT, "profiles")):
        try:
            return json.load(open(os.path.join(d, f"{RND}_pmc_calib.json")))
        except (OSError, ValueError):
            pass
    return None


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
    if "calib" in what:
        calibrate()
        what = [w for w in what if w != "calib"]
    req = None
    if "req" in what:
        flags_ = flags
        req = requests()
        vals = {}
        line = None
        for i, cs in enumerate(REQ_SETS):
            db, line = run_pass(cs, f"req{i}", flags_)
            for k, d in per_kernel(db).items():
                if "nadm" in k and "synth" not in k:
                    vals.setdefault(k, {}).update(d)
        req["src_hash"], req["workload_key"] = src, workload_key(line)
        req["kernels"] = {k[:90]: req_bytes(d) for k, d in vals.items()}
        dec = [k for k in vals if "decode_bce" in k]
        tot = {}
        for k in dec:
            for n, v in req_bytes(vals[k]).items():
                if v is not None:
                    tot[n] = tot.get(n, 0.0) + v
        req["pass2_per_step"] = tot
        json.dump(req, open(os.path.join(OUT, f"{RND}_pmc_req.json"), "w"), indent=1)
        print(json.dumps(req))
        what = [w for w in what if w != "req"]
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
        cal = load_calib()
        rl = line["roofline"]
        comp = rl.get("traffic_composition")       # known bytes of the launch per access pattern (bench.py)
        if cal and comp:
            rd = {"rows64_read": comp["x_pieces_read"], "stream_read": comp["param_stream_read"]}
            wr = {"rows64_write": comp["batch_copy_write"], "slab_write": comp["dq_slab_write"], "stream_write": comp["param_stream_write"]}
            r_read = sum(v * cal[k]["ratio"] for k, v in rd.items()) / max(1.0, sum(rd.values()))
            r_write = sum(v * cal[k]["ratio"] for k, v in wr.items()) / max(1.0, sum(wr.values()))
            traffic = f * 1024.0 / r_read + w * 1024.0 / r_write
            corr = ("counted KiB x 1024 / (counted-per-known-byte ratio of the launch's own access patterns, tools/calib_fetch.hip -> "
                    f"{RND}_pmc_calib.json): reads {r_read:.3f} (X pieces {cal['rows64_read']['ratio']:.3f}, parameter streams {cal['stream_read']['ratio']:.3f}), "
                    f"writes {r_write:.3f} (batch copy {cal['rows64_write']['ratio']:.3f}, dQ slab {cal['slab_write']['ratio']:.3f}, parameter streams "
                    f"{cal['stream_write']['ratio']:.3f}); all pass-2 launches of one step summed")
            known = {"reads": rd, "writes": wr, "expected_fetch_kib": sum(v * cal[k]["ratio"] for k, v in rd.items()) / 1024.0,
                     "expected_write_kib": sum(v * cal[k]["ratio"] for k, v in wr.items()) / 1024.0}
        else:
            traffic, known = (2 * f + w) * 1024.0, None
            corr = ("FETCH_SIZE x2 (gfx950 tallies the 128 B requests of wide coalesced streams at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as "
                    "counted; all pass-2 launches of one step summed (no calibration file)")
        if req is None:
            for d_ in (OUT, os.path.join(ROOT, "profiles")):
                try:
                    req = json.load(open(os.path.join(d_, f"{RND}_pmc_req.json")))
                    break
                except (OSError, ValueError):
                    pass
        three = {"raw_counters_bytes": (f + w) * 1024.0, "guide_rule_bytes": (2 * f + w) * 1024.0, "calibrated_bytes_r03_method": traffic if (cal and comp) else None,
                 "request_size_bytes": None}
        if req and req.get("src_hash") == src and req.get("workload_key") == workload_key(line):
            t_ = req["pass2_per_step"]
            if "read_bytes_by_request_size" in t_ and "write_bytes_by_request_size" in t_:
                three["request_size_bytes"] = t_["read_bytes_by_request_size"] + t_["write_bytes_by_request_size"]
                three["request_size_read_bytes"], three["request_size_write_bytes"] = t_["read_bytes_by_request_size"], t_["write_bytes_by_request_size"]
                traffic = three["request_size_bytes"]               # the physical figure: what bench.py reports as roofline.traffic
                corr = ("bytes = sum over request sizes of TCC_EA0_RDREQ_{32B,64B,128B} x size + TCC_EA0_WRREQ (64 B / 32 B) x size "
                        f"({RND}_pmc_req.json, separate --pmc passes); raw FETCH_SIZE + WRITE_SIZE and the guide's x2 rule are listed beside it")
        out = {"round": RND, "src_hash": src, "workload_key": workload_key(line), "kernels": [k[:80] for k in dec],
               "fetch_size_kib_raw": f, "write_size_kib_raw": w, "correction": corr, "known_composition": known,
               "three_figures": three,
               "traffic_bytes_per_launch": traffic, "launches_per_step": nl,
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
               "active_inst_lds_quad_cycles": tot("SQ_ACTIVE_INST_LDS"),
               "note": "per launch = all pass-2 launches of one step summed, counters summed over the 32 shader engines"}
        json.dump(out, open(os.path.join(OUT, f"{RND}_pmc_sq.json"), "w"), indent=1)
        open(os.path.join(OUT, f"{RND}_pmc_sq.txt"), "w").write(
            f"# {RND}: SQ counters per launch, two --pmc passes, tools/pmc_profile.py sq {' '.join(flags)}\n"
            "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_BUSY_CYCLES and SQ_VALU_MFMA_BUSY_CYCLES cycles\n" + "\n".join(txt) + "\n")
        print(json.dumps(out))


if __name__ == "__main__":
    main()
