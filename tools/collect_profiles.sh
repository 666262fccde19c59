#!/bin/bash
# Everything measured that profiles/r06_* holds, from the current sources, on the GPU box (about 6 minutes):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh'   then copy gpurun_out/r06_* into profiles/
# Kernel tables (rocprofv3 --kernel-trace --stats), counter passes (--pmc only, stamped with the source hash bench.py checks),
# the bench lines and the full-run wall-clocks.
T="timeout 600"
$T bash tools/step_profile.sh r06 > /dev/null
$T bash tools/step_profile.sh r06_noloss --no-loss > /dev/null
$T bash tools/step_profile.sh r06_ddp1 --force-ddp > /dev/null
$T bash tools/step_profile.sh r06_c2 --k 7 --rows 2504 --snps 600000 > /dev/null
$T bash tools/step_profile.sh r06_c3 --min-k 2 --max-k 10 --rows 2504 --snps 600000 > /dev/null
$T python tools/pmc_profile.py sq -- --no-loss > /dev/null 2>&1; cp gpurun_out/r06_pmc_sq.txt gpurun_out/r06_pmc_sq_noloss.txt
$T python tools/pmc_profile.py calib req hbm sq > gpurun_out/r06_pmc.log 2>&1
mkdir -p profiles; cp gpurun_out/r06_pmc_hbm.json gpurun_out/r06_pmc_sq.json gpurun_out/r06_pmc_calib.json gpurun_out/r06_pmc_req.json profiles/      # bench.py reads them from profiles/
$T python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err      # the driver's command
$T python bench.py --steps 100 --warmup 30 --no-loss --no-cpu-baseline > gpurun_out/r06_bench_noloss.json 2>/dev/null
$T python bench.py --steps 100 --warmup 30 --force-ddp --no-cpu-baseline > gpurun_out/r06_bench_ddp1.json 2>/dev/null
$T python bench.py --steps 50 --warmup 10 --min-k 2 --max-k 10 --rows 2504 --snps 600000 > gpurun_out/r06_bench_c3.json 2>/dev/null
$T python bench.py --steps 50 --warmup 10 --k 7 --rows 2504 --snps 600000 > gpurun_out/r06_bench_c2.json 2>/dev/null
$T python bench.py --rows 500000 --snps 1000000 --k 16 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null > gpurun_out/r06_bench_c5_1gpu.json
$T python bench.py --steps 100 --warmup 30 --force-ddp --batch 100 --no-cpu-baseline > gpurun_out/r06_bench_ddp1_b100.json 2>/dev/null
$T python bench.py --steps 100 --warmup 30 --force-ddp --emulate-world 8 --no-cpu-baseline > gpurun_out/r06_bench_ddp1_emul8.json 2>/dev/null
$T python bench.py --steps 100 --warmup 30 --force-ddp --emulate-world 8 --batch 100 --no-cpu-baseline > gpurun_out/r06_bench_ddp1_b100_emul8.json 2>/dev/null
$T python bench.py --steps 100 --warmup 30 --batch 100 --no-cpu-baseline > gpurun_out/r06_bench_b100.json 2>/dev/null
$T python bench.py --steps 100 --warmup 30 --parallelism snp --batch 6400 --snps 62500 --no-cpu-baseline > gpurun_out/r06_bench_snp1_b6400.json 2>/dev/null
for c in c2 c3 c4; do NADM_RSVD_PHASES=1 $T python tools/full_run.py $c 2>/dev/null | tail -1; done > gpurun_out/r06_full_runs.txt
python - <<'PY'
import json
for f in ("default", "noloss", "ddp1", "ddp1_emul8", "b100", "ddp1_b100", "ddp1_b100_emul8", "c2", "c3", "c5_1gpu", "snp1_b6400"):
    d = json.loads(open(f"gpurun_out/r06_bench_{f}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, round(d["ms_per_step"], 4), "host", round(d["host_queue_ms_per_step"], 4), "%.4g" % d["value"], {k: (round(v * 1e3, 1) if not isinstance(v, list) else [round(x * 1e3, 1) for x in v]) for k, v in r["kernel_ms"].items()}, round(r["frac_8d"], 4),
          round(r["frac_min"], 4), r["traffic"], r.get("issue_frac"), r.get("valu_busy_frac"), r.get("mfma_busy_frac"),
          (r["issue"] or {}).get("valu_insts_per_genotype"))
for l in open("gpurun_out/r06_full_runs.txt"):
    d = json.loads(l)
    print(d["config"], round(d["total_s"], 2), "s;", round(d["epoch_s"] * 1e3, 2), "ms/epoch", {k: round(v, 2) for k, v in d["train_phases"].items()}, "rsvd", round(d["rsvd_s"], 2), d.get("rsvd_phases"))
PY
head -9 gpurun_out/r06_kernel_stats.txt | cut -c1-130
