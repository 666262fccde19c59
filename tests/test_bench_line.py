"""The bench line's schema (r06): what the driver reads, and -- with several ranks -- every alternative multi-GPU leg and every
collective of the micro-benchmark, so that ONE `bench.py --gpus 8` run on a real node answers DESIGN.md section 5's open questions."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_share_gpu_line_of_8_ranks_has_every_alt_leg_and_collective():
    import bench
    assert bench.ALT_LEGS == ("dp_weak", "dp_global_batch", "snp_weak", "snp_global_batch", "dp_2buckets", "dp_comm_a")
    assert bench.ALT_COLLECTIVES == ("reduce_scatter_msg_a", "all_gather_msg_a", "reduce_scatter_msg_b", "all_gather_msg_b", "all_reduce_small")
    d = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_share_gpu_n8.json")).read())
    assert bench.validate_line(d, 8) == []
    assert d["config"]["share_gpu"] and d["rccl_ranks"] == 0                      # a functional run: says so, claims no RCCL ranks
    assert d["alt"]["dp_global_batch"]["rows_per_rank_per_step"] == 100           # the reference's semantics: 800 // 8 (neural_admixture.py:287)
    assert d["alt"]["snp_weak"]["global_batch"] == 6400 and d["alt"]["snp_global_batch"]["global_batch"] == 800
    assert d["collectives"]["reduce_scatter_msg_a"]["bytes"] == 16_000_000       # message A = all P: 500k x 8 floats
    assert d["collectives"]["all_reduce_small"]["bytes"] == 200_000
    broken = dict(d, alt={k: v for k, v in d["alt"].items() if k != "snp_weak"})
    assert bench.validate_line(broken, 8) == ["alt.snp_weak missing or empty"]
    assert "missing box" in bench.validate_line({k: v for k, v in d.items() if k != "box"}, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_bench_share_gpu_runs_every_leg_and_prints_one_valid_line(world):
    import bench
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu", "--rows", "8000", "--snps", "40000",
                        "--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--alt-steps", "2", "--alt-warmup", "1", "--coll-reps", "2",
                        "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                                 # ONE JSON line on stdout, whatever the libraries print
    d = json.loads(lines[0])
    assert bench.validate_line(d, world) == []
    assert d["alt"]["dp_global_batch"]["rows_per_rank_per_step"] == 800 // world
    assert "incomplete line" not in r.stderr


@pytest.mark.gpu
def test_bench_one_gpu_line_carries_a_measured_clock_and_a_box_fingerprint():
    import bench
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "8000", "--snps", "200000", "--steps", "5", "--warmup", "2",
                        "--ramp-ms", "50", "--no-cpu-baseline", "--no-epoch-loop"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert bench.validate_line(d, 1) == []
    box = d["box"]
    assert 1.0 < box["effective_sclk_ghz"] < 2.6 and box["calib_ms"] > 0.1 and 2000 < box["copy_gbs"] < 8000
    assert box["wall_clock_khz"] == 100000.0
    assert 1.0 < box["dominant_kernel_sclk_ghz"] <= box["effective_sclk_ghz"] * 1.03            # pass 2 clocks itself; denser than the calibration stream
    assert d["roofline"]["issue"] is None or d["roofline"]["issue"]["sclk_ghz"] == box["dominant_kernel_sclk_ghz"]


@pytest.mark.gpu
def test_bench_prints_its_headline_when_the_alternative_legs_run_out_of_time():
    """A stuck or slow leg must not cost a multi-GPU run its headline: with a 1-second deadline for the legs rank 0 still prints ONE valid
    line -- the headline, alt = null, alt_error saying why -- and the launcher exits 0."""
    import bench
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--rows", "8000", "--snps", "40000",
                        "--steps", "3", "--warmup", "1", "--ramp-ms", "0", "--alt-steps", "200", "--alt-deadline", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["alt"] is None and "did not complete within" in d["alt_error"] and d["value"] > 0 and d["n_gpus"] == 2
    assert [p for p in bench.validate_line(d, 2) if not p.startswith(("alt.", "collectives."))] == []
