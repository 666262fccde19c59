#!/bin/bash
# Per-rank cost of the sample-sharded step at world = W, emulated on ONE GPU (bench.py --force-ddp --emulate-world W: rank 0 of W,
# no-op collectives, Adam on 1/W of the parameters) -> gpurun_out/r06_rank_emulation.txt.  NOT a scaling measurement: nothing
# crosses xGMI; it shows what a rank's GPU and host have to do per step -- r06: for message B in 1 / 2 / 4 / 8 SNP-range buckets,
# pass 3 launched range by range or whole, one or two communicators.
out=gpurun_out/r06_rank_emulation.txt
mkdir -p gpurun_out
{
echo "# r06 rank emulation on one MI355X (tools/rank_emulation.sh): ms/step, host ms to queue a step, kernel_ms (us); b = rows per rank"
echo "# weak = 800 rows per rank; strong = the reference's batch_size // num_gpus at --batch_size 800 (neural_admixture.py:287)"
row() { python bench.py --no-cpu-baseline --steps 100 --warmup 30 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-64s' % '$*', '| ms/step', '%.4f' % d['ms_per_step'], '| host', '%.4f' % d['host_queue_ms_per_step'], '|', {k: (round(v*1e3,1) if not isinstance(v, list) else [round(x*1e3,1) for x in v]) for k,v in d['roofline']['kernel_ms'].items()})"; }
row
for nb in 1 2 4 8; do row --force-ddp --buckets $nb; done
row --force-ddp --buckets 4 --p3-whole
for nb in 1 2 4 8; do row --force-ddp --emulate-world 8 --buckets $nb; done
row --force-ddp --emulate-world 8 --buckets 4 --p3-whole
row --force-ddp --emulate-world 8 --buckets 4 --comm-a
for w in 2 4; do row --force-ddp --emulate-world $w --buckets 1; row --force-ddp --emulate-world $w --buckets 4; done
row --batch 100
for nb in 1 2 4 8; do row --force-ddp --batch 100 --emulate-world 8 --buckets $nb; done
row --force-ddp --batch 100 --emulate-world 8 --buckets 4 --p3-whole
row --batch 400; row --force-ddp --batch 400 --emulate-world 2 --buckets 1; row --force-ddp --batch 400 --emulate-world 2 --buckets 4
row --batch 200; row --force-ddp --batch 200 --emulate-world 4 --buckets 1; row --force-ddp --batch 200 --emulate-world 4 --buckets 4
} | tee $out
