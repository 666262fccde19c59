"""Thin command line with the reference's flags (entry.py:20-67): ``python -m neural_admixture_amd train|infer ...``.
Reads BED input straight into the packed layout, runs the RSVD + GMM initialisation, trains on the MI355X engine and
writes ``{name}.{K}.Q/.P``, ``{name}.pt`` and ``{name}_config.json`` exactly where the reference does
(src/main.py:38-44, src/inference.py:91-92).  BED and VCF inputs are read natively (io.read_bed_packed, io.read_vcf_packed);
PGEN needs pgenlib through the reference's own reader.  ``--num_gpus N`` spawns one process per GPU like the reference
(entry.py:186-190); ``--threads`` sets the size of the host thread pools like the reference's (entry.py:138-146): torch's pool
at once, the BLAS / OpenMP pools inside train() (host_threads=), the environment variables for child processes."""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

import numpy as np
import torch

logging.basicConfig(stream=sys.stdout, level=logging.INFO, format="%(message)s")
log = logging.getLogger(__name__)


def parse_train_args(argv):
    p = argparse.ArgumentParser(prog="neural-admixture train", description="Rapid population clustering with autoencoders - training mode")
    p.add_argument("--epochs", type=int, default=250)
    p.add_argument("--batch_size", type=int, default=800)
    p.add_argument("--learning_rate", type=float, default=20e-4)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--k", type=int)
    p.add_argument("--min_k", type=int)
    p.add_argument("--max_k", type=int)
    p.add_argument("--hidden_size", type=int, default=1024)
    p.add_argument("--save_dir", required=True, type=str)
    p.add_argument("--data_path", required=True, type=str)
    p.add_argument("--name", required=True, type=str)
    p.add_argument("--supervised_loss_weight", type=float, default=100)
    p.add_argument("--pops_path", type=str, default="")
    p.add_argument("--n_components", type=int, default=8)
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--threads", type=int, default=1,
                   help="host thread pools (torch, BLAS, OpenMP), like the reference's flag (entry.py:46,138-146; default 1 as there). "
                        "The GPU step does not use them; the decoder-init mixture fit and the host side of the readers do: 4 is a good value")
    p.add_argument("--parallelism", choices=("dp", "snp"), default="dp",
                   help="multi-GPU sharding: dp = samples (the reference's DDP), snp = SNPs (two tiny all-reduces per step)")
    p.add_argument("--gmm", choices=("auto", "sklearn", "native", "device", "em"), default="auto",
                   help="decoder-init mixture fit (model/train.py:60-68): sklearn = the reference's own scikit-learn GaussianMixture call; "
                        "auto (default) = its float64 restatement on host threads / HIP kernels (same means to 1e-10, no library import: "
                        "0.1 s instead of 1.5 s on a 1000-Genomes-sized run); see INTEGRATION.md section 3")
    p.add_argument("--share_gpu", action="store_true",
                   help="functional check of a --num_gpus N run on a ONE-GPU box: every rank uses cuda:0 and gloo carries the "
                        "tensors (RCCL refuses two ranks per device)")
    return p.parse_args(argv)


def parse_infer_args(argv):
    p = argparse.ArgumentParser(prog="neural-admixture infer", description="Rapid population clustering with autoencoders - inference mode")
    p.add_argument("--out_name", required=True, type=str)
    p.add_argument("--save_dir", required=True, type=str)
    p.add_argument("--data_path", required=True, type=str)
    p.add_argument("--name", required=True, type=str)
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--threads", type=int, default=1)
    return p.parse_args(argv)


def _read(path, device=None, keep_on_device=False):
    from .io import read_bed_packed, read_vcf_packed
    name = os.path.basename(path)
    if ".vcf" in name:                                   # src/snp_reader.py:103
        log.info("    Input format is VCF.")
        data = read_vcf_packed(path, device, keep_on_device)
        log.info(f"    Data contains {data.N} samples and {data.M} SNPs.")
        return data
    if ".bed" not in name:
        raise SystemExit("    Invalid format. Unrecognized file format. Make sure file ends with .bed or .vcf (.pgen needs pgenlib "
                         "through the reference's own reader).")
    log.info("    Input format is BED.")
    data = read_bed_packed(path, device, keep_on_device)
    log.info(f"    Data contains {data.N} samples and {data.M} SNPs.")
    return data


def _train_worker(rank, args, num_gpus, data, V, pops, t0):
    from .train import train
    from .io import save_model, write_outputs
    if num_gpus > 1:                                        # src/utils.py:69-95
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dev_id = 0 if args.share_gpu else rank
        torch.cuda.set_device(dev_id)
        torch.distributed.init_process_group("gloo" if args.share_gpu else "nccl", init_method="env://", rank=rank, world_size=num_gpus)
    else:
        dev_id = rank
    master = rank == 0
    device = torch.device(f"cuda:{dev_id}")
    K = args.k
    Ps, Qs, model = train(args.epochs, args.batch_size, args.learning_rate, K, args.seed, data, device, num_gpus, args.hidden_size,
                          master, V, pops, args.min_k, args.max_k, args.n_components, parallelism=args.parallelism,
                          host_threads=args.threads, gmm=args.gmm)
    if master:
        save_model(model, args.name, args.save_dir)
        write_outputs(Qs, args.name, K, args.min_k, args.max_k, args.save_dir, Ps)
        log.info("    Q and P matrices saved." if K is not None else "    Q and P matrices saved for all K.")
        log.info("")
        log.info(f"    Total elapsed time: {time.time() - t0:.2f} seconds.")
    if num_gpus > 1:
        torch.distributed.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    assert argv and argv[0] in ("train", "infer"), 'Please provide either the argument "train" or "infer" to choose running mode.'
    mode, t0 = argv[0], time.time()
    if not torch.cuda.is_available():
        raise SystemExit("neural_admixture_amd needs a ROCm GPU; use the reference for --num_gpus 0 (CPU) runs.")
    if mode == "train":
        args = parse_train_args(argv[1:])
        assert args.epochs > 0 and args.batch_size > 0 and args.learning_rate > 0 and args.hidden_size > 0 and args.n_components > 0
        pops = None
        if args.pops_path:                                  # src/utils.py:28-33: one population name per line
            with open(args.pops_path, "r") as fb:
                pops = [ln.strip() for ln in fb.readlines()]
            assert args.k is not None, "Supervised mode needs --k (the number of populations in --pops_path)."
        if args.k is not None:
            assert args.k > 1, "Please select K > 1."
            log.info(f"    Running on K = {args.k}.")
        elif args.min_k is not None and args.max_k is not None:
            assert args.min_k > 1 and args.max_k > args.min_k
            log.info(f"    Running from K={args.min_k} to K={args.max_k}.")
        else:
            raise ValueError("Please provide either --k or both --min_k and --max_k.")
        num_gpus = max(1, args.num_gpus if args.share_gpu else min(args.num_gpus, torch.cuda.device_count()))   # entry.py:168-173
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):       # entry.py:138-146
            os.environ[var] = str(args.threads)              # for child processes (the concurrent GMM fits, spawned ranks) ...
        torch.set_num_threads(max(1, args.threads))          # ... this process's pools are already up: torch here, BLAS / OpenMP
                                                             # through train(host_threads=) (threadpoolctl)
        from .svd import RSVD
        data = _read(args.data_path, torch.device("cuda:0"), keep_on_device=(num_gpus == 1))   # 2-bit transpose on the GPU
        log.info("")
        log.info("    Running SVD...")
        V = RSVD(data, data.N, data.M, args.n_components, args.seed)
        if num_gpus > 1:
            data.packed.share_memory_()
            torch.multiprocessing.spawn(_train_worker, args=(args, num_gpus, data, V, pops, t0), nprocs=num_gpus)
        else:
            _train_worker(0, args, 1, data, V, pops, t0)
        return 0
    args = parse_infer_args(argv[1:])
    from .model import Q_P
    from .io import write_outputs
    with open(f"{args.save_dir}/{args.name}_config.json") as fb:
        cfg = json.load(fb)
    sd = torch.load(f"{args.save_dir}/{args.name}.pt", map_location="cpu", weights_only=True)
    model = Q_P(int(cfg["hidden_size"]), int(cfg["num_features"]), ks_list=cfg["ks"], is_train=False)
    model.load_state_dict(sd, device=torch.device("cuda:0"), max_batch=args.batch_size)
    data = _read(args.data_path, torch.device("cuda:0"), keep_on_device=True)
    eng = model.engine
    eng.pack_from_host(data)
    idx = torch.arange(data.N, dtype=torch.int32, device=eng.device)
    outs = [[] for _ in cfg["ks"]]
    for s in range(0, data.N, args.batch_size):
        bb = min(args.batch_size, data.N - s)
        for h, q in enumerate(eng.infer_q(idx[s:s + bb], bb)):
            outs[h].append(q.cpu().numpy())
    Qs = [np.concatenate(o, axis=0) for o in outs]
    K = cfg["ks"][0] if len(cfg["ks"]) == 1 else None
    write_outputs(Qs, args.out_name, K, cfg["ks"][0], cfg["ks"][-1], args.save_dir)
    log.info(f"    Total elapsed time: {time.time() - t0:.2f} seconds.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
