#!/usr/bin/env python3
"""Instruction census of pass 2's headline instantiation (decode_bce_bf16_kernel<8, LOSS, UNIT_P, QIMG, !SLICED>) from `hipcc -S`
(no GPU needed):  python tools/isa_census.py [-DFLAG ...] > profiles/r06_p2_isa_census.txt

The kernel's VALU instructions are counted per REGION of the code (block prologue, per-64-sample-tile staging, the hot loop over 32-sample
pairs, the cold exact-loss fallback, block epilogue), weighted by how often a thread runs the region at b = 800 (12.5 tiles, 25 hot-loop
rounds of 32 genotypes per lane), and the hot loop is itemised by what the instructions are for.  SQ_INSTS_VALU of the profiled launch
(profiles/*_pmc_sq.json) is the dynamic total this static count is checked against."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "decode_bce_bf16_kernelILi8ELb1ELb1ELb1ELb0E"
B, TS = 800, 64


def ops(lines):
    c = collections.Counter()
    for l in lines:
        t = l.strip().split()
        if t and not t[0].startswith((";", ".", "_Z")) and not t[0].endswith(":"):
            c[t[0]] += 1
    return c


def valu(c):
    return sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))


def main():
    flags = sys.argv[1:]
    src = os.path.join(ROOT, "neural-admixture_amd", "csrc", "nadm_genotype_passes.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src] + flags,
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read().split("\n")
    i = next(k for k, l in enumerate(txt) if re.match(r"^_Z\w+:", l) and KERNEL in l)
    j = next(k for k in range(i, len(txt)) if txt[k].startswith("\t.end_amdhsa_kernel") or re.match(r"^\s*\.amdhsa_kernel", txt[k]))
    body = txt[i:j]
    meta = {}
    for l in txt[j:j + 200]:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|accum_offset)\s+(\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    # LLVM annotates every basic block with the loop it belongs to ("in Loop: Header=BBx Depth=d") and every header with its depth
    # ("This [Inner] Loop Header: Depth=d", parents listed above it): assign each instruction to its innermost loop's depth.  Depth 1 = the
    # tile loop, 2 = the loop over 32-sample pairs (the hot loop), 3 = the cold exact-loss fallback.  Depth-1 loops before / behind the
    # tile loop (P rows -> LDS, the epilogue's row pieces) are told apart by position.
    depth, cur, head_depth = [], 0, {}
    k = 0
    while k < len(body):
        l = body[k]
        m = re.match(r"^\.LBB\w+:\s*;(.*)$", l)
        if m:
            blk = m.group(1)
            kk = k + 1
            while kk < len(body) and body[kk].strip().startswith(";"):
                blk += body[kk]
                kk += 1
            mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", blk) or re.search(r"in Loop: Header=\w+ Depth=(\d+)", blk)
            cur = int(mm.group(1)) if mm else 0
        elif re.match(r"^\.LBB\w+:", l):
            cur = 0
        depth.append(cur)
        k += 1
    first2 = next(k for k, d_ in enumerate(depth) if d_ == 2)
    tile0 = max(k for k in range(first2) if depth[k] == 0) + 1                  # the tile loop's first depth-1 block in front of the pair loop
    last = max(k for k, d_ in enumerate(depth) if d_ >= 2)
    tile1 = next((k for k in range(last, len(body)) if depth[k] == 0), len(body))
    outer = (tile0, tile1)
    hot_lines = [body[k] for k in range(tile0, tile1) if depth[k] == 2]
    cold_lines = [body[k] for k in range(tile0, tile1) if depth[k] == 3]
    tile_lines = [body[k] for k in range(tile0, tile1) if depth[k] == 1]
    loops = []
    for lo_, hi_ in ((0, tile0), (tile1, len(body))):                            # depth-1 loops of the prologue / epilogue: 2 trips each at KP = 8
        k = lo_
        while k < hi_:
            if depth[k] >= 1:
                e = k
                while e < hi_ and depth[e] >= 1:
                    e += 1
                loops.append((k, e))
                k = e
            else:
                k += 1
    pro, epi = body[:outer[0]], body[outer[1]:]
    H, T, C_, P, E = ops(hot_lines), ops(tile_lines), ops(cold_lines), ops(pro), ops(epi)
    tiles, rounds, per_lane = B / TS, B / 32, 32
    geno = B                                                                     # genotypes per thread and block: 800 samples x 256 SNPs / 256 threads
    # loops inside prologue / epilogue (P rows -> LDS: 2 float4 per thread; finish(): 2 float4 per thread) are counted once statically and
    # run twice: add their bodies once more
    def inner_loops(lo, hi):
        return [(a, b_) for a, b_ in loops if lo <= a and b_ <= hi]
    extra = 0
    for a, b_ in inner_loops(0, outer[0]) + inner_loops(outer[1], len(body)):
        extra += valu(ops(body[a:b_]))
    print(f"# pass 2 headline instantiation: {meta}; static VALU counts per region (MFMA excluded)")
    print(f"region            VALU(static)  runs/thread  VALU/thread   per genotype")
    rows = [("block prologue", valu(P), 1), ("block epilogue", valu(E), 1), ("(their inner loops, 2nd trip)", extra, 1),
            ("per 64-sample tile", valu(T), tiles), ("hot loop (32-sample pair)", valu(H), rounds), ("cold exact-loss fallback", valu(C_), 0)]
    tot = 0.0
    for name, n, w in rows:
        tot += n * w
        print(f"{name:30s} {n:6d}   x {w:6.1f}   = {n * w:9.0f}    {n * w / geno:6.3f}")
    print(f"{'static estimate':30s}                     {tot:9.0f}    {tot / geno:6.3f}   (SQ_INSTS_VALU of the profiled launch: profiles/r06_pmc_sq.json -- 13.91 per genotype; 15.35 with the r03-r06 loss form)")
    # ---- the hot loop itemised: 32 genotypes per lane = 16 PAIRS (packed f32 math handles two genotypes per instruction)
    pairs = per_lane // 2
    want = [  # bucket, opcode, count per pair, what it is
        ("decode", "v_cvt_scalef32_pk_f32_fp4", 1, "x of the pair from the code word (FP4 E2M1 read)"),
        ("gradient", "v_pk_fma_f32", 1, "den = d - d^2"),
        ("gradient", "v_rcp_f32_e32", 2, "1 / den"),
        ("gradient", "v_mul_f32_e64", 2, "sat(1e-12 * rcp): the floor and the clamp mask in one multiply"),
        ("gradient", "v_pk_add_f32", 1, "d - x"),
        ("gradient", "v_pk_mul_f32", 1, "(d - x) * inv"),
        ("loss", "v_sub_f32_e64", 2, "o = sat(1 - d)"),
        ("loss", "v_pk_add_f32", 1, "q = o - x"),
        ("loss", "v_pk_fma_f32", 2, "x^2 - x = -[c == 1] / 4;  f = q^2 + that  (r06: | q^2 - x(1-x) | is f for all three calls)"),
        ("loss", "v_mul_f32_e32", 1, "f0 * f1"),
        ("loss", "v_log_f32_e64", 1, "one logarithm per pair, | . | as a source modifier"),
        ("loss", "v_add_f32_e32", 1, "accumulate"),
        ("bf16 split", "v_cvt_pk_bf16_f32", 2, "hi = bf16(dR), lo = bf16(dR - hi)"),
        ("bf16 split", "v_lshlrev_b32_e32", 1, "hi.x back to f32"),
        ("bf16 split", "v_and_b32_e32", 1, "hi.y back to f32"),
        ("bf16 split", "v_pk_add_f32", 1, "dR - hi"),
    ]
    left = collections.Counter({k: v for k, v in H.items() if k.startswith("v_") and not k.startswith("v_mfma")})
    # the compiler splits a few packed ops into scalar halves (v_fma_f32 / v_add_f32_e64): fold them back for the itemisation
    bucket = collections.Counter()
    detail = []
    for bk, op, n, what in want:
        take = min(left[op], n * pairs)
        left[op] -= take
        bucket[bk] += take
        detail.append((bk, op, n * pairs, take, what))
    split_scalar = left.pop("v_fma_f32", 0) + left.pop("v_add_f32_e64", 0)
    print(f"\n# hot loop, one round = 32 genotypes per lane = {pairs} pairs; MFMA {H['v_mfma_f32_16x16x32_bf16']}, LDS {sum(v for k, v in H.items() if k.startswith('ds_'))}, "
          f"s_nop {H['s_nop']}, s_waitcnt {H['s_waitcnt']}")
    print("bucket       opcode                        expected  found   what")
    for bk, op, e, t, what in detail:
        print(f"{bk:12s} {op:28s} {e:6d}  {t:6d}   {what}")
    missing = sum(e - t for _, _, e, t, _ in detail)
    print(f"(packed ops the compiler issued as scalar halves instead: {split_scalar} instructions for {missing} missing packed ones)")
    over = sum(v for v in left.values() if v > 0) + max(0, split_scalar - missing)
    bucket["address / mask / overhead"] = over
    print("overhead     " + ", ".join(f"{k} {v}" for k, v in sorted(left.items(), key=lambda x: -x[1]) if v > 0))
    print("\nbucket                        per round   per genotype")
    for bk in ("decode", "gradient", "loss", "bf16 split", "address / mask / overhead"):
        print(f"{bk:28s} {bucket[bk]:8d}     {bucket[bk] / per_lane:6.3f}")
    print(f"{'hot loop total':28s} {valu(H):8d}     {valu(H) / per_lane:6.3f}")
    print(f"\n# outside the hot loop, per genotype: tile staging {valu(T) * tiles / geno:.3f} (commit: clean_codes x 4, masks, LDS / batch-copy stores; dQ slab rows summed over "
          f"the waves), block prologue + epilogue {(valu(P) + valu(E) + extra) / geno:.3f} (P rows -> bf16 operand pieces; dP fold, Adam + clamp on the block's P rows)")
    print("# per-tile opcodes: " + ", ".join(f"{k} {v}" for k, v in sorted(T.items(), key=lambda x: -x[1]) if k.startswith("v_"))[:900])


main()
