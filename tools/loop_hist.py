#!/usr/bin/env python3
"""Instruction histogram of the hot loop (the innermost loop that holds MFMAs) of one kernel of nadm_genotype_passes.hip:
   python tools/loop_hist.py <kernel-substring> [-DFLAG ...]      (hipcc -S, no GPU needed)
Prints the VALU / MFMA / LDS / memory instruction counts of that loop body and an issue-cycle estimate from the per-form costs
measured by tools/ubench_valu_asm.hip (profiles/r02_ubench_valu_asm.txt)."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COST = [("v_pk_", 4.8), ("v_log", 8.3), ("v_rcp", 8.4), ("v_sqrt", 8.4), ("v_exp", 8.4), ("v_max3", 4.4), ("v_med3", 4.4), ("v_cvt", 4.4), ("v_cvt_scalef32", 5.1),
        ("v_perm", 4.5), ("v_lshl", 4.3), ("v_lshr", 4.3), ("v_ashr", 4.3), ("v_bfe", 4.5), ("v_max_", 4.3), ("v_min_", 4.3), ("v_cmp", 5.5), ("v_cndmask", 5.0),
        ("v_and_or", 4.4), ("v_lshl_or", 4.4), ("v_lshl_add", 4.4), ("v_or3", 4.4), ("v_bitop3", 4.4), ("v_bfi", 4.4), ("v_alignbit", 4.4), ("v_mfma", 17.7)]
def cost(op):
    c = 3.0
    for pre, v in COST:
        if op.startswith(pre): c = v
    return c
def main():
    pat, flags = sys.argv[1], sys.argv[2:]
    src = os.path.join(ROOT, "neural-admixture_amd", "csrc", "nadm_genotype_passes.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src] + flags,
                       check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read().split("\n")
    i = next(k for k, l in enumerate(txt) if re.match(r"^_Z\w+:", l) and pat in l)
    j = next(k for k in range(i, len(txt)) if txt[k].startswith("\t.end_amdhsa_kernel") or re.match(r"^\s*\.amdhsa_kernel", txt[k]))
    body = txt[i:j]
    labels = {m.group(1): k for k, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
    loops = []
    for k, l in enumerate(body):
        m = re.match(r"\s+s_cbranch\w*\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            loops.append((labels[m.group(1)], k + 1))
    # the hot loop = the loop with the most MFMAs of its own (lines of nested loops -- the cold fallback of pass 2 -- excluded)
    best, best_n = None, -1
    for lo, hi in loops:
        inner = [(a, b_) for a, b_ in loops if lo <= a and b_ <= hi and (a, b_) != (lo, hi)]
        own = [body[k] for k in range(lo, hi) if not any(a <= k < b_ for a, b_ in inner)]
        n = sum("v_mfma" in s_ for s_ in own)
        if n > best_n:
            best, best_n = own, n
    cnt = collections.Counter()
    for l in best:
        t = l.strip().split()
        if t and not t[0].startswith((";", ".", "_Z")) and not t[0].endswith(":"):
            cnt[t[0]] += 1
    valu = {k: v for k, v in cnt.items() if k.startswith("v_") and not k.startswith("v_mfma")}
    cyc = sum(cost(k) * v for k, v in valu.items())
    print("loop:", len(best), "lines;", "VALU", sum(valu.values()), "MFMA", cnt["v_mfma_f32_16x16x32_bf16"], "LDS", sum(v for k, v in cnt.items() if k.startswith("ds_")),
          "vmem", sum(v for k, v in cnt.items() if k.startswith(("global_", "buffer_", "scratch_"))), "s_nop", cnt["s_nop"], "s_waitcnt", cnt["s_waitcnt"], "barrier", cnt["s_barrier"])
    print("VALU issue estimate %.0f cycles + MFMA %.0f" % (cyc, 17.7 * cnt["v_mfma_f32_16x16x32_bf16"]))
    print(dict(sorted(valu.items(), key=lambda x: -x[1])))
main()
