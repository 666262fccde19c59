#!/bin/bash
# Per-kernel register / spill / instruction-mix summary of nadm_genotype_passes.hip for a set of -D flags (no GPU needed):
#   tools/isa_stats.sh <kernel-substring> [-DFLAG=..]...
set -e
pat=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -S --cuda-device-only -o $tmp/k.s $R/neural-admixture_amd/csrc/nadm_genotype_passes.hip "$@" 2>/dev/null
python3 - "$tmp/k.s" "$pat" <<'PY'
import re, sys, collections
txt = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
i = 0
while i < len(txt):
    m = re.match(r"^(_Z\w+):", txt[i])
    if m and pat in m.group(1):
        name = m.group(1)
        j = i + 1
        while j < len(txt) and not txt[j].startswith("\t.end_amdhsa_kernel") and not re.match(r"^\s*\.amdhsa_kernel", txt[j]):
            j += 1
        body = txt[i:j]
        # innermost hot loop = the longest basic-block span between a loop header label and its back edge
        cnt = collections.Counter()
        for l in body:
            t = l.strip().split()
            if t and not t[0].startswith((";", ".", "_Z")) and not t[0].endswith(":"):
                cnt[t[0]] += 1
        k = j
        meta = {}
        while k < len(txt) and not txt[k].startswith("\t.end_amdhsa_kernel"):
            mm = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size)\s+(\d+)", txt[k])
            if mm: meta[mm.group(1)] = int(mm.group(2))
            k += 1
        tot = sum(cnt.values())
        key = ["v_mfma_f32_16x16x32_bf16", "v_log_f32_e32", "v_rcp_f32_e32", "v_max_f32_e32", "v_max_f32_e64", "v_max3_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
               "v_cvt_pk_bf16_f32", "scratch_store_dword", "scratch_load_dword", "s_nop"]
        print(name[:70], meta, "insts", tot, {k_: cnt[k_] for k_ in key if cnt[k_]})
        i = k
    i += 1
PY
rm -rf $tmp
