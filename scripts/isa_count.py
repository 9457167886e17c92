#!/usr/bin/env python3
"""Static instruction census of one kernel in the ISA listing (`make -C code-robchar_amd/csrc asm`, or any `hipcc -S
--cuda-device-only` output): counts per instruction class and the resource metadata.  Development aid: the dynamic count
(`SQ_INSTS_VALU / SQ_WAVES`, scripts/pmc_quick.sh) needs a GPU, straight-line changes show up here without one.
usage: python scripts/isa_count.py <listing.s> <substring of the mangled kernel name> [...]"""
import collections
import re
import sys


def census(path, pat):
    body, meta, inside, name = [], {}, False, None
    for line in open(path):
        if not inside:
            m = re.match(r"^(\S+):\s*; @", line)
            if m and pat in m.group(1):
                inside, name = True, m.group(1)
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            break
        body.append(line)
    for line in open(path):
        m = re.match(r"\s*\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s*(\d+)", line)
        if m and meta.get("_hit"):
            meta.setdefault(m.group(1), int(m.group(2)))
        if ".name:" in line:
            meta["_hit"] = (name is not None and name in line)
            if meta["_hit"]:
                meta = {"_hit": True}
    for line in body:                                   # (the YAML block lists .group_segment_fixed_size BEFORE .name)
        m = re.match(r"\s*\.amdhsa_group_segment_fixed_size\s+(\d+)", line)
        if m:
            meta["group_segment_fixed_size"] = int(m.group(1))
    cnt = collections.Counter()
    for line in body:
        t = line.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_"):
            if "f64" in op:
                cls = "v_trans_f64" if re.search(r"rsq|rcp|sqrt", op) else ("v_cvt" if "cvt" in op else ("v_cmp" if "cmp" in op else "v_f64"))
            elif "f32" in op or "f16" in op:
                cls = "v_trans_f32" if re.search(r"rsq|rcp|sqrt|exp|log|sin|cos", op) else ("v_cvt" if "cvt" in op else ("v_cmp" if "cmp" in op else ("v_pk_f32" if "pk" in op else "v_f32")))
            elif re.match(r"v_(mov|accvgpr|readlane|writelane|readfirstlane|swap|cndmask|perm|bfi|bfe)", op):
                cls = "v_move/select"
            elif "cmp" in op:
                cls = "v_cmp"
            else:
                cls = "v_int/other"
        elif op.startswith("s_"):
            cls = "s_waitcnt/nop" if re.match(r"s_(waitcnt|nop|sleep|setprio|barrier)", op) else ("s_branch" if re.match(r"s_(cbranch|branch)", op) else ("s_load" if "load" in op else "s_alu"))
        elif op.startswith("ds_"):
            cls = "ds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls = "scratch" if op.startswith("scratch_") else "vmem"
        else:
            cls = "other"
        cnt[cls] += 1
    return name, cnt, {k: v for k, v in meta.items() if k != "_hit"}


if __name__ == "__main__":
    for pat in sys.argv[2:]:
        name, cnt, meta = census(sys.argv[1], pat)
        valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
        print(f"{name}\n  static VALU {valu}  " + "  ".join(f"{k} {v}" for k, v in sorted(cnt.items())) + f"\n  {meta}")
