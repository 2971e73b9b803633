#!/usr/bin/env python3
"""Per-instantiation register report for gemm_filter_kernel: VGPRs, spills, and how many scratch
loads/stores sit in basic blocks that also hold MFMAs (i.e. inside the K-loop rather than the epilogue).

    make -C innr_amd/csrc asm && python tools/gemm_spill_report.py
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, "innr_amd", "lib", "asm", "api.s")).read()
for name in re.findall(r"^(_ZN4innr18gemm_filter_kernel\S+):", s, flags=re.M):
    if name.endswith(".kd"):
        continue
    i = s.index("\n" + name + ":")
    j = s.index("s_endpgm", i)
    body = s[i:j]
    meta = s[j:s.index(".end_amdhsa_kernel", j)]
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)
    blocks = re.split(r"\n\.LBB\S+:", body)
    hot = sum(len(re.findall(r"scratch_(?:load|store)", b)) for b in blocks if "v_mfma" in b)
    tot = len(re.findall(r"scratch_(?:load|store)", body))
    t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name).groups()
    print("<%s,%s,%s,%s> vgpr=%s scratch_ops=%d in_mfma_blocks=%d" % (*t, vg, tot, hot))
