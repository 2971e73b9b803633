#!/usr/bin/env python3
"""Build-time guard for the hand-counted waits in gemm_filter_kernel (innr_amd/csrc/kernels_gemm.h).

The query operands are loaded by inline asm (`global_load_dwordx2`) and become valid only at the matching
`s_waitcnt vmcnt(N)` asm, which ties the registers with "+v". If the register allocator ever satisfied such a tie by
COPYING a register (copy before the wait = stale data), results would be silently wrong on some runs. This script reads
the ISA (`make -C innr_amd/csrc asm`) and fails unless, in every instantiation, the 8 operand register pairs are touched
only by: the asm loads themselves, MFMAs, their zero-initialisation, and address temporaries that feed the very next
load of the same pair.

A second check covers the hazard behind the one GPU memory fault this code has had: a VALU-written SGPR pair (a reloaded
spill, a v_readfirstlane) read as the BASE of a VMEM instruction needs wait states that hipcc does not insert in front of
inline asm. Every asm load therefore copies its base with `s_mov_b64` inside the same statement (SALU reads are
interlocked); the script fails if any `global_load*` inside an ;;#ASMSTART block takes a base that the block did not
just write with s_mov_b64.

    make -C innr_amd/csrc asm && python tools/check_gemm_asm.py
(also run by __graft_entry__.build() and tests/test_abi.py::test_gemm_asm_guard, both on the CPU box)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "innr_amd", "lib", "asm", "api.s")
s = open(path).read()
KERNELS = r"(?:18gemm_filter_kernel|23gemm_bf16_filter_kernel|21gemm_i8_filter_kernel|22gemm_i8h_filter_kernel)"
names = [n for n in re.findall(r"^(_ZN4innr" + KERNELS + r"\S+):", s, flags=re.M) if not n.endswith(".kd")]
if not names:
    sys.exit("no gemm_filter_kernel in " + path)
bad_total = 0
for name in names:
    i = s.index("\n" + name + ":")
    body = s[i:s.index("s_endpgm", i)].split("\n")
    # hazard check: inside every asm block, the SGPR base of a global load must have been written by an s_mov_b64 of the
    # same block (never a register the compiler may have produced on the VALU right before the statement)
    unsafe = []
    k = 0
    while k < len(body):
        if ";;#ASMSTART" in body[k]:
            j = k + 1
            moved = set()
            while j < len(body) and ";;#ASMEND" not in body[j]:
                t = body[j].split(";")[0].strip()
                m = re.match(r"s_mov_b64 (s\[\d+:\d+\]),", t)
                if m:
                    moved.add(m.group(1))
                m = re.match(r"global_load\S* .*?(s\[\d+:\d+\])", t)
                if m and m.group(1) not in moved:
                    unsafe.append(t)
                j += 1
            k = j
        k += 1
    pairs = set()
    for k, line in enumerate(body):
        if ";;#ASMSTART" in line:
            j = k + 1
            while j < len(body) and ";;#ASMEND" not in body[j]:
                m = re.match(r"\s*global_load_dwordx[24] v\[(\d+):(\d+)\]", body[j])
                if m:
                    pairs.add((int(m.group(1)), int(m.group(2))))
                j += 1
    flat = {r for a, b in pairs for r in range(a, b + 1)}
    bad = []
    # the operand registers carry in-flight loads from the first asm load to the last MFMA; outside that window they
    # are ordinary registers (zero-initialisation before, the final list compaction after)
    loads = [k for k, l in enumerate(body) if re.match(r"\s*global_load_dwordx[24] v\[", l)]
    mfmas = [k for k, l in enumerate(body) if l.strip().startswith("v_mfma")]
    lo, hi = (loads[0] if loads else 0), (mfmas[-1] if mfmas else len(body))
    for k, line in enumerate(body):
        if k < lo or k > hi:
            continue
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith(("v_mfma", "global_load_dwordx2", "global_load_dwordx4", "s_")):
            continue
        used = set()
        for mm in re.finditer(r"v\[(\d+):(\d+)\]", t):
            used.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        for mm in re.finditer(r"\bv(\d+)\b", t):
            used.add(int(mm.group(1)))
        if not (used & flat):
            continue
        ops = [p.strip() for p in t[len(op):].split(",")]
        dst = ops[0]
        # allowed: writes (zero-init, temporaries while the pair is dead); a READ of an operand register is not
        srcs = ",".join(ops[1:])
        src_regs = set()
        for mm in re.finditer(r"v\[(\d+):(\d+)\]", srcs):
            src_regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        for mm in re.finditer(r"\bv(\d+)\b", srcs):
            src_regs.add(int(mm.group(1)))
        if src_regs & flat:
            # reading a pair as the ADDRESS of its own reload (v_lshl_add_u64 vX, vX ... ; global_load vX, vX) is fine
            nxt = " ".join(body[k + 1:k + 4])
            if op.startswith("v_lshl_add_u64") and re.search(r"global_load_dwordx[24] " + re.escape(dst), nxt):
                continue
            bad.append(t)
    short = re.sub(r".*gemm_filter_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E.*", r"<\1,\2,\3,\4>", name)
    short = re.sub(r".*gemm_bf16_filter_kernelILi(\d+)ELi(\d+)E.*", r"bf16<\1,\2>", short)
    short = re.sub(r".*gemm_i8_filter_kernelILi(\d+)ELi(\d+)E.*", r"i8<\1,\2>", short)
    short = re.sub(r".*gemm_i8h_filter_kernelILi(\d+)ELi(\d+)E.*", r"i8h<\1,\2>", short)
    want_pairs = 8  # f32: 8 k-pairs of one K-step; bf16 / int8: 2 ring positions x 2 depths x 2 fragments
    status = "ok" if (len(pairs) == want_pairs and not bad and not unsafe) else "FAIL"
    print(f"{short}: {len(pairs)} operand pairs, {len(bad)} foreign reads, {len(unsafe)} asm loads without s_mov_b64 base  {status}")
    for b in (bad + unsafe)[:5]:
        print("     ", b)
    bad_total += (len(pairs) != want_pairs) + len(bad) + len(unsafe)
sys.exit(1 if bad_total else 0)
