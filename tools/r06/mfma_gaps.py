#!/usr/bin/env python3
"""Issue pattern of a kernel's MFMA streams: for every run of MFMAs inside a basic block, the histogram of how many VALU / transcendental /
LDS / VMEM / SALU instructions sit between consecutive MFMAs.   python tools/r06/mfma_gaps.py file.s <kernel-substring>"""
import re, sys, collections
asm = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
kernel = None
blocks = []      # (kernel, label, [ops])
cur = None
for l in asm:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kernel = m.group(1); cur = [kernel, "entry", []]; blocks.append(cur); continue
    m = re.match(r"^(\.LBB\d+_\d+):", l.strip())
    if m and kernel:
        cur = [kernel, m.group(1), []]; blocks.append(cur); continue
    s = l.split(";")[0].strip()
    if not s or s.startswith(".") or s.endswith(":") or kernel is None or cur is None:
        continue
    cur[2].append(s.split()[0])
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(TRANS): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
for k, label, ops in blocks:
    if want not in k: continue
    n_mfma = sum(1 for o in ops if kind(o) == "mfma")
    if n_mfma < 20: continue
    gaps = []
    curgap = None
    tot = collections.Counter()
    for o in ops:
        kd = kind(o)
        tot[kd] += 1
        if kd == "mfma":
            if curgap is not None: gaps.append(curgap)
            curgap = collections.Counter()
        elif curgap is not None:
            curgap[kd] += 1
    tail = curgap
    hist = collections.Counter(g["valu"] + g["trans"] for g in gaps)
    before = 0
    for o in ops:
        if kind(o) == "mfma": break
        before += 1
    print(f"{k[-40:]} {label}: {n_mfma} MFMA, totals {dict(tot)}; before first MFMA {before} instr, after last {sum(tail.values())} "
          f"(valu {tail['valu']}, trans {tail['trans']})")
    print("   VALU+trans per MFMA gap: " + ", ".join(f"{n}: {c}" for n, c in sorted(hist.items())))
    print("   trans per gap: " + ", ".join(f"{n}: {c}" for n, c in sorted(collections.Counter(g['trans'] for g in gaps).items())),
          "| waits in gaps:", sum(g["wait"] for g in gaps), "| nops:", sum(g["nop"] for g in gaps), "| scratch/vmem:", sum(g["vmem"] for g in gaps))
