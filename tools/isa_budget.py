"""Per-phase instruction budget of a kernel from its ISA: hipcc -S with line tables, every instruction attributed to the source phase
it was inlined from (`// @phase <name> <wave|item|matched|rare>` markers in the kernel source; `// @callsite matched|unmatched` on the
two instantiating calls of a lambda that exists once per kind of item), classified as plain VALU / transcendental / MFMA / LDS / VMEM /
SALU.  Static counts x how often a phase runs = the dynamic budget per item; with the workload's item counts, the SQ_INSTS_VALU a PMC run
should report.

    python tools/isa_budget.py video-gcp_amd/csrc/conv3x3_head_split.hip conv3x3_head_split_kernelILi1E [--items 130048 --matched 81920]
"""
import argparse, collections, os, re, subprocess, sys, tempfile

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--items", type=int, default=0, help="items per launch (frames x H/4 x W/16)")
    ap.add_argument("--matched", type=int, default=0, help="items of matched frames per launch")
    ap.add_argument("--flags", default="")
    ap.add_argument("--dump", default="", help="phase whose opcode histogram to print (name or name:runs-per)")
    args = ap.parse_args()
    src = os.path.abspath(args.source)
    base = os.path.basename(src)
    lines = open(src).read().split("\n")
    marks, callsites = [], {}
    for i, l in enumerate(lines, 1):
        m = re.search(r"//\s*@phase\s+(\S+)\s+(\S+)", l)
        if m:
            marks.append((i, m.group(1), m.group(2)))
        m = re.search(r"//\s*@callsite\s+(\S+)", l)
        if m:
            callsites[i] = m.group(1)

    def phase_of(line):
        cur = None
        for (i, n, f) in marks:
            if i <= line:
                cur = (n, f)
        return cur

    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        extra = ["-fno-slp-vectorize"] if base in ("conv3x3_head_split.hip", "loss.hip") else []      # per-file flags of csrc/build.sh
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-gline-tables-only", "-S", "--cuda-device-only"] + extra + args.flags.split() + [src, "-o", out]
        subprocess.run(cmd, check=True, cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
        asm = open(out).read().split("\n")
    start = next(i for i, l in enumerate(asm) if args.kernel in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(asm)) if asm[i].strip().startswith("s_endpgm"))
    first_mark = marks[0][0]
    counts = collections.defaultdict(collections.Counter)      # (phase, freq) -> class counts
    cur = ("unattributed", "item")
    dump = collections.Counter()
    for l in asm[start + 1:end + 1]:
        s = l.strip()
        if s.startswith(".loc"):
            frames = re.findall(r"([\w./+-]+):(\d+):\d+", s.split(";", 1)[1]) if ";" in s else []
            site = None
            chosen = None
            for fn, ln in frames:                                 # innermost first
                ln = int(ln)
                if os.path.basename(fn) != base:
                    continue
                if ln in callsites:
                    site = callsites[ln]
                if chosen is None and ln >= first_mark:
                    chosen = phase_of(ln)
            if chosen:
                n, f = chosen
                if site and f == "item":
                    f = site
                cur = (n, f)
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op = s.split()[0]
        counts[cur][classify(op)] += 1
        if args.dump and args.dump in (cur[0], cur[0] + ":" + cur[1]):
            dump[op] += 1
    if args.dump:
        print(args.dump, dict(dump.most_common()))
    cls = ["valu", "trans", "mfma", "lds", "vmem", "salu"]
    print(f"{'phase':24s} {'runs per':10s} " + " ".join(f"{c:>6s}" for c in cls))
    tot = collections.defaultdict(collections.Counter)
    for (n, f), c in sorted(counts.items(), key=lambda kv: (["wave", "item", "unmatched", "matched", "rare"].index(kv[0][1]) if kv[0][1] in ["wave", "item", "unmatched", "matched", "rare"] else 9, kv[0][0])):
        print(f"{n:24s} {f:10s} " + " ".join(f"{c[k]:6d}" for k in cls))
        tot[f] += c
    print()
    per = {"unmatched item": ["item", "unmatched"], "matched item": ["item", "matched"]}
    res = {}
    for name, fs in per.items():
        c = collections.Counter()
        for f in fs:
            c += tot[f]
        res[name] = c
        print(f"{name:24s} {'':10s} " + " ".join(f"{c[k]:6d}" for k in cls) + f"   VALU incl. transcendentals {c['valu'] + c['trans']}")
    print(f"{'off the main path':24s} {'(static)':10s} " + " ".join(f"{tot['rare'][k]:6d}" for k in cls))
    if args.items:
        um, mt = args.items - args.matched, args.matched
        v = um * (res["unmatched item"]["valu"] + res["unmatched item"]["trans"]) + mt * (res["matched item"]["valu"] + res["matched item"]["trans"])
        t = um * res["unmatched item"]["trans"] + mt * res["matched item"]["trans"]
        m = um * res["unmatched item"]["mfma"] + mt * res["matched item"]["mfma"]
        print(f"\nper launch ({args.items} items, {mt} of matched frames): VALU {v / 1e6:.1f} M wave instructions (of them transcendental {t / 1e6:.1f} M), MFMA {m / 1e6:.1f} M")
        # issue cost per wave instruction on a SIMD, from the round-5 ablation builds of the head kernel (profiles/r05_head_isa_budget.txt):
        # 4 cycles per plain VALU, ~11 per transcendental, 16 per v_mfma_f32_16x16x32_f16; nothing overlaps (two wavefronts per SIMD); 1024 SIMDs
        cyc = ((v - t) * 4 + t * 11 + m * 16) / 1024
        print(f"issue-time model: {cyc / 1e6:.3f} M cycles per SIMD = {cyc / 2.1e9 * 1e3:.3f} ms at 2.1 GHz (MFMA {m * 16 / 1024 / 2.1e9 * 1e3:.3f}, plain VALU {(v - t) * 4 / 1024 / 2.1e9 * 1e3:.3f}, transcendental {t * 11 / 1024 / 2.1e9 * 1e3:.3f})")

if __name__ == "__main__":
    main()
