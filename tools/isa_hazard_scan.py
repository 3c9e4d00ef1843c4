"""Scan the library's gfx950 ISA for the instruction forms around the round-4 corruption of the training head's gradient rows
(profiles/r05_head_store_hazard.txt).  Every source of csrc/ is compiled to assembly with the build's flags; per kernel:
  * `cross`: packed-f32 VALU instructions (v_pk_*_f32) whose LOW lane reads the HIGH half of a register pair (an op_sel bit set on a VGPR
    source) — the form every wrong value of the failing build came out of;
  * a packed-f32 result consumed as the data of a global / LDS store within `--raw` instructions, and a store's data registers overwritten
    by a packed-f32 VALU within `--war` instructions behind it (`--stores`: listed only on request — 400+ places, all bit-deterministic).

    python tools/isa_hazard_scan.py [--raw 2] [--war 3] [--flags "..."] [files...]
"""
import argparse, collections, concurrent.futures, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-gcp_amd", "csrc")
# per-file flags: the list csrc/build.sh reads (csrc/sources.sh)
NO_SLP = tuple(n + ".hip" for n in re.search(r'GCPX_NO_SLP="([^"]*)"', open(os.path.join(CSRC, "sources.sh")).read()).group(1).split())


def regs(tok):
    tok = tok.strip().strip(",")
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    if m:
        return set([int(m.group(1))])
    m = re.fullmatch(r"-?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(line):
    s = line.split(";")[0].strip()
    if not s or s.startswith(".") or s.endswith(":"):
        return None
    parts = s.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    return op, ops, s


def scan_file(path, flags, raw_n, war_n):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        extra = ["-fno-slp-vectorize"] if os.path.basename(path) in NO_SLP else []
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only"] + extra + flags + [path, "-o", out]
        subprocess.run(cmd, check=True, cwd=os.path.dirname(path), stderr=subprocess.DEVNULL)
        asm = open(out).read().split("\n")
    hits = []
    kernel, window = None, []
    counts = collections.Counter()
    for l in asm:
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            kernel, window = m.group(1), []
            continue
        if re.match(r"^\.LBB\d+_\d+:", l.strip()):
            window = []                      # (a hazard across a branch target would need both paths: distances are per basic block)
            continue
        p = parse(l)
        if p is None or kernel is None:
            continue
        op, ops, text = p
        is_store = op.startswith(("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write"))
        is_pk = op.startswith("v_pk_") and op.endswith("_f32")
        if is_pk:
            counts[(kernel, "pk")] += 1
            m = re.search(r"op_sel:\[([01,]+)\]", text)
            if m:
                bits = m.group(1).split(",")
                srcs = ops[1:1 + len(bits)]
                if any(b == "1" and regs(src.split()[0]) for b, src in zip(bits, srcs)):
                    counts[(kernel, "cross")] += 1
        if is_store:
            counts[(kernel, "store")] += 1
            data = regs(ops[0]) if op.startswith("buffer_store") else (regs(ops[1]) if len(ops) > 1 else set())
            for back, (pop, pdst, ptext, pstore, pdata) in enumerate(reversed(window[-raw_n:]), 1):
                if pop.startswith("v_pk_") and pop.endswith("_f32") and (pdst & data):
                    hits.append((kernel, "packed result -> store data", back, ptext, text))
        dst = regs(ops[0]) if op.startswith("v_") and ops else set()
        if is_pk:
            for back, (pop, pdst, ptext, pstore, pdata) in enumerate(reversed(window[-war_n:]), 1):
                if pstore and (pdata & dst):
                    hits.append((kernel, "store data overwritten by a packed op", back, ptext, text))
        window.append((op, dst, text, is_store, (regs(ops[0]) if op.startswith("buffer_store") else (regs(ops[1]) if len(ops) > 1 else set())) if is_store else set()))
    return os.path.basename(path), hits, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--raw", type=int, default=2)
    ap.add_argument("--war", type=int, default=3)
    ap.add_argument("--flags", default="")
    ap.add_argument("--stores", action="store_true", help="list the packed-result / store neighbourhoods too")
    args = ap.parse_args()
    files = args.files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    total = 0
    grand = collections.Counter()
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        for name, hits, counts in ex.map(lambda f: scan_file(os.path.abspath(f), args.flags.split(), args.raw, args.war), files):
            per = collections.defaultdict(list)
            for h in hits:
                per[h[0]].append(h)
            npk = sum(v for (k, kind), v in counts.items() if kind == "pk")
            ncross = sum(v for (k, kind), v in counts.items() if kind == "cross")
            print(f"== {name}: {npk} packed-f32 VALU instructions, {ncross} with a low lane reading a high half; {len(hits)} packed-result / store neighbourhood(s)")
            for (k, kind), v in sorted(counts.items()):
                if kind == "cross":
                    print(f"   cross x {v:4d}  {k[:120]}")
            grand["cross"] += ncross
            for k, hs in (per.items() if args.stores else []):
                kinds = collections.Counter(h[1] for h in hs)
                print(f"   {k[:110]}: " + ", ".join(f"{n} x {kind}" for kind, n in kinds.items()))
                for h in hs[:4]:
                    print(f"       [{h[1]}, distance {h[2]}]  {h[3]}   ...   {h[4]}")
            total += len(hits)
    print(f"packed-f32 instructions whose low lane reads a high half: {grand['cross']}; packed-result / store neighbourhoods: {total}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
