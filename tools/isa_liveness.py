#!/usr/bin/env python3
"""VGPR liveness of one kernel from hipcc's -S output: where the register demand peaks and what is live there.
    python tools/isa_liveness.py <file.s> <mangled kernel name> [--top N]
A backward data-flow over the kernel's basic blocks (labels / s_branch / s_cbranch_*), VGPRs v0..v255 only; every
instruction's first operand is taken as its definition unless it is a store / ds_write / atomic without return / compare
(what gfx9 assembly does).  Conservative about predication (a v_cndmask or an instruction under EXEC does not kill).
Prints the live count per instruction's maximum, and the source-line markers (.loc) around the peak."""
import re, sys

def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out

NO_DEF = ("global_store", "scratch_store", "buffer_store", "ds_write", "ds_store", "v_cmp", "v_cmpx", "s_", "global_atomic", "ds_add", "ds_max", "ds_min", "v_writelane", "v_readlane", "v_readfirstlane", "flat_store")

def main():
    path, name = sys.argv[1], sys.argv[2]
    text = open(path).read()
    start = re.search(r"^" + re.escape(name) + r":.*$", text, re.M)
    if not start:
        sys.exit("kernel not found")
    end = text.index(".Lfunc_end", start.end())
    lines = text[start.end():end].split("\n")
    ins = []  # (index in lines, mnemonic, defs, uses, branch target, falls through)
    labels = {}
    loc = None
    locs = {}
    for i, l in enumerate(lines):
        t = l.strip()
        lm = re.match(r"^(\.LBB\d+_\d+):", t)
        if lm:
            labels[lm.group(1)] = len(ins)
            continue
        if t.startswith(".loc"):
            loc = t
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        parts = t.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        d, u = [], []
        if mn.startswith(NO_DEF) or not ops:
            for o in ops:
                u += regs(o)
            if mn.startswith(("v_readlane", "v_readfirstlane")):
                u = regs(ops[1]) if len(ops) > 1 else []
            if mn.startswith("global_atomic") and len(ops) >= 3 and "sc0" in t:  # returning atomic: first operand is the result
                d = regs(ops[0]); u = [r for o in ops[1:] for r in regs(o)]
            if mn.startswith("v_writelane"):
                d = []; u = regs(ops[0])  # read-modify-write of a lane: keeps the register alive
        else:
            d = regs(ops[0])
            for o in ops[1:]:
                u += regs(o)
            if mn.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot")) or "op_sel" in t and False:
                u += d  # accumulators read their destination
        tgt = None
        bm = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if bm:
            tgt = bm.group(1)
        ins.append(dict(i=i, mn=mn, d=set(d), u=set(u), tgt=tgt, fall=not mn == "s_branch", text=t, loc=loc))
    n = len(ins)
    live_in = [set() for _ in range(n + 1)]
    changed = True
    while changed:
        changed = False
        for k in range(n - 1, -1, -1):
            x = ins[k]
            out = set()
            if x["fall"]:
                out |= live_in[k + 1]
            if x["tgt"] in labels:
                out |= live_in[labels[x["tgt"]]]
            # a definition under a partial EXEC mask does not kill the old value in the other lanes: only treat full-width
            # definitions in straight-line code as kills when the register is not live-in from a predicated path - we cannot
            # know EXEC here, so loads and plain VALU definitions kill (what the register allocator assumes too)
            new = (out - x["d"]) | x["u"]
            if new != live_in[k]:
                live_in[k] = new
                changed = True
    counts = [len(s) for s in live_in[:n]]
    peak = max(counts)
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 3
    print(f"{name}: {n} instructions, peak live VGPRs = {peak}")
    seen = 0
    last = -100
    for k in sorted(range(n), key=lambda k: -counts[k]):
        if abs(k - last) < 40:
            continue
        last = k
        x = ins[k]
        print(f"  live {counts[k]:3d} at instruction {k} (asm line {x['i']}): {x['text'][:70]}   [{x['loc']}]")
        seen += 1
        if seen >= top:
            break
    return 0

if __name__ == "__main__":
    main()
