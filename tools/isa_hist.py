"""Static histogram of an annotated gfx950 ISA listing (hipcc -S --cuda-device-only -gline-tables-only):
instructions per opcode class, per loop depth (the `Depth=` of the enclosing basic block's loop header comment)
and per source line.  Developer tool, not part of the product.

  python tools/isa_hist.py /tmp/isa/tpc_chain.s [kernel-substring] [--lines N] [--mindepth D] [--range LO HI]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_fma_f64", "v_add_f64", "v_mul_f64", "v_fmac_f64", "v_max_f64", "v_min_f64")):
        return "fp64"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_", "v_ldexp_f64", "v_frexp", "v_trig", "v_fract_f64", "v_rndne_f64", "v_cvt")):
        return "fp64x"
    if op.startswith("v_cndmask"):
        return "select"
    if op.endswith("_dpp") or "dpp" in op:
        return "dpp"
    if op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr", "v_pk_mov")):
        return "mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane")):
        return "lane"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_"):
        return "valu_int"
    if op.startswith("ds_bpermute"):
        return "bperm"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load"):
        return "sload"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    args = sys.argv[2:]
    kern = None
    nlines = 25
    mindepth = 0
    lo = hi = None
    i = 0
    while i < len(args):
        if args[i] == "--lines":
            nlines = int(args[i + 1]); i += 2
        elif args[i] == "--mindepth":
            mindepth = int(args[i + 1]); i += 2
        elif args[i] == "--range":
            lo, hi = int(args[i + 1]), int(args[i + 2]); i += 3
        else:
            kern = args[i]; i += 1
    files = {}
    cur_fn = None
    depth = 0
    loc = (0, 0)
    by_class = collections.Counter()
    by_depth = collections.Counter()
    by_line = collections.Counter()
    by_line_class = collections.defaultdict(collections.Counter)
    scratch_by_line = collections.Counter()
    re_loc = re.compile(r"\s*\.loc\s+(\d+)\s+(\d+)")
    re_file = re.compile(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?')
    re_fn = re.compile(r"^([A-Za-z_][\w.$]*):")
    re_depth = re.compile(r"Depth=(\d+)")
    re_bb = re.compile(r"^\.LBB\d+_\d+:")
    re_ins = re.compile(r"^\s+([a-z][a-z0-9_]+)\b")
    for ln in open(path, errors="replace"):
        m = re_file.match(ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        m = re_fn.match(ln)
        if m and not ln.startswith(".L"):
            cur_fn = m.group(1)
            depth = 0
            continue
        if re_bb.match(ln):
            md = re_depth.search(ln)
            depth = int(md.group(1)) if md else 0
            continue
        if "Depth=" in ln and ln.lstrip().startswith(";"):
            md = re_depth.search(ln)
            if md:
                depth = int(md.group(1))
            continue
        m = re_loc.match(ln)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        if kern and (cur_fn is None or kern not in cur_fn):
            continue
        m = re_ins.match(ln)
        if not m or ln.lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        if depth < mindepth:
            continue
        if lo is not None and not (files.get(loc[0], "").startswith("physics_tp") and lo <= loc[1] <= hi):
            continue
        c = classify(op)
        by_class[c] += 1
        by_depth[depth] += 1
        key = (files.get(loc[0], "?"), loc[1])
        by_line[key] += 1
        by_line_class[key][c] += 1
        if c == "scratch":
            scratch_by_line[(key, op.split("_")[1], depth)] += 1
    tot = sum(by_class.values())
    print("total", tot)
    print("by class:", ", ".join(f"{k} {v}" for k, v in by_class.most_common()))
    print("by depth:", dict(sorted(by_depth.items())))
    print("top lines:")
    for k, v in by_line.most_common(nlines):
        print(f"  {k[0]}:{k[1]}  {v}  " + ", ".join(f"{a} {b}" for a, b in by_line_class[k].most_common(5)))
    if scratch_by_line:
        print("scratch by line (file:line, load/store, depth):")
        for (k, kind, d), v in sorted(scratch_by_line.items(), key=lambda x: (-x[0][2], x[0][0])):
            print(f"  {k[0]}:{k[1]} {kind} depth {d}: {v}")


main()
