#!/usr/bin/env python3
"""ISA histogram of a kernel's hot loop: hipcc -S of csrc/svr_hip.hip for gfx950, the kernel's basic blocks, and for the block(s) that hold
the PSF evaluation (most v_pk_fma_f32) the instruction mix by class.

usage: python tools/isa_hist.py [--kernel MANGLED_SUBSTRING ...] [-DNAME=VALUE ...] [--asm FILE] [--dump DIR]
default kernels: the on-the-fly cell scatter and gather of SVR (support 16) and of the patch-based path (support 12)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fetalreconstruction_amd import build as B  # noqa: E402

DEFAULT = ["back_cell_kernelILi16ELb0ELb0E", "fwd_cell_kernelILi16ELb0ELb0ELb0E", "back_cell_kernelILi12ELb1ELb0E", "fwd_cell_kernelILi12ELb1ELb0ELb0E"]


def classify(op):
    if op.startswith("v_pk_"):
        return "VALU packed f32 (" + op + ")"
    if op in ("v_cndmask_b32", "v_cndmask_b32_e32", "v_cndmask_b32_e64"):
        return "VALU select (v_cndmask)"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "VALU compare"
    if op.startswith("v_rndne"):
        return "VALU v_rndne_f32"
    if op.startswith(("v_lshrrev", "v_lshlrev", "v_sub_u32", "v_sub_nc", "v_add_u32", "v_and", "v_or", "v_xor", "v_bfe", "v_lshl", "v_add3", "v_mad_u", "v_mul_lo", "v_mul_u", "v_ashr", "v_ffb", "v_bcnt", "v_not", "v_mbcnt", "v_alignbit", "v_perm")):
        return "VALU integer / bit"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "VALU move"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "VALU lane read / write"
    if op.startswith(("v_cvt", "v_ldexp", "v_frexp")):
        return "VALU convert / ldexp"
    if op.startswith(("v_min", "v_max", "v_med")):
        return "VALU min / max"
    if op.startswith("v_"):
        return "VALU other f32 (" + re.sub(r"_e(32|64)$", "", op) + ")"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        return "LDS ds_bpermute"
    if op.startswith("ds_"):
        return "LDS read / write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM " + op.split("_")[0]
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "SALU wait / nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "SALU branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def blocks_of(lines):
    """[(label, [ops], [branch targets])] of one function's text; a block ends at a label or after a branch"""
    out, cur, tg, name, k = [], [], [], "entry", 0
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            if cur:
                out.append((name, cur, tg))
            name, cur, tg, k = m.group(1), [], [], 0
            continue
        op = t.split()[0]
        if re.match(r"^[a-z]", op):
            cur.append(op)
            if op.startswith(("s_cbranch", "s_branch")):
                tg.append(t.split()[1])
                out.append((name, cur, tg))
                k += 1
                name, cur, tg = f"{name.split('+')[0]}+{k}", [], []
    if cur:
        out.append((name, cur, tg))
    return out


def main():
    argv = sys.argv[1:]
    defs = [a for a in argv if a.startswith("-D")]
    kernels = [argv[i + 1] for i, a in enumerate(argv) if a == "--kernel"] or DEFAULT
    asm = argv[argv.index("--asm") + 1] if "--asm" in argv else None
    dump = argv[argv.index("--dump") + 1] if "--dump" in argv else None
    if not asm:
        td = tempfile.mkdtemp()
        asm = os.path.join(td, "svr.s")
        cmd = [B.hipcc(), *[f for f in B.FLAGS if f not in ("-shared", "-fPIC")], *defs, "--cuda-device-only", "-S", "-o", asm, B.SRC]
        subprocess.check_call(cmd)
    text = open(asm).read().splitlines()
    print("# hipcc", " ".join([f for f in B.FLAGS if f not in ("-shared", "-fPIC")] + defs), "--cuda-device-only -S  csrc/svr_hip.hip   (tools/isa_hist.py)")
    for k in kernels:
        start = next((i for i, ln in enumerate(text) if re.match(r"^_ZN.*" + re.escape(k) + r".*:\s", ln)), None)
        if start is None:
            print(f"\n## {k}: not found")
            continue
        end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
        body = text[start + 1:end]
        bl = blocks_of(body)
        npk = lambda ops: sum(1 for o in ops if o.startswith("v_pk_fma_f32"))     # noqa: E731
        hi = max(range(len(bl)), key=lambda i: npk(bl[i][1]))
        hot = [bl[hi][:2]]
        # the innermost loop around the hot block: the closest back edge (a branch to a label at or before it from a block at or after it)
        first = {b[0].split("+")[0]: i for i, b in reversed(list(enumerate(bl)))}
        loops = [(first[t], j) for j, b in enumerate(bl) for t in b[2] if t in first and first[t] <= hi <= j]
        loop = min(loops, key=lambda ab: ab[1] - ab[0]) if loops else None
        if loop:                                                     # every back edge to that header belongs to the loop
            loop = (loop[0], max(j for a_, j in loops if a_ == loop[0]))
        dem = subprocess.run(["c++filt", text[start].split(":")[0]], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem)
        dem = re.sub(r"\(.*$", "", dem)
        print(f"\n## {dem}   ({len(bl)} basic blocks, {sum(len(b[1]) for b in bl)} instructions)")
        if loop:
            a, b = loop
            print(f"the loop over a slot's units ({bl[a][0]} .. {bl[b][0]}): its blocks in layout order -- label: instructions / VALU, ends with")
            lc = collections.Counter()
            for name, ops, tg in bl[a:b + 1]:
                c = collections.Counter(classify(o) for o in ops)
                valu = sum(v for kk, v in c.items() if kk.startswith("VALU"))
                rare = any(o.startswith(("s_swappc", "s_setpc", "v_div_", "v_rcp")) for o in ops) or (sum(o.startswith("v_cndmask") for o in ops) >= 12 and not any(o.startswith("ds_") for o in ops))
                print(f"  {name:>14}: {len(ops):4d} / {valu:4d}   {ops[-1] if ops else ''} {' '.join(tg)}" + ("   (rare path: out-of-line exponential / Taylor branch / NaN fix-up)" if rare else ""))
                if not rare:
                    lc.update(c)
            valu = sum(v for kk, v in lc.items() if kk.startswith("VALU"))
            print(f"sum over the loop's blocks without the rare paths: {sum(lc.values())} instructions, {valu} VALU (an upper bound of one trip: both sides of the remaining branches are counted)")
            for kk, v in sorted(lc.items(), key=lambda kv: (-kv[1], kv[0])):
                print(f"  {v:5d}  {kk}")
        for name, ops in hot:
            c = collections.Counter(classify(o) for o in ops)
            valu = sum(v for kk, v in c.items() if kk.startswith("VALU"))
            print(f"hot block {name}: {len(ops)} instructions, {valu} VALU -- one (pixel, plane) unit = one row of taps per lane")
            for kk, v in sorted(c.items(), key=lambda kv: (-kv[1], kv[0])):
                print(f"  {v:5d}  {kk}")
            if dump:
                os.makedirs(dump, exist_ok=True)
                open(os.path.join(dump, re.sub(r"[^A-Za-z0-9_]", "_", dem) + ".hot.txt"), "w").write("\n".join(ops) + "\n")


if __name__ == "__main__":
    main()
