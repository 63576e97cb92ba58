"""Where does a kernel wait outside its MFMA loop?  Reads the `.s` hipcc leaves with `-save-temps` and prints, for one kernel:

  * instruction counts: before the first MFMA (prologue), between first and last MFMA, after the last (epilogue, in LAYOUT order:
    branch targets of the loop that the compiler placed behind the last MFMA are counted here too -- read the skeleton, not the sum);
  * opcode histograms of prologue and epilogue;
  * the skeleton of either end: every memory operation, `s_waitcnt`, barrier and branch with its position, so that patterns such as
    `ds_read -> lgkmcnt(0) -> 4 FMAs -> ds_write` per piece, a `vmcnt(0)` that drains stores in front of a load's first use, or a
    load sunk into its consumer's predicate branch are visible at a glance.

    hipcc --offload-arch=gfx950 -O3 ... -save-temps -c csrc/conv3x3_w32.hip -o /tmp/x.o        (leaves <name>-hip-amdgcn-amd-amdhsa-gfx950.s)
    python tools/isa_segments.py conv3x3_w32-hip-amdgcn-amd-amdhsa-gfx950.s Li16ELi128ELi4ELi1ELb1ELb0ELb0ELb0E [--skeleton epilogue]

The second argument is any substring of the mangled kernel name (template arguments as Itanium mangling writes them).  This is the
tool behind round 5's epilogue / prologue work on the two wide kernels: s_waitcnt count of the GroupNorm
instantiation's epilogue 123 -> 32, no spill, no v_accvgpr_mov shuffle.
"""
import argparse
import collections
import re


def kernel_lines(path, pat):
    out, on = [], False
    for l in open(path):
        if not on and l.startswith("_Z") and pat in l and ": ;" in l:
            on = True
            continue
        if on:
            if ".end_amdhsa_kernel" in l or (l.startswith("_Z") and ": ;" in l):
                break
            out.append(l.rstrip("\n"))
    if not out:
        raise SystemExit("no kernel whose mangled name contains %r in %s" % (pat, path))
    return [l.strip() for l in out if l.strip() and not l.strip().startswith((";", ".", "//")) and not re.match(r"^[\w.$]+:", l.strip())]


def hist(seq, name, top):
    c = collections.Counter(l.split()[0] for l in seq)
    print("== %s: %d instructions" % (name, len(seq)))
    for k, v in c.most_common(top):
        print("   %-32s %d" % (k, v))


SKEL = re.compile(r"^(s_waitcnt|s_barrier|s_cbranch|s_branch|s_endpgm|global_|buffer_|flat_|scratch_|ds_)")


def skeleton(seq, name, per_line):
    print("== %s skeleton (position: instruction)" % name)
    items = ["%d:%s" % (i + 1, " ".join(l.split()[:2])[:30]) for i, l in enumerate(seq) if SKEL.match(l)]
    for i in range(0, len(items), per_line):
        print("   " + " | ".join(items[i:i + per_line]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--skeleton", choices=["prologue", "epilogue", "both", "none"], default="both")
    ap.add_argument("--per-line", type=int, default=5)
    ap.add_argument("--dump", default=None, help="write the epilogue's instructions to this file")
    a = ap.parse_args()
    lines = kernel_lines(a.asm, a.kernel)
    idx = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
    if not idx:
        raise SystemExit("kernel has no MFMA")
    pro, epi = lines[:idx[0]], lines[idx[-1] + 1:]
    spills = sum(1 for l in lines if l.startswith("scratch_"))
    print("MFMAs %d, instructions %d: prologue %d, epilogue (layout order) %d; scratch (spill) instructions %d; s_waitcnt prologue %d / epilogue %d"
          % (len(idx), len(lines), len(pro), len(epi), spills, sum(l.startswith("s_waitcnt") for l in pro), sum(l.startswith("s_waitcnt") for l in epi)))
    hist(pro, "prologue", a.top)
    hist(epi, "epilogue", a.top)
    if a.skeleton in ("prologue", "both"):
        skeleton(pro, "prologue", a.per_line)
    if a.skeleton in ("epilogue", "both"):
        skeleton(epi, "epilogue", a.per_line)
    if a.dump:
        with open(a.dump, "w") as f:
            f.write("\n".join(epi) + "\n")


if __name__ == "__main__":
    main()
