#!/usr/bin/env python3
"""List the s_waitcnt vmcnt(...) instructions hipcc inserted itself (outside inline asm) INSIDE loops of each kernel of an
ISA listing (hipcc -S --cuda-device-only).  A compiler vmcnt(0) inside a loop that runs an LDS-DMA ring drains the ring
every iteration: the counted hand-written waits are invisible to the compiler, so any ordinary VGPR load it still
considers pending at the loop header (issued before the loop, never consumed before it) gets waited for inside.

    python tools/vmcnt_scan.py /tmp/k/attention.s
"""
import re
import sys


def scan(path):
    kernel, in_asm, in_loop, depth = None, False, False, 0
    out = []
    for n, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, in_loop = m.group(1), False
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        lab = re.match(r"^\.LBB\d+_\d+:(.*)", line)
        if lab:
            in_loop = "Loop" in lab.group(1)
            d = re.search(r"Depth=(\d+)", lab.group(1))
            depth = int(d.group(1)) if d else 0
        if "s_waitcnt" in line and "vmcnt" in line and not in_asm and in_loop and kernel:
            out.append((kernel, n, "depth %d: %s" % (depth, line.strip())))
    return out


if __name__ == "__main__":
    for k, n, l in scan(sys.argv[1]):
        print("%s:%d: %s" % (k[:90], n, l))
