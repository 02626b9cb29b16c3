"""Every s_barrier of a kernel that stages through LDS-DMA (global_load_lds) must find vmcnt
drained by the issuing wave before other waves read the staged rows.  hipcc orders an LDS-DMA
only against the ISSUING wave's own LDS reads, so whether an `s_waitcnt vmcnt(0)` lands in front
of the barrier is not guaranteed (round 4: at a loop header of trsm_sweep_kernel it did not, the
back edge carried a fill in flight).  This walks the control-flow graph of every such kernel in the
ISA of `make asm` (build/asm/*.s) with one bit of state -- "an LDS-DMA may be in flight" -- and
lists the barriers that can be reached with it set.  Exit status 1 if any."""
import glob
import os
import re
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "build", "asm")


def analyse(name, body):
    # basic blocks
    blocks, cur, label_of = [], [], {}
    for l in body:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            if cur:
                blocks.append(cur)
            cur = []
            label_of[m.group(1)] = len(blocks)
        cur.append(l)
        if re.search(r"\ts_c?branch", l) or "s_endpgm" in l:
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b[-1]
        m = re.search(r"\ts_(c?branch\w*)\s+(\.LBB\w+)", last)
        if m:
            if m.group(2) in label_of:
                s.append(label_of[m.group(2)])
            if m.group(1).startswith("cbranch") and i + 1 < len(blocks):
                s.append(i + 1)
        elif "s_endpgm" not in last and i + 1 < len(blocks):
            s.append(i + 1)
        succ.append(s)
    state_in = [False] * len(blocks)
    flagged = set()
    changed = True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            p = state_in[i]
            for k, l in enumerate(b):
                if "global_load_lds" in l:
                    p = True
                elif "s_waitcnt" in l and "vmcnt(0)" in l:
                    p = False
                elif "s_barrier" in l and p:
                    flagged.add((i, k))
            for j in succ[i]:
                if p and not state_in[j]:
                    state_in[j] = True
                    changed = True
    nbar = sum(1 for b in blocks for l in b if "s_barrier" in l)
    return nbar, sorted(flagged)


bad = 0
for f in sorted(glob.glob(os.path.join(d, "k_*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
    name, body = None, []
    for line in open(f):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        if name:
            body.append(line)
        if name and line.startswith(".Lfunc_end"):
            if any("global_load_lds" in l for l in body):
                nbar, fl = analyse(name, body)
                bad += len(fl)
                print("%-70s %d barriers, %d reachable with an LDS-DMA in flight" % (name[:70], nbar, len(fl)))
            name = None
print("unsafe barriers:", bad)
sys.exit(1 if bad else 0)
