#!/usr/bin/env python3
"""Which spilled SGPRs does the pivot loop of a k_simplex_resident instance reload, and where were they written?  (CPU only.)

  python tools/spill_lanes.py file.s [mangled-name substring] [first segment of the phase-2 loop]

hipcc spills scalar registers into the lanes of one vector register (v_writelane_b32 vN, sX, lane) and reloads them with v_readlane_b32.
The static "SGPR spills" figure of tools/kernel_resources.py counts the WRITES, most of which happen once per launch (the kernel
arguments, in the prologue).  What a pivot pays is the reloads INSIDE the loop -- this tool lists, per stretch between two workgroup
barriers (the segment numbers of tools/isa_mix.py), the lanes read and the lanes written, and then, for the loop's segments, where each
reloaded lane was last written: `prologue` (segment < 2: a kernel argument or something derived from one at kernel entry), `preheader`
(the segment in front of the loop: invariants the loop's own set-up computes -- buffer descriptors, slot offsets, bounds) or `loop`.
Round 6 (VERDICT r05 #1a, "ResCtx through one pointer"): profiles/r06_sgpr_spill_lanes.md."""
import collections
import re
import sys

path = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "k_simplex_residentILi1024ELi2ELi8ELb0ELb1ELb0ELb0ELb0EE"
asm = open(path).read().split("\n")
start = [i for i, l in enumerate(asm) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l)][0]
end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
seg, rows, n_inst = 0, [], collections.Counter()
for l in asm[start:end]:
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    n_inst[seg] += 1
    if t.startswith("s_barrier"):
        seg += 1
    m = re.match(r"v_readlane_b32 (s\d+), (v\d+), (\d+)", t)
    if m:
        rows.append((seg, "R", m.group(2), int(m.group(3))))
    m = re.match(r"v_writelane_b32 (v\d+), (s\d+), (\d+)", t)
    if m:
        rows.append((seg, "W", m.group(1), int(m.group(3))))
n_seg = seg + 1
writes = collections.Counter(v for s, k, v, _ in rows if k == "W")
spill_v = writes.most_common(1)[0][0]  # (the DPP tails' readlanes read data registers, never written by v_writelane)
rows = [r for r in rows if r[2] == spill_v]
# the phase-2 loop: from the segment given (default: the one after the LAST segment that writes >= 8 lanes -- the loop's preheader) to the end
pre = max(s for s in range(n_seg) if sum(1 for r in rows if r[0] == s and r[1] == "W") >= 8)
loop0 = int(sys.argv[3]) if len(sys.argv) > 3 else pre + 1
print("kernel %s: spill register %s, %d lanes in use, %d v_writelane + %d v_readlane in %d segments; phase-2 loop = segments %d..%d (preheader %d)"
      % (name, spill_v, len({r[3] for r in rows}), sum(1 for r in rows if r[1] == "W"), sum(1 for r in rows if r[1] == "R"), n_seg, loop0, n_seg - 1, pre))
print("| segment | instructions | lanes reloaded (v_readlane) | lanes written (v_writelane) |")
print("|---|---|---|---|")
for s in range(n_seg):
    rd = sorted(r[3] for r in rows if r[0] == s and r[1] == "R")
    wr = sorted(r[3] for r in rows if r[0] == s and r[1] == "W")
    if rd or wr:
        print("| %d%s | %d | %d: %s | %d: %s |" % (s, " (loop)" if s >= loop0 else "", n_inst[s], len(rd), " ".join(map(str, rd)), len(wr), " ".join(map(str, wr))))
origin = collections.Counter()
last_w = {}
for s, k, v, lane in rows:
    if k == "W":
        last_w.setdefault(lane, []).append(s)
for s, k, v, lane in rows:
    if k == "R" and s >= loop0:
        ws = last_w.get(lane, [])
        if any(w >= loop0 for w in ws):
            origin["loop"] += 1
        elif pre in ws:
            origin["preheader"] += 1
        else:
            origin["prologue"] += 1
print("reloads inside the loop's segments by where the lane is written: %s (the last segment also holds the kernel's epilogue: its `prologue` reloads are the"
      " write-back's kernel arguments, once per launch)" % dict(origin))
