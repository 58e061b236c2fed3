#!/usr/bin/env python3
"""Static instruction mix of one kernel of libjslp_hip.so, segment by segment (CPU only: hipcc cross-compiles gfx950).

  python tools/isa_mix.py [--kernel SUBSTR] [--from N --to M] [--extra-flags "-D..."]

Compiles jslpsolver_amd/csrc/jslp_hip.hip with -save-temps into a scratch directory, cuts the kernel whose mangled name contains SUBSTR
(default: the headline instance of the lean register-resident kernel, cycle check off) out of the .s file and prints, for every
stretch between two workgroup barriers, how many vector / scalar / LDS / memory instructions, cross-lane reads (v_readlane: DPP tails,
broadcasts and SGPR-spill reloads), v_writelane (SGPR spills), waits and branches it holds.  With 16 waves on 4 SIMDs every instruction
that all waves execute costs ~16 cycles of the pivot, so these counts are what DESIGN.md section 8 prices the loop with.  The pivot loop
of phase 2 is the LAST long run of segments in front of the kernel's epilogue (pricing: three ds_min / ds_max rounds; then the ratio
test with its division, the update pass, the poll loop with s_sleep, the fetch loop, the normalisation with two or three divisions);
--from / --to restrict the listing to a range of segment numbers."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = "k_simplex_residentILi1024ELi2ELi8ELb0ELb1ELb0ELb0ELb0EE"  # <1024, 2, 8, OPT=0, LEAN=1, UNR=0, CHK=0, XL=0>

CLASSES = [
    ("valu", re.compile(r"^\s*v_(?!readlane|writelane|readfirstlane)")),
    ("salu", re.compile(r"^\s*s_(?!waitcnt|barrier|cbranch|branch|nop|sleep|endpgm)")),
    ("readlane", re.compile(r"^\s*v_read(first)?lane")),
    ("writelane", re.compile(r"^\s*v_writelane")),
    ("lds", re.compile(r"^\s*ds_")),
    ("mem", re.compile(r"^\s*(buffer_|global_|flat_|scratch_|s_load|s_buffer_load)")),
    ("waitcnt", re.compile(r"^\s*s_waitcnt")),
    ("branch", re.compile(r"^\s*s_c?branch")),
    ("nop/sleep", re.compile(r"^\s*s_(nop|sleep)")),
    ("div", re.compile(r"^\s*v_rcp_f64")),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default=DEFAULT)
    ap.add_argument("--from", dest="lo", type=int, default=0)
    ap.add_argument("--to", dest="hi", type=int, default=1 << 30)
    ap.add_argument("--extra-flags", default="")
    ap.add_argument("--asm", default=None, help="an existing .s file instead of compiling")
    a = ap.parse_args()
    if a.asm:
        asm = a.asm
    else:
        tmp = tempfile.mkdtemp(prefix="jslp_isa_")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
               "-save-temps=obj", "-o", os.path.join(tmp, "lib.so"), os.path.join(ROOT, "jslpsolver_amd", "csrc", "jslp_hip.hip")] + a.extra_flags.split()
        subprocess.run(cmd, check=True, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = next(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f)
    lines, on = [], False
    for l in open(asm):
        if not on and re.match(r"^_Z\w*" + re.escape(a.kernel) + r"\w*:", l):
            on = True
        if on:
            lines.append(l)
            if l.startswith(".Lfunc_end"):
                break
    if not lines:
        sys.exit("no kernel whose name contains %r in %s" % (a.kernel, asm))
    segs, cur = [], []
    for l in lines:
        cur.append(l)
        if re.match(r"^\s*s_barrier", l):
            segs.append(cur)
            cur = []
    segs.append(cur)
    print("kernel %s: %d lines, %d segments (each ends at an s_barrier)" % (lines[0].strip().rstrip(":"), len(lines), len(segs)))
    print("| seg | " + " | ".join(n for n, _ in CLASSES) + " | landmarks |")
    print("|---|" + "---|" * (len(CLASSES) + 1))
    # round 5: which v_readlane are RELOADS OF SPILLED SGPRs?  The register allocator parks spilled scalars in the lanes of a few reserved
    # VGPRs (the destinations of v_writelane); a v_readlane whose source is one of those is a reload, every other one is data movement the
    # source asked for (DPP tails, broadcast multipliers of the update pass).  Printed as readlane = total (reloads).
    spill_vgprs = {m.group(1) for l in lines for m in [re.match(r"^\s*v_writelane_b32 (v\d+),", l)] if m}
    reload_rx = re.compile(r"^\s*v_readlane_b32 s\d+, (v\d+),")
    tot = [0] * len(CLASSES)
    tot_reload = 0
    for i, sg in enumerate(segs):
        if not (a.lo <= i <= a.hi):
            continue
        c = [sum(1 for l in sg if rx.match(l)) for _, rx in CLASSES]
        n_reload = sum(1 for l in sg for m in [reload_rx.match(l)] if m and m.group(1) in spill_vgprs)
        tot_reload += n_reload
        marks = sorted({m for l in sg for m in re.findall(r"^\s*(ds_(?:min|max)\w*|buffer_wbl2|s_sleep|buffer_store_dwordx4|buffer_load_dwordx4|s_endpgm)", l)})
        tot = [x + y for x, y in zip(tot, c)]
        cs = [str(x) for x in c]
        cs[2] = "%d (%d)" % (c[2], n_reload)
        print("| %d | " % i + " | ".join(cs) + " | " + " ".join(marks) + " |")
    ts = [str(x) for x in tot]
    ts[2] = "%d (%d)" % (tot[2], tot_reload)
    print("| total | " + " | ".join(ts) + " | |")
    print("(readlane column: all v_readlane / v_readfirstlane, in brackets those that reload a spilled SGPR from %s)" % (", ".join(sorted(spill_vgprs)) or "no spill VGPR"))


if __name__ == "__main__":
    main()
