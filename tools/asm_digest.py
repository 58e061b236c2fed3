"""Per-kernel digest of a gfx950 assembly file (hipcc -save-temps), comments / debug directives / label numbers stripped: two source trees whose
kernels compile to the same instructions give the same table.  Round 6 used it to prune the rejected JSLP_PIPE_* switches out of
jslp_resident_pipe.hip.h: every k_simplex_resident instance before == after.
  python tools/asm_digest.py file.s [name filter]"""
import hashlib
import re
import sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, out = None, {}
for l in open(path):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1); out[cur] = []
        continue
    if l.startswith(".Lfunc_end"):
        cur = None
        continue
    if cur is None:
        continue
    t = l.split(";")[0].rstrip()
    if not t.strip() or re.match(r"^\s*\.(loc|file|cfi|p2align|type|size|ident|section)", t):
        continue
    t = re.sub(r"\.L(BB|tmp|func)\w*", "L", t)
    out[cur].append(t.strip())
for k in sorted(out):
    if flt in k:
        print(hashlib.md5("\n".join(out[k]).encode()).hexdigest()[:12], len(out[k]), k[:110])
