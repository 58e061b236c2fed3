"""Top self-time JS functions of a V8 .cpuprofile (node --cpu-prof): python tools/js_profile.py <file.cpuprofile> [n]"""
import collections, json, sys
d = json.load(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nodes = {n["id"]: n for n in d["nodes"]}
self_t = collections.Counter()
for s, t in zip(d["samples"], d["timeDeltas"]):
    cf = nodes[s]["callFrame"]
    self_t[(cf["functionName"] or "(anon)", cf["url"].split("/")[-1], cf["lineNumber"])] += t
tot = sum(self_t.values())
print("total %.1f ms sampled" % (tot / 1e3))
for k, v in self_t.most_common(top):
    print("%6.2f%% %8.2f ms  %s %s:%d" % (100.0 * v / tot, v / 1e3, k[0], k[1], k[2]))
