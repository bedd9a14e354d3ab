#!/usr/bin/env python3
"""usage: tools/prof/report.py <samples file> [top N]  -- resolve tools/prof/sampler.c's PCs to functions (addr2line) and print the
hottest functions and lines, per mapped object."""
import collections, subprocess, sys
path = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
maps, pcs = [], []
for l in open(path):
    if l.startswith("M "):
        f = l[2:].split()
        a, b = [int(x, 16) for x in f[0].split("-")]
        off = int(f[2], 16)
        maps.append((a, b, off, f[5] if len(f) > 5 else "?"))
    elif l.startswith("S "):
        pcs.append(int(l[2:], 16))
by_obj = collections.defaultdict(list)
for pc in pcs:
    for a, b, off, name in maps:
        if a <= pc < b:
            by_obj[name].append(pc - a + off); break
    else:
        by_obj["?"].append(pc)
print(f"{len(pcs)} samples")
for name, offs in sorted(by_obj.items(), key=lambda kv: -len(kv[1])):
    print(f"{len(offs):7d} {100.0 * len(offs) / len(pcs):5.1f}%  {name}")
for name, offs in sorted(by_obj.items(), key=lambda kv: -len(kv[1]))[:3]:
    if name.startswith("[") or name == "?":
        continue
    uniq = collections.Counter(offs)
    out = subprocess.run(["addr2line", "-f", "-C", "-i", "-e", name] + [hex(o) for o in uniq], capture_output=True, text=True).stdout.split("\n")
    # with -i one address can print several (function, line) pairs: re-run without -i for the counts by innermost frame
    out = subprocess.run(["addr2line", "-f", "-C", "-e", name] + [hex(o) for o in uniq], capture_output=True, text=True).stdout.split("\n")
    fn, ln = collections.Counter(), collections.Counter()
    for i, (o, c) in enumerate(uniq.items()):
        f, l = out[2 * i], out[2 * i + 1]
        fn[f[:90]] += c; ln[(f[:50], l.split("/")[-1])] += c
    print(f"\n== {name}: functions")
    for f, c in fn.most_common(top): print(f"{c:7d} {100.0 * c / len(pcs):5.1f}%  {f}")
    print(f"== {name}: lines")
    for (f, l), c in ln.most_common(top): print(f"{c:7d} {100.0 * c / len(pcs):5.1f}%  {l:40s} {f}")
