"""Prints the kernels of one replayed step from a rocprofv3 kernel trace, from `--from-us` on (default: the last 3 ms):
start, end, duration, queue, name.  usage: tools/tail_view.py <dir with *_kernel_trace.csv> [--from-us N] [--step K]"""
import csv, glob, re, sys
d = sys.argv[1]
frm = float(sys.argv[sys.argv.index("--from-us") + 1]) if "--from-us" in sys.argv else None
k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd" in r["Kernel_Name"]]
if "--step" not in sys.argv:
    # the replayed steps share one queue histogram (the eager warm-ups and the per-family timing passes have others): take
    # the middle one of the largest such group
    sig = {}
    for a, b in zip(idx[:-1], idx[1:]):
        h = {}
        for r in rows[a + 1:b + 1]:
            h[r["Queue_Id"]] = h.get(r["Queue_Id"], 0) + 1
        sig.setdefault(tuple(sorted(h.items())), []).append((a, b))
    grp = max((g for g in sig.values()), key=lambda g: (len(g), g[0][0]))
    ip, i = grp[len(grp) // 2]
else:
    i, ip = idx[k], idx[k - 1]
t0 = int(rows[ip]["End_Timestamp"])
span = (int(rows[i]["End_Timestamp"]) - t0) / 1e3
print("step kernels", i - ip, "span %.1f us" % span)
if frm is None:
    frm = span - 3000
busy = {}
for r in rows[ip + 1:i + 1]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)[:60]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy.setdefault(r["Queue_Id"], []).append((s, e))
    if s >= frm:
        print(f"{s:9.1f} {e:9.1f} {e - s:7.1f} q{r['Queue_Id']} {n}")
for q, v in busy.items():
    print("queue", q, "kernels", len(v), "busy %.1f us" % sum(e - s for s, e in v), "first %.1f last %.1f" % (v[0][0], v[-1][1]))
