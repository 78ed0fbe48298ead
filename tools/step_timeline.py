"""Timeline of one REPLAYED step from a rocprofv3 kernel trace of `bench.py --steps K --warmup W` (step windows are
delimited by nchw_to_nhwc4; windows W+1 .. W+K are hipGraph replays, the last three are the eager profiling steps):
idle gaps, queue hand-overs, and optionally every kernel.  usage: step_timeline.py <trace dir> [window index] [--all]"""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*kernel_trace.csv') + glob.glob(d + '/*/*kernel_trace.csv')
tr = list(csv.DictReader(open(f[0]))); tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'nchw_to_nhwc4' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip('-').isdigit() else len(idx) - 6
step = tr[idx[k]:idx[k + 1]]
t0 = int(step[0]['Start_Timestamp'])
span = int(tr[idx[k + 1]]['Start_Timestamp']) - t0
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
ev = sorted(step, key=lambda r: int(r['Start_Timestamp']))
ce, last, gaps, cov = int(ev[0]['End_Timestamp']), ev[0], [], 0
cs = int(ev[0]['Start_Timestamp'])
for r in ev[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > ce:
        gaps.append((s - ce, last, r)); cov += ce - cs; cs = s
    if e > ce:
        ce, last = e, r
cov += ce - cs
print("window %d of %d: span %.3f ms  covered %.3f ms  idle %.3f ms in %d gaps  kernels %d  sum of durations %.3f ms  queues %s" % (
    k, len(idx) - 1, span / 1e6, cov / 1e6, (span - cov) / 1e6, len(gaps), len(step),
    sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6, sorted(set(r['Queue_Id'] for r in step))))
c = collections.Counter()
for g, a, b in gaps:
    c[(nm(a), "hop" if a['Queue_Id'] != b['Queue_Id'] else "same queue", nm(b))] += g / 1e3
for key, v in c.most_common(14):
    print("  %7.1f us idle between %s" % (v, key))
if "--all" in sys.argv:
    for r in ev:
        print("%9.1f %7.1f q=%s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], nm(r)))
