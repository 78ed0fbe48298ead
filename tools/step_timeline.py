"""Timeline of one replayed step from a rocprofv3 kernel trace: idle gaps and the non-conv 'glue' kernels."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/*kernel_trace.csv') + glob.glob(d + '/*/*kernel_trace.csv')
tr = list(csv.DictReader(open(f[0]))); tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'nchw_to_nhwc4' in r['Kernel_Name']]
step = tr[idx[-2]:idx[-1]]
t0 = int(step[0]['Start_Timestamp'])
nm = lambda r: r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
cov, (cs, ce), gaps = 0, iv[0], []
for s, e in iv[1:]:
    if s > ce:
        cov += ce - cs; gaps.append((s - ce, ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
cov += ce - cs
span = max(e for _, e in iv) - t0
print("span %.3f ms  covered %.3f ms  idle %.3f ms  kernels %d  sum of durations %.3f ms" % (span / 1e6, cov / 1e6, (span - cov) / 1e6, len(step), sum(e - s for s, e in iv) / 1e6))
for g, at in sorted(gaps, reverse=True)[:8]:
    print("  gap %6.1f us at t=%8.1f us" % (g / 1e3, (at - t0) / 1e3))
glue = ('at::native', 'rocclr', 'loss_', 'assign', 'head_', 'sgd', 'pack_w', 'nchw', 'MODE_HEAD')
for i, r in enumerate(step):
    n = nm(r)
    if any(k in n for k in glue) or ', 2, true>' in n:
        print("%4d %9.1f %7.1f  %s" % (i, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, n))
