"""Per-validation-batch GPU timeline from a rocprofv3 kernel trace of tools/bench_eval.py (diagnostic)."""
import csv, glob, sys, collections
f = (glob.glob(sys.argv[1] + '/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[0]
tr = list(csv.DictReader(open(f)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'decode_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
seg = tr[a:b]
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
print(f"one batch: {len(seg)} kernels, span {span:.2f} ms, busy {busy:.2f} ms")
fam = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:50]
    fam[k][0] += 1; fam[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t:9.1f} us x{n:<4d} {k}")
gaps = sorted(((int(y['Start_Timestamp']) - int(x['End_Timestamp'])) / 1e3, x['Kernel_Name'][:45], y['Kernel_Name'][:45]) for x, y in zip(seg, seg[1:]))[::-1][:6]
for g in gaps:
    print("  gap %.0f us after %s before %s" % g)
