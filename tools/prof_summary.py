"""Summarise a rocprofv3 --kernel-trace --stats csv directory of bench.py: kernel families + per-layer conv table."""
import csv, glob, sys
sys.path.insert(0, '.')
from object_detection_cib_amd.engine.graph import build_graph

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open((glob.glob(d + '/*/*kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv'))[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("== kernel stats (all dispatches)")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print("%6.2f%% calls=%5s avg=%9.1fus  %s" % (100 * float(r['TotalDurationNs']) / tot, r['Calls'], float(r['AverageNs']) / 1e3, r['Name'][:100]))
tr = list(csv.DictReader(open((glob.glob(d + '/*/*kernel_trace.csv') + glob.glob(d + '/*kernel_trace.csv'))[0])))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'nchw_to_nhwc4' in r['Kernel_Name']]
# step windows are delimited by the input re-layout kernel; of `bench.py --steps K --warmup W` the last three windows
# are the eager profiling steps (one stream, no overlap), the K + 1 before them are hipGraph replays: take a replay
k = len(idx) - 6 if len(idx) >= 8 else max(len(idx) - 2, 0)
step = tr[idx[k]:idx[k + 1]] if len(idx) > 1 else tr[idx[-1]:]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (int(tr[idx[k + 1]]['Start_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6 if len(idx) > 1 else 0.0
print("== one replayed step (window %d of %d; under the profiler): %d dispatches on queues %s, sum of kernel time %.2f ms "
      "(kernels of different queues overlap), span %.2f ms" % (
          k, len(idx) - 1, len(step), sorted(set(r['Queue_Id'] for r in step)), sum(dur(r) for r in step) / 1e3, span))
fam = {}
for r in step:
    n = r['Kernel_Name']
    key = n.replace('(anonymous namespace)::', '').replace('void ', '')
    if key.startswith('_ZN'):
        import re as _re
        m = _re.match(r'_ZN\d+_GLOBAL__N_1(\d+)', key)
        key = key[m.end():m.end() + int(m.group(1))] if m else key
    key = key.split('(')[0][:60]
    fam.setdefault(key, [0, 0.0]); fam[key][0] += 1; fam[key][1] += dur(r)
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
    print("   %8.1f us  x%-4d %s" % (t, c, k))
