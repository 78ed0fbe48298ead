"""HBM traffic per launch of a kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
gfx950 corrections per MI355X_MICROARCH.md (HBM): counters are in KiB units, FETCH_SIZE tallies 128-B requests at
64 B (x2).  usage: pmc_traffic.py <fetch_dir> <write_dir> <substring of kernel name> [more substrings ...]"""
import csv, glob, sys


def collect(d, counter, pats):
    f = (glob.glob(d + '/*counter_collection.csv') + glob.glob(d + '/*/*counter_collection.csv'))[0]
    tot, n = 0.0, set()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter or not all(p in r['Kernel_Name'] for p in pats):
            continue
        tot += float(r['Counter_Value'])
        n.add(r['Dispatch_Id'])
    return tot, len(n)


if __name__ == "__main__":
    fd, wd, pats = sys.argv[1], sys.argv[2], sys.argv[3:]
    f, nf = collect(fd, 'FETCH_SIZE', pats)
    w, nw = collect(wd, 'WRITE_SIZE', pats)
    rd, wr = 2 * f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)
    print(f"kernels matching {pats}: {nf} / {nw} dispatches")
    print(f"read  {rd / 1e6:10.3f} MB per launch (2 x FETCH_SIZE KiB)")
    print(f"write {wr / 1e6:10.3f} MB per launch (WRITE_SIZE KiB)")
    print(f"total {(rd + wr) / 1e6:10.3f} MB per launch")
