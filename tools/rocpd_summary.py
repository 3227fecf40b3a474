"""Turn rocprofv3's rocpd sqlite output into the small CSV / JSON summaries kept under profiles/.
usage: rocpd_summary.py stats <results.db> <out.csv>          per-kernel duration statistics (the --stats table)
       rocpd_summary.py pmc <results.db> <out.csv>            per-kernel mean of every collected counter
"""
import csv, sqlite3, statistics, sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = {}
    for name, dur in c.execute("select name, duration from kernels"):
        rows.setdefault(name, []).append(dur)
    total = sum(sum(v) for v in rows.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 3)])


def pmc(db, out):
    c = sqlite3.connect(db)
    acc = {}
    for name, counter, value, dur in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        a = acc.setdefault((name, counter), [0, 0.0, 0.0])
        a[0] += 1; a[1] += value; a[2] += dur
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Kernel", "Counter", "Dispatches", "MeanValue", "MeanDurationNs"])
        for (name, counter), (n, v, d) in sorted(acc.items()):
            w.writerow([name, counter, n, round(v / n, 3), round(d / n, 1)])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
