"""Summarise a rocprofv3 *_kernel_stats.csv: per-step time per kernel (python tools/kstats.py file.csv [steps])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time per step: {tot / steps / 1e6:.3f} ms over {len(rows)} kernels')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']) / steps:7.1f} calls {float(r['TotalDurationNs']) / steps / 1e3:9.1f} us  avg {float(r['AverageNs']) / 1e3:8.1f} us")
