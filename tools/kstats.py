"""Print a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [name filter] [top n]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for r in rows[:top] if not flt else [r for r in rows if flt in r['Name']]:
    n = re.sub(r'\(.*', '', r['Name'])[:48]
    print(f"{n:48s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}  total {float(r['TotalDurationNs'])/1e3:9.1f}")
