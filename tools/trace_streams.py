"""Per-queue busy time of the LAST step in a rocprofv3 kernel trace (csv): the steps are separated by the largest idle gaps;
prints, for every queue that ran kernels in that step, launches / busy time / first start / last end, and the union.
python tools/trace_streams.py <p_kernel_trace.csv> [n_steps_back]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"]) for r in rows))
# find the optimizer kernel as the step delimiter
marks = [i for i, e in enumerate(ev) if "k_opt_adamw" in e[4]]
if len(marks) < back + 1:
    print("not enough steps"); sys.exit(0)
lo, hi = marks[-back - 1] + 1, marks[-back] + 1
step = ev[lo:hi]
t0 = step[0][0]
print(f"step: {len(step)} kernels, span {(max(e[1] for e in step) - t0) / 1e3:.1f} us")
qs = {}
for s, e, q, st, n in step:
    qs.setdefault((q, st), []).append((s, e, n))
for k, v in sorted(qs.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in v)
    print(f"  queue {k[0]:>3} stream {k[1]:>3}: {len(v):5d} kernels, busy {busy / 1e3:8.1f} us, first {(v[0][0] - t0) / 1e3:8.1f} us, last end {(max(e for _, e, _ in v) - t0) / 1e3:8.1f} us")
iv = sorted((s, e) for s, e, _, _, _ in step)
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"  union of busy intervals {u / 1e3:.1f} us, sum of kernel times {sum(e - s for s, e, *_ in step) / 1e3:.1f} us")
