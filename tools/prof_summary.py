"""Per-kernel summary of a rocprofv3 --kernel-trace run (sqlite results.db):  python tools/prof_summary.py <db> [steps]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
def short(n):
    n = re.sub(r'^_Z\d+', '', n); n = re.sub(r'\.kd$', '', n)
    return n[:70]
fam = {}
for n, c, t, a in rows:
    f = 'conv_pers' if 'k_conv_pers' in n else 'conv_patch' if 'k_conv_patch' in n else 'wgrad' if 'k_wgrad' in n else 'bn_bwd' if 'k_bn_bwd' in n else 'bn_finalize' if 'k_bn_finalize' in n else 'other'
    fam[f] = fam.get(f, 0) + t
tot = sum(r[2] for r in rows)
print('total kernel ms/step %.3f' % (tot / steps / 1e6), {k: round(v / steps / 1e6, 3) for k, v in fam.items()})
for n, c, t, a in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{c/steps:6.1f}/step {t/steps/1e3:9.1f} us/step  avg {a/1e3:8.1f} us  {short(n)}")
