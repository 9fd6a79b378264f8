"""Per-kernel summary (calls, avg / min / max us) of a rocprofv3 rocpd database; --seq prints the dispatch sequence of gpmpc kernels.
    python tools/debug/rocpd_kernels.py gpurun_out/x/x_results.db [--seq N] [--filter substr]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = 'kernel_name' if 'kernel_name' in cols else ('display_name' if 'display_name' in cols else cols[-1])
names = {r[0]: r[1] for r in cur.execute(f"select id, {namecol} from {ks}")}
rows = list(cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd} order by start"))
flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else None
d = collections.defaultdict(list)
for kid, s, e, g, w in rows:
    n = names.get(kid, str(kid))
    if flt and flt not in n:
        continue
    d[(n[:110], w)].append((e - s) / 1e3)
tot = sum(sum(v) for v in d.values())
for (n, w), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / tot < 0.002:
        continue
    print(f"{len(v):6d} x avg {sum(v)/len(v):10.2f} us  min {min(v):10.2f}  max {max(v):10.2f}  {100*sum(v)/tot:5.1f}%  wg {w:4d}  {n}")
if '--seq' in sys.argv:
    N = int(sys.argv[sys.argv.index('--seq') + 1])
    sel = [(names.get(k, ''), s, e, g, w) for k, s, e, g, w in rows if 'gpmpc' in names.get(k, '')]
    for n, s, e, g, w in sel[-N:]:
        print(f"{(e-s)/1e3:10.2f} us  grid {g:8d} wg {w:4d}  {n[:100]}")
