cd /tmp && export TMPDIR=/tmp
for s in 1 2 4 8 10; do
  export GPMPC_EIGH_SLOTS_PER_CU=$s
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sw$s -o sw -- python $GRAFT_REPO_ROOT/tools/bench_joint.py --car-only > /dev/null 2>&1
  echo "slots/CU=$s: $(grep eigh /tmp/sw$s/sw_kernel_stats.csv | cut -d, -f2-8)"
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("/tmp/sw$s/sw_kernel_trace.csv")) if "joint" in r["Kernel_Name"]]
out=[]
for i in range(len(rows)-1):
    if "joint_kernel" in rows[i]["Kernel_Name"] and "eigh" in rows[i+1]["Kernel_Name"]:
        out.append((rows[i]["Kernel_Name"][27:38], (int(rows[i+1]["End_Timestamp"])-int(rows[i+1]["Start_Timestamp"]))/1e3))
import collections
d=collections.defaultdict(list)
for k,v in out: d[k].append(v)
print({k: round(min(v),1) for k,v in d.items()})
PY
done
