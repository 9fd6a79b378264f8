"""Per-draw sums of the passes tools/profile_joint_r5.sh wrote: for either path the kernels of ONE k = 3 joint draw (the run's last
five draws are that iteration's: bench_joint.py --only-k 3), durations from the kernel trace, counters from the PMC passes."""
import collections, csv, glob, sys

out = sys.argv[1]
DRAWS = 5
# launches per draw, in stream order
PER_DRAW = {"valu": [("joint_kernel", ["all phases"]), ("joint_eigh_kernel", ["eigh root"])],
            "mfma": [("joint_test_mfma_kernel", ["factor mode", "test mode"]), ("joint_kernel", ["CHOL phase", "TAIL phase"]),
                     ("joint_eigh_kernel", ["eigh root"])]}


def short(name):
    for k in ("joint_test_mfma_kernel", "joint_eigh_kernel", "joint_kernel"):
        if k in name:
            return k
    return None


def last_draws(rows, path):
    """rows: [(dispatch id, kernel, value)] -> {(kernel, role): mean per launch over the last DRAWS draws}"""
    res = {}
    for kern, roles in PER_DRAW[path]:
        mine = sorted((d, v) for d, k, v in rows if k == kern)
        take = mine[-DRAWS * len(roles):]
        if len(take) < DRAWS * len(roles):
            continue
        for r, role in enumerate(roles):
            vals = [v for i, (d, v) in enumerate(take) if i % len(roles) == r]
            res[(kern, role)] = sum(vals) / len(vals)
    return res


for path in ("valu", "mfma"):
    print(f"== path {path}: one k = 3 draw (car, Ns = 1024, H = 40, 360 + 1 + 120 label rows; mean of the last {DRAWS} draws)")
    for f in glob.glob(f"{out}/{path}_trace/*kernel_trace.csv"):
        rows = [(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
                for r in csv.DictReader(open(f)) if short(r["Kernel_Name"])]
        res = last_draws(rows, path)
        for k, v in res.items():
            print(f"   duration  {k[0]:24s} {k[1]:12s} {v:9.1f} us")
        print(f"   duration  per draw {sum(res.values()):9.1f} us")
    totals = collections.defaultdict(float)
    for sub in ("fetch", "write", "sq", "sq2"):
        for f in glob.glob(f"{out}/{path}_{sub}/*counter_collection.csv"):
            by = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if short(r["Kernel_Name"]):
                    by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
            for cname, rows in sorted(by.items()):
                res = last_draws(rows, path)
                for k, v in res.items():
                    print(f"   {cname:26s} {k[0]:24s} {k[1]:12s} {v:12.5g}")
                    totals[cname] += v
    if totals:
        print("   per draw: " + "  ".join(f"{k} {v:.5g}" for k, v in sorted(totals.items())))
        if "FETCH_SIZE" in totals and "WRITE_SIZE" in totals:
            # FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them on gfx950: KB (MI355X_MICROARCH.md, HBM section)
            f_, w_ = totals['FETCH_SIZE'] * 1024 / 1e9, totals['WRITE_SIZE'] * 1024 / 1e9
            print(f"   FETCH + WRITE per draw: {f_ + w_:.3f} GB (fetch {f_:.3f}, write {w_:.3f}; as reported, the convention of "
                  f"profiles/r3_joint_counters.md and r4; with the guide's 2x for wide coalesced reads at most {2 * f_ + w_:.3f} GB)")
        if "SQ_WAIT_ANY" in totals and "SQ_WAVE_CYCLES" in totals:
            print(f"   SQ_WAIT_ANY / SQ_WAVE_CYCLES per draw: {totals['SQ_WAIT_ANY'] / totals['SQ_WAVE_CYCLES']:.3f}")
