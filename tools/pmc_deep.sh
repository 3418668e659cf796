cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" || exit 1
out=gpurun_out/pmc_deep; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o run --output-format csv -- python3 tools/time_cfg.py cfg2 > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o run --output-format csv -- python3 tools/time_cfg.py cfg2 > /dev/null 2> $out/write.err
python3 - <<'PY'
import csv, glob, collections
for what in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/pmc_deep/{what}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    print(what, "(KB per launch, min / median / max; launches)")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        if "k_" in k: print(f"  {k:72s} {v[0]:10.0f} {v[len(v)//2]:10.0f} {v[-1]:10.0f}  {len(v)}")
PY
