#!/bin/bash
# instruction-cache and issue counters of the kernels of tools/time_cfg.py cfg2@32 (gpurun -- tools/pmc_icache.sh <tag> [env...])
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run through gpurun}" || exit 1
out=gpurun_out/ic_$tag
mkdir -p $out
for e in "$@"; do export "$e"; done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $out/a -o run --output-format csv -- python3 tools/time_cfg.py cfg2@32 > /dev/null 2> $out/a.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_IFETCH -d $out/b -o run --output-format csv -- python3 tools/time_cfg.py cfg2@32 > /dev/null 2> $out/b.err
python3 - $out <<'P'
import collections, csv, glob, sys
for sub in ("a", "b"):
    f = glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:9]:
        w = v["SQ_WAVES"] or 1
        print(f"{k:60s} n {n[k]:3d} waves {w/n[k]:7.0f} " + " ".join(f"{c[3:] if c.startswith('SQ_') else c[4:]}/w {x/w:9.0f}" for c, x in sorted(v.items()) if c != "SQ_WAVES"))
P
