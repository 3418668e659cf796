"""Per-kernel instruction mix from a rocprofv3 --pmc ... --output-format csv run (run_counter_collection.csv):
instructions per wavefront by class, cycles per wavefront, share of cycles waiting / with the VALU active.

  python tools/pmc_mix.py <dir>
"""
import collections, csv, glob, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not n[k]:
        continue
    w = v["SQ_WAVES"] or 1
    cyc = v["SQ_WAVE_CYCLES"] or 1
    print(f"{k:46s} launches {n[k]:3d} waves {w / n[k]:8.0f} valu/w {v['SQ_INSTS_VALU'] / w:7.0f} salu/w {v['SQ_INSTS_SALU'] / w:6.0f} "
          f"lds/w {v['SQ_INSTS_LDS'] / w:5.0f} vmrd/w {v['SQ_INSTS_VMEM_RD'] / w:4.0f} cyc/w {cyc / w:8.0f} "
          f"wait {100 * v['SQ_WAIT_INST_ANY'] / cyc:5.1f}% valu_active {100 * v['SQ_ACTIVE_INST_VALU'] / cyc:5.1f}%")
