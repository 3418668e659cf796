"""Second PMC pass (tools/prof_kernels.sh, mix2): share of wavefront cycles waiting / issuing, LDS conflicts.
  python tools/pmc_mix2.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    if not n[k]:
        continue
    w = v["SQ_WAVES"] or 1; cyc = v["SQ_WAVE_CYCLES"] or 1
    print(f"{k:62s} waves {w / n[k]:8.0f} cyc/w {cyc / w:8.0f} wait_any {100 * v['SQ_WAIT_ANY'] / cyc:5.1f}% active_any {100 * v['SQ_ACTIVE_INST_ANY'] / cyc:5.1f}% "
          f"lds_active {100 * v['SQ_ACTIVE_INST_LDS'] / cyc:5.1f}% lds_conflict/w {v['SQ_LDS_BANK_CONFLICT'] / w:8.0f} vmwr/w {v['SQ_INSTS_VMEM_WR'] / w:5.0f} busy_cyc {v['SQ_BUSY_CYCLES'] / n[k]:10.0f}")
