"""Turns the rocprofv3 sqlite outputs that gpurun brought back (gpurun_out/<dir>/run_results.db) into the
summaries committed under profiles/: the per-kernel statistics CSV of the --kernel-trace --stats run and the
HBM traffic per launch from the two PMC passes (FETCH_SIZE, WRITE_SIZE; gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-byte request of wide coalesced reads).

  python tools/export_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir>
  python tools/export_profiles.py <tag> <stats_dir>                          (kernel statistics only)
"""
import csv, json, sqlite3, sys

SHORT = {'k_hq_unpack': 'hq_unpack', 'k_hq_pack': 'hq_pack', 'k_inv_fast<0, true': 'idwt_level_final',
         'k_inv_fast<0, false': 'idwt_level', 'k_fwd_fast<0, true': 'dwt_level_first', 'k_fwd_fast<0, false': 'dwt_level',
         'k_compact': 'slice_compact', 'k_scan_sizes': 'slice_offsets_scan', 'k_index_tables_nx': 'slice_index_tables',
         'k_index_group': 'slice_index_chain(group)', 'k_index_chain': 'slice_index_chain(chain)', 'k_index_emit': 'slice_index_emit'}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def main(tag, stats, fetch=None, write=None):
    c = sqlite3.connect(f'gpurun_out/{stats}/run_results.db')
    rows = c.execute("select name,count(*),sum(duration),avg(duration),min(duration),max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(f'profiles/{tag}_rocprofv3_kernel_stats.csv', 'w', newline='') as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(100 * r[2] / tot, 2), int(r[4]), int(r[5])])
    if fetch is None:
        return
    out = {}
    for name, d in (('FETCH_SIZE', fetch), ('WRITE_SIZE', write)):
        c = sqlite3.connect(f'gpurun_out/{d}/run_results.db')
        acc = {}
        for kn, cnt, tot in c.execute("select kernel_name,count(*),sum(value) from counters_collection where counter_name=? "
                                      "group by kernel_name", (name,)):
            s = short(kn)
            if s:  # several instantiations (and launch sizes) can share a short name: average over all their launches
                a = acc.setdefault(s, [0, 0.0])
                a[0] += cnt
                a[1] += tot
        for s, (cnt, tot) in acc.items():
            out.setdefault(s, {})[name + '_KiB'] = tot / cnt
            out[s]['launches_sampled'] = cnt
    res = {"_comment": "rocprofv3 PMC, two separate passes (--kernel-trace --pmc FETCH_SIZE ; --kernel-trace --pmc WRITE_SIZE) of "
                       "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` (batch 16 UHD cfg-2 pictures per launch), averages "
                       "per launch. FETCH_SIZE/WRITE_SIZE are in KiB. hbm_bytes_per_launch applies the gfx950 correction of "
                       "MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of wide coalesced reads): 2*FETCH + WRITE; calibrated on hq_pack, "
                       "which reads each of its 1062 MB of coefficients exactly once with 16-byte loads (FETCH_SIZE 535 MB). "
                       "Made by tools/export_profiles.py from the " + tag + " runs.",
           "pictures_per_launch": 16, "kernels": {}}
    for k, v in sorted(out.items()):
        f, w = v.get('FETCH_SIZE_KiB', 0) * 1024, v.get('WRITE_SIZE_KiB', 0) * 1024
        res['kernels'][k] = {"launches_sampled": v['launches_sampled'], "FETCH_SIZE_KiB": round(v.get('FETCH_SIZE_KiB', 0), 1),
                             "WRITE_SIZE_KiB": round(v.get('WRITE_SIZE_KiB', 0), 1), "hbm_bytes_per_launch": int(2 * f + w)}
        print(f"{k:28s} fetch {f / 1e6:8.1f} MB  write {w / 1e6:8.1f} MB  hbm {(2 * f + w) / 1e6:8.1f} MB")
    json.dump(res, open('profiles/r01_pmc_traffic.json', 'w'), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:5])
