"""Turns the rocprofv3 outputs that gpurun brought back from tools/prof_kernels.sh (gpurun_out/prof_<tag>/: CSV files of
the --kernel-trace --stats run and of the PMC passes) into the summaries committed under profiles/: the per-kernel
statistics, the instruction mix, and the HBM traffic per launch from the two traffic passes (FETCH_SIZE, WRITE_SIZE;
gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-byte request of wide
coalesced reads).  The traffic file records the digest of the kernel sources it was measured on (bench.py prints its
numbers only for those sources).

  python tools/export_profiles.py <tag>            (reads gpurun_out/prof_<tag>/)
"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHORT = {'k_hq_unpack': 'hq_unpack', 'k_hq_pack': 'hq_pack', 'k_inv_fast<0, true': 'idwt_level_final',
         'k_inv_fast<0, false': 'idwt_level', 'k_fwd_fast<0, true': 'dwt_level_first', 'k_fwd_fast<0, false': 'dwt_level',
         'k_inv_stream<0, true': 'idwt_level_final', 'k_inv_stream<0, false': 'idwt_level',
         'k_fwd_stream<0, true': 'dwt_level_first', 'k_fwd_stream<0, false': 'dwt_level',
         'k_fwd_pair<0, true': 'dwt_pair_first', 'k_fwd_pair<0, false': 'dwt_pair',          # round 5: two levels per launch
         'k_inv_pair<0, true': 'idwt_pair_final', 'k_inv_pair<0, false': 'idwt_pair',
         'k_compact': 'slice_compact', 'k_scan_sizes': 'slice_offsets_scan', 'k_index_tables_nx': 'slice_index_tables',
         'k_index_group': 'slice_index_chain(group)', 'k_index_chain': 'slice_index_chain(chain)', 'k_index_emit': 'slice_index_emit'}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def main(tag):
    import bench
    d = f'gpurun_out/prof_{tag}'
    shutil.copy(f'{d}/stats/run_kernel_stats.csv', f'profiles/{tag}_rocprofv3_kernel_stats.csv')
    if os.path.exists(f'{d}/bench_under_rocprof.json'):
        shutil.copy(f'{d}/bench_under_rocprof.json', f'profiles/{tag}_bench_under_rocprof.json')
    if os.path.exists(f'{d}/stats2/run_kernel_stats.csv'):   # the batch cut over two streams: kernels of the two halves overlap
        shutil.copy(f'{d}/stats2/run_kernel_stats.csv', f'profiles/{tag}_rocprofv3_kernel_stats_two_streams.csv')
        shutil.copy(f'{d}/bench_under_rocprof_two_streams.json', f'profiles/{tag}_bench_under_rocprof_two_streams.json')
    with open(f'profiles/{tag}_pmc_instruction_mix.txt', 'w') as f:
        f.write('# rocprofv3 --kernel-trace --pmc ... -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (tools/prof_kernels.sh); '
                'SQ_WAVE_CYCLES and the WAIT / ACTIVE counters are in 4-cycle units\n')
        f.write(subprocess.run([sys.executable, 'tools/pmc_mix.py', f'{d}/mix'], capture_output=True, text=True).stdout)
        f.write(subprocess.run([sys.executable, 'tools/pmc_mix2.py', f'{d}/mix2'], capture_output=True, text=True).stdout)
    out = {}
    for name, sub in (('FETCH_SIZE', 'fetch'), ('WRITE_SIZE', 'write')):
        acc = {}
        for r in csv.DictReader(open(glob.glob(f'{d}/{sub}/*counter_collection.csv')[0])):
            if r['Counter_Name'] != name:
                continue
            sn = short(r['Kernel_Name'])
            if sn:   # several instantiations can share a short name: average over their launches
                acc.setdefault(sn, []).append(float(r['Counter_Value']))
        for sn, vals in acc.items():
            # (round 6: bench.py also runs a few steps of 32 pictures -- `value_at_32_pictures_per_step`; a launch that moved
            # less than half of the name's largest launch is one of those, or of a smaller instantiation, and stays out of
            # the per-launch average of the full batch.  Names that only occur in those steps keep what they have.)
            full = [v for v in vals if v >= 0.5 * max(vals)]
            out.setdefault(sn, {})[name + '_KiB'] = sum(full) / len(full)
            out[sn]['launches_sampled'] = len(full)
    cfg = json.loads(open(f'{d}/bench_under_rocprof.json').read().strip().splitlines()[-1])["config"]   # the same command line as the PMC passes
    per_launch = cfg["pictures_per_gpu_per_step"] // max(1, cfg["streams"])
    res = {"_comment": "rocprofv3 PMC, two separate passes (--kernel-trace --pmc FETCH_SIZE ; --kernel-trace --pmc WRITE_SIZE) of "
                       f"`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-batch32` ({cfg['pictures_per_gpu_per_step']} UHD cfg-2 pictures per step on {cfg['streams']} stream(s): {per_launch} per launch), averages "
                       "per launch. FETCH_SIZE/WRITE_SIZE are in KiB. hbm_bytes_per_launch applies the gfx950 correction of "
                       "MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of wide coalesced reads): 2*FETCH + WRITE -- an upper bound for "
                       "kernels whose reads are narrower than 16 bytes per lane. Names with several launches per step (dwt_level, "
                       "idwt_level: the levels below the first) are averages over those launches. Made by tools/export_profiles.py "
                       "from the " + tag + " runs.",
           "csrc_digest": bench.csrc_digest(), "pictures_per_launch": per_launch, "kernels": {}}
    for k, v in sorted(out.items()):
        f, w = v.get('FETCH_SIZE_KiB', 0) * 1024, v.get('WRITE_SIZE_KiB', 0) * 1024
        res['kernels'][k] = {"launches_sampled": v['launches_sampled'], "FETCH_SIZE_KiB": round(v.get('FETCH_SIZE_KiB', 0), 1),
                             "WRITE_SIZE_KiB": round(v.get('WRITE_SIZE_KiB', 0), 1), "hbm_bytes_per_launch": int(2 * f + w)}
        print(f"{k:28s} fetch {f / 1e6:8.1f} MB  write {w / 1e6:8.1f} MB  hbm {(2 * f + w) / 1e6:8.1f} MB")
    json.dump(res, open(f'profiles/{tag}_pmc_traffic.json', 'w'), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
