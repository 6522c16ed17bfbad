"""Summarise the rocprofv3 CSVs written by tools/collect_profiles.sh into the files committed under profiles/.

  python tools/summarize_profiles.py gpurun_out/prof r01

writes profiles/<tag>_bench_kernel_stats.csv (copy of the --stats table), profiles/<tag>_pmc_hbm_traffic.json
(per-kernel FETCH_SIZE / WRITE_SIZE bytes per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
wide coalesced reads on gfx950) and profiles/<tag>_pmc_mfma_busy.csv (MFMA-busy share per conv launch).
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, 'profiles')


def short(name):
    name = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return name.split('(')[0]


def counters(prefix):
    """{(kernel, grid): {counter: [values per dispatch]}} in first-seen order."""
    agg = collections.OrderedDict()
    path = os.path.join(src, prefix + '_counter_collection.csv')
    if not os.path.exists(path):
        return agg
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (short(r['Kernel_Name']), int(r['Grid_Size']))
            agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


def summarize(prefix, suffix, dominant, nper, bench_cmd, layers_cmd):
    """prefix: '' (headline workload) or 'c5_'; dominant: the instantiation bench.py's roofline block is about; nper: how
    many launches of that instantiation one forward makes (the largest one is the dominant launch)."""
    stats = os.path.join(src, prefix + 'stats_kernel_stats.csv')
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, tag + '_bench%s_kernel_stats.csv' % suffix))
    fetch, write = counters(prefix + 'fetch'), counters(prefix + 'write')
    kernels = []
    for key, c in fetch.items():
        f = c['FETCH_SIZE']
        w = write.get(key, {}).get('WRITE_SIZE', [0.0])
        fe = sum(f) / len(f) * 1024.0                           # rocprofv3 reports KB
        wr = sum(w) / len(w) * 1024.0
        kernels.append({'kernel': key[0], 'grid_threads': key[1], 'launches': len(f), 'fetch_size_bytes': fe,
                        'fetch_bytes_corrected': 2 * fe, 'write_size_bytes': wr})
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, '
                   'tools/collect_profiles.sh) on `%s`; bytes per launch; fetch_bytes_corrected = 2 x FETCH_SIZE (gfx950 '
                   'reports half of a wide 16 B/lane coalesced stream, MI355X_MICROARCH.md section HBM); WRITE_SIZE '
                   'uncalibrated' % bench_cmd,
           'kernels': kernels, 'dominant_kernel': dominant, 'dominant_kernel_launches_per_forward': nper}
    keys = [k for k in fetch if k[0] == dominant]
    if keys:
        key = keys[0]
        f = fetch[key]['FETCH_SIZE']; w = write.get(key, {}).get('WRITE_SIZE', [])
        per = []
        for i in range(nper):
            fi = f[i::nper]; wi = w[i::nper] or [0.0]
            per.append({'launch_position': i, 'fetch_size_bytes': sum(fi) / len(fi) * 1024.0,
                        'fetch_bytes_corrected': 2 * sum(fi) / len(fi) * 1024.0, 'write_size_bytes': sum(wi) / len(wi) * 1024.0})
        out['dominant_kernel_by_launch_position'] = per
        d = max(per, key=lambda r: r['fetch_size_bytes'] + r['write_size_bytes'])
        out['dominant_kernel_mean_traffic_bytes_per_launch'] = d['fetch_bytes_corrected'] + d['write_size_bytes']
    if kernels:
        json.dump(out, open(os.path.join(dst, tag + '_pmc_hbm_traffic%s.json' % suffix), 'w'), indent=1)
    print(prefix or 'c3', 'dominant traffic', out.get('dominant_kernel_mean_traffic_bytes_per_launch'))

    mf = counters(prefix + 'mfma')
    if not mf:
        return
    path = os.path.join(dst, tag + '_pmc_mfma_busy%s.csv' % suffix)
    with open(path, 'w') as f:
        f.write('# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace '
                '-- %s (mean over the profiled launches of each (kernel, launch size); mfma_busy = share of SIMD '
                'cycles in which the matrix pipe is busy = issued MFMA FLOPs / peak at the clock the launch ran at)\n' % layers_cmd)
        f.write('kernel,grid_threads,launches,gui_active_cycles,mfma_busy_cycles,mfma_busy_pct_of_simd_cycles\n')
        # per (kernel, duration class): the same instantiation serves several layers with one grid size, so group the
        # dispatches of a kernel by their position inside a forward pass (they repeat with the same period)
        rows = collections.OrderedDict()
        with open(os.path.join(src, prefix + 'mfma_counter_collection.csv')) as g:
            disp = collections.OrderedDict()
            for r in csv.DictReader(g):
                d = disp.setdefault(int(r['Dispatch_Id']), {'name': short(r['Kernel_Name']), 'grid': int(r['Grid_Size'])})
                d[r['Counter_Name']] = float(r['Counter_Value'])
        for d in disp.values():
            if 'conv' not in d['name'] or 'GRBM_GUI_ACTIVE' not in d:
                continue
            # duration class: round the active cycles to 2 significant digits of their log2 bucket
            cls = int(round(2 * __import__('math').log2(max(d['GRBM_GUI_ACTIVE'], 1.0))))
            rows.setdefault((d['name'], d['grid'], cls), []).append(d)
        for (name, grid, cls), ds in rows.items():
            n = len(ds)
            gui = sum(x['GRBM_GUI_ACTIVE'] for x in ds) / n
            busy = sum(x['SQ_VALU_MFMA_BUSY_CYCLES'] for x in ds) / n
            # SQ_VALU_MFMA_BUSY_CYCLES accumulates over the 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE over 8 XCDs
            pct = 100.0 * busy / 1024.0 / (gui / 8.0) if gui else 0.0
            f.write('"%s",%d,%d,%.0f,%.0f,%.1f\n' % (name, grid, n, gui, busy, pct))
    print(open(path).read())


# enc.conv1+2 (the first block fused into the pooled 64 -> 64 layer) is the only launch of this instantiation
summarize('', '', 'conv_wino43_kernel<true, false, 8, true, false, false>', 1,
          'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline', 'python3 tools/bench_layers.py 64')
# fp16 workload: enc.conv1 evaluated inside enc.conv2 on the LDS-resident-weights kernel, one launch of this instantiation per forward
summarize('c5_', '_c5', 'conv_f16_res_kernel<32, true, false, 2, true>', 1,
          'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload c5', 'python3 tools/bench_layers.py 16 1024 1280 f16')

# the bench lines of tools/collect_all.sh (run AFTER the PMC passes were summarised: bench.py reads the traffic figure from the
# committed profiles/<tag>_pmc_hbm_traffic*.json) -> profiles/<tag>_bench_*.json, and the SQ counter summary of the fp16 convolutions
for name, out in (('bench_n1', 'bench_n1'), ('bench_c5', 'bench_c5_f16_n1'), ('bench_rccl', 'bench_rccl_world1'), ('bench_gen2', 'bench_gen2_only_n1'), ('bench_direct', 'bench_direct_n1'),
                  ('bench_c5_stream', 'bench_c5_streaming_kernel_n1')):
    path = os.path.join(src, name + '.json')
    if os.path.exists(path):
        lines = [l for l in open(path) if l.startswith('{')]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, '%s_%s.json' % (tag, out)), 'w'), indent=1)
if os.path.exists(os.path.join(src, 'latency.txt')):
    shutil.copy(os.path.join(src, 'latency.txt'), os.path.join(dst, tag + '_latency.txt'))
for name, out in (('sq_c5.txt', '_pmc_sq_counters_c5.txt'), ('sq_c3.txt', '_pmc_sq_counters_c3.txt'), ('bench_240x320.txt', '_bench_240x320.txt'),
                  ('ab_layers.txt', '_ab_layers.txt')):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, tag + out))
