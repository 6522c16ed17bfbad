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
    with open(os.path.join(src, prefix + '_counter_collection.csv')) as f:
        for r in csv.DictReader(f):
            key = (short(r['Kernel_Name']), int(r['Grid_Size']))
            agg.setdefault(key, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    return agg


shutil.copy(os.path.join(src, 'stats_kernel_stats.csv'), os.path.join(dst, tag + '_bench_kernel_stats.csv'))

# encoder conv1+conv2 (first block fused into the Winograd conv2) @480x640: one launch per step of this instantiation
DOMINANT = 'conv_wino_kernel<true, false, true>'

fetch, write = counters('fetch'), counters('write')
kernels = []
for key, c in fetch.items():
    f = c['FETCH_SIZE']
    w = write.get(key, {}).get('WRITE_SIZE', [0.0])
    fe = sum(f) / len(f) * 1024.0                           # rocprofv3 reports KB
    wr = sum(w) / len(w) * 1024.0
    kernels.append({'kernel': key[0], 'grid_threads': key[1], 'launches': len(f), 'fetch_size_bytes': fe,
                    'fetch_bytes_corrected': 2 * fe, 'write_size_bytes': wr})
out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only, '
               'tools/collect_profiles.sh) on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`; bytes per '
               'launch; fetch_bytes_corrected = 2 x FETCH_SIZE (gfx950 reports half of a wide 16 B/lane coalesced '
               'stream, MI355X_MICROARCH.md section HBM); WRITE_SIZE uncalibrated',
       'kernels': kernels, 'dominant_kernel': DOMINANT}
# the fused instantiation is launched once per step (conv4 / conv6 use conv_wino_kernel<true, false, false>)
NPER = 1
key = (DOMINANT, [k for k in fetch if k[0] == DOMINANT][0][1]) if any(k[0] == DOMINANT for k in fetch) else None
if key:
    f = fetch[key]['FETCH_SIZE']; w = write.get(key, {}).get('WRITE_SIZE', [])
    per = []
    for i in range(NPER):
        fi = f[i::NPER]; wi = w[i::NPER] or [0.0]
        per.append({'launch_position': i, 'fetch_size_bytes': sum(fi) / len(fi) * 1024.0,
                    'fetch_bytes_corrected': 2 * sum(fi) / len(fi) * 1024.0, 'write_size_bytes': sum(wi) / len(wi) * 1024.0})
    out['dominant_kernel_by_launch_position'] = per
    d = max(per, key=lambda r: r['fetch_size_bytes'])
    out['dominant_kernel_mean_traffic_bytes_per_launch'] = d['fetch_bytes_corrected'] + d['write_size_bytes']
json.dump(out, open(os.path.join(dst, tag + '_pmc_hbm_traffic.json'), 'w'), indent=1)
print('dominant traffic', out.get('dominant_kernel_mean_traffic_bytes_per_launch'))

mf = counters('mfma')
with open(os.path.join(dst, tag + '_pmc_mfma_busy.csv'), 'w') as f:
    f.write('# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace '
            '-- python3 tools/bench_layers.py 64 (mean over the profiled launches of each conv)\n')
    f.write('kernel,grid_threads,launches,gui_active_cycles,mfma_busy_cycles,mfma_busy_pct_of_simd_cycles\n')
    for key, c in mf.items():
        if 'conv' not in key[0]:
            continue
        n = len(c['GRBM_GUI_ACTIVE'])
        gui = sum(c['GRBM_GUI_ACTIVE']) / n
        busy = sum(c['SQ_VALU_MFMA_BUSY_CYCLES']) / n
        # SQ_VALU_MFMA_BUSY_CYCLES accumulates over the 1024 SIMDs (256 CUs x 4); GRBM_GUI_ACTIVE over 8 XCDs
        pct = 100.0 * busy / 1024.0 / (gui / 8.0) if gui else 0.0
        f.write('"%s",%d,%d,%.0f,%.0f,%.1f\n' % (key[0], key[1], n, gui, busy, pct))
print(open(os.path.join(dst, tag + '_pmc_mfma_busy.csv')).read())
