#!/bin/bash
# Developer tool (GPU box): where the waves of the fp16 convolutions spend their cycles (SQ counters, two passes).
#   bash tools/pmc_f16.sh gpurun_out/dir [bench_layers args...]
export TMPDIR=/tmp
OUT=$1; shift
ARGS=${@:-16 1024 1280 f16}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o p1 -- python3 tools/bench_layers.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o p2 -- python3 tools/bench_layers.py $ARGS > $OUT/p2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, collections, sys, os
out = sys.argv[1]
for pas in ('p1', 'p2'):
    path = os.path.join(out, pas + '_counter_collection.csv')
    if not os.path.exists(path):
        print('missing', path); continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].replace('void (anonymous namespace)::', '').split('(')[0], 'grid': int(r['Grid_Size'])})
        d[r['Counter_Name']] = float(r['Counter_Value'])
    agg = collections.OrderedDict()
    for d in disp.values():
        if 'conv' not in d['name']: continue
        import math
        cls = int(round(2 * math.log2(max(d.get('GRBM_GUI_ACTIVE', 1.0), 1.0))))
        agg.setdefault((d['name'], d['grid'], cls), []).append(d)
    for (name, grid, cls), ds in agg.items():
        n = len(ds)
        keys = [k for k in ds[0] if k not in ('name', 'grid')]
        m = {k: sum(x.get(k, 0.0) for x in ds) / n for k in keys}
        gui = m.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        line = '%-44s grid %7d n=%d cycles %9.0f' % (name[:44], grid, n, gui)
        wc = m.get('SQ_WAVE_CYCLES')
        for k in keys:
            if k in ('GRBM_GUI_ACTIVE',): continue
            if wc and k.startswith('SQ_') and k not in ('SQ_WAVE_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F16'):
                line += ' %s %.1f%%wc' % (k[3:], 100 * m[k] / wc)
            elif k == 'SQ_VALU_MFMA_BUSY_CYCLES':
                line += ' MFMA_BUSY %.1f%%' % (100 * m[k] / 1024.0 / gui if gui else 0)
            else:
                line += ' %s %.3g' % (k[3:], m[k])
        print(line)
PY
