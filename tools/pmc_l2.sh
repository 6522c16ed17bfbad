#!/bin/bash
# Developer tool (GPU box): the L2 -> LDS weight stream of the fp32 convolutions -- vector-L1 (TCP) and L2 (TCC) request / hit counters per
# launch, one rocprofv3 --pmc pass per counter group (kernel trace only).   bash tools/pmc_l2.sh gpurun_out/dir [bench_layers args...]
export TMPDIR=/tmp
OUT=$1; shift
ARGS=${@:-64}
mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for G in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d $OUT -o l2p$i -- python3 tools/bench_layers.py $ARGS > $OUT/l2p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, collections, sys, os, glob
out = sys.argv[1]
for path in sorted(glob.glob(os.path.join(out, 'l2p*_counter_collection.csv'))):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].replace('void (anonymous namespace)::', '').split('(')[0], 'grid': int(r['Grid_Size'])})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    agg = collections.OrderedDict()
    for d in disp.values():
        if 'conv' not in d['name'] and 'vprod' not in d['name'] and 'head' not in d['name']: continue
        agg.setdefault((d['name'], d['grid']), []).append(d)
    print('==', os.path.basename(path))
    for (name, grid), ds in agg.items():
        ds = ds[len(ds) // 2:]                       # the later launches (warm)
        keys = [k for k in ds[0] if k not in ('name', 'grid')]
        m = {k: sum(x.get(k, 0.0) for x in ds) / len(ds) for k in keys}
        print('%-50s grid %7d n=%d ' % (name[:50], grid, len(ds)) + ' '.join('%s %.4g' % (k, m[k]) for k in keys))
PY
