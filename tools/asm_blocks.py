"""Per basic block of the generated gfx950 code: instruction mix of the blocks that hold MFMAs (developer tool).
    python tools/asm_blocks.py conv_wino43.hip [kernel-name-substring] [extra hipcc flags...]"""
import collections, os, re, subprocess, sys, tempfile
src = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ''
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, 'k.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '--cuda-device-only', '-S',
                           os.path.join(root, 'multipoint_amd', 'csrc', src), '-o', out] + sys.argv[3:])
    lines = open(out).read().split('\n')
func = None; block = None; stats = collections.OrderedDict()
for l in lines:
    t = l.strip()
    m = re.match(r'^(_Z\w+):', l)
    if m: func = m.group(1); block = 'entry'; continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: block = m.group(1); continue
    if not func or not t or t.startswith(('.', ';')): continue
    op = t.split()[0]
    stats.setdefault((func, block), collections.Counter())[op] += 1
for (f, b), c in stats.items():
    if sub not in f: continue
    nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
    if nm < 8: continue
    valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
    lane = sum(v for k, v in c.items() if k in ('v_readlane_b32', 'v_writelane_b32', 'v_readfirstlane_b32'))
    dsr = sum(v for k, v in c.items() if k.startswith('ds_read')); dsw = sum(v for k, v in c.items() if k.startswith('ds_write'))
    dma = c.get('global_load_lds_dwordx4', 0); gl = sum(v for k, v in c.items() if k.startswith('global_load') and 'lds' not in k)
    gs = sum(v for k, v in c.items() if k.startswith('global_store'))
    print('%s %s: mfma %d valu %d (lane ops %d, pk %d, mov %d, accvgpr %d) ds_read %d ds_write %d dma %d gload %d gstore %d salu %d waitcnt %d' % (
        f[-40:], b, nm, valu, lane, sum(v for k, v in c.items() if k.startswith('v_pk_')), c.get('v_mov_b32', 0) + c.get('v_mov_b64', 0),
        sum(v for k, v in c.items() if 'accvgpr' in k), dsr, dsw, dma, gl, gs, sum(v for k, v in c.items() if k.startswith('s_') and k not in ('s_waitcnt', 's_nop', 's_barrier')), c.get('s_waitcnt', 0)))
