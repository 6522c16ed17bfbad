"""GPU (MI355X): the C ABI used from a plain C host program (examples/c_host_demo.c: gcc, HIP runtime for device
memory, no Python / PyTorch in the process) -- the drop-in boundary for a non-Python caller.  The program runs the
whole pair path (mp_forward -> mp_detect_keypoints -> mp_sample_descriptors -> mp_match_mutual_nn); its results must
equal the Python host path's bit for bit (same kernels) and the oracle's within the stated tolerances."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_program(tmp_path, oracle):
    exe = str(tmp_path / 'c_host_demo')
    build = subprocess.run(['gcc', '-O2', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include',
                            '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_host_demo.c'),
                            '-L' + os.path.join(ROOT, 'multipoint_amd'), '-lmultipoint_hip', '-L/opt/rocm/lib', '-lamdhip64',
                            '-Wl,-rpath,' + os.path.join(ROOT, 'multipoint_amd'), '-Wl,-rpath,/opt/rocm/lib', '-o', exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    cfg = dict(oracle.SHIPPED_MODEL_CONFIG)
    sd = oracle.make_weights(0, cfg)
    with open(tmp_path / 'weights.bin', 'wb') as f:
        items = [(k, v) for k, v in sd.items()]
        f.write(struct.pack('<i', len(items)))
        for k, v in items:
            a = v.detach().to(torch.float32).contiguous().numpy()
            f.write(struct.pack('<i', len(k))); f.write(k.encode()); f.write(struct.pack('<q', a.size)); f.write(a.tobytes())
    B, H, W, K = 4, 120, 160, 200
    img = oracle.make_images(7, B, H, W)
    with open(tmp_path / 'images.bin', 'wb') as f:
        f.write(struct.pack('<iii', B, H, W)); f.write(img.numpy().tobytes())
    run = subprocess.run([exe, str(tmp_path / 'weights.bin'), str(tmp_path / 'images.bin'), str(tmp_path / 'out.bin'), str(K)],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert 'multipoint_hip' in run.stdout and 'mutual matches' in run.stdout and 'error path:' in run.stdout
    raw = open(tmp_path / 'out.bin', 'rb').read()
    hdr = struct.unpack('<5i', raw[:20]); assert hdr == (B, H, W, 64, K)
    off = 20

    def take(n, dt):
        nonlocal off
        a = np.frombuffer(raw, dtype=dt, count=n, offset=off); off += a.nbytes
        return a
    prob = take(B * H * W, np.float32).reshape(B, 1, H, W)
    desc = take(B * (H // 8) * (W // 8) * 64, np.float32).reshape(B, H // 8, W // 8, 64)
    cnt = take(B, np.int32)
    kp = take(B * K * 2, np.int32).reshape(B, K, 2)
    kpdesc = take(B * K * 64, np.float32).reshape(B, K, 64)
    midx = take(B // 2 * K, np.int32).reshape(B // 2, K)
    mcnt = take(B // 2, np.int32)
    assert off == len(raw)

    # the Python host path on the same inputs: identical bits
    import multipoint_amd.models as models
    from multipoint_amd.pipeline import PairPipeline
    net = models.MultiPoint(cfg); net.load_state_dict(sd); net.to('cuda:0'); net.eval()
    out = net({'image': img.to('cuda:0')})
    assert np.array_equal(out['prob'].cpu().numpy(), prob)
    assert np.array_equal(out['desc'].permute(0, 2, 3, 1).contiguous().cpu().numpy(), desc)
    pred = {'nms': 4, 'detection_threshold': 0.015, 'topk': K,
            'matching': {'method': 'bfmatcher', 'method_kwargs': {'crossCheck': True}, 'knn_matches': False}}
    res = PairPipeline(net, pred, capacity=K)(img[0::2].to('cuda:0'), img[1::2].to('cuda:0'))
    assert np.array_equal(res.kp_count.cpu().numpy(), cnt) and np.array_equal(res.match_count.cpu().numpy(), mcnt)
    for b in range(B):
        n = min(int(cnt[b]), K)
        assert n > 20
        assert np.array_equal(res.kp_yx[b, :n].cpu().numpy(), kp[b, :n])
        assert np.array_equal(res.desc[b, :n].cpu().numpy(), kpdesc[b, :n])
    for p in range(B // 2):
        n = min(int(cnt[2 * p]), K)
        assert np.array_equal(res.match_idx[p, :n].cpu().numpy(), midx[p, :n]) and int(mcnt[p]) == int((midx[p, :n] >= 0).sum())
    # and the oracle within the stated tolerances
    ro = oracle.forward(sd, img, cfg)
    assert np.abs(ro['prob'].numpy() - prob).max() <= 1e-4
    assert np.abs(ro['desc'].permute(0, 2, 3, 1).numpy() - desc).max() <= 1e-4
