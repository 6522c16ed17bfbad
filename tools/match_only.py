"""Developer tool (GPU box): the matcher of one bench batch, a few calls (for rocprofv3 --kernel-trace --stats)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipoint_amd.utils as U
from multipoint_amd import _lib
P, K, D = (8, 2000, 64) if len(sys.argv) > 1 and sys.argv[1] == 'c5' else (32, 1000, 64)
g = torch.Generator().manual_seed(0)
desc = torch.nn.functional.normalize(torch.randn(2 * P, K, D, generator=g), dim=2).cuda()
cnt = torch.full((2 * P,), K, dtype=torch.int32, device='cuda')
midx = torch.empty((P, K), dtype=torch.int32, device='cuda'); mdist = torch.empty((P, K), device='cuda'); mcnt = torch.empty((P,), dtype=torch.int32, device='cuda')
h = _lib.get_handle(desc.device)
for _ in range(6):
    h.check(h.lib.mp_match_mutual_nn(h.ptr, _lib.ptr(desc), _lib.ptr(cnt), ctypes.c_void_p(desc.data_ptr() + K * D * 4),
                                     ctypes.c_void_p(cnt.data_ptr() + 4), 2 * K * D, 2, P, K, D, float('inf'),
                                     _lib.ptr(midx), _lib.ptr(mdist), _lib.ptr(mcnt), _lib.stream_ptr(desc.device)))
torch.cuda.synchronize()
print('matches', int(mcnt.sum()))
