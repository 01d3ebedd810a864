"""Times lnrf_hashgrid_fwd at BASELINE configs[2] size (786,432 evaluations, L = 16, T = 2^19) and at the coarse size
(262,144, L = 6).  LNRF_HASHGRID_LDS=0 disables the LDS-staged path of the 16^3 levels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf.instant_ngp import MultiresHashTableEncoding

for levels, m in ((16, 786432), (6, 262144)):
    enc = MultiresHashTableEncoding([2 ** 19] * levels, [2 ** (4 + i // 2) for i in range(levels)], (-1.0,) * 3, (1.0,) * 3)
    gen = torch.Generator().manual_seed(0)
    tables = (torch.rand(enc.num_table_floats(), generator=gen) * 2 - 1).cuda()
    x = (torch.rand(m, 3, generator=gen) * 2 - 1).cuda()
    for _ in range(3):
        enc.encode_t(tables, x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        enc.encode_t(tables, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"L={levels} m={m} LDS={os.environ.get('LNRF_HASHGRID_LDS', '1')}: {ms:.4f} ms = {m * levels * 64 / ms / 1e6:.0f} GB/s algorithmic")
