"""Per-level cost of the hash-grid gather and scatter at BASELINE configs[2] size (786,432 evaluations, T = 2^19):
one single-level encoding per grid size, timed alone.  Feeds the cost weights of the XCD-aware level assignment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learn-nerf_amd")); sys.path.insert(0, ROOT)
import torch
from learn_nerf import ops
from learn_nerf.instant_ngp import MultiresHashTableEncoding

m = 786432
gen = torch.Generator().manual_seed(0)
# samples along rays (neighbouring samples are close in space, as in the train step)
o = torch.rand(4096, 1, 3, generator=gen) * 0.2 - 0.1
d = torch.randn(4096, 1, 3, generator=gen); d = d / d.norm(dim=-1, keepdim=True)
t = torch.linspace(-0.9, 0.9, 192).reshape(1, 192, 1)
x = (o + d * t).clamp(-1, 1).reshape(-1, 3).contiguous().cuda()
for G in (16, 32, 64, 128, 256, 512, 1024, 2048):
    enc = MultiresHashTableEncoding([2 ** 19], [G], (-1.0,) * 3, (1.0,) * 3)
    tables = (torch.rand(enc.num_table_floats(), generator=gen) * 2 - 1).cuda()
    g = torch.randn(2, m, generator=gen).cuda()
    gt = torch.zeros_like(tables)
    res = []
    for fn in (lambda: enc.encode_t(tables, x), lambda: ops.hashgrid_bwd(enc.desc(), x, g, gt)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"G={G:5d} hashed={int(G ** 3 > 2 ** 19)}: gather {res[0]:7.1f} us  scatter {res[1]:7.1f} us")
