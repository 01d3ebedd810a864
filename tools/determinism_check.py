import sys, os
sys.path.insert(0, "learn-nerf_amd"); sys.path.insert(0, ".")
import torch, bench
from learn_nerf.rng import Key
dev = torch.device("cuda", 0)
outs = []
for rep in range(2):
    loop = bench.build_loop("nerf", "bf16", 19, dev)
    step = loop.step_fn(bench.BBOX_MIN, bench.BBOX_MAX)
    batch = bench.synthetic_batch(1024, 7, dev)
    for i in range(3):
        step(Key(i), batch)
    torch.cuda.synchronize()
    outs.append(loop.flat.clone())
print("params bit-identical after 3 steps:", torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max()))
