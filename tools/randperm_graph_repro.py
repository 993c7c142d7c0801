"""Minimal reproducer behind device/ppo.py's "permutations are drawn outside the captured graph": a HIP graph whose only content is
`torch.randperm(n, device="cuda")` (+ a copy into a slot of a result buffer), replayed K times back to back without a host wait, against the same with
one hipStreamSynchronize per replay.  Prints how many of the K results are valid permutations in each mode (ROCm 7.2 / torch 2.10 on MI355X)."""
import sys

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
out = torch.zeros((K, n), dtype=torch.int64, device=dev)
slot = torch.zeros(1, dtype=torch.int64, device=dev)
torch.manual_seed(0)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        p = torch.randperm(n, device=dev)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    p = torch.randperm(n, device=dev)
    out.view(-1).index_copy_(0, slot * n + torch.arange(n, device=dev), p)  # result k lands in row `slot`
    slot.add_(1)
want = torch.arange(n, device=dev)
for mode in ("stream sync per replay", "no host wait"):
    out.zero_()
    slot.zero_()
    torch.cuda.synchronize()
    for k in range(K):
        g.replay()
        if mode.startswith("stream"):
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    valid = sum(int(torch.equal(out[k].sort().values, want)) for k in range(K))
    distinct = len({int(out[k, :8].sum()) * 1000003 + int(out[k, 8:16].sum()) for k in range(K)})
    print(f"{mode:24s}: {valid} of {K} replays gave a valid permutation of {n}; {distinct} distinct results", flush=True)
