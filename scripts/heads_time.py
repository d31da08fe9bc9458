"""Forward + backward time of each loss-head stand-in of `bench.py --config trainer` (B=16, 256x256): python scripts/heads_time.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import loss_heads as LH
torch.manual_seed(0)
B = int(os.environ.get('B', 16))
x = (torch.rand(B, 3, 256, 256, device='cuda') * 2 - 1).requires_grad_(True)
y = (torch.rand(B, 3, 256, 256, device='cuda') * 2 - 1)
heads = {'id': LH.IdLoss().cuda().eval(), 'lpips': LH.LpipsShaped().cuda().eval(), 'deca': LH.ShapeModelStandIn().cuda().eval()}
for m in heads.values():
    for p in m.parameters(): p.requires_grad_(False)
def run(name):
    m = heads[name]
    if name == 'deca':
        p, _ = m(x); q, _ = m(y)
        loss = m.landmark_loss({k: v.detach() for k, v in q.items()}, p)
    else:
        loss = m(x, y)
    loss.backward()
for name in heads:
    for _ in range(3): run(name)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(name)
    torch.cuda.synchronize()
    print('%-6s fwd(x)+fwd(y)+bwd: %.2f ms' % (name, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
