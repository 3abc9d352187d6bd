import torch, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
x = torch.randn((64, 128256), device="cuda")
w = torch.randn((4096, 4096), device="cuda", dtype=torch.bfloat16)
def work(R):
    y = torch.log_softmax(x[:R].float(), dim=-1)
    t = torch.topk(y, 8, dim=-1)
    cu = t.values + t.values[:, :1]
    b = torch.topk(cu.view(-1), 8)
    k = torch.sort(b.indices).values
    s = torch.searchsorted(k, b.indices)
    h = (w[:R] * 2).to(torch.float32).sum(-1)
    return t.indices.reshape(-1)[b.indices] + s, h
graphs = []
for R in (64, 8, 16, 1):
    if mode != "nowarm":
        work(R); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = work(R)
    graphs.append((g, out))
    g.replay(); torch.cuda.synchronize(); print("captured+replayed", R, flush=True)
for i, (g, out) in enumerate(graphs):
    g.replay(); torch.cuda.synchronize(); print("replay", i, "ok", out[0][:3].tolist(), flush=True)
