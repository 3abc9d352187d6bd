"""determinism_stress.py [iters] -- replays of the same forward must give the same bits.  The draft head's tree forward (attention
"block") and the base runner's verify forward (attention "split", fused q|k|v, pairs kernel) are run `iters` times on fixed inputs per
row bucket; any output that differs from the first run's is reported (an intermittent difference = a race inside a kernel or between
launches).  Prints one line per case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sam-decoding_amd"), os.path.join(ROOT, "tests")]
import torch
import samd_hip
from samd_hip.llama import LlamaRunner
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad_total = 0

def stress(name, fn, outs):
    global bad_total
    fn(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs()]
    bad = 0
    for _ in range(iters):
        fn()
        cur = outs()
        if any(not torch.equal(a.view(torch.uint8).reshape(-1) if False else a, b) for a, b in zip(cur, ref)):
            bad += 1
    torch.cuda.synchronize()
    bad_total += bad
    print(f"{name}: {bad} of {iters} runs differ from the first", flush=True)

# ---- draft head, tree forward ------------------------------------------------------------------------------------------------------
from test_gpu_llama import tiny_llama
from samd.tree_model.device_head import DeviceHead
from samd.tree_model.eagle2 import Eagle2Head
lm = tiny_llama(2, seed=8)
base = LlamaRunner.from_hf(lm, max_cache_len=512, dtype=torch.float16)
head = Eagle2Head(dict(hidden_size=256, intermediate_size=512, num_attention_heads=2, num_key_value_heads=2, vocab_size=512, rms_norm_eps=1e-5, bias=True),
                  dtype=torch.float16, device="cuda")
head.random_init(seed=3, std=0.08)
dh = DeviceHead(head, base)
g = torch.Generator(device="cuda").manual_seed(0)
hs = torch.randn((70, 256), generator=g, device="cuda").half(); ids = torch.randint(3, 512, (70,), generator=g, device="cuda")
dh.extend(hs[:64], ids[:64]); dh.extend(hs[64:], ids[64:])
for n in (8, 16, 24, 40, 64):
    x = torch.randn((n, 256), generator=g, device="cuda").half()
    par = torch.randint(0, 8, (n,), generator=g, device="cuda")
    anc = torch.eye(n, device="cuda")
    depth = torch.zeros(n, dtype=torch.int32, device="cuda")
    for i in range(8, n):
        anc[i] += anc[int(par[i]) + 8 * ((i // 8) - 1)]; depth[i] = i // 8
    anc = (anc > 0).float()
    res = {}
    def run(x=x, depth=depth, anc=anc, res=res):
        res["o"] = dh.tree(x, depth, anc)
    stress(f"draft head tree forward, {n} rows", run, lambda res=res: [res["o"][0], res["o"][1]])

# ---- base runner, verify forward at Vicuna-7B width (4 layers) ------------------------------------------------------------------------
cfg = dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=4, num_attention_heads=32, num_key_value_heads=32, vocab_size=32000,
           max_position_embeddings=2048, rms_norm_eps=1e-6)
runner = LlamaRunner.random_init(cfg, 2048, torch.float16, seed=1)
sess = samd_hip.Session(4096)
runner.prefill(sess, torch.randint(3, 32000, (1, 700), device="cuda"))
v = sess.device_views()
for R in (8, 16, 32, 48, 64):
    n = R - 3
    runner.pf_n.fill_(n)
    toks = torch.randint(3, 32000, (64,), dtype=torch.int32, device="cuda")
    gr = torch.cuda.CUDAGraph()
    fwd = lambda R=R, toks=toks: runner.forward_rows(R, toks, runner.pf_relpos, runner.pf_mask, v["cache_length"], runner.pf_n)
    fwd(); torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        fwd()
    b = runner._buffers(R)
    stress(f"verify forward (hipGraph), {R}-row bucket", gr.replay, lambda b=b, n=n: [b["logits"][:n], b["argmax"][:n]])
print("TOTAL differing runs:", bad_total)
