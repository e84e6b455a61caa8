import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from recguru_amd import hip
import test_fused256_gpu as T
M, L, dff = 64, 7, 512
t = T._block_inputs(M, dff, L, False)
rm = torch.ones(M, device="cuda")
def run(t, tag):
    out, sv = hip.post_attn_fwd(t["ctx"], t["x"], T._pack(t["Wo"]), t["bo"], t["g1"], t["be1"], T._pack(t["W1"]), t["b1"], T._pack(t["W2"]), t["b2"],
                                t["g2"], t["be2"], rm, save=True, L=L, w_packed=True)
    ref = T._torch_block(t, rm, L, False)
    torch.cuda.synchronize()
    for k, r in (("y", ref["y1"]), ("h1", ref["h1"]), ("out", ref["out"])):
        v = (out if k == "out" else sv[k]).float()
        print(tag, k, "nan", int(torch.isnan(v).sum()), "of", v.numel(), "max err", float((v - r).abs().nan_to_num(1e9).max()), "| rows with nan", torch.isnan(v).any(1).nonzero().flatten()[:8].tolist(), "cols", torch.isnan(v).any(0).nonzero().flatten()[:8].tolist())
    print(tag, "rstd1 nan", int(torch.isnan(sv["rstd1"]).sum()), sv["rstd1"][:4].tolist())
run(t, "full")
t2 = dict(t); t2["ctx"] = torch.zeros_like(t["ctx"])
run(t2, "ctx=0")
t3 = dict(t); t3["Wo"] = torch.zeros_like(t["Wo"])
run(t3, "Wo=0")
t4 = dict(t3); t4["W1"] = torch.zeros_like(t["W1"]); t4["W2"] = torch.zeros_like(t["W2"])
run(t4, "all W=0")
