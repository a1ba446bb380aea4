import torch, sys
sys.path.insert(0, '.')
from consistentid_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
def rnd(*s, seed=0, scale=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).half()
B, C1, Cout, H = 8, 320, 320, 64
HW = H * H; M = B * HW; K = 9 * C1
x1 = rnd(M, C1, seed=1); w = rnd(Cout, K, seed=3, scale=K ** -0.5); bias = rnd(Cout, seed=4); res = rnd(M, Cout, seed=5)
out = torch.empty(M, Cout, dtype=torch.float16, device=dev)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
ops.gemm(x1.to(dev), w.to(dev), out, M=M, N=Cout, c1=C1, bias=bias.to(dev), res=res.to(dev), ldr=Cout, ws=ws, gn_hw=HW, taps=9, Hi=H, Wi=H, Ho=H, Wo=H)
torch.cuda.synchronize()
st, rows = out._gn_stats
u = Cout // 32
o = out.double().cpu().reshape(M // rows, rows, 32, u)
want = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1)
d = (st.double().cpu() - want)
rel = d.abs() / (want.abs() + rows * u * 1e-3)
i = rel.argmax()
idx = torch.unravel_index(i, rel.shape)
print("rows", rows, "max rel", rel.max().item(), "at", [int(x) for x in idx], "got", st.cpu()[idx].item(), "want", want[idx].item())
print("S max abs err", d[..., 0].abs().max().item(), "Q max abs err", d[..., 1].abs().max().item(), "Q max rel", (d[...,1].abs()/want[...,1]).max().item())
