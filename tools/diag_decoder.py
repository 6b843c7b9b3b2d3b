import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.test_gpu_parity import _synthetic_decoder
from tests.common import T
from aqualora_amd import _lib as L
from oracle import decoder_oracle as DO
dec = _synthetic_decoder(48); sd = {k[6:]: v.clone().float() for k, v in dec.state_dict().items()}
dec = dec.to("cuda").eval(); P = dec._pack(); st = L.stream_ptr()
B=2; x = T("dbg.x", (B,3,512,512), 0.5).clamp(-1,1)
def rel(a,b): return ((a.cpu()-b).abs().max()/(b.abs().max()+1e-12)).item()
xr = F.interpolate(x, size=(512,512), mode="bilinear")
ref = DO._cna(sd, "features.0", xr, 2, 3)
cur = torch.empty(B,512,512,3,device="cuda"); L.call("aql_resize_bilinear_nhwc", L.ptr(x.cuda()), B,3,512,512,512,512,L.ptr(cur),st)
print("resize", rel(cur.permute(0,3,1,2), xr))
h = torch.empty(B,256,256,32,device="cuda"); L.call("aql_stem_conv3x3s2_silu", L.ptr(cur), L.ptr(P["stem"][0]), L.ptr(P["stem"][1]), B,512,512,32,L.ptr(h),st)
print("stem", rel(h.permute(0,3,1,2), ref), ref.abs().max().item())
# first block pieces
d = P["blocks"][0]; p="features.1.0.block"
dwr = DO._cna(sd, p+".0", ref, 1, 3, groups=32)
dw = torch.empty(B,256,256,32,device="cuda"); L.call("aql_dwconv_silu", L.ptr(h), L.ptr(d["dw"][0]), L.ptr(d["dw"][1]), B,256,256,32,3,1,L.ptr(dw),st)
print("dw", rel(dw.permute(0,3,1,2), dwr), dwr.abs().max().item())
pool = torch.empty(B,32,device="cuda"); L.call("aql_avgpool_nhwc", L.ptr(dw), B, 65536, 32, L.ptr(pool), st)
print("pool", rel(pool, dwr.mean(dim=(2,3))))
g = dwr.mean(dim=(2,3),keepdim=True); g = F.silu(F.conv2d(g, sd[p+".1.fc1.weight"], sd[p+".1.fc1.bias"])); g = torch.sigmoid(F.conv2d(g, sd[p+".1.fc2.weight"], sd[p+".1.fc2.bias"]))
gate = torch.empty(B,32,device="cuda"); w1,b1,w2,b2 = d["se"]
L.call("aql_se_gate", L.ptr(pool), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), B, 32, w1.shape[0], L.ptr(gate), st)
print("gate", rel(gate, g.flatten(1)))
pr = DO._cna(sd, p+".2", dwr*g, 1, 1, act=False)
out = torch.empty(B,256,256,16,device="cuda")
L.call("aql_pwconv_f32", L.ptr(dw), L.ptr(d["proj"][0]), L.ptr(d["proj"][1]), L.ptr(gate), 65536, None, B*65536, 16, 32, 0, L.ptr(out), st)
print("proj", rel(out.permute(0,3,1,2), pr), pr.abs().max().item())
