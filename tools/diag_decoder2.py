import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.test_gpu_parity import _synthetic_decoder
from tests.common import T
from aqualora_amd import _lib as L
from oracle import decoder_oracle as DO
dec = _synthetic_decoder(48); sd = {k[6:]: v.clone().float() for k, v in dec.state_dict().items()}
dec = dec.to("cuda").eval(); P = dec._pack(); st = L.stream_ptr()
B=2; x = T("dbg.x", (B,3,512,512), 0.5).clamp(-1,1); dev="cuda"
def rel(a,b): return [round(((a[i].cpu()-b[i]).abs().max()/(b[i].abs().max()+1e-12)).item(),8) for i in range(B)]
xr = F.interpolate(x, size=(512,512), mode="bilinear"); r = DO._cna(sd, "features.0", xr, 2, 3)
cur = torch.empty(B,512,512,3,device=dev); L.call("aql_resize_bilinear_nhwc", L.ptr(x.cuda()), B,3,512,512,512,512,L.ptr(cur),st)
h = torch.empty(B,256,256,32,device=dev); L.call("aql_stem_conv3x3s2_silu", L.ptr(cur), L.ptr(P["stem"][0]), L.ptr(P["stem"][1]), B,512,512,32,L.ptr(h),st)
Hc=256; bi=0
for si,(t,k,s,cin,cout,n) in enumerate(DO.B1_STAGES, start=1):
    for i in range(n):
        p=f"features.{si}.{i}.block"; stride = s if i==0 else 1; c_in = cin if i==0 else cout
        inp_r = r; j=0
        if t!=1: r = DO._cna(sd, f"{p}.0", r, 1, 1); j=1
        r = DO._cna(sd, f"{p}.{j}", r, stride, k, groups=r.shape[1])
        g = r.mean(dim=(2,3),keepdim=True); g = F.silu(F.conv2d(g, sd[f"{p}.{j+1}.fc1.weight"], sd[f"{p}.{j+1}.fc1.bias"])); g = torch.sigmoid(F.conv2d(g, sd[f"{p}.{j+1}.fc2.weight"], sd[f"{p}.{j+1}.fc2.bias"]))
        r = DO._cna(sd, f"{p}.{j+2}", r*g, 1, 1, act=False)
        if stride==1 and c_in==cout: r = r + inp_r
        d = P["blocks"][bi]; bi+=1
        tt,kk,ss,ci,co = d["cfg"]; cexp = ci*tt; inp = h
        if tt!=1:
            e = torch.empty(B,Hc,Hc,cexp,device=dev); L.call("aql_pwconv_f32", L.ptr(h), L.ptr(d["exp"][0]), L.ptr(d["exp"][1]), None, 0, None, B*Hc*Hc, cexp, ci, 1, L.ptr(e), st); h=e
        Ho=(Hc+2*(kk//2)-kk)//ss+1
        dw = torch.empty(B,Ho,Ho,cexp,device=dev); L.call("aql_dwconv_silu", L.ptr(h), L.ptr(d["dw"][0]), L.ptr(d["dw"][1]), B,Hc,Hc,cexp,kk,ss,L.ptr(dw),st); Hc=Ho
        pool = torch.empty(B,cexp,device=dev); L.call("aql_avgpool_nhwc", L.ptr(dw), B, Hc*Hc, cexp, L.ptr(pool), st)
        gate = torch.empty(B,cexp,device=dev); w1,b1,w2,b2=d["se"]; L.call("aql_se_gate", L.ptr(pool), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), B, cexp, w1.shape[0], L.ptr(gate), st)
        out = torch.empty(B,Hc,Hc,co,device=dev); L.call("aql_pwconv_f32", L.ptr(dw), L.ptr(d["proj"][0]), L.ptr(d["proj"][1]), L.ptr(gate), Hc*Hc, L.ptr(inp) if d["res"] else None, B*Hc*Hc, co, cexp, 0, L.ptr(out), st); h=out
        print(si,i,"cfg",d["cfg"],"res",d["res"], "rel", rel(h.permute(0,3,1,2), r), "max", r.abs().max().item(), "gate", rel(gate, g.flatten(1)))
