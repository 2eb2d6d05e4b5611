"""diagnostic: which op of the UNet step is not bit-reproducible run to run?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from diffusionspatialcontrol_amd import ops
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig

torch.manual_seed(0)
dev = "cuda"
def rep(name, fn, n=4):
    outs = [fn().clone() for _ in range(n)]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{name:40s} reproducible={same}" + ("" if same else f"  maxdiff={max((outs[0].float()-o.float()).abs().max().item() for o in outs[1:]):.3e}"), flush=True)

x = torch.randn(2, 64, 16, 16, device=dev).half()
w = torch.randn(64, 64, 3, 3, device=dev).half()
rep("conv3x3 64->64 16x16", lambda: F.conv2d(x, w, padding=1))
x2 = torch.randn(2, 32, 16, 16, device=dev).half(); w2 = torch.randn(64, 32, 3, 3, device=dev).half()
rep("conv3x3 32->64", lambda: F.conv2d(x2, w2, padding=1))
rep("conv3x3 stride2", lambda: F.conv2d(x, w, padding=1, stride=2))
w1 = torch.randn(64, 64, 1, 1, device=dev).half()
rep("conv1x1", lambda: F.conv2d(x, w1))
a = torch.randn(2, 256, 64, device=dev).half(); wl = torch.randn(512, 64, device=dev).half()
rep("linear", lambda: F.linear(a, wl))
q = torch.randn(2, 256, 4, 16, device=dev).half()
rep("self_attention (interim SDPA)", lambda: ops.self_attention(q, q, q))
rep("layer_norm", lambda: F.layer_norm(a, (64,)))
g = torch.ones(64, device=dev).half(); b = torch.zeros(64, device=dev).half()
rep("groupnorm_silu", lambda: ops.groupnorm_silu(x, 8, g, b, 1e-5, True))
rep("geglu", lambda: ops.geglu(F.linear(a, wl)))
k = torch.randn(2, 77, 4, 16, device=dev).half()
wt = torch.zeros(2, 256, 77, device=dev); wt[:, :100, 2:4] = 0.5
rep("region_xattn", lambda: ops.region_xattn(q, k, k, wt, 3.0, layout="blhd"))
rep("interpolate nearest", lambda: F.interpolate(x, scale_factor=2.0, mode="nearest"))
cfg = UNetConfig.tiny()
unet = UNet2DConditionModel(cfg).half().to(dev)
xi = torch.randn(2, 4, 16, 16, device=dev).half(); t = torch.tensor([500.0, 500.0], device=dev)
enc = torch.randn(2, 77, 64, device=dev).half()
rep("tiny unet forward (eager)", lambda: unet(xi, t, enc).sample, n=5)
rs = {256: torch.zeros(2, 256, 77), 64: torch.zeros(2, 64, 77), 16: torch.zeros(2, 16, 77), 4: torch.zeros(2, 4, 77)}
for v in rs.values(): v[:, : v.shape[1] // 2, 2:4] = 0.5
sig = torch.tensor([4.0], device=dev)
rp = {"region_state": rs, "sigma": sig, "weight_func": lambda w, s, qk: w * s * qk.std()}
rep("tiny unet forward + region (eager)", lambda: unet(xi, t, enc, cross_attention_kwargs={"region_prompt": rp}).sample, n=5)

# ---- locate the first non-reproducible module
import collections
def run_record():
    rec = collections.OrderedDict()
    hooks = []
    for name, m in unet.named_modules():
        if len(list(m.children())) == 0 or m.__class__.__name__ == "Attention":
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: rec.__setitem__(name, (out if torch.is_tensor(out) else out[0]).clone())))
    unet(xi, t, enc, cross_attention_kwargs={"region_prompt": rp})
    for h in hooks: h.remove()
    return rec
bad = 0
for trial in range(6):
    r1, r2 = run_record(), run_record()
    for name in r1:
        if not torch.equal(r1[name], r2[name]):
            m = dict(unet.named_modules())[name]
            print("trial", trial, "first mismatch at", name, m.__class__.__name__, tuple(r1[name].shape), (r1[name].float()-r2[name].float()).abs().max().item(), flush=True)
            bad += 1
            break
print("mismatching trials:", bad)
torch.backends.cudnn.deterministic = True
rep("tiny unet + region, cudnn.deterministic=True", lambda: unet(xi, t, enc, cross_attention_kwargs={"region_prompt": rp}).sample, n=8)
torch.backends.cudnn.deterministic = False
xs = torch.randn(2, 128, 2, 2, device=dev).half(); ws = torch.randn(64, 128, 1, 1, device=dev).half()
rep("conv1x1 128->64 @2x2 (MIOpen)", lambda: F.conv2d(xs, ws), n=10)
rep("conv1x1 as baddbmm", lambda: torch.matmul(ws.view(64, 128), xs.view(2, 128, 4)), n=10)

