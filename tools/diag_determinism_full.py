"""diagnostic: first non-reproducible module of the full-size SD1.5 UNet step"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
torch.manual_seed(0)
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 4, 64, 64, generator=g).half().cuda(); enc = torch.randn(2, 77, 768, generator=g).half().cuda()
t = torch.tensor([500.5, 500.5], device="cuda")
if len(sys.argv) > 1 and sys.argv[1] == "det":
    torch.backends.cudnn.deterministic = True
def run_record():
    rec = collections.OrderedDict(); hooks = []
    for name, m in unet.named_modules():
        if len(list(m.children())) == 0 or m.__class__.__name__ in ("Attention", "ResnetBlock2D", "Transformer2DModel"):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: rec.__setitem__(name, (out if torch.is_tensor(out) else out[0]).clone())))
    with torch.no_grad(): unet(x, t, enc)
    for h in hooks: h.remove()
    return rec
seen = collections.Counter()
for trial in range(6):
    r1, r2 = run_record(), run_record()
    bad = [n for n in r1 if not torch.equal(r1[n], r2[n])]
    if bad:
        m = dict(unet.named_modules())[bad[0]]
        print("trial", trial, "first mismatch:", bad[0], m.__class__.__name__, tuple(r1[bad[0]].shape), "n_bad", len(bad), flush=True)
        seen[bad[0]] += 1
    else:
        print("trial", trial, "all equal")
