"""probe: does the HOST block in the fused loop?  Host time per iteration of (a) graph replay only, (b) replay + the eager sampler
kernel (the loop's form), (c) replay + an unrelated eager torch kernel - each 50 iterations without synchronising in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
import bench
from diffusionspatialcontrol_amd import ops
from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
emb, ids, state, tok = bench.synthetic_inputs(512, 2)
emb = emb.to(dev)
pipe = StableDiffusionPipeline(None, None, tok, unet, SD15Scheduler())
lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(1000)).half().to(dev)
def gen():
    return pipe.txt2img(None, height=512, width=512, num_inference_steps=25, guidance_scale=7.5, latents=lat, output_type="latent",
                        region_map_state=state, sampler_name="sample_dpmpp_2m", sampler_opt={"scheduler": "karras"},
                        prompt_embeds=emb[1:2], negative_prompt_embeds=emb[0:1], text_input_ids=ids)[0]
gen(); gen()
torch.cuda.synchronize()
st = next(iter(pipe._graphs.values()))
run = st["run"]
x = lat.clone(); old = torch.zeros_like(x)
a = torch.zeros(64, device=dev)
def variant(name, extra):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run()
        extra()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:34s}: host {1e3 * (t1 - t0) / 50:.3f} ms per iteration, device-bound {1e3 * (t2 - t0) / 50:.3f} ms", flush=True)
variant("replay only", lambda: None)
variant("replay + eager sampler kernel", lambda: ops.cfg_dpmpp2m_step(x, st["eps"], old, 5.0, 7.5, 0.5, 0.5, 0.0, 1.0, 500.0, 4.0, st["x_in"], st["t"], st["sigma"]))
variant("replay + eager torch add", lambda: a.add_(1.0))
variant("replay only (again)", lambda: None)
