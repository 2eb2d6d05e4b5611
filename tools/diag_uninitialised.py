"""Latent-initialisation check: fill free device memory with NaN bit patterns before the pipeline creates its buffers /
captures its graph.  A buffer that is read before it is (fully) written shows up as a NaN result instead of hiding behind
the zeros of freshly mapped memory.  mode: cache (poison torch's cached blocks) | driver (also return them to the driver)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
mode = sys.argv[1] if len(sys.argv) > 1 else "cache"
torch.manual_seed(3)
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
emb, ids, state, tok = bench.synthetic_inputs(512, 2)
emb = emb.half().cuda()
pipe = StableDiffusionPipeline(None, None, tok, unet, SD15Scheduler())
lat = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(1)).half().cuda()
kw = dict(height=512, width=512, num_inference_steps=4, guidance_scale=7.5, latents=lat, output_type="latent",
          region_map_state=state, sampler_name="sample_dpmpp_2m", sampler_opt={"scheduler": "karras"},
          prompt_embeds=emb[1:2], negative_prompt_embeds=emb[0:1], text_input_ids=ids)
if mode != "none":
    sizes = [1 << 14, 1 << 17, 1 << 20, 1 << 23, 1 << 26, 1 << 28]
    junk = [torch.full((n,), float("nan"), dtype=torch.float32, device="cuda") for n in sizes for _ in range(24)]
    torch.cuda.synchronize()
    del junk
    if mode == "driver":
        torch.cuda.empty_cache()
out = pipe.txt2img(None, **kw)[0]
print(mode, "first generation finite:", torch.isfinite(out).all().item(), flush=True)
out2 = pipe.txt2img(None, **kw)[0]
print(mode, "second generation finite:", torch.isfinite(out2).all().item(), "equal:", torch.equal(out, out2))
