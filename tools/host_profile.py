"""Host-side profile of back-to-back generations (cProfile): where the Python thread spends its time - i.e. where it waits for
the device (synchronising copies) instead of preparing the next generation under the current one's 25 graph replays."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline  # noqa: E402
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig  # noqa: E402

torch.manual_seed(0)
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
emb, ids, state, tok = bench.synthetic_inputs(512, 2)
emb = emb.cuda()
pipe = StableDiffusionPipeline(None, None, tok, unet, SD15Scheduler())
lat = torch.randn(1, 4, 64, 64).half().cuda()


def generate():
    return pipe.txt2img(None, height=512, width=512, num_inference_steps=25, guidance_scale=7.5, latents=lat,
                        output_type="latent", region_map_state=state, sampler_name="sample_dpmpp_2m",
                        sampler_opt={"scheduler": "karras"}, prompt_embeds=emb[1:2], negative_prompt_embeds=emb[0:1],
                        text_input_ids=ids)[0]


for _ in range(3):
    generate()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t0 = time.perf_counter()
for _ in range(n):
    generate()
torch.cuda.synchronize()
print(f"{n} generations, {1e3 * (time.perf_counter() - t0) / n:.2f} ms each (un-profiled)")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    generate()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
