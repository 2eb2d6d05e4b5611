"""long prompts (S = 154 / 231 text keys) at the SD1.5 bench shape: chunked prepared-operand kernels vs the library op sequence"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from diffusionspatialcontrol_amd.modules import attention_modify as am
from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
import bench
torch.manual_seed(0)
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
for S in (77, 154, 231):
    emb, ids, state, tok = bench.synthetic_inputs(512, 2, S=S)
    pipe = StableDiffusionPipeline(None, None, tok, unet, SD15Scheduler())
    for mode, lim in (("chunked kernels", 384), ("library sequence", 0)):
        if S == 77 and lim == 0:
            continue
        am._KERNEL_MAX_KEYS_PACKED = lim
        kw = dict(height=512, width=512, num_inference_steps=25, guidance_scale=7.5, output_type="latent", region_map_state=state,
                  sampler_name="sample_dpmpp_2m", sampler_opt={"scheduler": "karras"}, prompt_embeds=emb[1:2].cuda().half(),
                  negative_prompt_embeds=emb[:1].cuda().half(), text_input_ids=ids)
        for _ in range(2):
            out = pipe.txt2img(None, latents=torch.randn(1, 4, 64, 64).half(), **kw)[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            out = pipe.txt2img(None, latents=torch.randn(1, 4, 64, 64).half(), **kw)[0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"S={S:3d} {mode:17s}: {dt*1e3:7.1f} ms / image ({1/dt:5.2f} images/s) finite={bool(torch.isfinite(out).all())}", flush=True)
        pipe._graphs = {}
