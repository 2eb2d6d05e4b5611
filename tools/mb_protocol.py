"""protocol-mode samplers at the SD1.5 shape: eager UNet calls vs graph-backed model calls (DSC_PROTOCOL_GRAPH=0/1)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from diffusionspatialcontrol_amd import ops
from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
import bench
torch.manual_seed(0)
with torch.device("cuda"):
    unet = UNet2DConditionModel(UNetConfig.sd15())
unet = unet.half().eval()
emb, ids, state, tok = bench.synthetic_inputs(512, 2)
pipe = StableDiffusionPipeline(None, None, tok, unet, SD15Scheduler())
for name in ("sample_dpmpp_2m", "sample_euler", "sample_heun"):
    for mode in ("fused", "protocol+graph", "protocol eager"):
        if mode == "fused" and name != "sample_dpmpp_2m":
            continue
        ops.PROTOCOL_GRAPH = mode == "protocol+graph"
        kw = dict(height=512, width=512, num_inference_steps=25, guidance_scale=7.5, output_type="latent", region_map_state=state,
                  sampler_name=name, sampler_opt={"scheduler": "karras"}, prompt_embeds=emb[1:2].cuda().half(),
                  negative_prompt_embeds=emb[:1].cuda().half(), text_input_ids=ids, fused=(mode == "fused"))
        for _ in range(2):
            out = pipe.txt2img(None, latents=torch.randn(1, 4, 64, 64).half(), **kw)[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            out = pipe.txt2img(None, latents=torch.randn(1, 4, 64, 64).half(), **kw)[0]
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"{name:18s} {mode:15s}: {dt*1e3:7.1f} ms / image ({1/dt:5.2f} images/s)", flush=True)
# ControlNet (SD1.5-size, random weights) inside the captured step vs evaluated eagerly per model call
from diffusionspatialcontrol_amd.modules.controlnet import ControlNetModel
with torch.device("cuda"):
    cn = ControlNetModel(UNetConfig.sd15())
for conv in list(cn.controlnet_down_blocks) + [cn.controlnet_mid_block, cn.controlnet_cond_embedding.conv_out]:
    torch.nn.init.normal_(conv.weight, 0.0, 0.02)
pipe.setup_controlnet(cn.half().eval())
ctrl = torch.rand(1, 3, 512, 512)
for mode in ("protocol+graph", "protocol eager"):
    ops.PROTOCOL_GRAPH = mode == "protocol+graph"
    kw = dict(height=512, width=512, num_inference_steps=25, guidance_scale=7.5, output_type="latent", region_map_state=state,
              sampler_name="sample_dpmpp_2m", sampler_opt={"scheduler": "karras"}, prompt_embeds=emb[1:2].cuda().half(),
              negative_prompt_embeds=emb[:1].cuda().half(), text_input_ids=ids, control_img=ctrl, controlnet_conditioning_scale=1.0)
    for _ in range(2):
        out = pipe.txt2img(None, latents=torch.randn(1, 4, 64, 64).half(), **kw)[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        out = pipe.txt2img(None, latents=torch.randn(1, 4, 64, 64).half(), **kw)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{'dpmpp_2m + ControlNet':18s} {mode:15s}: {dt*1e3:7.1f} ms / image ({1/dt:5.2f} images/s)", flush=True)
