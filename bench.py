#!/usr/bin/env python3
"""bench.py - images/s of the spatially-controlled SD1.5 denoising path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A bench "step" is one pass of the hot path over one batch: ONE full generation (25 DPM++ 2M Karras denoising steps,
CFG 7.5, every cross-attention carrying the region bias) of `--images-per-gpu` 512x512 latents per GPU.  Workload =
BASELINE.json configs[1] ("SD1.5 512x512, 25-step DPM++2M Karras, 2 region masks, batch=1, 1xMI355X"); with N GPUs
every rank generates its own images (weak scaling, no data-path collective: one broadcast of the text embeddings
and region table before the timed region).  Synthetic data: random-init SD1.5-architecture UNet (seed 0) in fp16,
seeded text embeddings, rectangular 64-px-aligned region masks.  Inputs are resident in HBM when
the timed region starts; the final latents stay on the device (VAE decode is outside the path).

`--in-flight N` (default 2): the K generations are driven by N host threads, each with its own stream and generation
slot (static buffers, captured step, packed K/V, library workspace - DESIGN.md section 6), pulling the next generation
from a shared queue.  Every generation is still ONE batch-1 image with its own inputs and results equal to the
one-at-a-time results bit for bit; what changes is that the GPU interleaves two of them.  `value` / `ms_per_step` are
total images / wall time (throughput); `one_generation_at_a_time` on the same line repeats the K generations with one in
flight (`--in-flight 1` makes that the headline): its ms_per_generation is the latency of one image.  Leg order: one at a
time, rooflines, cpu_baseline, then the generations in flight behind a stall watchdog (`--stall-seconds`).

Extra objects on the JSON line:
  roofline     - the region cross-attention forward kernel (`xp_fwd`, L=4096 level of the same workload) timed live with
                 HIP events around graph-captured back-to-back launches on the stream they run on: algorithmic bytes
                 per launch / average launch duration vs the 8 TB/s HBM peak.  Algorithmic bytes per launch = 6.60 MB
                 per UNet row (2*(2*L*C + 2*S*C) + 4*L*S at L=4096, C=320, S=77) x Bc rows.
  roofline_self_attn - the flash self-attention kernel at the same level against the 2.5 PFLOP/s dense fp16 MFMA peak
                 (4*L^2*C FLOPs per row).
  roofline_conv3x3   - the 3x3 convolution kernel (64x64, 320->320, the most frequent one) against the same peak
  roofline_at_8_images - the same three kernels at Bc = 16 (BASELINE configs[2]: 8 images per GPU): achieved / frac / launch time
  one_generation_at_a_time - see above
  cpu_baseline - the oracle (oracle/unet_ref.py, torch fp32, op-for-op unfused like the reference) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N=1 only); its
                 gpu_vs_cpu_on_the_sample = the product loop on the same sample against the oracle's latents.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="timed generations (each = 25 denoising steps)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images-per-gpu", type=int, default=1)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--model", choices=["sd15", "sdxl-shape"], default="sd15",
                    help="UNet geometry: sd15 (BASELINE configs[0..3], the headline) or the SDXL-base SHAPE of configs[4] (3 levels, transformer "
                         "depth 0 / 2 / 10, 64-channel heads, context 2048; use with --size 1024 --images-per-gpu 2: 16 images over 8 GPUs).  "
                         "The reference has no SDXL pipeline (SURVEY.md 8d): the same processor contract on an SDXL-shaped UNet, random weights")
    ap.add_argument("--denoise-steps", type=int, default=25)
    ap.add_argument("--regions", type=int, default=2)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--miopen-find", action="store_true",
                    help="let MIOpen time its solvers (cudnn.benchmark + MIOPEN_FIND_MODE=1) during warm-up.  Off by default and "
                         "without effect on the SD1.5 step: none of its convolutions is MIOpen's any more (only channel counts "
                         "the hand-written kernels do not cover fall back to it)")
    ap.add_argument("--no-miopen-find", action="store_true", help="(default behaviour; kept for older command lines)")
    ap.add_argument("--deterministic-conv", action="store_true",
                    help="torch.backends.cudnn.deterministic: MIOpen then avoids its split-K (fp16 atomic) igemm kernels but "
                         "falls back to kernels >10x slower on this image; off by default")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="generations in flight per GPU: N host threads, each with its own stream and generation slot (static "
                         "buffers, captured step, packed K/V, library workspace); every generation is still one batch-1 image")
    ap.add_argument("--tuning-profile", choices=["auto", "latency", "throughput"], default="auto",
                    help="launch rules (dsc_set_tuning_profile): auto = 'latency' for the one-at-a-time leg, 'throughput' for the "
                         "generations in flight (each slot re-captures its step, untimed); or one profile for every leg (A/B)")
    ap.add_argument("--stall-seconds", type=float, default=120.0,
                    help="in-flight leg: no generation completed for this long -> print the line (status: stalled, value = the "
                         "one-at-a-time figure) and exit with status 3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched-roofline", action="store_true",
                    help="skip roofline_at_8_images (profiling runs: its Bc = 16 launches of the same kernels would mix into the "
                         "per-kernel averages of the rocprofv3 summary)")
    ap.add_argument("--decode", action="store_true",
                    help="also run the VAE decoder (SURVEY.md 8f rank 1) inside the timed region; off by default so the "
                         "headline stays the denoising path of BASELINE.json")
    ap.add_argument("--no-coalesced", action="store_true",
                    help="skip the coalesced_requests leg (k concurrent batch-1 requests in one captured step, txt2img_coalesced)")
    ap.add_argument("--coalesce-sweep", default="2x1,4x1,8x1,2x2,4x2,8x2",
                    help="the (requests per captured step) x (generations in flight) points of the coalesced_requests leg")
    ap.add_argument("--cpu-sample-steps", type=int, default=25,
                    help="denoising steps of the cpu_baseline leg (oracle on the host cores): all 25 by default (BASELINE.md section 2: "
                         "configs[1] timed in full, about two minutes on 16 cores); fewer -> extrapolated and labelled as such")
    ap.add_argument("--cpu-budget-s", type=float, default=300.0,
                    help="when the configs[0] single step predicts more than this for the cpu_baseline sample (few host cores), the "
                         "sample is cut to the steps that fit (>= 3) and labelled as extrapolated; 0 = no limit")
    return ap.parse_args()


def synthetic_inputs(size, regions, S=77, ctx=768):
    """text embeddings (seed 7), token ids, region masks: phrase r occupies columns 2+2r, 3+2r."""
    from inputs import FakeTokenizer
    g = torch.Generator().manual_seed(7)
    emb = torch.randn(2, S, ctx, generator=g)
    tok = FakeTokenizer()
    words = [f"object{r}a object{r}b" for r in range(regions)]
    ids = [49406, 320]
    for w in words:
        ids += tok(w).input_ids
    ids = ids + [49407] * (S - len(ids))
    import numpy as np
    pos = np.array([ids], dtype=np.int64)
    cells = size // 64
    state = {}
    for r, w in enumerate(words):
        m = np.full((size, size), 255, dtype=np.uint8)
        x0 = (r * cells) // regions
        x1 = ((r + 1) * cells) // regions
        m[(cells // 4) * 64:(3 * cells // 4) * 64, x0 * 64:x1 * 64] = 0
        state[w] = {"map": m, "weight": 0.5, "mask_outsides": 0.0}       # UI defaults app.py:1332-1336
    return emb, [pos.copy(), pos], state, tok


def xattn_source_sha16():
    """identifies the region cross-attention kernel source the PMC traffic figure was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("region_xattn_packed.hip", "xattn_shared.h"):
        h.update(open(os.path.join(ROOT, "diffusionspatialcontrol_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def graph_launch_time_us(fn, launches=50, replays=10):
    """Average duration of ONE launch of fn(): `launches` back-to-back launches are captured into a HIP graph (so the
    host's ~10 us ctypes/launch overhead is out of the picture), the graph is replayed `replays` times and timed with HIP
    events recorded on the stream the replays run on (torch's current stream)."""
    dev = torch.cuda.current_device()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            fn()
    g.replay()
    torch.cuda.synchronize(dev)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(replays):
        g.replay()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / (launches * replays) * 1e3


def csrc_sha16():
    """identifies the kernel sources the stamped files under profiles/ (in_step_kernels.json, pmc_mfma_busy.json) were measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffusionspatialcontrol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


_STAMPED = {}


def stamped(name):
    """profiles/<name> if it was measured on the kernel sources of this tree (its csrc_sha16 matches), else None"""
    if name not in _STAMPED:
        rec = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
            if rec.get("csrc_sha16") != csrc_sha16():
                rec = None
        except Exception:  # noqa: BLE001
            rec = None
        _STAMPED[name] = rec
    return _STAMPED[name]


def in_step_us(key):
    """average duration of a kernel INSIDE the captured UNet step (tools/make_in_step.py on a rocprofv3 kernel trace of this
    command, committed as profiles/in_step_kernels.json); None when the kernels have changed since it was taken"""
    rec = stamped("in_step_kernels.json")
    try:
        return rec["kernels"][key]["in_step"]["avg_us"]
    except Exception:  # noqa: BLE001
        return None


def mfma_busy(prefix):
    """{mfma_busy_pct, ..of_occupied_simds} of the first kernel whose name starts with `prefix` in profiles/pmc_mfma_busy.json
    (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... -- python3 tools/pmc_kernels.py, tools/make_mfma_busy.py)"""
    rec = stamped("pmc_mfma_busy.json")
    if rec:
        for k, v in rec["kernels"].items():
            if k.startswith(prefix):
                return {"kernel": k, "mfma_busy_pct": v["mfma_busy_pct"],
                        "mfma_busy_pct_of_occupied_simds": v["mfma_busy_pct_of_occupied_simds"]}
    return None


def roofline_region_xattn(dev, n_img):
    """The region cross-attention OP (the std over the scores is part of it) on the prepared-operand path the
    pipeline runs, at the L=4096 level of the workload (C=320, H=8, d=40, S=77), Bc = 2*n_img rows: the statistics launch
    (`xp_stats`) + the forward launch (`xp_fwd`).  HBM-bound by arithmetic intensity (61 FLOP/B).  `frac` is the OP's: the
    algorithmic bytes of SURVEY.md 8d over the time of BOTH launches; the forward launch alone is in `forward_only`."""
    from diffusionspatialcontrol_amd import ops
    Bc, H, L, S, d = 2 * n_img, 8, 4096, 77, 40
    C = H * d
    g = torch.Generator().manual_seed(3)
    q = torch.randn(Bc, L, C, generator=g).half().to(dev)
    k = torch.randn(Bc, S, C, generator=g).half().to(dev)
    v = torch.randn(Bc, S, C, generator=g).half().to(dev)
    w = torch.zeros(2, L, S)
    w[:, 1000:2000, 2:4] = 0.5
    w[:, 2500:3500, 4:6] = 0.5
    sig = torch.tensor([7.0], device=dev)
    out = torch.empty(Bc, L, H, d, dtype=torch.half, device=dev)
    q4, k4, v4 = q.view(Bc, L, H, d), k.view(Bc, S, H, d), v.view(Bc, S, H, d)
    packed = ops.xattn_kv_pack(k4, v4)
    ids, rows = ops.compress_region_table(w, pad_rows=True)
    rows = ops.pad_region_rows(rows)                                    # the form the pipeline uploads (model_k_diffusion._compress_tables)
    comp = (ids.to(dev), rows.to(dev))
    call = lambda **kw: ops.region_xattn_packed(q4, packed, S, comp, sig, n_std_groups=n_img, out=out,  # noqa: E731
                                                ref_fp16_rounding=False, **kw)
    call()                                                              # leaves the std partials in the workspace
    pair = graph_launch_time_us(lambda: call())
    fwd = graph_launch_time_us(lambda: call(reuse_stats=True))
    alg_bytes = Bc * (2 * (2 * L * C + 2 * S * C) + 4 * L * S)
    achieved = alg_bytes / (pair * 1e-6) / 1e9
    achieved_fwd = alg_bytes / (fwd * 1e-6) / 1e9
    traffic, traffic_note = None, None
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pj) and n_img == 1:                              # the PMC passes were taken at Bc = 2
        try:
            rec = json.load(open(pj))
            if rec.get("kernel_source_sha16") == xattn_source_sha16():
                traffic = rec.get("xp_fwd_hbm_bytes_per_launch")
                traffic_note = ("HBM bytes of the FORWARD launch: rocprofv3 PMC passes of tools/pmc_xattn.py (profiles/pmc_traffic.json), "
                                "taken on this kernel source; the statistics launch reads Q once more (+5.4 MB)")
            else:
                traffic_note = "profiles/pmc_traffic.json was measured on a different version of the kernel source: not reported"
        except Exception:  # noqa: BLE001
            traffic = None
    # what the launch has to move now that K/V arrive packed and the table compressed: Q + out, the packed K/V images
    # (21 KB per (b, h) at d = 40), uint16 row ids + the distinct rows
    must_move = Bc * (2 * 2 * L * C) + packed.numel() * packed.element_size() \
        + ids.numel() * ids.element_size() + rows.numel() * rows.element_size()
    r = {"kernel": "xp_stats<3,false> + xp_fwd<3,false,false> (region cross-attention incl. the std pass, L=4096 C=320 S=77, Bc=%d)" % Bc,
         "bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
         "frac": round(achieved / 8000.0, 4), "traffic": traffic,
         "traffic_note": traffic_note,
         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": round(pair, 2),
         "what_is_timed": "hot: isolated back-to-back launches of the statistics + forward pair (HIP events around graph-captured launches)",
         "forward_only": {"avg_launch_us": round(fwd, 2), "achieved": round(achieved_fwd, 1), "frac": round(achieved_fwd / 8000.0, 4)},
         "bytes_the_kernel_must_move": int(must_move),
         "frac_of_peak_on_bytes_it_must_move": round(must_move / (fwd * 1e-6) / 8e12, 4),
         "note": "algorithmic bytes = SURVEY.md 8d per-row figure (6.60 MB incl. the dense fp32 table) x Bc rows, charged ONCE for the op; "
                 "the kernels read the table as uint16 row ids + <=32 distinct rows"}
    if n_img == 1:
        st, fw = in_step_us("xp_stats"), in_step_us("xp_fwd")
        if st and fw:
            r["in_step_us"] = {"stats": st, "fwd": fw, "op": round(st + fw, 2),
                               "frac_op": round(alg_bytes / ((st + fw) * 1e-6) / 8e12, 4),
                               "frac_forward_only": round(alg_bytes / (fw * 1e-6) / 8e12, 4),
                               "source": "profiles/in_step_kernels.json: the launches inside the captured UNet step in a rocprofv3 kernel "
                                         "trace of this command on this kernel source (tools/make_in_step.py)"}
        else:
            r["in_step_us"] = None
        mb = mfma_busy("xp_fwd<3")
        r["mfma_busy_pct"] = mb["mfma_busy_pct"] if mb else None
        if mb:
            r["mfma_busy"] = mb
    return r


def roofline_self_attn(dev, n_img):
    """The flash self-attention kernel at the L=4096 level (C=320, H=8, d=40): MFMA-bound (AI = L/2 = 2048 FLOP/B)."""
    from diffusionspatialcontrol_amd import ops
    Bc, H, L, d = 2 * n_img, 8, 4096, 40
    C = H * d
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(Bc, L, 3 * C, generator=g).half().to(dev)
    q, k, v = (qkv[..., i * C:(i + 1) * C].unflatten(-1, (H, d)) for i in range(3))
    # the operand layout the pipeline hands the kernel: the fused QKV projection (dsc_linear_qkv_f16) writes K / V head-major
    # ([2, Bc, H, L, d]; a head's keys contiguous), Q token-major
    k, v = (t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3) for t in (k, v))
    q = q.contiguous()
    out = torch.empty(Bc, L, H, d, dtype=torch.half, device=dev)
    us = graph_launch_time_us(lambda: ops.self_attention(q, k, v, out=out), launches=20, replays=5)
    flops = 4.0 * L * L * C * Bc
    tf = flops / (us * 1e-6) / 1e12
    return {"kernel": "self_attn_fwd (flash self-attention, L=4096 C=320 d=40, Bc=%d; K/V head-major as the pipeline's QKV "
                      "projection writes them)" % Bc, "bound": "mfma",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "traffic": None,
            "algorithmic_flops_per_launch": flops, "avg_launch_us": round(us, 2),
            **_in_step_and_busy("self_attn_fwd", "self_attn_fwd<3", flops, n_img)}


def _in_step_and_busy(step_key, pmc_prefix, flops, n_img):
    """the stamped figures of profiles/ for an MFMA-bound kernel of the Bc = 2 workload: its duration inside the captured step
    and the matrix pipe's busy share from the PMC pass (the metric's "attn MFMA util%")"""
    if n_img != 1:
        return {}
    us = in_step_us(step_key)
    mb = mfma_busy(pmc_prefix)
    return {"in_step_us": us, "frac_in_step": round(flops / (us * 1e-6) / 2.5e15, 4) if us else None,
            "mfma_busy_pct": mb["mfma_busy_pct"] if mb else None,
            "mfma_busy_pct_of_occupied_simds": mb["mfma_busy_pct_of_occupied_simds"] if mb else None}


def roofline_conv3x3(dev, n_img):
    """The 3x3 convolution kernel at the 64x64 level (320 -> 320 channels), the most frequent convolution of a step: MFMA-bound."""
    from diffusionspatialcontrol_amd import ops
    Bc, C, hw = 2 * n_img, 320, 64
    g = torch.Generator().manual_seed(6)
    cl = torch.channels_last
    x = torch.randn(Bc, C, hw, hw, generator=g).half().to(dev).contiguous(memory_format=cl)
    w = (torch.randn(C, C, 3, 3, generator=g) / (3.0 * C ** 0.5)).half().to(dev).contiguous(memory_format=cl)
    us = graph_launch_time_us(lambda: ops.conv3x3(x, w, None), launches=20, replays=5)
    flops = 2.0 * Bc * hw * hw * C * C * 9
    tf = flops / (us * 1e-6) / 1e12
    return {"kernel": "conv3x3_kernel<16> (3x3 convolution, 64x64, 320->320, Bc=%d)" % Bc, "bound": "mfma",
            "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4), "traffic": None,
            "algorithmic_flops_per_launch": flops, "avg_launch_us": round(us, 2),
            **_in_step_and_busy("conv3x3_320_320_64x64", "conv3x3_kernel<16, 3, 0> [320", flops, n_img)}


def host_cores():
    """cores this process may actually use (the GPU box exposes 256 logical CPUs but grants a share of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count()
    quota = None
    try:                                        # cgroup v2 cpu.max = "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:  # noqa: BLE001
        pass
    n = min(n, quota) if quota else n
    if quota is None and n > 64:
        # no cgroup quota visible and the whole host in the affinity mask: a 1-GPU box of this pool is granted a 16-CPU share of
        # its host (more threads than that only contend), so that is what an unlabelled 256-CPU view is taken to mean
        n = 16
    if os.environ.get("DSC_CPU_THREADS"):
        n = int(os.environ["DSC_CPU_THREADS"])
    return max(1, n)


def cpu_baseline(unet, cfg, sigmas, text, region_state, latents, guidance, sample_steps, total_steps, config1=None, budget_s=0.0):
    """oracle on the host cores: configs[1] = the `total_steps` denoising steps of the same image, timed
    in full by default (`--cpu-sample-steps` < total: that many steps, extrapolated and labelled); `config1` = (region tables of
    ONE mask) -> configs[0], one denoise step of a 64x64 latent with one region mask, timed in full.  `budget_s` > 0: when the
    configs[0] step predicts more than that for the sample (a box that grants few host cores), the sample is cut to the steps
    that fit (at least 3) and labelled as extrapolated - the default run has to finish within minutes on any box"""
    from oracle import unet_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    sig = sigmas.float().cpu().tolist()
    lat = latents.float().cpu() * math.sqrt(sig[0] ** 2 + 1)
    c1 = None
    if config1 is not None:
        t0 = time.perf_counter()
        c1_ref = unet_ref.denoise_loop(sd, cfg, lat, sig, text.float().cpu(), config1, guidance, steps_limit=1)
        c1 = (time.perf_counter() - t0, c1_ref)
        if budget_s > 0 and c1[0] * sample_steps > budget_s:
            sample_steps = max(3, min(sample_steps, int(budget_s / c1[0])))
    t0 = time.perf_counter()
    ref = unet_ref.denoise_loop(sd, cfg, lat, sig, text.float().cpu(), region_state, guidance, steps_limit=sample_steps)
    dt = time.perf_counter() - t0
    per_image = dt / sample_steps * total_steps
    full = sample_steps >= total_steps
    res = {"value": round(1.0 / per_image, 5), "unit": "images/s", "cores": cores, "kind": "port", "steps_timed": sample_steps,
           "sample": (f"all {total_steps} denoising steps of one 512x512 image (BASELINE configs[1]) through the fp32 torch oracle, "
                      f"timed in full: {dt:.1f} s" if full else
                      f"{sample_steps} of {total_steps} denoising steps of one 512x512 image through the fp32 torch oracle "
                      f"({dt:.1f} s), scaled x{total_steps}/{sample_steps}"),
           ("seconds_per_image" if full else "seconds_per_image_extrapolated"): round(per_image, 1)}
    if c1 is not None:
        res["config1_single_step_s"] = round(c1[0], 2)
        res["config1"] = "BASELINE configs[0]: one CFG denoise step (UNet forward on 2 rows + DPM++ 2M update) of a 64x64 latent, 1 region mask"
    return res, ref, (c1[1] if c1 is not None else None)


def gpu_sample(pipe, sigmas, text, region_state, latents, guidance, sample_steps):
    """the product loop on the same truncated schedule as the cpu_baseline sample (same inputs, fp16, HIP kernels)"""
    import inspect
    wf = inspect.signature(pipe.txt2img).parameters["weight_func"].default
    x0 = latents * (sigmas[0] ** 2 + 1) ** 0.5                           # as txt2img does (reference model_k_diffusion.py:1043)
    return pipe._denoise_fused(x0, sigmas[:sample_steps + 1], text, region_state, wf, guidance, latents.shape[0], {}, -1, 0)


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILDREN (torch.distributed.run, one
    process per GPU) before this process has touched the GPU - never an exec of a process that has - relay their output
    (rank 0 prints the JSON line) and exit with the worst child status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # stdout of this command is ONE JSON line: whatever else the ranks' libraries print there (gloo's connection notes) goes to
    # stderr with the rest of the log
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith("{") and line.rstrip().endswith("}"):
            print(line, end="", flush=True)
        else:
            print(line, end="", file=sys.stderr, flush=True)
    return proc.wait()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(launch_ranks(a.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: run `python bench.py "
                         f"--gpus N` (it starts the N ranks itself) or `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`")
    if a.miopen_find and not a.no_miopen_find:
        os.environ.setdefault("MIOPEN_FIND_MODE", "1")         # must be set before MIOpen initialises
        torch.backends.cudnn.benchmark = True                   # let MIOpen time its solvers per conv shape (warm-up only)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = world > 1
    backend = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local = local % torch.cuda.device_count()                 # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DSC_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for single-GPU rehearsals
        if backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group(backend)

    if dist:
        # ranks must run the SAME kernels on the same image (per-image equality across ranks): a GEMM that is
        # left to hipBLASLt takes the heuristic's first algorithm instead of the per-process cold-timed one (linear_lt.hip)
        os.environ.setdefault("DSC_LT_TUNE", "1")
    from diffusionspatialcontrol_amd import ops
    from diffusionspatialcontrol_amd.modules.model_k_diffusion import SD15Scheduler, StableDiffusionPipeline
    from diffusionspatialcontrol_amd.modules.u_net_condition_modify import UNet2DConditionModel, UNetConfig
    from diffusionspatialcontrol_amd.parallel import broadcast_generation_inputs, shard_image_indices
    ops.GRAPHS_ENABLED = not a.no_graph
    torch.backends.cudnn.deterministic = bool(a.deterministic_conv)

    cfg = UNetConfig.sd15() if a.model == "sd15" else UNetConfig.sdxl_base()
    torch.manual_seed(0)
    with torch.device(dev):
        unet = UNet2DConditionModel(cfg)
    unet = unet.half().eval()
    emb, ids, state, tok = synthetic_inputs(a.size, a.regions, ctx=cfg.cross_attention_dim)
    emb = emb.to(dev)
    if dist:                                                    # rank 0's embeddings are THE embeddings
        emb = broadcast_generation_inputs(emb, src=0)
    vae = None
    if a.decode:
        from diffusionspatialcontrol_amd.modules.vae_decoder import AutoencoderKLDecoder
        with torch.device(dev):
            vae = AutoencoderKLDecoder()
        vae = vae.half().eval()
    pipe = StableDiffusionPipeline(vae, None, tok, unet, SD15Scheduler())
    n_img = a.images_per_gpu
    my_images = shard_image_indices(n_img * world, rank, world)
    def start_latents(indices):
        return torch.stack([torch.randn(4, a.size // 8, a.size // 8, generator=torch.Generator().manual_seed(1000 + i))
                            for i in indices]).half().to(dev)

    lat = lat_mine = start_latents(my_images)

    def generate(slot=0, lat=None):
        lat = lat_mine if lat is None else lat
        out = pipe.txt2img(None, height=a.size, width=a.size, num_inference_steps=a.denoise_steps, guidance_scale=7.5,
                           latents=lat, output_type="latent", region_map_state=state, sampler_name="sample_dpmpp_2m",
                           sampler_opt={"scheduler": "karras"}, prompt_embeds=emb[1:2], negative_prompt_embeds=emb[0:1],
                           text_input_ids=ids, num_images_per_prompt=n_img, slot=slot)[0]
        if a.decode:                                         # pixels stay on the device (decode_latents' .cpu() is host I/O)
            out = (vae.decode(out / vae.config.scaling_factor).sample / 2 + 0.5).clamp(0, 1)
        return out

    out = None
    ops.set_tuning_profile("latency" if a.tuning_profile == "auto" else a.tuning_profile)
    nfl = 1 if (a.decode or a.no_graph) else max(1, a.in_flight)     # slots exist for the captured denoising loop
    if nfl > 4:
        raise SystemExit("--in-flight: at most 4 generation slots (dsc_set_workspace_slot)")
    streams = [torch.cuda.Stream() for _ in range(nfl)] if nfl > 1 else None
    if nfl > 1:
        torch.cuda.synchronize()
        for s_i, st in enumerate(streams):                  # captures and algorithm timing happen here, one slot at a time
            with torch.cuda.stream(st):
                for _ in range(max(1, a.warmup)):
                    out = generate(s_i)
            torch.cuda.synchronize()
    else:
        for _ in range(a.warmup):
            out = generate()
    import contextlib
    import threading

    own = [0.0]

    def timed(fn):
        """K generations between barrier + synchronize on both sides; the MAX over ranks"""
        if dist:
            td.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        own[0] = time.perf_counter() - t0                   # this rank's own time, before it waits for the others
        if dist:
            td.barrier()
        dt_ = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([dt_], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            td.all_reduce(tt, op=td.ReduceOp.MAX)
            dt_ = tt.item()
        return dt_, r

    def per_rank():
        """every rank's own time for its K generations (taken before the closing barrier), gathered: images/s per rank"""
        fn_seconds = own[0]
        mine = torch.tensor([fn_seconds], device=dev if dist and backend == "nccl" else "cpu", dtype=torch.float64)
        if not dist:
            return [round(n_img * a.steps / fn_seconds, 4)]
        allr = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(allr, mine)
        return [round(n_img * a.steps / t.item(), 4) for t in allr]

    def one_at_a_time():
        o = None
        with (torch.cuda.stream(streams[0]) if streams else contextlib.nullcontext()):
            for _ in range(a.steps):
                o = generate(0)
        return o

    # Order: the one-at-a-time leg, the roofline / cpu_baseline legs, and only then the generations in flight - so that a
    # complete line exists before the only leg in which two streams share the chip, and a stall there (never observed;
    # hipBLASLt's stream-K kernels spin on each other's partial tiles) still ends with a valid line instead of a hung job.
    dt_seq, out = timed(one_at_a_time)
    rates_seq = per_rank()
    finite = bool(torch.isfinite(out).all().item())
    images = n_img * world * a.steps
    res = None
    if rank == 0:
        res = {
            "metric": ("512x512 images/sec, SD1.5 25-step DPM++2M Karras with region-biased cross-attention" if a.model == "sd15" and a.size == 512
                       else f"{a.size}x{a.size} images/sec, {a.model} UNet, 25-step DPM++2M Karras with region-biased cross-attention"),
            "value": round(images / dt_seq, 4), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt_seq / a.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{'SD1.5' if a.model == 'sd15' else 'SDXL-base-SHAPED UNet (configs[4] geometry)'} {a.size}x{a.size}, {a.denoise_steps}-step DPM++2M Karras, CFG 7.5, "
                                   f"{a.regions} region masks, {n_img} image(s) per generation, "
                                   f"1 generation(s) in flight per GPU",
                       "images_per_gpu": n_img, "parallelism": f"dp{world} (independent images, no per-step collective)",
                       "per_rank_images_per_s": rates_seq, "max_over_ranks_s": round(dt_seq, 4),
                       "generations_in_flight": 1, "images_per_generation": n_img, "slots_equal_one_at_a_time": None,
                       "hip_graph": bool(ops.GRAPHS_ENABLED), "outputs_finite": finite, "vae_decode_in_timed_region": bool(a.decode)},
        }
        res["roofline"] = roofline_region_xattn(dev, n_img)
        res["roofline_self_attn"] = roofline_self_attn(dev, n_img)
        res["roofline_conv3x3"] = roofline_conv3x3(dev, n_img)
        if n_img == 1 and not a.no_batched_roofline:
            # the same three kernels at the batch of BASELINE configs[2] (8 images per GPU, Bc = 16): what they reach once a
            # launch has enough workgroups to fill the chip - the bench workload above (Bc = 2) is launch-latency bound
            pick = lambda r: {k_: r[k_] for k_ in ("achieved", "unit", "frac", "avg_launch_us", "forward_only") if k_ in r}    # noqa: E731
            res["roofline_at_8_images"] = {"region_xattn": pick(roofline_region_xattn(dev, 8)),
                                           "self_attn": pick(roofline_self_attn(dev, 8)),
                                           "conv3x3": pick(roofline_conv3x3(dev, 8))}
        if world == 1 and not a.no_cpu_baseline and a.model == "sd15":
            from diffusionspatialcontrol_amd.modules.encode_region_map_function import encode_region_map
            rs = encode_region_map(pipe, state, a.size, a.size, 1, text_ids=ids)
            sig = pipe.get_sigmas(a.denoise_steps, {"scheduler": "karras"}).half()
            text = torch.cat([emb[0:1], emb[1:2]]).half()
            a.cpu_sample_steps = max(1, min(a.cpu_sample_steps, a.denoise_steps))
            # configs[0]: ONE region mask (the first of the workload's), one step
            first = next(iter(state))
            rs1 = encode_region_map(pipe, {first: state[first]}, a.size, a.size, 1, text_ids=ids)
            res["cpu_baseline"], ref, ref1 = cpu_baseline(unet, cfg, sig, text, rs, lat[:1], 7.5, a.cpu_sample_steps, a.denoise_steps,
                                                          config1=rs1, budget_s=a.cpu_budget_s)
            a.cpu_sample_steps = res["cpu_baseline"]["steps_timed"]
            # the checker's other use: the timed GPU path and the CPU baseline computed the same thing on this sample
            got = gpu_sample(pipe, sig.to(dev), text.to(dev), rs, lat[:1], 7.5, a.cpu_sample_steps).float().cpu()
            err = (got - ref).abs()
            res["cpu_baseline"]["gpu_vs_cpu_on_the_sample"] = {
                "max_abs_err": round(err.max().item(), 5), "mean_abs_err": round(err.mean().item(), 6),
                "ref_max_abs": round(ref.abs().max().item(), 4),
                "note": f"latents after {a.cpu_sample_steps} of {a.denoise_steps} steps: fp16 HIP path vs fp32 oracle"}
            got1 = gpu_sample(pipe, sig.to(dev), text.to(dev), rs1, lat[:1], 7.5, 1).float().cpu()
            err1 = (got1 - ref1).abs()
            res["cpu_baseline"]["config1_gpu_vs_cpu"] = {"max_abs_err": round(err1.max().item(), 5), "mean_abs_err": round(err1.mean().item(), 6),
                                                         "ref_max_abs": round(ref1.abs().max().item(), 4)}
    # self-describing multi-GPU lines: what each rank ran on
    me = f"{torch.cuda.get_device_name(dev)} (cuda:{local} of {torch.cuda.device_count()} visible)"
    if dist:
        names = [None] * world
        td.all_gather_object(names, me)
    else:
        names = [me]
    if res is not None:
        res["config"]["devices"] = names
    ranks_equal = True
    if dist and not a.decode:
        # per-image equality across ranks: every rank hashes the final latents of its images (one generation
        # at a time, latency rules); rank 0 generates EVERY rank's images itself and compares - the same bits are expected since
        # no kernel of the step is chosen per process any more (no library GEMM) and every kernel is bit-reproducible
        import hashlib
        ops.set_tuning_profile("latency" if a.tuning_profile == "auto" else a.tuning_profile)
        mine = hashlib.sha256(generate(0).float().cpu().numpy().tobytes()).hexdigest()
        hashes = [None] * world
        td.all_gather_object(hashes, mine)
        if res is not None:
            regen = [hashlib.sha256(generate(0, start_latents(shard_image_indices(n_img * world, r_, world))).float().cpu().numpy().tobytes()).hexdigest()
                   for r_ in range(world)]
            res["config"]["ranks_equal_single_process"] = ranks_equal = bool(regen == hashes)
            res["config"]["dist_backend"] = backend if backend == "nccl" else f"{backend} (NOT RCCL: a rehearsal, not a multi-GPU measurement)"
        td.barrier()
    if nfl > 1:
        # the launch rules for generations that SHARE the chip (include/dsc_hip.h, dsc_set_tuning_profile): the legs above ran
        # under "latency"; every slot re-captures its step under "throughput" here, outside the timed region
        ref_same_rules = out
        if a.tuning_profile == "auto":
            ops.set_tuning_profile("throughput")
            if res is not None and "cpu_baseline" in res:
                # the headline leg's launch rules against the oracle too (the check above ran under the latency rules)
                got = gpu_sample(pipe, sig.to(dev), text.to(dev), rs, lat[:1], 7.5, a.cpu_sample_steps).float().cpu()
                err = (got - ref).abs()
                res["cpu_baseline"]["gpu_vs_cpu_on_the_sample_throughput_profile"] = {
                    "max_abs_err": round(err.max().item(), 5), "mean_abs_err": round(err.mean().item(), 6)}
            for s_i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    o_ = generate(s_i)
                torch.cuda.synchronize()
                if s_i == 0:
                    ref_same_rules = o_              # one generation alone on the chip under the in-flight leg's launch rules
        progress = [0, time.monotonic()]
        outs, errs = [None] * nfl, []
        todo, todo_lock = iter(range(a.steps)), threading.Lock()

        def drive(s_i):
            try:
                torch.cuda.set_device(dev)                  # the current device is per host thread
                with torch.cuda.stream(streams[s_i]):
                    while True:                             # a.steps generations in total, whichever slot is free next
                        with todo_lock:
                            if next(todo, None) is None:
                                break
                        outs[s_i] = generate(s_i)
                        progress[0] += 1
                        progress[1] = time.monotonic()
            except BaseException as e:                      # noqa: BLE001 - re-raised on the main thread
                errs.append(e)

        def in_flight():
            progress[1] = time.monotonic()
            threads = [threading.Thread(target=drive, args=(i,), daemon=True) for i in range(nfl)]
            for t in threads:
                t.start()
            while any(t.is_alive() for t in threads):
                for t in threads:
                    t.join(timeout=0.25)
                if time.monotonic() - progress[1] > a.stall_seconds:
                    msg = f"stalled after {progress[0]} of {a.steps} generations with {nfl} in flight"
                    if res is not None:
                        res["status"] = "stalled"
                        res["config"]["in_flight_leg"] = msg + "; value is the one-at-a-time figure"
                        print(json.dumps(res), flush=True)
                    print(f"bench.py: rank {rank}: {msg}", file=sys.stderr, flush=True)
                    os._exit(3)          # a stall is a FAILURE (non-zero; under torchrun it takes the other ranks down too);
                    #                      the stuck threads cannot be joined, hence _exit
            if errs:
                raise errs[0]
            return [o for o in outs if o is not None]

        dt, slot_outs = timed(in_flight)
        rates_fl = per_rank()
        # every generation has the same inputs and every kernel is bit-reproducible: the slots' last results and the result of a
        # generation that ran ALONE under the same launch rules must be EQUAL - a free check on every run that the generations in
        # flight did not interfere.  (Against the one-at-a-time leg, which runs under the latency rules, the latents agree to
        # rounding: the two profiles' own kernels give equal bytes, but a library GEMM may be run by another algorithm.)
        slots_agree = all(torch.equal(o, ref_same_rules) for o in slot_outs)
        prof_diff = max(float((o.float() - out.float()).abs().max().item()) for o in slot_outs)
        if res is not None:
            res["one_generation_at_a_time"] = {"value": res["value"], "unit": "images/s", "ms_per_generation": res["ms_per_step"],
                                               "note": "the same K generations with one in flight (latency of one image)"}
            res["value"], res["ms_per_step"] = round(images / dt, 4), round(dt / a.steps * 1e3, 2)
            res["config"].update({"generations_in_flight": nfl, "slots_equal_one_at_a_time": slots_agree,
                                  "max_abs_diff_between_the_profiles_latents": round(prof_diff, 6),
                                  "tuning_profile": {"one_generation_at_a_time": "latency" if a.tuning_profile == "auto" else a.tuning_profile,
                                                     "in_flight": ops.tuning_profile()},
                                  "per_rank_images_per_s": rates_fl, "max_over_ranks_s": round(dt, 4),
                                  "outputs_finite": finite and all(bool(torch.isfinite(o).all().item()) for o in slot_outs),
                                  "workload": res["config"]["workload"].replace("1 generation(s) in flight", f"{nfl} generation(s) in flight")})
    if res is not None and world == 1 and n_img == 1 and not a.no_coalesced and not a.decode and not a.no_graph and a.model == "sd15":
        # Serving mode (not the headline, which stays BASELINE configs[1]: batch-1 generations): k CONCURRENT batch-1 requests of
        # the same workload - own start latent each, configs[1]'s prompt and masks - denoised by ONE captured step per sigma
        # (txt2img_coalesced: per-request std groups and tables, every kernel on k times the rows), f such generations in flight
        ops.set_tuning_profile("throughput" if a.tuning_profile == "auto" else a.tuning_profile)

        def coalesced(k, f):
            reqs = [{"prompt_embeds": emb[1:2], "negative_prompt_embeds": emb[0:1], "text_input_ids": ids, "region_map_state": state,
                     "latents": start_latents([i])} for i in range(k)]
            gen = lambda s_i: pipe.txt2img_coalesced(reqs, height=a.size, width=a.size, num_inference_steps=a.denoise_steps,   # noqa: E731
                                                     guidance_scale=7.5, sampler_opt={"scheduler": "karras"}, slot=s_i)
            sts = [torch.cuda.Stream() for _ in range(f)]
            for s_i, st_ in enumerate(sts):                   # capture + warm-up per slot, untimed
                with torch.cuda.stream(st_):
                    gen(s_i)
                torch.cuda.synchronize()
            n_gen = max(2 * f, (16 + k - 1) // k)
            todo_, lock_ = iter(range(n_gen)), threading.Lock()
            errs_ = []

            def drive_(s_i):
                try:
                    torch.cuda.set_device(dev)
                    with torch.cuda.stream(sts[s_i]):
                        while True:
                            with lock_:
                                if next(todo_, None) is None:
                                    break
                            gen(s_i)
                except BaseException as e:                  # noqa: BLE001
                    errs_.append(e)

            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            ths = [threading.Thread(target=drive_, args=(i,), daemon=True) for i in range(f)]
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join(timeout=a.stall_seconds)
            if any(t_.is_alive() for t_ in ths):
                raise RuntimeError(f"coalesced leg k={k} f={f} stalled")
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0_
            if errs_:
                raise errs_[0]
            return {"requests_per_step": k, "generations_in_flight": f, "images_per_s": round(n_gen * k / dt_, 3),
                    "ms_per_image_latency": round(dt_ / n_gen * f * 1e3, 1), "generations_timed": n_gen}

        try:
            sweep = [coalesced(*map(int, pt.split("x"))) for pt in a.coalesce_sweep.split(",") if pt]
            best = max(sweep, key=lambda r_: r_["images_per_s"])
            res["coalesced_requests"] = {
                "best": best, "sweep": sweep,
                "vs_headline": round(best["images_per_s"] / res["value"], 3),
                "note": "serving mode, NOT the headline: k concurrent batch-1 requests (own latent each; this workload's prompt and "
                        "2 masks) share one captured UNet step per sigma (StableDiffusionPipeline.txt2img_coalesced: per-request std "
                        "groups, rows [u_0..u_k-1, c_0..c_k-1]); ms_per_image_latency = wall time of a request from entering a "
                        "batch to its latents (every request of a batch finishes with the batch)"}
        except Exception as e:  # noqa: BLE001 - the headline line must survive a failure of this extra leg
            res["coalesced_requests"] = {"error": repr(e)}
    if res is not None:
        try:
            import ctypes
            from diffusionspatialcontrol_amd import _lib
            st3 = (ctypes.c_longlong * 3)()
            _lib.load_library().dsc_linear_lt_stats(st3)
            res["config"]["library_gemm_algorithms"] = {"shapes": int(st3[0]), "candidates_offered": int(st3[1]),
                                                        "dropped_needing_workspace": int(st3[2]),
                                                        "note": "shapes = 0: no GEMM of the run went to hipBLASLt (every linear on gemm_tn_f16 / "
                                                                "split-K, ops.USE_LIBRARY_GEMM off); DSC_LIBRARY_GEMM=1 restores the library "
                                                                "routing, workspace-free algorithms only (linear_lt.hip)"}
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(res), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()
    if not ranks_equal:
        # README / INTEGRATION claim the same bits on every rank: a run that shows otherwise must not look like rc 0
        print("bench.py: a rank's final latents differ from rank 0's regeneration of the same images", file=sys.stderr, flush=True)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
