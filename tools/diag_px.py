"""diagnostic: packed / generic region cross-attention outputs of two builds of the library (DSC_LIB_PATH), bit for bit"""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
if len(sys.argv) > 1:
    import torch
    from inputs import attn_inputs
    from diffusionspatialcontrol_amd import ops
    Bc, H, L, S, d, Bw, ng = 2, 8, 4096, 77, 40, 2, 1
    x = attn_inputs(f"packed/{Bc}/{H}/{L}/{S}/{d}", Bc=Bc, H=H, L=L, S=S, d=d, Bw=Bw)
    q, k, v, w = (torch.from_numpy(x[n]) for n in ("q", "k", "v", "w"))
    qd, kd, vd = q.cuda().half(), k.cuda().half(), v.cuda().half()
    res = {}
    for ref16 in (True, False):
        res[f"gen{ref16}"] = ops.region_xattn(qd, kd, vd, w.cuda(), 2.0, n_std_groups=ng, ref_fp16_rounding=ref16).cpu()
        res[f"gen0{ref16}"] = ops.region_xattn(qd, kd, vd, None, ref_fp16_rounding=ref16).cpu()
        packed = ops.xattn_kv_pack(kd, vd, layout="bhld")
        comp = ops.compress_region_table(w.cuda())
        res[f"pk{ref16}"] = ops.region_xattn_packed(qd.transpose(1, 2), packed, S, comp, 2.0, n_std_groups=ng, ref_fp16_rounding=ref16).transpose(1, 2).cpu()
        res[f"pk0{ref16}"] = ops.region_xattn_packed(qd.transpose(1, 2), packed, S, None, ref_fp16_rounding=ref16).transpose(1, 2).cpu()
    torch.save(res, sys.argv[1])
else:
    import torch
    outs = {}
    old_lib = os.environ.get("DSC_OLD_LIB", os.path.join(root, "tools", "_ab", "libdsc_old.so"))     # another build of the same ABI
    if not os.path.exists(old_lib):
        raise SystemExit(f"{old_lib} not found: build another commit's library there (python -m diffusionspatialcontrol_amd.build in a worktree) or set DSC_OLD_LIB")
    for name, lib in (("new", ""), ("old", old_lib)):
        env = dict(os.environ)
        if lib:
            env["DSC_LIB_PATH"] = lib
        f = f"/tmp/diag_{name}.pt"
        subprocess.check_call([sys.executable, __file__, f], env=env)
        outs[name] = torch.load(f)
    for key in outs["new"]:
        a, b = outs["new"][key].float(), outs["old"][key].float()
        print(f"{key:10s} new vs old: differing {int((a != b).sum())}  max {float((a - b).abs().max()):.3e}")
    for r in ("True", "False"):
        for n in ("new", "old"):
            a, b = outs[n]["pk" + r].float(), outs[n]["gen" + r].float()
            a0, b0 = outs[n]["pk0" + r].float(), outs[n]["gen0" + r].float()
            print(f"{n} ref16={r}: packed vs generic differing {int((a != b).sum())}, no-bias {int((a0 != b0).sum())}")
