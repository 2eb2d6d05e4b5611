"""per-workgroup timeline of dsc_conv3x3_nhwc_f16 (dsc_debug_set_conv_stamps): start skew, prologue / loop / epilogue
lengths, workgroups per CU"""
import sys, os, ctypes, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from diffusionspatialcontrol_amd import ops, _lib
lib = _lib.load_library(); dev = "cuda"
for (B, cin, cout, hw, splits) in [(2, 320, 320, 64, 1), (2, 960, 320, 64, 1), (2, 1280, 1280, 16, 5), (2, 1280, 1280, 8, 10)]:
    x = torch.randn(B, cin, hw, hw, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)).half().contiguous(memory_format=torch.channels_last)
    for _ in range(3): ops.conv3x3(x, w, None, splits=splits)
    nwg = 8192
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    lib.dsc_debug_set_conv_stamps(ctypes.c_void_p(buf.data_ptr()))
    ops.conv3x3(x, w, None, splits=splits)
    torch.cuda.synchronize()
    lib.dsc_debug_set_conv_stamps(None)
    t = buf.cpu().numpy().reshape(nwg, 8)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    start, ls, le, end = [(t[:, i] - t0) / 100.0 for i in range(4)]       # us
    hw_id = t[:, 7]
    cu_key = [(int(v) >> 32, (int(v) >> 8) & 0xf, (int(v) >> 13) & 0x7, (int(v) >> 16) & 0xf) for v in hw_id]  # xcc, cu, sh, se
    per_cu = collections.Counter(cu_key)
    multi = np.array([per_cu[k] for k in cu_key])
    print(f"B{B} {cin}->{cout} @{hw} s{splits}: {len(t)} workgroups on {len(per_cu)} CUs (max {max(per_cu.values())}/CU); span {end.max():.1f} us; "
          f"start {start.mean():.1f} (max {start.max():.1f}); prologue {np.mean(ls-start):.2f}; loop {np.mean(le-ls):.2f} (min {np.min(le-ls):.2f} max {np.max(le-ls):.2f}); epilogue {np.mean(end-le):.2f}")
    for m in sorted(set(multi)):
        sel = multi == m
        print(f"    {m} wg/CU: n={sel.sum()} loop us {np.mean((le-ls)[sel]):.2f}  loop cycles/step {np.mean(t[sel,5]) / ((cin//64//max(splits,1))*9):.0f}  prologue cyc {np.mean(t[sel,4]):.0f} epilogue cyc {np.mean(t[sel,6]):.0f}  end {np.mean(end[sel]):.1f}")
