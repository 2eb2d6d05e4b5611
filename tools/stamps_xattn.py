"""in-kernel stamps of xp_fwd workgroup 0 (diagnostic; cdna_hip_programming.md section 7 'In-kernel stamps')"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusionspatialcontrol_amd import ops, _lib
dev = "cuda"
lib = _lib.load_library()
buf = torch.zeros(256, dtype=torch.int64, device=dev)
lib.dsc_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
names = ["start", "loads issued+std+bias", "vmcnt(0)", "barrier", "scores", "softmax", "PV+store issue"]
for (Bc, H, L, d) in [(2, 8, 4096, 40), (2, 8, 64, 160)]:
    S, C = 77, H * d
    q = torch.randn(Bc, L, C, device=dev).half(); k = torch.randn(Bc, S, C, device=dev).half(); v = torch.randn(Bc, S, C, device=dev).half()
    w = torch.zeros(2, L, S, device=dev); w[:, : L // 3, 2:4] = 0.5
    q4, k4, v4 = q.view(Bc, L, H, d), k.view(Bc, S, H, d), v.view(Bc, S, H, d)
    packed = ops.xattn_kv_pack(k4, v4); comp = ops.compress_region_table(w, pad_rows=True); comp = (comp[0], ops.pad_region_rows(comp[1]))
    for ref16, tpwflag in ((False, 0), (False, 64), (False, 128)):
        for bias in (comp, None):
            for _ in range(20):
                ops.region_xattn_packed(q4, packed, S, bias, 3.0, ref_fp16_rounding=ref16, debug_flags=32 | tpwflag)
            torch.cuda.synchronize()
            st = buf.cpu().view(4, 32, 2)[0]          # wave 0
            if bias:
                tt = st[:, 0].tolist()
                print(f"      prologue detail: DMA + Q + row loads issued {tt[7]-tt[0]} | partial loads issued {tt[8]-tt[7]} | rows + partials stored to LDS (loads landed) {tt[9]-tt[8]}")
            t = st[:, 0].tolist(); r = st[:, 1].tolist()
            mhz = (t[6] - t[0]) / max(r[6] - r[0], 1) * 100.0
            print(f"L{L} d{d} tpwflag={tpwflag} ref16={ref16} bias={'y' if bias else 'n'}: clock ~{mhz:.0f} MHz; cycles:", " | ".join(f"{names[i+1]} {t[i+1]-t[i]}" for i in range(6)), f"| total {t[6]-t[0]} cyc = {(r[6]-r[0])*10} ns")
