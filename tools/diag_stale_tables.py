"""diagnostic: which generation-to-generation state does a captured step keep?  (tiny UNet, tables with 48 / 5 distinct rows)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

from test_full_size_parity_gpu import _random_tables, _run_tiny, _tiny_pipe  # noqa: E402
from diffusionspatialcontrol_amd import ops  # noqa: E402

wf = lambda w, s, qk: w * s * qk.std()  # noqa: E731
for distinct in (48, 5):
    for graphs in (True, False):
        ops.GRAPHS_ENABLED = graphs
        a, b = _random_tables(1, distinct), _random_tables(2, distinct)
        cfg, p1 = _tiny_pipe()
        a1 = _run_tiny(p1, cfg, a, wf)
        b1 = _run_tiny(p1, cfg, b, wf)
        b1b = _run_tiny(p1, cfg, b, wf)
        _, p2 = _tiny_pipe()
        b2 = _run_tiny(p2, cfg, b, wf)
        b2b = _run_tiny(p2, cfg, b, wf)
        a2 = _run_tiny(p2, cfg, a, wf)
        d = lambda x, y: (x - y).abs().max().item()  # noqa: E731
        print(f"distinct={distinct} graphs={graphs}: |a1-b1|={d(a1, b1):.3f} |b1-b1b|={d(b1, b1b):.3f} |b1-b2|={d(b1, b2):.3f} "
              f"|b2-b2b|={d(b2, b2b):.3f} |a1-a2|={d(a1, a2):.3f} scale={b2.abs().max().item():.1f}", flush=True)
