"""Microbenchmark: bf16 GEMM shapes of a frozen ViT tower and the whole frozen tower, fp32 vs bf16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dclip_amd import ops, config as dcfg, synth
from dclip_amd.clip_model import from_hf_state_dict

dev = torch.device("cuda:0")


def timeit(f, n=20, w=5):
    for _ in range(w):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(102400, 2304, 768), (102400, 768, 768), (102400, 3072, 768), (102400, 768, 3072), (12800, 2304, 768), (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072), (12544, 768, 3072),
                (65792, 3072, 1024), (65792, 1024, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    af, wf = a.float(), w.float()
    t16 = timeit(lambda: ops.gemm_bf16(a, w, bias=b))
    t32 = timeit(lambda: ops.gemm(af, wf, ops.LAYOUT_NT, bias=b))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: bf16 {t16:.3f} ms {fl / t16 / 1e9:.0f} TF/s | fp32 {t32:.3f} ms {fl / t32 / 1e9:.0f} TF/s", flush=True)

for name, mk, B in [("ViT-B/32", dcfg.vit_b32, 256), ("ViT-L/14", dcfg.vit_l14, 64)]:
    cfg = mk()
    m = from_hf_state_dict(cfg, synth.synth_clip_state_dict(cfg, seed=0), device=dev)
    pix = torch.randn(B, 3, cfg.vision.image_size, cfg.vision.image_size, device=dev)
    with torch.no_grad():
        t32 = timeit(lambda: m.get_image_features(pixel_values=pix), n=5, w=2)
        t16 = timeit(lambda: m.get_image_features(pixel_values=pix, precision="bf16"), n=5, w=2)
    print(f"{name} frozen tower B={B}: fp32 {t32:.2f} ms ({B / t32 * 1e3:.0f} img/s) | bf16 {t16:.2f} ms ({B / t16 * 1e3:.0f} img/s)", flush=True)
