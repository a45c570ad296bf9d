"""Image data path (zero_amd/loader.py + csrc/image.hip) on one MI355X: images/s and achieved HBM GB/s of
pad-mode and anyres preprocessing, next to PIL + the transformers CLIPImageProcessor on the host cores.
Algorithmic bytes per image: the u8 source read once + the normalised output written once (fp32 here, as the
reference's CPU tensors are; bf16 halves the write)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zero_amd import loader as L

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
N = 64
rs = np.random.RandomState(0)
host = [rs.randint(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(N)]
dev = [torch.from_numpy(h).cuda() for h in host]
proc = L.CLIPImageProcessorHip(336)
out = {}
for mode, kw in (("pad", {}), ("anyres", {"anyres_grid_pinpoints": "(1x1),...,(3x3)"})):
    for dt in (torch.float32, torch.bfloat16):
        for _ in range(2):
            for d in dev:
                y = L.load_image(d, proc, mode, dtype=dt, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            for d in dev:
                y = L.load_image(d, proc, mode, dtype=dt, **kw)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / (5 * N)
        nbytes = H * W * 3 + y.numel() * y.element_size()
        out[f"{mode}_{str(dt).split('.')[-1]}"] = {"us_per_image": round(dt_s * 1e6, 1), "images_per_s": round(1 / dt_s),
                                                   "patches": y.shape[0], "algorithmic_MB": round(nbytes / 1e6, 3),
                                                   "GBps": round(nbytes / dt_s / 1e9, 1)}
# host baseline: PIL + transformers (what the reference's DataLoader workers run), one core
try:
    from PIL import Image
    from transformers import CLIPImageProcessor
    p = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    bg = tuple(int(x * 255) for x in p.image_mean)
    t0 = time.perf_counter()
    for h in host[:16]:
        im = Image.fromarray(h, "RGB")
        w_, h_ = im.size
        s = max(w_, h_)
        sq = Image.new("RGB", (s, s), bg)
        sq.paste(im, ((s - w_) // 2, (s - h_) // 2))
        ref = p(sq, return_tensors="pt")["pixel_values"]
    cpu_s = (time.perf_counter() - t0) / 16
    out["cpu_pil_pad"] = {"us_per_image": round(cpu_s * 1e6, 1), "images_per_s": round(1 / cpu_s, 1), "cores": 1}
    got = L.load_image(dev[15], proc, "pad")
    out["bit_exact_vs_pil"] = bool(torch.equal(got.cpu(), ref))
except Exception as e:      # PIL / transformers absent on the box: GPU numbers only
    out["cpu_pil_pad"] = f"unavailable: {type(e).__name__}"
print(json.dumps({"image": [H, W], **out}))
