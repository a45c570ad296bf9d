"""In-process A/B of GEMM tuning hooks on the full C3 train step (interleaved rounds, one device:
boxes differ by several percent, so builds are never ranked across runs).
    python tools/ab_step.py "name=cfg,cfg;name2=cfg,..."      cfg = zk_gemm_set_tile codes"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import zero_oracle as zo
from zero_amd import ops
from zero_amd.build_decoder import load_hf_llama3
from zero_amd.build_encoder import load_clip
from zero_amd.configs.zero import train_llama as run_cfg
from zero_amd.methods.engine_ar_instruction_llama import forward_from_ids
from zero_amd.models.adapter import Zero
from zero_amd.train import Trainer

spec = sys.argv[1] if len(sys.argv) > 1 else "base=99999;nostagger=100000"
variants = [(v.split("=")[0], [int(c) for c in v.split("=")[1].split(",")]) for v in spec.split(";")]
env_variants = os.environ.get("AB_ENV")          # optional "NAME=VAL" toggled per variant index (unused by default)
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
cfg = copy.deepcopy(run_cfg.config); cfg.device = device; cfg.batch_size = 8
torch.manual_seed(1337)
so, sys.stdout = sys.stdout, open(os.devnull, "w")
try:
    encoder, _, img_size = load_clip(cfg)
    num_embd, decoder, _, _, _ = load_hf_llama3(cfg)
finally:
    sys.stdout = so
cfg.num_embd = num_embd
model = Zero(cfg, encoder, decoder)
with torch.no_grad():
    for n, p in model.named_parameters():
        if "norm" in n and n.endswith("weight"):
            p.fill_(1.0)
groups = run_cfg.func_optim_filter(cfg, model)
model.train()
imgs_cpu, ids, tgts, img_id, pad_id = zo.synth_batch(zo.EncoderCfg(), zo.DecoderCfg(vocab=128256), 8, 1337, n_pre=21, n_post=168, n_resp=128)
imgs = [i.to(device).to(torch.bfloat16) for i in imgs_cpu]
trainer = Trainer(cfg, model, forward_from_ids, groups)
step = lambda: trainer.micro_step(imgs, ids, tgts, img_id, pad_id)
for _ in range(3):
    step()
torch.cuda.synchronize()
res = {n: [] for n, _ in variants}
for rnd in range(5):
    for name, codes in variants:
        for c in codes:
            ops.set_gemm_tile(c)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 6 * 1e3)
for name, v in res.items():
    v = sorted(v)
    print(f"{name:16s} median {v[len(v) // 2]:7.2f} ms  min {v[0]:7.2f} ms   ({8e3 / v[len(v) // 2]:.1f} img/s)", flush=True)
