#!/usr/bin/env python
"""Headline benchmark: images/sec of the encoder + adapter + surrogate-decoder TRAIN STEP
(forward + backward + gradient all-reduce + grad-norm clip + AdamW) on synthetic 336 px
images plus text tokens — BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W          (N = 1: plain python;
    N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workloads (--workload):
    c3 (default)  Llama-3.2-3B surrogate (18 layers) step, B = 8 per GPU, S = 768   [configs[2]]
    c4            Llama-3.1-8B surrogate (19 layers) step, B = 8 per GPU, S = 768   [configs[3]]
    c2            ViT encoder + adapter forward/backward only, B = 64               [configs[1]]
    c5            zero-shot graft inference on Llama-3.1-70B, tensor parallel over --gpus ranks (RCCL);
                  on ONE GPU: rank 0's shard of a --tp-way split, collectives skipped     [configs[4]]
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0


_RESULT_FD = None


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to file descriptor 1 when
    its communicator comes up (from C, past sys.stdout), so descriptor 1 is pointed at stderr for the
    whole run and the result line is written to a saved duplicate of the real stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_RESULT_FD, line)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--new-tokens", type=int, default=32, help="c5: greedy decode steps that are timed")
    ap.add_argument("--tp", type=int, default=8, help="c5 on one GPU: the tensor-parallel degree whose rank-0 shard is timed")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 8; 64 for c2)")
    ap.add_argument("--stage", default="finetune-encoder", choices=["finetune-encoder", "pretrain"],
                    help="training stage of ref:configs/zero/train_llama.py (pretrain = adapter + translator layer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--n-resp", type=int, default=128,
                    help="labelled (response) tokens per sample, <= 168: the default 128 mimics response-only loss "
                         "(SURVEY.md §8d); 168 labels every text token after the image (label-dense variant)")
    ap.add_argument("--gemm-shapes", action="store_true", help="per-shape GEMM table on stderr")
    ap.add_argument("--gemm-gap", default="", metavar="FILE",
                    help="after the instrumented steps, replay every distinct GEMM call of the step in isolation "
                         "(same operands back to back = cache-hot; operands rotated through > 512 MB = cache-cold) "
                         "and write the in-step / isolated table as JSON to FILE")
    ap.add_argument("--force-dist", action="store_true",
                    help="one-GPU rehearsal of the RCCL path: 1-rank process group, bucket all-reduces still issued")
    return ap.parse_args()


def flops_per_image(enc_layers, ft_after, d, nh, nkv, f, Ld, V, S, T=577, D=1024, F=4096, H=16, n_valid=128,
                    full_lm_head=True, stage="finetune-encoder", n_translators=1):
    """SURVEY.md §8a counting rules: linear 2MKN; attention fwd 4 T^2 H hd (x1/2 causal);
    trainable bwd 2x fwd; frozen bwd 1x (GEMM) / 2x (attention)."""
    enc_layer = 2 * T * (4 * D * D + 2 * D * F) + 4 * T * T * D
    patch = 2 * 576 * 588 * D
    enc_fwd = enc_layers * enc_layer + patch
    enc_bwd = (enc_layers - ft_after) * 2 * enc_layer
    adapter = 2 * T * (D * d + d * d)
    hd = 128
    dec_gemm = 2 * S * (d * (nh + 2 * nkv) * hd + nh * hd * d + 3 * d * f)
    dec_attn = 4 * S * S * nh * hd * 0.5
    rows = S if full_lm_head else n_valid
    lm = 2 * rows * d * V
    if stage == "pretrain":        # no encoder backward; weight gradients of the translator layer(s)
        enc_bwd = 0
    total = enc_fwd + enc_bwd + 3 * adapter + Ld * (2 * dec_gemm + 3 * dec_attn) + 2 * lm
    if not full_lm_head:
        # executed: the last (frozen) layer forms its row-wise half (o-projection, MLP) for the labelled rows only,
        # forward and backward (functional.DecoderLayerFn, tail rows): nothing reads the other rows of its output
        total -= 2 * 2 * (S - rows) * (nh * hd * d + 3 * d * f)
    if stage == "pretrain":
        total += n_translators * dec_gemm
    return total


_TILES = {0: "128,128", 1: "256,128", 2: "256,256", 3: "128,96", 4: "384,192", 5: "192,256"}


def plan_name(plan: int) -> str:
    """zk_gemm_last_plan code -> the kernel symbol prefix rocprofv3 prints."""
    kind, tile, lean, epi = plan // 1000, plan // 100 % 10, plan // 10 % 10, plan % 10
    if kind == 4:
        return f"gemm_kernel<128,128> split-K x{plan - 4000} + splitk_finish_kernel"
    t = _TILES.get(tile, "?")
    if kind == 3:
        return f"gemm_pring_kernel<{t},lean={lean},epi={epi}>"
    if kind == 2:
        return f"gemm_ringx_kernel<{t},lean={lean}>"
    if kind == 1:
        return f"gemm_ring_kernel<{t}>"
    return f"gemm_kernel<{t}>"


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes over this same
    command (profiles/r03_pmc_summary.json, written by tools/pmc_bench.sh + tools/pmc_summary.py:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); None if no summary is present."""
    for name in ("r03_pmc_summary.json", "r02_pmc_summary.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                ent = json.load(f)["kernels"].get(kernel)
            if ent:
                return ent["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            continue
    return None


def gemm_gap_table(lib, calls, shp, path, device):
    """Where the step's GEMM time goes relative to the same launches in isolation.  For every distinct GEMM call of
    the instrumented steps: (a) its average duration INSIDE the step (HIP events around each launch), (b) the same
    call replayed back to back on the same operands (everything it reads stays in L2 / the 256 MB Infinity Cache:
    what tools/bench_gemm.py measures), (c) replayed with A and B rotated through copies that together exceed
    512 MB (every launch reads its operands from HBM, as a weight that was last touched a step ago is read).
    The activations the recorded pointers refer to have been recycled by then: the replays compute on whatever the
    buffers hold now (timing only), and write into output buffers that nothing reads afterwards."""
    rows = []
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def run(a, n_it, rot=None):
        for it in range(3):
            aa = a if rot is None else (rot[it % len(rot)] + a[2:])
            lib.zk_gemm_bf16(*aa)
        e0, e1 = ev(), ev()
        e0.record()
        for it in range(n_it):
            aa = a if rot is None else (rot[it % len(rot)] + a[2:])
            lib.zk_gemm_bf16(*aa)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n_it * 1e3

    for key, a in calls.items():
        plan, (ta, tb), (M, N, K), epi = key
        n_, ms_, fl = shp[key]
        lda, ldb = a[6], a[7]
        na = (K if ta else M) * lda
        nb = (K if tb else N) * ldb
        hot = run(a, 10)
        copies = max(2, min(24, int(512e6 // ((na + nb) * 2)) + 1))
        pool = []
        for _ in range(copies):
            ta_ = (torch.randn(na, device=device) * 0.5).to(torch.bfloat16)
            tb_ = (torch.randn(nb, device=device) * 0.5).to(torch.bfloat16)
            pool.append((ta_, tb_))
        cold = run(a, max(10, copies), rot=[(x.data_ptr(), y.data_ptr()) for x, y in pool])
        del pool
        rows.append({"kernel": plan_name(plan), "transA": ta, "transB": tb, "M": M, "N": N, "K": K, "epilogue": epi,
                     "launches_per_step": n_ // 2, "in_step_us": round(ms_ / n_ * 1e3, 1), "isolated_hot_us": round(hot, 1),
                     "isolated_cold_us": round(cold, 1), "gflop": round(fl / 1e9, 1),
                     "in_step_ms_per_step": round(ms_ / 2, 3),
                     "ms_per_step_if_hot": round(hot * (n_ // 2) / 1e3, 3),
                     "ms_per_step_if_cold": round(cold * (n_ // 2) / 1e3, 3)})
    rows.sort(key=lambda r: -r["in_step_ms_per_step"])
    tot = {k: round(sum(r[k] for r in rows), 2) for k in ("in_step_ms_per_step", "ms_per_step_if_hot", "ms_per_step_if_cold")}
    with open(path, "w") as f:
        json.dump({"total": tot, "rows": rows}, f, indent=1)
    print(f"GEMM ms/step: in the step {tot['in_step_ms_per_step']}, same launches isolated cache-hot "
          f"{tot['ms_per_step_if_hot']}, isolated cache-cold {tot['ms_per_step_if_cold']}", file=sys.stderr)
    for r in rows[:24]:
        print(f"  {r['kernel']:44s} tA{r['transA']}tB{r['transB']} epi{r['epilogue']} {r['M']:6d} {r['N']:6d} {r['K']:6d} "
              f"x{r['launches_per_step']:3d}  in-step {r['in_step_us']:7.1f} us  hot {r['isolated_hot_us']:7.1f}  cold "
              f"{r['isolated_cold_us']:7.1f}", file=sys.stderr)


def self_launch(n: int, argv=None) -> int:
    """`python bench.py --gpus N` without a launcher around it: run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <the same arguments>` as a child, pass its stderr through, print the one JSON line rank 0 wrote, return its exit
    code.  ZERO_AMD_BENCH_LAUNCHER (a command prefix that replaces `python -m torch.distributed.run`) exists for the
    CPU test of this path."""
    import shlex
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launcher = os.environ.get("ZERO_AMD_BENCH_LAUNCHER")
    head = shlex.split(launcher) if launcher else [sys.executable, "-m", "torch.distributed.run"]
    cmd = head + ["--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                  os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.lstrip().startswith("{")]
    for ln in proc.stdout.splitlines():
        if ln not in lines[-1:]:
            print(ln, file=sys.stderr)                  # anything else a rank printed is not the result
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the launcher exited 0 without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    args = parse()
    if os.environ.get("ZERO_AMD_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ZERO_AMD_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the N = 1 case is (`python bench.py --gpus N`): start the ranks ourselves.  Nothing has
        # touched the GPU yet, and the launcher is a CHILD process (never an exec): its one JSON line is relayed.
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_dist:
        _claim_stdout()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.force_dist:   # rehearsal of the RCCL path on one GPU: a 1-rank group, collectives still issued
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
            os.environ["ZERO_AMD_FORCE_ALLREDUCE"] = "1"
        dist.init_process_group("nccl", device_id=device)

    if args.workload == "c5":
        return bench_c5(args, world, rank, device)

    from zero_amd.synthetic import synth_batch
    from zero_amd import ops
    from zero_amd.build_decoder import load_hf_llama3
    from zero_amd.build_encoder import load_clip
    from zero_amd.configs.zero import train_llama as run_cfg
    from zero_amd.methods.engine_ar_instruction_llama import forward_from_ids
    from zero_amd.models.adapter import Zero
    from zero_amd.train import Trainer

    import copy
    cfg = copy.deepcopy(run_cfg.config)
    cfg.device = device
    if args.workload == "c4":
        cfg.llama_model_version, cfg.llama_model_scale = "3.1", "8"
        from zero_amd.configs.config_base import LLAMA_SHAPES
        cfg.translators_range = LLAMA_SHAPES[("3.1", "8")]["translators_range"]
    B = args.batch or (64 if args.workload == "c2" else 8)
    cfg.batch_size = B
    cfg.training_stage = args.stage
    if os.environ.get("ZERO_AMD_NAN_CHECK"):      # A/B hook: "sync" = the reference's blocking check
        cfg.check_nan_loss = os.environ["ZERO_AMD_NAN_CHECK"]

    # ---- model: identical seeded random init on every rank (DDP broadcasts rank 0's weights)
    torch.manual_seed(1337)
    quiet = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, quiet
    try:
        encoder, _, img_size = load_clip(cfg)
        if args.workload == "c2":
            from zero_amd.configs.config_base import LLAMA_SHAPES
            num_embd, decoder = LLAMA_SHAPES[("3.2", "3")]["hidden_size"], None
        else:
            num_embd, decoder, _, _, _ = load_hf_llama3(cfg)
    finally:
        sys.stdout = stdout
    cfg.num_embd = num_embd
    model = Zero(cfg, encoder, decoder)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
    groups = run_cfg.func_optim_filter(cfg, model)
    model.train()

    # ---- synthetic data: seeded per rank
    g = torch.Generator(device="cpu").manual_seed(1337 + rank)
    n_pre, n_post, n_resp = 21, 168, min(args.n_resp, 168)
    T = (img_size // 14) ** 2 + 1
    imgs_cpu, ids, tgts, img_id, pad_id = synth_batch(img_size, 128256, B, 1337 + rank, n_pre=n_pre, n_post=n_post,
                                                      n_resp=n_resp)
    imgs = [i.to(device).to(torch.bfloat16) for i in imgs_cpu]
    S = n_pre + (T + 2) + n_post

    if args.workload == "c2":
        def c2_forward(self, cfg_, imgs_):
            rows, _, _ = self._encode_rows(imgs_)
            # a scalar objective that reaches every trainable tensor: mean of the adapter output
            return _MeanFn.apply(rows)
        trainer = Trainer(cfg, model, c2_forward, groups)
        step = lambda: trainer.micro_step(imgs)
    else:
        trainer = Trainer(cfg, model, forward_from_ids, groups)
        step = lambda: trainer.micro_step(imgs, ids, tgts, img_id, pad_id)

    def barrier():
        if world > 1 or args.force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if rank == 0 and world == 1 and os.environ.get("ZERO_AMD_BENCH_CPROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        for _ in range(3):
            step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(28)
    if rank == 0 and world == 1 and os.environ.get("ZERO_AMD_BENCH_TORCHPROF"):
        # which Python lines launch the small torch kernels (fills, copies, cats) of a step
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            step()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=60,
                                                          max_name_column_width=60, max_src_column_width=110),
              file=sys.stderr)
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - h0
    barrier()
    dt = time.perf_counter() - t0
    if rank == 0 and os.environ.get("ZERO_AMD_BENCH_VERBOSE"):
        # host-side enqueue time per step (the step never synchronises; if this approaches ms_per_step
        # the GPU is waiting for Python)
        print(f"host enqueue {host_s / args.steps * 1e3:.2f} ms/step", file=sys.stderr)
    if world > 1 or args.force_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = dt / args.steps * 1e3
    images_per_s = B * world * args.steps / dt

    # ---- host cost of enqueuing one step, measured with an EMPTY device queue (the in-loop figure above also
    # counts the time Python spends blocked on a full launch queue, i.e. it approaches the GPU time of a step
    # whenever the GPU is the bottleneck).  Outside the timed region; every rank takes the same steps.
    host_enq = []
    for _ in range(3):
        barrier()
        h0 = time.perf_counter()
        step()
        host_enq.append((time.perf_counter() - h0) * 1e3)
    barrier()
    host_enqueue_ms = sorted(host_enq)[1]

    # ---- per-kernel timing of the GEMMs, live, with HIP events on the launch stream.  Every
    # zk_gemm_bf16 call (plain and fused) is bracketed by two events and attributed to the kernel
    # variant the library launched (zk_gemm_last_plan), so each group below is ONE kernel symbol of
    # the rocprofv3 table; the roofline line is the symbol with the largest total time.
    roofline = None
    if not args.no_kernel_timing:
        # EVERY rank takes the two extra steps (they contain the step's collectives: a rank stepping
        # alone would wait for its peers forever); only rank 0 instruments its launches.
        records = []
        calls = {}          # first call of every distinct (kernel, layout, shape, epilogue): replayed by --gemm-gap

        class _TimedLib:
            def __init__(self, real):
                self._real = real

            def __getattr__(self, name):
                return getattr(self._real, name)

            def zk_gemm_bf16(self, A, Bp, C, M, N, K, lda, ldb, ldc, ta, tb, epi, *rest):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = self._real.zk_gemm_bf16(A, Bp, C, M, N, K, lda, ldb, ldc, ta, tb, epi, *rest)
                e1.record()
                plan = self._real.zk_gemm_last_plan()
                records.append((plan, (ta, tb), 2.0 * M * N * K, e0, e1, (M, N, K), epi))
                calls.setdefault((plan, (ta, tb), (M, N, K), epi), (A, Bp, C, M, N, K, lda, ldb, ldc, ta, tb, epi) + rest)
                return rc

        # the bandwidth-bound kernels of the same two steps, each launch bracketed the same way; algorithmic
        # bytes per launch from its arguments (read + write of every operand row once)
        bw_records = []

        def _bw(name, nbytes_of):
            def call(self, *a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = getattr(self._real, name)(*a)
                e1.record()
                bw_records.append((name, float(nbytes_of(a)), e0, e1))
                return rc
            setattr(_TimedLib, name, call)

        _bw("zk_rmsnorm_fwd", lambda a: a[4] * a[5] * 4)
        _bw("zk_rmsnorm_bwd", lambda a: a[6] * a[7] * (8 if a[4] else 6))
        _bw("zk_layernorm_fwd", lambda a: a[6] * a[7] * 4)
        _bw("zk_layernorm_bwd", lambda a: a[10] * a[11] * ((6 if a[6] else 4) + (2 if a[5] else 0)))
        _bw("zk_rope", lambda a: a[3] * a[5] * a[6] * 4)
        _bw("zk_cross_entropy", lambda a: a[5] * a[6] * (4 if a[4] else 2))
        _bw("zk_adamw_step", lambda a: a[4] * 22)

        real_lib = ops.lib
        if rank == 0:
            ops.lib = _TimedLib(real_lib)
        try:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        finally:
            ops.lib = real_lib
        barrier()
    if rank == 0 and not args.no_kernel_timing:
        fam, shp = {}, {}
        for plan, tt, fl, e0, e1, mnk, epi in records:
            ms = e0.elapsed_time(e1)
            d = fam.setdefault(plan, dict(n=0, ms=0.0, flops=0.0))
            d["n"] += 1; d["ms"] += ms; d["flops"] += fl
            d_ = shp.setdefault((plan, tt, mnk, epi), [0, 0.0, fl])
            d_[0] += 1; d_[1] += ms
        if args.gemm_shapes:
            for (plan, tt, mnk, epi), (n_, ms_, fl) in sorted(shp.items(), key=lambda kv: -kv[1][1]):
                print(f"  {plan_name(plan):46s} tA{tt[0]}tB{tt[1]} epi{epi} M,N,K={mnk}  x{n_ // 2:3d}/step  "
                      f"{ms_ / 2:7.2f} ms/step  {fl * n_ / (ms_ * 1e-3) / 1e12:7.1f} TF/s", file=sys.stderr)
        if args.gemm_gap:
            gemm_gap_table(real_lib, calls, shp, args.gemm_gap, device)
        key = max(fam, key=lambda k: fam[k]["ms"])
        d = fam[key]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        all_ms = sum(v["ms"] for v in fam.values())
        all_fl = sum(v["flops"] for v in fam.values())
        roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(plan_name(key)),
                    "kernel": plan_name(key),
                    "launches_per_step": d["n"] // 2, "avg_launch_us": round(d["ms"] / d["n"] * 1e3, 1),
                    "avg_gflop_per_launch": round(d["flops"] / d["n"] / 1e9, 1),
                    "gemm_ms_per_step": round(all_ms / 2, 2),
                    "all_gemm_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 1),
                    "kernels": {plan_name(k): {"n": v["n"] // 2, "ms": round(v["ms"] / 2, 2),
                                               "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}}
        bwf = {}
        for name, nb, e0, e1 in bw_records:
            d_ = bwf.setdefault(name, dict(n=0, ms=0.0, bytes=0.0))
            d_["n"] += 1; d_["ms"] += e0.elapsed_time(e1); d_["bytes"] += nb
        # achieved = algorithmic bytes / launch time INSIDE the step (operands are often still in the 256 MB
        # Infinity Cache there; the cold-operand HBM rates are profiles/r01_bandwidth_kernels.json)
        roofline["hbm_kernels"] = {k[3:]: {"n": v["n"] // 2, "ms": round(v["ms"] / 2, 3),
                                           "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9),
                                           "frac_of_8TBps": round(v["bytes"] / (v["ms"] * 1e-3) / 8e12, 3)}
                                   for k, v in sorted(bwf.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}

    # ---- CPU baseline: the oracle on the host cores, rank 0, N = 1 only, bounded sample
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != "c2":
        from oracle import zero_oracle as zo   # the checker, timed as the CPU baseline: this leg only
        ecfg = zo.EncoderCfg()
        cores = os.cpu_count() or 1
        cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
        torch.set_num_threads(cores)
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        if "decoder.lm_head.weight" in sd and model.decoder.config.tie_word_embeddings:
            del sd["decoder.lm_head.weight"]
        dc = model.decoder.config
        odec = zo.DecoderCfg(hidden=dc.hidden_size, heads=dc.num_attention_heads, kv_heads=dc.num_key_value_heads,
                             head_dim=dc.head_dim, intermediate=dc.intermediate_size, layers=dc.num_hidden_layers,
                             vocab=dc.vocab_size, rope_factor=dc.rope_scaling["factor"], tie=dc.tie_word_embeddings)
        nb = min(2, B)      # bounded sample: two images of the batch are 10-12 s of host work
        t0 = time.perf_counter()
        r = zo.train_step(sd, ecfg, odec, [i.to(torch.bfloat16).float() for i in imgs_cpu[:nb]], ids[:nb], tgts[:nb],
                          img_id=img_id, pad_id=pad_id, encoder_ft_layers_after=cfg.encoder_ft_layers_after)
        cdt = time.perf_counter() - t0
        # full-size parity spot check: the HIP path's loss on the SAME images with the SAME (current) weights
        with torch.no_grad():
            gl = float(forward_from_ids.__get__(model, model.__class__)(cfg, imgs[:nb], ids[:nb], tgts[:nb], img_id, pad_id))
        cpu_baseline = {"value": round(nb / cdt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
                        "sample": f"{nb} images (S={S}) of the same workload: one fp32 forward+backward of the oracle "
                                  f"(oracle/zero_oracle.py) in {cdt:.1f} s, optimizer excluded",
                        "loss": round(float(r.loss), 4), "hip_loss_same_image": round(gl, 4),
                        "loss_abs_diff": round(abs(gl - float(r.loss)), 5)}

    if rank == 0:
        dc = model.decoder.config if decoder is not None else None
        if dc is not None:
            Ld = len(model.decoder.model.layers)
            fl_alg = flops_per_image(24, cfg.encoder_ft_layers_after, dc.hidden_size, dc.num_attention_heads,
                                     dc.num_key_value_heads, dc.intermediate_size, Ld, dc.vocab_size, S,
                                     stage=args.stage)
            fl_exec = flops_per_image(24, cfg.encoder_ft_layers_after, dc.hidden_size, dc.num_attention_heads,
                                      dc.num_key_value_heads, dc.intermediate_size, Ld, dc.vocab_size, S,
                                      n_valid=n_resp, full_lm_head=False, stage=args.stage)
        else:
            fl_alg = fl_exec = 0.680e12
        per_gpu = images_per_s / world
        out = {
            "metric": "images/sec (node) for encoder+surrogate train step",
            "value": round(images_per_s, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": {"c3": ("Llama-3.2-3B surrogate (18 layers) encoder-training step, encoder "
                                            "layers 16-23 + adapter trainable, surrogate frozen"
                                            if args.stage == "finetune-encoder" else
                                            "Llama-3.2-3B surrogate (18 layers) stage-1 pretrain step: adapter + "
                                            "post_layernorm + translator layer 16 trainable, encoder frozen"),
                                    "c4": "Llama-3.1-8B surrogate (19 layers) encoder-training step",
                                    "c2": "ViT-L/14-336 encoder + adapter forward/backward"}[args.workload],
                       "per_gpu_batch": B, "global_batch": B * world, "seq_len": S, "image": "336x336",
                       "parallelism": f"dp{world}", "grad_accum": 1, "weights": "seeded random init",
                       "stage": args.stage,
                       # exact dead-row elimination, counted in tflop_per_image_executed
                       "rows_computed": ("lm_head and the last decoder layer's row-wise half on the labelled rows only "
                                         "(loss and gradients of the full computation)"
                                         if args.workload != "c2" and os.environ.get("ZERO_AMD_TAIL_ROWS", "1") != "0"
                                         else "lm_head on the labelled rows only" if args.workload != "c2" else "all")},
            "loss": round(float(loss), 4),
            "tflop_per_image_algorithmic": round(fl_alg / 1e12, 3),
            "tflop_per_image_executed": round(fl_exec / 1e12, 3),
            "mfma_util_step": round(per_gpu * fl_exec / 1e12 / PEAK_BF16_TFLOPS, 4),
            "host_enqueue_ms_per_step": round(host_enqueue_ms, 2),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        _emit(out)
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


def bench_c5(args, world, rank, device):
    """Graft inference (ref:inference/vlm_infer.py:191-270) on the FULL Llama-3.1-70B: encoder + adapter
    prefix, prefill of the 768-token prompt, then greedy decoding with a KV cache.  Each rank holds 1/W
    of every layer (distributed/tensor_parallel.py); with one GPU the rank-0 shard of a --tp split is
    built directly (same shapes, random init) and the all-reduces are skipped (compute-only number)."""
    from zero_amd.build_encoder import load_clip
    from zero_amd.configs.config_base import LLAMA_SHAPES
    from zero_amd.configs.zero import train_llama as run_cfg
    from zero_amd.distributed.tensor_parallel import TPContext
    from zero_amd.inference.decode import KVCache, forward_cached
    from zero_amd.inference.vlm_infer import graft_inputs
    from zero_amd.models.adapter import Zero
    from zero_amd.models.llama import LlamaConfig, LlamaForCausalLM
    import copy
    W = world if world > 1 else args.tp
    sh = dict(LLAMA_SHAPES[("3.1", "70")])
    sh.pop("translators_range")
    factor = sh.pop("rope_factor")
    full = LlamaConfig(**sh)
    lcfg = LlamaConfig(**{**sh, "num_attention_heads": full.num_attention_heads // W,
                          "num_key_value_heads": full.num_key_value_heads // W,
                          "intermediate_size": full.intermediate_size // W})
    lcfg.rope_scaling = dict(rope_type="llama3", factor=factor, low_freq_factor=1.0, high_freq_factor=4.0,
                             original_max_position_embeddings=8192)
    cfg = copy.deepcopy(run_cfg.config)
    cfg.device, cfg.enable_translator, cfg.num_embd = device, False, lcfg.hidden_size
    torch.manual_seed(1337 + rank)                 # each rank draws its own shard
    quiet = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, quiet
    try:
        encoder, _, img_size = load_clip(cfg)
        decoder = LlamaForCausalLM(lcfg, device=device)          # shard-shaped layers, replicated embed / lm_head
    finally:
        sys.stdout = stdout
    torch.manual_seed(1337)                        # replicated tensors must agree across ranks
    with torch.no_grad():
        decoder.model.embed_tokens.weight.normal_(0.0, 0.02)
        decoder.lm_head.weight.normal_(0.0, 0.02)
    model = Zero(cfg, encoder, decoder)
    model.eval()
    tp = TPContext(rank, world, None) if world > 1 else TPContext()
    from zero_amd.synthetic import synth_batch
    imgs_cpu, ids, _, img_id, _ = synth_batch(336, 128256, 1, 1337, n_pre=21, n_post=168, n_resp=1)
    imgs = [i.to(device).to(torch.bfloat16) for i in imgs_cpu]
    nl, kvw = len(decoder.model.layers), lcfg.num_key_value_heads * lcfg.head_dim
    wte = decoder.model.get_input_embeddings()

    def one_pass(n_new):
        x = graft_inputs(model, imgs, ids, img_id)
        cache = KVCache(nl, 1, x.shape[1] + n_new, kvw, device)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
        t0.record()
        logits = forward_cached(decoder, x, cache, tp)
        t1.record()
        for _ in range(n_new):
            nxt = logits[:, 0].float().argmax(-1)
            logits = forward_cached(decoder, wte(nxt.view(1, 1)), cache, tp)
        t2.record()
        torch.cuda.synchronize()
        return x.shape[1], t0.elapsed_time(t1), t1.elapsed_time(t2)

    for _ in range(max(1, args.warmup // 2)):
        one_pass(4)
    if world > 1:
        dist.barrier()
    pre, decs = [], []
    for _ in range(max(1, args.steps // 3)):
        S, a, b = one_pass(args.new_tokens)
        pre.append(a); decs.append(b)
    pre_ms, dec_ms = sorted(pre)[len(pre) // 2], sorted(decs)[len(decs) // 2]
    if world > 1:
        t = torch.tensor([pre_ms, dec_ms], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pre_ms, dec_ms = t.tolist()
    if rank == 0:
        per_tok = 2.0 * sum(p.numel() for n, p in decoder.named_parameters() if "embed_tokens" not in n)
        _emit(({
            "metric": "zero-shot graft inference on Llama-3.1-70B: decode tokens/sec (one sequence)",
            "value": round(args.new_tokens / (dec_ms * 1e-3), 2), "unit": "tokens/sec", "n_gpus": world,
            "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
            "prefill_ms": round(pre_ms, 2), "prefill_tokens": S, "decode_ms_per_token": round(dec_ms / args.new_tokens, 3),
            "config": {"workload": "encoder + adapter prefix, 768-token prompt prefill, greedy KV-cache decoding on "
                                   "Llama-3.1-70B (80 layers), tensor parallel", "tp_degree": W,
                       "collectives": "RCCL all-reduce, 2 per layer" if world > 1 else
                                      "skipped: ONE rank's shard timed on one GPU (compute only)",
                       "weights": "seeded random init", "new_tokens": args.new_tokens},
            "decode_weight_gb_per_token_per_gpu": round(per_tok / 1e9, 2),
            "decode_weight_stream_gbs": round(per_tok / 1e9 / (dec_ms * 1e-3 / args.new_tokens), 1)}))
    if world > 1:
        dist.destroy_process_group()


class _MeanFn(torch.autograd.Function):
    """c2 objective: mean over all adapter outputs (bf16 rows) via the column-sum kernel."""

    @staticmethod
    def forward(ctx, rows):
        from zero_amd import ops
        acc = torch.empty((ops.colsum_parts(rows.shape[0]), rows.shape[1]), dtype=torch.float32, device=rows.device)
        ops.colsum_(rows, acc)             # per-part partial sums (deterministic), summed below
        ctx.shape = rows.shape
        return acc.sum() / rows.numel()

    @staticmethod
    def backward(ctx, g):
        n = ctx.shape[0] * ctx.shape[1]
        return (g / n).to(torch.bfloat16).expand(ctx.shape).contiguous()


if __name__ == "__main__":
    main()
