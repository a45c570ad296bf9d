"""bench.py host logic that needs no GPU: `python bench.py --gpus N` with no launcher around it starts its own
ranks as a CHILD process and relays the one JSON line (the driver starts the N = 1 case as plain `python bench.py`;
the N > 1 case must not die in argument handling if it is started the same way)."""
import json
import os
import stat
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(tmp_path, body: str) -> str:
    p = tmp_path / "fake_launcher.py"
    p.write_text("#!" + sys.executable + "\nimport json, os, sys\n" + body)
    p.chmod(p.stat().st_mode | stat.S_IEXEC)
    return str(p)


def _run(tmp_path, body, *args):
    env = dict(os.environ, ZERO_AMD_BENCH_LAUNCHER=f"{sys.executable} {_stub(tmp_path, body)}")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_n_without_launcher_starts_its_own_ranks(tmp_path):
    body = (
        "a = sys.argv[1:]\n"
        "assert '--nnodes=1' in a and '--nproc-per-node=4' in a, a\n"
        "assert a[a.index('--master-addr') + 1] == '127.0.0.1' and int(a[a.index('--master-port') + 1]) > 0\n"
        "i = next(k for k, x in enumerate(a) if x.endswith('bench.py'))\n"
        "assert a[i + 1:] == ['--gpus', '4', '--steps', '3', '--warmup', '1'], a\n"
        "assert 'WORLD_SIZE' not in os.environ\n"
        "print('NCCL version banner that is not the result')\n"
        "print(json.dumps({'metric': 'images/sec', 'value': 1.0, 'n_gpus': 4, 'steps': 3, 'warmup': 1}))\n")
    r = _run(tmp_path, body, "--gpus", "4", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1                                              # ONE line on stdout: the result
    assert json.loads(out[0])["n_gpus"] == 4
    assert "banner" in r.stderr                                       # everything else goes to stderr


def test_self_launch_passes_the_childs_failure_on(tmp_path):
    r = _run(tmp_path, "sys.exit(7)\n", "--gpus", "2")
    assert r.returncode == 7 and r.stdout.strip() == ""
    r = _run(tmp_path, "pass\n", "--gpus", "2")                       # exit 0 without a result line is a failure too
    assert r.returncode == 1


def test_product_synthetic_batch_equals_the_oracles():
    """bench.py draws its batch from zero_amd.synthetic; the parity tests draw theirs from the oracle: same bytes."""
    from oracle import zero_oracle as zo
    from zero_amd.synthetic import synth_batch
    for kw in (dict(), dict(ragged=True), dict(text_only=(1,)), dict(n_pre=3, n_post=9, n_resp=4)):
        enc, dec = zo.EncoderCfg(image=28), zo.DecoderCfg(vocab=128256)
        a = zo.synth_batch(enc, dec, 3, 11, **kw)
        b = synth_batch(28, 128256, 3, 11, **kw)
        assert all(torch.equal(x, y) for x, y in zip(a[0], b[0]))
        assert a[1:] == b[1:]
