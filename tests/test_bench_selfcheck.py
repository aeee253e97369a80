"""``bench.roofline_self_check``: the line's own plausibility test for its roofline figures (review r04, row d).

Round 4's driver-run line said ``kernel_ms_per_batch`` 17.07 inside an 18.6-ms batch (``frac`` 0.255 where the rocprofv3 summary
said 0.488): the clock sampler's stream had been serialised in front of the traced launch.  The checks must flag exactly that
line and pass this round's."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                                 # (defines functions only: main() runs under __main__)
    return mod


def test_round4_driver_line_is_flagged():
    b = _bench()
    # BENCH_r04: block sums 17.07 ms "in" an 18.6-ms batch; the 16-class launch 16.385 ms against 6.9 ms back to back at 2 397 MHz
    chk = b.roofline_self_check(17.07, 5.6, [16.385], [18.5, 18.6, 18.7, 18.6, 18.9], 6.9, 2397.0)
    assert chk["fits_in_batch"] is False and chk["class_launch_plausible"] is False and chk["ok"] is False
    # either symptom alone is enough
    assert b.roofline_self_check(17.07, 5.6, [], [18.6], None, None)["ok"] is False
    assert b.roofline_self_check(8.9, 5.6, [16.4], [18.6], 6.9, 2397.0)["ok"] is False


def test_round5_lines_pass():
    b = _bench()
    chk = b.roofline_self_check(8.977, 5.38, [8.297], [18.3, 18.4, 18.2, 18.3, 18.9], 7.45, 2010.0)
    assert chk["ok"] and chk["fits_in_batch"] and chk["class_launch_plausible"]
    assert chk["blocksum_plus_chain_ms"] == round(8.977 + 5.38, 3)
    # no clock sample (sampler discarded or several ranks): the in-situ clock of every run so far stands in
    assert b.roofline_self_check(8.977, 5.38, [8.297], [18.3], 7.45, None)["ok"]
    # nothing measured (--no-roofline-batch): nothing to object to
    assert b.roofline_self_check(0.0, 0.0, [], [], None, None)["ok"]


def test_replayed_counters_are_bound_to_the_kernel_sources(tmp_path):
    """``roofline.traffic`` & co. come from a committed counter summary; they are only printed while the kernel sources + flags
    the summary was taken on hash to what the tree holds (review r05, item 6)."""
    import json

    from basq_amd._build import source_hash

    b = _bench()
    h = source_hash()
    assert len(h) == 64 and h == source_hash()
    (tmp_path / "r06_a_pmc.json").write_text(json.dumps({"hbm_bytes_per_batch": 1, "kernel_source_sha256": "0" * 64}))
    (tmp_path / "r07_z_pmc.json").write_text(json.dumps({"hbm_bytes_per_batch": 2}))              # no hash at all: stale
    rec, name, stale = b.newest_counters(str(tmp_path))
    assert name == "r07_z_pmc.json" and stale is True
    (tmp_path / "r08_a_pmc.json").write_text(json.dumps({"hbm_bytes_per_batch": 3, "kernel_source_sha256": h}))
    rec, name, stale = b.newest_counters(str(tmp_path))
    assert name == "r08_a_pmc.json" and stale is False and rec["hbm_bytes_per_batch"] == 3
    assert b.newest_counters(str(tmp_path), tree_hash="f" * 64)[2] is True
    assert b.newest_counters(str(tmp_path / "nothing_here")) == (None, None, None)


def test_source_hash_follows_the_sources(tmp_path, monkeypatch):
    from basq_amd import _build

    h0 = _build.source_hash()
    monkeypatch.setattr(_build, "FLAGS", _build.FLAGS + ["-DX=1"])
    assert _build.source_hash() != h0


def test_bench_without_launcher_starts_its_ranks_itself():
    """``python bench.py --gpus N`` with no ``RANK`` in the environment is its own launcher (review r05, item 2a).  On this box there
    is no GPU: the parent must find that out by COUNTING devices (no HIP call) and say so, instead of asking for torchrun."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box: this is the CPU-side check")
    assert r.returncode != 0 and "torch.distributed.run" not in r.stderr.split("this node shows")[0][-300:]
    assert "this node shows" in r.stderr and "must be launched" not in r.stderr


def test_self_launch_relays_the_json_line_last(tmp_path, monkeypatch):
    """The launcher half on its own: the ranks' output is passed on, the JSON line is held back and printed LAST, the exit code is
    the launcher's; a launcher that ends without a line is never a success."""
    import io
    import subprocess
    import sys
    from contextlib import redirect_stdout

    b = _bench()
    fake = tmp_path / "fake_run.py"

    class FakeTorchCuda:
        @staticmethod
        def device_count():
            return 4

    real = subprocess.Popen

    def popen_factory(script_body, rc):
        def popen(cmd, **kw):
            assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
            assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
            assert kw["env"]["MASTER_ADDR"] == "127.0.0.1" and kw["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
            fake.write_text(script_body + f"\nimport sys; sys.exit({rc})\n")
            return real([sys.executable, str(fake)], stdout=subprocess.PIPE, text=True)

        return popen

    monkeypatch.setattr(b.torch, "cuda", FakeTorchCuda)
    monkeypatch.setattr(subprocess, "Popen", popen_factory("print('NCCL version banner'); print('{\"metric\": \"m\", \"value\": 1}'); "
                                                           "print('late rank chatter')", 0))
    out = io.StringIO()
    with redirect_stdout(out):
        rc = b.self_launch(4)
    lines = out.getvalue().strip().splitlines()
    assert rc == 0 and lines[-1].startswith('{"metric"') and "late rank chatter" in lines[:-1]
    monkeypatch.setattr(subprocess, "Popen", popen_factory("print('no line at all')", 0))
    with redirect_stdout(io.StringIO()):
        assert b.self_launch(4) != 0
    monkeypatch.setattr(subprocess, "Popen", popen_factory("print('{\"metric\": \"m\"}')", 3))
    with redirect_stdout(io.StringIO()):
        assert b.self_launch(4) == 3                             # the watchdog's code comes through


def test_watchdog_exit_code_is_not_success():
    b = _bench()
    assert b.WATCHDOG_EXIT_CODE not in (0, None)
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os._exit(0)" not in src and "os._exit(WATCHDOG_EXIT_CODE)" in src
