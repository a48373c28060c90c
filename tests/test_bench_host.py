"""Host-side logic of bench.py that needs no GPU: the self-launch of `--gpus N` (the driver runs `python bench.py
--gpus N` exactly as it runs `--gpus 1`), the steady-state block timing, and the guard that keeps counter-derived figures
out of the line when the kernels on disk are not the ones the committed counter pass was taken on."""
import json
import os
import subprocess
import sys
import threading
import time

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_the_drivers_own():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    # a free port is picked when none is given
    p = int(bench.launch_command(2, [])[bench.launch_command(2, []).index("--master-port") + 1])
    assert 1024 < p < 65536


def test_gpus_2_without_a_launcher_starts_its_own_ranks_or_says_why():
    """no GPU in this container: the self-launch path must be the one taken (not the old 'must be launched under
    torch.distributed.run' exit) and must fail with a message about GPUs, before any rank is started"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() == 0:
        assert r.returncode == 2 and "no GPU visible" in r.stderr and "torch.distributed.run" not in r.stderr
        assert r.stdout.strip() == ""


def test_block_count_and_block_times():
    class A:
        blocks = 0
        steps = 20
    assert bench.n_blocks(A) == 75                      # 75 x 20 = 1500: the steady window does not depend on --steps
    A.steps = 400
    assert bench.n_blocks(A) == 4
    A.steps = 5000
    assert bench.n_blocks(A) == 3
    A.blocks = 4
    assert bench.n_blocks(A) == 4
    # completions one per millisecond from t = 1 ms: every block of 10 lasts 10 ms whatever the warm-up
    done = [0.001 * (i + 1) for i in range(100)]
    bt = bench.block_times(done, 7, 10, 5, 0.0)
    assert all(abs(x - 0.010) < 1e-12 for x in bt) and len(bt) == 5
    # without warm-up the first block starts at t_start
    assert abs(bench.block_times(done, 0, 10, 1, 0.0)[0] - 0.010) < 1e-12


def test_gpus_are_counted_without_touching_hip(monkeypatch):
    """GPU_MAX_HW_QUEUES has to be in the environment before the HIP runtime starts, so the ranks-per-GPU arithmetic may
    not go through a torch device call"""
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus_without_hip() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3")
    assert bench.visible_gpus_without_hip() == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpus_without_hip() >= 0            # the KFD topology, 0 when there is none (this container)
    src = open(os.path.join(ROOT, "bench.py")).read()
    main_src = src[src.index("def main():"):]
    assert main_src.index('setdefault("GPU_MAX_HW_QUEUES"') < main_src.index("import torch\n    import numpy")


def test_no_rccl_collective_outside_the_sharded_leg():
    """the control plane is gloo: the only place bench.py may reach RCCL is open_data_group inside the sharded leg"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'init_process_group("gloo")' in src and 'init_process_group("nccl"' not in src
    assert src.count("open_data_group(") == 1
    leg = src[src.index("# ---- N > 1: proofs sharded over the ranks"):src.index("# ---- N = 1 diagnostic")]
    assert "open_data_group(" in leg
    watchdog_at = src.index("watchdog = threading.Timer")
    assert watchdog_at < src.index("open_data_group(dev")    # and it is under the watchdog


def test_steady_stream_keeps_n_in_flight_and_surfaces_errors():
    live, peak, lock = [0], [0], threading.Lock()

    def job(k):
        with lock:
            live[0] += 1
            peak[0] = max(peak[0], live[0])
        time.sleep(0.002)
        with lock:
            live[0] -= 1
    done = bench.steady_stream(job, 40, 4)
    assert len(done) == 40 and done == sorted(done) and peak[0] == 4

    def bad(k):
        if k == 5:
            raise RuntimeError("boom")
    with pytest.raises(RuntimeError):
        bench.steady_stream(bad, 20, 3)


def test_counter_pass_is_tied_to_the_kernel_sources():
    fp = bench.source_fingerprint()
    assert len(fp) == 16 and fp == bench.source_fingerprint()
    entry, current = bench.committed_counters("rs256-sd/gates/bits=0.90")
    if entry is not None:
        assert current == (entry.get("csrc_sha16") == fp)
    assert bench.committed_counters("no/such/workload") == (None, False)
    # the fingerprint moves with any kernel source byte
    import importlib.util
    spec = importlib.util.spec_from_file_location("cg_build_t", os.path.join(ROOT, "crescent-credentials_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old_flags = list(mod.EXTRA_FLAGS)
    try:
        mod.EXTRA_FLAGS.append("-DX")
        assert mod.source_fingerprint() != fp
    finally:
        mod.EXTRA_FLAGS[:] = old_flags
    assert mod.source_fingerprint() == fp


def test_content_digest_memo_holds_its_arrays_and_is_not_kept_for_views_of_writable_buffers():
    """the digest memo kept (id, address, size) of the arrays, which CPython and malloc recycle - a replaced
    array of the same size could be taken for the old one.  It now holds the arrays and compares by identity.  ADVICE r5
    (low): the hashed arrays are frozen, their parents are NOT (round 5 froze them too and changed unrelated views of the
    caller's buffer); a set that contains a view of a still-writable buffer is not memoised - hashed again at every use."""
    import numpy as np
    from crescent_credentials_amd import api
    base = np.arange(64 * 4, dtype=np.uint8)
    view = base[:64]
    held = api._freeze(view, b"abc")
    assert held.same_as((view, held.arrays[1])) and not held.same_as((base[:64], held.arrays[1]))     # another view object: not "the same"
    assert not view.flags.writeable and base.flags.writeable and not held.stable
    import pytest
    with pytest.raises(ValueError):
        view[0] = 1
    base[0] = 1                                              # the caller's buffer stays the caller's
    own = np.arange(64, dtype=np.uint8)
    assert api._freeze(own).stable and not own.flags.writeable
    # a replacing array that lands on the old one's address cannot be confused with it: the memo keeps the old one alive
    rows = [[(1, 0)], [(1, 1)]]
    cm = api.ConstraintMatrices.from_rows(rows, rows, rows, 1, 2)
    k1 = api._matrices_key(cm)
    assert api._matrices_key(cm) is k1
    old = cm.a.coeff
    cm.a.coeff = old.copy()
    cm.a.coeff.setflags(write=True)
    cm.a.coeff[0] ^= 1
    assert api._matrices_key(cm) != k1
    assert cm._content_key[0].arrays[2] is cm.a.coeff and old is not cm.a.coeff
    # matrices made of VIEWS of a writable buffer: no memo, and a write through the buffer is seen
    buf = np.zeros(64, np.uint8)
    buf[0] = 1
    cm2 = api.ConstraintMatrices.from_rows(rows, rows, rows, 1, 2)
    cm2.a.coeff = buf[:32]
    k2 = api._matrices_key(cm2)
    assert cm2._content_key is None
    buf[1] = 7
    assert api._matrices_key(cm2) != k2


def test_every_tool_parses():
    """tools/ holds the scripts the evidence under profiles/ was made with (profilers, A/B loops, stress reproducers): each of
    them at least parses - Python by compile(), shell by `bash -n` - and the experiments patch names the tree it applies to"""
    import glob
    tools = os.path.join(ROOT, "tools")
    pys = sorted(glob.glob(os.path.join(tools, "*.py")))
    shs = sorted(glob.glob(os.path.join(tools, "*.sh")))
    assert len(pys) >= 10 and len(shs) >= 5
    for f in pys:
        with open(f) as fh:
            compile(fh.read(), f, "exec")
    for f in shs:
        r = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stderr)
    with open(os.path.join(tools, "patches", "r06_load_stress_experiments.patch")) as fh:
        head = fh.read(2000)
    assert "NOT applied" in head and "fingerprint" in head
