"""bench.py launched the way the multi-GPU driver launches it, with two ranks that share ONE GPU and exchange
their gradients through gloo: the barrier, the max-over-ranks timing, the rank-0 print and the per-rank shards
of the N > 1 path, run for real.  Needs an MI355X.

The gradients cross through host memory (dist._through_host), staged explicitly."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


_launches = []


def _launch(ranks, extra, timeout=240, backend="gloo", env_extra=None):
    _launches.append(1)
    port = 29700 + (os.getpid() * 7 + len(_launches) * 13) % 250   # a fresh port per launch
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--backend", backend, "--no-cpu-baseline"] + extra
    # a rank still running after 120 s prints its threads' stacks and exits, so a hang names its place
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ICRL_BENCH_WATCHDOG="120", **(env_extra or {}))
    done = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0: %r" % done.stdout[-2000:]
    return json.loads(lines[0])


def test_two_rank_bench_prints_one_line_and_keeps_the_ranks_in_step():
    n = 64
    out = _launch(2, ["--steps", "3", "--warmup", "2", "--soak", "6", "--batch", str(n), "--vocab", "1000",
                      "--fdim", "512", "--caplen", "12"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 2
    assert out["config"]["global_batch"] == 2 * n and out["config"]["parallelism"] == "dp2"
    assert out["config"]["backend"] == "gloo" and out["scaling"] == "weak"
    assert out["parameters_identical_across_ranks"] is True
    assert out["launches_that_gave_up_waiting"] == 0
    assert out["value"] == pytest.approx(2 * n * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)
    assert out["soak"]["minibatches"] == 6 and out["soak"]["images_per_s"] > 0
    assert "cpu_baseline" not in out   # rank 0 at N = 1 only
    assert out["last_loss"] == out["last_loss"]   # not NaN


def test_two_rank_bench_with_persistent_chains_at_batch_512():
    """Both ranks run the persistent chain kernels of the batch-512 step on the SAME GPU, in lock step.  The f16
    launches hold every CU's registers (the forward pair: two workgroups per CU; the policy LSTM's BPTT: one per CU
    on all 256), so two processes that start such a launch at the same moment can each hold part of the device and
    wait for the rest - which a rank with a GPU of its own never sees.  What this rehearsal checks is the way out:
    the bounded wait gives up (counted), the minibatch's launches are repeated, and the ranks end on identical
    parameters and a finite loss.  (Vocabulary and feature width are cut down: they set the size of the gradients,
    which this rehearsal moves through host memory.)"""
    out = _launch(2, ["--steps", "2", "--warmup", "1", "--soak", "0", "--vocab", "1000", "--fdim", "512"])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 1024
    assert out["parameters_identical_across_ranks"] is True
    assert out["profiled_step"]["plans"]["forward"][:2] == ["segmented", 256]
    # (three minibatches; the forward pair, the value chain's BPTT or the policy LSTM's BPTT may each give up in one)
    assert out["launches_that_gave_up_waiting"] <= 6 and out["last_loss"] == out["last_loss"]


def test_rccl_collectives_beside_the_persistent_chain_kernels_on_one_rank():
    """The N > 1 exchange path over RCCL itself, as far as one GPU can run it: a single-rank `nccl` process group with
    every collective issued (ICRL_FORCE_COLLECTIVES=1) - the broadcast, the early all-reduce of the policy network's
    gradients on its own stream beside the value chain's persistent BPTT launch, the second piece, the max over
    ranks - at the metric's size.  No persistent launch may give up waiting for its workgroups, and the step may
    cost at most 5 % more than the same run with no process group (measured: 1-2 %, 0.05-0.1 ms: one all-reduce of
    the flat buffer at the end of every step - the policy network's gradients are complete about when the value
    network's are, so bench.py no longer sends them as a piece of their own; with three collectives a step it was
    1-8 %; a launch that gives up or a stream that serialises behind a collective costs 20 % and more).  (More than one
    RCCL rank needs more than one GPU: that stays unmeasured here.)"""
    size = ["--steps", "20", "--warmup", "5", "--soak", "0"]
    plain = _launch(1, size, backend="nccl")
    forced = _launch(1, size, backend="nccl", env_extra={"ICRL_FORCE_COLLECTIVES": "1"})
    assert forced["n_gpus"] == 1 and forced["launches_that_gave_up_waiting"] == 0
    assert forced["seam_fallbacks_in_run"] == plain["seam_fallbacks_in_run"]
    assert forced["parameters_identical_across_ranks"] is None   # one rank: nothing to compare
    print("ms per step: %.3f without a process group, %.3f with every collective issued over RCCL"
          % (plain["ms_per_step"], forced["ms_per_step"]))
    assert forced["ms_per_step"] <= 1.05 * plain["ms_per_step"]
