"""Throughput of the A2C caption-training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full a2c_training minibatch (trainers.py:426-496): policy rollout with
sampling, value + reward chains, loss, backward through policy and value networks,
gradient all-reduce across ranks (N > 1), Adam.  Workload: the batch-512 configuration the
headline metric is quoted on - batch 512 per GPU, 2048-d features, vocab 10k, caption length T = 20, float32 arithmetic.
Inputs (features, start tokens, uniforms) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.

Beside the K timed steps the line carries
  * `soak`: a further run of minibatches on the same networks (they keep training, and the value LSTM's
    memory - hence the chains' warm-up - grows with it): mean / p95 step time, the plans it ended on and
    how many minibatches had to be redone;
  * `roofline`: the step's dominant kernel - gemm_f32x3_kernel, the backward pass's products over all positions -
    from HIP events around every one of its launches in one timed step (on the stream it is launched on): element
    products / time against the f32 MFMA peak, the issued bf16 piece products against the bf16 peak beside it;
    `roofline.one_stream`: the same launches of one more minibatch run on ONE stream, with nothing beside them;
    `roofline_f32_gemm`: the same for the gemm_f32_kernel launches; `roofline_all_gemms`: every GEMM launch;
  * `steady_state`: the soak's mean as images/s, beside `value`;
  * `chains`: the recurrent-chain launches of the same step with the steps they walked, their MFMA work as
    executed (warm-up included) and as the reference's chains would need;
  * `cpu_baseline`: the reference-shaped float32 CPU port timed on this host (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "image-captioning-through-rl_amd"

F32_MFMA_PEAK_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32, dense, MI355X
BF16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16, dense, MI355X
# GEMM families that form float32 products from 16-bit pieces: piece products issued per element product
PIECE_KINDS = {"gemm_f32x3": 6, "gemm_f16x2s": 3, "gemm_f16x2": 3}
PIECE_KERNELS = {
    "gemm_f32x3": "gemm_f32x3_kernel (three bf16 pieces per element, cut on chip: the backward pass's long products with "
                  "--grad-products f32x3)",
    "gemm_f16x2s": "gemm_f16x2s_kernel + gemm_f16x2s_big_kernel (two scaled f16 pieces per element, cut on chip: the backward "
                   "pass's long products - G0 . W and G0^T . Hs on 256x256 tiles, the recurrent layers' weight and input "
                   "gradients on 128x128 tiles)",
    "gemm_f16x2": "gemm_f16x2_kernel (operands cut once into two scaled f16 pieces in fragment order: the rollout's "
                  "per-round vocabulary/gate product and the per-word tables)"}


def build_networks(syn, models, optim, vocab, fdim, seed, dev, value_forget_bias=0.0):
    w2i = syn.make_word_to_idx(vocab)
    nets = (models.PolicyNetwork(w2i, input_dim=fdim), models.ValueNetwork(w2i, feature_dim=fdim),
            models.RewardNetwork(w2i, feature_dim=fdim))
    params = [syn.make_params(k, vocab, fdim, seed) for k in ("policy", "value", "reward")]
    if value_forget_bias:
        params[1]["valrnn.lstm.bias_ih_l0"][512:1024] += value_forget_bias
    for net, p in zip(nets, params):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        net.to(dev)
    pol, val, rew = nets
    rew.requires_grad_(False)
    a2c = models.AdvantageActorCriticNetwork(val, pol).to(dev)
    flat = optim.flatten_module(a2c)
    return a2c, rew, flat, params


def pretrained_networks(syn, models, optim, trainers, vocab, fdim, n, minibatches, dev):
    """The three networks as the reference's pipeline hands them to the A2C loop (trainers.py:331-370): each trained
    by its own pre-training loop (train_reward_network, train_policy_network, train_value_network) for `minibatches`
    minibatches of n rows on a synthetic ragged-caption set of 4 n rows, then loaded from the checkpoints those
    loops wrote (their best-loss states, as the reference loads them)."""
    import math
    import tempfile
    torch.manual_seed(0)
    np.random.seed(0)
    import random
    random.seed(0)
    data = syn.make_ragged_train_data(4 * n, vocab, fdim, 0)
    epochs = max(1, math.ceil(minibatches / 4))
    tmp = tempfile.mkdtemp(prefix="icrl_pretrain_")
    paths = {k: os.path.join(tmp, k + ".pt") for k in ("reward_network", "policy_network", "value_network")}
    log = trainers.ScalarLog()
    t0 = time.perf_counter()
    trainers.train_reward_network(data, paths, log, False, epochs=epochs, batch_size=n)
    trainers.train_policy_network(data, paths, log, False, epochs=epochs, batch_size=n)
    trainers.train_value_network(data, paths, log, False, epochs=epochs, batch_size=n)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    w2i = data["word_to_idx"]
    nets = {"policy_network": models.PolicyNetwork(w2i, input_dim=fdim),
            "value_network": models.ValueNetwork(w2i, feature_dim=fdim),
            "reward_network": models.RewardNetwork(w2i, feature_dim=fdim)}
    for k, net in nets.items():
        net.load_state_dict(torch.load(paths[k], map_location="cpu", weights_only=True), strict=False)
        net.to(dev)
        os.remove(paths[k])
    os.rmdir(tmp)
    nets["reward_network"].requires_grad_(False)
    a2c = models.AdvantageActorCriticNetwork(nets["value_network"], nets["policy_network"]).to(dev)
    losses = {}
    for tag, value, _ in log.records:
        losses.setdefault(tag, []).append(value)
    info = {"minibatches_per_network": 4 * epochs, "rows_per_minibatch": n, "seconds": secs,
            "losses_first_last_best": {t: [v[0], v[-1], min(v)] for t, v in losses.items()}}
    return a2c, nets["reward_network"], optim.flatten_module(a2c), info


def host_cores():
    """(threads torch uses, CPUs this process may run on, physical cores among them)."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cores = set()
    try:
        for cpu in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            with open(base + "physical_package_id") as f, open(base + "core_id") as g:
                cores.add((f.read().strip(), g.read().strip()))
    except OSError:
        cores = set()
    return torch.get_num_threads(), len(allowed), len(cores) or None


def cpu_baseline(syn, params, vocab, caplen, fdim, sample_batch, repeats):
    """The reference-shaped float32 CPU port (oracle/cpu_port.py) on this host's cores: one untimed
    minibatch, then the median of `repeats` timed ones, `sample_batch` rows of the same workload each."""
    from oracle import cpu_port
    nets = cpu_port.PortNets(params[0], params[1], params[2], vocab, fdim)
    opt = cpu_port.make_optimizer(nets)
    feats, caps = syn.make_minibatch(sample_batch, vocab, caplen, fdim, 0)
    np.random.seed(0)
    times = []
    for i in range(repeats + 1):
        t0 = time.perf_counter()
        cpu_port.a2c_minibatch(nets, opt, feats, caps)
        if i > 0:
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    threads, allowed, physical = host_cores()
    return {"value": sample_batch / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "cpus_allowed": allowed, "physical_cores_allowed": physical,
            "sample": "median of %d A2C minibatches (after one untimed) of %d rows (vocab %d, T=%d, %d-d features), "
                      "float32 torch-CPU on %d threads, %.1f s each" % (repeats, sample_batch, vocab, caplen, fdim,
                                                                        threads, dt)}


def git_sha():
    """HEAD of this tree; where the tree travelled without its history, the sha the library was built at
    (csrc/BUILD_SHA, written by __graft_entry__.build())."""
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                             timeout=5).stdout.strip()
        if sha:
            return sha
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        with open(os.path.join(ROOT, PKG, "csrc", "BUILD_SHA")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def load_profile(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def plans_of(res, engine):
    fwd, bwd = res.chain_plan
    return {"forward": list(fwd), "reward_forward": list(res.reward_plan), "backward": list(bwd),
            "full_calls": res.full_calls, "replay_check": list(engine.last_replay) if engine.last_replay else None,
            "warm16_steps": list(engine.last_cheap), "warm16_steps_backward": engine.last_cheap_bwd}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="rows per GPU")
    ap.add_argument("--vocab", type=int, default=10000)
    ap.add_argument("--caplen", type=int, default=20)
    ap.add_argument("--fdim", type=int, default=2048)
    ap.add_argument("--soak", type=int, default=150, help="minibatches of the steady-state run after the timed steps (0: none)")
    ap.add_argument("--cpu-sample-batch", type=int, default=256)
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo: the ranks' collectives go through host memory (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--policy-dtype", default="f32", choices=("f32", "bf16"),
                    help="bf16: the policy network's products on the bf16 matrix instruction (NOT the headline "
                         "workload's arithmetic: reported under its own workload name and dtype)")
    ap.add_argument("--grad-products", default=None, choices=("f32", "f32x3", "f16x2"),
                    help="how the backward pass forms its products over all positions (default: the engine's)")
    ap.add_argument("--value-forget-bias", type=float, default=0.0,
                    help="added to the value LSTM's forget-gate bias: a recurrence with a longer memory (stress case, "
                         "reported under its own workload name)")
    ap.add_argument("--chain-semantics", default="batch", choices=("batch", "per_sample"),
                    help="per_sample: every image owns its value / reward recurrent state (NOT the headline "
                         "workload: a different computation, reported under its own workload name)")
    ap.add_argument("--pretrain", type=int, default=0,
                    help="run each of the three pre-training loops for this many minibatches first and bench the A2C "
                         "step on the networks they produce (trained-regime run, reported under its own workload name)")
    ap.add_argument("--rollout-arith", default=None, choices=("f32", "f16x2"),
                    help="the rollout's per-round vocabulary/gate product (default: the engine's, f16x2)")
    ap.add_argument("--chain-arith", default=None, choices=("f32", "f16x2"),
                    help="recurrent products of the forward chains' persistent launch (default: the engine's)")
    ap.add_argument("--value-bptt-first", action="store_true",
                    help="queue the value chain's BPTT ahead of the policy LSTM's (the order before the reordered schedule)")
    ap.add_argument("--max-streams", type=int, default=None, choices=(3, 4),
                    help="streams the engine uses, the caller's included (default: the engine's)")
    ap.add_argument("--no-late-policy-bptt", action="store_true",
                    help="the policy LSTM's BPTT as 3 T launches on the side stream beside the value chain's (the "
                         "schedule before the persistent policy BPTT launch)")
    ap.add_argument("--no-collapse", action="store_true", help="walk every call of the chains in full")
    ap.add_argument("--no-warm16", action="store_true", help="every warm-up step on the float32 matrix instruction")
    args = ap.parse_args()
    watchdog = float(os.environ.get("ICRL_BENCH_WATCHDOG", "0"))
    if watchdog > 0:   # a run that is still going after this many seconds prints every thread's stack and exits
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)

    syn = importlib.import_module(PKG + ".synthetic")
    hip = importlib.import_module(PKG + ".hipabi")
    models = importlib.import_module(PKG + ".models")
    optim = importlib.import_module(PKG + ".optim")
    trainers = importlib.import_module(PKG + ".trainers")
    ddp = importlib.import_module(PKG + ".dist")

    rank, world, local = ddp.init_from_env(args.backend)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    local = local % max(1, torch.cuda.device_count())   # more ranks than GPUs (a rehearsal): they share
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if os.environ.get("ICRL_BENCH_OWN_STREAM") == "1":   # diagnosis: the step's main stream is not the device's default one
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    n, vocab, caplen, fdim = args.batch, args.vocab, args.caplen, args.fdim
    steps_per_rollout = caplen - 1

    a2c, rew, flat, params = build_networks(syn, models, optim, vocab, fdim, 0, dev, args.value_forget_bias)
    pretrain_info = None
    if args.pretrain > 0:
        assert world == 1 and not args.value_forget_bias, "--pretrain: one GPU, no forget-bias shift"
        a2c, rew, flat, pretrain_info = pretrained_networks(syn, models, optim, trainers, vocab, fdim, n,
                                                            args.pretrain, dev)
    opt = optim.FusedAdam(flat, lr=1e-4)
    engine_opts = dict(chain_semantics=args.chain_semantics, collapse_replays=not args.no_collapse)
    if args.no_warm16:
        engine_opts["warm16"] = False
    if args.chain_arith is not None:
        engine_opts["chain_arith"] = args.chain_arith
    if args.rollout_arith is not None:
        engine_opts["rollout_arith"] = args.rollout_arith
    if args.no_late_policy_bptt:
        engine_opts["late_policy_bptt"] = False
    if args.policy_dtype != "f32":
        engine_opts["policy_dtype"] = args.policy_dtype
    if args.grad_products is not None:
        engine_opts["grad_products"] = args.grad_products
    engine = trainers.build_engine(a2c, rew, **engine_opts)
    if args.max_streams is not None:
        engine.max_streams = args.max_streams
    if args.value_bptt_first:
        engine.policy_bptt_first = False
    feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, rank)  # a different shard per rank
    feats_d = torch.from_numpy(feats).to(dev)
    prefix_d = torch.from_numpy(caps[:, :1]).to(dev)
    total_steps = args.warmup + args.steps + args.soak + 2
    uniforms = torch.from_numpy(
        np.random.RandomState(1000 + rank).random_sample((total_steps, steps_per_rollout * n))).to(dev)

    # N > 1: one all-reduce of the flat gradient buffer at the end of the minibatch.  (With the value chain's BPTT in
    # front of the policy LSTM's - the earlier schedule, --value-bptt-first - the policy network's gradients are
    # complete milliseconds before the value network's and cross the links in a piece of their own beside the rest of
    # the backward pass; with the policy BPTT first both are complete at about the same time, and two more collectives
    # would only add their launches.  The choice is a constructor-level one, so every rank takes it alike.)
    averager = ddp.GradientAverager(flat.flat_grad, flat.span("policy_network."), stream=engine.side_stream())
    if not engine.policy_bptt_first:
        engine.on_policy_grads = averager.start_early

    sync_each = os.environ.get("ICRL_BENCH_SYNC_EACH") == "1"   # diagnosis: the device drained after every minibatch

    head_start = int(os.environ.get("ICRL_BENCH_HEAD_START", "0"))   # diagnosis (kernel traces): the device idles this many
    # clock ticks in front of every minibatch, so that a host slowed down by a profiler still queues ahead of it

    def one_step(i):
        if head_start:
            torch.cuda._sleep(head_start)
        res = engine.step(feats_d, prefix_d, steps_per_rollout, uniforms=uniforms[i])
        averager.finish()
        opt.step()
        if sync_each:
            torch.cuda.synchronize()
        return res.check()  # reads the status word back: one host sync per step, as .item() in the reference loop

    for i in range(args.warmup):
        one_step(i)
    ddp.broadcast_parameters(flat.flat)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # per-launch HIP events (on the launching streams) bracket the launches of the FIRST timed step only:
    # a step's ~120 event records cost ~0.5 ms of it.  The plans that step ran with are kept beside them.
    hip.PROFILE = []
    prof = hip.PROFILE
    fallbacks_before = engine.fallbacks
    profiled_plans, profiled_clean = None, True
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == 1:
            hip.PROFILE = None
        res = one_step(args.warmup + i)
        if i == 0:
            profiled_plans = plans_of(res, engine)
            profiled_clean = engine.fallbacks == fallbacks_before   # no redone attempt among the recorded launches
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    elapsed = ddp.max_over_ranks(time.perf_counter() - t0, dev)
    hip.PROFILE = None
    timed_fallbacks = engine.fallbacks - fallbacks_before
    final_plans = plans_of(res, engine)
    last_loss = float(res.scalars[0])
    in_step = ddp.same_on_all_ranks(flat.flat)   # every rank applied the same Adam updates to the same parameters
    gave_up = int(ddp.max_over_ranks(engine.launch_timeouts, dev))

    # ---- steady state: the networks keep training, the value LSTM's memory (and with it the warm-up) moves
    soak = None
    if args.soak > 0:
        times, before = [], engine.fallbacks
        for i in range(args.soak):
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            res_s = one_step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - s0)
        worst = ddp.max_over_ranks(float(np.mean(times)), dev)
        soak = {"minibatches": args.soak, "after_minibatches": args.warmup + args.steps,
                "ms_per_step_mean": 1e3 * worst, "ms_per_step_p95": 1e3 * float(np.percentile(times, 95)),
                "ms_per_step_max": 1e3 * float(np.max(times)), "images_per_s": world * n / worst,
                "redone_minibatches": engine.fallbacks - before, "final_plans": plans_of(res_s, engine),
                "note": "each minibatch synchronised and timed on the host; the same networks keep training"}

    # ---- the same launches with nothing beside them: one more minibatch with every launch on ONE stream, each
    # bracketed by HIP events.  In the timed steps the GEMM launches share the CUs with the chain kernels and
    # with each other (three streams), so their durations there say how the step is packed, not how good the
    # kernel is; this pass gives the kernel's own figure for the very same shapes.
    one_stream, pieces_alone = None, {}
    if world == 1:
        engine.concurrent = False
        one_step(total_steps - 2)
        torch.cuda.synchronize()
        hip.PROFILE = []
        one_step(total_steps - 1)
        torch.cuda.synchronize()
        alone, hip.PROFILE = hip.PROFILE, None
        engine.concurrent = True
        fl = sum(r[1] for r in alone if r[0] == "gemm")
        se = sum(r[2].elapsed_time(r[3]) for r in alone if r[0] == "gemm") * 1e-3
        if se > 0:
            one_stream = {"achieved": fl / se / 1e12, "frac": fl / se / 1e12 / F32_MFMA_PEAK_TFLOPS,
                          "launches": sum(1 for r in alone if r[0] == "gemm"), "gflop": fl / 1e9,
                          "ms_in_kernel": 1e3 * se,
                          "note": "the step's gemm_f32_kernel launches with every launch of the minibatch on one "
                                  "stream (nothing runs beside them), HIP events around each"}

        pieces_alone = {}
        for kind, per_element in PIECE_KINDS.items():
            fx = sum(r[1] for r in alone if r[0] == kind)
            sx = sum(r[2].elapsed_time(r[3]) for r in alone if r[0] == kind) * 1e-3
            if sx > 0:
                pieces_alone[kind] = {
                    "achieved_tflops": fx / sx / 1e12, "frac_of_f32_mfma_peak": fx / sx / 1e12 / F32_MFMA_PEAK_TFLOPS,
                    "frac_of_16bit_mfma_peak": per_element * fx / sx / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                    "launches": sum(1 for r in alone if r[0] == kind), "gflop": fx / 1e9, "ms_in_kernel": 1e3 * sx}

    # ---- per-kernel accounting of the profiled step from the HIP events recorded on each launch's own stream
    acc, chain_steps, chain_flops = {}, {}, {}
    for kind, work, e0, e1, nbytes, steps, flops in prof:
        a = acc.setdefault(kind, [0.0, 0.0, 0, 0.0])
        a[0] += work
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += nbytes
        chain_steps[kind] = chain_steps.get(kind, 0) + steps
        chain_flops[kind] = chain_flops.get(kind, 0.0) + flops
    gemm_flops, gemm_s, gemm_launches, gemm_bytes = acc.get("gemm", [0.0, 1e-30, 0, 0.0])
    bf16 = None
    if "gemm_bf16" in acc:   # --policy-dtype bf16: the policy network's products, priced against the bf16 peak
        bf, bs, bl, bb = acc["gemm_bf16"]
        bf16 = {"kernel": "gemm_bf16_kernel (policy network products)", "launches_per_step": bl, "gflop_per_step": bf / 1e9,
                "ms_per_step_in_kernel": 1e3 * bs, "achieved_tflops": bf / bs / 1e12,
                "frac_of_bf16_mfma_peak": bf / bs / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                "algorithmic_bytes_per_launch": bb / max(1, bl)}
    # products formed from 16-bit pieces of float32 operands: element products (2 M N K) counted once, and the piece
    # products the kernel really issues (six bf16 ones / three f16 ones per element product) against the 16-bit peak
    pieces = {}
    for kind, per_element in PIECE_KINDS.items():
        if kind not in acc:
            continue
        xf, xs, xl, xb = acc[kind]
        pieces[kind] = {"kernel": PIECE_KERNELS[kind], "launches_per_step": xl, "gflop_per_step": xf / 1e9,
                        "ms_per_step_in_kernel": 1e3 * xs, "achieved_tflops": xf / xs / 1e12,
                        "frac_of_f32_mfma_peak": xf / xs / 1e12 / F32_MFMA_PEAK_TFLOPS,
                        "piece_products_per_element_product": per_element,
                        "mfma_16bit_tflops_issued": per_element * xf / xs / 1e12,
                        "frac_of_16bit_mfma_peak": per_element * xf / xs / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                        "algorithmic_bytes_per_launch": xb / max(1, xl), "one_stream": pieces_alone.get(kind)}
    peak = F32_MFMA_PEAK_TFLOPS
    achieved = gemm_flops / gemm_s / 1e12
    pmc_t, pmc_m = load_profile("pmc_traffic.json"), load_profile("pmc_mfma.json")

    def pmc_traffic(kernel):
        # (a family's kernels together: bytes per launch weighted by their launch counts in the profiled run)
        rows = [pmc_t[k] for k in (kernel, kernel.replace("_kernel", "_big_kernel")) if pmc_t and k in pmc_t]
        launches = sum(r["launches"] for r in rows)
        if launches:
            return sum((r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"] for r in rows) / launches
        return None
    traffic = pmc_traffic("gemm_f32_kernel")
    mfma_util = {k: v["mfma_utilisation"] for k, v in pmc_m.items() if isinstance(v, dict) and "mfma_utilisation" in v} \
        if pmc_m else None
    common = {"traffic_unit": "HBM bytes per launch (PMC, profiles/pmc_traffic.json)",
              "traffic_profile_git_sha": (pmc_t or {}).get("_git_sha"),
              "events": "HIP events around every launch of the first of the timed steps, on its own stream",
              "mfma_utilisation_pmc": mfma_util, "mfma_profile_git_sha": (pmc_m or {}).get("_git_sha"),
              "mfma_utilisation_note": "SQ_VALU_MFMA_BUSY_CYCLES / (dispatch cycles * 1024 SIMDs) per kernel, "
                                       "kernels run one at a time (profiles/pmc_mfma.json)"}
    roofline_f32 = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "algorithmic_bytes_per_launch": gemm_bytes / max(1, gemm_launches),
                    "kernel": "gemm_f32_kernel, all launches of a step: the forward pass's products (the rollout's "
                              "vocabulary/gate products, per-word tables, chain input projections, heads) and the "
                              "backward pass's short ones; with --grad-products f32 also its long ones",
                    "launches_per_step": gemm_launches, "gflop_per_step": gemm_flops / 1e9,
                    "ms_per_step_in_kernel": 1e3 * gemm_s, "one_stream": one_stream}
    # the step's dominant kernel is the one `roofline` prices: the GEMM family with the most in-kernel time in the
    # profiled step.  A family built on 16-bit pieces is priced on the piece products it issues against the dense
    # 16-bit MFMA peak (its element rate against the f32 MFMA peak is reported beside it).
    dominant = max(pieces, key=lambda k: pieces[k]["ms_per_step_in_kernel"]) if pieces else None
    if dominant is not None and pieces[dominant]["ms_per_step_in_kernel"] >= 1e3 * gemm_s * 0.5:
        d = pieces[dominant]
        roofline = {"bound": "mfma", "achieved": d["mfma_16bit_tflops_issued"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": d["frac_of_16bit_mfma_peak"], "traffic": pmc_traffic(PIECE_KERNELS[dominant].split(" ")[0]),
                    "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
                    "kernel": d["kernel"], "launches_per_step": d["launches_per_step"],
                    "gflop_per_step": d["gflop_per_step"], "ms_per_step_in_kernel": d["ms_per_step_in_kernel"],
                    "peak_note": "achieved counts the 16-bit piece products the kernel issues (%d per element product "
                                 "2 M N K, float32 sums) against the dense f16/bf16 MFMA peak; the element products "
                                 "themselves against the f32 MFMA peak, the rate an exact-f32 kernel could reach at "
                                 "best, are given beside it" % d["piece_products_per_element_product"],
                    "element_tflops": d["achieved_tflops"], "peak_f32_mfma": F32_MFMA_PEAK_TFLOPS,
                    "frac_of_f32_mfma_peak": d["frac_of_f32_mfma_peak"], "one_stream": d["one_stream"]}
    else:
        roofline = dict(roofline_f32)
    roofline.update(common)
    all_fl = sum(a[0] for k, a in acc.items() if k.startswith("gemm"))
    all_s = sum(a[1] for k, a in acc.items() if k.startswith("gemm"))
    roofline_all = None if all_s <= 0 else {
        "kernels": sorted(k for k in acc if k.startswith("gemm")),
        "launches_per_step": sum(a[2] for k, a in acc.items() if k.startswith("gemm")), "gflop_per_step": all_fl / 1e9,
        "ms_per_step_in_kernel": 1e3 * all_s, "achieved": all_fl / all_s / 1e12, "unit": "TFLOP/s",
        "frac_of_f32_mfma_peak": all_fl / all_s / 1e12 / peak,
        "note": "every GEMM launch of the profiled step, element products over the sum of the launches' durations "
                "(they overlap each other and the chains, so the durations add up to more than the step)"}

    # the chains: steps each launch walked, MFMA work executed and the work the reference's chains hold
    gates = {"lstm_chain_fwd": 4, "gru_chain_fwd": 3, "lstm_chain_bwd": 4}
    ref_positions = {"lstm_chain_fwd": hip.chain_positions(n, 1, steps_per_rollout),
                     "gru_chain_fwd": hip.chain_positions(n, 2, steps_per_rollout),
                     "lstm_chain_bwd": hip.chain_positions(n, 1, steps_per_rollout)}
    per_position = 2.0 * 512 * 512
    ref_flops = {k: gates[k] * ref_positions[k] * per_position for k in gates}
    ref_flops["chain_pair_fwd"] = ref_flops["lstm_chain_fwd"] + ref_flops["gru_chain_fwd"]
    chains, executed_chain_flops = {}, 0.0
    segs = profiled_plans["forward"][1] if profiled_plans and profiled_plans["forward"][0] == "segmented" else None
    for kind, (work, secs, launches, _) in acc.items():
        if kind.startswith("gemm"):
            continue
        entry = {"launches": launches, "positions_walked": int(work), "steps": chain_steps[kind],
                 "ms": 1e3 * secs, "us_per_step": 1e6 * secs / max(1, chain_steps[kind])}
        executed = None
        if chain_flops[kind] > 0:
            executed = chain_flops[kind]
        elif kind in gates and segs:
            executed = chain_steps[kind] * segs * gates[kind] * per_position
        if executed is not None:
            executed_chain_flops += executed
            tf = executed / secs / 1e12
            entry.update({"gflop_executed": executed / 1e9, "achieved_tflops": tf, "frac_of_f32_mfma_peak": tf / peak})
            if kind.endswith("f16x2") or (kind in ("policy_bptt", "replay_check") and engine.chain_arith == "f16x2"):
                # these launches form every element product from three f16 piece products
                entry["frac_of_16bit_mfma_peak"] = 3.0 * tf / BF16_MFMA_PEAK_TFLOPS
        if kind in ref_flops:
            entry["gflop_reference_chain"] = ref_flops[kind] / 1e9
        chains[kind] = entry

    # minimal-algorithm FLOPs of the policy path per image (fwd x3 for fwd+bwd)
    e_dim = h_dim = 512
    fwd = 2 * fdim * h_dim + steps_per_rollout * (2 * (e_dim + h_dim) * 4 * h_dim + 2 * h_dim * vocab)
    policy_flops_per_image = 3 * fwd
    value = world * n * args.steps / elapsed
    step_s = elapsed / args.steps

    if rank == 0:
        variant = []
        if args.chain_semantics != "batch":
            variant.append("NOT the metric's workload: per-image recurrent state")
        if args.value_forget_bias:
            variant.append("NOT the metric's workload: value LSTM forget bias +%g (long-memory stress case)" % args.value_forget_bias)
        if pretrain_info:
            variant.append("networks pre-trained for %d minibatches each by the three pre-training loops: a "
                           "trained-regime run, not the headline's fresh initialisation"
                           % pretrain_info["minibatches_per_network"])
        if args.policy_dtype != "f32":
            variant.append("NOT the metric's arithmetic: policy network products in %s" % args.policy_dtype)
        if (n, fdim, vocab, caplen) != (512, 2048, 10000, 20):
            variant.append("NOT the metric's size")
        out = {
            "metric": "images/sec per A2C train step (batch 512, T=20, vocab 10k)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.policy_dtype, "data": "synthetic",
            "config": {"workload": "A2C minibatch: batch %d per GPU, %d-d features, vocab %d, T=%d (%s%s)"
                                   % (n, fdim, vocab, caplen,
                                      "; ".join(variant) if variant else "the metric's batch-512 configuration",
                                      ", data-parallel over the GPUs" if world > 1 else ""),
                       "global_batch": world * n, "parallelism": "dp%d" % world,
                       "backend": args.backend if world > 1 else None},
            "parameters_identical_across_ranks": in_step if world > 1 else None,
            "launches_that_gave_up_waiting": gave_up,
            "soak": soak,
            "pretrain": pretrain_info,
            "steady_state": None if soak is None else {
                "value": soak["images_per_s"], "unit": "images/s", "ms_per_step": soak["ms_per_step_mean"],
                "note": "mean over the soak's %d further minibatches of the same networks (see `soak`); `value` above "
                        "is the state after %d minibatches" % (args.soak, args.warmup)},
            "roofline": roofline,
            "roofline_f32_gemm": roofline_f32,
            "roofline_all_gemms": roofline_all,
            "bf16_gemms": bf16,
            "piece_gemms": pieces,
            "grad_products": engine.vgd, "rollout_arith": engine.rollout_arith, "chain_arith": engine.chain_arith,
            "chains": chains,
            "profiled_step": {"plans": profiled_plans, "no_attempt_redone": profiled_clean,
                              "note": "the roofline and chain figures are this step's launches, with this step's plans"},
            "mfma_work_of_whole_step": {
                "gflop_executed_per_step": (gemm_flops + executed_chain_flops) / 1e9,
                "achieved_tflops": (gemm_flops + executed_chain_flops) / step_s / 1e12,
                "frac_of_f32_mfma_peak": (gemm_flops + executed_chain_flops) / step_s / 1e12 / peak,
                "note": "the gemm_f32_kernel launches' and the segment-parallel chain steps' FLOPs (warm-up included; "
                        "not the products formed from 16-bit pieces) over the mean step time"},
            "lstm_gemm_roofline_of_whole_step": {
                "policy_flops_per_image_fwd_bwd": policy_flops_per_image,
                "achieved_tflops": policy_flops_per_image * value / world / 1e12,
                "frac_of_f32_mfma_peak": policy_flops_per_image * value / world / 1e12 / peak},
            "chain_plan": final_plans, "seam_check": engine.last_seam,
            "seam_fallbacks_in_run": timed_fallbacks, "last_loss": last_loss, "git_sha": git_sha(),
        }
        if (world == 1 and not args.no_cpu_baseline and args.chain_semantics == "batch" and not args.value_forget_bias
                and not pretrain_info):
            out["cpu_baseline"] = cpu_baseline(syn, params, vocab, caplen, fdim, args.cpu_sample_batch, args.cpu_repeats)
            out["speedup_vs_cpu_port"] = value / out["cpu_baseline"]["value"]
            if soak:
                out["soak_speedup_vs_cpu_port"] = soak["images_per_s"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
