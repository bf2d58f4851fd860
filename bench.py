"""Throughput of the A2C caption-training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full a2c_training minibatch (trainers.py:426-496): policy rollout with
sampling, value + reward chains, loss, backward through policy and value networks,
gradient all-reduce across ranks (N > 1), Adam.  Workload: the batch-512 configuration the
headline metric is quoted on - batch 512 per GPU, 2048-d features, vocab 10k, caption length T = 20, float32 arithmetic.
Inputs (features, start tokens, uniforms) are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "image-captioning-through-rl_amd"

F32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32, dense, MI355X


def build_networks(syn, models, optim, vocab, fdim, seed, dev):
    w2i = syn.make_word_to_idx(vocab)
    nets = (models.PolicyNetwork(w2i, input_dim=fdim), models.ValueNetwork(w2i, feature_dim=fdim),
            models.RewardNetwork(w2i, feature_dim=fdim))
    params = [syn.make_params(k, vocab, fdim, seed) for k in ("policy", "value", "reward")]
    for net, p in zip(nets, params):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        net.to(dev)
    pol, val, rew = nets
    rew.requires_grad_(False)
    a2c = models.AdvantageActorCriticNetwork(val, pol).to(dev)
    flat = optim.flatten_module(a2c)
    return a2c, rew, flat, params


def cpu_baseline(syn, params, vocab, caplen, fdim, sample_batch):
    """The reference-shaped float32 CPU port (oracle/cpu_port.py), one minibatch of
    `sample_batch` rows of the same workload, on this host's cores."""
    from oracle import cpu_port
    nets = cpu_port.PortNets(params[0], params[1], params[2], vocab, fdim)
    opt = cpu_port.make_optimizer(nets)
    feats, caps = syn.make_minibatch(sample_batch, vocab, caplen, fdim, 0)
    np.random.seed(0)
    t0 = time.perf_counter()
    cpu_port.a2c_minibatch(nets, opt, feats, caps)
    dt = time.perf_counter() - t0
    return {"value": sample_batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 A2C minibatch of %d rows (vocab %d, T=%d, %d-d features), float32 torch-CPU, %.1f s"
                      % (sample_batch, vocab, caplen, fdim, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="rows per GPU")
    ap.add_argument("--vocab", type=int, default=10000)
    ap.add_argument("--caplen", type=int, default=20)
    ap.add_argument("--fdim", type=int, default=2048)
    ap.add_argument("--cpu-sample-batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chain-semantics", default="batch", choices=("batch", "per_sample"),
                    help="per_sample: every image owns its value / reward recurrent state (NOT the headline "
                         "workload: a different computation, reported under its own workload name)")
    args = ap.parse_args()

    syn = importlib.import_module(PKG + ".synthetic")
    hip = importlib.import_module(PKG + ".hipabi")
    models = importlib.import_module(PKG + ".models")
    optim = importlib.import_module(PKG + ".optim")
    trainers = importlib.import_module(PKG + ".trainers")
    ddp = importlib.import_module(PKG + ".dist")

    rank, world, local = ddp.init_from_env("nccl")
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n, vocab, caplen, fdim = args.batch, args.vocab, args.caplen, args.fdim
    steps_per_rollout = caplen - 1

    a2c, rew, flat, params = build_networks(syn, models, optim, vocab, fdim, 0, dev)
    opt = optim.FusedAdam(flat, lr=1e-4)
    engine = trainers.build_engine(a2c, rew, chain_semantics=args.chain_semantics)
    feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, rank)  # a different shard per rank
    feats_d = torch.from_numpy(feats).to(dev)
    prefix_d = torch.from_numpy(caps[:, :1]).to(dev)
    total_steps = args.warmup + args.steps
    uniforms = torch.from_numpy(
        np.random.RandomState(1000 + rank).random_sample((total_steps, steps_per_rollout * n))).to(dev)

    # N > 1: the policy network's gradients cross the links beside the value chain's backward pass
    averager = ddp.GradientAverager(flat.flat_grad, flat.span("policy_network."))
    engine.on_policy_grads = averager.start_early

    def one_step(i):
        res = engine.step(feats_d, prefix_d, steps_per_rollout, uniforms=uniforms[i])
        averager.finish()
        opt.step()
        return res.check()  # reads the status word back: one host sync per step, as .item() in the reference loop

    for i in range(args.warmup):
        one_step(i)
    ddp.broadcast_parameters(flat.flat)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # per-launch HIP events (on the launching streams) bracket the launches of the FIRST timed step only:
    # every step issues the same launches, and the ~120 event records of a step cost ~0.5 ms of it
    hip.PROFILE = []
    profiled_steps = 1
    prof = hip.PROFILE
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == profiled_steps:
            hip.PROFILE = None
        res = one_step(args.warmup + i)
    torch.cuda.synchronize()
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
    elapsed = ddp.max_over_ranks(time.perf_counter() - t0, dev)
    hip.PROFILE = None

    # per-kernel accounting from the HIP events recorded on each launch's own stream
    acc = {}
    for kind, work, e0, e1, nbytes, _steps, _flops in prof:
        a = acc.setdefault(kind, [0.0, 0.0, 0, 0.0])
        a[0] += work
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += nbytes
    gemm_flops, gemm_s, gemm_launches, gemm_bytes = acc["gemm"]
    # HBM traffic of the same launches from PMC counters (separate rocprofv3 --pmc passes, FETCH_SIZE
    # doubled as gfx950 needs), measured offline on this workload and committed under profiles/
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)["gemm_f32_kernel"]
        traffic = pmc["fetch_bytes_per_launch"] + pmc["write_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    mfma_util = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_mfma.json")) as f:
            mfma_util = {k: v["mfma_utilisation"] for k, v in json.load(f).items() if k != "_note"}
    except (OSError, KeyError, ValueError, TypeError):
        pass
    achieved = gemm_flops / gemm_s / 1e12
    chains = {k: {"dependent_steps_per_minibatch": int(v[0] / profiled_steps), "us_per_step": 1e6 * v[1] / v[0],
                  "ms_per_minibatch": 1e3 * v[1] / profiled_steps}
              for k, v in acc.items() if k != "gemm"}
    # minimal-algorithm FLOPs of the policy path per image (fwd x3 for fwd+bwd)
    e_dim = h_dim = 512
    fwd = 2 * fdim * h_dim + steps_per_rollout * (2 * (e_dim + h_dim) * 4 * h_dim + 2 * h_dim * vocab)
    policy_flops_per_image = 3 * fwd
    value = world * n * args.steps / elapsed
    # MFMA work the segment-parallel chain kernels executed (warm-up steps included): per launch one
    # [segments, 512] x [512, G*512] product; the chain events above time those launches
    seg_flops = 0.0
    plan_fwd, plan_bwd = res.chain_plan
    plan_rew = res.reward_plan
    flops_of = {}
    for kind, gates, p0, plan in (("lstm_chain_fwd", 4, 1, plan_fwd), ("gru_chain_fwd", 3, 2, plan_rew),
                                  ("lstm_chain_bwd", 4, 1, plan_bwd)):
        if plan[0] == "segmented":
            _, segs, warm = plan
            positions = hip.chain_positions(n, p0, steps_per_rollout)
            launches = warm + (positions + segs - 1) // segs
            flops_of[kind] = launches * 2.0 * segs * 512 * gates * 512
            seg_flops += flops_of[kind]
    if "lstm_chain_fwd" in flops_of and "gru_chain_fwd" in flops_of:
        flops_of["chain_pair_fwd"] = flops_of["lstm_chain_fwd"] + flops_of["gru_chain_fwd"]
    for kind, entry in chains.items():
        if kind in flops_of:   # the segment-parallel kernels are f32 MFMA kernels: price them like the GEMM
            tf = flops_of[kind] / (entry["ms_per_minibatch"] * 1e-3) / 1e12
            entry.update({"gflop_per_minibatch": flops_of[kind] / 1e9, "achieved_tflops": tf,
                          "frac_of_f32_mfma_peak": tf / F32_MFMA_PEAK_TFLOPS})
    step_s = elapsed / args.steps

    if rank == 0:
        out = {
            "metric": "images/sec per A2C train step (batch 512, T=20, vocab 10k)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "A2C minibatch: batch %d per GPU, %d-d features, vocab %d, T=%d "
                                   "(%s%s)" % (n, fdim, vocab, caplen,
                                               "the metric's batch-512 configuration" if args.chain_semantics == "batch"
                                               else "NOT the metric's workload: per-image recurrent state",
                                               ", data-parallel over the GPUs" if world > 1 else ""),
                       "global_batch": world * n, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC, profiles/pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": gemm_bytes / gemm_launches,
                         "kernel": "gemm_f32_kernel, all launches of a step (policy LSTM/vocab GEMMs fwd+bwd, "
                                   "chain input projections and weight gradients)",
                         "launches_per_step": gemm_launches // profiled_steps,
                         "gflop_per_step": gemm_flops / profiled_steps / 1e9,
                         "ms_per_step_in_kernel": 1e3 * gemm_s / profiled_steps,
                         "events": "HIP events around every launch of the first of the timed steps",
                         "mfma_utilisation_pmc": mfma_util,
                         "mfma_utilisation_note": "SQ_VALU_MFMA_BUSY_CYCLES / (dispatch cycles * 1024 SIMDs) per kernel, "
                                                  "kernels run one at a time (profiles/pmc_mfma.json)"},
            "chains": chains,
            "mfma_work_of_whole_step": {
                "gflop_executed_per_step": (gemm_flops / profiled_steps + seg_flops) / 1e9,
                "achieved_tflops": (gemm_flops / profiled_steps + seg_flops) / step_s / 1e12,
                "frac_of_f32_mfma_peak": (gemm_flops / profiled_steps + seg_flops) / step_s / 1e12 / F32_MFMA_PEAK_TFLOPS,
                "note": "all f32 MFMA FLOPs launched in a step (GEMM launches + segment-parallel chain steps, "
                        "their warm-up included) over the step's wall time"},
            "lstm_gemm_roofline_of_whole_step": {
                "policy_flops_per_image_fwd_bwd": policy_flops_per_image,
                "achieved_tflops": policy_flops_per_image * value / world / 1e12,
                "frac_of_f32_mfma_peak": policy_flops_per_image * value / world / 1e12 / F32_MFMA_PEAK_TFLOPS},
            "chain_plan": {"forward": list(plan_fwd), "reward_forward": list(plan_rew), "backward": list(plan_bwd)},
            "seam_check": engine.last_seam,
            "seam_fallbacks_in_run": engine.fallbacks, "last_loss": float(res.scalars[0]),
        }
        if world == 1 and not args.no_cpu_baseline and args.chain_semantics == "batch":
            out["cpu_baseline"] = cpu_baseline(syn, params, vocab, caplen, fdim, args.cpu_sample_batch)
            out["speedup_vs_cpu_port"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
