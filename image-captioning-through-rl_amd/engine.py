"""One A2C minibatch (rollout, rewards, loss, gradients) on the HIP kernels.

This is the host-side schedule of the hot path of trainers.py:436-480 (and the
curriculum variant, trainers.py:547-594).  It is NOT a step-by-step replay of the
reference's Python loop; it computes the same numbers in a different order:

  1. policy rollout, incremental: the LSTM state is carried from step to step and only
     the newest position is projected to the vocabulary (the reference reruns the whole
     prefix and projects every position each step, models.py:71-84, 286);
  2. once all tokens exist, the value network's batch-1 LSTM recurrence and the reward
     network's batch-1 GRU recurrence (models.py:133, 226 fed (N,1,E) without
     batch_first) run over every step's prefix, both chains at once;
  3. loss (trainers.py:471-475), then the backward pass: the policy LSTM's BPTT and the value
     chain's BPTT as one persistent launch each (the policy LSTM's first by default), the
     products over all (step, row) positions - vocabulary, weight and input gradients - on two
     more streams beside the value chain's BPTT and behind it; or, where the persistent launches
     do not apply, the policy BPTT step by step on one stream beside the value chain's on another.

The two recurrences can be evaluated in two ways (``chain_mode``):
  * "exact": the persistent kernels of chain.hip walk the ~10^5 dependent steps one by one;
  * "segmented": chain_seg.hip cuts the chain into segments that run side by side after a
    warm-up, and reports how far each segment's warmed-up state was from the true state at
    its seam.  A result is accepted only if every seam agrees to rounding level; otherwise
    it is redone with a longer warm-up and finally with the exact kernels, so the result
    never depends on the recurrence forgetting fast enough.  Forward and backward keep their
    own warm-up length; a rejected backward seam redoes only the value chain's BPTT.  With
    ``persistent`` (default) and at most 512 segments all steps of the two forward chains
    run in one launch, and all steps of the backward chain in another;
  * "auto" (default): segmented for chains of at least 4096 positions, else exact.

Replays left out (``collapse_replays``, default on, batch semantics only).  Call s of the value network walks
caption columns 0 .. p0+s-1 (models.py:168-169), so call s+1 starts by replaying every input of call s from
the state call s ended in, and only then reads its one new column; the reward network does the same one
column further (models.py:254-255).  A recurrence that forgets ends such a replay in the state it started
from, so the chains walk calls 0 .. fc-1 in full and then one column per later call: about a tenth of the
positions at batch 512, T = 20, forward and backward (the gate gradients of the short chain are the full
chain's summed over calls).  Nothing is assumed: beside the backward pass a third stream runs the first
steps of every replay that was left out, from the state the chain had there, and the step stands only if each
one has met the trajectory of the last fully walked call to rounding (icrl_chain_pair_replay_check), and if
the accepted backward warm-up, which bounds how far a gradient reaches, is shorter than the shortest replay
left out; otherwise the minibatch is redone with more calls in full, in the end with all of them.

``chain_semantics="per_sample"`` is an explicit change of what is computed, never a default: every
image owns its recurrent state (the reference's own classes called with one image at a time,
state kept across the calls of a minibatch), so row i's values and rewards no longer depend on the
rows before it.  The N chains then advance together, one [N,512] x W_hh product per step, through
the same kernels (a "segment" is a row: no warm-up, no seams, nothing approximate).

All device memory is PyTorch-owned and cached per problem shape; nothing here touches
the CPU oracle, and every kernel call raises if libicrl_a2c.so is missing.
"""
import torch

from . import hipabi as hip

H = 512


class A2CStepResult:
    """Device tensors of one minibatch.  tokens [N, p0+S] int64; log_probs, values,
    rewards [N, S] float32; scalars[0..4] = loss, actor, critic, mean reward, mean adv."""

    def __init__(self, tokens, log_probs, values, rewards, scalars, entropy, status, chain_plan=None, reward_plan=None,
                 status_host=None, full_calls=None):
        self.tokens, self.log_probs, self.values, self.rewards = tokens, log_probs, values, rewards
        self.scalars, self.entropy, self.status = scalars, entropy, status
        # the status words as read back together with the backward seam check (after every chain kernel of
        # the minibatch had finished), when the step made that read; check() then needs no second round trip
        self.status_host = status_host
        # (value chain forward, value chain backward), each ("exact",), ("segmented", segments, warmup) or ("rows",)
        self.chain_plan = chain_plan
        self.reward_plan = reward_plan  # the same for the reward chain (forward only)
        # calls the chains walked in full (= the number of calls unless the later calls' replays were left out)
        self.full_calls = full_calls

    def check(self):
        """Raises if a chain kernel gave up waiting for a peer workgroup.  Synchronises unless the step has
        already read the status words back (it does whenever it checks a backward seam)."""
        st = self.status_host if self.status_host is not None else self.status.cpu().tolist()
        if any(st):
            raise RuntimeError("a persistent chain kernel timed out (status %s)" % list(st))
        return self


def _runs_beside(main, cand, cycles=4000000):
    """True if a kernel on `cand` overlaps one on `main` (the current stream): two spins of `cycles`
    clock ticks (milliseconds, so that the cross-stream event costs do not matter), one per stream,
    take about as long as one."""
    if not hasattr(torch.cuda, "_sleep"):   # no spin kernel to probe with: take the candidate as it is
        return True

    def spin_pair(second):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(main)
        with torch.cuda.stream(main):
            torch.cuda._sleep(cycles)
            if second is None:
                torch.cuda._sleep(cycles)
        if second is not None:
            with torch.cuda.stream(second):
                second.wait_event(t0)
                torch.cuda._sleep(cycles)
                done = torch.cuda.Event()
                done.record(second)
            main.wait_event(done)
        t1.record(main)
        t1.synchronize()
        return t0.elapsed_time(t1)

    # the first timing in a process comes out up to a third too long (4.5 ms for two 1.7 ms spins), which is exactly
    # what makes a candidate that shares main's queue look as if it overlapped: one untimed pass, then the shorter
    # of two measurements each
    spin_pair(None)
    serial = min(spin_pair(None), spin_pair(None))
    return min(spin_pair(cand), spin_pair(cand)) < 0.75 * serial


class _WarmupLadder:
    """Where one chain (value forward, reward forward, value backward) starts in the warm-up ladder.

    Up: when a length is rejected; and before that can happen, when the seam error (as a fraction of
    its tolerance) is above one half, or would pass 1 within two more minibatches at the rate it has
    been growing from minibatch to minibatch at a fixed length (the parameters drift slowly, and while a
    network's memory is getting longer the error at a fixed length grows by a steady factor per update).
    Down: the error falls geometrically with the warm-up length until only rounding is left, at about a
    tenth of the tolerance; after enough accepted minibatches in a row at that floor (under 15 %) the
    next one probes one rung down.  A failed probe costs a redone minibatch and a successful one saves
    the rung's steps on every later minibatch, so short rungs wait longer (6 / 24 / 64 minibatches for
    rungs of >= 64 / >= 32 / fewer steps), and a probe that does not hold is not repeated for a while
    (32 minibatches, doubling up to 1024)."""

    def __init__(self, warmups, first=0):
        self.warmups = tuple(warmups)
        self.first, self.calm, self.cooldown, self.penalty, self.probing = first, 0, 0, 32, False
        self.prev, self.growth = None, 1.0

    def _calm_needed(self):
        if self.first == 0 or self.first >= len(self.warmups):
            return 6
        gain = self.warmups[self.first] - self.warmups[self.first - 1]
        return 6 if gain >= 64 else 24 if gain >= 32 else 64

    EXACT_RETRY = 16   # minibatches on the exact kernels before the longest warm-up is tried again

    def accepted(self, idx, checks, last):
        if idx != self.first:
            return
        if not checks:
            # the exact kernels ran: nothing was measured.  They cost many times the longest rung, so that rung
            # is tried again now and then (a failed try lengthens the wait like any failed probe)
            if idx >= len(self.warmups) > 0:
                self.cooldown = max(0, self.cooldown - 1)
                self.calm += 1
                if self.calm >= self.EXACT_RETRY and self.cooldown == 0:
                    self.first, self.calm, self.probing = len(self.warmups) - 1, 0, True
            return
        worst = max(e / t if t > 0 else float("inf") for e, t in checks)
        if self.prev is not None and self.prev[0] == idx and self.prev[1] > 0.1:
            self.growth = 0.5 * self.growth + 0.5 * min(max(worst / self.prev[1], 0.5), 3.0)
        self.prev = (idx, worst)
        ahead = max(1.0, self.growth) ** 2
        self.cooldown = max(0, self.cooldown - 1)
        if worst > 0.5 or worst * ahead > 1.0:
            self.first, self.calm = min(idx + 1, last), 0
            if self.probing:
                self._probe_failed()
        else:
            self.probing = False
            self.calm = self.calm + 1 if worst < 0.15 else 0
            if self.calm >= self._calm_needed() and self.cooldown == 0 and self.first > 0:
                self.first, self.calm, self.probing = self.first - 1, 0, True

    def rejected(self, new_idx):
        if self.probing:
            self._probe_failed()
        self.first, self.calm, self.prev = new_idx, 0, None

    def _probe_failed(self):
        self.probing, self.cooldown, self.penalty = False, self.penalty, min(2 * self.penalty, 1024)


class A2CEngine:
    SEAM_TOL_FWD = 1e-5   # max |state_warm - state_true| / max(1, |state_true|) over h and c
    SEAM_TOL_BWD = 1e-4   # max |d_warm - d_true| / max |d_true| over dh and dc
    REPLAY_TOL = 1e-5     # a replay that was left out vs the trajectory it must have met: same measure as SEAM_TOL_FWD
    COLLAPSED_MAX_SEGMENTS = 256   # at most one chain workgroup per CU, so that the replay check's fit beside them

    def __init__(self, policy_params, value_params, reward_params, policy_grads=None, value_grads=None,
                 reward_grads=None,
                 concurrent=True, chain_mode="auto", segments=None, persistent=True, entropy=False,
                 chain_semantics="batch", collapse_replays=True, max_segments=512, policy_dtype="f32",
                 frozen_reward=False, warm16=True, warm16_backward=False, grad_products="f16x2", late_policy_bptt=True,
                 chain_arith="f16x2", rollout_arith="f16x2", policy_bptt_first=True,
                 warmups=(48, 64, 80, 96, 112, 128, 144, 160, 192, 224, 256, 288, 320, 384, 448, 512, 640, 768, 1024,
                          1280, 1536, 2048, 2560, 3072, 4096, 6144, 8192), first_warmup=96):
        """*_params: dict state_dict-name -> contiguous float32 CUDA tensor (the live
        parameter storage).  *_grads: same keys -> gradient tensors to fill; parameters
        missing from the grads dicts (e.g. frozen pretrained embeddings) get no gradient."""
        self.P, self.V, self.R = policy_params, value_params, reward_params
        self.gP, self.gV, self.gR = policy_grads or {}, value_grads or {}, reward_grads or {}
        self.concurrent = concurrent
        self.want_entropy = entropy  # per-(step,row) policy entropy: a diagnostic, not part of the loss
        self.persistent = persistent  # persistent chain kernels (one launch for all steps) when the device can hold them
        assert chain_mode in ("auto", "exact", "segmented")
        assert chain_semantics in ("batch", "per_sample")
        self.per_sample = chain_semantics == "per_sample"
        self.chain_mode, self.segments, self.warmups = chain_mode, segments, tuple(warmups)
        self.fallbacks = 0  # how many times a seam or replay check rejected an attempt
        self.collapse = collapse_replays and not self.per_sample
        # arithmetic of the policy network's products (models.py:75, 80, 82 and their backward): "f32" = exact
        # float32 matrix instruction (default; greedy ids bit-exact), "bf16" = operands rounded to bf16 on chip,
        # float32 sums.  The chains, heads, softmax / sampler and every reduction stay float32 either way.
        assert policy_dtype in ("f32", "bf16")
        self.pd = policy_dtype
        # the backward pass's products over all (step, row) positions - the two vocabulary-sized ones (G0 . W and
        # G0^T . Hs, two fifths of the step's GEMM work) and the recurrent layers' weight and input gradients:
        # "f32x3" forms them from bf16 pieces of the operands at f32 rounding-level accuracy (hipabi.gemm); what
        # the forward pass computes (tokens, log-probabilities, values, rewards, the loss) stays on the exact kernel
        # "f16x2": the same products from two f16 pieces of the operands, each row (as the product sees it) scaled by its
        # own power of two: half the matrix instructions of "f32x3", 1.4-1.5 times its rate, same accuracy class
        assert grad_products in ("f32", "f32x3", "f16x2")
        self.gd = grad_products                                               # value network
        self.vgd = grad_products if policy_dtype == "f32" else policy_dtype   # policy network
        # frozen_reward: the caller promises not to write to the reward network's parameters between steps
        # without calling parameters_changed(); what depends on them alone is then computed once
        self.frozen_reward, self._param_epoch = bool(frozen_reward), 0
        self.max_segments = int(max_segments)
        # the first part of the forward warm-ups at 16-bit split precision (see _cheap_steps)
        self.warm16, self._warm16_off, self._warm16_penalty = bool(warm16), 0, 32
        self.last_cheap = (0, 0)     # 16-bit warm-up steps (value, reward chain) of the last forward pair launch
        self.last_cheap_bwd = 0      # 16-bit warm-up steps of the last value-chain BPTT launch
        # off unless asked for: alone the 16-bit backward launch walks a warm-up step in 8.9 us against the float32
        # launch's 11.5, but it holds every CU's registers, and inside the A2C step, where the policy network's
        # backward pass shares the CUs with the chain, the step gets slower with it (soak 10.9 -> 12.7 ms)
        self.warm16_backward = bool(warm16_backward)
        # recurrent products of the persistent chain launches (both forward chains, the value chain's BPTT, the policy
        # LSTM's BPTT): "f16x2" = every step on the f16 matrix
        # instruction, both operands cut into two scaled f16 pieces (float32-level accuracy: 2^-21 per element product;
        # W_hh in registers, 16-unit tiles), "f32" = the exact float32 instruction (with the 16-bit warm-up part, if on)
        assert chain_arith in ("f32", "f16x2")
        self.chain_arith = chain_arith
        # the rollout's per-round product h . [linear2vocab.weight ; lstm.weight_hh]^T (sampled and teacher-forced
        # rollouts, float32 policy arithmetic): "f16x2" = weights cut once per minibatch into two scaled f16 pieces,
        # h cut by the sampler launch that produces it, three piece products per element product (float32
        # rounding-level accuracy; a third of the exact kernel's time), "f32" = the exact float32 instruction.
        # Greedy decoding always takes the exact kernel.
        assert rollout_arith in ("f32", "f16x2")
        self.rollout_arith = rollout_arith
        # the policy LSTM's backward-through-time as one persistent launch on the main stream, in front of the value
        # chain's or behind it (policy_bptt_first; see _tail_ok)
        self.late_policy_bptt = bool(late_policy_bptt)
        self._warm16b_off, self._warm16b_penalty = 0, 32
        # positions a replay must have before it is left out (raised when a replay check fails; decays again)
        self._replay_need, self._replay_cool, self._replay_penalty = 0, 0, 32
        self.last_replay = None      # (full_calls, steps, err value chain, err reward chain) of the last step
        self._policy_give_ups = 0    # minibatches in a row whose one-launch policy BPTT gave up
        self.launch_timeouts = 0     # attempts repeated because a persistent launch gave up waiting for a workgroup
        self._check = None           # third stream: loss-independent policy backward, vocabulary gradients
        self._replay = None          # fourth stream (if one overlaps the others): the replay check
        self.trace = None  # set to [] to collect (label, stream-name, event) marks for phase timing
        self.host_trace = None  # set to [] beside it to collect (label, host clock) at the same points
        # on_policy_grads(event): called once per step as soon as the forward pass has been accepted, with
        # the event after which the policy network's gradients are complete; a data-parallel caller starts
        # their all-reduce there, beside the value chain's BPTT
        self.on_policy_grads = None
        self._ev_policy = None
        # streams the engine uses, the caller's included: 3 (main, side, third), or 4 with a stream of its own for the
        # replay check (see _replay_stream); side_stream() hands the side stream to a caller that wants its
        # collective queued there instead of on yet another stream
        self.max_streams = 3
        # order of the two BPTT launches when the policy LSTM's is one launch: the policy LSTM's first (0.2 ms alone on
        # the device), so that the weight gradients of BOTH networks' other layers and the replay check have the whole
        # of the value chain's BPTT (1.3 ms on half the CUs) to run beside (see _value_chain_backward)
        self.policy_bptt_first = bool(policy_bptt_first)
        # every chain starts at the first rung of at least first_warmup steps (freshly initialised networks
        # pass there); shorter rungs are reached only by probing down
        start = next((i for i, w in enumerate(self.warmups) if w >= first_warmup), len(self.warmups) - 1)
        self._ladder_fv, self._ladder_fr, self._ladder_bwd, self._ladder_rb = (
            _WarmupLadder(self.warmups, start) for _ in range(4))   # value fwd, reward fwd, value bwd, reward bwd
        self.last_seam = None
        self._ws = {}
        self._side = None
        # any subset of the three networks may be bound (the pre-training steps use one at a time)
        self.P, self.V, self.R = self.P or None, self.V or None, self.R or None
        groups = [g for g in (self.P, self.V, self.R) if g is not None]
        assert groups, "no parameters given"
        self.device = next(iter(groups[0].values())).device
        self.vocab, self.E, self.F = None, None, None
        if self.P is not None:
            self.vocab, self.E = self.P["caption_embedding.weight"].shape
            self.F = self.P["cnn2linear.weight"].shape[1]
            assert self.P["cnn2linear.weight"].shape[0] == H
        if self.V is not None:
            f = self.V["linear1.weight"].shape[1] - H
            assert self.V["linear1.weight"].shape[0] == 512 and (self.F is None or f == self.F), \
                "value head width != feature_dim + 512"
            self.F = f
        if self.R is not None:
            assert self.R["visual_embed.weight"].shape[0] == 512
            assert self.F is None or self.R["visual_embed.weight"].shape[1] == self.F
            self.F = self.R["visual_embed.weight"].shape[1]
        for group in groups:
            for k, t in group.items():
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), k

    def parameters_changed(self):
        """Tell the engine that parameters it was told are frozen have been written to (a checkpoint was
        loaded, say): everything cached from them is rebuilt on the next step."""
        self._param_epoch += 1

    # ------------------------------------------------------------------ workspaces
    MIN_REPLAY = 128      # shortest replay, in positions, that is ever left out
    MAX_SHAPES = 24       # problem shapes whose workspaces are kept (the least recently used one is dropped)

    def _buffers(self, n, p0, s):
        key = (n, p0, s)
        ws = self._ws.pop(key, None)
        if ws is not None:
            self._ws[key] = ws            # most recently used last
            return ws
        while len(self._ws) >= self.MAX_SHAPES:
            self._ws.pop(next(iter(self._ws)))
        dev = self.device
        t_steps, l_cols, lv = p0 - 1 + s, p0 + s, p0 + s - 1
        tv = hip.chain_positions(n, p0, s)

        def f32(*shape):
            return torch.empty(*shape, dtype=torch.float32, device=dev)

        ws = dict(tokT=torch.zeros(l_cols, n, dtype=torch.int64, device=dev), tv=tv, cap_v=0, cap_r=0,
                  scalars=torch.zeros(8, dtype=torch.float32, device=dev),
                  # seam errors [0:6], replay errors [6:8], status words [8:12]
                  ctl=torch.zeros(12, dtype=torch.float32, device=dev),
                  seam_host=torch.zeros(4, dtype=torch.float32).pin_memory(),
                  replay_host=torch.zeros(2, dtype=torch.float32).pin_memory(),
                  ctl_host=torch.zeros(8, dtype=torch.float32).pin_memory())
        ws["seam"], ws["status"] = ws["ctl"][0:6], ws["ctl"][8:12].view(torch.int32)
        widths = [H]
        if self.P is not None:
            v, e = self.vocab, self.E
            widths.append(e)
            ws.update(
                Hbuf=f32(t_steps + 1, n, H), Cbuf=f32(t_steps + 1, n, H), gates=f32(t_steps, n, 4 * H),
                X=f32(t_steps, n, e), logits=f32(n, v), logp=f32(s, n), entropy=f32(s, n),
                dlogp=f32(s * n), dlogits=f32(s, n, v),
                dHall=f32(t_steps, n, H), dgp=f32(t_steps, n, 4 * H), dc0=f32(n, H), dc1=f32(n, H), dh0=f32(n, H),
                dX=f32(t_steps * n, e), Uvocab=f32(s * n, H), HsC=f32(s * n, H))
            ws["Cbuf"][0].zero_()
        if self.V is not None:
            ev = self.V["valrnn.caption_embedding.weight"].shape[1]
            widths.append(ev)
            ws.update(
                Xv=f32(lv * n, ev), XPv=f32(lv * n, 4 * H), hv=f32(s, n, H), U=f32(n, 512),
                Y=f32(s * n, 512), values=f32(s * n), dvalue=f32(s * n),
                sync_v=torch.zeros(hip.chain_sync_bytes(), dtype=torch.uint8, device=dev),
                dY=f32(s * n, 512), dYsum=f32(n, 512), gw1h=f32(512, H), d_emit=f32(s, n, H),
                dGsum=f32(lv * n, 4 * H), dXv=f32(lv * n, ev))
        if self.R is not None:
            ws.update(
                XPr=f32(l_cols * n, 3 * H), hr=f32(s, n, H), se=f32(s * n, 512), ve=f32(n, 512), rewards=f32(s * n),
                sync_r=torch.zeros(hip.chain_sync_bytes(), dtype=torch.uint8, device=dev))
        # split-K partial slabs: up to 16 of a [4H,H] weight gradient, 8 of the vocabulary projection's,
        # 5 of an activation gradient
        scratch = max(16 * 4 * H * H, 8 * (self.vocab or 0) * H, 5 * l_cols * n * max(widths))
        ws["scratch_main"], ws["scratch_side"], ws["scratch_check"] = f32(scratch), f32(scratch), f32(scratch)
        # fixed-order reductions (bias gradients, embedding gradients): the same bits on every run
        cs = max(2048 * 64 + max(self.vocab or 0, 4 * H), n * 512)
        ws["cs_main"], ws["cs_side"], ws["cs_check"] = f32(cs), f32(cs), f32(cs)
        vmax = self._emb_vocab() if (self.P is not None or self.V is not None) else 1
        ws["emb_work"] = torch.zeros(hip.embedding_grad_work_ints(vmax, l_cols * n), dtype=torch.int32, device=dev)
        k = n if self.per_sample else self._segments_for(tv, tv)
        if k:
            ws["work_v"] = f32(hip.chain_seg_workspace_floats(k))
            ws["work_r"] = f32(hip.chain_seg_workspace_floats(k))
            if self.collapse and s > 1:   # the replay check's rows: one per call past the first
                ws["work_cv"] = f32(hip.chain_seg_workspace_floats(s))
                ws["work_cr"] = f32(hip.chain_seg_workspace_floats(s))
        ws["segments"] = k
        # scale of an operand bounded by 1 in the two-piece products (see _unit_scale): filled here, on the caller's
        # stream before the step's other streams branch off, because products on several streams read it
        ws["sc16_unit"] = torch.full((max(4096, 4 * H),), 16384.0, dtype=torch.float32, device=dev)
        if s == 1 or not self.collapse:
            self._chain_capacity(ws, tv, 0)
        self._ws[key] = ws
        return ws

    def _chain_capacity(self, ws, positions_v, positions_r):
        """Per-position buffers of the chains (the value chain's kept activations and gate gradients, the
        reward chain's kept activations when the replay check needs them), sized for the chain that is
        actually walked and grown when more calls have to be walked in full."""
        dev = self.device
        if self.V is not None and positions_v > ws["cap_v"]:
            ws["saved"] = ws["dgv"] = None
            ws["saved"] = torch.empty(positions_v, 6, H, dtype=torch.float32, device=dev)
            ws["dgv"] = torch.empty(positions_v, 4 * H, dtype=torch.float32, device=dev)
            ws["cap_v"] = positions_v
        if self.R is not None and positions_r > ws["cap_r"]:
            ws["saved_r"] = None
            ws["saved_r"] = torch.empty(positions_r, 5, H, dtype=torch.float32, device=dev)
            ws["cap_r"] = positions_r

    def _segments_for(self, positions, full_positions=None):
        """Number of segments for a chain of `positions` steps; 0 = use the exact kernels (decided on the
        length the chain has when every call is walked in full)."""
        full_positions = positions if full_positions is None else full_positions
        if self.chain_mode == "exact" or (self.chain_mode == "auto" and full_positions < 4096):
            return 0
        if self.segments:
            return int(self.segments)
        # a step costs about the same for 32 segments as for 512 (fixed per-step time dominates until the
        # device is full), so short chains are cut finer: fewer, fuller steps
        return max(32, min(self.max_segments, positions // 16 // 32 * 32))

    def _plans(self, ws):
        if self.per_sample:
            return [("rows",)]
        k = ws["segments"]
        plans = [("segmented", k, w) for w in self.warmups] if k else []
        return plans + [("exact",)]

    WARM16_MIN_WARMUP = 96   # shorter warm-ups gain nothing: the extra launch costs what its steps save (at 96: 7.13-7.29 against 7.37-7.42 ms per step)
    WARM16_TAIL = 16   # float32 steps that end a warm-up whose first part ran at 16-bit precision (at least W / 8)

    @staticmethod
    def _sudden(ladder, idx, ok):
        """Whether a seam check that failed (ok False) at ladder position idx came out of the blue: the last accepted
        minibatch ran the same warm-up with its error under a quarter of the tolerance.  A failure that was
        building up is the recurrence's memory growing, which is the ladder's business."""
        return ok or (ladder.prev is not None and ladder.prev[0] == idx and ladder.prev[1] < 0.25)

    def _note_cheap(self, plan, plan_r):
        self.last_cheap = self._cheap_steps(plan[1], plan[2], plan_r[2])
        return self.last_cheap

    def _warm16_active(self, pv, pr):
        if self.chain_arith == "f16x2" and pv[0] == "segmented" and hip.chain_f16x2_supported(pv[1]):
            return False   # the f16 launch walks every step alike: no 16-bit part to switch off
        return pv[0] == "segmented" and pr[0] == "segmented" and any(self._cheap_steps(pv[1], pv[2], pr[2]))

    def _cheap_steps(self, segments, w_v, w_r):
        """(value, reward): how many of each forward chain's warm-up steps run at 16-bit split precision
        (icrl_chain_pair_fwd_seg_warm16) before the float32 launch; a float32 tail of max(WARM16_TAIL, W / 8)
        steps washes their 2^-16 out.  Switched off for a while (32 minibatches, doubling) after a forward seam
        was rejected with it in use, so that a network it does not suit costs a redone minibatch now and then,
        not the warm-up ladder."""
        if (not self.warm16 or self._warm16_off > 0 or max(w_v, w_r) < self.WARM16_MIN_WARMUP
                or not hip.chain_warm16_supported(segments)):
            return 0, 0
        cut = lambda w: max(0, w - max(self.WARM16_TAIL, w // 8))
        return cut(w_v), cut(w_r)

    WARM16_BWD_TAIL = 8   # float32 steps that end a backward warm-up whose first part ran at 16-bit precision (at least W / 16)

    def _cheap_bwd(self, plan):
        """How many of the value chain's backward warm-up steps run at 16-bit split precision
        (icrl_lstm_chain_bwd_seg_warm16) before the float32 launch.  The backward seam tolerance (1e-4 of the
        largest state) is well above their 2^-16, so a short float32 tail is enough.  Switched off for a while
        after a backward seam was rejected out of the blue with it in use."""
        if (plan[0] != "segmented" or not self.warm16_backward or self._warm16b_off > 0
                or plan[2] < self.WARM16_MIN_WARMUP
                or not (self.persistent and hip.chain_pair_supported(plan[1]) and hip.chain_bwd_warm16_supported(plan[1]))):
            return 0
        return max(0, plan[2] - max(self.WARM16_BWD_TAIL, plan[2] // 16))

    def _replay_steps(self, w_fwd):
        """Steps of a left-out replay that the check walks: the longest forward warm-up in use (the seams say
        that within it even a zero state is forgotten; the replay starts from a state of the trajectory).  The
        check measures, so any length is safe; a short one only risks asking for more calls in full than
        needed.  It runs after the forward chains and, at long warm-ups, longer than the backward pass beside
        it, so it is not made longer than that."""
        return max(self.MIN_REPLAY, w_fwd)

    def _full_calls(self, n, p0, s, w_fwd, w_bwd):
        """How many calls the chains walk in full: the fewest for which every replay left out is at least
        as long as the steps the replay check walks, as the backward warm-up (a gradient that a seam check
        shows to be gone after `warmup` steps does not reach through a replay that long), as the floor
        MIN_REPLAY and as what failed replay checks have asked for."""
        if not self.collapse or s < 2:
            return s
        need = max(self._replay_need, self._replay_steps(w_fwd), w_bwd)
        for fc in range(1, s):
            if (p0 + fc - 1) * n >= need:
                # not worth a different geometry unless it removes a good part of the chain
                return fc if hip.chain_positions(n, p0, s, fc) * 4 <= hip.chain_positions(n, p0, s) * 3 else s
        return s

    def _choose(self, ws, n, p0, s, fv, fr, bi):
        """(full_calls, value plan, reward plan, backward plan) for the ladder positions fv, fr, bi."""
        if self.per_sample:
            return s, ("rows",), ("rows",), ("rows",)
        last = len(self.warmups)
        if not ws["segments"]:
            return s, ("exact",), ("exact",), ("exact",)
        if fv >= last:   # a value chain that no warm-up reproduced forward is not tried segmented backward
            bi = last
        idx = [i for i in (fv, fr, bi) if i < last]
        fc = s if len(idx) < 3 else self._full_calls(n, p0, s, max(self.warmups[fv], self.warmups[fr]), self.warmups[bi])
        k = self._segments_for(hip.chain_positions(n, p0, s, fc), ws["tv"])
        if fc < s and not self.segments:
            k = min(k, self.COLLAPSED_MAX_SEGMENTS)
        plan = lambda i: ("segmented", k, self.warmups[i]) if i < last else ("exact",)
        # (Measured and dropped: half as many segments for the forward pair once its warm-up is long.  Alone the launch
        # gains - a step costs 4.8 us with one workgroup per CU against 5.9 with two: warm-up 256, 1.62 against 1.73 ms -
        # but with one workgroup per CU the products queued behind it run beside it, and the step loses: 7.31-7.41
        # against 7.25 ms at steady state.)
        return fc, plan(fv), plan(fr), plan(bi)

    def _bwd_f16x2(self, rows):
        """Whether a backward chain over `rows` segments (or rows) runs as the owner-computes f16 launch."""
        return self.chain_arith == "f16x2" and self.persistent and hip.chain_bwd_f16x2_supported(rows)

    def _tail_ok(self, n, pv, pr, pb):
        """Whether this attempt runs the backward pass with the policy LSTM's BPTT and the value chain's BPTT as two
        persistent launches on the main stream (the policy LSTM's first unless policy_bptt_first is off), the weight
        gradients beside and behind the value chain's, and with everything of the policy network's
        backward pass that needs no BPTT (G0 . W, the vocabulary gradients) formed beside the value chain's BPTT
        instead of beside the forward chains.  As 3 T dependent launches on a stream of its own the policy LSTM's
        BPTT took as long as the value chain's (19 steps at 120 us, launch latency, not work) and its weight
        gradients then ended the step; as T + 1 steps of one launch it takes 0.3 ms, and the forward chains and
        the heads have the device to themselves.  Needs the persistent launches (all segmented plans, and N rows
        of the policy launch resident at once)."""
        return (self.late_policy_bptt and self.concurrent and self.persistent and not self.per_sample
                and self.P is not None and pv[0] == pr[0] == pb[0] == "segmented" and hip.chain_pair_supported(pv[1])
                and hip.chain_pair_supported(n))

    @staticmethod
    def _next_plan(plans, attempt, checks):
        """After a rejected attempt: the seam error shrinks geometrically with the warm-up length
        (from an O(1) initial mismatch), so estimate the length that reaches a quarter of the
        tolerance and jump to the first plan at least that long.  checks: [(error, tolerance)]."""
        import math
        w = plans[attempt][2]
        need = w + 1
        for err, tol in checks:
            if err > tol and tol <= 0:
                return len(plans) - 1
            if err > tol and 0 < err < 0.5:
                need = max(need, int(math.ceil(w * math.log(tol / 4.0) / math.log(err))))
            elif err >= 0.5 or err != err:
                need = max(need, 4 * w)
        for idx in range(attempt + 1, len(plans)):
            if plans[idx][0] == "exact" or plans[idx][2] >= need:
                return idx
        return len(plans) - 1

    def _next_rung(self, attempt, checks):
        """_next_plan on the warm-up ladder alone (index len(warmups) = the exact kernels)."""
        plans = [("segmented", 0, w) for w in self.warmups] + [("exact",)]
        return self._next_plan(plans, attempt, checks)

    def _mark(self, label):
        if self.trace is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self.trace.append((label, ev))
            if self.host_trace is not None:   # when the HOST got here (it queues ahead of the device, or falls behind)
                import time
                self.host_trace.append((label, time.perf_counter()))

    def _side_stream(self):
        """A second stream that really runs beside the current one.  HIP multiplexes its streams onto a
        few hardware queues (4 per process by default), and two streams that share a queue run one
        after the other; which streams collide depends on what else the process has created (an RCCL
        communicator brings several), so candidates are tried until one overlaps."""
        if self._side is None:
            self._side = self._stream_beside([torch.cuda.current_stream()])
        return self._side

    def side_stream(self):
        """The engine's second stream (None when it runs everything on the caller's): the policy network's gradients
        are completed there, so a collective on them that is queued on it needs no stream of its own."""
        return self._side_stream() if self.concurrent else None

    def _replay_stream(self):
        """A fourth stream for the replay check, a persistent launch of 1.5 ms that would otherwise hold the third
        stream's queue while the vocabulary gradients wait behind it (they then start 1.4 ms earlier, beside the
        value chain's BPTT instead of beside the weight gradients at the end of the step: 8.56-8.79 -> 8.47-8.48 ms).
        Only if a stream is found that really runs beside the other three (HIP has four hardware queues by
        default, and other libraries' streams share them); the third stream otherwise."""
        if self._replay is None:
            if self.max_streams < 4:
                # on the side stream, which is idle between the reward head and the vocabulary gradients: those then
                # start behind the check, 1.5 ms into the value chain's BPTT instead of with it, and the BPTT - which
                # every product beside it slows down - runs 1.8 ms instead of 2.4 (6.7-6.8 ms per step against 7.2
                # with a stream of the check's own).  It also leaves a hardware queue (HIP gives a process four) to
                # the streams of a process group
                self._replay = self._side_stream()
            else:
                others = [torch.cuda.current_stream(), self._side_stream(), self._check_stream()]
                self._replay = self._stream_beside(others, strict=True) or self._check_stream()
        return self._replay

    def _check_stream(self):
        """A third stream, beside both the current one and the side stream (the replay check)."""
        if self._check is None:
            self._check = self._stream_beside([torch.cuda.current_stream(), self._side_stream()])
        return self._check

    def _stream_beside(self, others, strict=False):
        """A new stream that overlaps every stream in `others`; if none of eight candidates does, the first one
        (strict: None)."""
        first = None
        for attempt in range(8):
            cand = torch.cuda.Stream(device=self.device)
            first = first or cand
            if all(_runs_beside(o, cand) for o in others):
                return cand
        return None if strict else first

    TABLE_ROWS_FACTOR = 16   # the table is built beside the rollout: worth it down to a sixteenth of the vocabulary in (step, row) pairs (batch 64 at vocabulary 10 000: 5.84 -> 5.56 ms)
    NEXT_CELL_FROM = 1   # first rollout step whose cell update is done by the previous step's sampler launch

    def _policy_table_wanted(self, n, p0, s, mode):
        """Whether the rollout takes the policy LSTM's input side from a per-word table and its cell update from
        the sampler launch: a sampled or forced rollout, and enough (step, row) pairs that a table over the whole
        vocabulary, built off the critical path, does not cost many times the products it replaces."""
        t_steps = p0 - 1 + s
        return mode != hip.GREEDY and self.vocab % 4 == 0 and s >= 4 and self.TABLE_ROWS_FACTOR * t_steps * n >= self.vocab

    def _padded_vocab(self):
        return (self.vocab + 63) // 64 * 64

    def _rollout_f16x2(self):
        return self.rollout_arith == "f16x2" and self.pd == "f32" and self.E % 16 == 0

    def _policy_table(self, ws, stacked_on=None, after=None):
        """Per-word input side of the policy LSTM, table[w] = W_ih emb[w] + b_ih + b_hh (models.py:73, 80), and
        the vocabulary projection stacked on W_hh, so that one product gives a step's logits and the recurrent
        half of the next step's gates.  They depend on the parameters only: built beside the first rollout steps.
        The vocabulary's rows are padded with zero rows to a multiple of 64 (the gate rows are one), so that the
        product has no ragged tile column: its workgroups take the loop with bounds checks and finish last (66 against
        60 us per rollout step at 10 000 words)."""
        P, V = self.P, self.vocab
        Vp = self._padded_vocab()
        if "table_p" not in ws:
            dev = self.device
            ws["table_p"] = torch.empty(V, 4 * H, dtype=torch.float32, device=dev)
            ws["Wcat"] = torch.zeros(Vp + 4 * H, H, dtype=torch.float32, device=dev)
            ws["bcat"] = torch.zeros(Vp + 4 * H, dtype=torch.float32, device=dev)
            ws["logits_ext"] = torch.empty(ws["logits"].shape[0], Vp + 4 * H, dtype=torch.float32, device=dev)
            if self._rollout_f16x2():
                rows = ws["logits"].shape[0]
                ws["Wcat_p"], ws["Wcat_d"] = hip.f16x2_pieces(Vp + 4 * H, H, self.device)
                ws["h_p"], ws["h_d"] = hip.f16x2_pieces(rows, H, self.device)
                ws["h_d"].fill_(1.0 / 4096.0)
        # the stacked weight on `stacked_on` (a stream beside the current one: both are on the path to the rollout's
        # first round, 0.05 and 0.13 ms), the table on the current stream
        with torch.cuda.stream(stacked_on if stacked_on is not None else torch.cuda.current_stream()):
            if stacked_on is not None and after is not None:
                stacked_on.wait_event(after)
            ws["Wcat"][:V].copy_(P["linear2vocab.weight"])
            ws["Wcat"][Vp:].copy_(P["lstm.weight_hh_l0"])
            ws["bcat"][:V].copy_(P["linear2vocab.bias"])
            if self._rollout_f16x2():
                hip.cut_f16x2(ws["Wcat"], ws["Wcat_p"], ws["Wcat_d"])
            # (the rounds before the table is ready use the vocabulary rows of the pieces already)
            ws["ev_wcat"] = torch.cuda.Event()
            ws["ev_wcat"].record(torch.cuda.current_stream())
        if self._rollout_f16x2():
            self._word_table_f16x2(ws, "p", P["caption_embedding.weight"], P["lstm.weight_ih_l0"], P["lstm.bias_ih_l0"],
                                   P["lstm.bias_hh_l0"], ws["table_p"])
        else:
            hip.gemm(P["caption_embedding.weight"], P["lstm.weight_ih_l0"], ws["table_p"], bias=P["lstm.bias_ih_l0"],
                     bias2=P["lstm.bias_hh_l0"], dtype=self.pd)

    def _word_table_f16x2(self, ws, tag, emb, w_ih, bias, bias2, out):
        """out[w] = W_ih emb[w] + bias (+ bias2) for every word on the f16 matrix instruction: both matrices are cut
        into two scaled f16 pieces (each row by its own power of two) and multiplied by hipabi.gemm_f16x2, float32
        rounding-level accuracy at half the exact kernel's time (a table is 21 GFLOP, built beside the first rollout
        rounds, whose launches it slows for as long as it runs)."""
        key = "tab16_" + tag
        if key not in ws:
            ws[key] = (hip.f16x2_pieces(emb.shape[0], emb.shape[1], self.device),
                       hip.f16x2_pieces(w_ih.shape[0], w_ih.shape[1], self.device),
                       torch.empty_like(bias))
        (ep, ed), (wp, wd), bsum = ws[key]
        hip.cut_f16x2(emb, ep, ed)
        hip.cut_f16x2(w_ih, wp, wd)
        if bias2 is not None:
            torch.add(bias, bias2, out=bsum)
        hip.gemm_f16x2(ep, ed, wp, wd, out, emb.shape[1], bias=bsum if bias2 is not None else bias)

    def _rollout(self, ws, features, n, p0, s, uniforms, mode, ev_table=None, between=None):
        """Policy rollout (models.py:71-84 run incrementally; trainers.py:443-458): LSTM state carried
        from column to column, only the newest position projected to the vocabulary and sampled (or
        arg-maxed, or scored against the forced token).  With ev_table (the event after which _policy_table's
        outputs are complete) a step is two launches: [logits | h.W_hh] as one product, then the sampler, whose
        workgroups go on to the next cell update of their row with the table row of the token they fixed.
        (Measured and dropped: the rows cut into two groups issued alternately on two streams, 2.30-2.37 ms
        against 2.26-2.33; the rollout on a high-priority stream while the tables are built beside it, 2.65 ms
        against 2.30.)"""
        P = self.P
        t_steps = p0 - 1 + s
        Hb, Cb, gates = ws["Hbuf"], ws["Cbuf"], ws["gates"]
        tokT, V = ws["tokT"], self.vocab
        if not ws.pop("h0_done", False):
            self._rollout_h0(ws, features)
        ws["rollout_id"] = ws.get("rollout_id", 0) + 1   # (what is derived from this rollout's outputs is kept under it)
        fused_step = self.E % 64 == 0 and self.pd == "f32"
        mode_s = mode if mode == hip.GREEDY else mode | hip.EMIT_DLOGITS
        cell_done = ws.pop("cell0_done", False)   # (step() has queued the first cell update already)
        h_cut = False   # the newest h is already in ws["h_p"] (left there by the sampler launch that computed it)
        wcat_ready = False
        k = -1
        while k + 1 < t_steps:
            k += 1
            if not cell_done:
                self._rollout_cell(ws, k, n, fused_step)
            cell_done = False
            if k < p0 - 1:
                continue
            st = k - (p0 - 1)
            # the sampler leaves G0 = onehot(token) - softmax of its row (d log_prob / d logits, all the backward pass
            # needs of the softmax) where a separate pass over the [rows, V] probabilities used to put it
            out = dict(uniforms=None if uniforms is None else uniforms[st * n:(st + 1) * n], log_prob=ws["logp"][st],
                       probs=ws["dlogits"][st] if mode != hip.GREEDY else None,
                       entropy=ws["entropy"][st] if (self.want_entropy and mode != hip.GREEDY) else None)
            if ev_table is not None and self.NEXT_CELL_FROM <= k + 1 < t_steps:
                if ev_table is not True:
                    torch.cuda.current_stream().wait_event(ev_table)
                    torch.cuda.current_stream().wait_event(ws["ev_wcat"])
                    ev_table = True
                ext = ws["logits_ext"][:n]
                if self._rollout_f16x2():
                    if not h_cut:
                        hip.cut_f16x2(Hb[k + 1], ws["h_p"], ws["h_d"], fixed_scale=4096.0)
                    if hip.PROFILE is None:
                        # every remaining round of this kind by one call into the library (the same launches; a host
                        # call per launch, 50 us a round, is more than the two launches take on the device)
                        # (`between`: what the caller wants queued on other streams once the rollout is under way - the
                        # device takes launches up in about the order the host queues them, so work queued before the
                        # rollout delays its first round and work queued after all of it starts when it ends)
                        total = t_steps - 1 - k
                        splits = (min(2, total), total - min(2, total)) if between is not None else (total,)
                        for rounds in splits:
                            if rounds > 0:
                                hip.policy_rollout_rounds(
                                    ws["Wcat_p"], ws["Wcat_d"], ws["bcat"], ws["h_p"], ws["h_d"], ext, n, V, self._padded_vocab(),
                                    rounds, mode_s, tokT[k + 1:k + 1 + rounds], ws["logp"][st:st + rounds], ws["table_p"],
                                    gates[k + 1:k + 1 + rounds], Cb[k + 1:k + 2 + rounds], Hb[k + 2:k + 2 + rounds],
                                    uniforms=None if uniforms is None else uniforms[st * n:(st + rounds) * n],
                                    probs=ws["dlogits"][st:st + rounds] if mode != hip.GREEDY else None,
                                    entropy=ws["entropy"][st:st + rounds] if (self.want_entropy and mode != hip.GREEDY) else None)
                                k, st = k + rounds, st + rounds
                            if between is not None:
                                between()
                                between = None
                        k, cell_done, h_cut = t_steps - 2, True, True
                        continue
                    hip.gemm_f16x2(ws["h_p"], ws["h_d"], ws["Wcat_p"], ws["Wcat_d"], ext, H, bias=ws["bcat"])
                    out["h_pieces"] = ws["h_p"]
                    h_cut = True
                else:
                    hip.gemm(Hb[k + 1], ws["Wcat"], ext, bias=ws["bcat"], tile=3, dtype=self.pd)
                hip.sample_tokens_lstm_cell(ext[:, :V], mode_s, tokT[k + 1], 1, ext[:, self._padded_vocab():], ws["table_p"], gates[k + 1],
                                            Cb[k + 1], Cb[k + 2], Hb[k + 2], **out)
                cell_done = True
            else:
                logits = ws["logits"][:n]
                if ev_table is not None and self._rollout_f16x2() and V % 4 == 0:
                    # (a round without the next cell update: the first ones, while the table is being built, and the last)
                    if not wcat_ready:
                        torch.cuda.current_stream().wait_event(ws["ev_wcat"])
                        wcat_ready = True
                    if not h_cut:
                        hip.cut_f16x2(Hb[k + 1], ws["h_p"], ws["h_d"], fixed_scale=4096.0)
                    hip.gemm_f16x2(ws["h_p"], ws["h_d"], ws["Wcat_p"], ws["Wcat_d"], logits, H, bias=ws["bcat"])
                else:
                    hip.gemm(Hb[k + 1], P["linear2vocab.weight"], logits, bias=P["linear2vocab.bias"], dtype=self.pd)
                h_cut = False
                hip.sample_tokens(logits, mode_s, tokT[k + 1], 1, **out)
        # the embedding rows of the steps that took their input side from the table are gathered when the
        # backward pass asks for them
        ws["X_pending"] = ev_table is not None
        if between is not None:
            between()

    def _rollout_h0(self, ws, features):
        """h0 = cnn2linear(features) (models.py:71): a 64-tile output with a long k, first on the step's critical
        path; k is split over workgroups (partial slabs summed on top of the bias)."""
        Hb = ws["Hbuf"]
        Hb[0].copy_(self.P["cnn2linear.bias"].expand_as(Hb[0]))
        hip.gemm_auto(features, self.P["cnn2linear.weight"], Hb[0], ws["scratch_main"], accumulate=True, dtype=self.pd)

    def _rollout_cell(self, ws, k, n, fused_step):
        """The policy LSTM's step k for all rows (models.py:73, 80): embedding lookup, both gate products, cell."""
        P = self.P
        tokT = ws["tokT"]
        Hb, Cb, gates, X = ws["Hbuf"], ws["Cbuf"], ws["gates"], ws["X"]
        if fused_step:  # gather + both gate products + cell in one launch
            hip.policy_lstm_step(P["caption_embedding.weight"], tokT[k], P["lstm.weight_ih_l0"],
                                 P["lstm.weight_hh_l0"], P["lstm.bias_ih_l0"], P["lstm.bias_hh_l0"], Hb[k], Cb[k],
                                 gates[k], Cb[k + 1], Hb[k + 1], X[k])
        else:
            hip.gather_rows(P["caption_embedding.weight"], tokT[k], 1, n, X[k])
            hip.gemm(X[k], P["lstm.weight_ih_l0"], gates[k], bias=P["lstm.bias_ih_l0"],
                     bias2=P["lstm.bias_hh_l0"], dtype=self.pd)
            hip.gemm(Hb[k], P["lstm.weight_hh_l0"], gates[k], accumulate=True, dtype=self.pd)
            hip.lstm_cell_fwd(gates[k], Cb[k], Cb[k + 1], Hb[k + 1])

    # ------------------------------------------------------------------ pre-training steps
    def policy_ce_step(self, features, captions, valid_steps, row_weight):
        """Teacher-forced cross-entropy step of the policy pre-training loop (trainers.py:228-252):
        captions [N, L] int64 cuda; position t < valid_steps[i] of row i contributes
        -row_weight[i] * log p(captions[i, t+1] | captions[i, :t+1]).  Fills the policy gradients;
        returns the loss as a 1-element device tensor."""
        n, l_cols = captions.shape
        s = l_cols - 1
        ws = self._buffers(n, 1, s)
        # an A2C step on the same shape shares this workspace: none of its pending work belongs to this pass
        ws["policy_early"] = False
        ws.pop("early_pending", None)
        ws.pop("head_grads_pending", None)
        self._forget_scales(ws)
        ws["tokT"].copy_(captions.t())
        self._rollout(ws, features, n, 1, s, None, hip.FORCED)
        self._order_tokens(ws, n, s * n, torch.cuda.current_stream(), torch.cuda.current_stream())
        hip.masked_nll(ws["logp"], valid_steps, row_weight, ws["scalars"][0:1], ws["dlogp"])
        self._policy_backward(ws, features, n, 1, s)
        return ws["scalars"][0:1]

    def value_mse_step(self, features, captions, targets):
        """Mean-squared-error step of the value pre-training loop (trainers.py:175-190): one
        ValueNetwork call on captions [N, p] (its batch-1 chain walked exactly), nn.MSELoss against
        targets [N], gradients of the value network.  Returns (loss 1-element tensor, values [N])."""
        V, gV = self.V, self.gV
        n, p0 = captions.shape
        ws = self._buffers(n, p0, 1)
        self._forget_scales(ws)
        main = torch.cuda.current_stream()
        ws["tokT"][:p0].copy_(captions.t())
        tok_flat = ws["tokT"].view(-1)
        hip.gather_rows(V["valrnn.caption_embedding.weight"], tok_flat, 1, p0 * n, ws["Xv"])
        hip.gemm(ws["Xv"], V["valrnn.lstm.weight_ih_l0"], ws["XPv"], bias=V["valrnn.lstm.bias_ih_l0"],
                 bias2=V["valrnn.lstm.bias_hh_l0"])
        self._order_tokens(ws, n, p0 * n, main, main)
        # the chain starts from zeros (the loop resets the state every minibatch), so a long one runs
        # segment-parallel under the same seam checks and warm-up ladders as the A2C step
        plans = self._plans(ws)
        last = len(plans) - 1
        fv, bi = min(self._ladder_fv.first, last), min(self._ladder_bwd.first, last)
        redo_forward = True
        while True:
            pv = plans[fv]
            if pv[0] == "exact":
                bi = last
            pb = plans[bi]
            if redo_forward:
                if pv[0] == "segmented":
                    hip.lstm_chain_fwd_seg(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], n, p0, 1, pv[1], pv[2], ws["hv"],
                                           ws["saved"], ws["work_v"], ws["seam"][0:2],
                                           persistent=self.persistent and hip.chain_pair_supported(pv[1]))
                else:
                    hip.lstm_chain_fwd(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], n, p0, 1, ws["hv"], ws["saved"],
                                       ws["sync_v"], ws["status"][0:1])
                w1 = V["linear1.weight"]
                hip.gemm(features, w1[:, :self.F], ws["U"], bias=V["linear1.bias"])
                hip.gemm(ws["hv"].view(n, H), w1[:, self.F:], ws["Y"])
                hip.value_head_fwd(ws["Y"], ws["U"], V["linear2.weight"], V["linear2.bias"], n, n, ws["values"])
                hip.mse_loss(ws["values"], targets, ws["scalars"][0:1], ws["dvalue"])
                self._value_head_vector(ws)
                self._value_head_backward(ws, features, n, 1)
            self._value_chain_backward(ws, n, p0, 1, 1, main, main, pb, None)
            seg_v, seg_b = pv[0] == "segmented", pb[0] == "segmented"
            if not (seg_v or seg_b):
                break
            seam = ws["seam"].tolist()
            self.last_seam = seam
            checks_v = [(seam[0], self.SEAM_TOL_FWD)] if seg_v else []
            checks_b = [(seam[4] / seam[5] if seam[5] > 0 else 0.0, self.SEAM_TOL_BWD)] if seg_b else []
            ok_v, ok_b = all(e <= t for e, t in checks_v), all(e <= t for e, t in checks_b)
            if ok_v and ok_b:
                self._ladder_fv.accepted(fv, checks_v, last)
                self._ladder_bwd.accepted(bi, checks_b, last)
                break
            self.fallbacks += 1
            redo_forward = not ok_v
            if not ok_v:
                fv = self._next_plan(plans, fv, checks_v)
                self._ladder_fv.rejected(fv)
            if not ok_b:
                bi = self._next_plan(plans, bi, checks_b)
                self._ladder_bwd.rejected(bi)
        return ws["scalars"][0:1], ws["values"]

    def reward_vse_step(self, features, captions, beta=0.2):
        """One minibatch of the reward pre-training loop (trainers.py:286-300): the GRU chain over the
        whole captions [N, L] (activations kept; segment-parallel when long enough), the two embeddings,
        VisualSemanticEmbeddingLoss on their [N, N] similarity matrix, and the gradients of every reward
        network parameter (BPTT through the batch-1 chain included).  Returns the loss (1-element tensor)."""
        R, gR = self.R, self.gR
        n, l_cols = captions.shape
        key = ("vse", n, l_cols)
        ws = self._ws.get(key)
        dev = self.device
        er = R["rewrnn.caption_embedding.weight"].shape[1]
        pos = n * l_cols
        if ws is None:
            def f32(*shape):
                return torch.empty(*shape, dtype=torch.float32, device=dev)
            ws = dict(tok=torch.zeros(l_cols, n, dtype=torch.int64, device=dev), X=f32(pos, er), XP=f32(pos, 3 * H),
                      hr=f32(1, n, H), saved=f32(pos, 5, H), se=f32(n, 512), ve=f32(n, 512), S=f32(n, n), dS=f32(n, n),
                      rows=f32(n), loss=torch.zeros(1, dtype=torch.float32, device=dev), dse=f32(n, 512),
                      dve=f32(n, 512), d_emit=f32(1, n, H), dgx=f32(pos, 3 * H), dgh=f32(pos, 3 * H), hprev=f32(pos, H),
                      dX=f32(pos, er), scratch=f32(max(16 * 3 * H * H, 16 * 512 * max(512, self.F))),
                      sync=torch.zeros(hip.chain_sync_bytes(), dtype=torch.uint8, device=dev),
                      status=torch.zeros(4, dtype=torch.int32, device=dev), cs=f32(2048 * 64 + 4 * H),
                      emb_work=torch.zeros(hip.embedding_grad_work_ints(R["rewrnn.caption_embedding.weight"].shape[0], pos),
                                           dtype=torch.int32, device=dev),
                      seam=torch.zeros(6, dtype=torch.float32, device=dev), segments=self._segments_for(pos))
            if ws["segments"]:
                ws["work"] = f32(hip.chain_seg_workspace_floats(ws["segments"]))
            self._ws[key] = ws
        ws["tok"].copy_(captions.t())
        tok = ws["tok"].view(-1)
        w_ih, w_hh = R["rewrnn.gru.weight_ih_l0"], R["rewrnn.gru.weight_hh_l0"]
        # ---- forward
        hip.gather_rows(R["rewrnn.caption_embedding.weight"], tok, 1, pos, ws["X"])
        hip.gemm(ws["X"], w_ih, ws["XP"], bias=R["rewrnn.gru.bias_ih_l0"])
        # the chain starts from zeros, so a long one runs segment-parallel, forward and backward, under the
        # same seam checks and warm-up ladders as the A2C step (a rejected attempt is redone)
        plans = self._plans(ws)
        last = len(plans) - 1
        fr, rb = min(self._ladder_fr.first, last), min(self._ladder_rb.first, last)
        while True:
            pf, pb = plans[fr], plans[rb]
            self._reward_vse_attempt(ws, features, n, l_cols, beta, tok, pf, pb)
            seg_f, seg_b = pf[0] == "segmented", pb[0] == "segmented"
            if not (seg_f or seg_b):
                break
            seam = ws["seam"].tolist()
            self.last_seam = seam
            checks_f = [(seam[2], self.SEAM_TOL_FWD)] if seg_f else []
            checks_b = [(seam[4] / seam[5] if seam[5] > 0 else 0.0, self.SEAM_TOL_BWD)] if seg_b else []
            ok_f, ok_b = all(e <= t for e, t in checks_f), all(e <= t for e, t in checks_b)
            if ok_f and ok_b:
                self._ladder_fr.accepted(fr, checks_f, last)
                self._ladder_rb.accepted(rb, checks_b, last)
                break
            self.fallbacks += 1
            if not ok_f:
                fr = self._next_plan(plans, fr, checks_f)
                self._ladder_fr.rejected(fr)
            if not ok_b:
                rb = self._next_plan(plans, rb, checks_b)
                self._ladder_rb.rejected(rb)
        self._vse_status = ws["status"]
        return ws["loss"]

    def _reward_vse_attempt(self, ws, features, n, l_cols, beta, tok, plan_f, plan_b):
        """Everything of reward_vse_step after the input projection, with the chain run as the plans say."""
        R, gR = self.R, self.gR
        pos = n * l_cols
        w_ih, w_hh = R["rewrnn.gru.weight_ih_l0"], R["rewrnn.gru.weight_hh_l0"]
        if plan_f[0] == "segmented":
            hip.gru_chain_fwd_seg(ws["XP"], w_hh, R["rewrnn.gru.bias_hh_l0"], n, l_cols, 1, plan_f[1], plan_f[2],
                                  ws["hr"], ws["work"], ws["seam"][2:4], saved=ws["saved"],
                                  persistent=self.persistent and hip.chain_pair_supported(plan_f[1]))
        else:
            hip.gru_chain_fwd(ws["XP"], w_hh, R["rewrnn.gru.bias_hh_l0"], n, l_cols, 1, ws["hr"], ws["sync"],
                              ws["status"][0:1], saved=ws["saved"])
        hr = ws["hr"].view(n, H)
        hip.gemm(hr, R["semantic_embed.weight"], ws["se"], bias=R["semantic_embed.bias"])
        hip.gemm(features, R["visual_embed.weight"], ws["ve"], bias=R["visual_embed.bias"])
        hip.gemm(ws["ve"], ws["se"], ws["S"])                                  # S = ve . se^T
        hip.vse_loss(ws["S"], beta, ws["rows"], ws["loss"], ws["dS"])
        # ---- backward: the two linear layers
        sc = ws["scratch"]
        hip.gemm(ws["dS"], ws["se"], ws["dve"], b_kmajor=True)                 # dve = dS . se
        hip.gemm(ws["dS"], ws["ve"], ws["dse"], a_kmajor=True, b_kmajor=True)  # dse = dS^T . ve
        hip.gemm_auto(ws["dve"], features, gR["visual_embed.weight"], sc, a_kmajor=True, b_kmajor=True)
        hip.colsum(ws["dve"], gR["visual_embed.bias"], scratch=ws["cs"])
        hip.gemm_auto(ws["dse"], hr, gR["semantic_embed.weight"], sc, a_kmajor=True, b_kmajor=True)
        hip.colsum(ws["dse"], gR["semantic_embed.bias"], scratch=ws["cs"])
        hip.gemm(ws["dse"], R["semantic_embed.weight"], ws["d_emit"].view(n, H), b_kmajor=True)
        # ---- backward: the chain, then its weight gradients as products over all positions
        if plan_b[0] == "segmented":
            hip.gru_chain_bwd_seg(w_hh, ws["saved"], ws["d_emit"], n, l_cols, 1, plan_b[1], plan_b[2], ws["dgx"],
                                  ws["dgh"], ws["work"], ws["seam"][4:6])
        else:
            hip.gru_chain_bwd(w_hh, ws["saved"], ws["d_emit"], n, l_cols, 1, ws["dgx"], ws["dgh"], ws["sync"],
                              ws["status"][1:2])
        hip.chain_hprev(ws["saved"], pos, ws["hprev"], gru=True)
        hip.gemm_auto(ws["dgh"], ws["hprev"], gR["rewrnn.gru.weight_hh_l0"], sc, a_kmajor=True, b_kmajor=True)
        hip.colsum(ws["dgh"], gR["rewrnn.gru.bias_hh_l0"], scratch=ws["cs"])
        hip.gemm_auto(ws["dgx"], ws["X"], gR["rewrnn.gru.weight_ih_l0"], sc, a_kmajor=True, b_kmajor=True)
        hip.colsum(ws["dgx"], gR["rewrnn.gru.bias_ih_l0"], scratch=ws["cs"])
        g_emb = gR.get("rewrnn.caption_embedding.weight")
        if g_emb is not None:
            hip.gemm_auto(ws["dgx"], w_ih, ws["dX"], sc, b_kmajor=True)
            hip.token_order(tok, 1, pos, g_emb.shape[0], ws["emb_work"])
            hip.embedding_grad(g_emb, pos, ws["dX"], ws["emb_work"])

    # ------------------------------------------------------------------ the step
    def step(self, features, prefix, n_steps, uniforms=None, forced_tokens=None, mode=hip.SAMPLE, backward=True):
        """features [N,F] float32 cuda; prefix [N,p0] int64 cuda (ground-truth start
        columns: captions[:, :1] for plain A2C, captions[:, :caplen-level] for the
        curriculum); uniforms: float64 cuda [n_steps*N], one draw per (step,row) in that
        order, required for mode SAMPLE; forced_tokens [N,p0+n_steps] for mode FORCED."""
        n, p0 = prefix.shape
        s = int(n_steps)
        assert s >= 1 and p0 >= 1
        assert features.shape == (n, self.F) and features.dtype == torch.float32 and features.is_contiguous()
        ws = self._buffers(n, p0, s)
        self._forget_scales(ws)
        P, V, R = self.P, self.V, self.R
        tokT = ws["tokT"]
        if mode == hip.FORCED:
            tokT.copy_(forced_tokens.t())
        else:
            tokT[:p0].copy_(prefix.t())
        if mode == hip.SAMPLE:
            assert uniforms is not None and uniforms.dtype == torch.float64 and uniforms.numel() >= s * n
        main = torch.cuda.current_stream()

        self._mark("start")
        side = self._side_stream() if self.concurrent else main
        # the two products of the image features that the heads need after the chains (visual embedding,
        # feature half of the value head's first layer) depend on nothing the step computes: they run on
        # the side stream beside the rollout, whose small launches leave most of the device idle
        ev_start = torch.cuda.Event()
        ev_start.record(main)
        # first on the critical path, queued before the other streams' launches (the host needs 0.2 ms for those):
        # h0 and, when the rollout starts at the first column, the first cell update
        self._rollout_h0(ws, features)
        ws["h0_done"] = True             # (for the _rollout call below)
        if self.E % 64 == 0 and self.pd == "f32":
            self._rollout_cell(ws, 0, n, True)
            ws["cell0_done"] = True
        # then what the rollout's first round waits for (the policy network's per-word table and the pieces of its
        # stacked weight), then the rollout itself; what is needed only after it - the image-feature products, the
        # chains' per-word tables - is queued last and runs beside it all the same
        tables_on = self._check_stream() if self.concurrent else main
        ev_table_p = None
        with torch.cuda.stream(tables_on):
            tables_on.wait_event(ev_start)
            if self._policy_table_wanted(n, p0, s, mode):
                self._policy_table(ws, stacked_on=side if self.concurrent else None,
                                   after=ev_start)
                ev_table_p = torch.cuda.Event()
                ev_table_p.record(tables_on)
        events = {}

        def beside_the_rollout():
            with torch.cuda.stream(side):
                side.wait_event(ev_start)
                hip.gemm(features, R["visual_embed.weight"], ws["ve"], bias=R["visual_embed.bias"])
                hip.gemm(features, V["linear1.weight"][:, :self.F], ws["U"], bias=V["linear1.bias"])
                if backward:
                    self._value_head_vector(ws)
                events["feat"] = torch.cuda.Event()
                events["feat"].record(side)
            with torch.cuda.stream(tables_on):
                self._word_tables(ws)
                if backward:
                    self._parameter_scales(ws)
                events["tables"] = torch.cuda.Event()
                events["tables"].record(tables_on)

        self._rollout(ws, features, n, p0, s, uniforms, mode, ev_table=ev_table_p, between=beside_the_rollout)
        self._mark("rollout_done")
        ev_feat, ev_tables = events["feat"], events["tables"]
        main.wait_event(ev_feat)
        main.wait_event(ev_tables)
        ws["policy_early"] = False
        if backward and mode != hip.GREEDY:
            # (queued behind the forward chains' launch, beside which it runs: _chains_forward_loss)
            ws["order_pending"] = (n, (p0 - 1 + s) * n)
            ws["early_pending"] = bool(self.concurrent)
        last = len(self.warmups)   # ladder index of the exact kernels
        fv, fr, bi = (min(lad.first, last) for lad in (self._ladder_fv, self._ladder_fr, self._ladder_bwd))
        with_bwd = backward and mode != hip.GREEDY
        status_host = None
        gave_up = 0   # attempts whose persistent launch gave up waiting for a workgroup (reported as an infinite error)
        # ---- chains forward + loss + backward, enqueued in one go; the forward seam errors come back
        # through pinned memory as soon as the chains are done, so the host knows whether the attempt
        # stands while the device is still busy with the backward pass.  A rejected forward seam or replay
        # check redoes everything after the rollout; a rejected backward seam only the value chain's BPTT,
        # unless its longer warm-up also asks for more calls in full.
        ev_replay = None
        while True:
            if ev_replay is not None:
                # a rejected attempt's replay check may still be reading the chains' buffers on its own stream:
                # nothing of the next attempt (which rewrites and may reallocate them) is queued in front of it
                main.wait_event(ev_replay)
            fc, pv, pr, pb = self._choose(ws, n, p0, s, fv, fr, bi)
            if pv[0] == "exact":
                bi = last
            tail = with_bwd and self._tail_ok(n, pv, pr, pb)
            # (G0 . W is queued behind the forward chains' launch on the third stream whichever way the policy LSTM's
            # BPTT is run: it starts when the chains end, beside the heads and the loss, and what flows into the LSTM
            # outputs is ready when the value chain's BPTT is)
            # (with the policy BPTT as one launch the third stream carries the vocabulary products on the path to it:
            # the token order, which only the embedding gradients at the end of the step read, is queued behind them)
            ws["order_late"] = bool(tail and self.concurrent)
            ws["defer_replay"] = bool(tail and self.policy_bptt_first)
            ws.pop("replay_deferred", None)
            ws.pop("ev_replay_late", None)
            ev_seam, ev_replay, replay_steps = self._chains_forward_loss(ws, features, n, p0, s, fc, backward, main, side,
                                                                         pv, pr)
            if with_bwd:
                self._backward(ws, features, n, p0, s, fc, main, side, pb, tail)
            if "replay_deferred" in ws:       # nobody has queued the replay check yet
                ev_replay = ws.pop("replay_deferred")()
            elif "ev_replay_late" in ws:
                ev_replay = ws.pop("ev_replay_late")
            self._mark("backward_done")
            seg_v, seg_r = pv[0] == "segmented", pr[0] == "segmented"
            checks_v, checks_r = [], []
            if seg_v or seg_r:
                ev_seam.synchronize()
                seam = ws["seam_host"].tolist()
                checks_v = [(seam[0], self.SEAM_TOL_FWD)] if seg_v else []
                checks_r = [(seam[2], self.SEAM_TOL_FWD)] if seg_r else []
                ok_v, ok_r = (all(e <= t for e, t in c) for c in (checks_v, checks_r))
                if not (ok_v and ok_r):
                    self.fallbacks += 1
                    if any(e == float("inf") for e, _ in checks_v + checks_r):
                        # +inf is what a launch reports whose workgroups were not all resident in time (other
                        # kernels held the CUs; the library prints who waited for whom): that says nothing about the
                        # warm-up, so the same plans run again - and after MAX_GIVE_UPS such attempts with one
                        # launch per step, which needs no workgroup to wait for another, instead of climbing the
                        # warm-up ladder to the exact kernels on a signal that is not about warm-ups
                        gave_up += 1
                        self.launch_timeouts += 1
                        if gave_up == self.MAX_GIVE_UPS:
                            self._degrade_to_per_step_launches()
                        if gave_up <= self.MAX_GIVE_UPS:
                            continue
                    if (self._warm16_active(pv, pr) and not (self._ladder_fv.probing or self._ladder_fr.probing)
                            and self._sudden(self._ladder_fv, fv, ok_v) and self._sudden(self._ladder_fr, fr, ok_r)):
                        # the 16-bit part of the warm-up may be what failed (not so when a shorter warm-up was being
                        # tried: then that is what failed): the same plans again without it
                        self._warm16_off, self._warm16_penalty = self._warm16_penalty, min(2 * self._warm16_penalty, 1024)
                        continue
                    if not ok_v:
                        fv = self._next_rung(fv, checks_v)
                        self._ladder_fv.rejected(fv)
                    if not ok_r:
                        fr = self._next_rung(fr, checks_r)
                        self._ladder_fr.rejected(fr)
                    continue
            if ev_replay is not None:
                # the replays that were left out: did they come back to the state they started from?
                ev_replay.synchronize()
                rep = ws["replay_host"].tolist()
                self.last_replay = (fc, replay_steps, rep[0], rep[1])
                if not all(e <= self.REPLAY_TOL for e in rep):
                    self.fallbacks += 1
                    self._replay_rejected(n, p0, fc)
                    continue
            else:
                self.last_replay = (fc, 0, 0.0, 0.0)
            if not self.per_sample and ws["segments"]:
                self._ladder_fv.accepted(fv, checks_v if seg_v else [], last)
                self._ladder_fr.accepted(fr, checks_r if seg_r else [], last)
                if with_bwd and pb[0] == "exact":
                    self._ladder_bwd.accepted(bi, [], last)
            self.last_seam = ws["seam_host"].tolist()[:4] + [0.0, 0.0]
            # ---- the backward seam
            restart = False
            while with_bwd and pb[0] == "segmented":
                # one read-back for the backward seam and the status words (the step's last host sync): it was
                # queued right behind the chain's BPTT, so the host is released while the weight gradients that
                # follow are still running and has the optimiser step and the next rollout queued by their end
                ws["ev_ctl"].synchronize()
                seam_b = ws["ctl_host"][0:2].tolist()
                status_host = ws["ctl_host"][4:8].view(torch.int32).tolist()
                self.last_seam[4:6] = seam_b
                checks_b = [(seam_b[0] / seam_b[1] if seam_b[1] > 0 else 0.0, self.SEAM_TOL_BWD)]
                if all(e <= t for e, t in checks_b):
                    self._ladder_bwd.accepted(bi, checks_b, last)
                    break
                self.fallbacks += 1
                if seam_b[0] == float("inf") and gave_up < self.MAX_GIVE_UPS:
                    # the BPTT launch gave up waiting for its workgroups (see above): the same plan once more
                    gave_up += 1
                    self.launch_timeouts += 1
                    if gave_up == self.MAX_GIVE_UPS:
                        self._degrade_to_per_step_launches()
                        pb = self._choose(ws, n, p0, s, fv, fr, bi)[3]
                    self._value_chain_backward(ws, n, p0, s, fc, main, side, pb, None, tail=False)
                    self._mark("backward_done")
                    continue
                if (self.last_cheap_bwd > 0 and not self._ladder_bwd.probing
                        and self._sudden(self._ladder_bwd, bi, False)):
                    # the 16-bit part of the warm-up may be what failed: the same plan again without it
                    self._warm16b_off, self._warm16b_penalty = self._warm16b_penalty, min(2 * self._warm16b_penalty, 1024)
                else:
                    bi = self._next_rung(bi, checks_b)
                    self._ladder_bwd.rejected(bi)
                if self._choose(ws, n, p0, s, fv, fr, bi)[0] != fc:
                    restart = True     # the longer warm-up reaches further than the replays left out are long
                    break
                pb = self._choose(ws, n, p0, s, fv, fr, bi)[3]
                self._value_chain_backward(ws, n, p0, s, fc, main, side, pb, None, tail=tail)
                self._mark("backward_done")
            if restart:
                continue
            if with_bwd and tail and status_host is not None and status_host[3] != 0 and gave_up <= self.MAX_GIVE_UPS:
                # the policy LSTM's one-launch BPTT gave up waiting for its workgroups (it needs every CU; another
                # process on the GPU held some): once more with a launch per step, which waits for nobody, and its
                # weight gradients again (everything else of the minibatch stands)
                gave_up += 1
                self.launch_timeouts += 1
                self._redo_policy_bptt(ws, features, n, p0, s, main)
                status_host[3] = 0
                self._policy_give_ups += 1
                if self._policy_give_ups == self.MAX_GIVE_UPS and self.late_policy_bptt:
                    # three minibatches in a row: something keeps the device from holding the launch (each give-up
                    # costs its 20 ms wait): the policy LSTM's BPTT goes back to a launch per step from here on
                    import warnings
                    warnings.warn("the policy LSTM's one-launch BPTT gave up waiting for its workgroups in %d minibatches in "
                                  "a row: this engine goes on with one launch per step for it" % self.MAX_GIVE_UPS,
                                  RuntimeWarning)
                    self.late_policy_bptt = False
            elif with_bwd and tail:
                self._policy_give_ups = 0
            if with_bwd and self.on_policy_grads is not None and self._ev_policy is not None:
                # the forward pass stands and no launch of the backward pass is repeated any more, so the policy
                # gradients being computed now are final
                self.on_policy_grads(self._ev_policy)
                self._ev_policy = None
            break
        self._replay_accepted()
        self._warm16_off = max(0, self._warm16_off - 1)
        self._warm16b_off = max(0, self._warm16b_off - 1)
        # the tensors handed back are final now; their copies queue behind the backward pass
        out = (tokT.t().contiguous(), ws["logp"].t().contiguous(), ws["values"].view(s, n).t().contiguous(),
               ws["rewards"].view(s, n).t().contiguous())
        return A2CStepResult(*out, ws["scalars"], ws["entropy"] if self.want_entropy else None, ws["status"],
                             (pv, pb), pr, status_host, full_calls=fc)

    # the backward pass's two vocabulary-sized products on 256 x 256 tiles (alone 433 against 481 us for G0 . W and 369
    # against 472 for G0^T . Hs; inside the step 5.09-5.11 against 5.19-5.25 ms with both, 5.12-5.14 with the second
    # alone; the weight-gradient shapes, whose outputs are 16 such tiles, gain nothing, the input-gradient shapes
    # 13 % alone and nothing inside the step)
    BIG = "ug"   # ("u": G0 . W, "g": G0^T . Hs)
    LONG_REPLAY = 160   # steps of a replay check from which the side stream is left to it alone (see _value_chain_backward)
    MAX_GIVE_UPS = 3   # attempts of one minibatch whose persistent launch may give up before the engine stops using them

    def _degrade_to_per_step_launches(self):
        """Persistent launches keep giving up (something else holds the CUs they need - another process on the same
        GPU, a stream that keeps the device full): from here on this engine walks the chains with one launch per
        step.  Slower by the launch overheads, but no workgroup waits for another."""
        if self.persistent:
            import warnings
            warnings.warn("persistent chain launches gave up waiting for their workgroups %d times in one minibatch "
                          "(see the [icrl] lines on stdout): this engine goes on with one launch per chain step"
                          % self.MAX_GIVE_UPS, RuntimeWarning)
            self.persistent = False

    def _word_tables(self, ws):
        """The chains' input projections for every WORD: table[w] = W_ih emb[w] + b (the value LSTM's with both
        biases, the reward GRU's with b_ih).  They depend on the parameters only, so they are formed beside the
        rollout; once the tokens exist a position's projection is a row gather - row by row the same sums as
        the product over the gathered embeddings."""
        V, R = self.V, self.R
        if "table_v" not in ws:   # allocated by the first A2C step of the shape (the pre-training steps need none)
            ws["table_r"] = torch.empty(R["rewrnn.caption_embedding.weight"].shape[0], 3 * H, dtype=torch.float32,
                                        device=self.device)
            ws["table_v"] = torch.empty(V["valrnn.caption_embedding.weight"].shape[0], 4 * H, dtype=torch.float32,
                                        device=self.device)
        f16 = self.rollout_arith == "f16x2" and V["valrnn.caption_embedding.weight"].shape[1] % 16 == 0
        if f16:
            self._word_table_f16x2(ws, "v", V["valrnn.caption_embedding.weight"], V["valrnn.lstm.weight_ih_l0"],
                                   V["valrnn.lstm.bias_ih_l0"], V["valrnn.lstm.bias_hh_l0"], ws["table_v"])
        else:
            hip.gemm(V["valrnn.caption_embedding.weight"], V["valrnn.lstm.weight_ih_l0"], ws["table_v"],
                     bias=V["valrnn.lstm.bias_ih_l0"], bias2=V["valrnn.lstm.bias_hh_l0"])
        # a reward network declared frozen (it is, during A2C training: trainers.py:372-373) keeps its table
        # until parameters_changed() is called
        if not (self.frozen_reward and ws.get("table_r_epoch") == self._param_epoch):
            if f16 and R["rewrnn.caption_embedding.weight"].shape[1] % 16 == 0:
                self._word_table_f16x2(ws, "r", R["rewrnn.caption_embedding.weight"], R["rewrnn.gru.weight_ih_l0"],
                                       R["rewrnn.gru.bias_ih_l0"], None, ws["table_r"])
            else:
                hip.gemm(R["rewrnn.caption_embedding.weight"], R["rewrnn.gru.weight_ih_l0"], ws["table_r"],
                         bias=R["rewrnn.gru.bias_ih_l0"])
            ws["table_r_epoch"] = self._param_epoch

    def _early_now(self, ws, n, p0, s, main):
        """Start the loss-independent part of the policy backward pass, if this step wants it and has not yet."""
        if ws.pop("early_pending", False):
            self._policy_backward_early(ws, n, p0, s, main, self._check_stream())

    def _policy_backward_early(self, ws, n, p0, s, main, stream):
        """The part of the policy network's backward pass that needs no loss: d logits up to the per-row
        coefficient dLoss/dlog_prob, G0 = onehot(token) - softmax, and their product with the vocabulary matrix
        U = G0 . W (what flows into the LSTM outputs is coef * U, row by row).  On `stream`, beside the chains."""
        P = self.P
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(stream):
            stream.wait_event(ev)
            G0 = ws["dlogits"].view(s * n, self.vocab)   # left by the rollout's sampler launches
            self._uvocab(ws, G0, ws["scratch_check"])
            done = torch.cuda.Event()
            done.record(stream)
            self._mark("check:policy_early_done")
        ws["policy_early"], ws["ev_early"] = True, done

    def _uvocab(self, ws, G0, scratch):
        """ws["Uvocab"] = G0 . linear2vocab.weight on the current stream (what flows into the LSTM outputs, up to the
        per-row coefficient)."""
        rows = self._g0_row_scales(ws, G0) if self.vgd == "f16x2" else None
        self._pgemm(ws, "uvoc", G0, self.P["linear2vocab.weight"], ws["Uvocab"], scratch, self.vgd, b_kmajor=True, shared=True,
                    a_scale=rows, b_scale=ws.get("psc_wvoc") if self.vgd == "f16x2" else None, big_tile="u" in self.BIG)

    def _vocab_weight_grad(self, ws, G0, scratch):
        """linear2vocab.weight's gradient G0^T . (coef * Hs) on the current stream."""
        colsc = None
        if self.vgd == "f16x2":
            colsc, ev = self._g0_col_scales(ws, G0)
            torch.cuda.current_stream().wait_event(ev)   # (they may have been computed on another stream, beside the chains)
        self._pgemm(ws, "gvoc", G0, ws["HsC"], self.gP["linear2vocab.weight"], scratch, self.vgd, a_kmajor=True, b_kmajor=True,
                    shared=True, a_scale=colsc, big_tile="g" in self.BIG)

    def _order_tokens(self, ws, n, rows, main, side, after=None):
        """The rows of the token matrix listed by token (what the reproducible embedding gradients sum by): it
        needs the tokens only, and both trainable embedding tables are looked up with the same ones, so it is
        built once per minibatch, on the side stream, where the gradients that use it are computed later."""
        if not any(g is not None and g.get(k) is not None for g, k in
                   ((self.gP, "caption_embedding.weight"), (self.gV, "valrnn.caption_embedding.weight"))):
            return
        if side is main:
            hip.token_order(ws["tokT"].view(-1), 1, rows, self._emb_vocab(), ws["emb_work"])
            ws["ev_order_done"] = None
            return
        # on the third stream (idle until the replay check), so that it delays neither the chains' input
        # products on the side stream nor the chains; the side stream picks the result up from there
        check = self._check_stream()
        ev = after    # an event after which the tokens are final (default: everything queued on main so far)
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(main)
        with torch.cuda.stream(check):
            check.wait_event(ev)
            hip.token_order(ws["tokT"].view(-1), 1, rows, self._emb_vocab(), ws["emb_work"])
            done = torch.cuda.Event()
            done.record(check)
        ws["ev_order"] = ws["ev_order_done"] = done   # side stream (policy table) and main stream (value table) wait on it

    def _emb_vocab(self):
        sizes = {g[k].shape[0] for g, k in ((self.P, "caption_embedding.weight"), (self.V, "valrnn.caption_embedding.weight"))
                 if g is not None}
        assert len(sizes) == 1, "the policy and value networks are built over one vocabulary"
        return sizes.pop()

    def _replay_rejected(self, n, p0, fc):
        """A replay that was left out did not come back to its start state within the steps checked: ask
        for replays twice as long (so more calls are walked in full), and do not come back down for a while."""
        self._replay_need = 2 * max(self.MIN_REPLAY, (p0 + fc - 1) * n)
        self._replay_cool, self._replay_penalty = self._replay_penalty, min(2 * self._replay_penalty, 1024)

    def _replay_accepted(self):
        """After a while without a rejection, try shorter replays again (half the length at a time)."""
        if self._replay_cool > 0:
            self._replay_cool -= 1
        elif self._replay_need > 0:
            self._replay_need = self._replay_need // 2 if self._replay_need // 2 > self.MIN_REPLAY else 0
            self._replay_cool = 8

    def _chains_forward_loss(self, ws, features, n, p0, s, fc, backward, main, side, plan, plan_r):
        """The two forward chains, the heads and the loss.  plan / plan_r: how the value / reward chain
        runs; fc: calls walked in full (fc < s: the later calls' replays are left out, and the check that
        this is right is started on a stream of its own).  Returns (an event after which ws["seam_host"][:4]
        holds the forward seam errors, an event after which ws["replay_host"] holds the replay check's two
        errors or None, the steps that check walks)."""
        P, V, R = self.P, self.V, self.R
        l_cols, lv = p0 + s, p0 + s - 1
        tv, tr = hip.chain_positions(n, p0, s, fc), hip.chain_positions(n, p0 + 1, s, fc)
        collapsed = fc < s
        self._chain_capacity(ws, tv, tr if collapsed else 0)
        tok_flat = ws["tokT"].view(-1)
        ws["Xv_gathered"] = False     # set below when this attempt gathers the value network's embedding rows early
        ws["psc_hbuf"] = None
        seg, seg_r, rows = plan[0] == "segmented", plan_r[0] == "segmented", plan[0] == "rows"
        keep = backward or collapsed      # the value chain's activations: for the BPTT, and for the replay check

        # ---- 2. the two forward chains: one persistent launch for both when the device can hold it,
        # else the reward GRU chain on the side stream beside the value LSTM chain on the main stream
        # (per-row chains: always one call for both, persistent or a launch per step)
        pair = rows or (seg and seg_r and self.persistent and hip.chain_pair_supported(plan[1]))
        assert pair or not collapsed or (seg and seg_r), "replays are only left out on the segmented kernels"
        if rows or not pair:
            self._early_now(ws, n, p0, s, main)
        # The device starts launches in about the order the host queues them, whatever their streams: the two
        # gathers the chains wait for go first, and what the chains do not need (the token order for the embedding
        # gradients, the value network's embedding rows) is queued behind the chains' launch and runs beside it
        # (queued in front, those five small launches ran one after the other, 0.2 ms between the last sampler and
        # the chains).
        ev_tok = torch.cuda.Event()
        ev_tok.record(main)
        hip.gather_rows(ws["table_v"], tok_flat, 1, lv * n, ws["XPv"])
        with torch.cuda.stream(side):
            side.wait_event(ev_tok)
            hip.gather_rows(ws["table_r"], tok_flat, 1, l_cols * n, ws["XPr"])
            if pair:
                ev_xpr = torch.cuda.Event()
                ev_xpr.record(side)
            elif seg_r:
                hip.gru_chain_fwd_seg(ws["XPr"], R["rewrnn.gru.weight_hh_l0"], R["rewrnn.gru.bias_hh_l0"], n, p0 + 1, s,
                                      plan_r[1], plan_r[2], ws["hr"], ws["work_r"], ws["seam"][2:4], full_calls=fc,
                                      saved=ws["saved_r"] if collapsed else None)
            else:
                hip.gru_chain_fwd(ws["XPr"], R["rewrnn.gru.weight_hh_l0"], R["rewrnn.gru.bias_hh_l0"], n, p0 + 1, s,
                                  ws["hr"], ws["sync_r"], ws["status"][1:2])
            if not pair:
                ev_gru = torch.cuda.Event()
                ev_gru.record(side)
                self._reward_head(ws, features, n, s, side, visual_done=True)
                ev_rew = torch.cuda.Event()
                ev_rew.record(side)
        if pair:
            main.wait_event(ev_xpr)
        if rows:
            hip.chain_pair_fwd_rows(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], ws["XPr"], R["rewrnn.gru.weight_hh_l0"],
                                    R["rewrnn.gru.bias_hh_l0"], n, p0, p0 + 1, s, ws["hv"],
                                    ws["saved"] if backward else None, ws["hr"], ws["work_v"], ws["work_r"],
                                    ws["status"][0:1], persistent=self.persistent and hip.chain_pair_supported(n))
        elif pair and self.chain_arith == "f16x2" and hip.chain_f16x2_supported(plan[1]):
            self.last_cheap = (0, 0)
            hip.chain_pair_fwd_seg(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], ws["XPr"], R["rewrnn.gru.weight_hh_l0"],
                                   R["rewrnn.gru.bias_hh_l0"], n, p0, p0 + 1, s, plan[1], plan[2], ws["hv"],
                                   ws["saved"] if keep else None, ws["hr"], ws["work_v"], ws["work_r"], ws["seam"][0:2],
                                   ws["seam"][2:4], warmup_r=plan_r[2], full_calls=fc,
                                   saved_r=ws["saved_r"] if collapsed else None, arith="f16x2")
            # G0 . W behind the launch, not beside it: the f16 pair holds every CU's registers, so what is queued beside
            # it either waits for it to end anyway or - when it gets to the CUs first, as a 1 520-workgroup product did
            # in a kernel trace - keeps the pair's workgroups from becoming resident (pair 0.82 -> 1.82 ms)
            self._early_now(ws, n, p0, s, main)
        elif pair:
            cheap = self._note_cheap(plan, plan_r)
            pair_args = (ws["XPv"], V["valrnn.lstm.weight_hh_l0"], ws["XPr"], R["rewrnn.gru.weight_hh_l0"],
                         R["rewrnn.gru.bias_hh_l0"], n, p0, p0 + 1, s, plan[1], plan[2], ws["hv"],
                         ws["saved"] if keep else None, ws["hr"], ws["work_v"], ws["work_r"], ws["seam"][0:2],
                         ws["seam"][2:4])
            pair_opts = dict(warmup_r=plan_r[2], full_calls=fc, saved_r=ws["saved_r"] if collapsed else None, cheap=cheap)
            if any(cheap):
                # the 16-bit launch needs every CU's registers: it goes first, and what runs beside the chains
                # on the other streams is released behind it
                hip.chain_pair_fwd_seg(*pair_args, phase=1, **pair_opts)
                self._early_now(ws, n, p0, s, main)
                hip.chain_pair_fwd_seg(*pair_args, phase=2, **pair_opts)
            else:
                self._early_now(ws, n, p0, s, main)
                hip.chain_pair_fwd_seg(*pair_args, **pair_opts)
        ws["ev_tok"] = ev_tok
        if "order_pending" in ws and not ws.get("order_late"):
            self._order_tokens(ws, *ws.pop("order_pending"), main, side, after=ev_tok)
        if pair and backward:
            # the value network's embedding rows, which only its input-side weight gradient reads: gathered
            # beside the chains instead of at the end of the step (the side stream's later events cover it)
            with torch.cuda.stream(side):
                hip.gather_rows(V["valrnn.caption_embedding.weight"], tok_flat, 1, lv * n, ws["Xv"])
                if self.vgd == "f16x2":
                    # and the column scales of the policy LSTM's outputs for its recurrent weight gradient
                    ws["psc_hbuf"] = self._scale_of(ws, "p_hbuf", ws["Hbuf"][:p0 - 1 + s].view((p0 - 1 + s) * n, H), True)
            ws["Xv_gathered"] = True
        ev_seam = None
        if pair:
            ev_seam = self._seam_to_host(ws, main)
            ev_pair = torch.cuda.Event()
            ev_pair.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_pair)
                self._reward_head(ws, features, n, s, side, visual_done=True)
                ev_rew = torch.cuda.Event()
                ev_rew.record(side)
                if backward and self.vgd == "f16x2" and ws.get("defer_replay"):
                    # the side stream is idle from here to the end of the policy LSTM's BPTT: the pass over G0 for its
                    # column scales (the vocabulary weight gradient's), beside G0 . W on the third stream
                    self._g0_col_scales(ws, ws["dlogits"].view(s * n, self.vocab))
        elif seg:
            hip.lstm_chain_fwd_seg(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], n, p0, s, plan[1], plan[2], ws["hv"],
                                   ws["saved"] if keep else None, ws["work_v"], ws["seam"][0:2], full_calls=fc)
        else:
            hip.lstm_chain_fwd(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], n, p0, s, ws["hv"],
                               ws["saved"] if backward else None, ws["sync_v"], ws["status"][0:1])
        ev_replay, replay_steps = None, 0
        ws["replay_steps"] = 0
        if collapsed:
            # beside everything that follows, on a stream of its own: the first steps of every replay that was
            # left out, from the state the chain had there
            replay_steps = min((p0 + fc - 1) * n, self._replay_steps(max(plan[2], plan_r[2])))
            ws["replay_steps"] = replay_steps
            ev_chains = torch.cuda.Event()
            ev_chains.record(main)
            replay_on = self._replay_stream() if self.concurrent else main

            def launch_replay(after=None):
                check = replay_on
                with torch.cuda.stream(check):
                    check.wait_event(ev_chains)
                    if after is not None:
                        check.wait_event(after)
                    if not pair:
                        check.wait_event(ev_gru)
                    hip.chain_pair_replay_check(ws["XPv"], V["valrnn.lstm.weight_hh_l0"], ws["saved"], ws["XPr"],
                                                R["rewrnn.gru.weight_hh_l0"], R["rewrnn.gru.bias_hh_l0"], ws["saved_r"],
                                                n, p0, p0 + 1, s, fc, replay_steps, ws["work_cv"], ws["work_cr"],
                                                ws["ctl"][6:8],
                                                arith="f16x2" if (self.chain_arith == "f16x2" and self.persistent
                                                                  and hip.chain_f16x2_supported(s - fc)) else "f32")
                    ws["replay_host"].copy_(ws["ctl"][6:8], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(check)
                    self._mark("check:replay_done")
                return ev

            if ws.get("defer_replay"):
                ws["replay_deferred"] = launch_replay   # (the backward pass queues it behind the policy BPTT's launch)
            else:
                ev_replay = launch_replay()
        w1 = V["linear1.weight"]
        hip.gemm(ws["hv"].view(s * n, H), w1[:, self.F:], ws["Y"])
        hip.value_head_fwd(ws["Y"], ws["U"], V["linear2.weight"], V["linear2.bias"], s * n, n, ws["values"])
        self._mark("value_fwd_done")
        main.wait_event(ev_rew)
        if ev_seam is None:
            ev_seam = self._seam_to_host(ws, main)

        # ---- 3. loss (trainers.py:471-476)
        hip.a2c_loss(ws["values"], ws["rewards"], ws["logp"].view(-1), ws["scalars"],
                     ws["dvalue"] if backward else None, ws["dlogp"] if backward else None)
        self._mark("loss_done")
        return ev_seam, ev_replay, replay_steps

    @staticmethod
    def _seam_to_host(ws, stream):
        ws["seam_host"].copy_(ws["seam"][:4], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def _reward_head(self, ws, features, n, s, stream, visual_done=False):
        """Semantic / visual embeddings and their cosine (models.py:235-240) on the current stream."""
        R = self.R
        hip.gemm(ws["hr"].view(s * n, H), R["semantic_embed.weight"], ws["se"], bias=R["semantic_embed.bias"])
        if not visual_done:
            hip.gemm(features, R["visual_embed.weight"], ws["ve"], bias=R["visual_embed.bias"])
        hip.cosine_reward(ws["ve"], ws["se"], s * n, n, ws["rewards"])
        self._mark("side:reward_done")

    # ------------------------------------------------------------------ gradients
    def _backward(self, ws, features, n, p0, s, fc, main, side, plan, tail=False):
        ev_loss = torch.cuda.Event()
        ev_loss.record(main)

        # ---- value network: what the chain's BPTT needs of the head's backward pass, then the BPTT (main stream);
        # the head's own weight gradients are not on that path: they are queued on the third stream behind the
        # vocabulary gradients (_policy_backward), where they run beside the end of the step
        late = self.concurrent
        self._value_head_backward(ws, features, n, s, weights=not late)
        ws["head_grads_pending"] = (features, n, s) if late else None
        self._value_chain_backward(ws, n, p0, s, fc, main, side, plan, ev_loss, features, tail=tail)
        ev_head = ws.pop("ev_head_grads", None)
        if ev_head is not None:
            main.wait_event(ev_head)
        elif ws.pop("head_grads_pending", None) is not None:   # nobody picked them up: here and now
            self._value_head_weight_grads(ws, features, n, s, ws["scratch_main"], ws["cs_main"])

    def _value_head_vector(self, ws):
        """uh = w2 . W1h, the 512-vector through which the value head's output depends on the chain's hidden state:
        there is no non-linearity between the head's two linear layers (models.py:175-178), so what flows back into
        the chain is dvalue[r] * uh, row by row.  It depends on the parameters only."""
        V = self.V
        if "uh" not in ws:
            ws["uh"] = torch.empty(1, H, dtype=torch.float32, device=self.device)
        hip.gemm(V["linear2.weight"], V["linear1.weight"][:, self.F:], ws["uh"], b_kmajor=True)

    def _value_head_backward(self, ws, features, n, s, weights=True):
        """Backward of the two linear layers of the value head (models.py:175-178) from ws["dvalue"]:
        d_emit, the gradient entering the chain at each call's last column - dvalue[r] * (w2 . W1h), an outer
        product with a vector formed off the critical path (_value_head_vector) instead of two products over all
        rows in front of the chain's BPTT - and (weights=True) the layers' weight gradients."""
        hip.scale_rows(ws["uh"].expand(s * n, H), ws["dvalue"], ws["d_emit"].view(s * n, H))
        if weights:
            self._value_head_weight_grads(ws, features, n, s, ws["scratch_main"], ws["cs_main"])

    @staticmethod
    def _forget_scales(ws):
        """Scales kept in the workspace from one product to the next (parameter columns, the policy LSTM's outputs)
        belong to the minibatch that computed them: every entry point that shares the workspace drops them first, so
        that a later pass never cuts operands with scales taken from older parameters (a row that has grown past its
        old scale would overflow f16)."""
        for k in [k for k in ws if isinstance(k, str) and k.startswith("psc_")]:
            ws[k] = None

    def _scale_of(self, ws, key, x, cols):
        """Power-of-two scales for a two-piece f16 product (hipabi.gemm, dtype "f16x2"): one per row of x, or with
        `cols` one per column, from one pass over x on the current stream.  Kept in ws under `key` (one per call site:
        products on different streams must not share them)."""
        r, c = x.shape
        count = c if cols else r
        buf = ws.get("sc16_" + key)
        if buf is None or buf[0].numel() < count or buf[1].numel() < r + c:
            buf = ws["sc16_" + key] = (torch.empty(count, dtype=torch.float32, device=self.device),
                                       torch.empty(r + c, dtype=torch.int32, device=self.device))
        hip.absmax_scales(x, buf[1], col_scale=buf[0] if cols else None, row_scale=None if cols else buf[0])
        return buf[0]

    def _scales_of(self, ws, key, x):
        """(row scales, column scales) of x from ONE pass (an operand that enters one product along its rows and
        another along its columns: the gate gradients)."""
        r, c = x.shape
        buf = ws.get("sc16_" + key)
        if buf is None or buf[0].numel() < r or buf[1].numel() < c:
            buf = ws["sc16_" + key] = (torch.empty(r, dtype=torch.float32, device=self.device),
                                       torch.empty(c, dtype=torch.float32, device=self.device),
                                       torch.empty(r + c, dtype=torch.int32, device=self.device))
        hip.absmax_scales(x, buf[2], row_scale=buf[0], col_scale=buf[1])
        return buf[0], buf[1]

    def _unit_scale(self, ws, count):
        """The scale of an operand whose entries lie in (-1, 1) (an LSTM's hidden state): 2^14 for every row, no pass."""
        t = ws["sc16_unit"]
        assert t.numel() >= count
        return t

    def _parameter_scales(self, ws):
        """Scales that depend on the parameters alone, computed beside the rollout (current stream): the columns of the
        embedding tables (a bound for the columns of any set of gathered rows), of the LSTMs' input weights and of the
        vocabulary matrix, as the backward pass's two-piece products take them."""
        if self.gd != "f16x2" and self.vgd != "f16x2":
            return
        P, V = self.P, self.V
        for key, w in (("embP", P["caption_embedding.weight"]), ("wihP", P["lstm.weight_ih_l0"]),
                       ("wvoc", P["linear2vocab.weight"]), ("embV", V["valrnn.caption_embedding.weight"]),
                       ("wihV", V["valrnn.lstm.weight_ih_l0"])):
            ws["psc_" + key] = self._scale_of(ws, "p_" + key, w, True)

    def _pgemm(self, ws, tag, a, b, out, scratch, dtype, a_kmajor=False, b_kmajor=False, shared=False, a_scale=None,
               b_scale=None, big_tile=False):
        """A product of the backward pass in the arithmetic `dtype` (hipabi.gemm_auto).  With "f16x2" the operands'
        scales are computed here, a pass over each, unless the caller hands them in (an operand of two products)."""
        if dtype == "f16x2":
            if a_scale is None:
                a_scale = self._scale_of(ws, tag + "a", a, a_kmajor)
            if b_scale is None:
                b_scale = self._scale_of(ws, tag + "b", b, b_kmajor)
        else:
            a_scale = b_scale = None
        hip.gemm_auto(a, b, out, scratch, a_kmajor=a_kmajor, b_kmajor=b_kmajor, shared=shared, dtype=dtype, a_scale=a_scale,
                      b_scale=b_scale, big_tile=big_tile)

    def _g0_row_scales(self, ws, G0):
        """Row scales of G0 = onehot(token) - softmax for the two-piece products, on the current stream, from the rows'
        log-probabilities (the largest magnitude of a row is 1 - p_token: hipabi.g0_row_scales)."""
        rows = ws.get("g0_row_scale")
        if rows is None or rows.numel() < G0.shape[0]:
            rows = ws["g0_row_scale"] = torch.empty(G0.shape[0], dtype=torch.float32, device=self.device)
        hip.g0_row_scales(ws["logp"].view(-1)[:G0.shape[0]], rows)
        return rows

    def _g0_col_scales(self, ws, G0):
        """Column scales of G0 (one pass over the matrix, 0.4 GB at the metric's size) on the current stream, once per
        rollout: returns (scales, the event after which they are complete).  Bound by memory traffic, so it is queued
        where it runs beside a product bound by the matrix instruction (_chains_forward_loss), not in the step's tail."""
        got = ws.get("g0_col_scales")
        if got is None or got[0] != ws["rollout_id"]:
            colsc = self._scale_of(ws, "g0c", G0, True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            got = ws["g0_col_scales"] = (ws["rollout_id"], colsc, ev)
        return got[1], got[2]

    def _value_head_weight_grads(self, ws, features, n, s, scratch, cs):
        """Weight and bias gradients of the value head's two linear layers, from ws["dvalue"] and ws["dY"], on the
        current stream; scratch / cs: split-K slabs and fixed-order column-sum partials of that stream."""
        V, gV = self.V, self.gV
        dvalue = ws["dvalue"].view(s * n, 1)
        dY = ws["dY"]
        hip.scale_rows(V["linear2.weight"].expand(s * n, 512), ws["dvalue"], dY)        # dY = dvalue (x) w2
        hip.gemm_auto(dvalue, ws["Y"], gV["linear2.weight"], scratch, a_kmajor=True, b_kmajor=True)  # dw2 = dvalue^T Y
        hip.colsum(dvalue, gV["linear2.bias"], scratch=cs)
        gw1 = gV["linear1.weight"]
        self._pgemm(ws, "gw1h", dY, ws["hv"].view(s * n, H), ws["gw1h"], scratch, self.gd, a_kmajor=True, b_kmajor=True,
                    b_scale=self._unit_scale(ws, H) if self.gd == "f16x2" else None)
        gw1[:, self.F:].copy_(ws["gw1h"])
        hip.colsum(dY.view(s, n * 512), ws["dYsum"].view(-1), scratch=cs)
        hip.gemm(ws["dYsum"], features, gw1[:, :self.F], a_kmajor=True, b_kmajor=True)
        hip.colsum(dY, gV["linear1.bias"], scratch=cs)

    def _value_chain_backward(self, ws, n, p0, s, fc, main, side, plan, ev_loss, features=None, tail=False):
        """The value chain's BPTT and everything that consumes it.  With ev_loss the policy network's backward
        pass is queued as well (first attempt of a step): on the side stream beside the chain, or (tail, see
        _tail_ok) its LSTM's BPTT as one launch in front of the chain's (policy_bptt_first) or behind it and its
        products over all positions beside the chain's launch.  Without ev_loss only the chain's part is redone
        (after a rejected seam check)."""
        V, gV = self.V, self.gV
        lv, tv = p0 + s - 1, hip.chain_positions(n, p0, s, fc)
        tok_flat = ws["tokT"].view(-1)
        saved, dgv = ws["saved"][:tv], ws["dgv"][:tv]
        tail = tail and ev_loss is not None
        def launch_value_chain():
            if plan[0] == "rows":
                hip.lstm_chain_bwd_rows(V["valrnn.lstm.weight_hh_l0"], saved, ws["d_emit"], n, p0, s, dgv,
                                        ws["work_v"], ws["status"][2:3],
                                        persistent=self.persistent and hip.chain_pair_supported(n))
            elif plan[0] == "segmented" and self._bwd_f16x2(plan[1]):
                # owner-computes launch on the f16 matrix instruction (a workspace of its own: the exchanged partial tiles)
                self.last_cheap_bwd = 0
                key = "work_b%d" % plan[1]
                if key not in ws:
                    ws[key] = torch.empty(hip.chain_bwd_f16x2_workspace_floats(plan[1]), dtype=torch.float32, device=self.device)
                hip.lstm_chain_bwd_seg(V["valrnn.lstm.weight_hh_l0"], saved, ws["d_emit"], n, p0, s, plan[1], plan[2],
                                       dgv, ws[key], ws["seam"][4:6], full_calls=fc, arith="f16x2")
            elif plan[0] == "segmented":
                self.last_cheap_bwd = self._cheap_bwd(plan)
                hip.lstm_chain_bwd_seg(V["valrnn.lstm.weight_hh_l0"], saved, ws["d_emit"], n, p0, s, plan[1], plan[2],
                                       dgv, ws["work_v"], ws["seam"][4:6], full_calls=fc,
                                       persistent=self.persistent and hip.chain_pair_supported(plan[1]),
                                       cheap=self.last_cheap_bwd)
            else:
                assert fc == s
                hip.lstm_chain_bwd(V["valrnn.lstm.weight_hh_l0"], saved, ws["d_emit"], n, p0, s, dgv,
                                   ws["sync_v"], ws["status"][2:3])

        pfirst = tail and self.policy_bptt_first
        ev_dhall = rest_of_third_stream = ev_pbptt = ev_vocab = ev_pw = None

        def policy_bptt():
            # the policy LSTM's backward-through-time: one persistent launch
            main.wait_event(ev_dhall)
            arith = "f16x2" if self._bwd_f16x2(n) else "f32"
            if "work_p" not in ws:
                floats = max(hip.chain_seg_workspace_floats(n), hip.chain_bwd_f16x2_workspace_floats(n))
                ws["work_p"] = torch.empty(floats, dtype=torch.float32, device=self.device)
            hip.policy_lstm_bptt_persist(ws["gates"], ws["Cbuf"], ws["dHall"], self.P["lstm.weight_hh_l0"], ws["dgp"],
                                         ws["dh0"], ws["work_p"], ws["status"][3:4], arith=arith)
            self._mark("policy_bptt_done")
            ev = torch.cuda.Event()
            ev.record(main)
            return ev

        if pfirst:
            # The policy LSTM's BPTT FIRST: its inputs (coef * G0 . W) are ready soon after the loss, it needs the whole
            # device for 0.25 ms either way, and with it out of the way the policy network's weight gradients, the
            # vocabulary weight gradient and the replay check all run beside the value chain's BPTT (which leaves half
            # of the CUs free) instead of after it.
            ev_dhall, rest_of_third_stream = self._policy_backward_beside_chain(ws, n, p0, s, ev_loss, side, vocab_later=True)
            ev_pbptt = policy_bptt()
        launch_value_chain()
        self._mark("value_chain_bwd_done")

        def read_back():
            # every word the host reads back (backward seam error, the chain kernels' status words) is final here.
            # Queued before the other streams' work that follows in this function: a device-to-host copy was seen to
            # start only once everything the host had queued before it, on any stream, had been dispatched (1.1 ms
            # after the BPTT it follows, in a kernel trace, with the policy network's weight gradients queued first)
            if plan[0] == "segmented":
                ws["ctl_host"].copy_(ws["ctl"][4:12], non_blocking=True)
                ws["ev_ctl"] = torch.cuda.Event()
                ws["ev_ctl"].record(main)

        if pfirst:
            # The read-back first, then the replay check on the side stream in front of the policy network's weight
            # gradients (64 half-CU workgroups: they fit beside the BPTT's 128 CUs only while the products have not
            # filled the rest; and the host waits for its result before it accepts the forward pass - queued behind the
            # products, on either stream, the step took 5.85 ms against 5.2), the vocabulary weight gradient and the
            # third stream's small launches; the value network's weight gradients follow on the main stream.
            read_back()
            long_replay = self.concurrent and ws.get("replay_steps", 0) >= self.LONG_REPLAY and "replay_deferred" in ws
            if "replay_deferred" in ws:
                ws["ev_replay_late"] = ws.pop("replay_deferred")(ev_pbptt)
            # The policy network's weight gradients: on the side stream behind the replay check - or, when that check
            # is long (it walks as many steps as the warm-ups, 2 ms once the value LSTM's memory has grown, and would
            # hold them back until the BPTT beside which they should run is over), on the third stream behind the
            # vocabulary weight gradient
            ev_pw = None
            wgrads_on = self._check_stream() if long_replay else side

            def policy_weight_grads():
                with torch.cuda.stream(wgrads_on):
                    wgrads_on.wait_event(ev_pbptt)
                    if "order_pending" in ws:
                        n_rows = ws.pop("order_pending")
                        if any(g is not None and g.get(k) is not None for g, k in
                               ((self.gP, "caption_embedding.weight"), (self.gV, "valrnn.caption_embedding.weight"))):
                            hip.token_order(ws["tokT"].view(-1), 1, n_rows[1], self._emb_vocab(), ws["emb_work"])
                            ws["ev_order_done"] = torch.cuda.Event()
                            ws["ev_order_done"].record(wgrads_on)
                    self._policy_weight_grads(ws, features, n, p0, s)
                    ev = torch.cuda.Event()
                    ev.record(wgrads_on)
                return ev

            if not long_replay:
                ev_pw = policy_weight_grads()
            ev_vocab = rest_of_third_stream(ev_pbptt)
            if long_replay:
                ev_pw = policy_weight_grads()
        elif tail:
            # queued BEHIND the chain's launch although it waits for the loss only: launches that the host queues
            # first get the device first, and with two vocabulary-sized products (36 KB of LDS per workgroup, four
            # per CU) in front of it the chain's persistent launch (84 KB per workgroup) found no CU to start on for
            # 0.9 ms (kernel trace: BPTT from 4.43 ms into the step, against 3.60 ms in this order)
            ev_dhall, rest_of_third_stream = self._policy_backward_beside_chain(ws, n, p0, s, ev_loss, side)
            # right behind the value chain's launch (measured: behind the value network's weight gradients instead,
            # 5.70 against 5.45 ms a minibatch)
            ev_pbptt = policy_bptt()
            read_back()
            ev_vocab = rest_of_third_stream(ev_pbptt)
        else:
            read_back()
        # ---- side stream: the policy network's BPTT beside the value chain
        ev_side = None

        def side_section():
            nonlocal ev_side
            if ev_loss is None:
                return
            with torch.cuda.stream(side):
                side.wait_event(ev_loss)
                self._mark("side:policy_bwd_start")
                if ws.get("ev_order") is not None:
                    side.wait_event(ws.pop("ev_order"))
                self._policy_backward(ws, features, n, p0, s)
                self._mark("side:policy_bwd_done")
                self._ev_policy = torch.cuda.Event()
                self._ev_policy.record(side)
                ev_side = self._ev_policy
        if not tail:
            side_section()

        # ---- main stream: the recurrent weight gradient, sum over the positions of dgates (x) h_prev.  The
        # previous position's h is read in place from the kept activations (slot 5 of the position before,
        # a strided view); the chain's first position(s) had h_prev = 0 and drop out of the sum.
        sm = ws["scratch_main"]
        back = n if plan[0] == "rows" else 1
        if tv > back:
            h_prev = saved.view(tv, 6 * H)[:tv - back, 5 * H:]
            f16 = self.gd == "f16x2"
            # one pass for the row and column scales of the gate gradients when the input-side products below read
            # the same matrix (else the positions of this product alone: the matrix may hold rows no launch wrote)
            shared_pass = f16 and fc == 1 and p0 == 1 and s > 1
            dg_rows, dg_cols = self._scales_of(ws, "vdg", dgv) if shared_pass else (None, None)
            self._pgemm(ws, "vhh", dgv[back:], h_prev, gV["valrnn.lstm.weight_hh_l0"], sm, self.gd, a_kmajor=True,
                        b_kmajor=True, a_scale=dg_cols, b_scale=self._unit_scale(ws, H) if f16 else None)
        else:
            gV["valrnn.lstm.weight_hh_l0"].zero_()
        self._mark("value_wgrads_done")
        # ---- then the value network's input-side gradients (the policy network's backward pass outlasts the
        # chain's on the side stream, so these follow the chain on this one)
        same = fc == 1 and p0 == 1 and s > 1 and tv > back and self.gd == "f16x2"
        if fc == 1 and p0 == 1 and s > 1:
            dgsum = dgv          # every column occurs once in the chain: nothing to sum over calls
        else:
            dgsum = ws["dGsum"]
            hip.chain_fold_calls(dgv, n, p0, s, 4 * H, dgsum, full_calls=fc)
        if self.gd == "f16x2":
            sum_rows, sum_cols = (dg_rows, dg_cols) if same else self._scales_of(ws, "vdgs", dgsum[:lv * n])
            emb_cols, wih_cols = ws.get("psc_embV"), ws.get("psc_wihV")
        else:
            sum_rows = sum_cols = emb_cols = wih_cols = None
        if ev_loss is not None and not ws.pop("Xv_gathered", False):
            # the A2C step took its input projections from the per-word table: gather x now
            hip.gather_rows(V["valrnn.caption_embedding.weight"], tok_flat, 1, lv * n, ws["Xv"])
        self._pgemm(ws, "vih", dgsum[:lv * n], ws["Xv"][:lv * n], gV["valrnn.lstm.weight_ih_l0"], sm, self.gd, a_kmajor=True,
                    b_kmajor=True, a_scale=sum_cols, b_scale=emb_cols)
        hip.colsum(dgsum, gV["valrnn.lstm.bias_ih_l0"], rows=lv * n, scratch=ws["cs_main"])
        gV["valrnn.lstm.bias_hh_l0"].copy_(gV["valrnn.lstm.bias_ih_l0"])
        g_emb = gV.get("valrnn.caption_embedding.weight")
        if g_emb is not None:
            if ws.get("ev_order_done") is not None:
                main.wait_event(ws["ev_order_done"])
            self._pgemm(ws, "vdx", dgsum[:lv * n], V["valrnn.lstm.weight_ih_l0"], ws["dXv"][:lv * n], sm, self.gd, b_kmajor=True,
                        a_scale=sum_rows, b_scale=wih_cols)
            hip.embedding_grad(g_emb, lv * n, ws["dXv"], ws["emb_work"])
        self._mark("value_input_grads_done")
        if tail:
            # queued behind the main stream's own products above: the device takes up a stream's launches in the
            # order the host queued them across streams (queued first, these held the main stream's first product back
            # by 0.7 ms in a kernel trace)
            with torch.cuda.stream(side):
                if not pfirst:
                    side.wait_event(ev_pbptt)
                    if ws.get("ev_order") is not None:
                        side.wait_event(ws.pop("ev_order"))
                    self._policy_weight_grads(ws, features, n, p0, s)
                elif ev_pw is not None:
                    side.wait_event(ev_pw)
                side.wait_event(ev_vocab)
                self._mark("side:policy_bwd_done")
                self._ev_policy = torch.cuda.Event()
                self._ev_policy.record(side)
                ev_side = self._ev_policy
        if ev_side is not None:
            main.wait_event(ev_side)

    def _policy_backward(self, ws, features, n, p0, s):
        """Backward of the policy network from ws["dlogp"] = dLoss/dlog_prob (one coefficient per (step, row)).
        d logits = coef * G0 with G0 = onehot - softmax, so the coefficient is applied to small matrices instead
        of the [rows, V] one: what flows into the LSTM outputs is coef * (G0 . W) (G0 . W formed before the loss
        when _policy_backward_early ran), the vocabulary matrix's gradient is G0^T . (coef * Hs) and its bias
        gradient sum_r coef[r] G0[r, :].  The two vocabulary-sized gradients run on the third stream beside the
        LSTM's backward-through-time."""
        P, gP = self.P, self.gP
        t_steps = p0 - 1 + s
        coef = ws["dlogp"]
        G0 = ws["dlogits"].view(s * n, self.vocab)
        Hs = ws["Hbuf"][p0:].view(s * n, H)          # LSTM outputs that fed the vocabulary projection
        ss = ws["scratch_side"]
        cur = torch.cuda.current_stream()
        if ws.get("policy_early"):
            cur.wait_event(ws["ev_early"])
        else:
            self._uvocab(ws, G0, ss)
        dHall = ws["dHall"]
        if p0 > 1:
            dHall[:p0 - 1].zero_()
        hip.scale_rows(ws["Uvocab"], coef, dHall[p0 - 1:].view(s * n, H))
        hip.scale_rows(Hs, coef, ws["HsC"])
        # the vocabulary matrix's gradients: beside the rest when a third stream exists
        other = self._check_stream() if (self.concurrent and ws.get("policy_early")) else cur
        ev_scaled = torch.cuda.Event()
        ev_scaled.record(cur)
        with torch.cuda.stream(other):
            other.wait_event(ev_scaled)
            sc, cs = (ws["scratch_check"], ws["cs_check"]) if other is not cur else (ss, ws["cs_side"])
            self._vocab_weight_grad(ws, G0, sc)
            hip.colsum(G0, gP["linear2vocab.bias"], scratch=cs, row_scale=coef)
            ev_vocab = torch.cuda.Event()
            ev_vocab.record(other)
            pending = ws.pop("head_grads_pending", None) if other is not cur else None
            if pending is not None:   # the value head's weight gradients: off the path to the chain's BPTT
                self._value_head_weight_grads(ws, *pending, sc, cs)
                ws["ev_head_grads"] = torch.cuda.Event()
                ws["ev_head_grads"].record(other)
            if other is not cur:
                self._mark("check:vocab_grads_done")
        # the LSTM's backward-through-time: 3 launches per step, queued by the library in one call (from Python the
        # 57 launches of a 19-step rollout cost about half a millisecond of host time)
        hip.policy_lstm_bptt(ws["gates"], ws["Cbuf"], dHall, P["lstm.weight_hh_l0"], ws["dgp"], ws["dc0"], ws["dc1"],
                             ws["dh0"], ss, shared=True, dtype=self.pd)
        self._policy_weight_grads(ws, features, n, p0, s)
        cur.wait_event(ev_vocab)

    def _redo_policy_bptt(self, ws, features, n, p0, s, main):
        """The policy LSTM's backward-through-time with one launch per step and what follows it, on the main stream
        behind everything the minibatch has queued: for a minibatch whose one-launch BPTT gave up (ws["dHall"], the
        kept gates and cell states are as that launch found them)."""
        ws["status"][3:4].zero_()
        hip.policy_lstm_bptt(ws["gates"], ws["Cbuf"], ws["dHall"], self.P["lstm.weight_hh_l0"], ws["dgp"], ws["dc0"],
                             ws["dc1"], ws["dh0"], ws["scratch_side"], shared=True, dtype=self.pd)
        self._policy_weight_grads(ws, features, n, p0, s)
        self._ev_policy = torch.cuda.Event()
        self._ev_policy.record(main)

    def _policy_weight_grads(self, ws, features, n, p0, s):
        """What follows the policy LSTM's BPTT, on the current stream: the gradients of its weights, biases and
        embedding table and of cnn2linear, from ws["dgp"] (gate gradients of all steps) and ws["dh0"]."""
        P, gP = self.P, self.gP
        t_steps = p0 - 1 + s
        ss = ws["scratch_side"]
        if ws.get("X_pending"):
            hip.gather_rows(P["caption_embedding.weight"], ws["tokT"].view(-1), 1, t_steps * n,
                            ws["X"].view(t_steps * n, -1))
            ws["X_pending"] = False
        dgp = ws["dgp"]
        dg_all = dgp.view(t_steps * n, 4 * H)
        if self.vgd == "f16x2":
            dg_rows, dg_cols = self._scales_of(ws, "pdg", dg_all)
            emb_cols, wih_cols = ws.get("psc_embP"), ws.get("psc_wihP")
        else:
            dg_rows = dg_cols = emb_cols = wih_cols = None
        self._pgemm(ws, "pih", dg_all, ws["X"].view(t_steps * n, -1), gP["lstm.weight_ih_l0"], ss, self.vgd, a_kmajor=True,
                    b_kmajor=True, shared=True, a_scale=dg_cols, b_scale=emb_cols)
        # (Hbuf's first rows are h0 = cnn2linear(features), not an LSTM output: no bound, so its columns get a pass -
        # beside the forward chains when the step has one, see _chains_forward_loss)
        self._pgemm(ws, "phh", dg_all, ws["Hbuf"][:t_steps].view(t_steps * n, H), gP["lstm.weight_hh_l0"], ss, self.vgd,
                    a_kmajor=True, b_kmajor=True, shared=True, a_scale=dg_cols, b_scale=ws.get("psc_hbuf"))
        hip.colsum(dg_all, gP["lstm.bias_ih_l0"], scratch=ws["cs_side"])
        gP["lstm.bias_hh_l0"].copy_(gP["lstm.bias_ih_l0"])
        g_emb = gP.get("caption_embedding.weight")
        if g_emb is not None:
            self._pgemm(ws, "pdx", dg_all, P["lstm.weight_ih_l0"], ws["dX"], ss, self.vgd, b_kmajor=True, shared=True,
                        a_scale=dg_rows, b_scale=wih_cols)
            hip.embedding_grad(g_emb, t_steps * n, ws["dX"], ws["emb_work"])
        hip.gemm_auto(ws["dh0"], features, gP["cnn2linear.weight"], ss, a_kmajor=True, b_kmajor=True, shared=True, dtype=self.pd)
        hip.colsum(ws["dh0"], gP["cnn2linear.bias"], scratch=ws["cs_side"])

    def _policy_backward_beside_chain(self, ws, n, p0, s, ev_loss, side, vocab_later=False):
        """The part of the policy network's backward pass that needs the loss but no BPTT, queued beside the value
        chain's BPTT (see _tail_ok).  Third stream: U = G0 . W (unless an earlier attempt of this minibatch has
        formed it), what flows into the LSTM outputs (coef * U), coef * Hs, then the value head's weight
        gradients.  Side stream: the vocabulary matrix's gradients G0^T . (coef * Hs) and sum_r coef[r] G0[r, :].
        Returns (event after which ws["dHall"] is complete, a function that queues the rest of the third stream's work
        behind the event it is given - the policy BPTT launch's - and returns the event after which the vocabulary
        gradients are complete)."""
        P, gP = self.P, self.gP
        coef = ws["dlogp"]
        G0 = ws["dlogits"].view(s * n, self.vocab)
        Hs = ws["Hbuf"][p0:].view(s * n, H)
        check = self._check_stream()
        with torch.cuda.stream(check):
            check.wait_event(ev_loss)
            if ws.get("policy_early"):
                check.wait_event(ws["ev_early"])
            else:
                self._uvocab(ws, G0, ws["scratch_check"])
                ws["policy_early"], ws["ev_early"] = True, torch.cuda.Event()
                ws["ev_early"].record(check)
            dHall = ws["dHall"]
            if p0 > 1:
                dHall[:p0 - 1].zero_()
            hip.scale_rows(ws["Uvocab"], coef, dHall[p0 - 1:].view(s * n, H))
            ev_dhall = torch.cuda.Event()
            ev_dhall.record(check)
            self._mark("check:policy_dH_done")
            # the vocabulary matrix's gradients follow on the same stream, beside the value chain's BPTT (whose
            # workgroups hold half of the CUs); on the side stream they queued behind the replay check and ran after
            # both BPTT launches, when nothing was left to run beside them
            if not vocab_later:
                hip.scale_rows(Hs, coef, ws["HsC"])
                self._vocab_weight_grad(ws, G0, ws["scratch_check"])

        def order_tokens():
            if "order_pending" in ws:
                n_rows = ws.pop("order_pending")
                if any(g is not None and g.get(k) is not None for g, k in
                       ((self.gP, "caption_embedding.weight"), (self.gV, "valrnn.caption_embedding.weight"))):
                    hip.token_order(ws["tokT"].view(-1), 1, n_rows[1], self._emb_vocab(), ws["emb_work"])
                    ws["ev_order"] = ws["ev_order_done"] = torch.cuda.Event()
                    ws["ev_order"].record(check)

        def after_policy_bptt(ev_pbptt):
            """The rest of the third stream's work, released behind the policy LSTM's BPTT launch: that launch needs
            every CU, so it starts only once the other streams have drained, and these small launches (queued right
            behind the vocabulary weight gradient they held it back by 0.35 ms in a kernel trace) do not need to
            run before it."""
            with torch.cuda.stream(check):
                check.wait_event(ev_pbptt)
                late_check_work()
            return ws.pop("ev_vocab_late")

        def late_check_work():
            if vocab_later:
                # (policy BPTT first: the token order, which the policy network's embedding gradient waits for beside
                # the value chain's BPTT, then the vocabulary weight gradient)
                order_tokens()
                hip.scale_rows(Hs, coef, ws["HsC"])
                self._vocab_weight_grad(ws, G0, ws["scratch_check"])
            hip.colsum(G0, gP["linear2vocab.bias"], scratch=ws["cs_check"], row_scale=coef)
            ev_vocab = torch.cuda.Event()
            ev_vocab.record(check)
            ws["ev_vocab_late"] = ev_vocab
            self._mark("check:vocab_grads_done")
            order_tokens()
            pending = ws.pop("head_grads_pending", None)
            if pending is not None:   # the value head's weight gradients: off the path to the chain's BPTT
                self._value_head_weight_grads(ws, *pending, ws["scratch_check"], ws["cs_check"])
                ws["ev_head_grads"] = torch.cuda.Event()
                ws["ev_head_grads"].record(check)

        with torch.cuda.stream(side):
            side.wait_event(ev_loss)
            self._mark("side:policy_bwd_start")
            if ws.get("X_pending"):   # the embedding rows of the steps that took their input side from the per-word table
                t_steps = p0 - 1 + s
                hip.gather_rows(P["caption_embedding.weight"], ws["tokT"].view(-1), 1, t_steps * n,
                                ws["X"].view(t_steps * n, -1))
                ws["X_pending"] = False
        return ev_dhall, after_policy_bptt
