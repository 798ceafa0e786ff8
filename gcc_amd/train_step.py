"""One MoCo pre-training step of train.py:378-434 as a fixed sequence of HIP
launches with no host synchronisation (the reference calls
``torch.cuda.synchronize()`` every step, train.py:433).

    sample(q, k) [side stream, one step ahead] -> positional embedding ->
    encoder(q; model) + encoder(k; model_ema) in shared launches -> MoCo/InfoNCE
    head (+ key all-gather, enqueue) -> head backward -> encoder backward ->
    (gradient all-reduce) -> clip + Adam -> EMA

Parameters of ``model`` and ``model_ema`` are re-homed into flat buffers so that
the optimiser, the gradient all-reduce and the EMA are one launch/collective
each; ``state_dict()`` is unaffected.
"""
from __future__ import annotations

import os
import time

import torch

from .contrast import MemoryMoCo, NceEngine
from .encoder import GraphEncoder, H, grad_params


def flatten_parameters(enc: GraphEncoder):
    """Re-home every parameter of ``enc`` into one flat fp32 buffer, live parameters
    (those that receive gradients on the GIN path, :func:`grad_params` order) first.
    Returns (flat, n_live).  Idempotent."""
    if getattr(enc, "_flat", None) is not None:
        return enc._flat, enc._n_live
    live = [p for _, _, p in grad_params(enc)]
    live_ids = {id(p) for p in live}
    dead = [p for p in enc.parameters() if id(p) not in live_ids]     # set2set.*, lin_readout.* (unused by GIN)
    # --hidden-size below 64: a channel-indexed parameter is the prefix of a zero-padded block (GraphEncoder.ensure_padded);
    # the flat buffers hold the BLOCKS, so clip + Adam, the EMA and the gradient all-reduce run over padded storage (the
    # padding has zero value and zero gradient and stays zero)
    pn = getattr(enc, "padded_numel", lambda t: t.numel())
    n_live = sum(pn(p) for p in live)
    total = n_live + sum(p.numel() for p in dead)
    dev = live[0].device
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    with torch.no_grad():
        for p in live + dead:
            n = p.numel()
            flat[off:off + n].copy_(p.reshape(-1))
            p.data = flat[off:off + n].view_as(p)
            off += pn(p) if id(p) in live_ids else n
    if getattr(enc, "is_padded", lambda: False)():
        # the running statistics (buffers, not parameters) get their padded homes first, then every channel tensor's current
        # storage -- the parameters' blocks inside ``flat`` included -- is recorded as its padded home
        enc._pad_ptrs = {}
        for name, m, a, is_param in enc._channel_tensors():
            if not is_param:
                t = getattr(m, a)
                block = torch.zeros((t.numel() // t.shape[0]) * H, dtype=t.dtype, device=t.device)
                block[: t.numel()].copy_(t.reshape(-1))
                m._buffers[a] = block[: t.numel()].view(t.shape)
        enc.mark_padded()
    enc._flat, enc._n_live = flat, n_live
    return flat, n_live


def moment_update(model, model_ema, m, engine: NceEngine | None = None):
    """train.py:169-172: model_ema = m * model_ema + (1 - m) * model over ALL parameters
    (the reference also averages the unused set2set / lin_readout weights)."""
    eng = engine or NceEngine()
    f1, f2 = getattr(model, "_flat", None), getattr(model_ema, "_flat", None)
    if f1 is not None and f2 is not None and f1.numel() == f2.numel():
        st = torch.cuda.current_stream(f1.device).cuda_stream if f1.is_cuda else None
        eng.ema(f2, f1, m, stream=st)
        return
    for p1, p2 in zip(model.parameters(), model_ema.parameters()):
        st = torch.cuda.current_stream(p2.device).cuda_stream if p2.is_cuda else None
        eng.ema(p2.data, p1.detach().data.contiguous(), m, stream=st)


def clip_grad_norm(params, max_norm):
    """train.py:340-347."""
    if max_norm > 0:
        return torch.nn.utils.clip_grad_norm_(params, max_norm)
    return torch.sqrt(sum(p.grad.data.norm() ** 2 for p in params if p.grad is not None))


class BatchProducer:
    """Runs sampler + positional embedding ahead of the training stream.  This replaces the reference's DataLoader
    worker pool (train.py:577-586) -- same role, same "prefetch" semantics, no processes.

    A producer launch is bound by its slowest subgraph (hub seeds need long eigen-iterations) while most CUs idle,
    so work of many future steps has to be in flight.  The command processor serves FEW queues well: with more than
    four hardware queues, or with two streams sharing one, the ~75 dependent short kernels of a training step take
    2-3x longer (tools/contention_probe.py).  So the look-ahead is not spread over many streams; each of a few
    ``lanes`` (own stream, sampler and eigensolver workspaces) produces a CHUNK of ``chunk`` consecutive steps per
    turn: ``chunk`` sampler calls, then ONE multi-view positional-embedding call over all 2 * chunk views
    (gcc_posemb_multi: the eigensolver kernels pull items from work lists over all views).  Chunk c is produced by
    lane c % lanes into slot (c // lanes) % depth of that lane's buffer rings."""

    def __init__(self, lanes, first_id_fn, device, depth=2, chunk=1, reserved_cus=0, cu_layout="interleaved", ahead=None,
                 gate_heavy=None):
        self.lanes = lanes                      # list of (sampler, posemb): sampler ring >= depth * chunk, posemb ring
        self.first_id = first_id_fn             # >= 2 * depth * chunk buffers, posemb.max_views >= 2 * chunk
        self.dev = device
        self.depth, self.chunk = depth, chunk
        # chunks kept launched beyond the one being consumed: at most lanes * depth - 1 (every ring slot but the
        # consumed chunk's), at least 1
        cap = len(lanes) * depth - 1
        self.ahead = max(1, min(cap, ahead if ahead is not None else len(lanes) * (depth - 1)))
        self.launched = 0                       # chunks launched so far (bench.py: produced == consumed in the window)
        self.late_chunks, self.late_wait_s = 0, 0.0   # chunks the consumer had to wait for, and for how long (host clock)
        self.cuda = torch.device(device).type == "cuda"
        self._owners = []
        if self.cuda and reserved_cus:          # producers stay off `reserved_cus` compute units (gcc_amd/streams.py)
            from .streams import MaskedStream, producer_cus
            self._owners = [MaskedStream(device, producer_cus(reserved_cus, cu_layout)) for _ in lanes]
            self.streams = [o.stream for o in self._owners]
        else:
            self.streams = [torch.cuda.Stream(device) for _ in lanes] if self.cuda else [None] * len(lanes)
        # Optional: the lanes' LDS-heavy eigensolver launches take turns (gcc_amd.posemb.HeavyGate), so that at most one
        # lane's heavy workgroups hold CUs at a time.  Measured (scripts/gpu/r3_call12.sh): no gain -- 1.29 vs 1.28 ms per
        # step sustained, the training kernels are as slow next to 64 heavy workgroups as next to 300 -- so it is OFF
        # unless GCC_POSEMB_GATE=1.
        import os
        if gate_heavy is None:
            gate_heavy = os.environ.get("GCC_POSEMB_GATE", "0") == "1"
        self.gate = None
        if self.cuda and gate_heavy and len(lanes) > 1:
            from .posemb import HeavyGate
            self.gate = HeavyGate()
        self.ready = {}                         # chunk -> (list of (q, k) per step, event)
        self.released = {}                      # chunk -> event recorded on the consumer stream after its last step
        self.next_chunk = 0
        self.prof = None
        self.snap = {}                          # chunk -> (status-snapshot token of its sampler, ring positions before it)
        self.regrown = 0                        # chunks re-sampled after an overflow (sampler scratch / edge capacity grown)

    def _produce(self, c):
        sampler, posemb = self.lanes[c % len(self.lanes)][:2]
        pr = self.prof or {}
        if self.prof is not None:
            self.prof["used"] = True            # this step's marks were recorded (only chunk-launching steps have them)
        s0 = c * self.chunk
        stride = self.first_id(s0 + 1) - self.first_id(s0)
        # ring positions before this chunk: an overflowed chunk is re-sampled into the SAME slots (_reproduce)
        pos = (getattr(sampler, "_next", None), getattr(posemb, "_next", None))
        if self.chunk > 1 and getattr(sampler, "max_steps", 1) > 1:
            # the whole chunk's batches in one launch set per max_steps steps (gcc_sample_multi): the sampler's five kernels
            # are latency chains that one step's 2 * batch_size subgraphs cannot fill the GPU with
            pairs = sampler.sample_multi(self.first_id(s0), self.chunk, stride, prof=pr.get("sampler"))
        else:
            pairs = []
            for step in range(s0, s0 + self.chunk):
                pairs.append(sampler.sample(self.first_id(step), prof=pr.get("sampler") if step == s0 else None))
        if hasattr(sampler, "status_snapshot"):
            # overflow flags of exactly this chunk's sampler calls -> pinned host word (read in get(), before the chunk is
            # consumed); the device word is reset so that the next chunk's snapshot is its own
            self.snap[c] = (sampler.status_snapshot(), pos)
        views = [g for pair in pairs for g in pair]
        pp = pr.get("posemb")
        if hasattr(posemb, "multi"):
            kw = {}
            if pp is not None:
                kw["prof"] = pp
            if self.gate is not None:
                kw["gate"] = self.gate
            posemb.multi(views, **kw)
        else:                                   # placeholder / CPU stand-ins: one view at a time
            for g in views:
                posemb(g)
        return pairs

    def _launch(self, c):
        self.launched += 1
        if not self.cuda:
            self.ready[c] = (self._produce(c), None)
            return
        lane = c % len(self.lanes)
        st = self.streams[lane]
        with torch.cuda.stream(st):
            prev_user = c - len(self.lanes) * self.depth       # the chunk whose buffers this one overwrites
            ev = self.released.pop(prev_user, None)
            if ev is not None:
                st.wait_event(ev)
            pairs = self._produce(c)
            done = torch.cuda.Event()
            done.record(st)
        self.ready[c] = (pairs, done)

    def _reproduce(self, c, bits):
        """Chunk ``c``'s sampler calls overflowed (``bits``): enlarge what overflowed and sample the chunk again into the
        same ring slots -- every subgraph is a pure function of its sample id, so the re-issued chunk is the batch an
        amply sized sampler would have produced.  (Round 3 raised at the next log line and the run was lost.)"""
        sampler, posemb = self.lanes[c % len(self.lanes)][:2]
        _, pos = self.snap.pop(c)
        sampler.grow(bits)                       # raises on a sizing error (node capacity) or beyond its growth limit
        self.regrown += 1
        keep = (getattr(sampler, "_next", None), getattr(posemb, "_next", None))
        if pos[0] is not None:
            sampler._next = pos[0]
        if pos[1] is not None:
            posemb._next = pos[1]
        self.launched -= 1                       # a re-issue, not a new chunk (bench.py: produced == consumed)
        self.ready.pop(c, None)
        self._launch(c)
        if keep[0] is not None:
            sampler._next = keep[0]
        if keep[1] is not None:
            posemb._next = keep[1]

    def prefill(self):
        """Launch the chunks the first :meth:`get` would launch (chunk 0 and the look-ahead) without consuming anything:
        the data pipeline primed, as a DataLoader's workers fill their prefetch queues before the first iteration
        (train.py:577-586).  bench.py calls it before the warm-up steps so that the step count it is asked for is the step
        count it runs."""
        horizon = self.ahead if self.cuda else 0
        while self.next_chunk <= horizon:
            if self.next_chunk not in self.ready:
                self._launch(self.next_chunk)
            self.next_chunk += 1

    def get(self, step, prof=None):
        """Batch of ``step`` (made ready on the current stream); keeps lanes * (depth - 1) chunks in flight."""
        self.prof = prof
        c = step // self.chunk
        horizon = c + self.ahead if self.cuda else c
        if self.next_chunk < c:
            self.next_chunk = c
        while self.next_chunk <= horizon:
            if self.next_chunk not in self.ready:
                self._launch(self.next_chunk)
            self.next_chunk += 1
        attempts = 0
        while c in self.snap:                    # overflow -> grow -> re-sample, at most a few times
            pairs, ev = self.ready[c]
            sampler = self.lanes[c % len(self.lanes)][0]
            if ev is not None:
                if not ev.query():               # the chunk was launched `ahead` chunks ago: normally long complete, and a
                    t_wait = time.perf_counter() # completed event costs one poll instead of a blocking call
                    ev.synchronize()
                    self.late_chunks += 1        # (bench.py reports both: a step that waits here is producer-bound)
                    self.late_wait_s += time.perf_counter() - t_wait
            elif hasattr(sampler, "snapshot_sync"):
                sampler.snapshot_sync()          # produced on the current stream (prefetch off): wait for it
            bits = sampler.read_snapshot(self.snap[c][0])
            if not bits:
                self.snap.pop(c)
                break
            if attempts == 4:                    # (the word just read is the LAST re-issue's: it really still overflows)
                raise RuntimeError(f"sampler overflow persists after growing {attempts} times (chunk {c})")
            self._reproduce(c, bits)
            attempts += 1
        pairs, ev = self.ready[c]
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)
            self.ready[c] = (pairs, None)       # later steps of the chunk need no second wait
        return pairs[step - c * self.chunk]

    def release(self, step):
        c = step // self.chunk
        if step == (c + 1) * self.chunk - 1:    # last step of its chunk
            self.ready.pop(c, None)
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.dev))
                self.released[c] = ev


_CAPTURE_ERROR_MARKS = ("capture", "hipErrorStreamCapture", "cudaErrorStreamCapture", "operation not permitted when stream is capturing",
                        "operation failed due to a previous error during capture")


def _is_capture_error(e):
    """an invalidated / unsupported stream capture (HIP: hipErrorStreamCapture*), as opposed to a failure of the captured body"""
    msg = str(e)
    return any(m in msg for m in _CAPTURE_ERROR_MARKS)


def _hint_rows_once(engine, q, k):
    if not hasattr(engine, "hint_rows"):         # (the any-width engine launches for the capacity)
        return
    _hint_rows_once_(engine, q, k)


def _hint_rows_once_(engine, q, k):
    """The FIRST batch of a run tells the encoder engine how many rows a batch has (one host read of two device integers, once
    per run: every later step stays free of host synchronisation); :func:`read_meters` raises the estimate when a log line finds a
    larger batch.  Graphs captured under an earlier estimate stay valid (any grid is correct)."""
    if engine.rows_hint is None:
        B = q.batch_size
        engine.hint_rows(max(int(q.node_off[B].item()), int(k.node_off[B].item())))


def _meter_buffers(dev):
    """(acc double[5]: sums of loss, prob, gnorm, nodes(q + k), steps; mx int32[2]: max nodes / edges of a q view)
    -- gcc_step_meters adds one step; :func:`read_meters` reads and zeroes them when a log line is due."""
    return torch.zeros(5, dtype=torch.float64, device=dev), torch.zeros(2, dtype=torch.int32, device=dev)


def read_meters(trainer):
    """-> (acc list[5], mx list[2]); synchronises (call once per log line, train.py:418-460)."""
    if hasattr(trainer, "join"):
        trainer.join()                      # the meters are written on the step's stream
    a, m = trainer.meter_acc.tolist(), trainer.meter_max.tolist()
    trainer.meter_acc.zero_()
    trainer.meter_max.zero_()
    if hasattr(getattr(trainer, "gin", None), "hint_rows") and m[0] > 0:
        before = trainer.gin.rows_hint
        trainer.gin.hint_rows(m[0], margin=1.04)       # (the largest q view since the last log line; the k views are as large)
        if before is not None and trainer.gin.rows_hint > before and getattr(trainer, "graphs", None):
            # batches have outgrown the grids the step graphs were captured with (a corpus whose shards differ in size: the first
            # batch came from a small one): a captured grid is baked in, and its workgroups would walk two or three tiles each
            # for the rest of the run.  The slots are captured again, under the new estimate, after their next (eager) step.
            trainer._retired_graphs = list(trainer.graphs.values())
            trainer.graphs = {}
    return a, m


class FlatAdam:
    """clip_grad_norm_ + torch.optim.Adam(lr, betas, eps=1e-8, weight_decay) (train.py:409,667-672) over one flat
    parameter buffer, as two HIP launches (gcc_adam_step).  ``param_groups`` / ``state_dict`` keep the shape
    train.py expects (it sets ``param_group["lr"]`` every step and checkpoints ``optimizer.state_dict()``)."""

    def __init__(self, param, grad, lr, betas, weight_decay, clip_norm, engine, eps=1e-8):
        self.param, self.grad, self.engine = param, grad, engine
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.clip_norm = clip_norm
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.steps = 0
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=param.device)
        self._scratch = torch.zeros(64, dtype=torch.float64, device=param.device)

    def step(self, grad_scale=1.0, ema=None, ema_src=None, ema_m=0.0, meters=None, scalars=None):
        """``grad_scale``: 1 / world when ``grad`` holds the SUM over ranks (folded into the two launches).
        ``ema`` / ``ema_src`` / ``ema_m``: moment_update of the flat EMA buffer from the flat parameter buffer
        (``param`` is its live prefix); ``meters`` = (acc, mx, loss, prob, graph_q, graph_k): one step of the
        device-side meters -- both inside the Adam launch (gcc_adam_ema_step) instead of launches of their own."""
        g = self.param_groups[0]
        self.steps += 1
        st = torch.cuda.current_stream(self.param.device).cuda_stream if self.param.is_cuda else None
        if ema is None and meters is None and scalars is None:
            self.engine.adam(self.param, self.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"], g["eps"],
                             g["weight_decay"], self.steps, self.clip_norm, self.grad_norm, self._scratch, stream=st,
                             grad_scale=grad_scale)
        else:
            self.engine.adam_ema(self.param, self.grad, self.exp_avg, self.exp_avg_sq, g["lr"], g["betas"], g["eps"],
                                 g["weight_decay"], self.steps, self.clip_norm, self.grad_norm, self._scratch, stream=st,
                                 grad_scale=grad_scale, ema=ema, ema_src=ema_src, ema_m=ema_m, meters=meters, scalars=scalars)
        return self.grad_norm

    def zero_grad(self):
        pass                 # the backward kernels overwrite the flat gradient

    def state_dict(self):
        return dict(state=dict(step=self.steps, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq),
                    param_groups=self.param_groups)


class _GraphedStep:
    """hipGraph replay of a fused training step (shared by MoCoTrainStep and E2ETrainStep).

    A step is a FIXED sequence of launches whose arguments depend only on (a) which ring slot holds the batch and (b) a few
    scalars: lr, Adam's step count, the queue's ring pointer, the dropout key.  (b) lives in a device struct
    (gcc_step_scalars) fed through a ring in pinned host memory: the host fills entry n before it submits step n, the
    step's first (captured) launch copies entry (device counter mod ring) into the struct -- no launch of its own between
    two replays.  For (a) there is one captured graph per ring slot (lanes x depth x chunk of them), captured right after the
    slot's first eager step.  What it buys is HOST time (0.43 ms of Python + ~45 launches per step -> 0.05 ms: one host store
    + one graph launch); the stream itself is as fast either way (tools/graph_probe.py), but the host thread also issues the
    producer lanes' launches."""

    RING_LEN = 2048                         # entries of the pinned scalars ring (tests shrink it to exercise the wrap)

    def _graph_init(self, graph):
        self.relaxed_streams = False        # see step(): drop the per-step stream hand-offs (bench.py / train.py loops)
        self._joined_caller = False
        self.use_scalars = False            # kernels read lr / ring pointer / dropout key from self.scalars (tests: eager)
        # default: on with prefetch on a device.  With collectives the step is replayed as SEGMENTS (captured graphs around
        # the RCCL calls, which stay eager launches on RCCL's own stream): the multi-GPU step takes the benchmarked launch path
        self.use_graph = bool(self.prefetch) if graph is None else bool(graph)
        if self.use_graph and self.collectives and self._staged_possible():
            if graph:
                raise ValueError("graph replay with collectives needs RCCL on device buffers (not the gloo staging path)")
            self.use_graph = False
        if self.use_graph and self.dev.type != "cuda":
            raise ValueError("graph replay needs a device")
        if self.use_graph and self.main is None:
            self.main = torch.cuda.Stream(self.dev, priority=-1)        # stream capture cannot run on the default stream
        self.scalars = torch.zeros(24, dtype=torch.uint8, device=self.dev)      # sizeof(gcc_step_scalars)
        self.ring_len = int(self.RING_LEN)
        self.ring = torch.zeros(24 * self.ring_len, dtype=torch.uint8)
        if self.dev.type == "cuda":
            self.ring = self.ring.pin_memory()
        self.ring_count = 0                                                    # host's count of ring steps
        self.ring_counter = torch.zeros(1, dtype=torch.int64, device=self.dev)  # the device's
        self._ring_events = []                                                  # (count, event): run-ahead guard
        self._ring_dirty = False            # a step raised between the host's and the device's count: re-aligned at the next step
        self.graphs = {}                    # ring-slot key -> (segments [(CUDAGraph | eager callable)], outs of the captured step)
        self._graphs_regrown = 0            # producer.regrown the cached graphs were captured under
        self._retired_graphs = []
        self.graph_replays = 0

    def _staged_possible(self):
        return bool(torch.distributed.is_available() and torch.distributed.is_initialized()
                    and torch.distributed.get_backend() == "gloo")

    def _slot_key(self, q, k):
        return tuple(t.data_ptr() for g in (q, k) for t in (g.node_off, g.edge_off, g.row_ptr, g.col_idx, g.graph_id,
                                                            g.pos_undirected))

    def _ring_guard(self):
        """the host must stay less than a ring ahead of the device: every ring_len / 8 ring steps an event is recorded, and a
        slot is only rewritten once the event recorded 3/4 of a ring earlier has completed (in practice it always has)."""
        if self.dev.type != "cuda":
            return
        n = self.ring_count
        every = max(1, self.ring_len // 8)
        if n % every == 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._ring_events.append((n, ev))
        while self._ring_events and self._ring_events[0][0] <= n - (self.ring_len * 3) // 4:
            self._ring_events.pop(0)[1].synchronize()

    @staticmethod
    def _run_segments(body_out):
        """``body_out`` = (segments, result): segments [(capturable, fn)] issued in order on the current stream."""
        segments, result = body_out
        for _, fn in segments:
            fn()
        return result()

    def _run_step(self, q, k, lr, seed, enqueue_index, explicit_masks, pr, st, body):
        """``body(scalars, pr) -> (segments, result)``: ``segments`` = [(capturable, fn)] whose ``fn()`` issue the step's
        launches on the current stream, ``result() -> dict(loss, prob, grad_norm)``.  Eager when dropout masks are injected or
        the caller wants in-step event marks; otherwise the slot's graphs are replayed (captured after the slot's first eager
        step: capture records the launches without executing them).  A step without collectives is ONE capturable segment;
        with collectives the RCCL calls are segments of their own that stay eager (issued between the graph launches)."""
        step_marks = any(n in pr for n in ("gin_fwd", "nce_fwd", "nce_bwd", "gin_bwd"))
        graphed = self.use_graph and not explicit_masks and not step_marks
        scalars = self.scalars if (graphed or self.use_scalars) and not explicit_masks else None
        if self._ring_dirty:
            # an earlier step raised somewhere between the host's ring count and the device's fetch: put them back in step
            torch.cuda.synchronize(self.dev)
            self.ring_counter.fill_(self.ring_count)
            self._ring_dirty = False
        regrown = getattr(getattr(self, "producer", None), "regrown", 0)
        if regrown != self._graphs_regrown:
            # an edge-capacity regrow replaced every ring slot's col_idx: every key changed, the graphs captured over the
            # retired buffers are never replayed again.  They are dropped one regrow later (launches of theirs may be in flight).
            self._retired_graphs = list(self.graphs.values())
            self.graphs = {}
            self._graphs_regrown = regrown
        try:
            if scalars is not None:
                g0 = self.optimizer.param_groups[0]
                self._ring_guard()
                self.nce.fill_scalars(self.ring, self.ring_count % self.ring_len, lr, g0["betas"], self.optimizer.steps + 1,
                                      enqueue_index, seed or 0)
                self.ring_count += 1
            if not graphed:
                return self._run_segments(body(scalars, pr))
            key = self._slot_key(q, k)
            hit = self.graphs.get(key)
            if hit is not None:                                          # replay: one launch per segment
                for item in hit[0]:
                    if isinstance(item, torch.cuda.CUDAGraph):
                        item.replay()
                    else:
                        item()
                self.optimizer.steps += 1
                self.graph_replays += 1
                return dict(hit[1])
            out = self._run_segments(body(scalars, pr))                  # first time this ring slot is consumed: eager ...
            first_ever = not self.graphs and not self._retired_graphs
            self._capture(key, body)                                     # ... then captured for the next time
            if first_ever:
                self._precapture_ready(st)
            return out
        except Exception:
            self._ring_dirty = scalars is not None
            raise

    def _capture(self, key, body):
        """Record ``body``'s launches for ring slot ``key`` (nothing executes): one CUDAGraph per capturable segment, all from
        one memory pool (later segments read what earlier ones allocated)."""
        steps0 = self.optimizer.steps
        self._drain_collectives()
        try:
            segments, result = body(self.scalars, {})
            items, pool = [], None
            for capturable, fn in segments:
                if not capturable:
                    items.append(fn)                                     # (collectives: not run now, issued at every replay)
                    continue
                gobj = torch.cuda.CUDAGraph()
                # (GCC_CAPTURE_MODE=relaxed was tried against the invalidated captures of the collectives path: no difference,
                #  2 of 6 runs died either way -- see _drain_collectives)
                mode = os.environ.get("GCC_CAPTURE_MODE") or "thread_local"
                gobj.capture_begin(capture_error_mode=mode, **({} if pool is None else dict(pool=pool)))
                try:
                    fn()
                finally:
                    gobj.capture_end()
                pool = pool or gobj.pool()
                items.append(gobj)
            cap = result()
        except RuntimeError as e:
            if not _is_capture_error(e):
                # a real failure inside the body (a C-ABI call's rc != 0, a shape or allocation error): not something a later
                # eager step would cure -- a silent fall-back would degrade a multi-hour run to the launch-by-launch path
                raise
            # A capture that the runtime invalidated (seen once in ~10 runs of the collectives path on ROCm 7.0: "operation failed
            # due to a previous error during capture", with RCCL's threads busy beside the capturing one).  Nothing executed and
            # nothing is lost: the slot stays uncaptured, its next step is issued launch by launch and captured again afterwards.
            self.graph_capture_failures = getattr(self, "graph_capture_failures", 0) + 1
            import warnings
            if self.graph_capture_failures > 4:
                # not a stray event but the rule on this system: stop trying, every step is issued launch by launch from here on
                self.use_graph = False
                warnings.warn(f"step graph capture failed {self.graph_capture_failures} times ({e}); graph replay is switched off")
            else:
                warnings.warn(f"step graph capture failed ({e}); the slot stays eager")
            try:
                torch.cuda.synchronize(self.dev)
            except RuntimeError:                                         # (a stream left in capture by a failed capture_end)
                pass
            return False
        finally:
            self.optimizer.steps = steps0                                # the captured body counted a step that did not run
        self.graphs[key] = (items, dict(loss=cap["loss"], prob=cap["prob"], grad_norm=cap["grad_norm"]))
        return True

    def _drain_collectives(self):
        """Before a capture with a process group alive: wait until the group's watchdog thread has nothing left to poll.  The watchdog
        queries the events of the collectives in flight every ~100 ms from ITS thread; on this runtime such a query while another
        thread captures invalidates the capture (in "thread_local" and in "relaxed" mode alike: 2 of 6 runs of 1024 steps, always in
        the start-up burst of captures right after the first eager step's collectives), the stream stays in the invalidated state,
        the next eager collective records its hand-off event on it and the watchdog -- and the process -- die on "an event last
        recorded in a capturing stream" (profiles/r6_collectives_soak_crash.txt; round 5 saw the first half of this once in ~10 runs
        and tolerated it).  With the device idle and the pending-work list empty the watchdog has no event to query, and a capture
        issues no collective (their segments stay eager)."""
        if not self.collectives or self.dev.type != "cuda":
            return
        dist = torch.distributed
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
            return
        torch.cuda.synchronize(self.dev)
        wait = getattr(dist.group.WORLD, "_wait_for_pending_works", None)
        if wait is not None:
            wait()
        else:                                    # (older torch: one watchdog period)
            import time
            time.sleep(0.25)

    def _precapture_ready(self, st):
        """Right after the FIRST step of a run (launched eagerly: every lazily allocated engine buffer exists now), the graphs
        of all the other ring slots the producers have already filled or are filling -- the look-ahead chunks -- are captured
        as well, without an eager step of their own: a capture only records.  The steps that consume those slots are replays
        from the start instead of each paying an eager issue + a capture (+ 0.6 ms per step over the first lanes x depth x
        chunk steps of a run: what a 20-step window right after 5 warm-up steps measured)."""
        prod = getattr(self, "producer", None)
        if prod is None or not hasattr(self, "_capture_body"):
            return
        self.graphs_precaptured = 0
        for c in sorted(prod.ready):
            pairs, _ev = prod.ready[c]
            for q, k in pairs:
                key = self._slot_key(q, k)
                if key not in self.graphs and self._capture(key, self._capture_body(q, k, st)):
                    self.graphs_precaptured += 1

    def _fetch_scalars(self, scalars, st):
        """first launch of a step that uses the device-resident scalars: this step's ring entry -> the device struct"""
        if scalars is not None:
            self.nce.fetch_scalars(scalars, self.ring, self.ring_len, self.ring_counter, stream=st)


class MoCoTrainStep(_GraphedStep):
    def __init__(self, model: GraphEncoder, model_ema: GraphEncoder, contrast: MemoryMoCo, sampler, posemb,
                 learning_rate=0.005, betas=(0.9, 0.999), weight_decay=1e-5, clip_norm=1.0, alpha=0.999,
                 world_size=1, rank=0, prefetch=True, extra_lanes=(), depth=2, lanes=None, chunk=1, reserved_cus=0, cu_layout="interleaved",
                 collectives=None, ahead=None, graph=None, flat_engine=None):
        """``sampler``/``posemb``: producer lane 0; ``extra_lanes``: more (sampler, posemb) pairs with their own
        workspaces for multi-stream prefetch (see :class:`BatchProducer`).
        ``graph``: replay the step's ~45 launches as ONE captured hipGraph per ring slot (default: on with prefetch on a
        device, off with collectives -- see :meth:`_step`)."""
        self.model, self.ema, self.contrast = model, model_ema, contrast
        self.sampler, self.posemb = sampler, posemb
        self.clip_norm, self.alpha = clip_norm, alpha
        self.world, self.rank = world_size, rank
        # collectives run whenever there is more than one rank (``collectives=True`` forces them at world_size 1: tests)
        self.collectives = world_size > 1 if collectives is None else bool(collectives)
        self.dev = next(model.parameters()).device
        self.flat, self.n_live = flatten_parameters(model)
        self.flat_ema, n2 = flatten_parameters(model_ema)
        assert n2 == self.n_live and self.flat.numel() == self.flat_ema.numel()
        # gradient buffer: views in grad_params order
        self.flat_grad = torch.zeros(self.n_live, dtype=torch.float32, device=self.dev)
        self.grad_views, off = [], 0
        for _, _, p in grad_params(model):                              # (blocks are zero-padded when hidden < 64)
            self.grad_views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += model.padded_numel(p)
        self.live = self.flat[: self.n_live]
        # --hidden-size above 64 (train.py:93): the any-width kernels of csrc/ginx.hip (one launch per operator; dense head) under the
        # SAME step -- producer lanes, flat buffers, clip + Adam + EMA + meters as two launches, key all-gather / gradient all-reduce
        # across ranks -- issued launch by launch (their enqueue index / learning rate are by-value arguments: no graph replay)
        self.wide = bool(model.wide or contrast.wide)
        if self.wide and not (model.wide and contrast.wide and model_ema.wide):
            raise ValueError("a wide step needs a wide encoder pair AND a wide head (hidden-size and MemoryMoCo's feature size above 64)")
        self.gin = model.wide_engine() if self.wide else model.engine()
        self.nce = contrast.engine()
        # Adam(lr, betas, weight_decay as L2) over exactly the parameters that get gradients, train.py:667-672
        self.optimizer = FlatAdam(self.live, self.flat_grad, learning_rate, betas, weight_decay, clip_norm,
                                  (flat_engine or NceEngine()) if self.wide else self.nce)      # (the flat-buffer kernels live in csrc/nce.hip; ``flat_engine``: tests)
        self.mask_fn = None          # tests inject explicit dropout keep-masks here; default = in-kernel Philox (wide: torch.rand)
        self.dropout_seed = 0x5EED0000
        self.B = sampler.batch_size
        if contrast.queueSize < self.B * world_size:
            # memory_moco.py:55-61 enqueues all keys of a step with fmod indices; more keys than queue rows would make
            # rows collide (the reference's index_copy_ result is then order dependent), gcc_queue_enqueue refuses it
            raise ValueError(f"nce_k = {contrast.queueSize} is smaller than the {self.B * world_size} keys enqueued per step "
                             f"(batch_size {self.B} x world {world_size}): raise --nce-k")
        self.L = len(model.gnn.ginlayers)
        self.keys_all = torch.empty(self.B * world_size, contrast.inputSize if self.wide else H, device=self.dev) if self.collectives else None
        self.one = torch.ones(1, device=self.dev)
        self.meter_acc, self.meter_max = _meter_buffers(self.dev)
        self.prefetch = prefetch and self.dev.type == "cuda"
        # the ~75 short training kernels of a step must not queue behind the producers' millisecond-long
        # eigensolver workgroups: the step runs on a high-priority stream
        self.main = torch.cuda.Stream(self.dev, priority=-1) if self.prefetch else None
        lanes = list(lanes) if lanes is not None else [(sampler, posemb)] + list(extra_lanes)
        self.producer = BatchProducer(lanes if self.prefetch else lanes[:1], self._first_id,
                                      self.dev if self.prefetch else "cpu", depth=depth if self.prefetch else 1,
                                      chunk=chunk if self.prefetch else 1, reserved_cus=reserved_cus, cu_layout=cu_layout,
                                      ahead=ahead)
        if not self.prefetch:
            self.producer.cuda = False
        if self.wide and graph:
            raise ValueError("graph replay is the 64-channel step's (device-resident scalars); the wide step is issued launch by launch")
        self._graph_init(False if self.wide else graph)
        model.train()                                                    # train.py:357-365
        model_ema.eval()
        for mod in model_ema.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.train()

    def _first_id(self, step):
        return (step * self.world + self.rank) * self.B

    def check_status(self, strict_posemb=False):
        """Synchronising check of the device status words of EVERY producer lane: sampler overflows (scratch / node /
        edge capacity) and refused positional embeddings raise -- the pack kernel leaves a valid but truncated
        subgraph behind and the eigensolver zeros, so training on them would be silent otherwise.  Returns the OR of
        the positional-embedding flag words (bit 8 = an eigen-iteration stopped at its restart cap)."""
        flags, err = 0, None
        for lane in self.producer.lanes:
            smp, pe = lane[0], lane[1]
            try:
                if hasattr(smp, "check_status"):
                    smp.check_status()
                if hasattr(pe, "check_status"):
                    flags |= int(pe.check_status(strict=strict_posemb) or 0)
            except RuntimeError as e:           # keep going: every rank must reach the agreement below
                err = err or e
        if self.collectives and torch.distributed.is_initialized():
            # a rank that raised alone would leave the others waiting in their next collective; the flag words are bit
            # masks, so they are OR-ed (a MAX of masks is not their union) and every rank returns the same word
            t = torch.tensor([1 if err is not None else 0, flags], dtype=torch.int64,
                             device="cpu" if self._staged() or self.dev.type != "cuda" else self.dev)
            every = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size())]
            torch.distributed.all_gather(every, t)
            any_err, flags = 0, 0
            for e in every:
                e = e.tolist()
                any_err |= int(e[0])
                flags |= int(e[1])
            if err is None and any_err:
                err = RuntimeError("another rank reported a sampler / positional-embedding overflow (see its log)")
        if err is not None:
            raise err
        return flags

    # collectives: RCCL on device buffers.  Only when several ranks share one device (bench.py --gpus N on a box with
    # fewer GPUs: a correctness run over gloo) are the same buffers staged through the host.
    def _staged(self):
        return self.dev.type == "cuda" and torch.distributed.get_backend() == "gloo"

    def _all_gather_begin(self, out, x):
        """Key all-gather issued right after the encoder forward: RCCL runs it on its own stream (after the
        work queued on the current one so far); nothing on the training stream waits for it until
        :meth:`_all_gather_end`, just before the enqueue -- the InfoNCE forward / backward and the whole encoder
        backward do not need ``keys_all`` (they read the queue as it was before the enqueue)."""
        if self._staged():
            return ("staged", out, x)
        return ("rccl", torch.distributed.all_gather_into_tensor(out, x, async_op=True))

    def _all_gather_end(self, pending):
        if pending[0] == "staged":
            _, out, x = pending
            o, xi = out.cpu(), x.cpu()
            torch.distributed.all_gather_into_tensor(o, xi)
            out.copy_(o)
        else:
            pending[1].wait()                   # the current stream waits for RCCL's stream; the host does not

    def _all_reduce(self, x):
        if self._staged():
            xi = x.cpu()
            torch.distributed.all_reduce(xi)
            x.copy_(xi)
        else:
            torch.distributed.all_reduce(x)

    # ---- one step
    def step(self, step, lr, prof=None):
        """``prof``: optional dict of gcc_amd.prof.Prof (sampler: 4 marks; gin_fwd/nce_fwd/nce_bwd/gin_bwd: 2).

        Lifetime of the returned batches: ``graph_q`` / ``graph_k`` are RING SLOTS of the producer lanes, valid until the NEXT
        ``step()`` call: that call hands them back on the step's stream, after the caller's stream has been waited for (per-step
        hand-offs, the default) -- kernels the caller enqueued on its own stream in between are covered.  With
        ``relaxed_streams`` read them on the step's stream (``with torch.cuda.stream(trainer.main)``) or synchronise first."""
        if self.main is None:
            return self._step(step, lr, prof)
        caller = torch.cuda.current_stream(self.dev)
        # The step runs on its own (high-priority) stream.  By default every call is bracketed by two stream hand-offs --
        # the step waits for what the caller enqueued, the caller waits for the step -- so that code written against
        # train.py's single-stream semantics stays correct.  Each hand-off is an event record + wait ON THE STEP'S CHAIN
        # (step n + 1 waits for the caller's stream, which waited for step n: ~35 us per hop, 0.07 ms per step in
        # tools/graph_probe.py).  Loops that only read results behind :meth:`join` / a device synchronisation set
        # ``relaxed_streams``: the first call still waits for the caller (weights, queue initialisation), later ones do not.
        if not self.relaxed_streams or not self._joined_caller:
            self.main.wait_stream(caller)
            self._joined_caller = True
        with torch.cuda.stream(self.main):
            out = self._step(step, lr, prof)
        if not self.relaxed_streams:
            caller.wait_stream(self.main)
        return out

    def join(self):
        """Make the caller's current stream wait for the steps issued so far (needed before reading a step's outputs,
        the meters or the weights from the caller's stream when ``relaxed_streams`` is set; harmless otherwise)."""
        if self.main is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.main)
            self._joined_caller = False          # whatever the caller enqueues next is waited for by the next step

    def _step(self, step, lr, prof=None):
        pr = prof or {}
        self._release_pending()
        q, k = self.producer.get(step, prof=prof)
        _hint_rows_once(self.gin, q, k)
        st = torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else None
        p_drop = self.model.gnn.drop.p
        keep = self.mask_fn() if self.mask_fn is not None else None
        seed = (self.dropout_seed + step * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF if p_drop > 0 else None
        c = self.contrast
        for grp in self.optimizer.param_groups:                          # train.py:411-416
            grp["lr"] = lr
        out = self._run_step(q, k, lr, seed, c.index, keep is not None, pr, st,
                             lambda scalars, marks: self._body(q, k, keep, seed, scalars, marks, st))
        c.index = (c.index + self.B * (self.world if self.collectives else 1)) % c.queueSize
        self._pending_release = step
        return dict(out, graph_q=q, graph_k=k)

    def _release_pending(self):
        """Hand the PREVIOUS step's ring slots back to their lane -- at the start of the next step, on the step's stream: the
        "slot free" event then also covers what the caller enqueued after that step returned (reads of ``graph_q`` / ``graph_k`` on
        the step's stream, or on the caller's own stream when the per-step hand-offs are on: ``main.wait_stream(caller)`` has run
        by now).  Rounds 1-5 recorded the event at the END of the step that consumed the slot, before any caller read: a refill
        could overtake a caller's kernels (tests/test_pipeline_gpu.py needed a synchronisation per step)."""
        step = getattr(self, "_pending_release", None)
        if step is not None:
            self._pending_release = None
            self.producer.release(step)

    def _body_wide(self, q, k, keep, S, st):
        """:meth:`MoCoTrainStep._body` on the any-width kernels (csrc/ginx.hip; train.py:93 ``--hidden-size`` above 64): the same five
        stages -- forward of both views | key all-gather begins | head + encoder backward | gradient all-reduce, gather joined | clip +
        Adam + EMA + meters, enqueue -- with explicit dropout masks (torch.rand on the device, as nn.Dropout draws them: gin.py:202,230)
        and by-value scalars."""
        c, enc = self.contrast, self.model
        p_drop = enc.gnn.drop.p

        def fwd():
            kq = keep
            if kq is None and p_drop > 0:
                kq = (torch.rand(len(enc.gnn.ginlayers) + 1, self.B, enc.output_dim, device=self.dev) >= p_drop).float()
            S["pq"], S["bufq"] = self.gin.make_pass(enc, q, training=True, keep=kq, slot=("step", 0))
            S["pk"], S["bufk"] = self.gin.make_pass(self.ema, k, training=True, keep=None, slot=("step", 1))
            self.gin.forward(S["pq"], stream=st)                             # train.py:389
            self.gin.forward(S["pk"], stream=st)                             # train.py:390-391
            self.last_bufs = (S["bufq"], S["bufk"])                          # (tests / bench.py's parity step read the embeddings here)

        def gather_begin():
            S["gathering"] = self._all_gather_begin(self.keys_all, S["bufk"]["feat"])

        def head_and_backward():
            # logits, loss and d loss / d q against the queue BEFORE the enqueue (memory_moco.py:31 clones it): train.py:393,407-408
            S["outs"] = self.nce.forward(S["bufq"]["feat"], S["bufk"]["feat"], c.memory, c.T, 0, stream=st)
            self.gin.backward(enc, S["pq"], S["outs"]["grad_rows"], self.grad_views, stream=st)

        def reduce_and_join():
            self._all_reduce(self.flat_grad)
            self._all_gather_end(S["gathering"])

        def update():
            outs = S["outs"]
            S["gnorm"] = self.optimizer.step(grad_scale=1.0 / self.world if self.collectives else 1.0,
                                             ema=self.flat_ema, ema_src=self.flat, ema_m=self.alpha,
                                             meters=(self.meter_acc, self.meter_max, outs["loss"], outs["prob"], q, k), scalars=None)
            keys = self.keys_all if self.collectives else S["bufk"]["feat"]
            self.nce.enqueue(c.memory, keys, c.index, stream=st)             # memory_moco.py:55-61

        def result():
            return dict(loss=S["outs"]["loss"], prob=S["outs"]["prob"], grad_norm=S["gnorm"])

        if self.collectives:
            segments = [(False, fwd), (False, gather_begin), (False, head_and_backward), (False, reduce_and_join), (False, update)]
        else:
            segments = [(False, lambda: (fwd(), head_and_backward(), update()))]
        return segments, result

    def _capture_body(self, q, k, st):
        seed = 0 if self.model.gnn.drop.p > 0 else None              # (with scalars only dropout on / off matters here)
        return lambda scalars, marks: self._body(q, k, None, seed, scalars, marks, st)

    def _body(self, q, k, keep, seed, scalars, pr, st):
        """The launches of one step on the current stream as (segments, result) for :meth:`_run_step` (eager, or under stream
        capture).  ``scalars``: device gcc_step_scalars the Adam / enqueue / dropout kernels read instead of by-value
        arguments.  Without collectives the step is one capturable segment; with them:
            forward (q, k)  |  key all-gather begins (RCCL, own stream)  |  head fwd + bwd, encoder bwd  |
            gradient all-reduce, all-gather joined  |  clip + Adam + EMA + meters, enqueue
        -- three captured graphs with the two RCCL hand-offs issued between their launches."""
        S = {}
        c = self.contrast
        if self.wide:
            return self._body_wide(q, k, keep, S, st)

        def fwd():
            self._fetch_scalars(scalars, st)
            # (with scalars the by-value seed is an addend to the device-resident one: 0 here)
            S["pq"], S["bufq"] = self.gin.make_pass(self.model, q, training=True, keep=keep, slot=("step", 0),
                                                    dropout_seed=(0 if seed is not None else None) if scalars is not None else seed,
                                                    scalars=scalars)
            S["pk"], S["bufk"] = self.gin.make_pass(self.ema, k, training=True, keep=None, slot=("step", 1), backward=False)
            self.gin.forward([S["pq"], S["pk"]], stream=st, prof=pr.get("gin_fwd"))      # train.py:389-391

        def gather_begin():                    # RCCL, overlapped with everything up to the enqueue
            S["gathering"] = self._all_gather_begin(self.keys_all, S["bufk"]["feat"])

        def head_and_backward():
            feat_q, feat_k = S["bufq"]["feat"], S["bufk"]["feat"]
            S["outs"] = self.nce.forward(feat_q, feat_k, c.kernel_memory(), c.T, 0, stream=st, prof=pr.get("nce_fwd"))   # train.py:393,407
            # The enqueue (memory_moco.py:55-61) is the LAST thing the step does with the queue: logits and their backward
            # are taken against the queue before the update (the reference clones it, memory_moco.py:31), so deferring the
            # update is the same computation -- and it takes the key all-gather off the critical chain.
            dq = self.nce.backward(feat_q, feat_k, c.kernel_memory(), c.T, 0, S["outs"], self.one, stream=st,
                                   prof=pr.get("nce_bwd"))                    # loss.backward(), train.py:408
            self.gin.backward(self.model, S["pq"], S["bufq"], dq, targets=self.grad_views, stream=st, prof=pr.get("gin_bwd"))

        def reduce_and_join():
            self._all_reduce(self.flat_grad)                             # SUM of one flat bucket (248 KiB) over xGMI
            self._all_gather_end(S["gathering"])

        def update():
            # clip (train.py:409) + Adam (train.py:417); the mean over ranks is folded into the two launches
            # ... moment_update (train.py:430-431) and train.py:418-428's meters ride in the Adam launch: the meters read
            # this batch's offsets BEFORE its ring slot is handed back
            outs = S["outs"]
            S["gnorm"] = self.optimizer.step(grad_scale=1.0 / self.world if self.collectives else 1.0,
                                             ema=self.flat_ema, ema_src=self.flat, ema_m=self.alpha,
                                             meters=(self.meter_acc, self.meter_max, outs["loss"], outs["prob"], q, k),
                                             scalars=scalars)
            keys = self.keys_all if self.collectives else S["bufk"]["feat"]
            self.nce.enqueue(c.kernel_memory(), keys, c.index, save=False, stream=st, scalars=scalars)

        def result():
            return dict(loss=S["outs"]["loss"], prob=S["outs"]["prob"], grad_norm=S["gnorm"])

        if self.collectives:
            # (the RCCL calls recorded INTO one graph per slot instead -- torch captures them without complaint -- replay at 1.18 ms per
            #  step against 0.88 for these segments on one rank: profiles/r5_collectives_in_graph.txt; not kept)
            segments = [(True, fwd), (False, gather_begin), (True, head_and_backward), (False, reduce_and_join), (True, update)]
        else:
            segments = [(True, lambda: (fwd(), head_and_backward(), update()))]
        return segments, result


class E2ETrainStep(_GraphedStep):
    """The E2E / in-batch-negatives step of train.py:396-417 (``--nce-k = batch_size - 1`` without ``--moco``,
    BASELINE configs[0]) as a fixed sequence of launches: both views go through ``model`` (two forward launch sets,
    one after the other, so that the BatchNorm running statistics are updated in the reference's order),
    ``out = feat_k feat_q^T / T`` with labels on the diagonal (``NCESoftmaxLossNS``, criterions.py:20-33), gradients of
    both passes accumulated into one flat buffer, clip + Adam as two launches.  Same producer pipeline as
    :class:`MoCoTrainStep`; single GPU (the reference has no data-parallel E2E mode)."""

    def __init__(self, model: GraphEncoder, sampler, posemb, nce_t=0.07, learning_rate=0.005, betas=(0.9, 0.999),
                 weight_decay=1e-5, clip_norm=1.0, prefetch=True, depth=2, lanes=None, chunk=1, ahead=None, engine=None,
                 graph=None):
        self.model = model
        self.sampler, self.posemb = sampler, posemb
        self.T, self.clip_norm = nce_t, clip_norm
        self.world, self.rank, self.collectives = 1, 0, False
        self.dev = next(model.parameters()).device
        self.flat, self.n_live = flatten_parameters(model)
        self.flat_grad = torch.zeros(self.n_live, dtype=torch.float32, device=self.dev)
        self.grad_views, off = [], 0
        for _, _, p in grad_params(model):
            self.grad_views.append(self.flat_grad[off:off + p.numel()].view_as(p))
            off += model.padded_numel(p)
        self.live = self.flat[: self.n_live]
        self.gin = model.engine()
        self.nce = engine if engine is not None else NceEngine()
        self.optimizer = FlatAdam(self.live, self.flat_grad, learning_rate, betas, weight_decay, clip_norm, self.nce)
        self.mask_fn = None          # tests: () -> (keep_q, keep_k); default = in-kernel Philox
        self.dropout_seed = 0x5EED0000
        self.B = sampler.batch_size
        self.one = torch.ones(1, device=self.dev)
        self.meter_acc, self.meter_max = _meter_buffers(self.dev)
        self.prefetch = prefetch and self.dev.type == "cuda"
        self.main = torch.cuda.Stream(self.dev, priority=-1) if self.prefetch else None
        lanes = list(lanes) if lanes is not None else [(sampler, posemb)]
        self.producer = BatchProducer(lanes if self.prefetch else lanes[:1], self._first_id,
                                      self.dev if self.prefetch else "cpu", depth=depth if self.prefetch else 1,
                                      chunk=chunk if self.prefetch else 1, ahead=ahead)
        if not self.prefetch:
            self.producer.cuda = False
        # graph replay is available but OFF by default here: measured 1.241 vs 1.240 ms per step at bsz 256 and 0.872 vs
        # 0.806 ms at bsz 32 (scripts/gpu/r4_call8.sh) -- the E2E step waits for its own latency chain, not for the host
        self._graph_init(False if graph is None else graph)
        model.train()

    def _first_id(self, step):
        return step * self.B

    _staged = MoCoTrainStep._staged
    check_status = MoCoTrainStep.check_status
    step = MoCoTrainStep.step
    join = MoCoTrainStep.join

    def _step(self, step, lr, prof=None):
        pr = prof or {}
        self._release_pending()
        q, k = self.producer.get(step, prof=prof)
        _hint_rows_once(self.gin, q, k)
        st = torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else None
        p_drop = self.model.gnn.drop.p
        keep_q, keep_k = self.mask_fn() if self.mask_fn is not None else (None, None)
        s0 = (self.dropout_seed + 2 * step * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF if p_drop > 0 else None
        for grp in self.optimizer.param_groups:                          # train.py:411-416
            grp["lr"] = lr
        out = self._run_step(q, k, lr, s0, 0, keep_q is not None, pr, st,
                             lambda scalars, marks: self._body(q, k, keep_q, keep_k, s0, scalars, marks, st))
        self._pending_release = step
        return dict(out, graph_q=q, graph_k=k)

    _release_pending = MoCoTrainStep._release_pending

    def _capture_body(self, q, k, st):
        s0 = 0 if self.model.gnn.drop.p > 0 else None
        return lambda scalars, marks: self._body(q, k, None, None, s0, scalars, marks, st)

    def _body(self, q, k, keep_q, keep_k, s0, scalars, pr, st):
        S = {}

        def whole():
            self._fetch_scalars(scalars, st)
            # the k pass's dropout key is the q pass's + the golden-ratio increment; with device-resident scalars the by-value
            # seed of a pass is its addend to the device's (gcc_gin_pass.scalars)
            G = 0x9E3779B97F4A7C15
            if s0 is None:
                sq = sk = None
            elif scalars is not None:
                sq, sk = 0, G
            else:
                sq, sk = s0, (s0 + G) & 0xFFFFFFFFFFFFFFFF
            pq, bufq = self.gin.make_pass(self.model, q, training=True, keep=keep_q, slot=("e2e", 0), dropout_seed=sq, scalars=scalars)
            pk, bufk = self.gin.make_pass(self.model, k, training=True, keep=keep_k, slot=("e2e", 1), dropout_seed=sk, scalars=scalars)
            self.gin.forward([pq], stream=st, prof=pr.get("gin_fwd"))          # feat_q = model(graph_q), train.py:397
            self.gin.forward([pk], stream=st)                                  # feat_k = model(graph_k), train.py:398
            feat_q, feat_k = bufq["feat"], bufk["feat"]
            # out = feat_k feat_q^T / T, CE against arange (train.py:400, criterions.py:27-33): rows = feat_k
            outs = self.nce.forward(feat_k, None, feat_q, self.T, 1, stream=st, prof=pr.get("nce_fwd"))
            dk = self.nce.backward(feat_k, None, feat_q, self.T, 1, outs, self.one, stream=st, prof=pr.get("nce_bwd"))
            dq = self.nce.backward(feat_q, None, feat_k, self.T, 1, outs, self.one, by_mem_row=True, stream=st)
            self.gin.backward(self.model, pq, bufq, dq, targets=self.grad_views, stream=st, prof=pr.get("gin_bwd"))
            self.gin.backward(self.model, pk, bufk, dk, targets=self.grad_views, accumulate=True, stream=st)
            # clip (train.py:409) + Adam (train.py:417), train.py:418-428's meters inside the Adam launch
            gnorm = self.optimizer.step(meters=(self.meter_acc, self.meter_max, outs["loss"], outs["prob"], q, k), scalars=scalars)
            S.update(loss=outs["loss"], prob=outs["prob"], grad_norm=gnorm)

        return [(True, whole)], lambda: dict(S)
