"""Data-parallel gradient averaging for the training step (the reference wraps the model in
DistributedDataParallel(find_unused_parameters=True), train_DFOLD_dynamics.py:548-551,615).

One process per GPU; the minibatch of trajectory windows is sharded over ranks and nothing but the parameter
gradients is exchanged.  `GradReducer` keeps every live gradient as a VIEW into one persistent flat fp32
buffer (no cat / copy-back around the collective; the fused Adam's pointer table never changes) cut into buckets in the
order in which gradients become final during backward, and launches each bucket's all-reduce (RCCL over xGMI: backend
"nccl"; "gloo" in the CPU tests) the moment its last gradient has been accumulated -- from autograd's
post-accumulate hooks, or from the conv tower, which finalises the gradient of one shared conv layer at a time while the
remaining layers of its last backward application are still computing (89 % of the bytes, ops.ConvTower.backward).

The first step is a discovery step (plain backward, then one blocking reduction): it observes which parameters receive
a gradient at all (the reference's 91,540 dead ones never do -- no find_unused_parameters graph walk afterwards), how
many accumulation events each sees per step (one for an autograd leaf however often it is used; whatever a custom node
such as the conv tower reports) and in what order they complete; rank 0's observation is broadcast so that every rank
cuts identical buckets.

A step whose graph differs from the discovered one (a parameter used more often, or one that had no gradient in the
discovery step) is still reduced exactly -- `finish()` agrees on it across ranks with one small flag collective and reduces
the late / stray gradients on their own -- and the next step re-discovers.  A bucket in flight is never written: the
moment a bucket's all-reduce is launched its parameters' `.grad` are detached from the flat buffer (set to None), so a
late accumulation lands in a fresh tensor of its own (autograd / ConvTower.finalize_layer allocate one) instead of racing
with the collective on the RCCL stream or being overwritten by the host-staged copy-back; `finish()` averages those late
tensors separately, adds them to the reduced views and re-attaches the views.

`broadcast_parameters` is DDP's start-up broadcast (train_DFOLD_dynamics.py:615: rank 0's 737.7 MB of parameters, and the
optimizer state on a resume) -- the reference seeds every rank differently (:419), so without it N ranks would train N
different models."""
import torch
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def broadcast_parameters(tensors, src=0, group=None):
    """In-place broadcast of rank `src`'s values into every rank's tensors (parameters, buffers, optimizer state), a few
    large collectives: the tensors are packed into flat fp32/any-dtype staging chunks of <= 256 MB (xGMI rings are
    per-link bound -- few, large transfers), broadcast, and copied back.  Returns the number of bytes broadcast."""
    if not _dist_on() or dist.get_world_size(group) == 1:
        return 0
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    by_kind = {}
    for t in tensors:
        if t is None or t.numel() == 0:
            continue
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    total = 0
    cap = 256 << 20
    for (dtype, dev), ts in by_kind.items():
        chunk, size = [], 0
        def flush():
            nonlocal chunk, size
            if not chunk:
                return
            flat = torch.cat([t.detach().reshape(-1) for t in chunk])
            dist.broadcast(flat, src=gsrc, group=group)
            off = 0
            with torch.no_grad():
                for t in chunk:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
            chunk, size = [], 0
        for t in ts:
            nb = t.numel() * t.element_size()
            if chunk and size + nb > cap:
                flush()
            chunk.append(t)
            size += nb
            total += nb
        flush()
    return total


class GradReducer:
    def __init__(self, params, bucket_bytes=32 << 20, group=None, force=False, payload_dtype=None, static_graph=None):
        """params: the trainable parameters in a fixed order (identical on all ranks).  static_graph (default: the environment
        variable DFOLD_DP_STATIC_GRAPH, off): the caller promises that every rank runs the same autograd graph in every step
        (DFOLDv2 training does: the reference wraps it in DistributedDataParallel without find_unused_parameters,
        train_DFOLD_dynamics.py:373-376).  After two clean steps the per-step flag collective of finish() -- and with it the
        host's wait for the device once per step -- is skipped; a late or stray gradient seen LOCALLY then raises instead of
        being repaired (a change on one rank cannot be agreed on without the collective: loud failure, never a silent one).  force: run the collectives even
        in a single-rank world (test / single-GPU coverage of the multi-GPU code path).  payload_dtype: torch.bfloat16
        sends every bucket as bf16 (half the bytes per xGMI link; the bucket is rounded once before and the average once
        after the collective -- relative error 2^-8 per gradient entry, opt-in; default / None = fp32, exact); the
        environment variable DFOLD_DP_GRAD_BF16=1 selects it without a code change."""
        self.params = list(params)
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.group = group
        self.world = dist.get_world_size(group) if _dist_on() else 1
        self.active = self.world > 1 or force
        if self.active and not _dist_on():
            raise RuntimeError("GradReducer(force=True) needs an initialised process group")
        self.bucket_bytes = bucket_bytes
        self.flat = None
        self.buckets = []            # [(start, end)] element ranges of self.flat
        self._bucket_of = {}         # id(param) -> bucket index
        self._expected = {}          # id(param) -> accumulations per step
        self._views = {}             # id(param) -> its view into self.flat
        self._fired, self._order = {}, {}
        self._remaining, self._works, self._launched = [], [], []
        self._late, self._stray = set(), set()
        self._hooks = []
        self._towers = []
        self.rediscoveries = 0
        self._pending_rebuild = False
        backend = dist.get_backend(group) if self.active else None
        self._avg = backend == "nccl"     # RCCL averages in the collective; gloo sums and each bucket is scaled once
        self._stage_host = None           # gloo without device-tensor support: buckets go through host memory (tests)
        self.timing = False               # record how long the stream waits for the collectives in finish()
        self._ev, self.wait_ms = None, []
        # timing mode also records, per bucket, how long before the end of backward its collective was launched (the
        # time available for overlap): events on the compute stream at each launch and at the start of finish()
        self._launch_ev, self._bwd_end_ev, self.bucket_lead_ms = {}, None, []
        import os
        if payload_dtype is None and os.environ.get("DFOLD_DP_GRAD_BF16", "0") == "1":
            payload_dtype = torch.bfloat16
        self.payload_dtype = payload_dtype
        if static_graph is None:
            static_graph = os.environ.get("DFOLD_DP_STATIC_GRAPH", "0") == "1"
        self.static_graph = bool(static_graph)
        self._clean_steps = 0
        self._missing, self._was_discovery = [], False
        self.static_check_every = int(os.environ.get("DFOLD_DP_STATIC_CHECK_EVERY", "64"))
        self._steps_since_check = 0

    # ---------------------------------------------------------------- wiring
    def attach(self, model=None):
        if not self.active:
            return self
        if model is not None:
            for m in model.modules():
                tower_of = getattr(m, "tower", None)
                if callable(tower_of) and hasattr(m, "_tower"):
                    self._towers.append(m)
        # A conv tower hands its layers' gradients over itself (ops.ConvTower.finalize_layer -> mark_ready), one layer at a
        # time while its last backward application is still running.  Its parameters get NO autograd hook: torch fires a
        # post-accumulate hook even when the node returned no gradient for the parameter (observed with torch 2.10: one
        # firing per backward pass, AFTER the tower's last application has returned), which the discovery step would count as
        # a second accumulation event -- every conv bucket (89 % of the gradient bytes) would then wait for it and start its
        # all-reduce only at the end of the tower's backward instead of layer by layer.
        managed = set()
        for m in self._towers:
            ws, bs = m._params()
            managed.update(id(p) for p in list(ws) + list(bs))
        self._tower_managed = managed
        for p in self.params:
            if id(p) not in managed:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_accumulated))
        return self

    def _bind_towers(self):
        for m in self._towers:
            m.tower().on_final = self.mark_ready

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for m in self._towers:
            if m._tower is not None:
                m._tower.on_final = None

    # ---------------------------------------------------------------- per-step protocol
    def reset_structure(self):
        """the caller is about to run a DIFFERENT graph (e.g. another step mode): discover the gradient structure again and,
        under static_graph, run the per-step flag collective again until two clean steps have been seen."""
        self._pending_rebuild = self.flat is not None
        self._clean_steps = 0

    def begin_step(self):
        """call before forward: zero the gradients (flat buffer once built) and re-arm the buckets."""
        if self._ev is not None:             # last step's wait time (events of the previous step have long completed)
            a, b = self._ev
            b.synchronize()
            self.wait_ms.append(a.elapsed_time(b))
            self._ev = None
            if self._bwd_end_ev is not None:
                self.bucket_lead_ms = [round(self._launch_ev[i].elapsed_time(self._bwd_end_ev), 3) if i in self._launch_ev else None
                                       for i in range(len(self.buckets))]
        self._launch_ev, self._bwd_end_ev = {}, None
        if not self.active:
            for p in self.params:
                p.grad = None
            return
        self._bind_towers()
        self._late, self._stray = set(), set()
        if self._pending_rebuild:            # last step's graph differed from the discovered one: discover again
            self._pending_rebuild = False
            self.rebuild()
        if self.flat is None:
            for p in self.params:
                p.grad = None
            self._fired, self._order = {}, {}
            return
        self.flat.zero_()
        # every bucketed gradient must still BE its view of the flat buffer: optimizer.zero_grad(set_to_none=True) or a
        # caller's `p.grad = None` would make autograd allocate a fresh tensor and the bucket would reduce zeros
        for p in self.params:
            v = self._views.get(id(p))
            if v is None:
                p.grad = None
            elif p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        self._remaining = [sum(self._expected[id(p)] for p in ps) for ps in self._bucket_params]
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def _on_accumulated(self, p):
        self.mark_ready(p)

    def mark_ready(self, p):
        """one accumulation into p.grad has been enqueued on the current stream"""
        if not self.active:
            return
        k = id(p)
        if self.flat is None:                     # discovery step
            self._fired[k] = self._fired.get(k, 0) + 1
            self._order.pop(k, None)
            self._order[k] = True                 # position of the LAST accumulation decides the bucket order
            return
        b = self._bucket_of.get(k)
        if b is None:
            # no gradient in the discovery step, one now: its (separate) .grad is reduced on its own in finish(), and the
            # next step discovers again
            self._stray.add(self._index[k])
            return
        if self._launched[b]:
            # more accumulations than discovered: the bucket is in flight (or done) and p.grad was detached from it at
            # launch, so this accumulation went into a fresh tensor; finish() averages it on its own and adds it to the view
            self._late.add(self._index[k])
            return
        self._remaining[b] -= 1
        if self._remaining[b] == 0:
            self._launch(b)

    def _reduce_async(self, t):
        """all-reduce (average for RCCL, sum for gloo) of a flat tensor; returns a closure that completes it"""
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self._stage_host is None and not self._avg and t.is_cuda:
            try:                                  # gloo builds without device-tensor support: stage through the host
                dist.all_reduce(torch.zeros(1, device=t.device), group=self.group)
                self._stage_host = False
            except RuntimeError:
                self._stage_host = True
        if self._stage_host and t.is_cuda:
            h = t.detach().cpu()
            w = dist.all_reduce(h, op=op, group=self.group, async_op=True)

            def done():
                w.wait()
                t.copy_(h)
                if not self._avg and self.world > 1:
                    t.div_(self.world)
            return done
        if self.payload_dtype is not None and t.dtype != self.payload_dtype:
            pay = t.to(self.payload_dtype)        # rounded once; the collective moves half the bytes
            w = dist.all_reduce(pay, op=op, group=self.group, async_op=True)

            def done():
                w.wait()
                t.copy_(pay)
                if not self._avg and self.world > 1:
                    t.div_(self.world)
            return done
        w = dist.all_reduce(t, op=op, group=self.group, async_op=True)

        def done():
            w.wait()                              # stream-ordered for RCCL: the current stream waits, the host does not
            if not self._avg and self.world > 1:
                t.div_(self.world)
        return done

    def _launch(self, b):
        s, e = self.buckets[b]
        if self.timing and self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._launch_ev[b] = ev
        self._works[b] = self._reduce_async(self.flat[s:e])
        self._launched[b] = True
        # nothing may write the bucket while its collective is in flight (RCCL stream / gloo thread + host copy-back): a late
        # accumulation must find no .grad and allocate its own tensor (re-attached in finish())
        for p in self._bucket_params[b]:
            p.grad = None

    def finish(self):
        """call after backward: reduce whatever has not been launched yet, wait for every bucket, average; then agree
        across ranks on whether this step's graph matched the discovered one and repair / re-discover if not."""
        if not self.active:
            return
        dev = None
        if self.timing and torch.cuda.is_available() and self.params and self.params[0].is_cuda:
            self._bwd_end_ev = torch.cuda.Event(enable_timing=True)
            self._bwd_end_ev.record()             # the compute stream has reached the end of backward
        discovery = self.flat is None
        if discovery:
            self._build()
            for b in range(len(self.buckets)):
                self._launch(b)
        self._missing = [b for b in range(len(self.buckets)) if not self._launched[b]]   # buckets whose gradients did not all arrive
        for b in self._missing:
            self._launch(b)
        self._was_discovery = discovery
        dev = self.flat.device
        if self.timing and dev.type == "cuda":
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        for w in self._works:
            w()
        if self.timing and dev.type == "cuda":
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self._ev = (ev0, ev1)
        # re-attach the views; whatever sits in .grad now is a late accumulation (its own tensor, see _launch)
        late = {}
        for p in self.params:
            v = self._views.get(id(p))
            if v is None:
                continue
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                late[self._index[id(p)]] = p.grad
                self._late.add(self._index[id(p)])
            p.grad = v
        self._reconcile(dev, late)

    def _reconcile(self, dev, late):
        """one small collective per step: flags [late parameter i ...| stray parameter i ...], MAX over ranks.  late: this
        rank's late accumulations {parameter index: tensor}"""
        n = len(self.params)
        self._steps_since_check += 1
        if self.static_graph and self._clean_steps >= 2 and self._steps_since_check < self.static_check_every:
            # a MISSING gradient (a parameter of a bucket that received fewer accumulations than discovered: its bucket was
            # only launched by finish()) is as much a graph change as an extra one -- its zeros would be averaged in silently
            if self._late or self._stray or self._missing:
                raise RuntimeError("GradReducer(static_graph=True): the autograd graph of this rank changed (late accumulations of "
                                   f"parameters {sorted(self._late)}, stray gradients of {sorted(self._stray)}, buckets with "
                                   f"missing gradients {self._missing}); run without static_graph to have such steps repaired "
                                   "collectively.  The other ranks are NOT notified: they block in their next collective until "
                                   "the process group's timeout (init_process_group(timeout=...)) ends them")
            return                                # no collective, no host wait: the stream waits, the host runs ahead
        self._steps_since_check = 0               # (static mode: one flag collective every `static_check_every` steps catches a
                                                  #  divergence that only another rank can see)
        # (2 n per-parameter flags + one for "a bucket of this rank was short of gradients while the static promise is armed":
        #  on the periodic check step of the static mode every rank learns of it in the same collective and they fail TOGETHER --
        #  between check steps the rank that sees it raises alone, see above)
        armed = self.static_graph and self._clean_steps >= 2
        flags = torch.zeros(2 * n + 1, dtype=torch.int32)
        for i in self._late:
            flags[i] = 1
        for i in self._stray:
            flags[n + i] = 1
        if armed and self._missing:
            flags[2 * n] = 1
        if self.world > 1:
            stage = self._stage_host or dev.type != "cuda"
            f = flags if stage else flags.to(dev)
            dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.group)
            flags = f.cpu()
        if bool(flags[2 * n]):
            raise RuntimeError("GradReducer(static_graph=True): on some rank a bucket was short of gradients (a parameter stopped "
                               f"receiving one; this rank: buckets {self._missing}) -- every rank raises on this check step; run "
                               "without static_graph to have such steps repaired collectively")
        flags = flags[:2 * n]
        if not bool(flags.any()):
            # two clean steps in the STEADY state arm the static mode: the discovery step (one blocking reduction of
            # everything) verifies nothing about the hook-driven path, and a step that needed finish() to launch a bucket
            # is not clean either
            if not getattr(self, "_was_discovery", False) and not self._missing:
                self._clean_steps += 1
            return
        self._clean_steps = 0
        for i in torch.nonzero(flags[:n]).flatten().tolist():
            p = self.params[i]
            g = late.get(i)
            if g is None:                         # another rank's late accumulation: this rank contributes zeros
                g = torch.zeros_like(p)
            g = g.contiguous()
            self._reduce_async(g.view(-1))()
            p.grad.add_(g)                        # view = avg(in time) + avg(late)
        for i in torch.nonzero(flags[n:]).flatten().tolist():
            p = self.params[i]
            if p.grad is None:                    # another rank's stray: this rank contributes zeros
                p.grad = torch.zeros_like(p)
            self._reduce_async(p.grad.view(-1))()
        self.rediscoveries += 1
        # this step's (now exact) gradients stay where they are for the optimizer; the next begin_step forgets the
        # structure and discovers again
        self._pending_rebuild = True

    def profile_buckets(self, reps=3):
        """Each bucket's collective in isolation (blocking, back to back; call between steps, on every rank): milliseconds
        and the achieved bus bandwidth 2 (n - 1) / n * bytes / time -- what one link of the xGMI ring sustains for that
        message size.  Leaves the gradients untouched (works on a scratch copy)."""
        out = []
        if not self.active or self.flat is None or not self.flat.is_cuda:
            return out
        n = self.world
        for b, (s, e) in enumerate(self.buckets):
            scratch = self.flat[s:e].clone()
            self._reduce_async(scratch)()             # warm-up (communicator set-up, first-touch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                self._reduce_async(scratch)()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            nbytes = (e - s) * (2 if self.payload_dtype == torch.bfloat16 else 4)
            out.append({"bucket": b, "mb": round(nbytes / 1e6, 2), "isolated_ms": round(ms, 3),
                        "bus_GBps": round(2.0 * (n - 1) / max(n, 1) * nbytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else None})
        return out

    def rebuild(self):
        """forget the discovered structure (the next step is a discovery step again)"""
        for p in self.params:
            p.grad = None
        self.flat, self.buckets, self._bucket_of, self._expected, self._views = None, [], {}, {}, {}

    # ---------------------------------------------------------------- discovery -> buckets
    def _build(self):
        live = [p for p in self.params if p.grad is not None]
        index = self._index
        # order of completion and accumulation counts as seen by rank 0, identical buckets everywhere
        seen = [(index[k], self._fired[k]) for k in self._order if k in index]
        missing = [index[id(p)] for p in live if id(p) not in self._fired]       # gradient assigned without any hook
        obj = [seen + [(i, 1) for i in missing]]
        if self.world > 1:
            dist.broadcast_object_list(obj, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                       group=self.group)
        plan = obj[0]
        self._plan = plan
        mine = sorted(index[id(p)] for p in live)
        ok = sorted(i for i, _ in plan) == mine
        if self.world > 1:                        # the verdict is collective: a lone raise would leave the others hanging
            oks = [None] * self.world
            dist.all_gather_object(oks, ok, group=self.group)
            ok = all(oks)
        if not ok:
            raise RuntimeError("ranks disagree on which parameters receive gradients in the discovery step")
        dev, total = live[0].device, sum(p.numel() for p in live)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets, self._bucket_params, self._bucket_of, self._expected, self._views = [], [], {}, {}, {}
        off, start, cur = 0, 0, []
        for i, cnt in plan:
            p = self.params[i]
            if p.dtype != torch.float32:
                raise ValueError("GradReducer expects fp32 parameters")
            n = p.numel()
            if cur and (off - start + n) * 4 > self.bucket_bytes:
                self.buckets.append((start, off))
                self._bucket_params.append(cur)
                start, cur = off, []
            view = self.flat[off:off + n].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            self._views[id(p)] = view
            self._bucket_of[id(p)] = len(self.buckets)
            self._expected[id(p)] = cnt
            cur.append(p)
            off += n
        self.buckets.append((start, off))
        self._bucket_params.append(cur)
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._remaining = [0] * len(self.buckets)
