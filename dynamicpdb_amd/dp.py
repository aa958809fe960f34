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
such as the conv tower reports) and in what order they complete; rank 0's observation is broadcast so that every rank cuts identical buckets."""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_bytes=32 << 20, group=None, force=False):
        """params: the trainable parameters in a fixed order (identical on all ranks).  force: run the collectives even
        in a single-rank world (test / single-GPU coverage of the multi-GPU code path)."""
        self.params = list(params)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or force
        if self.active and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradReducer(force=True) needs an initialised process group")
        self.bucket_bytes = bucket_bytes
        self.flat = None
        self.buckets = []            # [(start, end)] element ranges of self.flat
        self._bucket_of = {}         # id(param) -> bucket index
        self._expected = {}          # id(param) -> accumulations per step
        self._fired, self._order = {}, {}
        self._remaining, self._works, self._launched = [], [], []
        self._hooks = []
        self._towers = []
        backend = dist.get_backend(group) if self.active else None
        self._avg = backend == "nccl"     # RCCL averages in the collective; gloo sums and we scale once afterwards

    # ---------------------------------------------------------------- wiring
    def attach(self, model=None):
        if not self.active:
            return self
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_accumulated))
        if model is not None:
            for m in model.modules():
                tower_of = getattr(m, "tower", None)
                if callable(tower_of) and hasattr(m, "_tower"):
                    self._towers.append(m)
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
    def begin_step(self):
        """call before forward: zero the gradients (flat buffer once built) and re-arm the buckets."""
        if not self.active:
            for p in self.params:
                p.grad = None
            return
        self._bind_towers()
        if self.flat is None:
            for p in self.params:
                p.grad = None
            self._fired, self._order = {}, {}
            return
        self.flat.zero_()
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
            raise RuntimeError("a parameter without a gradient in the discovery step received one now; rebuild the reducer")
        if self._launched[b]:
            raise RuntimeError("a gradient arrived after its bucket had been reduced (the step's graph differs from the "
                               "discovery step's); call GradReducer.rebuild()")
        self._remaining[b] -= 1
        if self._remaining[b] == 0:
            self._launch(b)

    def _launch(self, b):
        s, e = self.buckets[b]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works[b] = dist.all_reduce(self.flat[s:e], op=op, group=self.group, async_op=True)
        self._launched[b] = True

    def finish(self):
        """call after backward: reduce whatever has not been launched yet, wait for every bucket (stream-ordered for
        RCCL: the current stream waits, the host does not), average."""
        if not self.active:
            return
        if self.flat is None:
            self._build()
            for b in range(len(self.buckets)):
                self._launch(b)
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for w in self._works:
            w.wait()
        if not self._avg and self.world > 1:
            self.flat.div_(self.world)

    def rebuild(self):
        """forget the discovered structure (the next step is a discovery step again)"""
        for p in self.params:
            p.grad = None
        self.flat, self.buckets, self._bucket_of, self._expected = None, [], {}, {}

    # ---------------------------------------------------------------- discovery -> buckets
    def _build(self):
        live = [p for p in self.params if p.grad is not None]
        index = {id(p): i for i, p in enumerate(self.params)}
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
        if sorted(i for i, _ in plan) != mine:
            raise RuntimeError("ranks disagree on which parameters receive gradients")
        dev, total = live[0].device, sum(p.numel() for p in live)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets, self._bucket_params, self._bucket_of, self._expected = [], [], {}, {}
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
            self._bucket_of[id(p)] = len(self.buckets)
            self._expected[id(p)] = cnt
            cur.append(p)
            off += n
        self.buckets.append((start, off))
        self._bucket_params.append(cur)
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._remaining = [0] * len(self.buckets)
