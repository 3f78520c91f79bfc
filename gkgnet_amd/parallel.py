"""Data-parallel plumbing: one process per GPU, gradients averaged with ONE flat all-reduce.

The path shards over the batch only (every (b, g) k-NN problem is independent — reference
torch_vertex.py:199-202); the single exchange step is the gradient all-reduce the reference gets from
MMDistributedDataParallel (mmcls/apis/train.py:117-125).  On ROCm ``backend='nccl'`` is RCCL over xGMI.
All parameter gradients live in one contiguous bucket (``p.grad`` are views into it), so the step issues a
single large collective instead of one per tensor — the right shape for point-to-point xGMI links.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialises the default process group from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GKG_DIST_BACKEND") == "gloo" and torch.cuda.is_available():
        local %= torch.cuda.device_count()             # ranks may share a device in the single-GPU exercise
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GKG_DIST_BACKEND=gloo lets a multi-rank run share ONE GPU (RCCL refuses duplicate devices): used to
            # exercise the N>1 control flow on a single-GPU box; production is RCCL.
            backend = os.environ.get("GKG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_ZERO_DEFER = []     # [defer(t) -> bool, flush()]: fused registers planes.defer_zero / planes.flush_deferred_zero
_FLUSH = []          # callables that complete gradients still queued for a batched launch (fused.flush_wgrads registers itself)


def flush_pending_grads():
    for f in _FLUSH:
        f()


class GradBucket:
    """Flat gradient storage for a set of parameters + the gradient all-reduce(avg) of the step.

    * ``p.grad`` are views into ONE contiguous buffer.  The fused blocks' backward kernels write weight / BN-parameter
      gradients STRAIGHT into those views (``grad_view``: the autograd engine then adopts the returned view as
      ``p.grad`` without a copy), so :meth:`pack` only moves what ordinary autograd produced elsewhere (stem /
      downsample convolutions, embeddings, the head) — at GKGNet-576 a few MB instead of the 138 MB re-pack.
    * The buffer is cut into ``bucket_bytes`` chunks at parameter boundaries, in REVERSE parameter order (the order the
      backward finishes them).  :meth:`all_reduce` reduces everything at once (what a hipGraph-replayed step uses);
      :meth:`install_overlap_hooks` starts each chunk's all-reduce from a post-accumulate hook as soon as its last
      gradient exists, on the collective stream RCCL picks — the reference's DDP-reducer behaviour
      (mmcls/apis/train.py:117-125) — and :meth:`wait` joins them.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20,
                 find_unused_parameters: bool = False):
        # find_unused_parameters (the reference's DDP flag, apis/train.py:124): a parameter without a gradient on THIS
        # rank may have one on another rank, so after a multi-rank reduce its slot holds the others' average and must be
        # cleared again next step.  False (default): a parameter unused here is unused everywhere (GKGNet: the conv biases
        # in front of train-mode BN), its slot stays zero through the all-reduce and is never re-filled.
        self.find_unused = bool(find_unused_parameters)
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self._zeros = None
        self._offset = {}
        # layout: reverse parameter order, so that a chunk completes while the backward is still running
        o = 0
        self.chunks = []                        # (start, end, [params])
        cur_start, cur_params = 0, []
        per = max(1, bucket_bytes // self.flat.element_size())
        for p in reversed(self.params):
            self._offset[p] = o
            p._gkg_bucket = (self.flat, o)
            cur_params.append(p)
            o += p.numel()
            if o - cur_start >= per:
                self.chunks.append((cur_start, o, cur_params))
                cur_start, cur_params = o, []
        if cur_params:
            self.chunks.append((cur_start, o, cur_params))
        es, base = self.flat.element_size(), self.flat.data_ptr()
        self._slot_ptr = {p: base + self._offset[p] * es for p in self.params}     # (the flat buffer is never reallocated)
        self._views = {p: self.flat[self._offset[p]:self._offset[p] + p.numel()].view_as(p) for p in self.params}
        self._point_grads()
        self._pending = []
        self._hooks = []
        self._ready = {}
        self._complete = set()                      # chunks whose gradients all exist (this backward)
        self._issued = 0                            # chunks 0 .. _issued-1 have had their all-reduce started
        self._zero = {id(p) for p in self.params}   # slots known to hold zeros (the buffer starts zeroed)

    def _view(self, p):
        """p's slot as a tensor of p's shape: ONE view object per parameter, made at construction (40 parameters x a slice and a
        view per step were 0.04 ms of the eager cfg2 step's host time).  parallel.grad_view hands out FRESH views instead: a
        gradient tensor autograd is to adopt as p.grad must not be referenced from anywhere else."""
        return self._views[p]

    def _fill_slot(self, q):
        """Slot of a non-resident parameter: copy its gradient in, or clear it (once) when there is none."""
        v = self._view(q)
        if q.grad is None:
            if id(q) not in self._zero:
                v.zero_()
                self._zero.add(id(q))
        else:
            self._zero.discard(id(q))
            v.copy_(q.grad.reshape(q.shape))
        q.grad = v

    def _point_grads(self):
        for p in self.params:
            p.grad = self._view(p)
            p._gkg_handed = False
            p._gkg_clean = False

    def zero(self):
        self.flat.zero_()
        self._zero = {id(p) for p in self.params}

    def release(self, prezero: bool = False):
        """Detach ``p.grad`` from the bucket so the next backward writes fresh gradients (no per-parameter accumulate
        kernels); follow the backward with :meth:`pack`.

        ``prezero``: clear the WHOLE flat buffer now, with ONE fill launch, and mark every slot as holding zeros.  Backward
        kernels that accumulate into their output with atomics (the streaming weight-gradient kernels: slabs of rows are
        added into a zeroed dW) then skip their own per-weight zero-fill — ``grad_view`` tells them the slot is clean — and
        slots of parameters that receive no gradient are already what :meth:`pack` would make them.  At cfg2 this replaces
        five 4.8 us fill launches per step by one; only valid for callers that do not keep ``.grad``
        across backward passes, which is what release() means anyway."""
        for p in self.params:
            p.grad = None
            p._gkg_handed = False
            p._gkg_deferred = False
            p._gkg_clean = prezero
        if prezero:
            # inside a captured step the clear rides in the step's first launch (the weight-plane refresh: planes.defer_zero)
            if not (_ZERO_DEFER and _ZERO_DEFER[0](self.flat)):
                self.flat.zero_()
            self._zero = {id(p) for p in self.params}
        self._ready = {}
        self._complete = set()
        self._issued = 0

    def _resident(self, p) -> bool:
        g = p.grad
        if g is not None and g.data_ptr() == self._slot_ptr[p] and g.is_contiguous():
            return True
        if getattr(p, "_gkg_deferred", False):
            # the batched weight-gradient launch wrote THIS slot after the backward node had returned its view; if autograd did not
            # adopt that view as p.grad (it cloned: extra references, a hook that replaced the tensor), p.grad is a copy of the
            # still-zero slot and copying it back would wipe the real gradient: the slot is the truth
            p.grad = self._view(p)
            return True
        return False

    def pack(self):
        """Bring every gradient into the flat bucket: gradients the fused kernels already wrote in place are left alone,
        the others are copied with one batched copy; parameters without a gradient get zeros.  Re-points ``p.grad`` at
        the bucket views."""
        flush_pending_grads()
        if _ZERO_DEFER:
            _ZERO_DEFER[1]()             # a clear handed to a weight-plane refresh that never ran (no x6 projection, no gradient slot
                                         # taken in this step): nothing has been written yet, so it is still correct to clear now
        src, dst, zero = [], [], []
        for p in self.params:
            if self._resident(p):
                self._zero.discard(id(p))
                continue
            v = self._view(p)
            if p.grad is None:
                # e.g. the conv biases in front of train-mode BN (exactly zero gradient, never materialised): their slots
                # are cleared ONCE and stay clear — a fill launch per such parameter per step cost 4.5 us each inside
                # the cfg2 step (8 biases: 37 of 1020 us)
                if id(p) not in self._zero:
                    zero.append(v)
                    self._zero.add(id(p))
            else:
                self._zero.discard(id(p))
                src.append(p.grad.reshape(p.shape))
                dst.append(v)
        if zero:
            torch._foreach_zero_(zero)
        if src:
            torch._foreach_copy_(dst, src)
        self._point_grads()

    def _issue_in_order(self, world, upto_all=False):
        """Start the all-reduce of every chunk that may go next: chunks are reduced in FIXED chunk order on every rank
        (chunk i only after chunks 0..i-1), whatever order the hooks fired in — ranks whose backward completes the
        chunks in different orders (unused parameters on some ranks) still issue identical collective sequences."""
        while self._issued < len(self.chunks) and (upto_all or self._issued in self._complete):
            if _ZERO_DEFER:
                _ZERO_DEFER[1]()
            if world > 1:
                flush_pending_grads()             # weight gradients queued for the batched launch: this chunk is about to be read
            start, end, plist = self.chunks[self._issued]
            for q in plist:                                      # gradients not written in place: copy this chunk now
                if not self._resident(q):
                    self._fill_slot(q)
                else:
                    self._zero.discard(id(q))
            if world > 1:
                chunk = self.flat[start:end]
                chunk.div_(world)
                self._pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
                if self.find_unused:
                    self._zero.difference_update(id(q) for q in plist)
            self._issued += 1

    def clip_grad_norm_(self, max_norm: float, eps: float = 1e-6):
        """``torch.nn.utils.clip_grad_norm_(params, max_norm)`` (the reference's ``grad_clip`` of the optimizer hook,
        configs/gkgnet/gkgnet_coco_576.py:126) over every gradient of the bucket, as a norm and a scale launch on the flat
        buffer instead of multi-tensor passes over ~300 parameters.  Call after :meth:`pack` / :meth:`wait` (all gradients
        resident; slots of parameters without a gradient hold zeros).  Returns the total norm (a 0-d tensor)."""
        flush_pending_grads()
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + eps), max=1.0))
        return total

    def all_reduce(self, async_op: bool = False):
        """Average over ranks with ONE collective over the whole buffer.  No-op in a single process."""
        flush_pending_grads()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        self.flat.div_(dist.get_world_size())
        if self.find_unused:
            self._zero.clear()
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)

    # ---- overlapped form (eager training loops)
    def install_overlap_hooks(self):
        """After this call a backward pass all-reduces each chunk as soon as all of its gradients exist.  Use
        ``release()`` before the backward and ``wait()`` after it (instead of ``pack()`` + ``all_reduce()``)."""
        if self._hooks:
            return
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        for ci, (start, end, plist) in enumerate(self.chunks):
            for p in plist:
                def hook(param, ci=ci, n=len(plist), world=world):
                    self._ready[ci] = self._ready.get(ci, 0) + 1
                    if self._ready[ci] < n:
                        return
                    self._complete.add(ci)
                    self._issue_in_order(world)
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))

    def wait(self):
        """Join the chunk all-reduces started during the backward; parameters that received no gradient at all (unused
        this step) are zero-filled and their chunks reduced now, so every rank issues the same collectives."""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # every chunk not reduced during the backward — hooks that never fired (all of its parameters unused: nothing is
        # distinguishable from "done" by the counters alone, hence the explicit _issued cursor) or fired partially — is
        # filled (gradient copied in, or zeros) and reduced now, in chunk order
        self._issue_in_order(world, upto_all=True)
        for w in self._pending:
            w.wait()
        self._pending = []
        self._ready = {}
        self._complete = set()
        self._issued = 0            # re-arm the cursor: a following backward (zero() + backward, or an accumulating one)
        self._point_grads()         # without release() must start its chunk all-reduces again


def grad_view(p: torch.nn.Parameter, shape=None):
    """A fresh view of ``p``'s slot in its gradient bucket (None when ``p`` is not bucketed or its ``.grad`` is currently
    attached, i.e. the step accumulates): backward kernels write the gradient there and return the view, which autograd
    adopts as ``p.grad`` — no separate gradient tensor, no re-pack.  A slot is handed out ONCE per backward
    (``GradBucket.release`` re-arms it)."""
    b = getattr(p, "_gkg_bucket", None)
    if _ZERO_DEFER:
        _ZERO_DEFER[1]()                           # a deferred clear nobody has picked up yet: before the first gradient lands
    if b is None or p.grad is not None or getattr(p, "_gkg_handed", False):
        # handed out already in this backward (a block or weight used twice in one graph): the second backward node gets
        # a fresh tensor and autograd ADDS it to the adopted view — two kernels writing the same slot would keep only
        # the last contribution
        return None
    p._gkg_handed = True
    flat, o = b
    v = flat[o:o + p.numel()]
    v = v.view(p.shape if shape is None else shape)
    v._gkg_slot = True                            # a bucket slot: kernels may fill it after the backward node has returned it
    v._gkg_owner = p                              # (fused._wgrad_defer marks the parameter: see GradBucket._resident)
    if getattr(p, "_gkg_clean", False):           # zeroed by release(prezero=True) and not written since
        p._gkg_clean = False
        v._gkg_zero = True
    return v


def shard_batch(global_batch: int, rank: int, world: int) -> range:
    """Contiguous slice of a global batch owned by ``rank`` (remainder spread over the first ranks)."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank ``src``'s parameters and buffers (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
