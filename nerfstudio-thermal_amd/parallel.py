"""Data-parallel training: one process per GPU, rays sharded by rank, ONE gradient all-reduce per step.

The reference wraps the model in torch DDP (pipelines/base_pipeline.py:281-283), whose reducer all-reduces every
parameter gradient in 25 MB buckets over NCCL.  Here the gradients already live in one flat fp32 arena
(arena.py), so the exchange is a single RCCL all-reduce (mean) over the live range -- 77.6 MB in shared mode --
or, optionally, a few large chunks so that the tail of the reduction overlaps the first Adam launches.
Works with backend "nccl" (= RCCL over xGMI on ROCm) and, for CPU tests, "gloo".
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def free_port() -> int:
    """A TCP port nobody listens on right now (for single-process rendezvous; a fixed default collides as soon as two jobs share a host)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world_size).
    MASTER_ADDR / MASTER_PORT come from the launcher (torchrun always sets them); there is deliberately no fixed fallback port."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: launch with torch.distributed.run (it picks and exports the port)")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradAllReducer:
    """Mean all-reduce of the arena's live gradient range AFTER the backward pass: the simple schedule (no side streams beyond the plain step's,
    no collective beside a kernel).  Use as `grad_hook` of RenderEngine.train_step.  force=True issues the collective on a one-rank group too
    (bench.py --force-dp --dp-chunks 0: the measurement wants RCCL's launch in the step)."""

    def __init__(self, world_size: int, chunks: int = 1, group=None, force: bool = False):
        self.world = world_size
        self.chunks = max(1, chunks)
        self.group = group
        self.force = force

    def __call__(self, arena) -> None:
        if self.world <= 1 and not (self.force and dist.is_initialized()):
            return
        lo, hi = arena.live_range
        flat = arena.grads[lo:hi]
        n = flat.numel()
        step = (n + self.chunks - 1) // self.chunks
        # async_op=False: ProcessGroupNCCL enqueues a synchronous collective on the CURRENT stream (torch >= 2.7; older versions run it on the
        # communicator's stream and make the current one wait -- same result).  No stream beyond the plain step's takes part, so this schedule
        # cannot meet the hardware-queue stall of DESIGN.md section 8.0 whatever GPU_MAX_HW_QUEUES says.  The host does not wait either way.
        for s in range(0, n, step):
            dist.all_reduce(flat[s : s + step], op=dist.ReduceOp.SUM, group=self.group, async_op=False)
        if self.world > 1:
            flat.mul_(1.0 / self.world)


class ScheduleGuard:
    """Picks the data-parallel schedule from MEASURED step times, in process, before the run proper.

    The overlapped schedule (OverlappedGradReducer: table level ranges exchanged beside the folds, the proposal networks on a side stream) keeps
    three streams busy.  On this runtime, kernels of two busy hardware queues slow each other down by far more than the work they share, and
    which streams share a queue is decided by GPU_MAX_HW_QUEUES and stream creation order (profiles/r05_dp_hwq_sweep.json:
    the one-rank overlapped step runs at 0.94 ms with the default 4 queues and at 1.25-1.5 ms with 5-16).  The schedule was cut down to three
    streams so that the default suffices, and this guard is the belt to those braces: the caller times a few steps WITHOUT any exchange (the
    plain step), a few with the overlapped schedule and a few with the simple one (GradAllReducer: one all-reduce after the backward, no stream
    beyond the plain step's), and the faster of the two real schedules runs.  Never re-execs, never touches the environment.

    All ranks must take the same branch (the collectives of the two schedules differ): the times are MAX-reduced over the group first.
    The plain steps apply UN-exchanged gradients: afterwards `resync(arena)` makes rank 0's parameters and Adam moments everybody's."""

    def __init__(self, world_size: int, ratio: float = 1.5, group=None):
        self.world, self.ratio, self.group = world_size, float(ratio), group
        self.decision: Optional[dict] = None

    def decide(self, plain_ms: float, overlapped_ms: float, simple_ms: Optional[float] = None) -> dict:
        """Times in ms per step, measured by the caller on THIS rank; every rank gets the same answer (MAX over the group first).
        Without `simple_ms`: the overlapped schedule is dropped when it costs more than ratio x the plain step (the stall signature).
        With `simple_ms` (the simple schedule timed too -- what bench.py does): the FASTER of the two real schedules runs; the overlapped one is
        kept unless the simple one beats it by more than 5 % (on N > 1 an overlapped step may legitimately exceed ratio x plain -- the exchange of
        78 MB over one xGMI link takes longer than the whole backward -- and still be the better schedule)."""
        t = torch.tensor([float(plain_ms), float(overlapped_ms), float("nan") if simple_ms is None else float(simple_ms)], dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")
            nan = torch.isnan(t)
            t = torch.where(nan, torch.full_like(t, float("inf")), t).to(dev)  # (MAX over ranks must not lose a NaN)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            t = t.cpu()
            t = torch.where(torch.isinf(t), torch.full_like(t, float("nan")), t)
        plain, over, simple = float(t[0]), float(t[1]), float(t[2])
        stalled = not (over <= self.ratio * plain)  # (a NaN time counts as a stall)
        if simple_ms is None or simple != simple:
            pick = "simple" if stalled else "overlapped"
        else:
            pick = "overlapped" if (over == over and over <= simple / 0.95) else "simple"
        self.decision = {"schedule": pick, "plain_ms": plain, "overlapped_ms": over, "simple_ms": None if simple != simple else simple,
                         "ratio": over / plain if plain > 0 else float("inf"), "threshold": self.ratio, "stalled": stalled}
        return self.decision

    def resync(self, arena, src: int = 0, grad_scaler=None) -> None:
        """Rank `src`'s parameters and Adam moments become everybody's -- and, with a grad scaler (optim.DeviceGradScaler), its loss scale, growth
        tracker, per-group skip counts and schedule lag too: a rank whose plain leg skipped or backed off would otherwise keep scaling its loss
        differently from the others for the rest of the run (every rank unscales by its OWN scale behind the all-reduce)."""
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            tensors = [arena.params, arena.exp_avg, arena.exp_avg_sq]
            if grad_scaler is not None and getattr(grad_scaler, "enabled", True):
                tensors += [grad_scaler.scale, grad_scaler.growth_tracker, grad_scaler.skipped, grad_scaler.found_inf]
            for t in tensors:
                dist.broadcast(t, src=src, group=self.group)


class InRunScheduleGuard:
    """ScheduleGuard for a run that has no iterations to spare for timing legs (what `ns-train thermal-nerfacto-hip` runs on N > 1:
    trainer.FusedTrainerMixin): the FIRST iterations of the run are the measurement.  `steps_per_leg` iterations run with the overlapped schedule,
    the next `steps_per_leg` with the simple one -- both exchange the same mean gradient, so every one of them is an ordinary training step: no
    plain leg, nothing to re-synchronise afterwards --, the caller times each (a synchronise around the step) and hands the time to `record()`;
    after the second leg the ranks agree on the faster schedule (ScheduleGuard.decide: MAX over the group, overlapped unless the simple one beats
    it by more than 5 %) and `hook` is that schedule's reducer for the rest of the run.  Never re-execs, never touches the environment.
    `log(decision, step)` is called once, on every rank, when the decision falls."""

    def __init__(self, world_size: int, overlapped, simple, steps_per_leg: int = 6, skip: int = 2, group=None, log=None):
        assert steps_per_leg > skip >= 0
        self.world, self.group = world_size, group
        self.hooks = {"overlapped": overlapped, "simple": simple}
        self.steps_per_leg, self.skip, self.log = int(steps_per_leg), int(skip), log
        self.times = {"overlapped": [], "simple": []}
        self.decision: Optional[dict] = None

    @property
    def measuring(self) -> bool:
        return self.decision is None

    @property
    def leg(self) -> str:
        if self.decision is not None:
            return self.decision["schedule"]
        return "overlapped" if len(self.times["overlapped"]) < self.steps_per_leg else "simple"

    @property
    def hook(self):
        return self.hooks[self.leg]

    def record(self, ms: float, step: int = -1) -> Optional[dict]:
        """the wall time of the iteration that just ran with `hook`; returns the decision when this call completed the second leg"""
        if self.decision is not None:
            return None
        self.times[self.leg].append(float(ms))
        if len(self.times["simple"]) < self.steps_per_leg:
            return None
        med = lambda v: float(sorted(v[self.skip:])[len(v[self.skip:]) // 2])  # noqa: E731
        over, simple = med(self.times["overlapped"]), med(self.times["simple"])
        dec = ScheduleGuard(self.world, group=self.group).decide(simple, over, simple)  # (no plain leg: the simple step stands for it)
        dec = {**dec, "measured_on": f"iterations 0-{2 * self.steps_per_leg - 1} of the run ({self.steps_per_leg} per schedule, first {self.skip} of each skipped)"}
        self.decision = dec
        if self.log is not None:
            self.log(dec, step)
        return dec


class _StreamWork:
    """wait() of a collective that was enqueued synchronously on a stream of ours: the current stream waits for the event recorded behind it"""

    def __init__(self, event):
        self.event = event

    def wait(self) -> None:
        torch.cuda.current_stream().wait_event(self.event)


class OverlappedGradReducer:
    """The same mean all-reduce, issued in pieces WHILE the backward pass is still running (what DDP's bucketed reducer does for the
    reference, pipelines/base_pipeline.py:281-283, laid out for this path's gradient arena and xGMI):

      * RenderEngine.loss_and_backward calls `reduce_range(lo, hi)` as soon as a slice of the gradient arena is final -- after each level
        range of the main table's scatter (ops.field_bwd_phase), after the proposal networks' backward on their side stream.  The collective
        is asynchronous: RCCL's stream waits for the work enqueued so far on the CURRENT stream and then runs beside the next scatter
        (atomic-request bound, so the two do not compete for the same resource).  xGMI rings are per-link bound: 4 x 16 MB table slices
        plus one ~10 MB proposal slice keep every message large.
      * `finish()` all-reduces whatever part of the live range has not been reduced yet (MLP weights, embeddings, poses), waits for all
        of it on the current stream and applies the 1/world scale when the backend has no AVG.

    Every rank runs the same Python schedule, so the collectives are issued in the same order everywhere."""

    pipelined = True
    adam_per_range = False  # True: RenderEngine.train_step launches Adam per exchanged range (finish_iter) instead of once after finish()

    def __init__(self, world_size: int, group=None, level_chunks=(6, 6, 4), dense_exchange: bool = False, side_group="auto",
                 transport_dtype: Optional[torch.dtype] = None):
        """level_chunks: how many table levels each successive exchange covers (an int n means n equal ranges).  The last range cannot hide
        behind any fold, so it is the smallest (4 levels = 16 MB, ~50 us on 8 GPUs); the first takes the six coarsest levels: their fold is
        the cheapest part of the backward and nothing can be exchanged before it anyway.  Three ranges, not more: every range costs a fold
        launch, a collective and an Adam launch on the HOST (a torch.distributed collective is ~10x a kernel launch), and the schedule is
        host-bound on a loaded host (bench.py --force-dp: 1.59 ms with 3 ranges / 1.63 ms with 6-4-3-2-1 on a quiet host, far apart on a
        busy one); finer ranges only shave the exposed tail.
        dense_exchange: exchange the coarse levels as dense per-cell sums (2.65 MB instead of 20 MB; tn_field_bwd_scatter_dense + dense fold).
        It needs the round-1 atomic scatter for those levels and two extra launches; with the binned scatter the plain level ranges are
        faster on one rank (bench.py --force-dp), so it is off by default.
        side_group: process group for the proposal networks' exchange, which is issued from the side stream.  A communicator runs its
        collectives in issue order on ONE stream: in the same group as the table ranges, the first table range would queue behind the
        proposal exchange, which in turn waits for the whole level-0 proposal backward.  "auto" creates a second group over the same ranks.
        transport_dtype: None (fp32 on the wire, what DDP sends for the reference) or torch.bfloat16: slices of at least 1 M gradients are
        converted, exchanged and converted back -- half the bytes per link (N = 2 / 4 are link-bound) for two
        element-wise launches per slice and gradients rounded to 8 mantissa bits BEFORE the mean; off by default (it changes numerics)."""
        self.world = world_size
        self.transport_dtype = transport_dtype
        self._casts: dict = {}
        self.group = group
        self.level_chunks = level_chunks
        self.dense_exchange = dense_exchange
        self._side_group = side_group
        self._comm_stream = None
        self._works: List = []
        self._ranges: List[Tuple[int, int]] = []
        self._arena = None
        self._avg = None
        self._after = {}
        self._pieces = {}

    def level_ranges(self, num_levels: int) -> List[Tuple[int, int]]:
        if isinstance(self.level_chunks, int):
            per = -(-num_levels // max(1, self.level_chunks))
            sizes = [per] * max(1, self.level_chunks)
        else:
            sizes = list(self.level_chunks)
        out, lb = [], 0
        for n in sizes:
            if lb >= num_levels:
                break
            out.append((lb, min(num_levels, lb + n)))
            lb += n
        if lb < num_levels:
            out.append((lb, num_levels))
        return out

    def _op(self):
        if self._avg is None:
            # AVG only where it divides by something: RCCL runs AVG as "multiply, then sum" -- with ONE rank a copy kernel over the whole slice
            # (oneRankReduce<FuncPreMulSum>: 116 us of kernel time per step for the 100 MB of gradients, beside the folds), whereas an in-place
            # SUM over one rank launches nothing.  The mean over one rank is the identity either way.
            # (TN_DP_ONE_RANK_AVG=1 keeps AVG there: the A/B switch of that measurement.)
            self._avg = dist.get_backend(self.group) == "nccl" and (self.world > 1 or os.environ.get("TN_DP_ONE_RANK_AVG", "0") == "1")
        return dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM

    def begin(self, arena) -> None:
        assert not self._works, "finish() was not called for the previous step"
        self._arena = arena
        self._ranges = []
        self._after = {}
        self._pieces = {}

    def side_group(self):
        if isinstance(self._side_group, str):  # "auto": every rank reaches this at the same point of the same schedule (collective call)
            self._side_group = dist.new_group() if dist.is_initialized() else None
        return self._side_group

    def _comm(self):
        """The stream the collectives run on (RCCL backend): ONE stream of this object's own, created once.  A synchronous collective is
        enqueued on the current stream (torch >= 2.7), so issuing it inside `with torch.cuda.stream(comm)` puts RCCL's kernels where this
        schedule wants them instead of on the communicator's internal stream -- the step's stream set is then exactly {main, proposal side
        stream, this one}, all from torch's pool.  TN_DP_COMM_STREAM=0: the communicator's own stream (async_op=True), as before round 5."""
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        return self._comm_stream

    def _issue(self, tensor, group=None) -> None:
        if dist.is_initialized():  # a 1-process group still goes through the backend (GPU tests drive RCCL that way)
            if self.transport_dtype is not None and tensor.numel() >= (1 << 20):
                wire = tensor.to(self.transport_dtype)
                self._casts[len(self._works)] = (tensor, wire)  # converted back when the collective has been waited for (finish_iter)
                tensor = wire
            grp = group if group is not None else self.group
            if tensor.is_cuda and dist.get_backend(grp) == "nccl" and os.environ.get("TN_DP_COMM_STREAM", "1") != "0":
                cs, cur = self._comm(), torch.cuda.current_stream()
                cs.wait_stream(cur)  # ordered after everything already enqueued on the current stream
                tensor.record_stream(cs)
                with torch.cuda.stream(cs):
                    dist.all_reduce(tensor, op=self._op(), group=grp, async_op=False)
                    ev = torch.cuda.Event()
                    ev.record(cs)
                self._works.append(_StreamWork(ev))
            else:
                self._works.append(dist.all_reduce(tensor, op=self._op(), group=grp, async_op=True))
        else:
            self._works.append(None)

    def reduce_range(self, lo: int, hi: int, side: bool = False) -> None:
        """Asynchronous mean all-reduce of arena.grads[lo:hi]; ordered after everything already enqueued on the current stream.
        side=True: on the second communicator (the proposal networks, issued from their side stream)."""
        if hi <= lo:
            return
        self._ranges.append((lo, hi))
        # one collective, handed to the optimiser group by group (a range may run over a group boundary: MLP weights + pose behind the table)
        cuts = sorted({b for g in self._arena.optimised_groups for b in self._arena.group_range[g]})
        pts = [lo] + [c for c in cuts if lo < c < hi] + [hi]
        if len(pts) > 2:
            self._pieces[len(self._ranges) - 1] = list(zip(pts[:-1], pts[1:]))
        self._issue(self._arena.grads[lo:hi], self.side_group() if side else None)

    def reduce_tensor(self, tensor, covers: Tuple[int, int], after) -> None:
        """Asynchronous mean all-reduce of a side tensor that STANDS FOR arena.grads[covers[0]:covers[1]] (the dense per-cell sums of the coarse
        table levels: 2.65 MB instead of a 20 MB slice that is almost all zeros).  `after()` is called once the current stream has been made to
        wait for the collective, before the range is handed to the optimiser: it turns the exchanged tensor into the gradient slice."""
        self._ranges.append(tuple(covers))
        self._after[len(self._ranges) - 1] = (tensor, after)
        self._issue(tensor)

    def finish_iter(self, skip: Optional[List[Tuple[int, int]]] = None):
        """Issue the exchange of the rest of the live range (minus `skip`: ranges whose gradients are not used this step; split at the
        optimiser-group boundaries), then yield every exchanged range (lo, hi) in issue order, each as soon as the current stream has been
        made to wait for it -- so the caller's Adam launch for an early range runs while later ranges are still on the wire."""
        lo, hi = self._arena.live_range
        cuts = sorted({b for g in self._arena.optimised_groups for b in self._arena.group_range[g]})
        covered = sorted(self._ranges + list(skip or []))
        cur = lo
        rest = []
        for a, b in covered + [(hi, hi)]:
            if a > cur:
                pts = [cur] + [c for c in cuts if cur < c < min(a, hi)] + [min(a, hi)]
                rest += [(p, q) for p, q in zip(pts[:-1], pts[1:]) if q > p]
            cur = max(cur, b)
        # adjacent leftovers travel as ONE collective (small messages are latency-bound: ~20-30 us each on the ring), but are handed to the
        # optimiser group by group
        merged: List[List[Tuple[int, int]]] = []
        for a, b in rest:
            if merged and merged[-1][-1][1] == a:
                merged[-1].append((a, b))
            else:
                merged.append([(a, b)])
        parts = dict(self._pieces)
        for grp in merged:
            self.reduce_range(grp[0][0], grp[-1][1])
            parts[len(self._ranges) - 1] = grp
        scale = dist.is_initialized() and not self._avg and self.world > 1
        works, ranges, after, casts = self._works, self._ranges, self._after, self._casts
        self._works, self._casts = [], {}
        for k, (w, (a, b)) in enumerate(zip(works, ranges)):
            if w is not None:
                w.wait()  # the current stream waits for the collective
            if k in casts:
                casts[k][0].copy_(casts[k][1])  # reduced-precision transport: back into the fp32 gradient slice
            if k in after:
                tensor, fn = after[k]
                if scale:
                    tensor.mul_(1.0 / self.world)
                fn()
            elif scale:
                self._arena.grads[a:b].mul_(1.0 / self.world)
            for piece in parts.get(k, [(a, b)]):
                yield piece

    def finish(self, skip: Optional[List[Tuple[int, int]]] = None) -> None:
        for _ in self.finish_iter(skip):
            pass

    def __call__(self, arena) -> None:  # plain grad_hook use: everything at the end
        self.begin(arena)
        self.finish()


class ShardedGradReducer(OverlappedGradReducer):
    """The same overlapped exchange with the optimiser state SHARDED over the ranks: a large slice of the gradient arena (the table level
    ranges, the proposal tables) is REDUCE-SCATTERED instead of all-reduced -- rank r receives the mean of its 1/world piece only --, every
    rank runs Adam on the pieces it owns, and the updated PARAMETERS of the slice are all-gathered.  Reduce-scatter + all-gather move the bytes
    one ring all-reduce moves (2 (N-1)/N x the slice per rank), but Adam's 28 B per parameter of HBM traffic -- 543 MB = 70 us per step in
    shared mode -- shrinks to 1/world of it, and only 1/world of the two moment arenas is ever touched per rank.  Small slices (MLP weights,
    embeddings, poses: latency-bound messages) stay all-reduced and replicated.  The parameters stay bit-identical across ranks: every element
    is computed once, by its owner, and copied.

    RenderEngine.train_step drives it (`sharded = True`): begin -> backward with reduce_range() -> finish_iter() yields the ranges THIS rank
    runs Adam on -> gather_params().  Backends without reduce-scatter (gloo: the CPU tests) all-reduce the slice instead; everything behind
    the collective -- ownership, Adam on the owned piece, the parameter all-gather -- is the same code.
    GradScaler semantics: `reduce_flags` (the per-group found_inf flags of what every rank is about to apply, MAX-reduced: one tiny collective);
    with a scaler the Adam launches follow that collective instead of riding behind each exchange (RenderEngine.optimizer_step(ranges=...))."""

    sharded = True
    adam_per_range = True

    def __init__(self, world_size: int, rank: int, min_shard: int = 1 << 20, **kw):
        super().__init__(world_size, **kw)
        assert self.transport_dtype is None, "the sharded exchange moves fp32"
        self.rank = rank
        self.min_shard = min_shard
        self._sharded: dict = {}   # index into _ranges -> (lo, hi) of a reduce-scattered slice
        self._gather: List[Tuple[int, int]] = []

    def _own(self, lo: int, hi: int) -> Tuple[int, int]:
        n = (hi - lo) // self.world
        return lo + self.rank * n, lo + (self.rank + 1) * n

    def _native_scatter(self, group) -> bool:
        """the backend has reduce_scatter_tensor / all_gather_into_tensor (RCCL); gloo does not (tests/test_parallel_cpu.py drives these branches
        through stand-ins built on gloo's collectives)"""
        return dist.get_backend(group) == "nccl"

    def begin(self, arena) -> None:
        super().begin(arena)
        self._sharded = {}
        self._gather = []

    def reduce_range(self, lo: int, hi: int, side: bool = False) -> None:
        n = hi - lo
        cuts = {b for g in self._arena.optimised_groups for b in self._arena.group_range[g]}
        shard = self.world > 1 and n >= self.min_shard and n % (4 * self.world) == 0 and not any(lo < c < hi for c in cuts)
        if not shard:
            return super().reduce_range(lo, hi, side)
        self._ranges.append((lo, hi))
        self._sharded[len(self._ranges) - 1] = (lo, hi)
        grads = self._arena.grads
        group = self.side_group() if side else self.group
        if self._native_scatter(group):
            a, b = self._own(lo, hi)
            self._works.append(dist.reduce_scatter_tensor(grads[a:b], grads[lo:hi], op=self._op(), group=group, async_op=True))
        else:  # no reduce-scatter in this backend: the whole slice is reduced, the owned piece is what gets used
            self._works.append(dist.all_reduce(grads[lo:hi], op=self._op(), group=group, async_op=True))

    def finish_iter(self, skip: Optional[List[Tuple[int, int]]] = None):
        """As OverlappedGradReducer.finish_iter, but a reduce-scattered slice yields only the piece this rank owns (and is remembered for
        gather_params)."""
        # Ownership is resolved PER PIECE, as it is yielded: the parent's generator issues the leftover ranges through self.reduce_range before
        # its first yield, and that may shard them too (separate mode: a whole idle-free proposal group as one leftover of >= min_shard floats).
        # A snapshot of self._sharded taken before the loop would miss those: the full range would get Adam on gradients of which only the
        # owned piece holds the mean, and the parameters would never be gathered.
        for piece in super().finish_iter(skip):
            if piece in self._sharded.values():
                self._gather.append(piece)
                yield self._own(*piece)
            else:
                yield piece

    def reduce_flags(self, found_inf: torch.Tensor) -> None:
        """GradScaler semantics for the sharded schedule: a non-finite value in a reduce-scattered slice reaches its OWNER only (the other ranks
        never see the reduced piece), while torch.amp.GradScaler decides over the whole gradient of an optimiser
        (torch/amp/grad_scaler.py: found_inf per optimiser).  Every rank checks what it is about to apply -- the pieces it owns, the replicated
        small slices -- and the per-group flags are MAX-reduced over the ranks: one collective of num_groups floats behind the last exchange."""
        if self.world > 1:
            dist.all_reduce(found_inf, op=dist.ReduceOp.MAX, group=self.group)

    def gather_params(self) -> None:
        """All-gather of the parameters of every sharded slice (each rank contributes the piece its Adam launch just updated); the current
        stream waits for them: the next forward reads the parameters."""
        works = []
        params = self._arena.params
        for lo, hi in self._gather:
            a, b = self._own(lo, hi)
            if self._native_scatter(self.group):
                works.append(dist.all_gather_into_tensor(params[lo:hi], params[a:b], group=self.group, async_op=True))
            else:
                n = (hi - lo) // self.world
                outs = [params[lo + r * n: lo + (r + 1) * n] for r in range(self.world)]
                works.append(dist.all_gather(outs, params[a:b].clone(), group=self.group, async_op=True))
        for w in works:
            w.wait()
        self._gather = []


def broadcast_params(arena, src: int = 0, group=None) -> None:
    """DDP's initial parameter broadcast from rank 0 (pipelines/base_pipeline.py:282)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(arena.params, src=src, group=group)


def rank_seed(base_seed: int, rank: int) -> int:
    """Every rank draws its own rays with seed + rank (scripts/train.py:97)."""
    return base_seed + rank
