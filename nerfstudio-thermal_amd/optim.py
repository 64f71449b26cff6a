"""Optimiser side of the drop-in path: torch.optim.Adam's interface over the flat parameter arena.

The reference builds one `torch.optim.Adam(lr, eps=1e-15)` per parameter group from `AdamOptimizerConfig._target`
(engine/optimizers.py:40-112, configs/method_configs.py:274-307) and steps them all after `loss.backward()`
(engine/trainer.py:489-499).  `HipFusedAdam` is a `torch.optim.Optimizer` with the same constructor arguments and state layout
(`state[p] = {"step", "exp_avg", "exp_avg_sq"}`), so `Optimizers`, LR schedulers, `state_dict()` / `load_state_dict()` and checkpoints
work unchanged, but `step()` is one `tn_adam_step_ranges` launch over the group's slice of the arena: the moments are views of the arena's
moment buffers, and a gradient that autograd left aliased to the arena's gradient buffer is consumed in place.

Parameters whose `.grad` is None are skipped and keep their step count, exactly as torch.optim.Adam does (the proposal networks on
iterations where the sampler ran them under no_grad, model_components/ray_samplers.py:605-610).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from . import ops
from .arena import ParamArena

_FUSED_SCALER_STEP = os.environ.get("TN_FUSED_SCALER_STEP", "1") != "0"  # A/B switch (read once at import): 0 = one GradScaler.step() per optimiser


class HipFusedAdam(torch.optim.Optimizer):
    # torch.amp.GradScaler.step() hands `grad_scale` / `found_inf` (device tensors) to optimisers that declare this and lets THEM skip the step
    # on the device (torch/amp/grad_scaler.py: the branch torch's fused Adam takes) instead of synchronising the host on found_inf.item()
    _step_supports_amp_scaling = True

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, **kwargs):
        if weight_decay != 0.0 or kwargs.get("amsgrad", False) or kwargs.get("maximize", False):
            raise NotImplementedError("HipFusedAdam implements plain Adam (weight_decay = 0, no amsgrad): what the thermal-nerfacto optimisers use")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        self._where: Dict[int, tuple] = {}  # id(p) -> (arena, offset, numel)
        self._steps: Dict[int, int] = {}    # id(p) -> Adam step count (torch keeps a tensor per parameter; a Python int costs nothing per step)
        self._plans: Dict[int, tuple] = {}
        self._skipped: Optional[torch.Tensor] = None  # device int32: steps skipped on found_inf (created with the first AMP step)
        for group in self.param_groups:
            for p in group["params"]:
                arena = ParamArena.owner_of(p)
                if arena is None:
                    raise ValueError("HipFusedAdam only optimises parameters that live in a ParamArena (ThermalNerfactoModel.get_param_groups())")
                off = (p.data_ptr() - arena.params.data_ptr()) // 4
                self._where[id(p)] = (arena, off, p.numel())
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": arena.exp_avg[off:off + p.numel()].view(p.shape),
                                 "exp_avg_sq": arena.exp_avg_sq[off:off + p.numel()].view(p.shape)}

    def _group_plan(self, gi: int):
        """Per group, once: (parameter, address its gradient has when autograd left it aliased to the arena, arena offset, numel) in arena
        order, and the merged ranges of the whole group (alignment padding between neighbours carries a zero gradient: Adam leaves it alone)."""
        plan = self._plans.get(gi)
        if plan is None:
            rows, runs = [], []
            for p in self.param_groups[gi]["params"]:
                arena, off, n = self._where[id(p)]
                rows.append((p, arena.grads.data_ptr() + 4 * off, off, n, arena))
                if runs and runs[-1][0] is arena and 0 <= off - runs[-1][2] < ParamArena.ALIGN:
                    runs[-1][2] = off + n
                else:
                    runs.append([arena, off, off + n])
            plan = self._plans[gi] = (rows, runs)
        return plan

    @torch.no_grad()
    def collect_runs(self):
        """This step's work as (arena, lo, hi, step count, lr, beta1, beta2, eps) per contiguous run of parameters that have a gradient; advances
        the per-parameter step counts (what step() does before it launches)."""
        steps = self._steps
        out = []
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            rows, full_runs = self._group_plan(gi)
            # fast path (every iteration of a normal run): every parameter of the group has a gradient, in place in the arena, same step count
            k0 = steps.get(id(rows[0][0]), 0) if rows else 0
            fast = bool(rows)
            for p, gptr, _, _, _ in rows:
                g = p.grad
                if g is None or g.data_ptr() != gptr or steps.get(id(p), 0) != k0:
                    fast = False
                    break
            if not fast and rows and any(p.grad is None for p, _, _, _, _ in rows):
                self._partial_steps = True  # (see _sync_step_tensors: skipped steps are counted per optimiser, not per parameter)
            if fast:
                k = k0 + 1
                for row in rows:
                    steps[id(row[0])] = k
                runs = [(r[0], r[1], r[2], k) for r in full_runs]
            else:
                runs = []
                for p, gptr, off, n, arena in rows:
                    g = p.grad
                    if g is None:
                        continue  # torch.optim.Adam skips it too, and does not advance its step count
                    if g.data_ptr() != gptr:  # autograd kept its own buffer (e.g. two gradient sources were summed)
                        arena.grads[off:off + n].copy_(g.reshape(-1))
                        arena.grads_clean = False
                    k = steps.get(id(p), 0) + 1
                    steps[id(p)] = k
                    if runs and runs[-1][0] is arena and runs[-1][3] == k and 0 <= off - runs[-1][2] < ParamArena.ALIGN:
                        runs[-1][2] = off + n
                    else:
                        runs.append([arena, off, off + n, k])
            out += [(r[0], r[1], r[2], r[3], group["lr"], b1, b2, group["eps"]) for r in runs]
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # set by torch.amp.GradScaler.step() for this call only (and deleted by it afterwards)
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        amp = found_inf is not None or grad_scale is not None
        runs = self.collect_runs()
        launches = []
        i = 0
        while i < len(runs):
            a, b1, b2, eps = runs[i][0], runs[i][5], runs[i][6], runs[i][7]
            chunk = []
            while i < len(runs) and len(chunk) < 8 and runs[i][0] is a and runs[i][5:] == (b1, b2, eps):
                chunk.append(runs[i])
                i += 1
            rng = [(r[1], (r[2] + 3) // 4 * 4, r[3], r[4]) for r in chunk]
            if amp:
                launches.append((a, rng, b1, b2, eps))
            else:
                ops.adam_step_ranges(a.params, a.grads, a.exp_avg, a.exp_avg_sq, rng, beta1=b1, beta2=b2, eps=eps)
        if launches:
            # GradScaler path: the skip / unscale decision stays on the device; only the LAST launch of this step() counts a skipped step
            if self._skipped is None:
                self._skipped = torch.zeros(1, dtype=torch.int32, device=launches[0][0].params.device)
            inv = None if grad_scale is None else grad_scale.to(torch.float32).reciprocal().reshape(1)
            fi = None if found_inf is None else found_inf.to(torch.float32).reshape(1)
            for k, (a, rng, b1, b2, eps) in enumerate(launches):
                ops.adam_step_ranges_amp(a.params, a.grads, a.exp_avg, a.exp_avg_sq, rng, beta1=b1, beta2=b2, eps=eps, inv_scale=inv, found_inf=fi,
                                         skipped=self._skipped, count_skip=(k == len(launches) - 1))  # one optimiser = one flag, one counter
        return loss

    def num_skipped(self) -> int:
        """Steps that GradScaler's found_inf turned into no-ops on the device (synchronises: checkpoints and tests only)."""
        return int(self._skipped.item()) if self._skipped is not None else 0

    def _sync_step_tensors(self) -> None:
        """Skipped steps (GradScaler's found_inf, decided on the device) are counted per OPTIMISER -- one counter, as GradScaler decides per
        optimiser -- and subtracted from every parameter's host step count here and in the kernel's bias correction.  torch.optim.Adam
        subtracts per parameter (`step -= found_inf` for the parameters that took part in the call): the two agree whenever all parameters of
        the optimiser receive their gradients together, which is how every optimiser of this model is used (one per parameter group).  An
        optimiser that has stepped with only SOME of its parameters holding gradients and has skipped a step since cannot tell which
        parameters the skip belongs to: that is reported instead of written silently into a checkpoint."""
        sk = self.num_skipped()  # the host-side counts include the skipped steps; torch's step tensors do not
        if sk and getattr(self, "_partial_steps", False):
            import warnings

            warnings.warn("HipFusedAdam: steps were skipped on found_inf while only some parameters of the optimiser held gradients; the per-parameter "
                          "`step` values written now subtract the skips from every parameter (torch.optim.Adam would subtract them only from the "
                          "parameters that took part)", RuntimeWarning, stacklevel=3)
        if sk:
            for k in list(self._steps):
                self._steps[k] = max(0, self._steps[k] - sk)
            self._skipped.zero_()
        for group in self.param_groups:
            for p in group["params"]:
                self.state[p]["step"] = torch.tensor(float(self._steps.get(id(p), 0)))

    def state_dict(self):
        self._sync_step_tensors()  # the per-parameter step counts are kept as Python ints between checkpoints
        return super().state_dict()

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad(set_to_none=True) without its per-call machinery (profiler range, foreach grouping, hooks): the
        gradients live in the arena, whose zero-fill is the engine's; here the .grad references are dropped, exactly like torch's default."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors; copy them back into the arena so that the moments stay views of it.  A checkpoint
        written by torch.optim.Adam (or by the reference Trainer) has NO entry for parameters that never received a gradient -- Adam creates
        its state lazily -- e.g. the thermal twins in shared mode: those get zero moments and step 0 here, and a full entry again."""
        super().load_state_dict(state_dict)
        if self._skipped is not None:
            self._skipped.zero_()  # the loaded step counts are final
        for group in self.param_groups:
            for p in group["params"]:
                arena, off, n = self._where[id(p)]
                st = self.state.get(p)
                views = {"exp_avg": arena.exp_avg[off:off + n].view(p.shape), "exp_avg_sq": arena.exp_avg_sq[off:off + n].view(p.shape)}
                if not st or "exp_avg" not in st:
                    for v in views.values():
                        v.zero_()
                    self._steps[id(p)] = 0
                    self.state[p] = {"step": torch.tensor(0.0), **views}
                    continue
                for key, view in views.items():
                    if st[key].data_ptr() != view.data_ptr():
                        view.copy_(st[key].to(view.device, view.dtype))
                        st[key] = view
                self._steps[id(p)] = int(float(st["step"]))
                st["step"] = torch.tensor(float(self._steps[id(p)]))

def _cut_launches(work: list, max_ranges: int = 8) -> List[list]:
    """work: ranges in optimiser order, each (flag, arena, lo, hi, step, lr, (beta1, beta2, eps)).  Launches of <= max_ranges consecutive ranges
    over one arena with one (beta1, beta2, eps), cut at optimiser boundaries: an optimiser is split only when it alone exceeds a launch, so
    that at most its LAST launch has to count a skipped step (the kernel counts once per flag and launch)."""
    launches, i = [], 0
    while i < len(work):
        j = i + 1
        while j < len(work) and j - i < max_ranges and work[j][1] is work[i][1] and work[j][6] == work[i][6]:
            j += 1
        if j < len(work) and work[j][0] == work[j - 1][0] and work[j - 1][0] != work[i][0]:
            while work[j - 1][0] == work[j][0]:  # the launch would end inside an optimiser that did not start it: end it before that optimiser
                j -= 1
        launches.append(work[i:j])
        i = j
    return launches


class DeviceGradScaler:
    """torch.amp.GradScaler's state machine (torch/amp/grad_scaler.py) for the FUSED step (RenderEngine.train_step), kept on the device: the
    reference Trainer runs every iteration through `grad_scaler.scale(loss).backward()`, `optimizer_scaler_step_some`, `grad_scaler.update()` and
    steps the LR schedulers only when the scale did not drop (engine/trainer.py:470-495; `mixed_precision=True` in thermal-nerfacto's method
    config, configs/method_configs.py:260).

    What carries over to the fused step, which computes in fp32 end to end (autocast has nothing to cast in it):
      * the multiplication of the loss by the scale and the division of the gradients by it cancel exactly in fp32 (powers of two), so they are
        not performed; the scale itself evolves exactly as GradScaler's (torch._amp_update_scale_ on the device: growth after `growth_interval`
        clean steps, backoff on a non-finite gradient) and can be read back with get_scale();
      * a step whose gradients contain an inf / NaN changes NOTHING: tn_grad_nonfinite raises `found_inf` over the live gradient range and
        tn_adam_step_ranges_amp returns without touching parameters or moments, counts the skipped step on the device, and evaluates the
        bias corrections and the LR schedule at (count - skipped) -- the trainer's "do not step the scheduler" -- without any host sync.
    (The overflow threshold differs: GradScaler sees the gradients times the scale, this sees them unscaled.)"""

    def __init__(self, device, num_groups: int = 6, init_scale: float = 2.0**16, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.num_groups = num_groups
        self.scale = torch.full((1,), float(init_scale), device=device)
        self.growth_tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.found_inf = torch.zeros(num_groups, device=device)                         # one flag per optimiser group (GradScaler decides per optimiser)
        self.skipped = torch.zeros(num_groups + 1, dtype=torch.int32, device=device)   # per-group skipped steps, then the LR-schedule lag
        self._done = torch.zeros(65 * 16, dtype=torch.int32, device=device)             # block counters of the Adam launch that performs update() (TN_ADAM_DONE_WORDS)

    def fused_update_args(self):
        """what ops.adam_step_ranges_amp(scaler_update=...) needs to perform update() in the Adam launch's last block"""
        return (self.scale, self.growth_tracker, self._done, self.growth_factor, self.backoff_factor, self.growth_interval)

    @property
    def lag_index(self) -> int:
        return self.num_groups

    def begin_step(self) -> None:
        """found_inf is zero here: created so, and update() clears it again in the launch that consumed it."""

    def check(self, group: int, grads: torch.Tensor) -> None:
        ops.grad_nonfinite(grads, self.found_inf[group:group + 1])

    def check_ranges(self, arena_grads: torch.Tensor, ranges, groups) -> None:
        """every group's gradient range in one launch"""
        ops.grad_nonfinite_ranges(arena_grads, ranges, groups, self.found_inf)

    def update(self) -> None:
        ops.grad_scaler_update(self.scale, self.growth_tracker, self.found_inf, self.skipped[self.num_groups:], self.growth_factor, self.backoff_factor,
                               self.growth_interval)

    def get_scale(self) -> float:
        return float(self.scale.item())

    def num_skipped(self, group: int = 0) -> int:
        return int(self.skipped[group].item())

    def schedule_lag(self) -> int:
        return int(self.skipped[self.num_groups].item())

    def state_dict(self) -> Dict[str, object]:
        """torch.amp.GradScaler.state_dict()'s keys (what Trainer.save_checkpoint stores under "scalers", engine/trainer.py:444) plus the two
        device-side counters that GradScaler keeps implicitly in the optimisers / schedulers it did not step: per-group skipped steps and the
        LR-schedule lag."""
        sk = self.skipped.tolist()
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(self.growth_tracker.item()), "skipped": sk[:self.num_groups], "schedule_lag": sk[self.num_groups]}

    def load_state_dict(self, state: Dict[str, object]) -> None:
        """Accepts its own state_dict and a plain torch.amp.GradScaler one (no counters: a reference checkpoint's optimiser / scheduler state
        already excludes the skipped steps)."""
        if not state:
            return
        self.scale.fill_(float(state["scale"]))
        self.growth_factor, self.backoff_factor = float(state["growth_factor"]), float(state["backoff_factor"])
        self.growth_interval = int(state["growth_interval"])
        self.growth_tracker.fill_(int(state.get("_growth_tracker", 0)))
        sk = list(state.get("skipped", [0] * self.num_groups))[:self.num_groups]
        sk += [0] * (self.num_groups - len(sk))
        self.skipped.copy_(torch.tensor(sk + [int(state.get("schedule_lag", 0))], dtype=torch.int32))
        self.found_inf.zero_()


class ExponentialDecayLR:
    """ExponentialDecayScheduler without warm-up (engine/schedulers.py:109-141): lr(step) = exp(lerp(log lr_init, log lr_final, step/max_steps)).
    Same numbers as the LambdaLR the reference builds, same `last_epoch` in its state_dict, a few microseconds per step instead of ~150."""

    def __init__(self, optimizer: torch.optim.Optimizer, lr_init: float, lr_final: float, max_steps: int):
        self.optimizer, self.lr_init, self.lr_final, self.max_steps = optimizer, lr_init, lr_final, max_steps
        self.last_epoch = 0
        self._apply()

    def _apply(self) -> None:
        t = min(max(self.last_epoch / self.max_steps, 0.0), 1.0)
        lr = float(np.exp(np.log(self.lr_init) * (1 - t) + np.log(self.lr_final) * t))  # (numpy's exp / log: the LambdaLR of the reference evaluates them in numpy)
        for g in self.optimizer.param_groups:
            g["lr"] = lr

    def step(self) -> None:
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1, "base_lrs": [self.lr_init]}

    def load_state_dict(self, state) -> None:
        self.last_epoch = int(state["last_epoch"])
        self._apply()


class Optimizers:
    """engine/optimizers.py:73-210 for this model: one optimiser (+ scheduler) per parameter group, stepped together."""

    def __init__(self, param_groups: Dict[str, List[torch.nn.Parameter]], table: Optional[Dict[str, tuple]] = None, optimizer_cls=HipFusedAdam,
                 max_norm: Optional[Dict[str, float]] = None):
        """max_norm: per-group gradient-norm clip (OptimizerConfig.max_norm, engine/optimizers.py:47; None everywhere in thermal-nerfacto's
        method config, configs/method_configs.py:274-307)."""
        from .engine import OPTIMIZERS

        table = table or OPTIMIZERS
        self.optimizers, self.schedulers, self.parameters = {}, {}, {}
        self.max_norm: Dict[str, float] = dict(max_norm or {})
        for name, params in param_groups.items():
            lr, lr_final, max_steps = table[name]
            self.optimizers[name] = optimizer_cls(params, lr=lr, eps=1e-15)
            self.parameters[name] = params
            self.schedulers[name] = ExponentialDecayLR(self.optimizers[name], lr, lr_final, max_steps)

    def zero_grad_all(self) -> None:
        for o in self.optimizers.values():
            o.zero_grad()

    def zero_grad_some(self, param_groups: List[str]) -> None:
        """engine/optimizers.py:139-143"""
        for name in param_groups:
            self.optimizers[name].zero_grad()

    def _amp_state(self, device):
        """Device state of the fused GradScaler step, one entry per optimiser in dict order: found_inf (float32; cleared and raised every
        iteration), a persistent 1-element view of each entry (what GradScaler.update() collects), the skipped-step counters (int32)."""
        st = self.__dict__.get("_amp")
        if st is None or st[0].device != device:
            n = len(self.optimizers)
            found = torch.zeros(n, dtype=torch.float32, device=device)
            st = self._amp = (found, [found[i:i + 1] for i in range(n)], torch.zeros(n, dtype=torch.int32, device=device))
        return st

    def _fused_scaler_step(self, grad_scaler, names: List[str]) -> bool:
        """torch.amp.GradScaler.step() for ALL the named optimisers in three launches instead of per optimiser: one non-finite check over every
        gradient range (found_inf per optimiser: GradScaler decides optimiser by optimiser), one reciprocal of the scale, one Adam launch that
        unscales while it reads and leaves an optimiser with a non-finite gradient untouched.  GradScaler's own bookkeeping is kept as its
        step() keeps it (torch/amp/grad_scaler.py:360-470: stage = STEPPED, found_inf_per_device), so get_scale() / update() / state_dict()
        behave as the reference Trainer expects (engine/trainer.py:488-495).  Per optimiser GradScaler.step() costs ~130 us of host time
        (inspect.signature of step(), a foreach unscale pass with a dummy scale, the torch.optim step wrapper): 0.38 ms of a 1.5 ms iteration.
        Returns False -- nothing done -- whenever the general path must run (another optimiser class, hooks, max_norm, unscale_() already
        called, a GradScaler without the private fields read here)."""
        if not isinstance(grad_scaler, torch.amp.GradScaler) or not grad_scaler.is_enabled() or self.max_norm or not _FUSED_SCALER_STEP:
            return False
        opts = [self.optimizers[n] for n in names]
        if any(type(o) is not HipFusedAdam or o._optimizer_step_pre_hooks or o._optimizer_step_post_hooks for o in opts):
            return False
        try:
            from torch.amp.grad_scaler import OptState

            states, get_scale, check = grad_scaler._per_optimizer_states, grad_scaler._get_scale_async, grad_scaler._check_scale_growth_tracker
        except (ImportError, AttributeError):
            return False
        live = [(n, o) for n, o in zip(names, opts) if any(p.grad is not None for g in o.param_groups for p in g["params"])]
        if not live:
            return True
        if any(states[id(o)]["stage"] is not OptState.READY for _, o in live):
            return False
        check("step")  # (raises like GradScaler.step when scale() was never called)
        index = self.__dict__.get("_opt_index")
        if index is None or len(index) != len(self.optimizers):
            index = self._opt_index = {n: i for i, n in enumerate(self.optimizers)}
        device = next(iter(live[0][1]._where.values()))[0].params.device
        found, views, skipped = self._amp_state(device)
        # only the entries of the optimisers stepped by THIS call are cleared: the views handed to GradScaler alias this buffer, and a second call in
        # the same iteration (two disjoint group lists before grad_scaler.update()) must not wipe the flags the first one raised
        mine_idx = [index[n] for n, _ in live]
        if len(mine_idx) == len(self.optimizers):
            found.zero_()
        elif mine_idx == list(range(mine_idx[0], mine_idx[-1] + 1)):
            found[mine_idx[0]:mine_idx[-1] + 1].zero_()
        else:
            # (the index tensor is built once per group list: torch.as_tensor of a Python list is a pageable host-to-device copy, i.e. a
            # synchronisation, and this is the optimiser's hot path)
            cache = self.__dict__.setdefault("_clear_index", {})
            key = (tuple(mine_idx), str(device))
            if key not in cache:
                cache[key] = torch.as_tensor(mine_idx, device=device)
            found.index_fill_(0, cache[key], 0.0)
        work = []  # (flag, arena, lo, hi, step, lr, (beta1, beta2, eps))
        for n, o in live:
            gi = index[n]
            mine = skipped[gi:gi + 1]
            if o._skipped is None or o._skipped.data_ptr() != mine.data_ptr():
                if o._skipped is not None:
                    mine.copy_(o._skipped)  # steps it skipped through its own step() so far
                o._skipped = mine
            work += [(gi, r[0], r[1], (r[2] + 3) // 4 * 4, r[3], r[4], tuple(r[5:8])) for r in o.collect_runs()]
        # launches of <= 8 ranges over one arena with one (beta1, beta2, eps); an optimiser is never split over two launches that both count
        # a skipped step (the kernel counts once per flag and launch)
        launches = _cut_launches(work)
        for ch in launches:
            ops.grad_nonfinite_ranges(ch[0][1].grads, [(w[2], w[3]) for w in ch], [w[0] for w in ch], found)
        inv = get_scale().to(torch.float32).reciprocal().reshape(1)
        for k, ch in enumerate(launches):
            a, (b1, b2, eps) = ch[0][1], ch[0][6]
            last = k + 1 == len(launches) or launches[k + 1][0][0] != ch[-1][0]  # (an optimiser of > 8 ranges: only its last launch counts)
            ops.adam_step_ranges_amp(a.params, a.grads, a.exp_avg, a.exp_avg_sq, [(w[2], w[3], w[4], w[5]) for w in ch], beta1=b1, beta2=b2, eps=eps,
                                     inv_scale=inv, found_inf=found, flags=[w[0] for w in ch], skipped=skipped, count_skip=last)
        for n, o in live:
            st = states[id(o)]
            st["found_inf_per_device"] = {device: views[index[n]]}
            st["stage"] = OptState.STEPPED
        return True

    def optimizer_scaler_step_some(self, grad_scaler, param_groups: List[str]) -> None:
        """engine/optimizers.py:160-173: unscale + clip when the group has a max_norm, then GradScaler.step -- which skips the optimiser when an
        inf / NaN was found (on the device for HipFusedAdam, through found_inf.item() for torch.optim.Adam) -- for groups with any gradient.
        HipFusedAdam optimisers without hooks / max_norm go out together (_fused_scaler_step)."""
        if self._fused_scaler_step(grad_scaler, param_groups):
            return
        for name in param_groups:
            optimizer = self.optimizers[name]
            max_norm = self.max_norm.get(name)
            if max_norm is not None:
                grad_scaler.unscale_(optimizer)
                torch.nn.utils.clip_grad_norm_(self.parameters[name], max_norm)
            if any(any(p.grad is not None for p in g["params"]) for g in optimizer.param_groups):
                grad_scaler.step(optimizer)

    def optimizer_scaler_step_all(self, grad_scaler) -> None:
        """engine/optimizers.py:145-158"""
        self.optimizer_scaler_step_some(grad_scaler, list(self.optimizers.keys()))

    def optimizer_step_all(self, step: int = 0) -> None:
        """engine/optimizers.py:175-183.  When every optimiser is a HipFusedAdam over one arena the groups' ranges go out in ONE launch
        (same arithmetic as one step() per optimiser: Adam is element-wise; ~0.15 ms of per-optimiser Python / torch.optim wrapper time less)."""
        opts = list(self.optimizers.values())
        if opts and all(type(o) is HipFusedAdam and not o._optimizer_step_pre_hooks and not o._optimizer_step_post_hooks for o in opts) and not self.max_norm:
            runs = []
            for o in opts:
                runs += o.collect_runs()
            by_arena: Dict[int, list] = {}
            for r in runs:
                by_arena.setdefault(id(r[0]), []).append(r)
            for rs in by_arena.values():
                a = rs[0][0]
                i = 0
                while i < len(rs):  # chunks of <= 8 runs that share (beta1, beta2, eps), exactly as HipFusedAdam.step cuts them
                    j = i + 1
                    while j < len(rs) and j - i < 8 and rs[j][5:8] == rs[i][5:8]:
                        j += 1
                    ops.adam_step_ranges(a.params, a.grads, a.exp_avg, a.exp_avg_sq, [(r[1], (r[2] + 3) // 4 * 4, r[3], r[4]) for r in rs[i:j]],
                                         beta1=rs[i][5], beta2=rs[i][6], eps=rs[i][7])
                    i = j
            return
        for name, o in self.optimizers.items():
            max_norm = self.max_norm.get(name)
            if max_norm is not None:
                torch.nn.utils.clip_grad_norm_(self.parameters[name], max_norm)
            o.step()

    def scheduler_step_all(self, step: int = 0) -> None:
        for s in self.schedulers.values():
            s.step()

    def state_dict(self):
        return {"optimizers": {k: v.state_dict() for k, v in self.optimizers.items()}, "schedulers": {k: v.state_dict() for k, v in self.schedulers.items()}}

    def load_state_dict(self, state) -> None:
        for k, v in state["optimizers"].items():
            self.optimizers[k].load_state_dict(v)
        for k, v in state.get("schedulers", {}).items():
            self.schedulers[k].load_state_dict(v)
