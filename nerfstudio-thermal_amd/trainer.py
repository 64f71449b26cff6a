"""The reference Trainer on the FUSED step: `Trainer.train_iteration` (engine/trainer.py:455-499) replaced by ONE call of
`ThermalNerfactoModel.train_iteration` -- forward, every loss, backward, (gradient exchange), Adam and the GradScaler bookkeeping as the HIP
library enqueues them (`tn_train_step`), with no autograd tape and no host synchronisation -- while everything else of the Trainer (its loop and
callbacks, logging, evaluation, checkpoints, the viewer) stays the reference's own code.

    from nerfstudio.engine.trainer import Trainer, TrainerConfig
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin

    class HipTrainer(FusedTrainerMixin, Trainer):
        pass
    # TrainerConfig(_target=HipTrainer, ...): what nerfstudio_thermal_amd.plugin does for the method `thermal-nerfacto-hip`

Why a Trainer-side seam at all: the reference's iteration reads `grad_scaler.get_scale()` twice (two host synchronisations) and walks through
autograd between three library calls; on the drop-in classes that sequence runs at 1.06-1.4 ms per 4096-ray iteration where this one runs at
0.78 (profiles/r04_experiments.md).  The semantics are the Trainer's: a step whose gradients hold an inf / NaN changes nothing, the scale grows /
backs off as torch.amp.GradScaler's, the LR schedulers do not advance on such a step (optim.DeviceGradScaler: decided on the device).

The mixin only reads what the reference Trainer has (engine/trainer.py:85-140, 455-499): `pipeline` (`.model`, `.datamanager.next_train(step)`),
`optimizers` (`.optimizers`, `.schedulers`), `grad_scaler`, `mixed_precision`, `gradient_accumulation_steps`, `config.log_gradients`.  Anything it
does not cover -- gradient accumulation, `log_gradients`, a model that is not this package's -- goes to the reference's own `train_iteration`."""
from __future__ import annotations

import functools
from typing import Any, Dict, Tuple

import torch


def _model_of(trainer):
    m = trainer.pipeline.model  # (VanillaPipeline.model is already the bare module of a DistributedDataParallel wrap, pipelines/base_pipeline.py:284-288)
    return getattr(m, "module", m)


def fused_ready(trainer) -> bool:
    """True when this Trainer's iteration is one the fused step reproduces."""
    m = _model_of(trainer)
    if not (hasattr(m, "engine") and hasattr(m, "train_iteration")):
        return False
    if any(int(v) != 1 for v in getattr(trainer, "gradient_accumulation_steps", {}).values()):
        return False
    return not bool(getattr(getattr(trainer, "config", None), "log_gradients", False))


def _device_scaler(trainer):
    """The Trainer's GradScaler as optim.DeviceGradScaler (same state, kept on the device), created on first use from the Trainer's own."""
    from .optim import DeviceGradScaler

    gs = getattr(trainer, "grad_scaler", None)
    if gs is None or not gs.is_enabled():
        return None
    ds = trainer.__dict__.get("_tn_device_scaler")
    if ds is None:
        m = _model_of(trainer)
        ds = DeviceGradScaler(m.device, num_groups=len(m.arena.optimised_groups))
        ds.load_state_dict(gs.state_dict())
        trainer.__dict__["_tn_device_scaler"] = ds
    return ds


def _log_schedule(decision: Dict[str, Any], step: int) -> None:
    """the guard's decision through the Trainer's writer (utils/writer.py: put_dict goes to every configured writer) and to the console"""
    scalars = {k: float(v) for k, v in decision.items() if isinstance(v, (int, float)) and v == v}
    scalars["overlapped_schedule_runs"] = 1.0 if decision.get("schedule") == "overlapped" else 0.0
    try:
        from nerfstudio.utils import writer  # type: ignore

        writer.put_dict(name="data-parallel schedule (thermal-nerfacto-hip)", scalar_dict=scalars, step=max(int(step), 0))
    except Exception:  # noqa: BLE001  (no nerfstudio here, or no writer set up yet: the console line below is the record)
        pass
    print("nerfstudio_thermal_amd: data-parallel schedule = %s (overlapped %.3f ms, simple %.3f ms per iteration; %s)" % (
        decision.get("schedule"), decision.get("overlapped_ms", float("nan")), decision.get("simple_ms") or float("nan"), decision.get("measured_on", "")), flush=True)


def _dp_guard(trainer):
    """Data parallel (scripts/train.py:138-151 starts one process per GPU and the pipeline wraps the model in DistributedDataParallel): the fused
    step has no autograd hooks for DDP's reducer to hang on, so the same mean all-reduce is issued by this package's reducers -- overlapped with the
    backward (parallel.OverlappedGradReducer) or after it (parallel.GradAllReducer), whichever the first iterations of THIS run measure as the
    faster one (parallel.InRunScheduleGuard).  TN_DP_SCHEDULE=overlapped|simple pins the schedule instead.  None on one rank."""
    import os

    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return None
    guard = trainer.__dict__.get("_tn_dp_guard")
    if guard is None:
        from .parallel import GradAllReducer, InRunScheduleGuard, OverlappedGradReducer

        world = dist.get_world_size()
        guard = InRunScheduleGuard(world, OverlappedGradReducer(world), GradAllReducer(world), log=_log_schedule)
        pinned = os.environ.get("TN_DP_SCHEDULE", "")
        if pinned in ("overlapped", "simple"):
            guard.decision = {"schedule": pinned, "pinned_by": "TN_DP_SCHEDULE"}
        trainer.__dict__["_tn_dp_guard"] = guard
    return guard


def fused_train_iteration(trainer, step: int) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], Dict[str, Any]]:
    """engine/trainer.py:455-499 -> (loss, loss_dict, metrics_dict).  The Trainer's loop runs the model's BEFORE / AFTER_TRAIN_ITERATION callbacks
    around this call (engine/trainer.py:258-276), so the step does not run the sampler's counter callback itself (step_callback=False)."""
    m = _model_of(trainer)
    ray_bundle, batch = trainer.pipeline.datamanager.next_train(step)  # pipelines/base_pipeline.py:296-301
    pose_metrics = m.engine.pose_metrics()  # (the reference's metrics hold the pose norms of the forward, i.e. before this iteration's Adam step)
    guard = _dp_guard(trainer)
    timing = guard is not None and guard.measuring  # (N > 1: the run's first iterations time the two exchange schedules)
    if timing:
        import time

        torch.cuda.synchronize()
        t0 = time.perf_counter()
    loss_dict = m.train_iteration(ray_bundle, batch, step, grad_hook=None if guard is None else guard.hook, grad_scaler=_device_scaler(trainer),
                                  step_callback=False)
    if timing:
        torch.cuda.synchronize()
        guard.record((time.perf_counter() - t0) * 1e3, step)
    metrics_dict = m.engine.train_metrics(pose_metrics)  # + PSNR per spectrum, distortion: one small launch behind the step, no synchronisation
    loss = functools.reduce(torch.add, loss_dict.values())
    return loss, loss_dict, metrics_dict


def push_state_to_trainer(trainer) -> None:
    """Before the Trainer writes a checkpoint (engine/trainer.py:424-447 reads optimizers / schedulers / grad_scaler state_dicts): the fused step's
    state -- Adam moments and step counts in the arena, LR-schedule position, scale / growth tracker -- goes into the Trainer's objects, in their
    own state_dict layouts (RenderEngine.optimizer_state_dict), so the file is the one the reference would have written for the same history."""
    m = _model_of(trainer)
    ds = trainer.__dict__.get("_tn_device_scaler")
    sd = m.engine.optimizer_state_dict(ds)
    opts = trainer.optimizers
    for g, o in opts.optimizers.items():
        if g in sd["optimizers"]:
            o.load_state_dict(sd["optimizers"][g])
    for g, sch in getattr(opts, "schedulers", {}).items():
        if g in sd["schedulers"]:
            st = dict(sch.state_dict())
            st.update({k: v for k, v in sd["schedulers"][g].items() if k in ("last_epoch", "_step_count")})
            sch.load_state_dict(st)
    if ds is not None and "scalers" in sd:
        trainer.grad_scaler.load_state_dict({k: sd["scalers"][k] for k in ("scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker")})


def pull_state_from_trainer(trainer) -> None:
    """After the Trainer loaded a checkpoint (engine/trainer.py:386-422 fills pipeline / optimizers / schedulers / grad_scaler): the same state
    into the fused step's arena and counters (RenderEngine.load_optimizer_state_dict takes the reference's layouts as they are)."""
    m = _model_of(trainer)
    opts = trainer.optimizers
    state = {"optimizers": {g: o.state_dict() for g, o in opts.optimizers.items()},
             "schedulers": {g: s.state_dict() for g, s in getattr(opts, "schedulers", {}).items()}}
    gs = getattr(trainer, "grad_scaler", None)
    trainer.__dict__.pop("_tn_device_scaler", None)  # rebuilt from the Trainer's (just loaded) scaler
    ds = _device_scaler(trainer)
    if gs is not None and gs.is_enabled():
        state["scalers"] = gs.state_dict()
    m.engine.load_optimizer_state_dict(state, ds)


class FusedTrainerMixin:
    """Mix in AHEAD of nerfstudio's Trainer (see the module docstring).  Three overrides, each falling through to the reference's method when
    the fused step does not cover the configuration."""

    def train_iteration(self, step: int):
        if not fused_ready(self):
            return super().train_iteration(step)
        return fused_train_iteration(self, step)

    def save_checkpoint(self, step: int) -> None:
        if fused_ready(self):
            push_state_to_trainer(self)
        return super().save_checkpoint(step)

    def _load_checkpoint(self) -> None:
        super()._load_checkpoint()
        if fused_ready(self):
            pull_state_from_trainer(self)
