"""trainer.FusedTrainerMixin: the reference Trainer's iteration (engine/trainer.py:455-499) replaced by the fused step, everything else of the
Trainer left alone.  nerfstudio is not importable on the GPU box, so the Trainer here is a stand-in that holds exactly what the mixin reads
(pipeline.model / pipeline.datamanager.next_train, optimizers, grad_scaler, mixed_precision, gradient_accumulation_steps, config.log_gradients)
and whose own train_iteration / save_checkpoint / _load_checkpoint are the reference's statements on this package's drop-in classes."""
import copy
import functools
import types

import pytest
import torch

from test_trainer_sequence_gpu import _setup, _snapshot, _sync, assert_same_training_state

pytestmark = pytest.mark.gpu


class _Datamanager:
    def __init__(self, rb, batch):
        self.rb, self.batch = rb, batch

    def next_train(self, step):
        return self.rb[...], self.batch


class RefTrainer:
    """engine/trainer.py, as far as an iteration and a checkpoint go."""

    def __init__(self, model, optimizers, rb, batch, accumulation=1, log_gradients=False):
        self.pipeline = types.SimpleNamespace(model=model, datamanager=_Datamanager(rb, batch))
        self.optimizers = optimizers
        self.mixed_precision = True
        self.grad_scaler = torch.amp.GradScaler("cuda", enabled=True)
        self.gradient_accumulation_steps = {g: accumulation for g in optimizers.optimizers}
        self.config = types.SimpleNamespace(log_gradients=log_gradients)
        self.callbacks = model.get_training_callbacks()
        self.file = None
        self.ref_iterations = 0

    def train_iteration(self, step):  # engine/trainer.py:455-499
        self.ref_iterations += 1
        model, opts = self.pipeline.model, self.optimizers
        opts.zero_grad_some(list(opts.optimizers.keys()))
        with torch.autocast(device_type="cuda", enabled=self.mixed_precision):
            rb, batch = self.pipeline.datamanager.next_train(step)
            out = model(rb)
            metrics_dict = model.get_metrics_dict(out, batch)
            loss_dict = model.get_loss_dict(out, batch, metrics_dict)
            loss = functools.reduce(torch.add, loss_dict.values())
        self.grad_scaler.scale(loss).backward()
        opts.optimizer_scaler_step_some(self.grad_scaler, list(opts.optimizers.keys()))
        scale = self.grad_scaler.get_scale()
        self.grad_scaler.update()
        if scale <= self.grad_scaler.get_scale():
            opts.scheduler_step_all(step)
        return loss, loss_dict, metrics_dict

    def save_checkpoint(self, step):  # engine/trainer.py:424-447
        self.file = copy.deepcopy({"step": step, "pipeline": self.pipeline.model.state_dict(),
                                   "optimizers": {k: v.state_dict() for k, v in self.optimizers.optimizers.items()},
                                   "schedulers": {k: v.state_dict() for k, v in self.optimizers.schedulers.items()},
                                   "scalers": self.grad_scaler.state_dict()})

    def _load_checkpoint(self):  # engine/trainer.py:386-422
        f = self.file
        self.pipeline.model.load_state_dict(f["pipeline"])
        for k, v in f["optimizers"].items():
            self.optimizers.optimizers[k].load_state_dict(v)
        for k, v in f["schedulers"].items():
            self.optimizers.schedulers[k].load_state_dict(v)
        self.grad_scaler.load_state_dict(f["scalers"])

    def loop(self, steps, seed_base=100):  # engine/trainer.py:258-276: callbacks around train_iteration
        from nerfstudio_thermal_amd.model import TrainingCallbackLocation as Loc

        out = None
        for step in steps:
            torch.manual_seed(seed_base + step)  # (two trainers that are compared draw the same jitter)
            self.pipeline.model.engine.__dict__.pop("_rand", None)
            for cb in self.callbacks:
                cb.run_callback_at_location(step, Loc.BEFORE_TRAIN_ITERATION)
            out = self.train_iteration(step)
            for cb in self.callbacks:
                cb.run_callback_at_location(step, Loc.AFTER_TRAIN_ITERATION)
        return out


def _trainers(golden_dir, mode, **kw):
    from nerfstudio_thermal_amd.trainer import FusedTrainerMixin

    class HipTrainer(FusedTrainerMixin, RefTrainer):
        pass

    mA, oA, rb, batch, _ = _setup(golden_dir, mode)
    mB, oB, _, _, _ = _setup(golden_dir, mode)
    return HipTrainer(mA, oA, rb, batch, **kw), RefTrainer(mB, oB, rb, batch, **kw)


@pytest.mark.parametrize("mode", ["shared", "separate"])
def test_fused_trainer_takes_the_reference_trainers_steps(golden_dir, mode):
    """13 iterations (the 11th is the first without a proposal update), each from the reference trainer's state: same losses, same metrics, same
    parameters / moments afterwards (up to the run-to-run noise of the scatter's float atomics), same sampler counters."""
    hip, ref = _trainers(golden_dir, mode)
    seen_idle = False
    for step in range(13):
        mh, mr = hip.pipeline.model, ref.pipeline.model
        # the fused trainer starts every iteration from the reference trainer's state: parameters, moments + step counts (through the checkpoint
        # route of the mixin: Trainer objects -> arena), sampler counters
        _sync(mh, hip.optimizers, mr, ref.optimizers)
        hip.grad_scaler.load_state_dict(ref.grad_scaler.state_dict())
        from nerfstudio_thermal_amd.trainer import pull_state_from_trainer

        pull_state_from_trainer(hip)
        lh, ldh, mdh = hip.loop([step])
        lr, ldr, mdr = ref.loop([step])
        assert hip.ref_iterations == 0  # every iteration went through the fused step
        assert set(ldh) == set(ldr), (sorted(ldh), sorted(ldr))
        for k in ldr:
            a, b = float(ldh[k].detach()), float(ldr[k].detach())
            assert abs(a - b) <= 2e-4 * abs(b) + 1e-9, (step, k, a, b)
        assert abs(float(lh.detach()) - float(lr.detach())) <= 2e-4 * abs(float(lr.detach()))
        for k in ("psnr_rgb", "psnr_thermal", "distortion", "camera_opt_translation", "camera_opt_rotation"):
            assert abs(float(mdh[k]) - float(mdr[k])) <= 2e-4 * abs(float(mdr[k])) + 1e-6, (step, k, float(mdh[k]), float(mdr[k]))
        assert mh.engine.last_updated == mr.engine.last_updated and mh.engine.steps_since_update == mr.engine.steps_since_update
        seen_idle = seen_idle or not mr.engine.last_updated
        assert_same_training_state(_snapshot(mh), _snapshot(mr), f"fused trainer vs reference trainer, iteration {step}")
    assert seen_idle


def test_fused_trainer_checkpoint_is_the_reference_trainers(golden_dir):
    """3 fused iterations -> Trainer.save_checkpoint -> a REFERENCE trainer loads the file and goes on for 2 iterations; against a fused trainer
    that loads the same file and goes on.  The file carries the reference's layouts (torch Adam state per parameter, LambdaLR state, GradScaler
    state) with the fused step's history in them."""
    hip, ref = _trainers(golden_dir, "shared")
    hip.loop(range(3))
    hip.save_checkpoint(3)
    f = hip.file
    assert set(f) == {"step", "pipeline", "optimizers", "schedulers", "scalers"}
    for g, sd in f["optimizers"].items():
        steps = {float(st["step"]) for st in sd["state"].values()}
        assert steps <= {3.0}, (g, steps)  # every parameter that was stepped has been stepped three times
    assert all(int(s["last_epoch"]) == 3 for s in f["schedulers"].values())
    assert float(f["scalers"]["scale"]) == 65536.0 and int(f["scalers"]["_growth_tracker"]) == 3
    # both kinds of trainer resume from it
    hip2, ref2 = _trainers(golden_dir, "shared")
    for t in (hip2, ref2):
        t.file = copy.deepcopy(f)
        t._load_checkpoint()
    for step in (3, 4):
        hip2.loop([step])
        ref2.loop([step])
        if step == 3:
            assert_same_training_state(_snapshot(hip2.pipeline.model), _snapshot(ref2.pipeline.model), "resumed fused vs resumed reference trainer")
    assert hip2.ref_iterations == 0 and ref2.ref_iterations == 2


def test_fused_trainer_leaves_uncovered_configurations_to_the_reference(golden_dir):
    """gradient accumulation / log_gradients are the reference's own iteration."""
    hip, _ = _trainers(golden_dir, "shared", log_gradients=True)
    hip.loop([0])
    assert hip.ref_iterations == 1
    hip, _ = _trainers(golden_dir, "shared", accumulation=2)
    hip.loop([0])
    assert hip.ref_iterations == 1
