"""trainer.FusedTrainerMixin without a GPU: what it does NOT take over goes to the Trainer it is mixed into, untouched."""
import types

from nerfstudio_thermal_amd.trainer import FusedTrainerMixin, fused_ready


class _Trainer:
    def __init__(self, model, accumulation=1, log_gradients=False):
        self.pipeline = types.SimpleNamespace(model=model, datamanager=None)
        self.gradient_accumulation_steps = {"fields": accumulation}
        self.config = types.SimpleNamespace(log_gradients=log_gradients)
        self.calls = []

    def train_iteration(self, step):
        self.calls.append(("train_iteration", step))
        return "reference"

    def save_checkpoint(self, step):
        self.calls.append(("save_checkpoint", step))

    def _load_checkpoint(self):
        self.calls.append(("_load_checkpoint",))


class _Hip(FusedTrainerMixin, _Trainer):
    pass


def test_a_foreign_model_is_the_reference_trainers_business():
    t = _Hip(model=object())  # not this package's model: no engine, no train_iteration
    assert not fused_ready(t)
    assert t.train_iteration(7) == "reference"
    t.save_checkpoint(7)
    t._load_checkpoint()
    assert t.calls == [("train_iteration", 7), ("save_checkpoint", 7), ("_load_checkpoint",)]


def test_accumulation_and_gradient_logging_are_not_covered():
    ours = types.SimpleNamespace(engine=object(), train_iteration=lambda *a, **k: None)
    assert fused_ready(_Hip(ours))
    assert not fused_ready(_Hip(ours, accumulation=2))
    assert not fused_ready(_Hip(ours, log_gradients=True))
    wrapped = types.SimpleNamespace(module=ours)  # a DistributedDataParallel wrap exposes the model as .module
    assert fused_ready(_Hip(wrapped))
