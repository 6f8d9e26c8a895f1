"""PyTorch-Lightning strategy for the HIP training step (replaces ``pl.trainer.strategy=ddp`` of the reference's launch
line, README.md:84-94 / config/pl/default.yaml:2): one process per GPU and the usual rank / sampler / seed plumbing of
Lightning's DDP strategy, but the LightningModule is NOT wrapped in ``DistributedDataParallel`` -- the gradient exchange
is ``sgdm_amd.ddp``'s bucketed RCCL all-reduce inside the backward program (overlapped with the remaining launches), and
nothing is broadcast per step (EMA shadows and schedule tables are rank-deterministic).

    trainer = pl.Trainer(strategy=HipDDPStrategy(), devices=8, accelerator="gpu", ...)

What Lightning's ``DDPStrategy`` does with the wrapper, and what this class does instead (both generations of the hook
are covered -- ``tests/test_ddp_gloo.py::test_pl_strategy_hooks`` drives them through a stand-in base class):

* Lightning 1.6-1.9 (the reference pins 1.6.3 / 1.8.0, README.md:141-145): ``setup()`` -> ``configure_ddp()`` ->
  ``self.model = self._setup_model(LightningDistributedModule(self.model))`` + ``_register_ddp_hooks()``;
* Lightning 2.x: ``setup()`` -> ``configure_ddp()`` -> ``self.model = self._setup_model(self.model)`` +
  ``_register_ddp_hooks()`` (which asserts ``isinstance(self.model, DistributedDataParallel)`` on CUDA).

Here ``configure_ddp`` keeps the bare module, registers no hooks and does what the wrapper's constructor would have done
once: broadcast rank 0's parameters AND buffers (the LitEma shadows included) to every rank
(``sgdm_amd.ddp.sync_initial_state``).  ``_setup_model`` is overridden too, for callers that reach it directly.

pytorch_lightning is an optional dependency of this package: without it the name raises on use, nothing else is affected.
Under the UNCHANGED ``strategy=ddp`` the drop-in still trains correctly: ``train._UNetTrainFn`` detects the wrapper and
leaves the exchange to torch's reducer."""


def make_strategy(base):
    """the strategy class on top of `base` (pytorch_lightning.strategies.DDPStrategy, or a stand-in in the tests)"""

    class HipDDPStrategy(base):
        strategy_name = "hip_ddp"

        def configure_ddp(self):
            """no DistributedDataParallel wrapper, no DDP comm hooks: the HIP backward program reduces the gradients
            itself.  Replicas start identical, as under the wrapper."""
            from .ddp import sync_initial_state
            self.model = self._setup_model(self.model)
            sync_initial_state(self.model)

        def _setup_model(self, model):
            return model

        def _register_ddp_hooks(self):       # nothing to hook: there is no torch reducer
            return None

    return HipDDPStrategy


try:
    from pytorch_lightning.strategies import DDPStrategy as _Base
except Exception:                                    # pragma: no cover - Lightning is absent in the build container
    _Base = None

if _Base is not None:
    HipDDPStrategy = make_strategy(_Base)
else:
    class HipDDPStrategy:                           # noqa: D101
        def __init__(self, *a, **k):
            raise ImportError("sgdm_amd.pl_strategy.HipDDPStrategy needs pytorch_lightning")
