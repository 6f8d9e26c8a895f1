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

Step dispatch: in Lightning 1.6-1.9 ``DDPStrategy.training_step`` is ``return self.model(*args, **kwargs)`` and counts on
the ``LightningDistributedModule`` wrapper to turn that ``forward`` into ``training_step`` -- with the bare module it
would run ``LightningModule.forward(batch, batch_idx)``.  The four ``*_step`` methods are therefore overridden to call
the LightningModule's own step under the precision plugin's context (what 2.x does for an unwrapped module).

RCCL's half of the CU reserve (``sgdm_amd.ddp.cap_exchange_channels``) is applied in ``setup_environment``, i.e. before
Lightning creates the process group and RCCL reads its environment.

pytorch_lightning is not installed in the build image: the hooks are driven through stand-in bases that reproduce both
generations' call sequences (``tests/test_ddp_gloo.py::test_pl_strategy_hooks``, ``::test_pl_strategy_step_dispatch``);
the class has NOT been run under a real Lightning Trainer.

pytorch_lightning is an optional dependency of this package: without it the name raises on use, nothing else is affected.
Under the UNCHANGED ``strategy=ddp`` the drop-in still trains correctly: ``train._UNetTrainFn`` detects the wrapper and
leaves the exchange to torch's reducer."""


def make_strategy(base):
    """the strategy class on top of `base` (pytorch_lightning.strategies.DDPStrategy, or a stand-in in the tests)"""
    import contextlib

    class HipDDPStrategy(base):
        strategy_name = "hip_ddp"

        def setup_environment(self):
            """before the process group exists: RCCL may take no more workgroups than the backward program leaves free"""
            from .ddp import cap_exchange_channels
            cap_exchange_channels()
            parent = getattr(super(), "setup_environment", None)
            return parent() if parent is not None else None

        def _hip_step(self, name, ctx_name, *args, **kwargs):
            module = getattr(self, "lightning_module", None) or self.model
            plugin = getattr(self, "precision_plugin", None)
            ctx = getattr(plugin, ctx_name, None) if plugin is not None else None
            with (ctx() if ctx is not None else contextlib.nullcontext()):
                return getattr(module, name)(*args, **kwargs)

        def training_step(self, *args, **kwargs):
            return self._hip_step("training_step", "train_step_context", *args, **kwargs)

        def validation_step(self, *args, **kwargs):
            return self._hip_step("validation_step", "val_step_context", *args, **kwargs)

        def test_step(self, *args, **kwargs):
            return self._hip_step("test_step", "test_step_context", *args, **kwargs)

        def predict_step(self, *args, **kwargs):
            return self._hip_step("predict_step", "predict_step_context", *args, **kwargs)

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
