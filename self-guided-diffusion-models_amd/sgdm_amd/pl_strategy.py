"""PyTorch-Lightning strategy for the HIP training step (replaces ``pl.trainer.strategy=ddp`` of the reference's launch
line, README.md:84-94 / config/pl/default.yaml:2): one process per GPU and the usual rank / sampler / seed plumbing of
Lightning's DDP strategy, but the LightningModule is NOT wrapped in ``DistributedDataParallel`` -- the gradient exchange
is ``sgdm_amd.ddp``'s bucketed RCCL all-reduce inside the backward program (overlapped with the remaining launches), and
nothing is broadcast per step (EMA shadows and schedule tables are rank-deterministic).

    trainer = pl.Trainer(strategy=HipDDPStrategy(), devices=8, accelerator="gpu", ...)

pytorch_lightning is an optional dependency of this package: without it the name raises on use, nothing else is affected.
Under the UNCHANGED ``strategy=ddp`` the drop-in still trains correctly: ``train._UNetTrainFn`` detects the wrapper and
leaves the exchange to torch's reducer."""
try:
    from pytorch_lightning.strategies import DDPStrategy as _Base
except Exception:                                    # pragma: no cover - Lightning is absent in the build container
    _Base = None


if _Base is not None:
    class HipDDPStrategy(_Base):
        strategy_name = "hip_ddp"

        def configure_ddp(self):
            """no DistributedDataParallel wrapper: the HIP backward program reduces the gradients itself"""
            self.model = self.model                  # keep the bare LightningModule

        def _setup_model(self, model):
            return model
else:
    class HipDDPStrategy:                           # noqa: D101
        def __init__(self, *a, **k):
            raise ImportError("sgdm_amd.pl_strategy.HipDDPStrategy needs pytorch_lightning")
