"""CPU oracle for the self-guided-diffusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and there only as the checker (or, for the
bench leg, as the timed CPU baseline) -- never as the thing measured or shipped.
The product path (``self-guided-diffusion-models_amd/``) never imports this
package and fails loudly when its HIP library is missing.

What it is: a from-scratch, state-dict driven, functional restatement (plain
PyTorch-CPU fp32 + numpy float64 host math) of the reference's

  * ``dynamic/diffusionmodules/openaimodel.py``     UNetModel   (``unet_fast``)
  * ``dynamic/diffusionmodules/openaimodel_ca.py``  UNetModel   (``unetca_fast``)
  * ``dynamic/crossattetion_lr.py``                 Attention_LR
  * ``dynamic/diffusionmodules/util.py``            schedules, timestep embedding
  * ``diffusion/ddpm.py``                           LatentDiffusion (loss, sampler dispatch)
  * ``diffusion/sampler/ddpm_sampler.py``           Schedule_DDPM (native 1000-step sampler)
  * ``diffusion/sampler/ddim_plms_sampler.py``      DDIMSampler (ddim)
  * ``dynamic/ema.py``, ``diffusion_utils/lr_scheduler.py``

Each function cites the reference file:line it follows.

Parity pinning: the reference's own tests hold no golden vectors for this path
(``test_unittest.py`` is a crash-only launcher), so the
oracle is pinned against outputs of the reference itself: the script
``tests/golden/make_golden.py`` imports the reference from ``/root/reference``
(with throw-away stubs for absent third-party modules), runs it on seeded
inputs and commits the input/output vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this oracle against every one of them.
"""
