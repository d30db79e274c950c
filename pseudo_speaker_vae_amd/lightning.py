"""PseudoSpeakerVAE: the reference's LightningModule API (/root/reference/ps_vae/lightning.py:10-214)
over the HIP hot path.

Same `**hparams` schema, attributes (`model`, `classifier`, `consistency_classifier`, `kl_loss_weight`,
`classifier_loss_weight`, `consitency_loss_weight` [sic, lightning.py:56], `use_cos_loss`, `hparams`),
logged keys and return values.  When pytorch_lightning is importable it is the base class; this image
has no Lightning (and no network), so a minimal stand-in base supplies what the reference's drivers
use: save_hyperparameters / hparams / log / device / load_from_checkpoint.

training_step / validation_step with no classifier branches run forward + ELBO as ONE fused autograd
Function on the GPU (model.ELBOStepFunction); so does the single-label LatentClassifier head
(model.ELBOClassifierStepFunction).  With a consistency head (EmbeddingClassifier on x_hat) the VAE runs through the
HIP forward/backward (model.VAEForwardFunction) and each head is its own fused HIP autograd node
(latent_classifier._ClassifierCEFunction, embedding_classifier._ConsistencyCEFunction); only head shapes those
kernels do not cover (multi-label, other activations, hidden > 128) fall to torch ops.
"""
from __future__ import annotations

import inspect
from typing import Any, Dict, Optional

import torch
import torch.nn as nn
from torch.optim import Adam

from .latent_classifier import LatentClassifier
from .model import VAEModel

try:  # pragma: no cover - not installed in this image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError here
    pl = None
    HAVE_LIGHTNING = False

    class _HParams(dict):
        """dict with attribute access: `self.hparams["optimizer"]` (lightning.py:205) and
        `vae_model.hparams.model['latent_dim']` (inference.py:22) both work."""
        __getattr__ = dict.__getitem__

    class _Base(nn.Module):
        """what the reference needs from pl.LightningModule, nothing more"""

        def __init__(self):
            super().__init__()
            self._hparams = _HParams()
            self.logged: Dict[str, Any] = {}

        def save_hyperparameters(self):
            # Lightning semantics: the caller's __init__ arguments, a **kwargs dict merged in flat
            frame = inspect.currentframe().f_back
            code, loc = frame.f_code, frame.f_locals
            nargs = code.co_argcount + code.co_kwonlyargcount
            hp = {n: loc[n] for n in code.co_varnames[1:nargs] if n in loc}
            if code.co_flags & inspect.CO_VARKEYWORDS:
                kwname = code.co_varnames[nargs + (1 if code.co_flags & inspect.CO_VARARGS else 0)]
                hp.update(loc.get(kwname, {}))
            self._hparams = _HParams(hp)

        @property
        def hparams(self):
            return self._hparams

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, name, value, **kwargs):
            self.logged[name] = value  # kept as a tensor: no host sync

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
            """Lightning .ckpt layout: {'state_dict': ..., 'hyper_parameters': ...}.
            Loaded with weights_only=True: nothing in the file is executed."""
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            hp.pop("vae_checkpoint", None)  # the full state_dict below already holds the VAE
            obj = cls(**hp)
            obj.load_state_dict(ckpt["state_dict"])
            return obj

        def save_checkpoint(self, path):
            torch.save({"state_dict": self.state_dict(), "hyper_parameters": dict(self.hparams)}, path)


def _accuracy(task="multiclass", num_classes=None):
    try:  # pragma: no cover
        from torchmetrics import Accuracy
        return Accuracy(task=task, num_classes=num_classes)
    except Exception:
        class _Acc(nn.Module):
            def forward(self, y_hat, y):
                return (y_hat.argmax(-1) == y).float().mean()
        return _Acc()


class PseudoSpeakerVAE(_Base):
    def __init__(self, **hparams):
        super().__init__()
        self.save_hyperparameters()
        self.model = VAEModel(**hparams["model"])                          # lightning.py:16

        if hparams.get("vae_checkpoint", None):                            # lightning.py:18-20
            self.load_vae_from_checkpoint(hparams["vae_checkpoint"])
        if hparams.get("freeze_vae", False):                               # lightning.py:22-25
            for p in self.model.parameters():
                p.requires_grad = False

        if "classifier" in hparams:                                        # lightning.py:27-42
            chp = hparams["classifier"]
            self.classifier = LatentClassifier(**chp)
            if isinstance(chp["num_classes"], int):
                self.multilabel = False
                self.accuracy = _accuracy(num_classes=chp["num_classes"])
            else:
                self.multilabel = True
                self.accuracy = nn.ModuleDict({k: _accuracy(num_classes=n) for k, n in chp["num_classes"].items()})
        else:
            self.classifier = None

        if "consistency_classifier_ckpt" in hparams:                       # lightning.py:44-52
            from .embedding_classifier import EmbeddingClassifier  # (it derives from _Base above)
            self.consistency_classifier = EmbeddingClassifier.load_from_checkpoint(hparams["consistency_classifier_ckpt"])
            for param in self.consistency_classifier.parameters():
                param.requires_grad = False
            self.consistency_classifier.eval()
        else:
            self.consistency_classifier = None

        self.kl_loss_weight = hparams.get("kl_loss_weight", 1.0)           # lightning.py:54-57
        self.classifier_loss_weight = hparams.get("classifier_loss_weight", 1.0)
        self.consitency_loss_weight = hparams.get("consistency_loss_weight", 1.0)
        self.use_cos_loss = hparams.get("use_cos_loss", False)

    # ---- reference API -------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> tuple:                           # lightning.py:59-61
        return self.model(x)

    def decode(self, z: torch.Tensor) -> torch.Tensor:                     # lightning.py:63-65
        return self.model.decode(z)

    def _step(self, batch, prefix: str):
        x, y = batch
        log_kw = dict(sync_dist=True) if prefix == "train" else dict(batch_size=x.size(0), sync_dist=True)
        vae_trainable = any(p.requires_grad for p in self.model.ordered_params())  # (cached list: no module walk)
        cons = self.consistency_classifier
        if cons is None and self.classifier is None and (vae_trainable or not torch.is_grad_enabled()):
            # fused: forward + ELBO on the GPU, one autograd node (lightning.py:110-124); from the third step of a batch
            # shape on, one captured HIP graph (api_graph.py).  Only the losses are needed here: outputs=False
            total, recon, kl, _, _, _ = self.model.elbo_step(x, self.kl_loss_weight, self.use_cos_loss, outputs=False)
        elif (x.is_cuda and not isinstance(y, dict) and (vae_trainable or not torch.is_grad_enabled())
              and (self.classifier is None or (not self.multilabel and self.classifier.fused_params() is not None))
              and (cons is None or cons.fused_ok())):
            # fused: forward + ELBO + the heads' cross entropies + (when training) the whole backward, one autograd node
            total, recon, kl, ce, hits, cons_loss, cons_hits, _, _, _ = self.model.elbo_heads_step(
                x, y, self.classifier, cons, self.kl_loss_weight, self.classifier_loss_weight,
                self.consitency_loss_weight, self.use_cos_loss)
            if self.classifier is not None:
                self.log(f"{prefix}_classifier_acc", hits / x.size(0), sync_dist=True)
                self.log(f"{prefix}_classifier_loss", ce, sync_dist=True)
            if cons is not None:
                self.log(f"{prefix}_consistency", cons_hits / x.size(0), sync_dist=True)
                self.log(f"{prefix}_consistency_loss", cons_loss, sync_dist=True)
        else:
            x_hat, mu, log_sigma = self(x)
            classifier_loss = 0
            if self.classifier is not None:                                # lightning.py:73-98
                fused_head = (not self.multilabel and mu.is_cuda and self.classifier.fused_params() is not None
                              and not isinstance(y, dict))
                if fused_head:  # fused HIP step: CE, accuracy and all classifier gradients in 3 launches
                    classifier_loss, hits = self.classifier.cross_entropy_step(mu, y)
                    self.log(f"{prefix}_classifier_acc", hits / x.size(0), sync_dist=True)
                    self.log(f"{prefix}_classifier_loss", classifier_loss, sync_dist=True)
                    y_hat = None
                elif (mu.is_cuda and isinstance(y, dict) == self.multilabel and self.classifier.general_spec() is not None):
                    # the general fused step (csrc/heads_general.hip): several labels (the MEAN of their cross
                    # entropies, lightning.py:85-95), tanh / sigmoid / leaky-ReLU trunks, up to 32 classes
                    fused_head = True
                    classifier_loss, hits, per_label = self.classifier.cross_entropy_heads_step(mu, y)
                    if not self.multilabel:
                        self.log(f"{prefix}_classifier_acc", hits[None] / x.size(0), sync_dist=True)
                        self.log(f"{prefix}_classifier_loss", classifier_loss, sync_dist=True)
                    else:
                        running = 0  # (the reference logs the labels' RUNNING sum under each label's key, lightning.py:93)
                        for name in per_label:
                            running = running + per_label[name]
                            self.log(f"{prefix}_classifier_acc_{name}", hits[name] / x.size(0), sync_dist=True)
                            self.log(f"{prefix}_classifier_loss_{name}", running, sync_dist=True)
                    y_hat = None
                else:
                    y_hat = self.classifier(mu)
                if fused_head:
                    pass
                elif not self.multilabel:
                    classifier_loss = nn.functional.cross_entropy(y_hat, y)
                    self.log(f"{prefix}_classifier_acc", self.accuracy(y_hat, y), sync_dist=True)
                    self.log(f"{prefix}_classifier_loss", classifier_loss, sync_dist=True)
                else:
                    for name in y_hat:
                        classifier_loss = classifier_loss + nn.functional.cross_entropy(y_hat[name], y[name])
                        self.log(f"{prefix}_classifier_acc_{name}", self.accuracy[name](y_hat[name], y[name]), sync_dist=True)
                        self.log(f"{prefix}_classifier_loss_{name}", classifier_loss, sync_dist=True)
                    classifier_loss = classifier_loss / len(y_hat)
            consistency_loss = 0
            if cons is not None:                                           # lightning.py:100-108
                # the reference takes the accuracy from self.accuracy, which only exists with a latent classifier
                # configured (lightning.py:33 vs 103); here it comes from the head's own step and needs neither
                if x_hat.is_cuda and cons.fused_ok() and not isinstance(y, dict):
                    consistency_loss, hits = cons.cross_entropy_step(x_hat, y)   # fused HIP step, 4 launches
                    consistency_acc = hits / x.size(0)
                else:
                    y_hat_consistency = cons(x_hat)
                    consistency_loss = nn.functional.cross_entropy(y_hat_consistency, y)
                    consistency_acc = (y_hat_consistency.argmax(-1) == y).float().mean()
                self.log(f"{prefix}_consistency", consistency_acc, sync_dist=True)
                self.log(f"{prefix}_consistency_loss", consistency_loss, sync_dist=True)
            # lightning.py:110-117: both ELBO terms and their gradients come from the HIP kernels (psvae_elbo_f32 /
            # psvae_elbo_grad_f32), also in this general branch: no ATen loss on the VAE math anywhere
            recon, kl = self.model.elbo_terms(x_hat, mu, log_sigma, x, self.use_cos_loss)
            total = (recon + self.kl_loss_weight * kl + self.classifier_loss_weight * classifier_loss
                     + self.consitency_loss_weight * consistency_loss)                  # lightning.py:119-124
        self.log(f"{prefix}_loss", total, **log_kw)                        # lightning.py:127-129 / 193-195
        self.log(f"{prefix}_recon_loss", recon, **log_kw)
        self.log(f"{prefix}_kl_loss", kl, **log_kw)
        return {"loss": total}

    def training_step(self, batch: tuple, batch_idx: int):                 # lightning.py:67-131
        return self._step(batch, "train")

    def validation_step(self, batch: tuple, batch_idx: int):               # lightning.py:133-197
        return self._step(batch, "val")

    def load_vae_from_checkpoint(self, checkpoint_path):                   # lightning.py:199-202
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        sd = {k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        self.model.load_state_dict(sd)

    def configure_optimizers(self):                                        # lightning.py:204-214
        # a torch.optim.Adam either way (same param_groups / state / checkpoint format): FusedAdam's step() is one HIP launch
        # over all parameter tensors and falls through to torch's own step() for anything it does not cover (optim.py)
        import os
        if os.environ.get("PSVAE_FUSED_ADAM", "1") != "0":
            from .optim import FusedAdam
            optimizer = FusedAdam(self.parameters(), **self.hparams["optimizer"])
        else:
            optimizer = Adam(self.parameters(), **self.hparams["optimizer"])
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, **self.hparams["scheduler"])
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": scheduler, "interval": "epoch", "frequency": 1}}
