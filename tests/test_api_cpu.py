"""Host-side mirror of the reference interface, checked without a GPU: names, shapes, init order,
hparams access, checkpoint round trip, label maps, flat layout."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load
from oracle import psvae_oracle as O


def test_vaemodel_state_dict_matches_reference_layout():
    from ps_vae.model import PsVAE, VAEModel
    assert PsVAE is VAEModel
    m = VAEModel(256, 64)
    sd = m.state_dict()
    assert tuple(sd.keys()) == O.PARAM_KEYS                      # model.py:14-36 names, creation order
    shapes = O.param_shapes(256, 64)
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], k
    assert sum(v.numel() for v in sd.values()) == 1_281_408
    assert [p is q for p, q in zip(m.ordered_params(), m.parameters())] == [True] * 18
    d = VAEModel()                                                # defaults of model.py:8
    assert (d.input_dim, d.latent_dim, d.normalize_decoder) == (512, 64, False)


@pytest.mark.parametrize("seed", [0, 1, 3])
def test_default_init_consumes_rng_like_the_reference(seed):
    """fixtures assert oracle.init_params == the reference's parameters bit for bit (make_golden.py);
    here the product's module must equal oracle.init_params bit for bit."""
    from ps_vae.model import VAEModel
    torch.manual_seed(seed)
    m = VAEModel(256, 32)
    ref = O.init_params(256, 32, seed)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k


def test_lightning_module_surface():
    from ps_vae.lightning import PseudoSpeakerVAE
    hp = dict(model=dict(input_dim=256, latent_dim=64, normalize_decoder=False), kl_loss_weight=0.5,
              use_cos_loss=True, optimizer=dict(lr=2e-3), scheduler=dict(T_max=7, eta_min=1e-5))
    m = PseudoSpeakerVAE(**hp)
    assert m.hparams["optimizer"] == dict(lr=2e-3) and m.hparams.model["latent_dim"] == 64
    assert (m.kl_loss_weight, m.classifier_loss_weight, m.consitency_loss_weight, m.use_cos_loss) == (0.5, 1.0, 1.0, True)
    assert m.classifier is None and m.consistency_classifier is None
    assert all(k.startswith("model.") for k in m.state_dict())
    cfg = m.configure_optimizers()                                # lightning.py:204-214
    assert isinstance(cfg["optimizer"], torch.optim.Adam)
    assert cfg["lr_scheduler"]["interval"] == "epoch" and cfg["lr_scheduler"]["frequency"] == 1
    assert isinstance(cfg["lr_scheduler"]["scheduler"], torch.optim.lr_scheduler.CosineAnnealingLR)
    with pytest.raises(Exception, match="GPU only"):
        m(torch.zeros(2, 256))                                    # no CPU fallback


def test_classifier_state_dict_keys():
    from ps_vae.latent_classifier import LatentClassifier
    c = LatentClassifier(64, 3, num_layers=3, hidden_dim=128, activation="relu")
    assert list(c.state_dict().keys()) == ["layers.0.weight", "layers.0.bias", "layers.2.weight", "layers.2.bias",
                                           "layers.4.weight", "layers.4.bias"]
    mh = LatentClassifier(256, {"label1": 10, "label2": 5}, 2, 128, "relu")  # latent_classifier.py:72-82 smoke
    out = mh(torch.randn(32, 256))
    assert out["label1"].shape == (32, 10) and out["label2"].shape == (32, 5)
    with pytest.raises(ValueError):
        LatentClassifier(4, 2, activation="gelu")


def test_checkpoint_round_trip_and_partial_vae_load(tmp_path):
    from ps_vae.lightning import PseudoSpeakerVAE
    hp = dict(model=dict(input_dim=64, latent_dim=16), classifier=dict(input_dim=16, num_classes=3, num_layers=2,
                                                                      hidden_dim=32, activation="relu"))
    torch.manual_seed(0)
    a = PseudoSpeakerVAE(**hp)
    path = os.path.join(tmp_path, "best-checkpoint.ckpt")
    a.save_checkpoint(path)
    b = PseudoSpeakerVAE.load_from_checkpoint(path)               # inference.py:134
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.equal(v, v2)
    assert b.hparams.model["latent_dim"] == 16
    torch.manual_seed(5)
    c = PseudoSpeakerVAE(model=dict(input_dim=64, latent_dim=16), vae_checkpoint=path, freeze_vae=True)  # lightning.py:18-25
    for k, v in a.model.state_dict().items():
        assert torch.equal(c.model.state_dict()[k], v)
    assert not any(p.requires_grad for p in c.model.parameters())


def test_embedding_classifier_surface_and_consistency_checkpoint(tmp_path):
    """EmbeddingClassifier (embedding_classifier.py:6-62): constructor, state_dict keys, hparams, optimizer; loaded frozen
    by PseudoSpeakerVAE from a Lightning-layout checkpoint (lightning.py:44-52); no CPU path for the fused step"""
    from ps_vae.embedding_classifier.embedding_classifier import EmbeddingClassifier
    from ps_vae.lightning import PseudoSpeakerVAE
    from pseudo_speaker_vae_amd._ffi import PsvaeError
    torch.manual_seed(3)
    head = EmbeddingClassifier(input_dim=64, num_classes=3, hidden_dim=32, optimizer_cfg=dict(lr=1e-2))
    assert [k for k in head.state_dict()] == ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias"]
    assert head.fc1.weight.shape == (32, 64) and head.fc2.weight.shape == (32, 32) and head.fc3.weight.shape == (3, 32)
    assert dict(head.hparams) == dict(input_dim=64, num_classes=3, hidden_dim=32, optimizer_cfg=dict(lr=1e-2))
    opt = head.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 1e-2
    assert head(torch.randn(5, 64)).shape == (5, 3)               # plain forward (logits) works anywhere
    with pytest.raises(PsvaeError):
        head.cross_entropy_step(torch.randn(5, 64), torch.zeros(5, dtype=torch.int64))
    path = os.path.join(tmp_path, "head.ckpt")
    head.save_checkpoint(path)
    m = PseudoSpeakerVAE(model=dict(input_dim=64, latent_dim=16), consistency_classifier_ckpt=path,
                         consistency_loss_weight=0.25)
    c = m.consistency_classifier
    assert isinstance(c, EmbeddingClassifier) and not c.training and m.consitency_loss_weight == 0.25
    assert not any(q.requires_grad for q in c.parameters())
    for k, v in head.state_dict().items():
        assert torch.equal(c.state_dict()[k], v)
    assert any(k.startswith("consistency_classifier.fc1") for k in m.state_dict())


def test_label_maps_match_reference_fixture():
    from ps_vae.utils import METADATA_TRANSFORMS, map_cv_age_to_label, map_cv_gender_to_label, map_vctk_gender_to_label
    fx = load(os.path.join(GOLDEN, "label_maps.npz"))
    for s, none, lab in zip(fx["ages"], fx["age_is_none"], fx["age_labels"]):
        assert map_cv_age_to_label(None if none else str(s)) == int(lab)
    for s, none, cv, vc in zip(fx["genders"], fx["gender_is_none"], fx["cv_gender_labels"], fx["vctk_gender_labels"]):
        key = None if none else str(s)
        assert map_cv_gender_to_label(key) == int(cv) and map_vctk_gender_to_label(key) == int(vc)
    assert METADATA_TRANSFORMS["age_and_gender"]({"age": "fifties", "gender": "female"}) == {"age": 1, "gender": 1}


def test_flat_layout_buckets():
    from pseudo_speaker_vae_amd.engine import FlatLayout
    shapes = [torch.Size(s) for s in O.param_shapes(256, 64).values()]
    lay = FlatLayout(shapes)
    flat = torch.arange(lay.total, dtype=torch.float32)
    views = lay.views(flat)
    assert [tuple(v.shape) for v in views] == [tuple(s) for s in shapes]
    assert all(o % 64 == 0 for o in lay.offsets)                  # 256-B aligned
    assert lay.offsets[11] < lay.loss_offset < lay.enc_end <= lay.offsets[12]  # encoders | loss | decoder
    # views do not overlap
    ends = [o + s.numel() for o, s in zip(lay.offsets, shapes)]
    assert all(e <= o2 for e, o2 in zip(ends[:-1], lay.offsets[1:]))


def test_conditional_synthesis_signature_is_the_reference_one():
    """inference.py:29-38: same parameter names, order and defaults (device="cpu" included); the extra keyword-only
    hooks come after them.  The default device has no CPU path here and says so."""
    import inspect
    from ps_vae.inference import conditional_synthesis, unconditional_synthesis
    sig = inspect.signature(conditional_synthesis)
    ref = [("vae_model", inspect._empty), ("num_samples", inspect._empty), ("classifier_target", inspect._empty),
           ("step_size", 0.01), ("num_steps", 100), ("noise_weight", 1.0), ("return_history", False), ("device", "cpu")]
    assert [(n, p.default) for n, p in list(sig.parameters.items())[:len(ref)]] == ref
    assert list(inspect.signature(unconditional_synthesis).parameters)[:3] == ["vae_model", "num_samples", "device"]
    with pytest.raises(RuntimeError, match="GPU only"):
        conditional_synthesis(object(), 4, 1)


def test_fused_adam_is_a_torch_adam_and_falls_through_on_cpu():
    """optim.FusedAdam (what configure_optimizers returns, lightning.py:204-205): a torch.optim.Adam subclass whose step() is
    torch's own whenever the HIP launch does not apply -- on CPU parameters it must reproduce torch.optim.Adam bit for bit,
    and its state_dict must load into a plain Adam and back"""
    import torch
    from pseudo_speaker_vae_amd.optim import FusedAdam
    torch.manual_seed(0)
    w0 = [torch.randn(7, 5), torch.randn(5)]
    pa = [torch.nn.Parameter(w.clone()) for w in w0]
    pb = [torch.nn.Parameter(w.clone()) for w in w0]
    oa = FusedAdam(pa, lr=3e-3, weight_decay=0.01)
    ob = torch.optim.Adam(pb, lr=3e-3, weight_decay=0.01)
    assert isinstance(oa, torch.optim.Adam)
    for it in range(3):
        for p, q in zip(pa, pb):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for p, q in zip(pa, pb):
        assert torch.equal(p, q)
    ob.load_state_dict(oa.state_dict())
    oa.load_state_dict(ob.state_dict())
    assert set(oa.state_dict()["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
