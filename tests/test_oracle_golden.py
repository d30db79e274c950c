"""The CPU oracle against every golden vector generated from the reference (tests/golden/make_golden.py).

No GPU needed.  This is what 'pins' the oracle: forward, losses, all 18 parameter gradients (sampled
+ checksums), decode, the Langevin sampler and the label maps, on the reference's own outputs.
"""
import os

import numpy as np
import pytest
import torch

from oracle import psvae_oracle as O
from helpers import GOLDEN, assert_close, check_grads_against_fixture, fixture_params, load, step_fixtures

ORACLE_TOL = 2e-5  # oracle vs reference: same ATen kernels, closed-form vs autograd ordering only


@pytest.mark.parametrize("path", step_fixtures(), ids=lambda p: os.path.basename(p)[:-4])
def test_train_step_matches_reference(path):
    fx = load(path)
    p = fixture_params(fx)
    x, eps = torch.from_numpy(fx["x"]), torch.from_numpy(fx["eps"])
    losses, grads, acts, _ = O.train_step(p, x, eps, float(fx["beta"]), bool(fx["normalize"]), bool(fx["cos"]))
    assert_close(acts["x_hat"], fx["x_hat"], ORACLE_TOL, "x_hat")
    assert_close(acts["mu"], fx["mu"], ORACLE_TOL, "mu")
    assert_close(acts["logvar"], fx["logvar"], ORACLE_TOL, "logvar")
    assert_close(acts["z"], fx["z"], ORACLE_TOL, "z")
    for k in ("recon", "kl", "total"):
        assert abs(float(losses[k]) - float(fx[k])) <= ORACLE_TOL * max(1.0, abs(float(fx[k]))), k
    check_grads_against_fixture(grads, fx, ORACLE_TOL)


@pytest.mark.parametrize("name", ["decode_d256_z64_n33", "decode_d256_z64_n33_norm"])
def test_decode_matches_reference(name):
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    xh = O.decode(p, torch.from_numpy(fx["z"]), bool(fx["normalize"]))
    assert_close(xh, fx["x_hat"], ORACLE_TOL, "decode")


@pytest.mark.parametrize("name", ["langevin_d256_z64_n8_t5", "langevin_d256_z64_n8_t5_lin"])
def test_langevin_matches_reference(name):
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    layers = [(torch.from_numpy(fx[f"cls_w{i}"]), torch.from_numpy(fx[f"cls_b{i}"]))
              for i in range(int(fx["n_cls_layers"]))]
    z, hist = O.langevin(layers, torch.from_numpy(fx["z0"]), torch.from_numpy(fx["noises"]),
                         int(fx["target"]), float(fx["step_size"]), float(fx["noise_weight"]))
    assert_close(torch.stack(hist), fx["hist"], ORACLE_TOL, "langevin history")
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    assert_close(O.decode(p, z), fx["x_hat"], ORACLE_TOL, "langevin decode")


def test_label_maps_match_reference():
    fx = load(os.path.join(GOLDEN, "label_maps.npz"))
    for s, none, lab in zip(fx["ages"], fx["age_is_none"], fx["age_labels"]):
        assert O.map_cv_age_to_label(None if none else str(s)) == int(lab)
    for s, none, cv, vc in zip(fx["genders"], fx["gender_is_none"], fx["cv_gender_labels"], fx["vctk_gender_labels"]):
        key = None if none else str(s)
        assert O.map_cv_gender_to_label(key) == int(cv)
        assert O.map_vctk_gender_to_label(key) == int(vc)


def test_backward_closed_form_equals_autograd():
    """The oracle's closed-form backward vs torch autograd on the oracle's own forward."""
    D, Z, B = 96, 24, 11
    p = {k: v.clone().requires_grad_() for k, v in O.init_params(D, Z, 3).items()}
    torch.manual_seed(5)
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    for norm, cos in ((False, False), (True, False), (False, True), (True, True)):
        acts = O.forward(p, x, eps, norm)
        _, _, total = O.elbo(acts["x_hat"], x, acts["mu"], acts["logvar"], 0.7, cos)
        auto = torch.autograd.grad(total, list(p.values()))
        with torch.no_grad():
            _, grads, _, _ = O.train_step({k: v.detach() for k, v in p.items()}, x, eps, 0.7, norm, cos)
        for k, ga in zip(O.PARAM_KEYS, auto):
            assert_close(grads[k], ga, 2e-5, f"{k} norm={norm} cos={cos}")


def test_cond_gather_form_equals_concat_definition():
    """cfg3 extension (parity unpinned by the reference): gather-add form == explicit one-hot concat."""
    D, Z, B = 64, 16, 9
    torch.manual_seed(0)
    p_cat = O.init_params(D, Z, 1)
    for blk, n_in in (("encoder_mu", D), ("encoder_sigma", D), ("decoder", Z)):
        p_cat[f"{blk}.0.weight"] = torch.randn(O.HIDDEN, n_in + O.COND_DIM) * 0.05
    x, eps = torch.randn(B, D), torch.randn(B, Z)
    gender = torch.tensor([0, 1, 2, -1, 0, 1, 2, 2, 0])
    age = torch.tensor([2, 1, 0, 0, -1, 1, 2, 0, 1])
    a = O.forward_cond_concat(p_cat, x, eps, gender, age)
    p, tabs = O.split_cond_params(p_cat, D, Z)
    b = O.forward_cond(p, tabs, x, eps, gender, age)
    for k in ("mu", "logvar", "z", "x_hat"):
        assert_close(b[k], a[k], 1e-5, k)
    # gradient of the tables vs autograd through the concat definition
    pc = {k: v.clone().requires_grad_() for k, v in p_cat.items()}
    a = O.forward_cond_concat(pc, x, eps, gender, age)
    _, _, total = O.elbo(a["x_hat"], x, a["mu"], a["logvar"], 1.0)
    total.backward()
    g_xh, g_mu, g_lv = O.elbo_grads(b["x_hat"], x, b["mu"], b["logvar"], 1.0)
    grads, tgrads, _ = O.backward_cond(p, tabs, x, b, g_xh, g_mu, g_lv, gender, age)
    for blk, n_in in (("encoder_mu", D), ("encoder_sigma", D), ("decoder", Z)):
        gW = pc[f"{blk}.0.weight"].grad
        assert_close(grads[f"{blk}.0.weight"], gW[:, :n_in], 2e-5, blk)
        assert_close(tgrads[f"cond.{blk}"], gW[:, n_in:].t(), 2e-5, "table " + blk)


def test_philox_known_answer_and_moments():
    # Random123 known-answer test for philox4x32-10: ctr=key=0 and the all-ones vector
    r = O.philox4x32_10(np.zeros((1, 4), np.uint32), np.zeros(2, np.uint32))[0]
    assert [hex(v) for v in r] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    r = O.philox4x32_10(np.full((1, 4), 0xFFFFFFFF, np.uint32), np.full(2, 0xFFFFFFFF, np.uint32))[0]
    assert [hex(v) for v in r] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    n = O.philox_normal(1 << 18, seed=1234, offset=77)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1) < 0.01
    assert abs((n ** 3).mean()) < 0.03 and abs((n ** 4).mean() - 3) < 0.1
    # offset semantics: element e of stream(offset) == element e-4 of stream(offset+1)
    a = O.philox_normal(64, 9, 10)
    b = O.philox_normal(60, 9, 11)
    np.testing.assert_array_equal(a[4:], b)


def test_adam_matches_torch():
    torch.manual_seed(0)
    w = torch.randn(1000)
    ref = w.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=0.01)
    m = torch.zeros_like(w)
    v = torch.zeros_like(w)
    cur = w.clone()
    for step in range(1, 6):
        g = torch.randn(1000)
        ref.grad = g.clone()
        opt.step()
        cur, m, v = O.adam_step(cur, g, m, v, step, 1e-3, weight_decay=0.01)
        assert_close(cur, ref.detach(), 1e-6, f"adam step {step}")
    sched_p = torch.nn.Parameter(torch.zeros(1))
    o2 = torch.optim.Adam([sched_p], lr=0.01)
    s = torch.optim.lr_scheduler.CosineAnnealingLR(o2, T_max=10, eta_min=1e-4)
    for ep in range(10):
        assert abs(o2.param_groups[0]["lr"] - O.cosine_lr(0.01, ep, 10, 1e-4)) < 1e-9
        o2.step()
        s.step()


@pytest.mark.parametrize("name", ["clfstep_d256_z64_b24_h128", "clfstep_d256_z64_b19_lin"])
def test_training_step_with_classifier_head_matches_reference(name):
    """oracle.train_step_with_classifier against the reference's own training_step + backward with a LatentClassifier
    head (lightning.py:67-131): losses, accuracy, all 18 VAE gradients and the classifier's"""
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    layers = [(torch.from_numpy(fx[f"cls_w{i}"]), torch.from_numpy(fx[f"cls_b{i}"])) for i in range(int(fx["n_cls_layers"]))]
    x, y, eps = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"]), torch.from_numpy(fx["eps"])
    losses, grads, cgrads = O.train_step_with_classifier(p, layers, x, y, eps, float(fx["beta"]), float(fx["cw"]))
    for k in ("total", "recon", "kl", "ce"):
        assert abs(float(losses[k]) - float(fx[k])) <= 2e-5 * max(1.0, abs(float(fx[k]))), k
    assert abs(losses["hits"] / x.shape[0] - float(fx["acc"])) < 1e-6
    check_grads_against_fixture(grads, fx, tol=2e-5)
    for i, (gW, gb) in enumerate(cgrads):
        assert_close(gW, torch.from_numpy(fx[f"cls_gw{i}"]), 2e-5, f"classifier dW{i}")
        assert_close(gb, torch.from_numpy(fx[f"cls_gb{i}"]), 2e-5, f"classifier db{i}")


CONS_CASES = ["consstep_d256_z64_b21_h128", "consstep_d512_z128_b9_h64"]


def cons_layers(fx, prefix="cons"):
    return [(torch.from_numpy(fx[f"{prefix}_w{i}"]), torch.from_numpy(fx[f"{prefix}_b{i}"])) for i in range(3)]


@pytest.mark.parametrize("name", CONS_CASES)
def test_training_step_with_consistency_head_matches_reference(name):
    """oracle.train_step_with_heads against the reference's own training_step + backward with a LatentClassifier on mu
    AND a frozen EmbeddingClassifier on x_hat (lightning.py:44-52,100-108): the four loss terms, both accuracies, all
    18 VAE gradients and the latent classifier's"""
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    layers = [(torch.from_numpy(fx[f"cls_w{i}"]), torch.from_numpy(fx[f"cls_b{i}"])) for i in range(2)]
    x, y, eps = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"]), torch.from_numpy(fx["eps"])
    losses, grads, cgrads = O.train_step_with_heads(p, layers, cons_layers(fx), x, y, eps, float(fx["beta"]),
                                                    float(fx["cw"]), float(fx["sw"]))
    for k in ("total", "recon", "kl", "ce", "cons"):
        assert abs(float(losses[k]) - float(fx[k])) <= 2e-5 * max(1.0, abs(float(fx[k]))), k
    assert abs(losses["hits"] / x.shape[0] - float(fx["acc"])) < 1e-6
    assert abs(losses["cons_hits"] / x.shape[0] - float(fx["cons_acc"])) < 1e-6
    check_grads_against_fixture(grads, fx, tol=2e-5)
    for i, (gW, gb) in enumerate(cgrads):
        assert_close(gW, torch.from_numpy(fx[f"cls_gw{i}"]), 2e-5, f"classifier dW{i}")
        assert_close(gb, torch.from_numpy(fx[f"cls_gb{i}"]), 2e-5, f"classifier db{i}")


@pytest.mark.parametrize("name", CONS_CASES)
def test_embedding_classifier_head_alone_matches_reference(name):
    """oracle.classifier_ce on the 3-layer head against the reference EmbeddingClassifier's own forward, loss_fn and
    autograd (embedding_classifier.py:49-62,75-77): logits, loss, d loss / d input and the six parameter gradients"""
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    layers = cons_layers(fx)
    xin, y = torch.from_numpy(fx["head_x"]), torch.from_numpy(fx["y"])
    assert_close(O.classifier_forward(layers, xin), torch.from_numpy(fx["head_logits"]), 2e-5, "logits")
    loss, _, gx, pg = O.classifier_ce(layers, xin, y)
    assert abs(float(loss) - float(fx["head_loss"])) <= 2e-5
    assert_close(gx, torch.from_numpy(fx["head_gx"]), 2e-5, "d loss / d x")
    for i, (gW, gb) in enumerate(pg):
        assert_close(gW, torch.from_numpy(fx[f"head_gw{i}"]), 2e-5, f"dW{i}")
        assert_close(gb, torch.from_numpy(fx[f"head_gb{i}"]), 2e-5, f"db{i}")


# ---- the general LatentClassifier head: several labels, one-logit head, non-ReLU trunks (csrc/heads_general.hip's checker) ----
GEN_LANGEVIN = ["langevin_ml_tanh_d256_z64_n8_t5", "langevin_ml_lin_d256_z64_n8_t5", "langevin_onelogit_d256_z64_n8_t5",
                "langevin_c20_leaky_d256_z32_n8_t5", "langevin_sigmoid_d256_z64_n8_t5"]


def general_head_from_fixture(fx):
    """(trunk, all heads, names, sizes, activation) of a fixture written by make_golden._save_classifier"""
    trunk = [(torch.from_numpy(fx[f"trunk_w{i}"]), torch.from_numpy(fx[f"trunk_b{i}"])) for i in range(int(fx["n_trunk"]))]
    sizes = [int(v) for v in fx["sizes"]]
    heads = [(torch.from_numpy(fx[f"head_w{i}"]), torch.from_numpy(fx[f"head_b{i}"])) for i in range(len(sizes))]
    names = [str(v) for v in fx["names"]]
    return trunk, heads, names, sizes, str(fx["activation"])


def targeted_heads(fx):
    trunk, heads, names, sizes, act = general_head_from_fixture(fx)
    if int(fx["multi"]):
        tn = [str(v) for v in fx["target_names"]]
        pick = [names.index(k) for k in tn]
    else:
        pick = [0]
    targets = [int(v) for v in fx["target_vals"]]
    one_logit = (not int(fx["multi"])) and sizes[0] == 1
    return trunk, [heads[i] for i in pick], targets, act, (float(fx["N"]) if one_logit else 1.0)


@pytest.mark.parametrize("name", GEN_LANGEVIN)
def test_general_langevin_oracle_matches_reference_histories(name):
    """oracle.langevin_heads against the reference's own conditional_synthesis (inference.py:29-110, unmodified) driving the
    reference's LatentClassifier: dict targets over several labels (one label left out of the target), a linear multi-label
    head, the single one-logit head with its N-fold prior, 20 classes behind leaky-ReLU, a sigmoid trunk"""
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    trunk, heads, targets, act, prior_scale = targeted_heads(fx)
    z, hist = O.langevin_heads(trunk, heads, targets, act, torch.from_numpy(fx["z0"]), torch.from_numpy(fx["noises"]),
                               float(fx["step_size"]), float(fx["noise_weight"]), prior_scale)
    assert_close(torch.stack(hist), torch.from_numpy(fx["hist"]), 2e-5, "history")
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    assert_close(O.decode(p, z), torch.from_numpy(fx["x_hat"]), 2e-5, "decoded samples")


@pytest.mark.parametrize("name", ["mlstep_d256_z64_b24_tanh", "mlstep_d256_z64_b19_lin"])
def test_multilabel_training_step_oracle_matches_reference(name):
    """oracle.train_step_with_label_heads against the reference's own training_step + backward on its multi-label branch
    (lightning.py:85-95): total / recon / kl, the labels' logged running losses and accuracies, all 18 VAE gradients, the
    trunk's and every label head's"""
    fx = load(os.path.join(GOLDEN, name + ".npz"))
    p = O.init_params(int(fx["D"]), int(fx["Z"]), int(fx["pseed"]))
    trunk, heads, names, sizes, act = general_head_from_fixture(fx)
    x, eps = torch.from_numpy(fx["x"]), torch.from_numpy(fx["eps"])
    ys = [torch.from_numpy(fx[f"y_{k}"]) for k in names]
    losses, grads, tg, hg = O.train_step_with_label_heads(p, trunk, heads, x, ys, eps, float(fx["beta"]), float(fx["cw"]), act)
    for k in ("total", "recon", "kl"):
        assert abs(float(losses[k]) - float(fx[k])) <= 2e-5 * max(1.0, abs(float(fx[k]))), k
    running = 0.0
    for k, q, h in zip(names, losses["per_label"], losses["hits"]):
        running += float(q)
        assert abs(running - float(fx[f"logged_loss_{k}"])) <= 2e-5 * max(1.0, running), k   # the running sum (lightning.py:93)
        assert abs(h / x.shape[0] - float(fx[f"logged_acc_{k}"])) < 1e-6, k
    check_grads_against_fixture(grads, fx, tol=2e-5)
    for i, (gW, gb) in enumerate(tg):
        assert_close(gW, torch.from_numpy(fx[f"trunk_gw{i}"]), 2e-5, f"trunk dW{i}")
        assert_close(gb, torch.from_numpy(fx[f"trunk_gb{i}"]), 2e-5, f"trunk db{i}")
    for i, (gW, gb) in enumerate(hg):
        assert_close(gW, torch.from_numpy(fx[f"head_gw{i}"]), 2e-5, f"head {names[i]} dW")
        assert_close(gb, torch.from_numpy(fx[f"head_gb{i}"]), 2e-5, f"head {names[i]} db")
