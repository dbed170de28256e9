"""The two registry backbones against fixtures of the REFERENCE's own classes (tests/golden/make_backbone_fixtures.py):
the pairwise `croco` backbone (backbone_croco.py:61-286) and the `croco_multi` trunk with the 'linear' intrinsics embedding and with no
embedding at all (backbone_croco_multiview.py:59-78,123-145,190-227).  VERDICT r04 "what's missing" 2 / 3."""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests.helpers import deterministic_init_

F = np.load(Path(__file__).resolve().parent / "golden" / "backbone_variants.npz")
TINY = dict(enc_depth=2, dec_depth=3, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512))


def _cfg(name, loc, typ):
    from styl3r_amd.encoder import BackboneCrocoCfg
    return BackboneCrocoCfg(name=name, intrinsics_embed_loc=loc, intrinsics_embed_type=typ)


def test_backbone_registry_keys_and_unsupported_embeddings():
    from styl3r_amd.encoder import BACKBONES, AsymmetricCroCo, AsymmetricCroCoMulti, get_backbone
    assert BACKBONES["croco"] is AsymmetricCroCo and BACKBONES["croco_multi"] is AsymmetricCroCoMulti
    for tag, name, loc, typ in (("pair", "croco", "encoder", "token"), ("lin", "croco_multi", "encoder", "linear"), ("none", "croco_multi", "none", "token")):
        m = get_backbone(_cfg(name, loc, typ), 3, TINY)
        assert sorted(m.state_dict().keys()) == list(F[f"{tag}_keys"]), tag
    # combinations the reference's own forward cannot run (backbone_croco_multiview.py:217: the multi-view trunk never hands the decoder-side
    # embedding to its decoder; backbone_croco.py:99-101: a non-pixelwise decoder-side embedding does not fit decoder_embed)
    with pytest.raises(NotImplementedError):
        get_backbone(_cfg("croco_multi", "decoder", "pixelwise"), 3, TINY)
    with pytest.raises(NotImplementedError):
        get_backbone(_cfg("croco", "decoder", "token"), 3, TINY)


@pytest.mark.gpu
def test_pairwise_croco_backbone_matches_the_reference_class():
    from styl3r_amd.encoder import get_backbone
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(get_backbone(_cfg("croco", "encoder", "token"), 3, TINY).eval()).to(dev)
    T = lambda k: torch.tensor(F[k], device=dev)
    img = T("pair_image").requires_grad_(True)
    dec1, dec2, s1, s2 = m(dict(image=img, intrinsics=T("pair_K")))
    assert len(dec1) == len(dec2) == int(F["pair_n"]) and np.array_equal(s1.numpy(), F["pair_shape"]) and np.array_equal(s2.numpy(), F["pair_shape"])
    for i in range(len(dec1)):
        assert_close_rel(dec1[i].detach().cpu().numpy(), F[f"pair_d1_{i}"], 1e-4, f"dec1[{i}]")
        assert_close_rel(dec2[i].detach().cpu().numpy(), F[f"pair_d2_{i}"], 1e-4, f"dec2[{i}]")
    ((dec1[-1] * T("pair_w1")).sum() + (dec2[-1] * T("pair_w2")).sum() + (dec2[1] * T("pair_w3")).sum()).backward()
    assert_close_rel(img.grad.cpu().numpy(), F["pair_gimage"], 1e-4, "d image")
    assert_close_rel(m.dec_blocks2[1].attn.qkv.weight.grad.cpu().numpy(), F["pair_g_dec2_qkv"], 1e-4, "d dec_blocks2[1].attn.qkv")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,loc,typ", [("lin", "encoder", "linear"), ("none", "none", "token")])
def test_multiview_backbone_linear_and_no_intrinsics_embedding_match_the_reference(tag, loc, typ):
    from styl3r_amd.encoder import get_backbone
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(get_backbone(_cfg("croco_multi", loc, typ), 3, TINY).eval()).to(dev)
    T = lambda k: torch.tensor(F[f"{tag}_{k}"], device=dev)
    img = T("image").requires_grad_(True)
    feat, pos, dec_feat, shape, images = m(dict(image=img, intrinsics=T("K")))
    assert np.array_equal(pos.cpu().numpy(), F[f"{tag}_pos"])                    # no token position appended
    assert_close_rel(feat.detach().cpu().numpy(), F[f"{tag}_feat"], 1e-4, "encoder features")
    for k, t in (("d_0", dec_feat[0]), ("d_1", dec_feat[1]), ("d_last", dec_feat[-1])):
        assert_close_rel(t.detach().cpu().numpy(), F[f"{tag}_{k}"], 1e-4, k)
    ((dec_feat[-1] * T("wa")).sum() + (dec_feat[1] * T("wb")).sum()).backward()
    assert_close_rel(img.grad.cpu().numpy(), F[f"{tag}_gimage"], 1e-4, "d image")


# ------------------------------------------------------------------ pixelwise intrinsics embedding (VERDICT r04 "what's missing" 2)
P = np.load(Path(__file__).resolve().parent / "golden" / "backbone_pixelwise.npz")


def test_intrinsic_embedding_matches_the_reference_function():
    """styl3r_amd.camera.intrinsic_embedding (rays + real SH by recurrence) against `get_intrinsic_embedding` (src/geometry/camera_emb.py:7-31,
    whose SH come from generated per-degree tables, src/misc/sht.py) at degrees 0 / 4 / 8, full resolution and one row per 16 x 16 patch."""
    from styl3r_amd.camera import intrinsic_embedding
    b, v, c, h, w = (int(x) for x in P["emb_image_shape"])
    ctx = dict(image=torch.zeros(b, v, c, h, w), intrinsics=torch.tensor(P["emb_K"]))
    for deg, bar in ((0, 2e-6), (4, 5e-6), (8, 5e-5)):      # float32 on both sides; the degree-8 polynomials of the tables cancel harder
        full = intrinsic_embedding(ctx, deg).numpy()[:1, :2]
        tok = intrinsic_embedding(ctx, deg, downsample=16, merge_hw=True).numpy()
        assert full.shape == P[f"emb_full_{deg}"].shape and tok.shape == P[f"emb_tok_{deg}"].shape
        assert np.abs(full - P[f"emb_full_{deg}"]).max() <= bar, (deg, np.abs(full - P[f"emb_full_{deg}"]).max())
        assert np.abs(tok - P[f"emb_tok_{deg}"]).max() <= bar, (deg, np.abs(tok - P[f"emb_tok_{deg}"]).max())


def test_pixelwise_backbones_have_the_reference_state_dict():
    from styl3r_amd.encoder import get_backbone
    m = get_backbone(_cfg("croco_multi", "encoder", "pixelwise"), 3, TINY)
    assert sorted(m.state_dict().keys()) == list(P["pixe_keys"]) and tuple(m.patch_embed.proj.weight.shape) == tuple(P["pixe_patch_shape"])
    m = get_backbone(_cfg("croco", "decoder", "pixelwise"), 3, TINY)
    assert sorted(m.state_dict().keys()) == list(P["pixd_keys"]) and tuple(m.decoder_embed.weight.shape) == tuple(P["pixd_embed_shape"])


@pytest.mark.gpu
def test_multiview_backbone_with_the_encoder_side_pixelwise_embedding_matches_the_reference():
    from styl3r_amd.encoder import get_backbone
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(get_backbone(_cfg("croco_multi", "encoder", "pixelwise"), 3, TINY).eval()).to(dev)
    T = lambda k: torch.tensor(P[f"pixe_{k}"], device=dev)
    img = T("image").requires_grad_(True)
    feat, pos, dec_feat, shape, images = m(dict(image=img, intrinsics=T("K")))
    assert np.array_equal(pos.cpu().numpy(), P["pixe_pos"]) and images.shape[2] == 3 + 25
    assert_close_rel(images.detach().cpu().numpy()[:1, :1], P["pixe_images"], 1e-5, "images with the embedding channels")
    assert_close_rel(feat.detach().cpu().numpy(), P["pixe_feat"], 1e-4, "encoder features")
    for k, t in (("d_0", dec_feat[0]), ("d_1", dec_feat[1]), ("d_last", dec_feat[-1])):
        assert_close_rel(t.detach().cpu().numpy(), P[f"pixe_{k}"], 1e-4, k)
    ((dec_feat[-1] * T("wa")).sum() + (dec_feat[1] * T("wb")).sum()).backward()
    assert_close_rel(img.grad.cpu().numpy(), P["pixe_gimage"], 1e-4, "d image")
    assert_close_rel(m.patch_embed.proj.weight.grad.cpu().numpy()[::64], P["pixe_g_patch"], 1e-4, "d patch_embed.proj (28 input channels)")


@pytest.mark.gpu
def test_pairwise_backbone_with_the_decoder_side_pixelwise_embedding_matches_the_reference():
    from styl3r_amd.encoder import get_backbone
    from tests.gpu_utils import assert_close_rel
    dev = "cuda:0"
    m = deterministic_init_(get_backbone(_cfg("croco", "decoder", "pixelwise"), 3, TINY).eval()).to(dev)
    T = lambda k: torch.tensor(P[f"pixd_{k}"], device=dev)
    img = T("image").requires_grad_(True)
    dec1, dec2, s1, s2 = m(dict(image=img, intrinsics=T("K")))
    assert len(dec1) == len(dec2) == int(P["pixd_n"])
    for i in range(len(dec1)):
        assert_close_rel(dec1[i].detach().cpu().numpy(), P[f"pixd_d1_{i}"], 1e-4, f"dec1[{i}]")
        assert_close_rel(dec2[i].detach().cpu().numpy(), P[f"pixd_d2_{i}"], 1e-4, f"dec2[{i}]")
    ((dec1[-1] * T("w1")).sum() + (dec2[1] * T("w2")).sum()).backward()
    assert_close_rel(img.grad.cpu().numpy(), P["pixd_gimage"], 1e-4, "d image")
    assert_close_rel(m.decoder_embed.weight.grad.cpu().numpy(), P["pixd_g_embed"], 1e-4, "d decoder_embed (1024 + 25 input columns)")
