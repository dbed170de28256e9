"""Checkpoint ingestion with synthetic state dicts of the real key layout (the blobs are absent)."""
import torch

from styl3r_amd import checkpoint as ck
from styl3r_amd.encoder import EncoderNoPoSplatMultiTokenStyle, EncoderNoPoSplatTokenStyleCfg

TINY = dict(enc_depth=1, dec_depth=12, enc_embed_dim=1024, dec_embed_dim=128, enc_num_heads=16, dec_num_heads=2,
            pos_embed="RoPE100", img_size=(512, 512))


def _enc():
    return EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=TINY)


def test_mast3r_model_dict_is_remapped_and_conf_channel_dropped():
    enc = _enc()
    # a MASt3R-style 'model' dict: un-prefixed trunk keys, no dec_blocks2, mean heads with a 4th (confidence) channel
    src = {}
    for k, v in enc.backbone.state_dict().items():
        if not k.startswith("dec_blocks2") and not k.startswith("intrinsic_encoder"):
            src[k] = torch.randn_like(v)
    for h in ("downstream_head1", "downstream_head2"):
        for k, v in getattr(enc, h).state_dict().items():
            t = torch.randn_like(v)
            if k.startswith("dpt.head.4."):
                t = torch.randn((4, *v.shape[1:]))
            src[f"{h}.{k}"] = t
    missing, unexpected = ck.load_pretrained_encoder(enc, {"model": src})
    assert not unexpected
    assert all(m.startswith(("backbone.intrinsic_encoder", "token_stylizer", "gaussian_")) for m in missing), missing[:5]
    assert torch.equal(enc.backbone.enc_blocks[0].attn.qkv.weight, src["enc_blocks.0.attn.qkv.weight"])
    # DUSt3R-style decoder duplication (backbone_croco_multiview.py:99-106)
    assert torch.equal(enc.backbone.dec_blocks2[5].mlp.fc1.weight, src["dec_blocks.5.mlp.fc1.weight"])
    assert torch.equal(enc.downstream_head1.dpt.head[4].weight, src["downstream_head1.dpt.head.4.weight"][:3])
    assert torch.equal(enc.downstream_head1.dpt.scratch.layer_rn[2].weight, enc.downstream_head1.dpt.scratch.layer3_rn.weight)


def test_wrapper_checkpoint_roundtrip_and_stylizer_init():
    a, b = _enc(), _enc()
    wrapper = {"state_dict": {"encoder." + k: v.clone() for k, v in a.state_dict().items()}}
    wrapper["state_dict"]["losses.0.lpips.net.weight"] = torch.zeros(3)          # non-encoder entries are ignored
    ck.load_wrapper_checkpoint(b, wrapper, strict=True)
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    c = _enc()
    ck.init_token_stylizer(c, wrapper)
    assert torch.equal(c.token_stylizer.enc_blocks[0].mlp.fc2.weight, a.backbone.enc_blocks[0].mlp.fc2.weight)
    assert torch.equal(c.token_stylizer.patch_embed.proj.weight, a.backbone.patch_embed.proj.weight)


def test_noposplat_state_dict_splits_gs_head_into_structure_and_appearance():
    """a NoPoSplat checkpoint's gaussian_param_head.dpt.head.4 has raw_gs_dim = structure + 3*d_sh rows;
    the structure rows load into gaussian_param_head{,2}, the trailing 3*d_sh rows seed the appearance head
    (src/main_style.py:139-150)."""
    enc = _enc()
    d3 = 3 * enc.gaussian_adapter.d_sh
    sd = {"encoder." + k: torch.randn_like(v) for k, v in enc.state_dict().items()
          if not k.startswith(("gaussian_appearance_head", "token_stylizer"))}
    rows = enc.gaussian_param_head.dpt.head[4].weight.shape[0]
    for h in ("gaussian_param_head", "gaussian_param_head2"):
        w = enc.state_dict()[f"{h}.dpt.head.4.weight"]
        sd[f"encoder.{h}.dpt.head.4.weight"] = torch.randn(rows + d3, *w.shape[1:])
        sd[f"encoder.{h}.dpt.head.4.bias"] = torch.randn(rows + d3)
    missing, unexpected = ck.load_pretrained_encoder(enc, {"state_dict": sd})
    assert not unexpected
    assert all(m.startswith(("gaussian_appearance_head", "token_stylizer")) for m in missing), missing[:5]
    src_w = sd["encoder.gaussian_param_head.dpt.head.4.weight"]
    assert torch.equal(enc.gaussian_param_head.dpt.head[4].weight, src_w[:rows])
    assert torch.equal(enc.gaussian_param_head2.dpt.head[4].weight, sd["encoder.gaussian_param_head2.dpt.head.4.weight"][:rows])
    assert torch.equal(enc.gaussian_appearance_head.dpt.head[4].weight, src_w[-d3:])
    assert torch.equal(enc.gaussian_appearance_head.dpt.head[4].bias, sd["encoder.gaussian_param_head.dpt.head.4.bias"][-d3:])
    # the rest of the appearance head is seeded from the gs head's trunk where the shapes agree
    assert torch.equal(enc.gaussian_appearance_head.dpt.head[0].weight, sd["encoder.gaussian_param_head.dpt.head.0.weight"])


def test_noposplat_state_dict_seeds_the_two_view_encoders_structure_and_appearance_heads():
    """EncoderNoPoSplatTokenStyle has `gaussian_structure_head` / `gaussian_appearance_head` instead of
    gaussian_param_head{,2}; the reference seeds exactly these two from NoPoSplat's gaussian_param_head: rows [:-3 d_sh] of
    dpt.head.4 -> structure head (src/main_style.py:144-146), rows [-3 d_sh:] -> appearance head (:148-150)."""
    from styl3r_amd.encoder import EncoderNoPoSplatTokenStyle
    enc = EncoderNoPoSplatTokenStyle(EncoderNoPoSplatTokenStyleCfg(name="noposplat_token_style"), trunk_params=TINY)
    d3 = 3 * enc.gaussian_adapter.d_sh
    g = torch.Generator().manual_seed(3)
    # a NoPoSplat checkpoint: backbone + downstream heads + ONE gs head family whose head.4 has raw_gs_dim rows
    sd = {"encoder." + k: torch.randn(v.shape, generator=g) for k, v in enc.state_dict().items() if k.startswith(("backbone.", "downstream_head1."))}
    rows = enc.gaussian_structure_head.dpt.head[4].weight.shape[0]
    for k, v in enc.gaussian_structure_head.state_dict().items():
        shape = (rows + d3, *v.shape[1:]) if k.startswith("dpt.head.4.") else v.shape
        sd["encoder.gaussian_param_head." + k] = torch.randn(shape, generator=g)
    before_sb = enc.structure_builder.state_dict()
    before_sb = {k: v.clone() for k, v in before_sb.items()}
    missing, unexpected = ck.load_pretrained_encoder(enc, {"state_dict": sd})
    assert all(u.startswith("gaussian_param_head.") for u in unexpected)            # this encoder has no module of that name
    w4 = sd["encoder.gaussian_param_head.dpt.head.4.weight"]; b4 = sd["encoder.gaussian_param_head.dpt.head.4.bias"]
    assert torch.equal(enc.gaussian_structure_head.dpt.head[4].weight, w4[:-d3]) and torch.equal(enc.gaussian_structure_head.dpt.head[4].bias, b4[:-d3])
    assert torch.equal(enc.gaussian_appearance_head.dpt.head[4].weight, w4[-d3:]) and torch.equal(enc.gaussian_appearance_head.dpt.head[4].bias, b4[-d3:])
    for mod in (enc.gaussian_structure_head, enc.gaussian_appearance_head):          # the shared trunk of both heads is seeded too
        assert torch.equal(mod.dpt.head[0].weight, sd["encoder.gaussian_param_head.dpt.head.0.weight"])
        assert torch.equal(mod.dpt.scratch.refinenet1.out_conv.weight, sd["encoder.gaussian_param_head.dpt.scratch.refinenet1.out_conv.weight"])
    assert all(torch.equal(v, enc.structure_builder.state_dict()[k]) for k, v in before_sb.items())     # untouched, as in the reference


def test_checkpoint_filter_matches_the_reference_function():
    """`convert_mast3r_state_dict` against outputs of the reference's own `checkpoint_filter_fn` (weight_modify.py:144-197) on a
    synthetic MASt3R 'model' dict with 8 x 8 patches (-> `resample_patch_embed`), a 4-channel mean head (confidence dropped)
    and no second decoder; plus the adapters' known answers.  Fixture: tests/golden/make_loss_fixtures.py."""
    from pathlib import Path
    import numpy as np
    G = np.load(Path(__file__).resolve().parent / "golden" / "checkpoint_ref.npz")
    tiny = dict(enc_depth=1, dec_depth=12, enc_embed_dim=128, dec_embed_dim=64, enc_num_heads=2, dec_num_heads=1,
                pos_embed="RoPE100", img_size=(512, 512))
    enc = EncoderNoPoSplatMultiTokenStyle(EncoderNoPoSplatTokenStyleCfg(), trunk_params=tiny)
    # rebuild the generator's source dict from its seed (same call order => same values); the stored inputs double-check it
    gg = torch.Generator().manual_seed(int(G["seed"]))
    src = {}
    for k, t in enc.backbone.state_dict().items():
        if k.startswith(("dec_blocks2", "intrinsic_encoder")):
            continue
        src[k] = torch.randn(t.shape, generator=gg)
    src["patch_embed.proj.weight"] = torch.randn(128, 3, 8, 8, generator=gg)
    src["decoder_embed.weight"] = torch.randn(64, 128, generator=gg)
    for h in ("downstream_head1", "downstream_head2"):
        for k, t in getattr(enc, h).state_dict().items():
            src[f"{h}.{k}"] = torch.randn((4, *t.shape[1:]) if k.startswith("dpt.head.4.") else t.shape, generator=gg)
    for k in G.files:
        if k.startswith("src:"):
            assert np.array_equal(src[k[4:]].numpy(), G[k]), k
    out = ck.convert_mast3r_state_dict(src, enc)
    ref_keys = set(G["out_keys"].tolist())
    assert ref_keys <= set(out.keys())
    assert all(k.startswith("backbone.dec_blocks2") for k in set(out.keys()) - ref_keys)   # the only addition: the decoder duplication
    for k in G.files:
        if k.startswith("out:"):
            got, want = out[k[4:]].numpy(), G[k]
            assert got.shape == want.shape, k
            assert np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-12), k
    got = ck.resample_patch_embed(torch.tensor(G["rpe_in"]), (12, 20)).numpy()
    assert np.abs(got - G["rpe_out_12x20"]).max() <= 2e-5 * np.abs(G["rpe_out_12x20"]).max()
    assert np.allclose(ck._adapt_input_conv(1, torch.tensor(G["aic_in"])).numpy(), G["aic_out_1"], atol=1e-6)
    assert np.allclose(ck._adapt_input_conv(7, torch.tensor(G["aic_in"])).numpy(), G["aic_out_7"], atol=1e-6)
    assert np.allclose(ck._adapt_linear(torch.tensor(G["al_in"])).numpy(), G["al_out"], atol=1e-6)
    missing, unexpected = ck.load_pretrained_encoder(enc, {"model": src})
    assert not unexpected
