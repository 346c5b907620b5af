"""CPU tests (run under -m "not gpu") of the real-format checkpoint paths:
  * the SAM checkpoint prefix split + freezes of build_model_with_query_support_feat(sam_checkpoint_path=...)
    (ref: lib/build_model.py:96-120), on a synthetic SAM-style .pth (keys image_encoder.* / mask_decoder.* / prompt_encoder.*);
  * an open_clip-style SigLIP state_dict .bin through siglip_checkpoint_path (ref: lib/support_model/siglip_openclip.py:12);
  * the CORE checkpoint loader of my_test.py:118-145 ("model_state_dict", "module." prefix, strict).
No GPU work: parameters only."""
import pytest
import torch

from cor_amd import utils
from cor_amd.lib.build_model import build_model_with_query_support_feat

SAM, SIG = "sam_base", "ViT-B-16-SigLIP-384"


@pytest.fixture(scope="module")
def donor():
    m = build_model_with_query_support_feat(SAM, SIG, None, None, "MaskAdapterPooling")
    utils.randomize_parameters(m, seed=5)
    return m


def test_sam_checkpoint_prefix_split_and_freezes(donor, tmp_path):
    ck = {f"image_encoder.{k}": v.clone() for k, v in donor.image_encoder.state_dict().items()}
    ck.update({f"mask_decoder.{k}": v.clone() for k, v in donor.mask_decoder.state_dict().items()})
    ck.update({f"prompt_encoder.{k}": v.clone() for k, v in donor.prompt_encoder.state_dict().items()})
    ck["prompt_encoder.point_embeddings.0.weight"] = torch.randn(1, 256)          # keys of the original SAM that CORE has no use for
    ck["prompt_encoder.mask_downscaling.0.weight"] = torch.randn(4, 1, 2, 2)
    path = str(tmp_path / "sam_vit_b_synthetic.pth")
    torch.save(ck, path)
    m = build_model_with_query_support_feat(SAM, SIG, sam_checkpoint_path=path, siglip_checkpoint_path=None, mask_pooling="MaskAdapterPooling")
    for k, v in donor.image_encoder.state_dict().items():
        assert torch.equal(m.image_encoder.state_dict()[k], v), k
    for k, v in donor.mask_decoder.state_dict().items():
        assert torch.equal(m.mask_decoder.state_dict()[k], v), k
    # the reference filters "prompt_encoder.dense_embedding*" (build_model.py:102-104): nothing of CORE's prompt encoder matches,
    # so it keeps its initial values (strict=False)
    assert not torch.equal(m.prompt_encoder.no_mask_embed.weight, donor.prompt_encoder.no_mask_embed.weight)
    # freezes (build_model.py:113-119)
    assert all(not p.requires_grad for p in m.image_encoder.parameters())
    assert all(not p.requires_grad for p in m.support_branch.siglip.parameters())
    assert all(not p.requires_grad for p in m.mask_decoder.iou_prediction_head.parameters())
    assert all(p.requires_grad for n, p in m.mask_decoder.named_parameters() if not n.startswith("iou_prediction_head."))
    assert all(p.requires_grad for n, p in m.support_branch.named_parameters() if not n.startswith("siglip."))
    # without a checkpoint nothing is frozen (the freezes sit inside the `if`, as in the reference)
    assert all(p.requires_grad for p in donor.image_encoder.parameters())


@pytest.mark.parametrize("wrapped", [False, True])
def test_openclip_style_siglip_checkpoint(donor, tmp_path, wrapped):
    sd = {k: v.clone() for k, v in donor.support_branch.siglip.model.state_dict().items()}
    assert "visual.trunk.pos_embed" in sd and "text.token_embedding.weight" in sd and "logit_scale" in sd and "logit_bias" in sd
    path = str(tmp_path / "open_clip_pytorch_model.bin")
    torch.save({"state_dict": sd} if wrapped else sd, path)
    m = build_model_with_query_support_feat(SAM, SIG, None, siglip_checkpoint_path=path, mask_pooling="MaskedPooling")
    got = m.support_branch.siglip.model.state_dict()
    assert sorted(got) == sorted(sd)
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    bad = dict(sd); bad.pop("text.ln_final.weight")
    torch.save(bad, path)
    with pytest.raises(RuntimeError):                                              # strict, like open_clip's load_checkpoint
        build_model_with_query_support_feat(SAM, SIG, None, siglip_checkpoint_path=path, mask_pooling="MaskedPooling")


@pytest.mark.parametrize("ddp_prefix", [False, True])
def test_core_checkpoint_loader(donor, tmp_path, ddp_prefix):
    from cor_amd import harness
    sd = {(f"module.{k}" if ddp_prefix else k): v.clone() for k, v in donor.state_dict().items()}
    path = str(tmp_path / "checkpoint_epoch_10.pth")
    torch.save({"model_state_dict": sd, "epoch": 10}, path)
    m = build_model_with_query_support_feat(SAM, SIG, None, None, "MaskAdapterPooling")
    f0 = m._fingerprint()
    res, epoch = harness.load_core_checkpoint(m, path)
    assert epoch == 10 and not res.missing_keys and not res.unexpected_keys
    assert m._fingerprint() != f0                                                 # the packed-weight cache notices the load
    for k, v in donor.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    sd.pop(next(iter(sd)))
    torch.save({"model_state_dict": sd}, path)
    with pytest.raises(RuntimeError):
        harness.load_core_checkpoint(m, path)
