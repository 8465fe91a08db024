"""Checkpoint ingestion: the published `.pth` files are `torch.save`d state-dicts, bare or under
['model'] (cwm/models/prediction.py:81-107; demo notebook cells that call `load_state_dict`, README.md:67-78).  No
published checkpoint is reachable offline, so what is pinned is the schema (key names, shapes, parameter counts:
SURVEY.md Appendix B, probed from the reference) and the loader's behaviour on a saved random state-dict."""
import os

import numpy as np
import pytest
import torch

from counterfactualworldmodels_amd import config as C, conjoined_vmae, prediction, synthetic as S, vmae

TINY = C.VmaeConfig(name="tiny_ckpt", img_size=(32, 32), patch=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=1,
                    dec_heads=2)


def _state(cfg, seed):
    return {k: torch.from_numpy(v) for k, v in S.synthetic_state_dict(cfg, seed).items()}


@pytest.mark.parametrize("wrap", [False, True])
def test_saved_state_dict_round_trip(tmp_path, wrap, capsys):
    sd = _state(TINY, 3)
    path = os.path.join(tmp_path, "tiny.pth")
    torch.save({"model": sd, "epoch": 7} if wrap else sd, path)
    net = vmae.PretrainVisionTransformer(TINY)
    G = prediction.PredictorBasedGenerator(predictor=net, predictor_load_path=path, imagenet_normalize_inputs=True)
    assert "<All keys matched successfully>" in capsys.readouterr().out   # what the reference prints (prediction.py:103)
    assert G._predictor_load_path == path
    got = net.state_dict()
    assert list(got.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(got[k], sd[k]), k


def test_published_schemas_load_strictly():
    """Every published predictor: a state-dict with exactly the reference's keys and shapes loads with strict=True, and the
    parameter counts are the published ones (demo notebook cell outputs)."""
    for name, count, ntensors in [("base_8x8patch_2frames_1tube", 92_661_312, 218), ("large_4x4patch_2frames_1tube", 340_709_936, 478)]:
        cfg = C.CONFIGS[name]
        schema = C.state_dict_schema(cfg)
        assert len(schema) == ntensors
        assert sum(int(np.prod(s)) for s in schema.values()) == count
        with torch.device("meta"):
            net = vmae.PretrainVisionTransformer(cfg)
        assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(s) for k, s in schema.items()}


def test_wrong_or_missing_keys_are_reported_like_torch():
    net = vmae.PretrainVisionTransformer(TINY)
    sd = _state(TINY, 4)
    bad = dict(sd)
    bad.pop("decoder.head.bias")
    bad["unexpected.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="decoder.head.bias"):
        net.load_state_dict(bad)
    res = net.load_state_dict(bad, strict=False)          # notebook cell for the IMU model uses strict=False
    assert res.missing_keys == ["decoder.head.bias"] and res.unexpected_keys == ["unexpected.weight"]
    wrong = dict(sd)
    wrong["encoder.patch_embed.proj.weight"] = torch.zeros(128, 3, 1, 4, 4)
    with pytest.raises(RuntimeError, match="size mismatch"):
        net.load_state_dict(wrong)


def test_conjoined_checkpoint_schema_round_trip(tmp_path):
    """IMU-conditioned model: 634 tensors / 148,265,040 parameters incl. the loaded-but-unused
    `context_stream.pos_embed_encoder.*`; saved under ['model'], loaded with strict=False as the notebook does."""
    with torch.device("meta"):
        net = conjoined_vmae.imu400_base_4x4patch_2frames_1tube()
    sd = net.state_dict()
    assert len(sd) == 634 and sum(v.numel() for v in sd.values()) == 148_265_040
    assert "context_stream.pos_embed_encoder.weight" in sd and tuple(sd["main_stream.null_token_enc"].shape) == (1, 1, 768)
    tiny = conjoined_vmae.ConjoinedPaddedVisionTransformer(conjoined_vmae.TINY_CONJ) if hasattr(conjoined_vmae, "TINY_CONJ") else None
    if tiny is not None:
        path = os.path.join(tmp_path, "conj.pth")
        torch.save({"model": tiny.state_dict()}, path)
        other = conjoined_vmae.ConjoinedPaddedVisionTransformer(conjoined_vmae.TINY_CONJ)
        G = prediction.PredictorBasedGenerator(predictor=other, imagenet_normalize_inputs=False)
        G.load_predictor(path)
        for k, v in tiny.state_dict().items():
            assert torch.equal(other.state_dict()[k], v), k
